import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure; built on demand with gcc)."""
    from oracle import orc as _orc

    _orc.build()
    return _orc


@pytest.fixture(scope="session")
def ur10():
    from robotic_mpc_amd import robots

    return robots.builtin_chain("ur10")


@pytest.fixture(scope="session")
def ur10_rb(orc, ur10):
    return orc.make_robot(ur10)


def has_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False
