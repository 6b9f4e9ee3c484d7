"""The C-ABI library builds for gfx950 here (no GPU), loads, exports every symbol that
include/mpcbatch.h declares, and refuses loudly to run without a device."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from robotic_mpc_amd import build, engine

    build.build()
    return engine.load_library()


def test_header_symbols_are_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "mpcbatch.h")).read()
    body = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(mpcb_[a-z_]+)\s*\(", body)))
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"include/mpcbatch.h declares {n} but libmpcbatch.so does not export it"


def test_version_and_sizes(lib):
    from robotic_mpc_amd import engine, packing

    assert lib.mpcb_version() == 100
    pb = engine.MpcbProblem(256, 100, 600, 1, 100, 50, 0, 0)
    ws = lib.mpcb_workspace_bytes(C.byref(pb))
    assert 256 * 101 * 600 * 8 < ws < 256 * 101 * 1200 * 8  # a few hundred doubles per stage
    per_sim = lib.mpcb_result_bytes_per_sim(C.byref(pb))
    assert per_sim == (39 * 601 + 6 * 600) * 8 + 3 * 600 * 4
    assert packing.NPARAM == 64
    bad = engine.MpcbProblem(0, 100, 600, 1, 100, 50, 0, 0)
    assert lib.mpcb_workspace_bytes(C.byref(bad)) == 0


def test_struct_layouts_match_header():
    from robotic_mpc_amd import engine

    assert C.sizeof(engine.MpcbProblem) == 32
    assert C.sizeof(engine.MpcbResult) == 11 * 8


@pytest.mark.skipif(__import__("conftest").has_gpu(), reason="checks the no-device behaviour")
def test_no_device_fails_loudly(lib):
    from robotic_mpc_amd import engine

    assert lib.mpcb_device_count() == 0
    h = C.c_void_p()
    assert lib.mpcb_create(C.byref(h), 0) == -2  # MPCB_ENODEV
    assert not h.value
    with pytest.raises(engine.EngineError, match="no CPU fallback"):
        engine.MpcBatchEngine(0)
    assert lib.mpcb_last_error(None) == b"invalid handle"


def test_missing_library_fails_loudly(tmp_path):
    from robotic_mpc_amd import engine

    with pytest.raises(engine.EngineError, match="no CPU fallback"):
        engine.load_library(str(tmp_path / "libmpcbatch.so"))


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under robotic-mpc_amd/ may reference it."""
    pkg = os.path.join(ROOT, "robotic-mpc_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                txt = open(os.path.join(dp, f)).read()
                assert "from oracle" not in txt and "import oracle" not in txt and "libmpc_oracle" not in txt, f
                assert "mpc_oracle.h" not in txt, f


def test_param_packing_layout():
    from robotic_mpc_amd import config, packing

    cfg = config.resolve_config(config.base_params())
    p = packing.pack_params(cfg)
    assert p.shape == (64,) and p[0] == 0.01 and p[2] == 1e-8 and p[1] == 1e-6
    np.testing.assert_array_equal(p[8:14], [200.0] * 6)
    np.testing.assert_array_equal(p[14:20], cfg["q0"])
    np.testing.assert_array_equal(p[38:44], cfg["umin"])
    np.testing.assert_array_equal(p[50:56], [-0.15, 0.15, -0.01, 0.01, 0.01, 0.0])
    np.testing.assert_array_equal(p[56:61], [50.0] * 5)
