"""Oracle (and emulated engine) against the committed golden vectors (tests/golden)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden as mg  # noqa: E402


@pytest.mark.parametrize("name", sorted(mg.CASES))
def test_oracle_reproduces_golden(orc, ur10_rb, name):
    g = np.load(os.path.join(HERE, "golden", f"{name}.npz"))
    o = orc.run(ur10_rb, orc.make_params(mg.case_config(name)))
    for k in ("z", "u", "ee_pose", "ee_rpy", "ee_vel"):
        np.testing.assert_allclose(o[k], g[k], atol=1e-12, rtol=0, err_msg=k)  # libm differences only
    for k in ("status", "sqp_iter", "qp_iter"):
        np.testing.assert_array_equal(o[k], g[k])


@pytest.mark.parametrize("robot", ["ur10", "ur5"])
def test_oracle_fk_against_independent_kinematic_vectors(orc, robot):
    from robotic_mpc_amd import robots

    g = np.load(os.path.join(HERE, "golden", f"kin_{robot}.npz"))
    rb = orc.make_robot(robots.builtin_chain(robot))
    for q, p, R in zip(g["q"], g["p"], g["R"]):
        pose = orc.fk(rb, q)
        np.testing.assert_allclose(pose[:3], p, atol=1e-13)
        np.testing.assert_allclose(pose[3:].reshape(3, 3), R, atol=1e-13)
