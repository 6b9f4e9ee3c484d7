"""Host emulation of the product's device code (tests/emu) against the oracle.

The HIP engine (robotic_mpc_amd/csrc/mpc_core.h) is a single-source template; here it is
instantiated with an executor that runs the 64 lanes of each phase sequentially on the CPU.
This catches lane-mapping / phase-hazard / algebra bugs without a GPU.  It is test
infrastructure: the product library has no CPU path.
"""
import os
import shutil
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc needed to build the emulation harness")


def _cfg(**kw):
    from robotic_mpc_amd import config

    return config.resolve_config(config.base_params(**kw))


# pool = LDS chunk pool in doubles: 2048 forces chunks of 3-7 stages (many chunk seams and halos),
# 0 = the default 128 KiB pool (one or two chunks)
# waves = wavefronts cooperating on one simulation (64 lanes each)
@pytest.mark.parametrize("N,T,solver,chunk,pool,waves", [
    (20, 0.6, "SQP_RTI", 0, 0, 1), (20, 0.3, "SQP", 0, 0, 4), (1, 0.1, "SQP_RTI", 0, 0, 4), (2, 0.1, "SQP", 0, 2048, 2),
    (65, 0.2, "SQP_RTI", 7, 0, 4), (130, 0.05, "SQP_RTI", 0, 0, 4), (23, 0.3, "SQP_RTI", 0, 2048, 4),
    (11, 0.2, "SQP", 3, 2048, 1), (100, 0.05, "SQP_RTI", 0, 4096, 2), (30, 0.1, "SQP", 0, 0, 8), (5, 0.05, "SQP_RTI", 0, 4096, 4), (2, 0.05, "SQP", 0, 2048, 1),
    # horizons beyond the LDS-resident limit with the whole pool: SEGMENT-resident sweeps (2 and 3 segments, a segment of exactly SEG_T)
    (224, 0.04, "SQP_RTI", 0, 0, 8), (300, 0.03, "SQP_RTI", 0, 0, 4), (126, 0.04, "SQP", 0, 0, 4),
    # half a CU's pool (two simulations per CU on the device), four wavefronts: one and two segments of the register-resident sweeps
    (100, 0.05, "SQP_RTI", 0, 9156, 4), (130, 0.03, "SQP", 0, 9156, 4), (7, 0.05, "SQP", 0, 9156, 4),
    # a horizon whose y (6 doubles per stage) does not fit the smallest pool: the residual pass takes its Y | S,D phases in two stage
    # ranges (401 x 6 > 2048); interior-point loop (qp_fast_path off) and fast path
    (400, 0.02, "SQP_RTI", 0, 2048, 4), (400, 0.02, "SQP_RTI_IPM", 0, 2048, 4)])
def test_emulated_engine_matches_oracle(orc, ur10, ur10_rb, N, T, solver, chunk, pool, waves):
    import emu

    ipm_only = solver.endswith("_IPM")
    cfg = _cfg(prediction_horizon=N, simulation_time=T, solver_options={"nlp_solver_type": solver.replace("_IPM", "")}, qp_fast_path=not ipm_only)
    ref = orc.run(ur10_rb, orc.make_params(cfg))
    out = emu.run([cfg], ur10, step_chunk=chunk, pool_doubles=pool, waves=waves)
    for k in ("z", "u", "ee_pose", "ee_rpy", "ee_vel"):
        np.testing.assert_allclose(out[k][0], ref[k], atol=1e-11, rtol=0, err_msg=k)
    np.testing.assert_allclose(out["cost"][0], ref["cost"], atol=1e-10, rtol=1e-10)
    np.testing.assert_array_equal(out["status"][0], ref["status"])
    np.testing.assert_array_equal(out["sqp_iter"][0], ref["sqp_iter"])
    np.testing.assert_array_equal(out["qp_iter"][0], ref["qp_iter"])


@pytest.mark.parametrize("N,T,pool,waves", [(15, 0.3, 0, 4), (15, 0.3, 9156, 4), (120, 0.06, 9156, 4), (120, 0.06, 0, 8), (130, 0.05, 0, 4)])
def test_emulated_engine_active_bounds_and_batch(orc, ur10, ur10_rb, N, T, pool, waves):
    """Tight input bounds (active from the first step) and non-default per-instance parameters -- through the streaming sweeps
    (short horizon), the register-resident sweeps (half a pool; N = 120: two segments), the LDS-resident factor (N = 120, whole
    pool) and the LDS segments (N = 130)."""
    import emu

    cfgs = [
        _cfg(prediction_horizon=N, simulation_time=T, qdot_min=np.full(6, -0.8), qdot_max=np.full(6, 0.8),
             qdot_0=np.array([0.5, 0.7, 0.5, 0, 0, 0.0])),
        _cfg(prediction_horizon=N, simulation_time=T, wcv=np.array([150., 180., 200., 120., 90., 60.]), w_u=0.001,
             w_qddot=0.05, px_ref=0.5, vy_ref=-0.02, surface_coeffs=dict(a=-0.1, b=0.12, c=0.0, d=0.02, e=-0.01, f=0.05)),
    ]
    out = emu.run(cfgs, ur10, pool_doubles=pool, waves=waves)
    # At N = 120 the first QP of the tight-bound simulation stops at qp_solver_iter_max (50 iterations, status 0 under SQP_RTI): an
    # interior-point iterate that has NOT converged is reproduced to 1e-9 .. 1e-10 only -- by every sweep implementation alike
    # (streaming sweeps 1.2e-9 / 3e-10 depending on the chunking, register sweeps 1.1e-9, resident factor 5e-11): 1e-8 there.
    atol = 1e-9 if N < 100 else 1e-8
    for i, cfg in enumerate(cfgs):
        ref = orc.run(ur10_rb, orc.make_params(cfg))
        np.testing.assert_allclose(out["z"][i], ref["z"], atol=atol, rtol=0)
        np.testing.assert_allclose(out["u"][i], ref["u"], atol=atol, rtol=0)
        np.testing.assert_array_equal(out["status"][i], ref["status"])
        np.testing.assert_array_equal(out["qp_iter"][i], ref["qp_iter"])
    assert np.abs(out["u"][0][:, 1:]).max() > 0.8 - 1e-6  # the bound is really hit


@pytest.mark.parametrize("N,T,solver,pool,waves", [(30, 0.2, "SQP_RTI", 0, 4), (30, 0.1, "SQP", 2048, 2), (100, 0.08, "SQP_RTI", 0, 8),
                                                    (100, 0.08, "SQP_RTI", 9156, 4), (140, 0.05, "SQP_RTI", 0, 8)])
def test_emulated_random_parameter_records(orc, ur10, ur10_rb, N, T, solver, pool, waves):
    """Every field of the parameter record drawn at random (the draws of tests/test_gpu_parity.py
    test_random_parameter_records_match_oracle, six simulations each) through the host emulation of the device code: streaming,
    LDS-resident, register-resident and segment sweeps against the oracle, strict up to the first flagged step."""
    import emu
    import helpers as hp

    so = {"nlp_solver_type": solver, "qp_solver_iter_max": 50 if N < 100 else 120}
    cfgs = hp.random_parameter_cfgs(12, seed=1000 + N + len(solver), prediction_horizon=N, simulation_time=T, solver_options=so)[:6]
    out = emu.run(cfgs, ur10, pool_doubles=pool, waves=waves)
    for i, c in enumerate(cfgs):
        ref = orc.run(ur10_rb, orc.make_params(c))
        bad = np.nonzero((ref["status"] != 0) | (out["status"][i] != 0))[0]
        n = int(bad[0]) if bad.size else ref["status"].shape[0]
        for k in ("status", "sqp_iter", "qp_iter"):
            np.testing.assert_array_equal(out[k][i][:n], ref[k][:n], err_msg=f"sim {i} {k}")
        for k in ("z", "u", "ee_pose"):
            np.testing.assert_allclose(out[k][i][:, :n + 1], ref[k][:, :n + 1], atol=1e-9, rtol=0, err_msg=f"sim {i} {k}")


@pytest.mark.parametrize("method", ["Euler", "RK2", "RK3"])
def test_emulated_plant_integrators(orc, ur10, ur10_rb, method):
    """simulation_model.py:39-49: the plant integrator is a per-simulation parameter."""
    import emu

    cfg = _cfg(prediction_horizon=8, simulation_time=0.1, integration_method=method)
    ref = orc.run(ur10_rb, orc.make_params(cfg))
    out = emu.run([cfg], ur10, waves=4)
    np.testing.assert_allclose(out["z"][0], ref["z"], atol=1e-11, rtol=0)
    rk4 = orc.run(ur10_rb, orc.make_params(_cfg(prediction_horizon=8, simulation_time=0.1)))
    assert np.abs(rk4["z"] - ref["z"]).max() > 1e-6   # the option does change the closed loop


def test_emulated_folded_passes_equal_separate_passes(ur10, tmp_path):
    """Round 4: SQP_RTI folds the fast path's commit and right-hand-side item passes into the NLP pass (mpc_core.h nlp_direct
    fuse_commit / want_rhs, build switch MPCB_FUSE).  The folded build must reproduce the build with separate passes BIT FOR BIT --
    one rhs_item function serves both places a right-hand side is formed, the commit's arithmetic is the same additions -- on a run with
    accepted, rejected and suspended fast-path attempts (tight input bounds at the start), in one launch and cut into launches, with
    the whole pool (G2 records leave LDS in one store) and with a small one (8-byte stores from the items)."""
    import subprocess

    import emu

    cfgs = [_cfg(prediction_horizon=30, simulation_time=0.4, qdot_max=np.full(6, 0.6), qdot_min=-np.full(6, 0.6),
                 q_0=np.array([0.9, -1.2, 1.1, 0.2, 0.4, 0.1])),
            _cfg(prediction_horizon=30, simulation_time=0.4)]
    code = ("import sys, numpy as np; sys.path.insert(0, %r); sys.path.insert(0, %r); import emu, pickle; "
            "cfgs, chain, kw, dst = pickle.load(open(sys.argv[1], 'rb')); o = emu.run(cfgs, chain, **kw); np.savez(dst, **o)"
            % (os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"), os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import pickle

    for kw in (dict(waves=4), dict(waves=4, step_chunk=7), dict(waves=4, pool_doubles=2048)):
        out = emu.run(cfgs, ur10, **kw)
        assert (out["qp_iter"] == 1).any() and (out["qp_iter"] > 1).any()       # both branches of the QP solve were taken
        job, dst = str(tmp_path / "job.pkl"), str(tmp_path / "sep.npz")
        pickle.dump((cfgs, ur10, kw, dst), open(job, "wb"))
        env = dict(os.environ, MPC_EMU_DEFINES="-DMPCB_FUSE=0")
        subprocess.check_call([sys.executable, "-c", code, job], env=env)
        sep = np.load(dst)
        for k in ("z", "u", "ee_pose", "ee_vel", "errors", "cost", "residuals", "status", "sqp_iter", "qp_iter"):
            np.testing.assert_array_equal(out[k], sep[k], err_msg=f"{kw} {k}")


def test_joint_sincos_against_libm():
    """csrc/mpc_kin.h sincos_joint (Cody-Waite by pi/2 in two fused steps + the fdlibm kernels; replaces the library sincos, whose
    large-argument path cost the 256-register builds their spills) against numpy's libm: <= 1 ulp of 1 in absolute terms over the
    joint range, near the multiples of pi/2 where the reduction cancels, and out to 1e6 rad."""
    import ctypes as C

    import emu

    lib = C.CDLL(emu.build())
    dp = C.POINTER(C.c_double)
    rng = np.random.default_rng(0)
    k = np.arange(-40, 41)[:, None] * (np.pi / 2)
    grids = [np.linspace(-7.0, 7.0, 400001), rng.uniform(-50, 50, 200000), (k + np.linspace(-1e-6, 1e-6, 2001)[None, :]).ravel(),
             (k + rng.uniform(-1e-12, 1e-12, (81, 200))).ravel(), rng.uniform(-1e6, 1e6, 200000), np.array([0.0, -0.0, 1e-300, -1e-8])]
    for th in grids:
        th = np.ascontiguousarray(th, dtype=np.float64)
        sn, cs = np.empty_like(th), np.empty_like(th)
        lib.emu_sincos(C.c_int(th.size), th.ctypes.data_as(dp), sn.ctypes.data_as(dp), cs.ctypes.data_as(dp))
        assert np.abs(sn - np.sin(th)).max() <= 2.3e-16 and np.abs(cs - np.cos(th)).max() <= 2.3e-16
        assert np.abs(sn * sn + cs * cs - 1.0).max() <= 5e-16
