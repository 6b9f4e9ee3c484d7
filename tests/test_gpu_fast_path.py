"""The bound-inactive fast path of the QP solve (csrc/mpc_ipm.h; config key qp_fast_path, parameter record slot [66]) on the GPU:
both engines against the oracle with the switch in both positions, on against off over BASELINE configs[1]'s 600 closed-loop
steps, and the attempt / back-off state across split launches.  Replaces nothing but the NUMBER of Riccati factorisations behind
solver.solve() (simulator.py:212): qp_iter is not a reference output (simulator.py:217-221)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu

ENGINES = [("latency", {}), ("stream", {"MPCB_ENGINE": "stream"})]


def _run(cfgs, chain, step_chunk=0):
    from robotic_mpc_amd import engine

    e = engine.MpcBatchEngine(0)
    out = e.run(cfgs, chain, step_chunk=step_chunk)
    info = e.launch_info()
    e.close()
    return out, info


def _configs1(n, **so):
    import bench

    cfgs = bench.workload_configs(n, 100, 6.0, seed=0, solver="SQP_RTI")
    return cfgs


@pytest.mark.parametrize("name,env", ENGINES)
def test_fast_path_on_against_off_over_configs1(orc, ur10, ur10_rb, monkeypatch, name, env):
    """32 of BASELINE configs[1]'s simulations (N=100, 600 steps, SQP_RTI, flat surface, seeded q_0 jitter), fast path on / off.
    * qp_tol = 1e-12: on == off to 1e-9 on q, qdot, u over all 600 steps (the fast path returns the QP's solution);
    * qp_tol = 1e-8 (the reference's, trajectory_optimizer.py:63): within 5e-6 (median over configs[1]'s 256 simulations 2.4e-7, largest
      1.6e-6, on the oracle) wherever no interior-point QP of the OFF run stopped at qp_solver_iter_max.  That distance is the
      interior-point iterates' own: a QP with weakly active bounds (lam ~ t ~ sqrt(tol)) is solved to ~sqrt(qp_tol) only
      (tests/test_oracle.py::test_qp_warm_start_reaches_same_solution), and the start-up transient of these runs rides the input bounds;
    * identical status; >= 90 % of the steps take ONE factorisation; HIP == oracle (1e-9, identical qp_iter) either way."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    res = {}
    for tol, itmax in ((1e-8, 50), (1e-12, 200)):
        for fast in (1, 0):
            cfgs = _configs1(32)
            for c in cfgs:
                c["qp_fast_path"] = fast; c["qp_tol"] = tol; c["qp_iter_max"] = itmax
            out, info = _run(cfgs, ur10)
            assert info["engine"] == (1 if name == "stream" else 0), info
            res[(tol, fast)] = (cfgs, out)
    on, off = res[(1e-12, 1)][1], res[(1e-12, 0)][1]
    np.testing.assert_array_equal(on["status"], off["status"])
    for k in ("z", "u"):
        np.testing.assert_allclose(on[k], off[k], atol=1e-9, rtol=0, err_msg=f"qp_tol 1e-12 {k}")
    on, off = res[(1e-8, 1)][1], res[(1e-8, 0)][1]
    np.testing.assert_array_equal(on["status"], off["status"])
    worst = 0.0
    for i in range(32):
        capped = np.nonzero(off["qp_iter"][i] >= 50)[0]
        n = int(capped[0]) if capped.size else 600
        for k in ("z", "u"):
            d = float(np.abs(on[k][i][:, :n + 1] - off[k][i][:, :n + 1]).max())
            worst = max(worst, d)
            assert d <= 5e-6, (i, k, d)
    print(f"{name}: fast path on vs off at qp_tol 1e-8: max deviation {worst:.2e}; "
          f"factorisations per step {on['qp_iter'].mean():.3f} vs {off['qp_iter'].mean():.3f}")
    assert (on["qp_iter"] == 1).mean() >= 0.9 and (off["qp_iter"] >= 2).all()
    for fast in (1, 0):
        cfgs, out = res[(1e-8, fast)]
        for i in (0, 17):
            ref = orc.run(ur10_rb, orc.make_params(cfgs[i]))
            for k in ("status", "sqp_iter", "qp_iter"):
                np.testing.assert_array_equal(out[k][i], ref[k], err_msg=f"fast {fast} sim {i} {k}")
            for k in ("z", "u", "ee_pose", "errors") if "errors" in ref else ("z", "u", "ee_pose"):
                np.testing.assert_allclose(out[k][i], ref[k], atol=1e-9, rtol=0, err_msg=f"fast {fast} sim {i} {k}")
            np.testing.assert_allclose(out["residuals"][i], ref["residuals"], atol=1e-7)


@pytest.mark.parametrize("name,env", ENGINES)
@pytest.mark.parametrize("solver,N,T", [("SQP_RTI", 15, 0.6), ("SQP", 12, 0.3), ("SQP_RTI", 130, 0.08)])
def test_fast_path_rejections_and_backoff_match_the_oracle(orc, ur10, ur10_rb, monkeypatch, name, env, solver, N, T):
    """Input bounds tight enough to stay active (every attempt rejected, attempts thinning out to one QP in nine) next to a
    simulation whose bounds let go after the first steps (rejections, then acceptances) -- in ONE launch cut into pieces of seven
    steps, so that the attempt / back-off counters travel through the workspace: equal to the oracle step by step, and equal to
    the single-launch run bit for bit."""
    from robotic_mpc_amd import config

    for k, v in env.items():
        monkeypatch.setenv(k, v)
    so = {"nlp_solver_type": solver, "qp_solver_iter_max": 120 if N > 100 else 50}
    kw = dict(prediction_horizon=N, simulation_time=T, solver_options=so)
    cfgs = [config.resolve_config(config.base_params(qdot_min=np.full(6, -0.8), qdot_max=np.full(6, 0.8), qdot_0=np.array([0.5, 0.7, 0.5, 0, 0, 0.0]), **kw)),
            config.resolve_config(config.base_params(qdot_min=np.full(6, -1.4), qdot_max=np.full(6, 1.4), qdot_0=np.array([0.9, 1.2, 0.8, 0, 0, 0.0]), **kw)),
            config.resolve_config(config.base_params(**kw))]
    whole, _ = _run(cfgs, ur10)
    pieces, _ = _run(cfgs, ur10, step_chunk=7)
    for k in ("z", "u", "status", "sqp_iter", "qp_iter", "cost"):
        np.testing.assert_array_equal(whole[k], pieces[k], err_msg=k)
    tried_any = False
    for i, c in enumerate(cfgs):
        ref = orc.run(ur10_rb, orc.make_params(c))
        off = orc.run(ur10_rb, orc.make_params({**c, "qp_fast_path": 0}))
        bad = np.nonzero((ref["status"] != 0) | (whole["status"][i] != 0) | (ref["qp_iter"] >= c["qp_iter_max"]))[0]
        n = int(bad[0]) if bad.size else ref["status"].shape[0]
        assert n >= (5 if solver == "SQP" else 0.5 * ref["status"].shape[0]), (i, n)    # (full SQP against bounds it cannot leave: flagged early)
        for k in ("status", "sqp_iter", "qp_iter"):
            np.testing.assert_array_equal(whole[k][i][:n], ref[k][:n], err_msg=f"sim {i} {k}")
        for k in ("z", "u"):
            np.testing.assert_allclose(whole[k][i][:, :n + 1], ref[k][:, :n + 1], atol=1e-9, rtol=0, err_msg=f"sim {i} {k}")
        tried_any |= bool((ref["qp_iter"] != off["qp_iter"]).any())
    assert tried_any
