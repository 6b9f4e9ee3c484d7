"""Randomised parity sweep: seeded random configurations (weights, bandwidths, bounds, references, surfaces, initial
states, plant integrators, solver options) over short horizons, on BOTH engines, against the CPU oracle.  Same
tolerances as tests/test_gpu_parity.py: 1e-9 absolute on the closed-loop logs, integer outputs identical."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ATOL = 1e-9
SEED = 20261004


def _random_cfg(rng, N, solver):
    from robotic_mpc_amd import config

    b = config.BASE_PARAMS
    scale = rng.uniform(0.35, 1.0)                           # tight velocity bounds: some become active
    opts = {"nlp_solver_type": solver}
    if rng.random() < 0.3:
        opts["levenberg_marquardt"] = float(10.0 ** rng.uniform(-4, -2))
    if solver == "SQP" and rng.random() < 0.5:
        opts["nlp_solver_tol_stat"] = float(10.0 ** rng.uniform(-7, -4))
    return config.resolve_config(config.base_params(
        prediction_horizon=N, simulation_time=0.25,
        q_0=b["q_0"] + rng.uniform(-0.3, 0.3, 6), qdot_0=rng.uniform(-1, 1, 6) * np.array([1, 1, 1, 0.5, 0.5, 0.5]),
        wcv=rng.uniform(60.0, 260.0, 6), w_u=float(10.0 ** rng.uniform(-3, -1)), w_qddot=float(rng.uniform(0.005, 0.08)),
        px_ref=float(rng.uniform(0.3, 0.55)), vy_ref=float(rng.uniform(-0.08, 0.08)),
        qdot_min=b["qdot_min"] * scale, qdot_max=b["qdot_max"] * scale,
        surface_coeffs=dict(a=rng.uniform(-0.15, 0.15), b=rng.uniform(-0.15, 0.15), c=rng.uniform(-0.03, 0.03),
                            d=rng.uniform(-0.03, 0.03), e=rng.uniform(-0.03, 0.03), f=rng.uniform(-0.05, 0.05)),
        integration_method=str(rng.choice(["Euler", "RK2", "RK3", "RK4"])), solver_options=opts))


@pytest.mark.parametrize("engine_name", ["latency", "stream"])
def test_randomised_configurations_match_oracle(engine_name, monkeypatch, orc, ur10, ur10_rb):
    from robotic_mpc_amd import engine

    monkeypatch.setenv("MPCB_ENGINE", engine_name)
    eng = engine.MpcBatchEngine(0)
    rng = np.random.default_rng(SEED)
    worst, n_active, n_checked = 0.0, 0, 0
    try:
        for N in (4, 9, 17, 26):
            for solver in ("SQP_RTI", "SQP"):
                cfgs = [_random_cfg(rng, N, solver) for _ in range(6)]
                # one launch per (N, solver, options) bucket: solver options are batch-uniform
                for c in cfgs:
                    out = eng.run([c], ur10)
                    assert eng.launch_info()["engine"] == (1 if engine_name == "stream" else 0)
                    ref = orc.run(ur10_rb, orc.make_params(c))
                    for k in ("z", "u", "ee_pose", "ee_rpy", "ee_vel"):
                        np.testing.assert_allclose(out[k][0], ref[k], atol=ATOL, rtol=0, err_msg=f"{k} N={N} {solver}")
                        worst = max(worst, float(np.abs(out[k][0] - ref[k]).max()))
                    np.testing.assert_allclose(out["cost"][0], ref["cost"], atol=1e-9, rtol=1e-9)
                    for k in ("status", "sqp_iter", "qp_iter"):
                        np.testing.assert_array_equal(out[k][0], ref[k], err_msg=f"{k} N={N} {solver}")
                    n_active += int(np.any(np.abs(out["u"][0][:, 1:]) >= c["umax"][:, None] - 1e-9))
                    n_checked += 1
    finally:
        eng.close()
    print(f"{engine_name}: {n_checked} random configurations, max |gpu - oracle| = {worst:.2e}, {n_active} with an active velocity bound")
    assert n_checked == 48 and n_active >= 5
