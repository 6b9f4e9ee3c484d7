"""The reference's OWN example regime on the engine (main.py:13-33, examples/simple_sim.ipynb cell 1): dt = 0.0005, 8 s = 16 000
closed-loop steps, N = 200, full SQP, wcv = [228.9 ... 1547.76], q_0 / limits of main.py, UR5 and UR10.  Every other test runs
dt = 0.01 and at most 600 steps; here a12 / a22 / b1 / b2 are far from wcv dt = 2, the 8-column log flush runs over 16 001
columns and the work queue over 1 600 chunks.  (main.py omits w_qddot -- a stale signature, SURVEY fact 0.6 --: 0.02 as in BASE_PARAMS.)"""
import os
import sys

import numpy as np
import pytest

import helpers as hp

pytestmark = pytest.mark.gpu
WCV = np.array([228.9, 262.09, 517.3, 747.44, 429.9, 1547.76])


def ref_cfg(robot, steps, N=200, dt=0.0005, **kw):
    from robotic_mpc_amd import config

    base = dict(robot_name=robot, dt=dt, simulation_time=(steps + 0.5) * dt, surface_limits=((-1, -0.1), (-1, 1)), surface_origin=np.zeros(3),
                surface_orientation_rpy=np.zeros(3), qdot_0=np.zeros(6), q_0=np.array([np.pi / 3, -np.pi / 3, np.pi / 4, -np.pi / 2, -np.pi / 2, 0.0]),
                wcv=WCV.copy(), q_min=np.array([-2 * np.pi, -2 * np.pi, -np.pi, -2 * np.pi, -2 * np.pi, -2 * np.pi]),
                q_max=np.array([2 * np.pi, 2 * np.pi, np.pi, 2 * np.pi, 2 * np.pi, 2 * np.pi]), qdot_min=np.full(6, -np.pi), qdot_max=np.full(6, np.pi),
                scene=False, prediction_horizon=N, w_qddot=0.02, solver_options={"nlp_solver_type": "SQP", "qp_solver": "PARTIAL_CONDENSING_HPIPM"})
    base.update(kw)
    return config.resolve_config(base)


_oracle_cache = {}


def _oracle(orc, robot, cfg):
    from robotic_mpc_amd import robots

    if robot not in _oracle_cache:
        _oracle_cache[robot] = orc.run(orc.make_robot(robots.builtin_chain(robot), cfg["t_ee"]), orc.make_params(cfg))
    return _oracle_cache[robot]


@pytest.mark.parametrize("engine_name,env", [("latency", {}), ("stream", {"MPCB_ENGINE": "stream"})])
@pytest.mark.parametrize("robot", ["ur5", "ur10"])
def test_reference_regime_2000_steps_against_the_oracle(orc, monkeypatch, robot, engine_name, env):
    """2 000 steps of the regime (1 s) on both engines against the oracle: 1e-9 on q, qdot, u, poses; identical status, sqp_iter,
    qp_iter at every step (none is flagged)."""
    from robotic_mpc_amd import engine, robots

    for k, v in env.items():
        monkeypatch.setenv(k, v)
    chain = robots.builtin_chain(robot)
    cfg = ref_cfg(robot, 2000)
    assert cfg["Nsim"] == 2000 and cfg["solver_type"] == 0
    ref = _oracle(orc, robot, cfg)
    e = engine.MpcBatchEngine(0)
    out = e.run([cfg, ref_cfg(robot, 2000, px_ref=0.45)], chain)
    assert e.launch_info()["engine"] == (1 if engine_name == "stream" else 0)
    e.close()
    assert (ref["status"] == 0).all()
    for k in ("status", "sqp_iter", "qp_iter"):
        np.testing.assert_array_equal(out[k][0], ref[k], err_msg=k)
    for k in ("z", "u", "ee_pose", "ee_vel"):
        np.testing.assert_allclose(out[k][0], ref[k], atol=1e-9, rtol=0, err_msg=k)
    np.testing.assert_allclose(out["cost"][0], ref["cost"], atol=1e-10, rtol=1e-9)
    assert np.abs(out["z"][0] - out["z"][1]).max() > 1e-4          # the second simulation really is another problem


def _check_properties(out, cfgs, chain, steps):
    z, u = out["z"], out["u"]
    assert np.isfinite(z).all() and (out["status"] == 0).all()
    assert (np.abs(u[:, :, 1:]) <= np.pi + 1e-9).all()
    # plant recurrence z[:, i+1] = RK4(z[:, i], u[:, i+1]) with per-joint bandwidths (simulation_model.py:79-83,111-117)
    w, dt = WCV[None, :, None], cfgs[0]["dt"]
    q, v, uu = z[:, :6, :-1], z[:, 6:, :-1], u[:, :, 1:]
    f = lambda vv: -w * vv + w * uu
    k1 = f(v); v2 = v + 0.5 * dt * k1; k2 = f(v2); v3 = v + 0.5 * dt * k2; k3 = f(v3); v4 = v + dt * k3; k4 = f(v4)
    np.testing.assert_allclose(z[:, 6:, 1:], v + dt / 6 * k1 + dt / 3 * k2 + dt / 3 * k3 + dt / 6 * k4, atol=1e-12)
    np.testing.assert_allclose(z[:, :6, 1:], q + dt / 6 * v + dt / 3 * v2 + dt / 3 * v3 + dt / 6 * v4, atol=1e-12)
    # every column of the log was written (the 8-column flush over steps + 1 columns) and is the FK of its joint angles
    for i in range(z.shape[0]):
        for t in (0, 1, 7, 8, 9, steps // 2 + 3, steps - 9, steps - 1, steps):
            T, _, _ = hp.fk_homogeneous(chain, z[i, :6, t])
            np.testing.assert_allclose(out["ee_pose"][i, :3, t], T[:3, 3], atol=1e-12)
            np.testing.assert_allclose(out["ee_pose"][i, 3:, t].reshape(3, 3), T[:3, :3], atol=1e-12)
    assert (np.abs(np.diff(z[:, :6, :], axis=2)).max(axis=(1, 2)) > 0).all()
    # resources/cost.png / g1.png (figures of this regime; their weights are not recoverable, and with a 0.1 s horizon -- N = 200 steps of
    # 0.5 ms -- the approach is slow): the cost falls by more than 5x within the 8 s, |e1| (distance to the surface) by more than half
    c = out["cost"]
    assert (c[:, 0] > 0.5).all() and (c[:, -1] < 0.2 * c[:, 0]).all(), (c[:, 0], c[:, -1])
    e1 = out["errors"][:, 0, :]
    assert (np.abs(e1[:, 0]) > 0.1).all() and (np.abs(e1[:, -1]) < 0.5 * np.abs(e1[:, 0])).all(), (e1[:, 0], e1[:, -1])


def test_reference_regime_16000_steps_latency_engine():
    """The whole 8 s of the regime, UR5 (the notebook's robot) with two surfaces, on the latency engine: properties."""
    from robotic_mpc_amd import engine, robots

    chain = robots.builtin_chain("ur5")
    cfgs = [ref_cfg("ur5", 16000), ref_cfg("ur5", 16000, surface_coeffs=dict(a=-0.1, b=0.1, c=-0.01, d=0.01, e=0.01, f=0.0))]
    e = engine.MpcBatchEngine(0)
    out = e.run(cfgs, chain)
    assert e.launch_info()["engine"] == 0
    again = e.run(cfgs, chain)
    e.close()
    for k in ("z", "u", "qp_iter"):
        assert np.array_equal(out[k], again[k]), k                  # deterministic
    _check_properties(out, cfgs, chain, 16000)
    print(f"16 000 steps, N=200, full SQP, latency engine: QPs per step {out['sqp_iter'].mean():.2f}, factorisations per QP "
          f"{out['qp_iter'].sum() / max(out['sqp_iter'].sum(), 1):.2f}, final cost {out['cost'][:, -1]}")


def test_reference_regime_16000_steps_through_the_work_queue(monkeypatch):
    """The same 8 s on the throughput engine's work-queue launch (four simulations, two slots: 1 600 chunks of ten steps per
    simulation, every hand-off through the workspace): properties, and equal to the latency engine's run over the first
    2 000 steps to 1e-9."""
    from robotic_mpc_amd import engine, robots

    chain = robots.builtin_chain("ur5")
    cfgs = [ref_cfg("ur5", 16000, px_ref=p) for p in (0.40, 0.42, 0.44, 0.46)]
    monkeypatch.setenv("MPCB_ENGINE", "stream"); monkeypatch.setenv("MPCB_STREAM_SLOTS", "2"); monkeypatch.setenv("MPCB_STREAM_CHUNK", "10")
    e = engine.MpcBatchEngine(0)
    out = e.run(cfgs, chain)
    assert e.launch_info()["engine"] == 1
    e.close()
    _check_properties(out, cfgs, chain, 16000)
    monkeypatch.delenv("MPCB_ENGINE"); monkeypatch.delenv("MPCB_STREAM_SLOTS"); monkeypatch.delenv("MPCB_STREAM_CHUNK")
    e = engine.MpcBatchEngine(0)
    lat = e.run([ref_cfg("ur5", 2000, px_ref=p) for p in (0.40, 0.46)], chain)
    e.close()
    for j, i in enumerate((0, 3)):
        np.testing.assert_array_equal(out["qp_iter"][i][:2000], lat["qp_iter"][j])
        np.testing.assert_allclose(out["z"][i][:, :2001], lat["z"][j], atol=1e-9, rtol=0)
