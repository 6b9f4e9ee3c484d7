#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz.

The reference (acados + CasADi + Pinocchio) cannot run offline and holds no golden vectors
(SURVEY.md 8c), so these fixtures are outputs of the CPU ORACLE (oracle/mpc_oracle.c), which is
itself pinned by the independent checks in tests/test_oracle.py.  They (a) freeze the oracle
against regressions and (b) let the GPU box compare the HIP engine with fixed vectors.
Kinematic known-answer vectors come from an independent numpy homogeneous-transform chain.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

_SURF0 = dict(a=0.0, b=0.0, c=0.0, d=0.0, e=0.0, f=0.0)
CASES = {
    # name: (overrides of BASE_PARAMS).  The first four pin the HPIPM-style interior-point path (every QP through the loop:
    # qp_fast_path=False, the reference's solver.solve() semantics); the *_fp cases pin the bound-inactive fast path (the default).
    "rti_n20": dict(prediction_horizon=20, simulation_time=0.5, solver_options={"nlp_solver_type": "SQP_RTI"}, qp_fast_path=False),
    "sqp_n10": dict(prediction_horizon=10, simulation_time=0.3, solver_options={"nlp_solver_type": "SQP"}, qp_fast_path=False),
    "rti_n100_flat": dict(prediction_horizon=100, simulation_time=0.2, solver_options={"nlp_solver_type": "SQP_RTI"},
                          surface_coeffs=_SURF0, qp_fast_path=False),
    "rti_tight_bounds": dict(prediction_horizon=15, simulation_time=0.3, qdot_min=[-0.8] * 6, qdot_max=[0.8] * 6,
                             qdot_0=[0.5, 0.7, 0.5, 0.0, 0.0, 0.0], solver_options={"nlp_solver_type": "SQP_RTI"}, qp_fast_path=False),
    "rti_n20_fp": dict(prediction_horizon=20, simulation_time=0.5, solver_options={"nlp_solver_type": "SQP_RTI"}),
    "sqp_n10_fp": dict(prediction_horizon=10, simulation_time=0.3, solver_options={"nlp_solver_type": "SQP"}),
    "rti_n100_flat_fp": dict(prediction_horizon=100, simulation_time=0.2, solver_options={"nlp_solver_type": "SQP_RTI"}, surface_coeffs=_SURF0),
    "rti_tight_bounds_fp": dict(prediction_horizon=15, simulation_time=0.6, qdot_min=[-0.8] * 6, qdot_max=[0.8] * 6,
                                qdot_0=[0.5, 0.7, 0.5, 0.0, 0.0, 0.0], solver_options={"nlp_solver_type": "SQP_RTI"}),
}
KEYS = ("z", "u", "ee_pose", "ee_rpy", "ee_vel", "status", "sqp_iter", "qp_iter", "residuals", "cost")


def case_config(name):
    from robotic_mpc_amd import config

    kw = {k: (np.asarray(v, dtype=float) if isinstance(v, list) else v) for k, v in CASES[name].items()}
    return config.resolve_config(config.base_params(**kw))


def main():
    import helpers as hp
    from oracle import orc
    from robotic_mpc_amd import robots

    chain = robots.builtin_chain("ur10")
    rb = orc.make_robot(chain)
    for name in CASES:
        o = orc.run(rb, orc.make_params(case_config(name)))
        path = os.path.join(HERE, f"{name}.npz")
        if os.path.exists(path) and "--all" not in sys.argv:
            # an existing fixture is only rewritten on request: it pins the oracle against regressions
            g = np.load(path)
            print(name, "exists; max |oracle - fixture| on z:", float(np.abs(o["z"] - g["z"]).max()))
            continue
        np.savez_compressed(path, **{k: o[k] for k in KEYS})
        print(name, "steps", o["status"].shape[0], "qp_iter mean", o["qp_iter"].mean(), "fast-path steps", int((o["qp_iter"] == 1).sum()))
    # independent kinematic KATs (numpy 4x4 chain, not the oracle)
    rng = np.random.default_rng(123)
    for robot in ("ur10", "ur5"):
        ch = robots.builtin_chain(robot)
        q = rng.uniform(-np.pi, np.pi, (16, 6))
        T = np.stack([hp.fk_homogeneous(ch, qi)[0] for qi in q])
        np.savez_compressed(os.path.join(HERE, f"kin_{robot}.npz"), q=q, p=T[:, :3, 3], R=T[:, :3, :3])


if __name__ == "__main__":
    main()
