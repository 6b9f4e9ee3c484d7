"""Diagnostic: long horizons (N up to 300: many chunks per pass) and SQP against the oracle on the GPU."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import orc
from robotic_mpc_amd import engine, robots, config
orc.build()
ch = robots.builtin_chain("ur10"); rb = orc.make_robot(ch)
eng = engine.MpcBatchEngine(0)
for N, T, solver in [(300, 0.05, "SQP_RTI"), (200, 0.05, "SQP"), (57, 0.01, "SQP_RTI"), (100, 0.01, "SQP")]:
    cfgs = [config.resolve_config(config.base_params(prediction_horizon=N, simulation_time=T, q_0=config.BASE_PARAMS["q_0"] + 0.05 * i,
                                                     solver_options={"nlp_solver_type": solver})) for i in range(2)]
    out = eng.run(cfgs, ch)
    for i, c in enumerate(cfgs):
        ref = orc.run(rb, orc.make_params(c))
        d = max(np.abs(out[k][i] - ref[k]).max() for k in ("z", "u", "ee_pose"))
        print(N, solver, i, "max diff %.2e" % d, "qp_iter equal", (out["qp_iter"][i] == ref["qp_iter"]).all(), "status equal", (out["status"][i] == ref["status"]).all())
