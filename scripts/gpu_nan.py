import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from robotic_mpc_amd import robots, config, engine
eng = engine.MpcBatchEngine(0)
ch = robots.builtin_chain("ur10")
for N, T in ((20, 0.05), (100, 0.03)):
    cf = [config.resolve_config(config.base_params(prediction_horizon=N, simulation_time=T)) for _ in range(2)]
    r = eng.run(cf, ch)
    print("N", N, "nan z", np.isnan(r["z"]).sum(), "qp_iter", r["qp_iter"][0], "status", r["status"][0], "cost", r["cost"][0], "res", r["residuals"][0][:2].tolist(), flush=True)
