"""One GPU's share of BASELINE configs[2] on an 8-GPU node: 512 simulations, 128 per horizon N in {20, 50, 100, 200}, 600 steps,
through SimulationManager.run_all (one launch per horizon, the launches overlapping on the device) -- with the default launch
geometry of each bucket (128 <= #CUs: one simulation per CU, the whole LDS pool) and with MPCB_SIMS_PER_CU=2 (half a pool each,
so that all 512 workgroups are resident at once)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from robotic_mpc_amd import SimulationManager, base_params
import test_gpu_configs as tg

sets = tg.surface_coeff_sets(32)
_w = SimulationManager(base_params(simulation_time=0.05))
_w.grid_search({"prediction_horizon": [20]}, surface_coeff_sets=sets[:1])
_w.run_all()
for rep in range(2):
    for label, env in (("default geometry", None), ("MPCB_SIMS_PER_CU=2", "2")):
        if env: os.environ["MPCB_SIMS_PER_CU"] = env
        else: os.environ.pop("MPCB_SIMS_PER_CU", None)
        m = SimulationManager(base_params())
        m.grid_search({"prediction_horizon": [20, 50, 100, 200], "w_qddot": [0.02, 0.05], "w_u": [0.01, 0.001]}, surface_coeff_sets=sets)
        t = time.time()
        res = m.run_all()
        wall = time.time() - t
        info = m.last_run_info
        print(f"{label}: {len(res)} sims, buckets {info['buckets']}, wall {wall:.3f} s (setup {info['setup_s']:.2f}, run {info['run_s']:.3f}, "
              f"kernels {info['kernel_ms']/1e3:.3f} s summed, d2h {info['d2h_s']:.2f}) -> {len(res)*600/wall:.0f} MPC-steps/s end to end, "
              f"{len(res)*600/info['run_s']:.0f} over the run phase", flush=True)
