"""BASELINE configs[2] per GPU share: 16 grid points x S coefficient sets (S = 32 / 64 / 128 / 256: what one GPU of 8 / 4 / 2 / 1 gets),
600 steps each, through SimulationManager.run_all -- as ONE ragged launch of the throughput engine, and as one launch per horizon
(the default engine of each bucket).  Re-derives packing.RAGGED_MIN_BATCH (the merge threshold) and shows what the N = 200 bucket costs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from robotic_mpc_amd import SimulationManager, base_params, packing
import test_gpu_configs as tg

sets = tg.surface_coeff_sets(256)
_w = SimulationManager(base_params(simulation_time=0.05))
_w.grid_search({"prediction_horizon": [20]}, surface_coeff_sets=sets[:1])
_w.run_all()
for S in (32, 64, 128, 256):
    for label, thr in (("ragged (one launch)", 1), ("one launch per horizon", 10**9)):
        packing.RAGGED_MIN_BATCH = thr
        best = None
        for rep in range(2):
            m = SimulationManager(base_params())
            m.grid_search({"prediction_horizon": [20, 50, 100, 200], "w_qddot": [0.02, 0.05], "w_u": [0.01, 0.001]}, surface_coeff_sets=sets[:S])
            t = time.time()
            m.run_all(return_results=False)
            wall = time.time() - t
            info = m.last_run_info
            if best is None or info["run_s"] < best[1]["run_s"]:
                best = (wall, info)
        wall, info = best
        n = 16 * S
        print(f"{n:5d} sims ({S} sets), {label}: buckets {info['buckets']}, wall {wall:.3f} s (setup {info['setup_s']:.2f}, run {info['run_s']:.3f}, "
              f"kernels {info['kernel_ms']/1e3:.3f} s summed) -> {n*600/info['run_s']:.0f} MPC-steps/s over the run phase, {n*600/wall:.0f} end to end", flush=True)
