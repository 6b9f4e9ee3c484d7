"""BASELINE configs[2] in full on ONE GPU: grid {N:[20,50,100,200]} x {w_qddot} x {w_u} x 256 coefficient sets = 4096
simulations, 600 steps each, through SimulationManager.run_all -- as one ragged launch of the throughput engine, and as
one launch per horizon (the latency engine at 1024 simulations each)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from robotic_mpc_amd import SimulationManager, base_params, packing
import test_gpu_configs as tg

sets = tg.surface_coeff_sets(256)
# one-time costs (library load, HIP context, allocator warm-up) are not part of either variant
_w = SimulationManager(base_params(simulation_time=0.05))
_w.grid_search({"prediction_horizon": [20]}, surface_coeff_sets=sets[:1])
_w.run_all()
for label, thr in (("ragged (one launch)", 2048), ("one launch per horizon", 10**9)):
    packing.RAGGED_MIN_BATCH = thr
    m = SimulationManager(base_params())
    m.grid_search({"prediction_horizon": [20, 50, 100, 200], "w_qddot": [0.02, 0.05], "w_u": [0.01, 0.001]}, surface_coeff_sets=sets)
    t = time.time()
    res = m.run_all()
    wall = time.time() - t
    info = m.last_run_info
    wr = np.array([r["summary"]["weighted_rmse"] for r in res])
    fails = sum(r["summary"]["num_failures"] for r in res)
    print(f"{label}: {len(res)} sims, buckets {info['buckets']}, wall {wall:.2f} s (setup {info['setup_s']:.2f}, run {info['run_s']:.2f}, "
          f"kernels {info['kernel_ms']/1e3:.2f} s summed, d2h {info['d2h_s']:.2f}) -> {len(res)*600/wall:.0f} MPC-steps/s end to end; "
          f"weighted RMSE median {np.median(wr):.4f} [{wr.min():.3f}, {wr.max():.3f}], solver failures {fails}", flush=True)
