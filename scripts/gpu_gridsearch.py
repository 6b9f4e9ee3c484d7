"""BASELINE.json configs[2] on one GPU's share: grid_search {N in [20,50,100,200], w_qddot in [0.02,0.05],
w_u in [0.01,0.001]} x random surface coefficient sets (SURVEY.md 8d Config 3), through the public
SimulationManager API: bucketing, batched launches, result objects, summaries.

usage: python scripts/gpu_gridsearch.py [n_coeff_sets=32] [simulation_time=6.0]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from robotic_mpc_amd import SimulationManager, base_params

n_sets = int(sys.argv[1]) if len(sys.argv) > 1 else 32
T = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
np.random.seed(42)                                   # surface_stats.ipynb cells 1+7
mean = {"a": -0.1, "b": 0.1, "c": -0.01, "d": 0.01, "e": 0.01, "f": 0.0}
sets = [{k: float(np.random.normal(mean[k], 0.01)) for k in "abcdef"} for _ in range(n_sets)]
mgr = SimulationManager(base_params(simulation_time=T, solver_options={"nlp_solver_type": "SQP_RTI"}))
mgr.grid_search({"prediction_horizon": [20, 50, 100, 200], "w_qddot": [0.02, 0.05], "w_u": [0.01, 0.001]},
                surface_coeff_sets=sets)
t0 = time.time()
res = mgr.run_all()
wall = time.time() - t0
steps = sum(r["simulator"].Nsim for r in res)
fails = sum(r["summary"]["num_failures"] for r in res)
wr = np.array([r["summary"]["weighted_rmse"] for r in res])
print(f"{len(res)} simulations in {mgr.last_run_info['buckets']} buckets: {wall:.2f} s wall (launches + copies + host analysis), "
      f"{steps / wall:.0f} MPC-steps/s end to end; solver failures {fails}; weighted RMSE min/median/max "
      f"{wr.min():.3f}/{np.median(wr):.3f}/{wr.max():.3f}")
for N in (20, 50, 100, 200):
    sel = [r for r in res if r["simulator"].prediction_horizon == N]
    print(f"  N={N:3d}: {len(sel)} sims, median final cost {np.median([r['simulator'].cost_history[-1] for r in sel]):.2e}")
