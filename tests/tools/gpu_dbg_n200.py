"""Test tooling (it uses the oracle, so it lives under tests/): the long-horizon grid corner of configs[2] on the GPU against the oracle,
step by step -- where the two drift apart and that the solver decisions stay equal."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import orc
from robotic_mpc_amd import SimulationManager, base_params, engine, robots
import test_gpu_configs as tg
orc.build()
ch = robots.builtin_chain("ur10"); rb = orc.make_robot(ch)
m = SimulationManager(base_params())
m.grid_search({"prediction_horizon": [int(os.environ.get("DBG_N", "200"))], "w_qddot": [0.02, 0.05], "w_u": [0.01, 0.001]},
              surface_coeff_sets=tg.surface_coeff_sets(16))
from robotic_mpc_amd.simulator import Simulator
sims = [Simulator(**s["config"]) for s in m.simulations]
cfgs = [s.resolved for s in sims]
eng = engine.MpcBatchEngine(0)
out = eng.run(cfgs, ch)
for i in [int(v) for v in os.environ.get("DBG_I", "56,0,17").split(",")]:
    ref = orc.run(rb, orc.make_params(cfgs[i]))
    du = np.abs(out["u"][i] - ref["u"]).max(axis=0)
    bad = np.nonzero(du > 1e-9)[0]
    dq = np.nonzero(out["qp_iter"][i] != ref["qp_iter"])[0]
    print(f"sim {i}: max|du| {du.max():.3e} first bad col {bad[:1]}, qp_iter mismatches at steps {dq[:10]} (n={dq.size}); "
          f"status gpu {np.unique(out['status'][i])} orc {np.unique(ref['status'])}")
    if bad.size:
        b = bad[0] - 1
        for s in range(max(b - 2, 0), min(b + 3, 600)):
            print(f"   step {s}: qp_iter gpu {out['qp_iter'][i][s]} orc {ref['qp_iter'][s]}  |du| {du[s+1]:.3e} res gpu {out['residuals'][i][s]} orc {ref['residuals'][s]}")
