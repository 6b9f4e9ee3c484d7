import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from oracle import orc
from robotic_mpc_amd import engine, robots
eng = engine.MpcBatchEngine(0)
ch = robots.builtin_chain("ur10"); rb = orc.make_robot(ch)
cfgs = bench.workload_configs(32, 100, 3.0, seed=1, solver="SQP")
out = eng.run(cfgs, ch)
st = out["status"]
print("status histogram", np.bincount(st.ravel(), minlength=5))
bad = np.argwhere(st != 0)
print("first bad (inst, step):", bad[:10].tolist())
insts = sorted(set(bad[:, 0].tolist()))[:2] or [0]
for i in insts:
    o = orc.run(rb, orc.make_params(cfgs[i]))
    print("inst", i, "oracle status hist", np.bincount(o["status"], minlength=5), "gpu", np.bincount(st[i], minlength=5))
    d = np.abs(out["z"][i] - o["z"]).max(); du = np.abs(out["u"][i] - o["u"]).max()
    print("   max|z diff|", d, "max|u diff|", du, "status equal", (st[i] == o["status"]).all(), "sqp_iter equal", (out["sqp_iter"][i] == o["sqp_iter"]).all())
    k = np.argwhere(st[i] != 0)[:3].ravel()
    for s in k:
        print("   step", s, "gpu status", st[i][s], "sqp", out["sqp_iter"][i][s], "qp", out["qp_iter"][i][s], "res", out["residuals"][i][s], "| oracle", o["status"][s], o["sqp_iter"][s], o["qp_iter"][s], o["residuals"][s])
