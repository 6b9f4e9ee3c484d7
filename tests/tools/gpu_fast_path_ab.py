"""Fast path of the QP solve (csrc/mpc_ipm.h) ON vs OFF on BASELINE configs[1] (256 simulations, N=100, 600 steps, SQP_RTI):
kernel rates, factorisations per step, ON-vs-OFF deviation at qp_tol 1e-8 and 1e-12, and HIP-vs-oracle spot checks with the
switch in both positions.  Diagnostic (calls the oracle as the checker: lives under tests/)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from oracle import orc
from robotic_mpc_amd import engine, robots

eng = engine.MpcBatchEngine(0, lib_path=os.environ.get("MPCB_LIB"))
ch = robots.builtin_chain("ur10")
B = int(os.environ.get("AB_BATCH", 256)); N = int(os.environ.get("AB_N", 100)); T = float(os.environ.get("AB_T", 6.0))
res = {}
for tol in (1e-8, 1e-12):
    for fast in (1, 0):
        cfgs = bench.workload_configs(B, N, T, seed=0, solver="SQP_RTI")
        for c in cfgs:
            c["qp_fast_path"] = fast; c["qp_tol"] = tol; c["qp_iter_max"] = 50 if tol > 1e-9 else 200
        best = 1e30
        for rep in range(3):
            pb, bufs = eng.run_device(cfgs, ch)
            best = min(best, sum(eng.last_kernel_ms))
        out = {k: v.cpu().numpy() for k, v in bufs.items()}
        res[(tol, fast)] = (cfgs, out)
        qi = out["qp_iter"]
        print(f"qp_tol {tol:g} fast {fast}: kernel {best:7.1f} ms {B*pb.Nsim/(best*1e-3):9.0f} steps/s  factorisations/step {qi.mean():.3f} "
              f"(==1: {100*(qi==1).mean():.1f}%)  slowest sim {qi.sum(1).max()} fastest {qi.sum(1).min()} fails {(out['status']!=0).sum()} info {eng.launch_info()}", flush=True)
for tol in (1e-8, 1e-12):
    a, b = res[(tol, 1)][1], res[(tol, 0)][1]
    dz = np.abs(a['z'] - b['z']).max(axis=(1, 2))
    capped = (b['qp_iter'] >= 50).any(axis=1) if tol > 1e-9 else np.zeros(len(dz), bool)
    print(f"qp_tol {tol:g}: ON vs OFF max|dz| {dz.max():.2e} (sim {dz.argmax()}); without the {capped.sum()} simulations whose OFF run has a capped QP: "
          f"max {dz[~capped].max():.2e} median {np.median(dz[~capped]):.2e}; |du| {np.abs(a['u']-b['u']).max():.2e} status equal {np.array_equal(a['status'], b['status'])}")
rb = orc.make_robot(ch)
for (tol, fast) in ((1e-8, 1), (1e-8, 0)):
    cfgs, out = res[(tol, fast)]
    worst = 0.0
    for i in (0, 7, 100, 255)[: (4 if B >= 256 else 1)]:
        ref = orc.run(rb, orc.make_params(cfgs[i]))
        d = max(np.abs(out[k][i] - ref[k]).max() for k in ("z", "u", "ee_pose"))
        worst = max(worst, d)
        assert np.array_equal(out["status"][i], ref["status"]) and np.array_equal(out["qp_iter"][i], ref["qp_iter"]), (i, fast)
    print(f"HIP vs oracle, fast {fast}: max dev {worst:.2e}, status / qp_iter identical")
