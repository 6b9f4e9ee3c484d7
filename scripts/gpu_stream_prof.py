"""Per-pass device time of the throughput engine (diagnostic build -DMPCB_SPROF)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MPCB_ENGINE"] = "stream"
import numpy as np
import bench
from robotic_mpc_amd import engine, robots
ch = robots.builtin_chain("ur10")
from robotic_mpc_amd import build as _b
eng = engine.MpcBatchEngine(0, lib_path=_b.build_variant("sprof_lin" if os.environ.get("LIN") else "sprof", ["MPCB_SPROF"] + (["MPCB_SPROF_LIN"] if os.environ.get("LIN") else [])))     # (built on demand)
for B in [int(v) for v in sys.argv[1:]] or [1024, 2048]:
    cfgs = bench.workload_configs(B, 100, float(os.environ.get("TSIM", "0.5")), seed=1, solver=os.environ.get("SWEEP_SOLVER", "SQP_RTI"))
    pb, bufs = eng.run_device(cfgs, ch)
    ms = sum(eng.last_kernel_ms)
    out = np.zeros(16)
    eng.lib.mpcb_debug_profile(eng._h, B // 2, out.ctypes.data_as(C.POINTER(C.c_double)))
    it = max(out[7], 1.0)
    names = ["fact", "fwd_aff", "corr", "fwd", "resid", "ipm_total", "nlp_step_total"]
    nf = max(out[15], 1.0)
    print(f"B={B}: sim {B//2}: {out[15]:.0f} of {pb.Nsim} steps solved by the fast path; per fast step (us): right-hand side + previous step's NLP residuals "
          f"{out[10]/nf*1e6:.0f}, factorisation {out[11]/nf*1e6:.0f}, forward sweep {out[12]/nf*1e6:.0f}, commit {out[13]/nf*1e6:.0f}; per step: linearisation "
          f"{out[8]/pb.Nsim*1e6:.0f}, plant + log {out[14]/pb.Nsim*1e6:.0f}, whole solve {out[6]/pb.Nsim*1e6:.0f}", flush=True)
    if os.environ.get("LIN"):
        print(f"B={B}: lin_pass per step (us): loads + update + wait {out[0]/pb.Nsim*1e6:.1f}, task_lin + store issue {out[1]/pb.Nsim*1e6:.1f}, wait for the stores {out[2]/pb.Nsim*1e6:.1f}", flush=True)
    print(f"B={B}: kernel {ms:.1f} ms for {pb.Nsim} steps; sim {B//2}: {it:.0f} IPM iterations; per IPM iteration and stage (us): " +
          ", ".join(f"{n} {out[i]/it/101*1e6:.2f}" for i, n in enumerate(names[:5])) +
          f"; per step: ipm {out[5]/pb.Nsim*1e6:.0f} us, nlp_step {out[6]/pb.Nsim*1e6:.0f} us, lin_pass {out[8]/pb.Nsim*1e6:.0f} us, nlp_res {out[9]/pb.Nsim*1e6:.0f} us", flush=True)
