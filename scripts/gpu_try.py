"""Scratch GPU check: engine vs oracle on a small batch, then a timing run."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import orc
from robotic_mpc_amd import robots, config, engine

ch = robots.builtin_chain("ur10")
rb = orc.make_robot(ch)
eng = engine.MpcBatchEngine(0)
print("kernel info", eng.kernel_info(), flush=True)

def cfgs_for(N, T, st, B, seed=0):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(B):
        q0 = config.BASE_PARAMS["q_0"] + (rng.uniform(-0.1, 0.1, 6) if i else 0)
        out.append(config.resolve_config(config.base_params(prediction_horizon=N, simulation_time=T, q_0=q0,
                                                            solver_options={"nlp_solver_type": st})))
    return out

for (N, T, st, B) in [(20, 0.5, "SQP_RTI", 4), (20, 0.3, "SQP", 2), (100, 0.5, "SQP_RTI", 2)]:
    cf = cfgs_for(N, T, st, B)
    t = time.time(); r = eng.run(cf, ch); el = time.time() - t
    worst = {}
    for i, c in enumerate(cf):
        o = orc.run(rb, orc.make_params(c))
        for k in ("z", "u", "ee_pose", "ee_rpy", "ee_vel", "cost", "residuals"):
            worst[k] = max(worst.get(k, 0), float(np.abs(r[k][i] - o[k]).max()))
        worst["status"] = max(worst.get("status", 0), int((r["status"][i] != o["status"]).sum()))
        worst["qp_iter"] = max(worst.get("qp_iter", 0), int((r["qp_iter"][i] != o["qp_iter"]).sum()))
    print(f"N={N} T={T} {st} B={B}: {el:.3f}s kernel_ms={eng.last_kernel_ms} worst={worst}", flush=True)

for B in (64, 256):
    cf = cfgs_for(100, 6, "SQP_RTI", B)
    for c in cf: c["coeffs"][:] = 0.0
    t = time.time(); pb, bufs = eng.run_device(cf, ch); el = time.time() - t
    ms = sum(eng.last_kernel_ms)
    print(f"bench B={B} N=100 Nsim=600: wall {el:.3f}s kernel {ms:.1f} ms -> {B*600/(ms*1e-3):.0f} MPC-steps/s; "
          f"qp_iter mean {bufs['qp_iter'].double().mean().item():.2f} status!=0 {(bufs['status']!=0).sum().item()}", flush=True)
