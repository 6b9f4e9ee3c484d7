"""Diagnostic: per-section device time of the rollout (libmpcbatch_prof.so, -DMPCB_PROFILE)."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from robotic_mpc_amd import robots, config, engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
T = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
SOLVER = sys.argv[4] if len(sys.argv) > 4 else "SQP_RTI"
from robotic_mpc_amd import build as _b
lib = os.environ.get("MPCB_LIB") or _b.build_variant("prof", ["MPCB_PROFILE"])     # (built on demand)
eng = engine.MpcBatchEngine(0, lib_path=lib)
print("kernel info", eng.kernel_info())
ch = robots.builtin_chain("ur10")
rng = np.random.default_rng(0)
cf = []
for i in range(B):
    c = config.resolve_config(config.base_params(prediction_horizon=N, simulation_time=T,
                                                 q_0=config.BASE_PARAMS["q_0"] + rng.uniform(-0.1, 0.1, 6),
                                                 surface_coeffs=dict(a=0, b=0, c=0, d=0, e=0, f=0),
                                                 solver_options={"nlp_solver_type": SOLVER}))
    cf.append(c)
pb, bufs = eng.run_device(cf, ch)
ms = sum(eng.last_kernel_ms)
print(f"B={B} N={N} Nsim={pb.Nsim}: kernel {ms:.1f} ms -> {B*pb.Nsim/(ms*1e-3):.0f} steps/s, per step {ms/pb.Nsim*1e3:.1f} us")
names = ["nlp", "res", "fact", "bwd", "fwd", "merit", "plant", "total", "ipm_iters", "io", "seq_fact", "seq_bwd", "seq_fwd", "res_A|nlp_update", "res_B|nlp_linearise", "res_C|nlp_norms+store"]
out = (C.c_double * 16)()
eng.lib.mpcb_debug_profile.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
for inst in (0, B // 2):
    eng.lib.mpcb_debug_profile(eng._h, inst, out)
    v = np.array(out[:])
    its = v[8]
    print(f"inst {inst}: ipm iters {its:.0f} ({its/pb.Nsim:.2f}/step)")
    for n, x in zip(names[:16], v[:16]):
        print(f"   {n:6s} {x*1e3:9.2f} ms  {x/pb.Nsim*1e6:8.1f} us/step  {100*x/max(v[7]+v[6],1e-12):5.1f}%")
