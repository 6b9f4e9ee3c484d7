"""Quick kernel-rate check of the latency engine: python scripts/gpu_quick.py [B N T] (env MPCB_WAVES_PER_SIM etc. apply)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from robotic_mpc_amd import robots, config, engine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 100
T = float(sys.argv[3]) if len(sys.argv) > 3 else 6.0
eng = engine.MpcBatchEngine(0, lib_path=os.environ.get("MPCB_LIB"))
ch = robots.builtin_chain("ur10")
rng = np.random.default_rng(0)
cf = [config.resolve_config(config.base_params(prediction_horizon=N, simulation_time=T, q_0=config.BASE_PARAMS["q_0"] + rng.uniform(-0.1, 0.1, 6),
                                               surface_coeffs=dict(a=0, b=0, c=0, d=0, e=0, f=0))) for _ in range(B)]
for rep in range(2):
    pb, bufs = eng.run_device(cf, ch)
ms = sum(eng.last_kernel_ms)
print(f"B={B} N={N} Nsim={pb.Nsim} geo={eng.launch_info()} {eng.kernel_info()}: kernel {ms:.1f} ms -> {B*pb.Nsim/(ms*1e-3):.0f} steps/s; qp_iter mean {bufs['qp_iter'].double().mean().item():.3f} fails {(bufs['status']!=0).sum().item()}")
