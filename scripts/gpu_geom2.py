"""Latency-engine launch geometries at a fixed workload. usage: gpu_geom2.py B:N:T ... (env combos inside)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MPCB_ENGINE"] = "latency"
import bench
from robotic_mpc_amd import engine, robots
ch = robots.builtin_chain("ur10")
combos = [dict(), dict(MPCB_WAVES_PER_SIM="2", MPCB_SIMS_PER_CU="2"), dict(MPCB_WAVES_PER_SIM="4", MPCB_SIMS_PER_CU="1"),
          dict(MPCB_WAVES_PER_SIM="1", MPCB_SIMS_PER_CU="4"), dict(MPCB_WAVES_PER_SIM="2", MPCB_SIMS_PER_CU="1")]
for spec in sys.argv[1:]:
    B, N, T = spec.split(":"); B, N, T = int(B), int(N), float(T)
    cfgs = bench.workload_configs(B, N, T, seed=1, solver="SQP_RTI")
    for c in combos:
        for k in ("MPCB_WAVES_PER_SIM", "MPCB_SIMS_PER_CU", "MPCB_WPE"):
            os.environ.pop(k, None)
        os.environ.update(c)
        eng = engine.MpcBatchEngine(0)
        pb, bufs = eng.run_device(cfgs, ch)
        ms = sum(eng.last_kernel_ms)
        print(f"B={B} N={N} Nsim={pb.Nsim} {c or 'default'} {eng.launch_info()} vgpr {eng.kernel_info()['vgprs']}: {ms:8.1f} ms {B*pb.Nsim/(ms*1e-3):10.0f} steps/s", flush=True)
        eng.close()
