"""Rate of one engine over batch sizes / horizons (diagnostic).  usage: gpu_stream_sweep.py ENGINE PREC B:N:T [B:N:T ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MPCB_ENGINE"] = sys.argv[1]
import bench
from robotic_mpc_amd import engine, robots
ch = robots.builtin_chain("ur10")
eng = engine.MpcBatchEngine(0, lib_path=os.environ.get("MPCB_LIB"))
prec = sys.argv[2]
for spec in sys.argv[3:]:
    B, N, T = spec.split(":")
    B, N, T = int(B), int(N), float(T)
    cfgs = bench.workload_configs(B, N, T, seed=1, solver=os.environ.get("SWEEP_SOLVER", "SQP_RTI"))
    for c in cfgs:
        c["precision"] = 1 if prec == "fp32" else 0
    pb, bufs = eng.run_device(cfgs, ch)
    ms = sum(eng.last_kernel_ms)
    qp = bufs["qp_iter"].double().mean().item()
    gb = B * pb.Nsim * qp * (N + 1) * 1678 * 8 / 1e9
    print(f"{os.environ.get('SWEEP_SOLVER', 'SQP_RTI')} {sys.argv[1]} {prec} B={B} N={N} Nsim={pb.Nsim} {eng.launch_info()} vgpr {eng.kernel_info()['vgprs']}: {ms:8.1f} ms {B*pb.Nsim/(ms*1e-3):10.0f} steps/s "
          f"qp_it {qp:.2f} fail {(bufs['status'] != 0).sum().item()} ~{gb/(ms*1e-3):.0f} GB/s of pass traffic", flush=True)
