"""Does splitting the closed loop into launches help the throughput engine's load balance? (diagnostic)
usage: gpu_split_launch.py B splits...   e.g.  4096 0 100 50,100 25,50,100,200"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MPCB_ENGINE"] = "stream"
import torch
import bench
from robotic_mpc_amd import engine, robots
ch = robots.builtin_chain("ur10")
eng = engine.MpcBatchEngine(0)
B = int(sys.argv[1])
cfgs = bench.workload_configs(B, 100, float(os.environ.get("TSIM", "6.0")), seed=1, solver="SQP_RTI")
pb = eng.setup(cfgs, ch)
bufs = eng.alloc_results(pb)
ref = None
for spec in sys.argv[2:]:
    cuts = [0] + [int(v) for v in spec.split(",") if int(v) > 0] + [pb.Nsim]
    ms = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        eng.rollout(bufs, a, b)
        eng.sync()
        ms.append(eng.kernel_ms())
    z = bufs["z"].clone()
    same = True if ref is None else bool(torch.equal(z, ref))
    ref = z if ref is None else ref
    print(f"B={B} Nsim={pb.Nsim} launches at {cuts[:-1]}: " + " + ".join(f"{m:.1f}" for m in ms) + f" = {sum(ms):.1f} ms -> {B*pb.Nsim/sum(ms)*1e3:.0f} steps/s; identical to the first variant: {same}", flush=True)
