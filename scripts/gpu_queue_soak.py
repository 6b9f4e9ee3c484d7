"""Soak of the work-queue launch: the same batch several times, queued, against ONE plain launch -- any lost hand-off
(stale line, early start of a chunk) would show as a bitwise difference somewhere in 4096 x 600 steps."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MPCB_ENGINE"] = "stream"
import torch
import bench
from robotic_mpc_amd import engine, robots
ch = robots.builtin_chain("ur10")
B, reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4096, int(sys.argv[2]) if len(sys.argv) > 2 else 6
cfgs = bench.workload_configs(B, 100, 6.0, seed=7, solver=os.environ.get("SWEEP_SOLVER", "SQP_RTI"))
os.environ["MPCB_STREAM_CHUNK"] = "0"
eng = engine.MpcBatchEngine(0)
pb = eng.setup(cfgs, ch)
ref = eng.alloc_results(pb)
eng.rollout(ref, 0, pb.Nsim); eng.sync()
print(f"plain launch: {eng.kernel_ms():.1f} ms", flush=True)
keys = [k for k in ref if k not in ("solver_time", "plant_time")]
bad = 0
for chunk in ["10", "3", "25"] * ((reps + 2) // 3):
    os.environ["MPCB_STREAM_CHUNK"] = chunk
    out = eng.alloc_results(pb)
    eng.rollout(out, 0, pb.Nsim); eng.sync()
    same = all(torch.equal(out[k], ref[k]) for k in keys)
    bad += 0 if same else 1
    print(f"queued, chunk {chunk}: {eng.kernel_ms():.1f} ms, bitwise equal to the plain launch: {same}", flush=True)
    del out
print("SOAK", "ok" if bad == 0 else f"FAILED ({bad} runs differ)")
sys.exit(0 if bad == 0 else 1)
