"""Where a closed-loop step of the throughput engine goes (diagnostic): kernel time per step against the in-kernel clocks
(solver_time = residual<0> .. nlp_res, plant_time = plant step + logging)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MPCB_ENGINE"] = "stream"
import bench
from robotic_mpc_amd import engine, robots
ch = robots.builtin_chain("ur10")
eng = engine.MpcBatchEngine(0, lib_path=os.environ.get("MPCB_LIB"))
for B in [int(v) for v in sys.argv[1:]] or [1024, 2048]:
    cfgs = bench.workload_configs(B, 100, float(os.environ.get("TSIM", "1.0")), seed=1, solver="SQP_RTI")
    pb, bufs = eng.run_device(cfgs, ch)
    ms = sum(eng.last_kernel_ms)
    st = bufs["solver_time"].double().mean().item() * 1e6
    pt = bufs["plant_time"].double().mean().item() * 1e6
    rounds = max(1.0, B / 2048.0)
    tot = (bufs["solver_time"].double().sum(dim=1) + bufs["plant_time"].double().sum(dim=1)) * 1e3    # ms per simulation
    its = bufs["qp_iter"].double().sum(dim=1)
    print(f"   per-simulation in-kernel total: mean {tot.mean().item():.1f} ms, min {tot.min().item():.1f}, max {tot.max().item():.1f}, "
          f"std {tot.std().item():.1f}; IPM iterations per simulation: mean {its.mean().item():.0f}, max {its.max().item():.0f}", flush=True)
    print(f"B={B}: kernel {ms:.1f} ms / {pb.Nsim} steps / {rounds:.1f} rounds = {ms/pb.Nsim/rounds*1e3:.0f} us per simulation-step; "
          f"in-kernel clocks: solver {st:.0f} us + plant/log {pt:.0f} us = {st+pt:.0f} us", flush=True)
