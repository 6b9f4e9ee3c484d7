"""Where the time of one bench pass goes on the host side: mpcb_setup, rollout (+ sync), summary kernel, D2H of the 14 arrays."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from robotic_mpc_amd import engine, robots, distributed as dmod
torch.cuda.set_device(0)
ch = robots.builtin_chain("ur10")
cfgs = bench.workload_configs(256, 100, 6.0, seed=0, solver="SQP_RTI")
eng = engine.MpcBatchEngine(0)
pb, params, robot = eng.prepare(cfgs, ch)
eng.setup_packed(pb, params, robot)
bufs = eng.alloc_results(pb)
host_bufs = {k: torch.empty(v.shape, dtype=v.dtype, pin_memory=True) for k, v in bufs.items()}
host_bufs["summary"] = torch.empty((256, engine.NSUMMARY), dtype=torch.float64, pin_memory=True)
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); eng.setup_packed(pb, params, robot); t1 = time.perf_counter()
    eng.rollout(bufs, 0, pb.Nsim); t2 = time.perf_counter()
    local = dict(bufs); local["summary"] = eng.summary(bufs); t3 = time.perf_counter()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    host = {k: dmod.to_host(v, host_bufs[k]) for k, v in local.items()}; t5 = time.perf_counter()
    eng.sync(); t6 = time.perf_counter()
    print(f"rep {rep}: setup {1e3*(t1-t0):.2f} ms | rollout call {1e3*(t2-t1):.2f} | summary call {1e3*(t3-t2):.2f} | wait for kernels {1e3*(t4-t3):.2f} (kernel {eng.kernel_ms():.2f}) | "
          f"D2H {1e3*(t5-t4):.2f} ms ({sum(v.nbytes for v in host.values())/1e6:.1f} MB) | sync {1e3*(t6-t5):.2f} | total {1e3*(t6-t0):.2f}", flush=True)
