"""Diagnostic: kernel time of BASELINE configs[1] for every library under variants/ (built with -D knobs such as
MPCB_FWD_BLK, MPCB_POLL_SLEEP) beside the default build."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, bench
from robotic_mpc_amd import engine, robots
ch = robots.builtin_chain("ur10")
cfgs = bench.workload_configs(256, 100, 6.0, seed=0, solver="SQP_RTI")
for lib in [None] + sorted(os.path.join("variants", f) for f in os.listdir("variants")):
    eng = engine.MpcBatchEngine(0, lib_path=os.path.abspath(lib) if lib else None)
    best = 1e9
    for rep in range(3):
        pb, bufs = eng.run_device(cfgs, ch)
        best = min(best, sum(eng.last_kernel_ms))
    print(f"{lib or 'default':40s} {best:8.2f} ms  {256*600/(best*1e-3):10.0f} steps/s", flush=True)
    eng.close()
