"""Throughput engine vs oracle on small cases, then its rate at large batch (diagnostic driver)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
os.environ["MPCB_ENGINE"] = "stream"
import bench
from oracle import orc
from robotic_mpc_amd import config, engine, robots
orc.build()
ch = robots.builtin_chain("ur10"); rb = orc.make_robot(ch)
eng = engine.MpcBatchEngine(0)
rng = np.random.default_rng(0)
cases = [(3, 0.03, 2), (5, 0.05, 3), (20, 0.3, 3), (100, 0.3, 2), (130, 0.05, 2)]
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    cases = cases[:2]
for N, T, B in cases:
    for prec in ("fp64", "fp32"):
        cfgs = [config.resolve_config(config.base_params(prediction_horizon=N, simulation_time=T, riccati_precision=prec,
                                                         q_0=config.BASE_PARAMS["q_0"] + rng.uniform(-0.1, 0.1, 6))) for _ in range(B)]
        out = eng.run(cfgs, ch)
        info = eng.launch_info()
        worst, ok_it = 0.0, True
        for i, c in enumerate(cfgs):
            ref = orc.run(rb, orc.make_params(c))
            d = max(np.abs(out[k][i] - ref[k]).max() for k in ("z", "u"))
            worst = max(worst, d)
            same = np.array_equal(out["qp_iter"][i], ref["qp_iter"]) and np.array_equal(out["status"][i], ref["status"])
            ok_it &= same
            if (d > 1e-9 or not same) and prec == "fp64":
                bad = np.nonzero(np.abs(out["u"][i] - ref["u"]).max(axis=0) > 1e-9)[0]
                print(f"   sim {i}: first bad col {bad[:3]} qp gpu {out['qp_iter'][i][:8]} orc {ref['qp_iter'][:8]} st {out['status'][i][:8]} "
                      f"res gpu {out['residuals'][i][0]} orc {ref['residuals'][0]} cost {out['cost'][i][:3]} {ref['cost'][:3]}")
        print(f"N={N} steps={cfgs[0]['Nsim']} B={B} {prec} engine={info['engine']}: max|gpu-oracle| {worst:.3e} iters_equal {ok_it} "
              f"qp_it {out['qp_iter'].mean():.2f} kernel {eng.kernel_info()}", flush=True)
if os.environ.get("STREAM_BENCH", "1") == "1":
    for B, T in ((4096, 1.5), (2048, 1.5)):
        for prec in ("fp64", "fp32"):
            cfgs = bench.workload_configs(B, 100, T, seed=1, solver="SQP_RTI")
            for c in cfgs:
                c["precision"] = 1 if prec == "fp32" else 0
            pb, bufs = eng.run_device(cfgs, ch)
            ms = sum(eng.last_kernel_ms)
            print(f"B={B} N=100 Nsim={pb.Nsim} {prec}: {ms:8.1f} ms {B*pb.Nsim/(ms*1e-3):10.0f} steps/s qp_it {bufs['qp_iter'].double().mean().item():.2f} "
                  f"fail {(bufs['status'] != 0).sum().item()}", flush=True)
