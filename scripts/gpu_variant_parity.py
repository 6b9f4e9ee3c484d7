"""A variant build of the engine against the default build on the same simulations (diagnostic; the default build is what the
parity tests hold against the oracle).  usage: MPCB_LIB=.../libmpcbatch_X.so python scripts/gpu_variant_parity.py [B:N:T:SOLVER ...]
Environment switches (MPCB_SIMS_PER_CU ...) apply to the variant only -- unless VARIANT_ENV_BOTH=1 (e.g. MPCB_ENGINE=stream for an A/B of two
builds of the throughput engine); the solver's own outputs (cost, residual norms) are compared too, at 1e-9 relative to max(|value|, 1e-3)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from robotic_mpc_amd import engine, robots

ch = robots.builtin_chain("ur10")
specs = sys.argv[1:] or ["6:100:0.4:SQP_RTI", "6:100:0.2:SQP", "5:200:0.2:SQP_RTI", "4:30:0.3:SQP", "3:130:0.2:SQP"]
both = os.environ.pop("VARIANT_ENV_BOTH", "") == "1"
env = {} if both else {k: os.environ.pop(k) for k in list(os.environ) if k.startswith("MPCB_") and k != "MPCB_LIB"}
bad = 0
for spec in specs:
    B, N, T, solver = spec.split(":")
    cfgs = bench.workload_configs(int(B), int(N), float(T), seed=3, solver=solver)
    e = engine.MpcBatchEngine(0)
    a = e.run(cfgs, ch)
    e.close()
    os.environ.update(env)
    e = engine.MpcBatchEngine(0, lib_path=os.environ["MPCB_LIB"])
    b = e.run(cfgs, ch)
    geo = e.launch_info()
    e.close()
    for k in env:
        os.environ.pop(k)
    same = all(np.array_equal(a[k], b[k]) for k in ("status", "sqp_iter", "qp_iter"))
    err = max(float(np.abs(a[k] - b[k]).max()) for k in ("z", "u", "ee_pose", "errors"))
    # (residual norms of a converged step are rounding-level numbers: 1e-12 absolute + 1e-9 relative)
    rel = max(float((np.abs(a[k] - b[k]) / (1e-3 + np.maximum(np.abs(a[k]), np.abs(b[k])))).max()) for k in ("cost", "residuals"))
    ok = same and err < 1e-10 and rel < 1e-9
    bad += not ok
    print(f"{spec}: geometry {geo}: decisions {'same' if same else 'DIFFER'}, max |diff| {err:.2e}, cost / residual norms rel {rel:.1e} {'ok' if ok else 'FAIL'}", flush=True)
sys.exit(1 if bad else 0)
