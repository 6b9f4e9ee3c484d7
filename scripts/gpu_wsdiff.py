"""Diagnostic: run the first closed-loop step with 1 and with 4 wavefronts per simulation and diff
the HBM workspaces group by group."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from robotic_mpc_amd import robots, config, engine
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ch = robots.builtin_chain("ur10")
IT = int(sys.argv[2]) if len(sys.argv) > 2 else 50
cf = [config.resolve_config(config.base_params(prediction_horizon=N, simulation_time=0.01, solver_options={"nlp_solver_type": "SQP_RTI", "qp_solver_iter_max": IT}))]
W = dict(G1=96, G2=112, G3=144, G4=228, G5=116)
def run(w, lib=None):
    os.environ["MPCB_WAVES_PER_SIM"] = str(w)
    eng = engine.MpcBatchEngine(0, lib_path=lib) if lib else engine.MpcBatchEngine(0)
    pb = eng.setup(cf, ch); bufs = eng.alloc_results(pb)
    eng.rollout(bufs, 0, 1); eng.sync()
    n = (N + 1) * 696 + 64
    out = np.zeros(n)
    eng.lib.mpcb_debug_workspace.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_size_t]
    rc = eng.lib.mpcb_debug_workspace(eng._h, 0, out.ctypes.data_as(C.POINTER(C.c_double)), n)
    assert rc == 0
    return out, {k: v.cpu().numpy() for k, v in bufs.items()}
a, ra = run(1, os.environ.get("MPCB_LIB"))
b, rb = run(4, os.environ.get("MPCB_LIB"))
off = 0
for g, w in W.items():
    A = a[off:off + (N + 1) * w].reshape(N + 1, w); B = b[off:off + (N + 1) * w].reshape(N + 1, w)
    d = np.abs(A - B); d[np.isnan(d)] = np.inf
    bad = np.argwhere(d > 1e-9)
    print(g, "max diff", d.max(), "nan in 4w", np.isnan(B).sum(), "first bad (stage,col):", bad[:4].tolist())
    for st in (N, N - 1, 1, 0):
        cols = np.where(d[st] > 1e-9)[0]
        if len(cols): print("    stage", st, "bad cols", cols[:40].tolist(), "n", len(cols))
    off += (N + 1) * w
print("qp_iter", ra["qp_iter"][0], rb["qp_iter"][0], "cost", ra["cost"][0], rb["cost"][0])
