"""Launch-geometry exploration at large batch: waves per simulation x simulations per CU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from robotic_mpc_amd import engine, robots

ch = robots.builtin_chain("ur10")
B = int(os.environ.get("GEOM_B", "4096"))
T = float(os.environ.get("GEOM_T", "1.5"))
cfgs = bench.workload_configs(B, 100, T, seed=1, solver="SQP_RTI")
combos = [(2, 2), (1, 2), (1, 4), (2, 4), (4, 1), (4, 2), (1, 8), (2, 8)]
if len(sys.argv) > 1:
    combos = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for lib in os.environ.get("GEOM_LIBS", "").split(",") or [""]:
    for nw, spc in combos:
        os.environ["MPCB_WAVES_PER_SIM"] = str(nw)
        os.environ["MPCB_SIMS_PER_CU"] = str(spc)
        try:
            eng = engine.MpcBatchEngine(0, lib_path=lib or None)
            pb, bufs = eng.run_device(cfgs, ch)
            ms = sum(eng.last_kernel_ms)
            print(f"lib={os.path.basename(lib) or 'default'} B={B} waves/sim={nw} sims/CU={spc} {eng.launch_info()} {eng.kernel_info()['vgprs']} vgpr: "
                  f"{ms:8.1f} ms {B*pb.Nsim/(ms*1e-3):10.0f} steps/s qp_it {bufs['qp_iter'].double().mean().item():.2f} "
                  f"fail {(bufs['status'] != 0).sum().item()}", flush=True)
            eng.close()
        except Exception as e:
            print(f"lib={lib} waves/sim={nw} sims/CU={spc}: FAILED {e!r}", flush=True)
