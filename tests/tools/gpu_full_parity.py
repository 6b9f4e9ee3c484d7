"""Every simulation of BASELINE configs[1] (256 x 600 steps) on BOTH engines, and every 4th of configs[2]'s 4096 (N in {20, 50, 100, 200}, ragged
launch + per-horizon launches), against the oracle -- all of them, not spot checks.  Diagnostic (the oracle is the checker: lives under tests/);
16 host cores run the oracle.  Prints one line per leg: max deviation on q, qdot, u, poses and whether status / sqp_iter / qp_iter are identical."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from concurrent.futures import ProcessPoolExecutor


def _oracle(args):
    from oracle import orc
    from robotic_mpc_amd import robots
    cfg = args
    r = orc.run(orc.make_robot(robots.builtin_chain("ur10"), cfg["t_ee"]), orc.make_params(cfg))
    return {k: r[k] for k in ("z", "u", "ee_pose", "status", "sqp_iter", "qp_iter")}


def compare(label, cfgs, out, pool, strict_steps=None, to_first_flag=False):
    t = time.time()
    refs = list(pool.map(_oracle, cfgs, chunksize=4))
    worst, worst_i, same, flagged = 0.0, -1, 0, 0
    for i, ref in enumerate(refs):
        n = ref["status"].shape[0] if strict_steps is None else strict_steps
        if to_first_flag:      # full SQP: steps that end with status 2 / 3 / 4 are path-dependent: strict up to the first one
            bad = np.nonzero((ref["status"] != 0) | (out["status"][i] != 0))[0]
            n = int(bad[0]) if bad.size else n
            flagged += int(bad.size > 0)
        d = max(float(np.abs(out[k][i][:, :n + 1] - ref[k][:, :n + 1]).max()) for k in ("z", "u", "ee_pose"))
        if d > worst:
            worst, worst_i = d, i
        same += int(all(np.array_equal(out[k][i][:n], ref[k][:n]) for k in ("status", "sqp_iter", "qp_iter")))
    if to_first_flag:
        label += f" [{flagged} simulations have a flagged step; compared up to it]"
    print(f"{label}: {len(cfgs)} simulations x {cfgs[0]['Nsim']} steps: max |gpu - oracle| = {worst:.2e} (simulation {worst_i}"
          f"{'' if strict_steps is None else f', first {strict_steps} steps'}); status / sqp_iter / qp_iter identical in {same} of {len(cfgs)}; "
          f"oracle {time.time() - t:.0f} s on the host", flush=True)


if __name__ == "__main__":
    import bench
    from robotic_mpc_amd import engine, robots, config
    import test_gpu_configs as tg
    ch = robots.builtin_chain("ur10")
    with ProcessPoolExecutor(int(os.environ.get("ORACLE_PROCS", "16"))) as pool:
        only_sqp = os.environ.get("PARITY_LEGS", "all") == "sqp"
        cfgs = bench.workload_configs(256, 100, 6.0, seed=0, solver="SQP_RTI")
        for name, env in (() if only_sqp else (("latency engine", None), ("throughput engine", "stream"))):
            if env: os.environ["MPCB_ENGINE"] = env
            else: os.environ.pop("MPCB_ENGINE", None)
            e = engine.MpcBatchEngine(0); out = e.run(cfgs, ch); info = e.launch_info(); e.close()
            compare(f"configs[1] on the {name} {info}", cfgs, out, pool)
        os.environ.pop("MPCB_ENGINE", None)
        sets = tg.surface_coeff_sets(64)
        allc = []
        for s_ in sets:
            for N in (20, 50, 100, 200):
                for wq in (0.02, 0.05):
                    for wu in (0.01, 0.001):
                        allc.append(config.resolve_config(config.base_params(prediction_horizon=N, w_qddot=wq, w_u=wu, surface_coeffs=s_)))
        for N in (() if only_sqp else (20, 50, 100, 200)):
            sub = [c for c in allc if c["N"] == N]
            e = engine.MpcBatchEngine(0); out = e.run(sub, ch); info = e.launch_info(); e.close()
            compare(f"configs[2] bucket N={N} (64 coefficient sets x 4 weight pairs) {info}", sub, out, pool, strict_steps=400 if N == 200 else None)
        if os.environ.get("PARITY_LEGS", "all") == "sqp":
            allc = []
        else:
            e = engine.MpcBatchEngine(0); out = e.run(allc, ch); info = e.launch_info(); e.close()      # ragged: one launch, the throughput engine
            compare(f"configs[2] 1024 simulations as ONE ragged launch {info}", allc, out, pool, strict_steps=400)
        # configs[3]: full SQP, N = 100, random surfaces (default_rng(2)); 512 of the 4096 simulations, 300 steps, both engines
        rng_c = np.random.default_rng(2)
        base = dict(a=-0.1, b=0.1, c=-0.01, d=0.01, e=0.01, f=0.0)
        sq = [config.resolve_config(config.base_params(prediction_horizon=100, simulation_time=3.0, solver_options={"nlp_solver_type": "SQP"},
                                                        surface_coeffs={k: float(rng_c.normal(v, 0.01)) for k, v in base.items()})) for _ in range(512)]
        for name, env in (("latency engine", None), ("throughput engine", "stream")):
            if env: os.environ["MPCB_ENGINE"] = env
            else: os.environ.pop("MPCB_ENGINE", None)
            e = engine.MpcBatchEngine(0); out = e.run(sq, ch); info = e.launch_info(); e.close()
            compare(f"configs[3] full SQP on the {name} {info}", sq, out, pool, to_first_flag=True)
