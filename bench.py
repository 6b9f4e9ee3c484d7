#!/usr/bin/env python3
"""Headline benchmark: MPC-steps/sec of the batched closed-loop rollout (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One bench "step" = one pass of the hot path over one batch: `run_all` of BASELINE.json
configs[1] -- 256 UR10 simulations, prediction horizon N=100, dt=0.01, 6 s (600 closed-loop MPC
steps each), SQP_RTI, flat surface, seeded q_0 jitter -- i.e. 153 600 MPC steps per GPU per
bench step.  Weak scaling: every rank (one per GPU) runs its own batch of 256; the only
exchange is the gather of the result logs to rank 0 (RCCL), inside the timed region.
Parameters are uploaded before the timed region; results stay in HBM (torch tensors).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md)
FP64_VALU_PEAK_TFLOPS = 78.6  # vector fp64


def workload_configs(batch: int, N: int, sim_time: float, seed: int, solver: str):
    """BASELINE.json configs[1] / SURVEY.md 8(d) Config 2."""
    from robotic_mpc_amd import config

    rng = np.random.default_rng(seed)
    flat = dict(a=0.0, b=0.0, c=0.0, d=0.0, e=0.0, f=0.0)
    cfgs = []
    for _ in range(batch):
        q0 = config.BASE_PARAMS["q_0"] + rng.uniform(-0.1, 0.1, 6)
        cfgs.append(config.resolve_config(config.base_params(
            prediction_horizon=N, simulation_time=sim_time, q_0=q0, surface_coeffs=flat,
            solver_options={"nlp_solver_type": solver})))
    return cfgs


def bytes_per_mpc_step(N: int) -> int:
    """Algorithmic HBM bytes of one MPC step (SURVEY.md 8d / BASELINE.md 4): solver state
    54N+24 doubles in and out plus 47 output doubles."""
    return 8 * (108 * N + 95)


def pmc_traffic(batch, N, Nsim, solver):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes of this same workload
    (profiles/r01_pmc_summary.json, written by scripts/profile_gpu.sh); None when absent."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
    try:
        with open(path) as f:
            d = json.load(f)
        if (d.get("batch"), d.get("N"), d.get("Nsim"), d.get("solver")) == (batch, N, Nsim, solver):
            return float(d["hbm_bytes_per_launch"]), "profiles/r01_pmc_summary.json (separate --pmc passes)"
    except (OSError, ValueError, KeyError):
        pass
    return None, None


def cpu_baseline(cfgs, chain, budget_s: float = 20.0):
    """Oracle (plain-C port of the reference algorithm) on the host cores, bounded sample."""
    from oracle import orc

    orc.build()
    rb = orc.make_robot(chain)
    Nsim = cfgs[0]["Nsim"]
    # 1 thread, like the reference's single-threaded acados
    t0 = time.time()
    done = 0
    for c in cfgs:
        orc.run(rb, orc.make_params(c))
        done += 1
        if time.time() - t0 > budget_s / 2 or done >= 4:
            break
    el = time.time() - t0
    one = done * Nsim / el
    out = {"value": one, "unit": "MPC-steps/s", "cores": 1, "kind": "port",
           "sample": f"{done} of the {len(cfgs)} simulations x {Nsim} steps, oracle/libmpc_oracle.so, 1 thread"}
    # all host cores: independent simulations in a process pool (reported as extra fields)
    try:
        import multiprocessing as mp

        ncpu = min(os.cpu_count() or 1, 16)
        sample = cfgs[: max(ncpu, 1)]
        t0 = time.time()
        with mp.get_context("fork").Pool(ncpu) as pool:
            pool.map(_oracle_one, [(chain, c) for c in sample])
        el = time.time() - t0
        out.update({"value_all_cores": len(sample) * Nsim / el, "cores_all": ncpu})
    except Exception as e:  # pragma: no cover
        out["all_cores_error"] = repr(e)
    return out


def _oracle_one(args):
    from oracle import orc

    chain, c = args
    orc.run(orc.make_robot(chain), orc.make_params(c))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--horizon", type=int, default=100)
    ap.add_argument("--sim-time", type=float, default=6.0)
    ap.add_argument("--solver", default="SQP_RTI")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, the product path) | gloo (rehearsal of the "
                    "multi-rank control flow on a box with fewer GPUs than ranks: tensors are gathered via the host)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from robotic_mpc_amd import engine, robots

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 or world > 1:
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} needs one process per GPU (torch.distributed.run), WORLD_SIZE={world}")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "gloo":   # rehearsal: ranks may share a GPU
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
            torch.cuda.set_device(local_rank)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)

    chain = robots.builtin_chain("ur10")
    cfgs = workload_configs(args.batch, args.horizon, args.sim_time, seed=rank, solver=args.solver)
    eng = engine.MpcBatchEngine(local_rank)
    pb = eng.setup(cfgs, chain)            # parameters resident in HBM before the timed region
    bufs = eng.alloc_results(pb)
    gather_note = "n/a"

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def one_pass():
        nonlocal gather_note
        eng.rollout(bufs, 0, pb.Nsim)  # step0 = 0 restarts every simulation from its initial state
        if world > 1 and not args.no_gather:
            try:
                for name in ("z", "u", "status", "cost"):
                    t = bufs[name] if args.backend == "nccl" else bufs[name].cpu()
                    lst = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
                    dist.gather(t, lst, dst=0)
                gather_note = ("rccl" if args.backend == "nccl" else "gloo (via host)") + " gather of z,u,status,cost to rank 0"
            except Exception as e:  # keep the measurement alive if the collective is unavailable
                gather_note = f"gather failed: {e!r}"

    for _ in range(args.warmup):
        one_pass()
    barrier()
    kernel_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_pass()
        if world == 1:
            kernel_ms.append(eng.kernel_ms())  # HIP events on the launch stream (syncs on the kernel)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        kernel_ms.append(eng.kernel_ms())
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    steps_per_pass = args.batch * pb.Nsim
    total_units = steps_per_pass * args.steps * world
    value = total_units / elapsed
    avg_kernel_s = float(np.mean(kernel_ms)) * 1e-3
    algo_bytes = bytes_per_mpc_step(pb.N) * steps_per_pass
    achieved = algo_bytes / avg_kernel_s / 1e9
    qp_iters = float(bufs["qp_iter"].double().mean().item())
    qp_hist = torch.bincount(bufs["qp_iter"].flatten().long().clamp(max=15), minlength=16).tolist()   # last bin: >= 15
    failures = int((bufs["status"] != 0).sum().item())
    # secondary: fp64 VALU estimate: ipm_iters*N*1e4 + n_lin*(N+1)*3e3 flop per MPC step
    flops_step = qp_iters * pb.N * 1.0e4 + 1.0 * (pb.N + 1) * 3.0e3
    info = eng.kernel_info()
    geo = eng.launch_info()
    traffic, traffic_src = pmc_traffic(args.batch, pb.N, pb.Nsim, args.solver)

    line = {
        "metric": "MPC-steps/sec (whole node), UR10 N=100 dt=0.01 batch; 1/2/4/8 GPU",
        "value": value, "unit": "MPC-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1]: batch={args.batch} UR10 sims per GPU, N={pb.N}, dt=0.01, "
                               f"{pb.Nsim} closed-loop steps, {args.solver}, flat surface, q_0 jitter U(-0.1,0.1) rng({rank})",
                   "batch_per_gpu": args.batch, "horizon": pb.N, "closed_loop_steps": pb.Nsim, "solver": args.solver,
                   "parallelism": f"{world} GPU x {args.batch} workgroups (one simulation each) x {geo['waves_per_sim']} wavefronts", "gather": gather_note,
                   "mean_qp_iters_per_step": qp_iters, "qp_iter_histogram_rank0": qp_hist,
                   "ipm_iterations_per_sec": value * qp_iters, "solver_failures": failures},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                     "kernel": "mpc_rollout_kernel", "avg_launch_ms": avg_kernel_s * 1e3,
                     "algorithmic_bytes_per_launch": algo_bytes,
                     "note": "latency / fp64-VALU bound by construction (sequential Riccati), see DESIGN.md",
                     "fp64_valu_est_tflops": flops_step * steps_per_pass / avg_kernel_s / 1e12,
                     "fp64_valu_frac": flops_step * steps_per_pass / avg_kernel_s / 1e12 / FP64_VALU_PEAK_TFLOPS,
                     "vgprs": info["vgprs"], "lds_bytes": info["lds_bytes"], "scratch_bytes": info["scratch_bytes"]},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(cfgs, chain)
    elif rank == 0:
        line["cpu_baseline"] = None
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
