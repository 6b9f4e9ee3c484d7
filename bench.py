#!/usr/bin/env python3
"""Headline benchmark: MPC-steps/sec of the batched closed-loop rollout (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One bench "step" = one pass of the hot path over one batch, as BASELINE.md section 4 defines the metric
(wall of mpcb_run incl. H2D params and D2H/gather of results): for BASELINE.json configs[1] -- 256 UR10
simulations, prediction horizon N=100, dt=0.01, 6 s (600 closed-loop MPC steps each), SQP_RTI, flat surface,
seeded q_0 jitter, i.e. 153 600 MPC steps per GPU per bench step -- the timed region holds

    mpcb_setup (validation, model constants, upload of the 72-double parameter records)  ->  rollout kernel  ->  summary kernel  ->
    [N > 1: gather of every result array to rank 0, RCCL on the device tensors]  ->  D2H of all logs (rank 0: of all ranks)

The synthetic configs (already flattened to parameter records, as Simulator.__init__ would leave them) are resident
in host memory and the device buffers are allocated before the timed region.
The kernel-only rate (HIP events around the rollout launch on its stream) is reported beside it as
`kernel_steps_per_s`; `roofline` is computed from that kernel time.  Weak scaling: every rank (one per GPU) runs
its own batch of 256; the only exchange is the final gather.

The CPU baseline (oracle port, `-O3 -march=native`, built on this host) is measured in a child process that
is started BEFORE this process touches torch / HIP (no fork of a process with a live HIP runtime) and that
waits for the GPU measurement to finish before it starts its own (the two never compete for the host cores).
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
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
    (profiles/r02_pmc_summary.json, written by scripts/profile_gpu.sh); None when absent or when it was
    taken on another workload.  The summary names the commit it was profiled at."""
    for name in ("r04_pmc_summary.json", "r03_pmc_summary.json", "r02_pmc_summary.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
            if (d.get("batch"), d.get("N"), d.get("Nsim"), d.get("solver")) == (batch, N, Nsim, solver):
                return float(d["hbm_bytes_per_launch"]), f"profiles/{name} (separate --pmc passes, profiled_at_commit {d.get('commit', '?')})"
        except (OSError, ValueError, KeyError):
            pass
    return None, None


# ------------------------------------------------------------------------------------------- CPU baseline (child)
def cpu_baseline_main(argv):
    """Child-process entry: `bench.py --cpu-baseline-child batch N sim_time solver budget_s`.  Builds the oracle
    with -O3 -march=native on THIS host and times it on a bounded sample of the bench workload.  Prints JSON."""
    batch, N, sim_time, solver, budget = int(argv[0]), int(argv[1]), float(argv[2]), argv[3], float(argv[4])
    import tempfile

    sys.stdin.readline()      # the parent says "go" once its GPU measurement is over: the two never share the host cores

    from oracle import orc
    from robotic_mpc_amd import robots

    flags = ["-O3", "-march=native", "-ffp-contract=off"]
    lib = os.path.join(tempfile.gettempdir(), f"libmpc_oracle_native_{os.getpid()}.so")
    subprocess.check_call(["gcc", *flags, "-fPIC", "-std=c11", "-shared", "-o", lib, os.path.join(ROOT, "oracle", "mpc_oracle.c"), "-lm"])
    orc._LIB_PATH = lib
    orc._lib = None
    chain = robots.builtin_chain("ur10")
    cfgs = workload_configs(batch, N, sim_time, seed=0, solver=solver)
    rb = orc.make_robot(chain)
    Nsim = cfgs[0]["Nsim"]

    def timed(fast, limit):
        t0 = time.time()
        done = 0
        for c in cfgs:     # 1 thread, like the reference's single-threaded acados
            orc.run(rb, orc.make_params({**c, "qp_fast_path": fast}))
            done += 1
            if time.time() - t0 > limit or done >= 8:
                break
        return done, done * Nsim / (time.time() - t0)

    # `value`: the REFERENCE's algorithm -- every QP through the HPIPM-style interior-point loop (qp_fast_path off), which is what
    # solver.solve() does (simulator.py:212); `value_fast_path`: the oracle with the same fast path the GPU engine runs by default
    done, rate = timed(0, budget / 3)
    done_f, rate_f = timed(1, budget / 6)
    out = {"value": rate, "unit": "MPC-steps/s", "cores": 1, "kind": "port",
           "sample": f"{done} of the {len(cfgs)} simulations x {Nsim} steps of the bench workload, oracle/mpc_oracle.c "
                     f"(dense C restatement of the reference algorithm: every QP through the interior-point loop) built here with gcc "
                     f"{' '.join(flags)}, 1 thread",
           "flags": " ".join(flags), "value_fast_path": rate_f,
           "value_fast_path_note": f"the same oracle with the bound-inactive fast path of the QP solve (the GPU engine's default), {done_f} simulations, 1 thread"}
    import multiprocessing as mp

    ncpu = min(os.cpu_count() or 1, 16)
    sample = cfgs[:ncpu]
    t0 = time.time()
    with mp.get_context("fork").Pool(ncpu, initializer=_oracle_init, initargs=(lib,)) as pool:
        pool.map(_oracle_one, [(chain, {**c, "qp_fast_path": 0}) for c in sample])
    out.update({"value_all_cores": len(sample) * Nsim / (time.time() - t0), "cores_all": ncpu})
    try:
        os.unlink(lib)
    except OSError:
        pass
    print("CPU_BASELINE " + json.dumps(out), flush=True)


def _oracle_init(lib):
    from oracle import orc

    orc._LIB_PATH = lib
    orc._lib = None


def _oracle_one(args):
    from oracle import orc

    chain, c = args
    orc.run(orc.make_robot(chain), orc.make_params(c))
    return 0


def start_cpu_baseline(args):
    """Launch the child BEFORE torch / HIP are touched; its result is collected after the GPU measurement."""
    return subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", str(args.batch), str(args.horizon),
                             str(args.sim_time), args.solver, "20"], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                            stderr=subprocess.PIPE, text=True)


def finish_cpu_baseline(proc):
    try:
        out, err = proc.communicate("go\n", timeout=240)
    except subprocess.TimeoutExpired:
        proc.kill()
        return {"error": "cpu baseline child timed out"}
    for line in out.splitlines():
        if line.startswith("CPU_BASELINE "):
            return json.loads(line[len("CPU_BASELINE "):])
    return {"error": "cpu baseline child failed", "stderr": err[-400:]}


# ------------------------------------------------------------------------------------------- self launch
def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: run `python -m torch.distributed.run --nproc-per-node N bench.py ...`
    as a CHILD process and forward its output and exit code.  This parent has not imported torch or touched HIP (a
    process that has initialised the GPU must never exec or fork GPU work), and it only waits."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    # --standalone: the launcher binds its own rendezvous port (no probe-then-reuse race with other processes on the box)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={n}", os.path.abspath(__file__), *sys.argv[1:]]
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:           # rank 0's JSON line (and nothing else of ours) arrives here
        sys.stdout.write(line)
        sys.stdout.flush()
    rc = proc.wait()
    if rc != 0:
        print(f"bench.py: the {n}-rank child launch exited with {rc}", file=sys.stderr)
    return rc


# ------------------------------------------------------------------------------------------- main
def stream_roofline(kernel_ms: float, B: int, N: int, Nsim: int):
    """roofline block of the throughput engine's kernel for the `secondary` record: algorithmic bytes as for the headline
    (SURVEY 8d), traffic from the committed PMC passes of the same workload scaled to this launch's step count."""
    algo = bytes_per_mpc_step(N) * B * Nsim
    achieved = algo / (kernel_ms * 1e-3) / 1e9
    traffic = src = None
    for name in ("r04_stream_b4096_pmc.json", "r03_stream_b4096_pmc.json", "r02_stream_b4096_pmc.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                d = json.load(f)
            k = next(v for kk, v in d.items() if isinstance(v, dict) and "hbm_bytes_per_launch" in v)
            wl = d.get("workload", {"batch": 4096, "N": 100, "Nsim": 100})
            if (wl["batch"], wl["N"]) == (B, N):
                # bytes per IPM iteration are what the passes move; per launch they scale with the iteration count, which the
                # profile records -- quoted per closed-loop step of the profiled run, times this launch's steps
                traffic = float(k["hbm_bytes_per_launch"]) / wl["Nsim"] * Nsim
                src = f"profiles/{name} ({wl['Nsim']} profiled steps scaled to {Nsim}; separate --pmc passes, 2*FETCH+WRITE)"
                break
        except (OSError, ValueError, KeyError, StopIteration):
            continue
    return {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic, "traffic_source": src, "kernel": "mpc_stream_kernel<double>", "avg_launch_ms": kernel_ms,
            "algorithmic_bytes_per_launch": algo}


def throughput_leg(eng, chain, args):
    """Secondary figure (NOT `value`): kernel rate of the throughput engine at batch 4096, N=100, 600 closed-loop steps,
    SQP_RTI (one work-queue launch after one warm-up launch of 60 steps), measured with HIP events like the roofline."""
    import torch

    B = 4096
    cfgs = workload_configs(B, args.horizon, args.sim_time, seed=1, solver="SQP_RTI")
    pb, params, robot = eng.prepare(cfgs, chain)
    eng.setup_packed(pb, params, robot)
    bufs = eng.alloc_results(pb)
    eng.rollout(bufs, 0, min(60, pb.Nsim))
    eng.sync()
    eng.rollout(bufs, 0, pb.Nsim)
    eng.sync()
    ms = eng.kernel_ms()
    geo = eng.launch_info()
    out = {"workload": f"batch={B} UR10 sims on one GPU, N={pb.N}, {pb.Nsim} closed-loop steps, SQP_RTI, flat surface, q_0 jitter rng(1)",
           "engine": "throughput (one wavefront per simulation, work-queue launch)" if geo["engine"] == 1 else "latency",
           "kernel_ms": ms, "kernel_steps_per_s": B * pb.Nsim / (ms * 1e-3),
           "mean_qp_iters_per_step": float(bufs["qp_iter"].double().mean().item()),
           "solver_failures": int((bufs["status"] != 0).sum().item()),
           "roofline": stream_roofline(ms, B, pb.N, pb.Nsim)}
    del bufs
    torch.cuda.empty_cache()
    return out


def config2_surface_sets(n):
    """SURVEY 8(d) Config 3 / examples/surface_stats.ipynb cells 1+7: every coefficient ~ N(mean, 0.01) around
    {a:-0.1, b:0.1, c:-0.01, d:0.01, e:0.01, f:0}, np.random.seed(42), keys in dict order a..f per set."""
    base = dict(a=-0.1, b=0.1, c=-0.01, d=0.01, e=0.01, f=0.0)
    rs = np.random.RandomState(42)
    return [{k: float(rs.normal(v, 0.01)) for k, v in base.items()} for _ in range(n)]


def config2_main(args, world, rank, multi, cpu_proc):
    """BASELINE configs[2] as the reference's user would run it: SimulationManager.grid_search over
    {prediction_horizon: [20, 50, 100, 200], w_qddot: [0.02, 0.05], w_u: [0.01, 0.001]} x coefficient sets, then run_all --
    sharded over the ranks by distributed.run_partitioned (every rank the same mix of horizons), results gathered to rank 0.
    One bench step = one run_all of the whole grid (config resolution, packing, launches, gather, D2H): STRONG scaling."""
    import torch
    import torch.distributed as dist

    from robotic_mpc_amd import SimulationManager, base_params

    m = SimulationManager(base_params(simulation_time=args.sim_time, solver_options={"nlp_solver_type": args.solver}))
    m.grid_search({"prediction_horizon": [20, 50, 100, 200], "w_qddot": [0.02, 0.05], "w_u": [0.01, 0.001]},
                  surface_coeff_sets=config2_surface_sets(args.coeff_sets))
    n_sims = len(m.simulations)
    Nsim = int(args.sim_time / 0.01)

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def one_pass():
        return m.run_all(distributed=multi, results=args.results)

    for _ in range(args.warmup):
        one_pass()
    barrier()
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(args.steps):
        res = one_pass()
        kernel_ms.append(m.last_run_info["kernel_ms"])
    barrier()
    elapsed = time.perf_counter() - t0
    if multi:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    if rank == 0:
        info = m.last_run_info
        fails = sum(int(s["num_failures"]) for s in m.last_summaries)
        line = {
            "metric": "MPC-steps/sec (whole node), UR10 N=100 dt=0.01 batch; 1/2/4/8 GPU",
            "value": n_sims * Nsim * args.steps / elapsed, "unit": "MPC-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[2]: grid_search {{N in [20,50,100,200], w_qddot in [0.02,0.05], w_u in [0.01,0.001]}} x "
                                   f"{args.coeff_sets} surface coefficient sets = {n_sims} simulations x {Nsim} steps, {args.solver}, "
                                   f"through SimulationManager.run_all(results='{args.results}'), sharded over {world} rank(s)",
                       "simulations": n_sims, "closed_loop_steps": Nsim, "buckets": info.get("buckets"), "results": args.results,
                       "timed_region": "run_all: Simulator(**config) x n -> parameter records -> launches of this rank's shard of every "
                                       "bucket -> summary kernels -> gather to rank 0 -> D2H",
                       "rank0_kernel_ms_per_pass": float(np.mean(kernel_ms)), "rank0_setup_s": info.get("setup_s"),
                       "rank0_run_s": info.get("run_s"), "rank0_d2h_s": info.get("d2h_s"), "solver_failures": fails,
                       "returned": len(res) if res is not None else 0},
            "roofline": None,
            "cpu_baseline": finish_cpu_baseline(cpu_proc) if cpu_proc is not None else None,
        }
        print(json.dumps(line), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-baseline-child":
        return cpu_baseline_main(sys.argv[2:])
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--horizon", type=int, default=100)
    ap.add_argument("--sim-time", type=float, default=6.0)
    ap.add_argument("--solver", default="SQP_RTI")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-only", action="store_true", help="time the rollout alone (parameters resident, results stay in HBM)")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, the product path) | gloo (rehearsal of the "
                    "multi-rank control flow on a box with fewer GPUs than ranks: tensors are gathered via the host)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the extra (untimed-region) throughput-geometry measurement")
    ap.add_argument("--force-dist", action="store_true", help="initialise the process group and take the gather path even with "
                    "one rank (rehearsal of the RCCL code path on a one-GPU box; launch under torch.distributed.run)")
    ap.add_argument("--workload", default="config1", choices=["config1", "config2"],
                    help="config1 (default, the driver's line): BASELINE configs[1], --batch simulations PER GPU (weak scaling); "
                         "config2: BASELINE configs[2], the 4096-simulation grid search through SimulationManager.grid_search / run_all, "
                         "sharded over the ranks (strong scaling)")
    ap.add_argument("--results", default="full", choices=["full", "summary"], help="config2: what run_all gathers to rank 0")
    ap.add_argument("--coeff-sets", type=int, default=256, help="config2: surface coefficient sets (x 16 grid points)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus)      # plain `python bench.py --gpus N`: start the N ranks ourselves (no torch / HIP touched yet)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    cpu_proc = None
    if rank == 0 and not args.no_cpu_baseline:
        cpu_proc = start_cpu_baseline(args)      # before the first torch / HIP call of this process

    import torch
    import torch.distributed as dist

    from robotic_mpc_amd import distributed as dmod, engine, robots

    multi = args.gpus > 1 or world > 1 or args.force_dist
    if multi:
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} needs one process per GPU (torch.distributed.run), WORLD_SIZE={world}")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.backend == "gloo":   # rehearsal: ranks may share a GPU
            local_rank = local_rank % max(torch.cuda.device_count(), 1)
            os.environ["LOCAL_RANK"] = str(local_rank)
            torch.cuda.set_device(local_rank)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)

    if args.workload == "config2":
        return config2_main(args, world, rank, multi, cpu_proc)
    chain = robots.builtin_chain("ur10")
    cfgs = workload_configs(args.batch, args.horizon, args.sim_time, seed=rank, solver=args.solver)
    eng = engine.MpcBatchEngine(local_rank)
    pb, params, robot = eng.prepare(cfgs, chain)   # config dicts -> parameter records (host; Simulator.__init__'s dict handling)
    eng.setup_packed(pb, params, robot)            # sizes the workspace (allocation is not part of a pass)
    bufs = eng.alloc_results(pb)
    sizes = [args.batch] * world
    # caller-owned host result buffers (pinned), allocated once like the device ones: allocation is not part of a pass
    host_bufs = {k: torch.empty(v.shape, dtype=v.dtype, pin_memory=True) for k, v in bufs.items()}
    host_bufs["summary"] = torch.empty((args.batch, engine.NSUMMARY), dtype=torch.float64, pin_memory=True)
    gather_cache = {}

    def barrier():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    def one_pass():
        """mpcb_run as BASELINE.md section 4 defines it; returns the host arrays on rank 0."""
        if args.kernel_only:
            eng.rollout(bufs, 0, pb.Nsim)
            eng.sync()                        # mpcb_sync also reports a work-queue hand-off that did not complete
            return None
        eng.setup_packed(pb, params, robot)   # mpcb_setup: validate, derive model constants, upload the records (H2D)
        eng.rollout(bufs, 0, pb.Nsim)      # step0 = 0 restarts every simulation from its initial state
        local = dict(bufs)
        local["summary"] = eng.summary(bufs)
        if multi:
            dmod.agree_ok(eng)                # mpcb_sync on every rank + one all-reduce of the verdict: all ranks raise together
            out = dmod.gather_to_root(local, sizes, gather_cache)   # product path of run_all: device tensors -> RCCL -> one D2H
            torch.cuda.current_stream().synchronize()         # a pass ends when this rank's results have left its buffers
            return out
        host = {k: dmod.to_host(v, host_bufs[k]) for k, v in local.items()}  # (a failed collective raises: the run exits non-zero)
        eng.sync()                            # the copies have waited for the kernel; this reads the launch's error flag
        return host

    for _ in range(args.warmup):
        one_pass()
    barrier()
    kernel_ms = []
    host = None
    pass_ms = []
    # (like timeit: no cyclic garbage collection inside the timed region -- a generation-2 sweep of this interpreter, torch loaded, is
    # ~40 ms, 40 % of a pass, and whether one lands inside the three timed passes depends on how many objects the imports created)
    import gc
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tp = time.perf_counter()
        host = None                        # (frees the previous pass's pinned arrays before the next ones are taken)
        host = one_pass()
        kernel_ms.append(eng.kernel_ms())  # HIP events on the launch stream (the pass has already waited for the kernel,
        pass_ms.append((time.perf_counter() - tp) * 1e3)
    barrier()                              # except with --kernel-only, where this wait is the only sync)
    elapsed = time.perf_counter() - t0
    gc.enable()
    if multi:
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
    if args.kernel_only:
        region = "rollout kernel only (parameters resident, results left in HBM)"
    elif multi:
        region = ("per rank: mpcb_setup (H2D params) -> rollout -> summary kernel -> "
                  f"{'RCCL' if args.backend == 'nccl' else 'gloo (via host)'} gather of all {len(bufs) + 1} result arrays to rank 0 -> D2H on rank 0")
    else:
        region = "mpcb_setup (H2D params) -> rollout -> summary kernel -> D2H of all result arrays (pinned)"
    d2h_mb = None if host is None else sum(v.nbytes for v in host.values()) / 1e6

    line = {
        "metric": "MPC-steps/sec (whole node), UR10 N=100 dt=0.01 batch; 1/2/4/8 GPU",
        "value": value, "unit": "MPC-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1]: batch={args.batch} UR10 sims per GPU, N={pb.N}, dt=0.01, "
                               f"{pb.Nsim} closed-loop steps, {args.solver}, flat surface, q_0 jitter U(-0.1,0.1) rng({rank})",
                   "batch_per_gpu": args.batch, "horizon": pb.N, "closed_loop_steps": pb.Nsim, "solver": args.solver,
                   "parallelism": f"{world} GPU x {args.batch} workgroups (one simulation each) x {geo['waves_per_sim']} wavefronts",
                   "timed_region": region, "host_bytes_per_pass_MB": d2h_mb,
                   "kernel_steps_per_s": steps_per_pass / avg_kernel_s, "pass_ms_rank0": pass_ms, "kernel_ms_rank0": kernel_ms,
                   "mean_qp_iters_per_step": qp_iters, "qp_iter_histogram_rank0": qp_hist,
                   "qp_iter_meaning": "Riccati factorisations per MPC step: 1 = the bound-inactive fast path solved the QP outright "
                                      "(csrc/mpc_ipm.h); otherwise interior-point iterations (+1 for a rejected attempt)",
                   "fast_path_step_fraction": float((bufs["qp_iter"] == 1).double().mean().item()),
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
    if rank == 0 and not multi and not args.no_secondary:
        # outside the timed region, for the record: the throughput geometry in the same process
        line["secondary"] = throughput_leg(eng, chain, args)
    if rank == 0:
        line["cpu_baseline"] = finish_cpu_baseline(cpu_proc) if cpu_proc is not None else None
        print(json.dumps(line), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main() or 0)
