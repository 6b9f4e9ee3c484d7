"""Sharding of a grid-search batch over the GPUs of one node and the final results gather.

The simulations of ``run_all`` are independent (simulator.py:654-674 runs them one after the
other with no coupling), so the batch is partitioned with NO data-path collective: every
bucket (same N, Nsim, solver options, robot) is cut into ``world_size`` contiguous chunks, rank
r takes chunk r of every bucket, so all GPUs see the same mix of horizons (cost per simulation
is ~ N * Nsim * iterations).  The single exchange step is the gather of the per-simulation
result logs to rank 0 at the end -- ``torch.distributed.gather`` on the device tensors, which is
RCCL over xGMI with the ``nccl`` backend (and gloo on CPU in the tests).
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import packing


def _dist():
    try:
        import torch.distributed as dist

        return dist if dist.is_available() and dist.is_initialized() else None
    except Exception:
        return None


def is_distributed() -> bool:
    d = _dist()
    return d is not None and d.get_world_size() > 1


def world_size() -> int:
    d = _dist()
    return d.get_world_size() if d else 1


def rank() -> int:
    d = _dist()
    return d.get_rank() if d else 0


def local_device() -> int:
    """GPU index of this process: one process per GPU (LOCAL_RANK from torchrun)."""
    return int(os.environ.get("LOCAL_RANK", "0"))


def chunk_bounds(n: int, parts: int) -> List[Tuple[int, int]]:
    """Contiguous split of range(n) into ``parts`` chunks whose sizes differ by at most one."""
    base, extra = divmod(n, parts)
    out, lo = [], 0
    for r in range(parts):
        hi = lo + base + (1 if r < extra else 0)
        out.append((lo, hi))
        lo = hi
    return out


def group_buckets(resolved: Sequence[Dict], parts: int = 1, merge_ragged: bool = True) -> "OrderedDict[tuple, List[int]]":
    """Queue indices grouped by launch bucket, first-appearance order.

    SQP_RTI buckets that differ only in the prediction horizon are merged into one RAGGED bucket when every one of
    the `parts` ranks then gets at least packing.RAGGED_MIN_BATCH simulations of it: the throughput engine takes the
    horizon per simulation, and a grid search over prediction_horizon (BASELINE configs[2]) then fills the GPU with
    one launch instead of one under-filled launch per horizon.  Inside a ragged bucket the simulations are ordered so
    that every contiguous shard holds the same mix of horizons, longest first (they are dispatched first)."""
    buckets: "OrderedDict[tuple, List[int]]" = OrderedDict()
    for i, c in enumerate(resolved):
        buckets.setdefault(packing.bucket_key(c), []).append(i)
    if not merge_ragged:
        return buckets
    groups: "OrderedDict[tuple, List[tuple]]" = OrderedDict()
    for key, idxs in buckets.items():
        c = resolved[idxs[0]]
        if c["solver_type"] == packing.SOLVER_RTI:
            groups.setdefault(packing.bucket_key(c, ragged=True), []).append(key)
    out: "OrderedDict[tuple, List[int]]" = OrderedDict()
    merged_members = {}
    for rkey, keys in groups.items():
        total = sum(len(buckets[k]) for k in keys)
        if len(keys) > 1 and total // max(parts, 1) >= packing.RAGGED_MIN_BATCH:
            for k in keys:
                merged_members[k] = rkey
    done = set()
    for key, idxs in buckets.items():
        rkey = merged_members.get(key)
        if rkey is None:
            out[key] = idxs
        elif rkey not in done:
            done.add(rkey)
            members = [k for k in groups[rkey]]
            # longest horizon first, then round-robin over the shards so that each contiguous chunk gets the same mix
            allidx = sorted((i for k in members for i in buckets[k]), key=lambda i: -resolved[i]["N"])
            p = max(parts, 1)
            shards = [allidx[r::p] for r in range(p)]
            out[("ragged",) + rkey] = [i for sh in shards for i in sh]
    return out


def gather_to_root(local: Dict[str, object], sizes: List[int], cache: Optional[Dict[str, object]] = None) -> Optional[Dict[str, np.ndarray]]:
    """The one exchange step of a sharded run: gather per-rank arrays [n_r, ...] (n_r = sizes[r]) to rank 0.

    ``local`` holds torch tensors (device tensors from the engine) or numpy arrays.  With the ``nccl`` backend
    (RCCL over xGMI) the DEVICE tensors go into ``dist.gather`` as they are -- no host hop before the collective.
    Rank 0 receives every array into ONE contiguous device buffer [world * max(sizes), ...] (the gather's output list
    are views of it), so the result needs no concatenation on the host, and copies it to pinned host memory on a
    side stream while the next array is being gathered; one wait at the end.  With ``gloo`` (CPU rehearsal) tensors
    are moved to the host first and the gathered buffer IS the result.  The collective is chosen by the backend alone
    (``dist.gather`` exists for both), never per rank or per exception: ranks that disagree on the collective would
    deadlock.  Shards are padded to max(sizes) because gather needs equal shapes; the padding rows are cut on rank 0.
    ``cache`` (a dict the caller keeps between calls of the same shapes) makes rank 0 reuse its gather buffers and pinned
    host arrays instead of allocating them per call -- the returned arrays are then overwritten by the next call."""
    import torch
    import torch.distributed as dist

    ws, me = dist.get_world_size(), dist.get_rank()
    on_device = dist.get_backend() == "nccl"
    dev = torch.device("cuda", local_device()) if on_device else torch.device("cpu")
    nmax = max(sizes)
    uniform = all(n == nmax for n in sizes)
    copy_stream = torch.cuda.Stream(device=dev) if on_device and me == 0 else None
    gathered: Dict[str, "torch.Tensor"] = {}
    for name in sorted(local):
        a = local[name]
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
        if t.device != dev:
            t = t.to(dev)
        if t.shape[0] < nmax:
            pad = torch.zeros((nmax - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
            t = torch.cat([t, pad], dim=0)
        t = t.contiguous()
        big = None
        if me == 0:
            shape = (ws * nmax,) + tuple(t.shape[1:])
            big = cache.get("dev:" + name) if cache is not None else None
            if big is None or tuple(big.shape) != shape or big.dtype != t.dtype:
                big = torch.empty(shape, dtype=t.dtype, device=dev)
                if cache is not None:
                    cache["dev:" + name] = big
        dist.gather(t, list(big.split(nmax, dim=0)) if me == 0 else None, dst=0)
        if me != 0:
            continue
        if not on_device:
            gathered[name] = big
            continue
        # (dist.gather has made the current stream wait for the collective: an event recorded now covers it)
        done = torch.cuda.Event()
        done.record()
        host = cache.get("host:" + name) if cache is not None else None
        if host is None or host.shape != big.shape or host.dtype != big.dtype:
            host = torch.empty(big.shape, dtype=big.dtype, pin_memory=True)
            if cache is not None:
                cache["host:" + name] = host
        copy_stream.wait_event(done)
        with torch.cuda.stream(copy_stream):
            host.copy_(big, non_blocking=True)
        big.record_stream(copy_stream)       # the allocator must not hand `big` out again before the copy has run
        gathered[name] = host
    if me != 0:
        return None
    if copy_stream is not None:
        copy_stream.synchronize()
    out: Dict[str, np.ndarray] = {}
    for name, h in gathered.items():
        arr = h.numpy()
        out[name] = arr if uniform else np.concatenate([arr[r * nmax: r * nmax + sizes[r]] for r in range(ws)], axis=0)
    return out


def agree_ok(check=None, error: Optional[BaseException] = None) -> None:
    """Cross-rank agreement before a collective: ``check`` (an engine -- its sync() is called -- or a callable) runs on
    this rank, then one all-reduce (MIN) of the ok flags; if ANY rank failed every rank raises here, so no rank walks
    into the gather alone and blocks.  ``error``: a failure this rank has already caught.  Without a process group it
    is just the check."""
    err = error
    if err is None and check is not None:
        try:
            check.sync() if hasattr(check, "sync") else check()
        except Exception as e:   # noqa: BLE001 -- reported below, on every rank
            err = e
    d = _dist()
    if d is not None:
        import torch

        on_device = d.get_backend() == "nccl"
        flag = torch.tensor([0 if err is not None else 1], dtype=torch.int32,
                            device=torch.device("cuda", local_device()) if on_device else torch.device("cpu"))
        try:
            d.all_reduce(flag, op=d.ReduceOp.MIN)
        except Exception:   # noqa: BLE001
            # after a sticky HIP error on this device the collective itself throws: what the caller needs is the original failure
            if err is not None:
                raise err
            raise
        if int(flag.item()) == 0 and err is None:
            raise RuntimeError(f"rank {d.get_rank()}: another rank failed before the results gather; stopping with it")
    if err is not None:
        raise err


def complete_runner_output(local: Dict[str, object], cfgs: Sequence[Dict]) -> Dict[str, object]:
    """Runner contract: 13 arrays [batch, ...] (engine.RESULT_FIELDS).  A runner written to the older 11-array contract
    (no ``errors`` / ``plant_time``) is completed here: the task errors are recomputed from the pose / velocity logs
    (analysis.compute_errors, the numpy mirror of simulator.py:265-344) and plant_time is zero.  Anything else that is
    missing is reported up front instead of as a KeyError deep inside the gather."""
    from .engine import RESULT_FIELDS

    need = [n for n, _, _ in RESULT_FIELDS]
    missing = [n for n in need if n not in local]
    if not missing:
        return local
    fixable = {"errors", "plant_time"}
    if not set(missing) <= fixable:
        raise KeyError(f"runner output lacks {sorted(set(missing) - fixable)}; a runner returns {need} (+ optional 'summary')")
    from . import analysis

    out = dict(local)
    if "plant_time" in missing:
        out["plant_time"] = np.zeros(np.shape(to_host(local["solver_time"])))
    if "errors" in missing:
        pose, vel = to_host(local["ee_pose"]), to_host(local["ee_vel"])
        out["errors"] = np.stack([analysis.errors_rows(analysis.compute_errors(pose[i], vel[i], c["coeffs"], c["t_ee"], c["px_ref"],
                                                                               c["vy_ref"])) for i, c in enumerate(cfgs)]) \
            if len(cfgs) else np.zeros((0, 7, pose.shape[-1]))
    return out


def to_host(v, out=None) -> np.ndarray:
    """Device tensor -> numpy through a pinned staging buffer (one DMA at PCIe rate); numpy stays numpy.  ``out``: a pinned
    host tensor of the same shape to copy into (caller-owned result buffer, reused from call to call)."""
    if isinstance(v, np.ndarray):
        return v
    if v.device.type == "cpu":
        return v.numpy()
    import torch

    host = out if out is not None else torch.empty(v.shape, dtype=v.dtype, pin_memory=True)
    host.copy_(v, non_blocking=False)
    return host.numpy()


def run_partitioned(resolved: Sequence[Dict], runner: Callable, chain_for: Callable, use_dist: bool,
                    info: Optional[Dict] = None, only: Optional[Sequence[str]] = None) -> Optional[List[Dict[str, np.ndarray]]]:
    """Run all simulations bucket by bucket (sharded when ``use_dist``) and return one record per
    simulation in queue order (rank 0; None on other ranks).  ``info`` (optional dict) receives
    ``kernel_ms`` (sum over the launches of this rank) and ``d2h_s`` (device -> host copy time).
    ``only``: names of the arrays that leave the device -- ``("summary",)`` is the summary-only mode of SURVEY 8(e): 24 doubles
    per simulation are gathered / copied instead of the 14 result arrays (~262 KB per simulation at 600 steps)."""
    import time

    ws, me = (world_size(), rank()) if use_dist else (1, 0)
    records: List[Optional[Dict[str, np.ndarray]]] = [None] * len(resolved)
    buckets = list(group_buckets(resolved, ws, getattr(runner, "supports_ragged", False)).items())
    kernel_ms, d2h_s = 0.0, 0.0
    # a runner with submit()/collect() launches buckets before the first result is awaited (buckets smaller than
    # the GPU then overlap); a plain callable runs them one after the other.  At most `max_in_flight` buckets hold
    # an engine, a workspace and result buffers at any time.
    max_in_flight = getattr(runner, "max_in_flight", 8)
    tickets: Dict[tuple, object] = {}
    pending = list(buckets)

    def launch_more():
        while pending and len(tickets) < max_in_flight and hasattr(runner, "submit"):
            key, idxs = pending.pop(0)
            lo, hi = chunk_bounds(len(idxs), ws)[me]
            mine = [resolved[i] for i in idxs[lo:hi]]
            if mine:
                tickets[key] = runner.submit(mine, chain_for(resolved[idxs[0]]))
            else:
                tickets[key] = None

    launch_more()
    for key, idxs in buckets:
        bounds = chunk_bounds(len(idxs), ws)
        lo, hi = bounds[me]
        mine = [resolved[i] for i in idxs[lo:hi]]
        failure = None
        try:
            if key in tickets:
                t = tickets.pop(key)
                local = runner.collect(t) if t is not None else None
                launch_more()
            else:
                local = runner(mine, chain_for(resolved[idxs[0]])) if mine else None
            if local is not None:
                local = complete_runner_output(local, mine)
        except Exception as e:   # noqa: BLE001
            if not (use_dist and _dist() is not None):
                raise
            failure, local = e, None
        if local is not None and "_kernel_ms" in local:
            kernel_ms += float(local.pop("_kernel_ms"))
        t0 = time.perf_counter()
        if use_dist and _dist() is not None:   # also a one-rank group: distributed=True asks for the collective path
            agree_ok(error=failure) if failure is not None else agree_ok()   # every rank raises, or none: no lone rank in the gather
            if local is None:  # this rank got no simulation of the bucket: contribute empty arrays
                local = _empty_like_bucket(resolved[idxs[0]], with_summary=True)
            elif "summary" not in local:
                local = dict(local)
                local["summary"] = _host_summary(local, mine)
            if only is not None:
                local = {k: local[k] for k in only}
            full = gather_to_root(local, [b[1] - b[0] for b in bounds])
        else:
            if only is not None and "summary" in only and "summary" not in local:
                local = dict(local)
                local["summary"] = _host_summary(local, mine)
            full = {k: to_host(v) for k, v in local.items() if only is None or k in only}
            if "summary" not in full and only is None:
                full["summary"] = _host_summary(full, mine)
        d2h_s += time.perf_counter() - t0
        if me == 0:
            for pos, qi in enumerate(idxs):
                records[qi] = {k: v[pos] for k, v in full.items()}
    if info is not None:
        info["kernel_ms"] = kernel_ms
        info["d2h_s"] = d2h_s
    return records if me == 0 else None  # type: ignore[return-value]


def _host_summary(local: Dict[str, object], cfgs: Sequence[Dict]) -> np.ndarray:
    """Summary records for a runner that delivers only logs (the engine runner brings its own from the device)."""
    from . import analysis

    a = {k: to_host(v) for k, v in local.items()}
    if len(cfgs) == 0:
        return np.zeros((0, 24))
    return analysis.batch_summary(a["errors"], a["sqp_iter"], a["qp_iter"], a["status"], a["residuals"], a["solver_time"],
                                  a["plant_time"], np.array([c["dt"] for c in cfgs]), np.stack([c["w_task"] for c in cfgs]))


def _empty_like_bucket(cfg: Dict, with_summary: bool = False) -> Dict[str, np.ndarray]:
    from .engine import NSUMMARY, RESULT_FIELDS

    S, T = cfg["Nsim"], cfg["Nsim"] + 1
    out = {name: np.zeros((0,) + shp(S, T), dtype=np.float64 if ty == "f8" else np.int32)
           for name, ty, shp in RESULT_FIELDS}
    if with_summary:
        out["summary"] = np.zeros((0, NSUMMARY))
    return out
