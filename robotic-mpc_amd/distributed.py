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


def group_buckets(resolved: Sequence[Dict]) -> "OrderedDict[tuple, List[int]]":
    """Queue indices grouped by launch bucket, first-appearance order."""
    buckets: "OrderedDict[tuple, List[int]]" = OrderedDict()
    for i, c in enumerate(resolved):
        buckets.setdefault(packing.bucket_key(c), []).append(i)
    return buckets


def _gather_to_root(local: Dict[str, np.ndarray], sizes: List[int]) -> Optional[Dict[str, np.ndarray]]:
    """Gather per-rank arrays [n_r, ...] (n_r = sizes[r]) to rank 0; pads to max(sizes)."""
    import torch
    import torch.distributed as dist

    ws, me = dist.get_world_size(), dist.get_rank()
    backend = dist.get_backend()
    dev = torch.device("cuda", local_device()) if backend == "nccl" else torch.device("cpu")
    nmax = max(sizes)
    out: Dict[str, np.ndarray] = {}
    for name in sorted(local):
        a = local[name]
        t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a))
        t = t.to(dev)
        if t.shape[0] < nmax:
            pad = torch.zeros((nmax - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
            t = torch.cat([t, pad], dim=0)
        t = t.contiguous()
        bufs = [torch.empty_like(t) for _ in range(ws)] if me == 0 else None
        try:
            dist.gather(t, bufs, dst=0)
        except (RuntimeError, NotImplementedError):  # backend without gather: fall back to all_gather
            bufs_all = [torch.empty_like(t) for _ in range(ws)]
            dist.all_gather(bufs_all, t)
            bufs = bufs_all if me == 0 else None
        if me == 0:
            out[name] = np.concatenate([b[: sizes[r]].cpu().numpy() for r, b in enumerate(bufs)], axis=0)
    return out if me == 0 else None


def run_partitioned(resolved: Sequence[Dict], runner: Callable, chain_for: Callable, use_dist: bool
                    ) -> Optional[List[Dict[str, np.ndarray]]]:
    """Run all simulations bucket by bucket (sharded when ``use_dist``) and return one record per
    simulation in queue order (rank 0; None on other ranks)."""
    ws, me = (world_size(), rank()) if use_dist else (1, 0)
    records: List[Optional[Dict[str, np.ndarray]]] = [None] * len(resolved)
    buckets = list(group_buckets(resolved).items())
    # a runner with submit()/collect() launches every bucket before the first result is awaited
    # (buckets smaller than the GPU then overlap); a plain callable runs them one after the other
    tickets = {}
    if hasattr(runner, "submit"):
        for key, idxs in buckets:
            lo, hi = chunk_bounds(len(idxs), ws)[me]
            mine = [resolved[i] for i in idxs[lo:hi]]
            if mine:
                tickets[key] = runner.submit(mine, chain_for(resolved[idxs[0]]))
    for key, idxs in buckets:
        bounds = chunk_bounds(len(idxs), ws)
        lo, hi = bounds[me]
        mine = [resolved[i] for i in idxs[lo:hi]]
        chain = chain_for(resolved[idxs[0]])
        if key in tickets:
            local = runner.collect(tickets.pop(key))
        else:
            local = runner(mine, chain) if mine else None
        if ws > 1:
            if local is None:  # this rank got no simulation of the bucket: contribute empty arrays
                local = _empty_like_bucket(resolved[idxs[0]])
            full = _gather_to_root({k: _np(v) for k, v in local.items()}, [b[1] - b[0] for b in bounds])
        else:
            full = {k: _np(v) for k, v in local.items()}
        if me == 0:
            for pos, qi in enumerate(idxs):
                records[qi] = {k: v[pos] for k, v in full.items()}
    return records if me == 0 else None  # type: ignore[return-value]


def _np(v):
    return v if isinstance(v, np.ndarray) else v.cpu().numpy()


def _empty_like_bucket(cfg: Dict) -> Dict[str, np.ndarray]:
    from .engine import RESULT_FIELDS

    S, T = cfg["Nsim"], cfg["Nsim"] + 1
    return {name: np.zeros((0,) + shp(S, T), dtype=np.float64 if ty == "f8" else np.int32)
            for name, ty, shp in RESULT_FIELDS}
