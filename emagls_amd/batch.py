"""Batches of independent filter-design jobs across the GPUs of one node.

Jobs (one per array radius / HRTF subject / ATF set) share nothing but read-only inputs, so the data path
has no collective: every rank designs its share and ONE gather brings the finished filters to rank 0
(torch.distributed: backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).
"""
from __future__ import annotations

import numpy as np


def shard_jobs(costs, world_size):
    """Longest-processing-time assignment: returns, per rank, the job indices it runs (each list sorted).
    `costs` ~ relative run time of each job (e.g. (simulation order + 1)^2 for a radius sweep)."""
    costs = np.asarray(costs, dtype=np.float64)
    order = np.argsort(-costs, kind="stable")
    load = np.zeros(world_size)
    shards = [[] for _ in range(world_size)]
    for j in order:
        r = int(np.argmin(load))
        shards[r].append(int(j))
        load[r] += costs[j]
    return [sorted(s) for s in shards]


def lane_groups(shape_keys, max_batch=8):
    """Group a rank's jobs into lane batches: a `Batch` runs in lane mode (one launch of every kernel for all its designs,
    one XCD per design in the sweep) only when its plans have identical shapes.  `shape_keys[i]` is any hashable that
    determines the shape of job i -- for a radius sweep `simulation_order(order, fs, radius)` -- and the result is a list of
    index lists of at most `max_batch` (<= 16) jobs each, largest classes first, job order kept inside a class."""
    if not 1 <= max_batch <= 16:
        raise ValueError("a batch holds 1..16 designs")
    classes = {}
    for i, k in enumerate(shape_keys):
        classes.setdefault(k, []).append(i)
    groups = []
    for k in sorted(classes, key=lambda k: (-len(classes[k]), str(k))):
        idx = classes[k]
        # equal-sized batches inside a class (e.g. 9 jobs -> 5 + 4, not 8 + 1: the sweep launch costs the same for 1..8 designs)
        nb = -(-len(idx) // max_batch)
        size = -(-len(idx) // nb)
        groups += [idx[i:i + size] for i in range(0, len(idx), size)]
    return groups


def simulation_order(order, fs, radius, c=343.0, raw=False):
    """max(N, ceil(fs*pi*r/c)) (dependencies/getSMAIRMatrix.m:95): the shape class of an array-radius job.
    raw=True is getEMagLs2Filters, which never sets params.order (lib/getEMagLs2Filters.m:51-63), so that
    getSMAIRMatrix.m:39-41 uses its default 4 in place of `order`."""
    import math
    return max(4 if raw else int(order), int(math.ceil(fs * math.pi * radius / c)))


def run_batch(jobs, design_fn, costs=None, group=None, device=None):
    """Run `design_fn(job) -> (wL, wR)` (equal shapes/dtypes for every job) for this rank's share of `jobs`
    and gather everything on rank 0.  Returns the list of (wL, wR) in job order on rank 0, None elsewhere.
    Works without an initialised process group (single process)."""
    import torch
    import torch.distributed as dist

    have_pg = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if have_pg else 0
    world = dist.get_world_size(group) if have_pg else 1
    n = len(jobs)
    shards = shard_jobs(costs if costs is not None else np.ones(n), world)
    mine = shards[rank]
    local = [design_fn(jobs[j]) for j in mine]
    if world == 1:
        return local
    if device is None:
        device = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    # every rank needs the result shape even when it has no job: agree on it through one tiny collective
    meta = torch.zeros(4, dtype=torch.int64, device=device)
    if local:
        w = np.asarray(local[0][0])
        meta = torch.tensor([w.shape[0], w.shape[1], int(np.iscomplexobj(w)), 1], dtype=torch.int64, device=device)
    dist.all_reduce(meta, op=dist.ReduceOp.MAX, group=group)
    rows, cols, cplx = int(meta[0]), int(meta[1]), bool(meta[2])
    nmax = max(len(s) for s in shards)
    dt = torch.complex128 if cplx else torch.float64
    buf = torch.zeros((nmax, 2, rows, cols), dtype=dt, device=device)
    for i, (wL, wR) in enumerate(local):
        buf[i, 0] = torch.as_tensor(np.ascontiguousarray(wL), device=device)
        buf[i, 1] = torch.as_tensor(np.ascontiguousarray(wR), device=device)
    if cplx:
        buf = torch.view_as_real(buf).contiguous()
    gathered = [torch.zeros_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, gathered, dst=0, group=group)  # the only collective on the data path
    if rank != 0:
        return None
    out = [None] * n
    for r, s in enumerate(shards):
        g = gathered[r]
        if cplx:
            g = torch.view_as_complex(g)
        g = g.cpu().numpy()
        for i, j in enumerate(s):
            out[j] = (g[i, 0], g[i, 1])
    return out
