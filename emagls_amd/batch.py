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


def batch_cost(n, sim_order):
    """Relative run time of one lane batch of `n` designs laid out for `sim_order` (measured on MI355X, eMagLS2 with 32
    microphones and 1024 taps: 13.6 ms for 8 radii at simulation order 23, 24 ms at order 44): a part that does not depend
    on the number of designs (the resident sweep costs the same for 1..8 designs, the latency chains of the per-bin
    factorisation) and a part that does (G_k of every bin, HBM bound)."""
    S = (int(sim_order) + 1) ** 2
    return (0.45 + 0.55 * n / 8.0) * (1.0 + 7.6e-4 * S)


def padded_lane_batches(sim_orders, max_batch=8):
    """Lane batches across neighbouring simulation-order classes: the jobs sorted by simulation order (stable) and cut into
    ceil(n / max_batch) consecutive chunks of equal size (+-1).  Every design of a chunk is laid out for the chunk's highest
    simulation order (`sim_order_pad` of the plan: b_n = 0 above the design's own order, the same filters), so a chunk has ONE
    shape and runs in lane mode.  Returns [(job indices, pad order), ...], lowest orders first.
    BASELINE config 4 (256 radii on 2..10 cm, 36 classes of 7-8 radii): 32 batches of 8 instead of 36 of 7-8 (or, sharded per
    job, 200 of 1-2)."""
    if not 1 <= max_batch <= 16:
        raise ValueError("a batch holds 1..16 designs")
    n = len(sim_orders)
    if n == 0:
        return []
    order = sorted(range(n), key=lambda i: (sim_orders[i], i))
    nb = -(-n // max_batch)
    base, extra = divmod(n, nb)
    out, pos = [], 0
    for b in range(nb):
        size = base + (1 if b < extra else 0)
        idx = order[pos:pos + size]
        pos += size
        out.append((idx, max(sim_orders[i] for i in idx)))
    return out


def shard_lane_batches(batches, world_size, cost=batch_cost):
    """Whole lane batches to ranks by longest-processing-time on `cost(len(indices), pad order)`.  Returns, per rank, its
    batches in the order it should run them (cheapest first: the first results arrive early) and the rank loads."""
    load = [0.0] * world_size
    shards = [[] for _ in range(world_size)]
    costs = [cost(len(idx), pad) for idx, pad in batches]
    for b in sorted(range(len(batches)), key=lambda b: (-costs[b], b)):
        r = min(range(world_size), key=lambda r: (load[r], r))
        shards[r].append(b)
        load[r] += costs[b]
    return [[batches[b] for b in sorted(s, key=lambda b: (costs[b], b))] for s in shards], load


def _gather_on_rank0(local, shards, n, group, device):
    """`local`: this rank's (wL, wR) list in the order of shards[rank]; one gather brings everything to rank 0 (the only
    collective on the data path).  Returns the list in job order on rank 0, None elsewhere."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
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
    dist.gather(buf, gathered, dst=0, group=group)
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


def run_batch(jobs, design_fn, costs=None, group=None, device=None):
    """Run `design_fn(job) -> (wL, wR)` (equal shapes/dtypes for every job) for this rank's share of `jobs`
    and gather everything on rank 0.  Returns the list of (wL, wR) in job order on rank 0, None elsewhere.
    Works without an initialised process group (single process)."""
    import torch.distributed as dist

    have_pg = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if have_pg else 0
    world = dist.get_world_size(group) if have_pg else 1
    n = len(jobs)
    shards = shard_jobs(costs if costs is not None else np.ones(n), world)
    local = [design_fn(jobs[j]) for j in shards[rank]]
    if world == 1:
        return local
    return _gather_on_rank0(local, shards, n, group, device)


def run_lane_batches(jobs, sim_orders, batch_fn, group=None, device=None, max_batch=8):
    """The class-aware form of `run_batch` for shape-dependent jobs (a sweep over array radii): the jobs are cut into padded
    lane batches (`padded_lane_batches`), whole batches go to ranks (`shard_lane_batches`), and every rank calls
    `batch_fn([jobs of one batch], pad_order) -> [(wL, wR), ...]` for each of its batches.  One gather to rank 0."""
    import torch.distributed as dist

    have_pg = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if have_pg else 0
    world = dist.get_world_size(group) if have_pg else 1
    n = len(jobs)
    per_rank, _ = shard_lane_batches(padded_lane_batches(list(sim_orders), max_batch), world)
    shards = [[j for idx, _ in bl for j in idx] for bl in per_rank]
    local = []
    for idx, pad in per_rank[rank]:
        res = batch_fn([jobs[j] for j in idx], pad)
        if len(res) != len(idx):
            raise ValueError("batch_fn must return one (wL, wR) per job of the batch")
        local += list(res)
    if world == 1:
        out = [None] * n
        for j, r in zip(shards[0], local):
            out[j] = r
        return out
    return _gather_on_rank0(local, shards, n, group, device)


# --------------------------------------------------------------------------------------------
# the two job lists BASELINE.json names, on the plan / batch API (one process per GPU)
# --------------------------------------------------------------------------------------------
def emagls2_radius_sweep(hL, hR, hrirGridAziRad, hrirGridZenRad, radii, micGridAziRad, micGridZenRad, order, fs, length,
                         shDefinition="real", group=None, max_batch=8):
    """getEMagLs2Filters (lib/getEMagLs2Filters.m:1-2) for every array radius of `radii` (BASELINE config 4): padded lane batches,
    whole batches per rank, one gather.  Returns [(wMlsL, wMlsR), ...] in the order of `radii` on rank 0, None elsewhere."""
    from . import Batch, Plan, _lib as L
    hL = np.asfortranarray(hL, dtype=np.float64)
    hR = np.asfortranarray(hR, dtype=np.float64)
    radii = [float(r) for r in radii]
    so = [simulation_order(order, fs, r, raw=True) for r in radii]
    nmics = int(np.asarray(micGridAziRad).size)

    def batch_fn(rs, pad):
        plans = []
        try:
            for r in rs:
                p = Plan(L.KIND_EMAGLS2, shDefinition, int(order), float(fs), int(length), hL.shape[0], hL.shape[1], r, nmics,
                         sim_order_pad=int(pad))
                p.set_hrir_grid(hrirGridAziRad, hrirGridZenRad)
                p.set_mic_grid(micGridAziRad, micGridZenRad)
                p.set_hrirs(hL, hR)
                plans.append(p)
            if len(plans) == 1:
                plans[0].execute()
                return [plans[0].get_filters()]
            b = Batch(plans)
            try:
                b.execute()
                return b.get_filters()
            finally:
                b.close()
        finally:
            for p in plans:
                p.close()
    return run_lane_batches(radii, so, batch_fn, group=group, max_batch=max_batch)


def emagls_from_atf_subjects(subjects, hrirGridAziZenRad, atfIrs, atfGridAziZenRad, fs, filterLen, fTrans, group=None,
                             max_batch=8):
    """getEMagLsFiltersFromAtf (lib/getEMagLsFiltersFromAtf.m:1) for every HRTF subject of `subjects` = [(hL, hR), ...] on ONE
    ATF set and HRIR grid (BASELINE config 5): the subjects are spread over the ranks, each rank runs its share in batches
    that compute the ATF side once (Batch.shares_atf_side) and sweep all their subjects in one resident launch; one gather.
    Returns [(wMlsL, wMlsR), ...] in the order of `subjects` on rank 0, None elsewhere."""
    import torch.distributed as dist
    from . import Batch, Plan, _lib as L
    hg = np.asarray(hrirGridAziZenRad, dtype=np.float64)
    ag = np.asarray(atfGridAziZenRad, dtype=np.float64)
    atf = np.asfortranarray(atfIrs, dtype=np.float64)
    taps, M, Da = atf.shape
    have_pg = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if have_pg else 0
    world = dist.get_world_size(group) if have_pg else 1
    n = len(subjects)
    shards = shard_jobs(np.ones(n), world)
    local = []
    mine = shards[rank]
    for i in range(0, len(mine), max_batch):
        plans = []
        try:
            for j in mine[i:i + max_batch]:
                hL = np.asfortranarray(subjects[j][0], dtype=np.float64)
                hR = np.asfortranarray(subjects[j][1], dtype=np.float64)
                p = Plan(L.KIND_FROM_ATF, "real", 0, float(fs), int(filterLen), hL.shape[0], hL.shape[1], nmics=M, f_trans=float(fTrans),
                         atf_taps=taps, natf=Da)
                p.set_hrir_grid(hg[:, 0], hg[:, 1])
                p.set_hrirs(hL, hR)
                p.set_atfs(atf, ag[:, 0], ag[:, 1])
                plans.append(p)
            if len(plans) == 1:
                plans[0].execute()
                local.append(plans[0].get_filters())
            else:
                b = Batch(plans)
                try:
                    b.execute()
                    local += b.get_filters()
                finally:
                    b.close()
        finally:
            for p in plans:
                p.close()
    if world == 1:
        return local
    return _gather_on_rank0(local, shards, n, group, None)


def emagls_hrir_sets(subjects, hrirGridAziRad, hrirGridZenRad, micRadius, micGridAziRad, micGridZenRad, order, fs, length,
                     shDefinition="real", kind="emagls", group=None, max_batch=8):
    """getEMagLsFilters (lib/getEMagLsFilters.m:32; kind 'emagls2': getEMagLs2Filters, 'emainch': getEMagLsFiltersEMAinCH with
    micGridZenRad None) for every HRIR set of `subjects` = [(hL, hR), ...] on ONE HRIR grid and ONE array: the loop over
    subjects a user of the reference writes around the call.  The sets are spread over the ranks; each rank runs its share in
    batches that compute the geometry stages once (Batch.share_geometry: SH matrices, array model, every bin's regularised
    inverse) and sweep all their sets in one resident launch; one gather.  Same filters as the single calls.
    Returns [(wL, wR), ...] in the order of `subjects` on rank 0, None elsewhere."""
    import torch.distributed as dist
    from . import Batch, Plan, _lib as L
    K = {"emagls": L.KIND_EMAGLS, "emagls2": L.KIND_EMAGLS2, "emainch": L.KIND_EMA_CH}[kind]
    azi = np.asarray(hrirGridAziRad, dtype=np.float64)
    zen = np.asarray(hrirGridZenRad, dtype=np.float64)
    maz = np.asarray(micGridAziRad, dtype=np.float64)
    mzn = None if micGridZenRad is None else np.asarray(micGridZenRad, dtype=np.float64)
    have_pg = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if have_pg else 0
    world = dist.get_world_size(group) if have_pg else 1
    n = len(subjects)
    shards = shard_jobs(np.ones(n), world)
    local = []
    mine = shards[rank]
    for i in range(0, len(mine), max_batch):
        plans = []
        try:
            for j in mine[i:i + max_batch]:
                hL = np.asfortranarray(subjects[j][0], dtype=np.float64)
                hR = np.asfortranarray(subjects[j][1], dtype=np.float64)
                p = Plan(K, shDefinition, int(order), float(fs), int(length), hL.shape[0], hL.shape[1], float(micRadius), maz.size)
                p.set_hrir_grid(azi, zen)
                p.set_mic_grid(maz, mzn)
                p.set_hrirs(hL, hR)
                plans.append(p)
            if len(plans) == 1:
                plans[0].execute()
                local.append(plans[0].get_filters())
            else:
                b = Batch(plans)
                try:
                    b.share_geometry(True)
                    b.execute()
                    local += b.get_filters()
                finally:
                    b.close()
        finally:
            for p in plans:
                p.close()
    if world == 1:
        return local
    return _gather_on_rank0(local, shards, n, group, None)


def magls_hrir_sets(subjects, hrirGridAziRad, hrirGridZenRad, order, fs, length, shDefinition="real", group=None, max_batch=8):
    """getMagLsFilters (lib/getMagLsFilters.m:30; hrirGridZenRad None: getMagLsFilters2D on a horizontal grid) for every HRIR
    set of `subjects` = [(hL, hR), ...] on ONE grid: spread over the ranks, each rank's share in batches that compute the SH
    side once (Batch.share_geometry) and sweep all their sets in one resident launch; one gather.  Same filters as the single
    calls.  Returns [(wL, wR), ...] in the order of `subjects` on rank 0, None elsewhere."""
    import torch.distributed as dist
    from . import Batch, Plan, _lib as L
    azi = np.asarray(hrirGridAziRad, dtype=np.float64)
    zen = None if hrirGridZenRad is None else np.asarray(hrirGridZenRad, dtype=np.float64)
    K = L.KIND_MAGLS if zen is not None else L.KIND_MAGLS_2D
    have_pg = dist.is_available() and dist.is_initialized()
    rank = dist.get_rank(group) if have_pg else 0
    world = dist.get_world_size(group) if have_pg else 1
    n = len(subjects)
    shards = shard_jobs(np.ones(n), world)
    local = []
    mine = shards[rank]
    for i in range(0, len(mine), max_batch):
        plans = []
        try:
            for j in mine[i:i + max_batch]:
                hL = np.asfortranarray(subjects[j][0], dtype=np.float64)
                hR = np.asfortranarray(subjects[j][1], dtype=np.float64)
                p = Plan(K, shDefinition, int(order), float(fs), int(length), hL.shape[0], hL.shape[1], 0.0, 0)
                p.set_hrir_grid(azi, zen)
                p.set_hrirs(hL, hR)
                plans.append(p)
            if len(plans) == 1:
                plans[0].execute()
                local.append(plans[0].get_filters())
            else:
                b = Batch(plans)
                try:
                    b.share_geometry(True)
                    b.execute()
                    local += b.get_filters()
                finally:
                    b.close()
        finally:
            for p in plans:
                p.close()
    if world == 1:
        return local
    return _gather_on_rank0(local, shards, n, group, None)
