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
    index lists of at most `max_batch` (<= 32) jobs each, largest classes first, job order kept inside a class."""
    if not 1 <= max_batch <= 32:
        raise ValueError("a batch holds 1..32 designs")
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
    """Run time in ms of one lane batch of `n` designs laid out for `sim_order`, as measured on MI355X at the end of round 4
    (eMagLS2, 32 microphones, order 4, 48 kHz, 1024 taps: every lane batch of BASELINE config 4's 256 radii alone on the GPU,
    tools/experiments/config4_costs.py): a part per launch sequence and a part per design, both growing with the simulated SH
    channels -- 10.1 ... 16.1 ms for 8 designs and 13.7 ... 22.4 ms for 16 between simulation orders 18 and 44 (the resident sweep
    costs 16 designs 1.3 times what it costs 8: lane batches of 16 do the job list in 0.72 of the time of batches of 8).  Below
    simulation order ~19 the designs of this shape keep materialised sweep operands (their ill-conditioned low bins reach beyond
    k_cut: `sweep_form` 1) and cost 12 - 13 ms per 8, 18 - 21 ms per 16, whatever the order.  Only the ratios matter
    (shard_lane_batches)."""
    S = (int(sim_order) + 1) ** 2
    g = n / 8.0
    if int(sim_order) < 19:
        return 5.6 + 7.2 * g
    return (6.4 + 2.4 * g) + (1.67 + 1.93 * g) * 1e-3 * S


def padded_lane_batches(sim_orders, max_batch=16, balance=True):
    """Lane batches across neighbouring simulation-order classes: the jobs sorted by simulation order (stable) and cut into
    ceil(n / max_batch) consecutive chunks.  Every design of a chunk is laid out for the chunk's highest simulation order
    (`sim_order_pad` of the plan: b_n = 0 above the design's own order, the same filters), so a chunk has ONE shape and runs in
    lane mode.  Returns [(job indices, pad order), ...], lowest orders first.
    balance=True (default): the chunks are cut at equal COST, not equal size -- the smallest bound on `batch_cost(size, highest
    order)` that needs no more than ceil(n / max_batch) chunks, so chunks of high orders hold fewer designs (never more than 32, the
    library's limit; `max_batch` is then the AVERAGE chunk size).  Whole chunks go to ranks (shard_lane_batches): with chunks of
    equal size the most loaded of 8 ranks carried 8 % more than the least loaded one on BASELINE config 4; with chunks of equal
    cost 2 %.  balance=False: chunks of equal size (+-1).
    BASELINE config 4 (256 radii on 2..10 cm, 36 classes of 7-8 radii): 16 batches (32 with max_batch = 8) instead of 36 of 7-8
    (or, sharded per job, 200 of 1-2).  A batch of more than 8 designs needs emagls_set_batch_max (the job lists of this module
    go through emagls_jobs_run, which raises it for the call)."""
    if not 1 <= max_batch <= 32:
        raise ValueError("a batch holds 1..32 designs")
    n = len(sim_orders)
    if n == 0:
        return []
    order = sorted(range(n), key=lambda i: (sim_orders[i], i))
    nb = -(-n // max_batch)
    out, pos = [], 0
    if balance and nb > 1:
        so = [sim_orders[i] for i in order]

        def cut(T):
            """greedy chunks of cost <= T each (a chunk's cost: batch_cost at its size and its highest order); None if a single design exceeds T"""
            chunks, pos = [], 0
            while pos < n:
                size = 1
                if batch_cost(1, so[pos]) > T:
                    return None
                while pos + size < n and size < 32 and batch_cost(size + 1, so[pos + size]) <= T:
                    size += 1
                chunks.append((pos, size))
                pos += size
            return chunks
        lo, hi = 0.0, batch_cost(32, max(so)) + 1.0
        for _ in range(40):   # the smallest cost bound that needs no more than nb chunks
            mid = 0.5 * (lo + hi)
            c = cut(mid)
            if c is not None and len(c) <= nb:
                hi = mid
            else:
                lo = mid
        for pos, size in cut(hi):
            idx = order[pos:pos + size]
            out.append((idx, max(sim_orders[i] for i in idx)))
        return out
    base, extra = divmod(n, nb)
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


def _pg(group):
    """(have a process group, rank, world size)"""
    import torch.distributed as dist
    have = dist.is_available() and dist.is_initialized()
    return have, (dist.get_rank(group) if have else 0), (dist.get_world_size(group) if have else 1)


def _collective_device(group, device=None):
    import torch.distributed as dist
    if device is not None:
        return device
    return "cuda" if dist.get_backend(group) == "nccl" else "cpu"


def _agree_or_raise(err, group, device=None, force=False):
    """Every rank enters the gather or none does: the ranks agree on an error flag first (one 8-byte all-reduce).  A rank whose
    share failed (an unsupported shape, out of memory) raises its own exception, the others a RuntimeError that names the
    failing rank -- nobody is left waiting inside a collective."""
    import torch
    import torch.distributed as dist
    have, rank, world = _pg(group)
    if not have or (world == 1 and not force):
        if err is not None:
            raise err
        return
    flag = torch.tensor([rank + 1 if err is not None else 0], dtype=torch.int64, device=_collective_device(group, device))
    dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=group)
    if err is not None:
        raise err
    if int(flag.item()):
        raise RuntimeError("rank %d failed in its share of the job list; no rank entered the gather" % (int(flag.item()) - 1))


def _gather_on_rank0(local, shards, n, group, device):
    """`local`: this rank's (wL, wR) list in the order of shards[rank] as host arrays (the results of a caller-supplied design
    function); one gather brings everything to rank 0 (the only collective on the data path).  Returns the list in job order on
    rank 0, None elsewhere.  (The job lists of this module that run on plans keep their filters on the device: `_Results`.)"""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    device = _collective_device(group, device)
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
    have_pg, rank, world = _pg(group)
    n = len(jobs)
    shards = shard_jobs(costs if costs is not None else np.ones(n), world)
    local, err = [], None
    try:
        local = [design_fn(jobs[j]) for j in shards[rank]]
    except Exception as e:   # (agreed on with the other ranks before anybody enters the gather)
        err = e
    _agree_or_raise(err, group, device)
    if world == 1:
        return local
    return _gather_on_rank0(local, shards, n, group, device)


def run_lane_batches(jobs, sim_orders, batch_fn, group=None, device=None, max_batch=8):
    """The class-aware form of `run_batch` for shape-dependent jobs (a sweep over array radii): the jobs are cut into padded
    lane batches (`padded_lane_batches`), whole batches go to ranks (`shard_lane_batches`), and every rank calls
    `batch_fn([jobs of one batch], pad_order) -> [(wL, wR), ...]` for each of its batches.  One gather to rank 0."""
    have_pg, rank, world = _pg(group)
    n = len(jobs)
    per_rank, _ = shard_lane_batches(padded_lane_batches(list(sim_orders), max_batch), world)
    shards = [[j for idx, _ in bl for j in idx] for bl in per_rank]
    local, err = [], None
    try:
        for idx, pad in per_rank[rank]:
            res = batch_fn([jobs[j] for j in idx], pad)
            if len(res) != len(idx):
                raise ValueError("batch_fn must return one (wL, wR) per job of the batch")
            local += list(res)
    except Exception as e:
        err = e
    _agree_or_raise(err, group, device)
    if world == 1:
        out = [None] * n
        for j, r in zip(shards[0], local):
            out[j] = r
        return out
    return _gather_on_rank0(local, shards, n, group, device)


# --------------------------------------------------------------------------------------------
# job lists on the plan / batch API (one process per GPU): the filters stay where the gather wants them
# --------------------------------------------------------------------------------------------
class _Results:
    """The finished filters of one rank, slot i = the i-th job of its share, in ONE buffer laid out for the gather:
    [slots][ear][column][row] (the library's column-major len x channels matrices), complex as (re, im) pairs.  With RCCL the
    buffer is a device tensor that the library writes device-to-device (emagls_batch_get_filters / emagls_plan_get_filters take
    device addresses) and the gather reads directly -- no host round trip; with gloo, or without a process group, a host array."""

    def __init__(self, nslots, on_device):
        self.nslots, self.on_device, self.buf, self.shape = int(nslots), bool(on_device), None, None

    def ensure(self, rows, cols, cplx):
        if self.buf is not None:
            if self.shape != (rows, cols, cplx):
                raise ValueError("the jobs of one list must give filters of one shape")
            return
        self.shape = (int(rows), int(cols), bool(cplx))
        dims = (max(self.nslots, 1), 2, int(cols), int(rows)) + ((2,) if cplx else ())
        if self.on_device:
            import torch
            # (the library fills the buffer device to device on its own streams: nothing of torch's may still be writing it)
            self.buf = torch.zeros(dims, dtype=torch.float64, device="cuda")
            torch.cuda.current_stream().synchronize()
        else:
            self.buf = np.zeros(dims, dtype=np.float64)

    def _addr(self, i, ear):
        if self.on_device:
            return self.buf[i, ear].data_ptr()
        return self.buf[i, ear].ctypes.data

    def ptrs(self, first, count):
        return [self._addr(first + j, 0) for j in range(count)], [self._addr(first + j, 1) for j in range(count)]

    @staticmethod
    def unpack(block, cplx):
        """[ear][column][row](re, im) of one job as a host array -> (wL, wR), len x channels like the single calls return."""
        if cplx:
            block = block[..., 0] + 1j * block[..., 1]
        return np.asfortranarray(block[0].T), np.asfortranarray(block[1].T)


def _out_shape(job):
    """(rows, columns, complex?) of a job's filters, from its descriptor alone (emagls_design_out_shape: no plan, no device
    memory)."""
    import ctypes as C
    from . import _lib as L
    kw = dict(job)
    nsamp, ndirs = np.asarray(kw["hL"]).shape
    atf = kw.get("atf")
    nmics = np.asarray(atf).shape[1] if atf is not None else (0 if kw.get("mic_azi") is None else int(np.asarray(kw["mic_azi"]).size))
    desc = L.DesignDesc(int(kw["kind"]), L.BASIS[kw["basis"]], int(kw["order"]), float(kw["fs"]), int(kw["length"]), int(nsamp), int(ndirs),
                        float(kw.get("mic_radius", 0.0)), int(nmics), float(kw.get("f_trans", 0.0)), 0, 0, 0, 0, int(kw.get("sim_order_pad", 0)))
    rows, cols, cplx = C.c_int64(0), C.c_int64(0), C.c_int(0)
    L.check(L.load().emagls_design_out_shape(C.byref(desc), C.byref(rows), C.byref(cols), C.byref(cplx)))
    return int(rows.value), int(cols.value), bool(cplx.value)


def _run_share(jobs, res, max_batch, share_geometry=False):
    """This rank's share of a job list on the library's scheduler (emagls_jobs_run: chunks of one shape as lane batches, several
    chunks in flight from the library's own threads).  `jobs` = keyword dictionaries of jobs.JobList.add; the filters go straight
    into `res` (slot i = job i of the share), device to device when `res` lives on the GPU."""
    from .jobs import JobList
    if not jobs:
        return
    res.ensure(*_out_shape(jobs[0]))
    jl = JobList()
    for i, kw in enumerate(jobs):
        pl, pr = res.ptrs(i, 1)
        jl.add(out=(pl[0], pr[0]), **kw)
    jl.run(batch_size=max_batch, in_flight=4, share_geometry=share_geometry)


def _run_job_list(n, shards, make_job, group=None, max_batch=16, share_geometry=False):
    """The shared loop of the job lists below.  `shards[r]` = the job indices of rank r in the order it runs them (jobs of one shape
    next to each other: the library cuts chunks where the shape changes); `make_job(j)` = the keyword dictionary of job j.  Every
    rank runs its share through the library's scheduler, the ranks agree that nobody failed, ONE gather of the device buffers
    brings the filters to rank 0.  Returns [(wL, wR), ...] in job order on rank 0 (and in a single process), None elsewhere."""
    import torch
    import torch.distributed as dist
    import os
    have_pg, rank, world = _pg(group)
    # (EMAGLS_FORCE_COLLECTIVE=1: a single rank goes through the agreement, the device buffers and the gather as well -- how the
    # RCCL path of this loop is exercised on a box with one GPU)
    force = have_pg and os.environ.get("EMAGLS_FORCE_COLLECTIVE", "0") == "1"
    on_device = have_pg and (world > 1 or force) and dist.get_backend(group) == "nccl"
    res = _Results(len(shards[rank]), on_device)
    err = None
    try:
        _run_share([make_job(j) for j in shards[rank]], res, max_batch, share_geometry)
        if on_device:
            torch.cuda.synchronize()   # (the library wrote the buffer on its own streams)
    except Exception as e:
        err = e
    _agree_or_raise(err, group, force=force)
    if world == 1 and not force:
        out = [None] * n
        for i, j in enumerate(shards[0]):
            out[j] = _Results.unpack(res.buf[i], res.shape[2])
        return out
    # the result shape, for ranks without a job
    device = _collective_device(group)
    meta = torch.tensor(list(res.shape) + [1] if res.shape else [0, 0, 0, 0], dtype=torch.int64, device=device)
    dist.all_reduce(meta, op=dist.ReduceOp.MAX, group=group)
    rows, cols, cplx = int(meta[0]), int(meta[1]), bool(meta[2])
    nmax = max(len(s) for s in shards)
    send = _Results(nmax, on_device)
    send.ensure(rows, cols, cplx)
    if res.buf is not None and len(shards[rank]):
        if on_device:
            send.buf[:len(shards[rank])].copy_(res.buf[:len(shards[rank])])
        else:
            send.buf[:len(shards[rank])] = res.buf[:len(shards[rank])]
    buf = send.buf if on_device else torch.from_numpy(send.buf)
    gathered = [torch.zeros_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, gathered, dst=0, group=group)
    if rank != 0:
        return None
    out = [None] * n
    for r, s in enumerate(shards):
        g = gathered[r].cpu().numpy()   # (the one copy to the host, on rank 0)
        for i, j in enumerate(s):
            out[j] = _Results.unpack(g[i], cplx)
    return out


def _even_shards(n, world):
    """Equal-shape jobs: longest-processing-time shares (the library cuts a share into chunks of at most max_batch)."""
    return shard_jobs(np.ones(n), world)


def emagls2_radius_sweep(hL, hR, hrirGridAziRad, hrirGridZenRad, radii, micGridAziRad, micGridZenRad, order, fs, length,
                         shDefinition="real", group=None, max_batch=16, _one_share_of=None):
    """getEMagLs2Filters (lib/getEMagLs2Filters.m:1-2) for every array radius of `radii` (BASELINE config 4): padded lane batches
    (of 16 designs: one resident sweep launch per batch, batch_cost), whole batches per rank, one gather.  Returns [(wMlsL, wMlsR), ...] in the order of `radii` on rank 0, None elsewhere."""
    from . import _lib as L
    hL = np.asfortranarray(hL, dtype=np.float64)
    hR = np.asfortranarray(hR, dtype=np.float64)
    radii = [float(r) for r in radii]
    so = [simulation_order(order, fs, r, raw=True) for r in radii]
    maz, mzn = np.asarray(micGridAziRad, dtype=np.float64), np.asarray(micGridZenRad, dtype=np.float64)
    azi, zen = np.asarray(hrirGridAziRad, dtype=np.float64), np.asarray(hrirGridZenRad, dtype=np.float64)
    nmics = int(maz.size)
    _, _, world = _pg(group)
    per_rank, load = shard_lane_batches(padded_lane_batches(so, max_batch), world if _one_share_of is None else int(_one_share_of))
    pad_of = {j: pad for bl in per_rank for idx, pad in bl for j in idx}
    shards = [[j for idx, _ in bl for j in idx] for bl in per_rank]
    if _one_share_of is not None:   # (measurement: the share of the most loaded of `_one_share_of` ranks, run in this single process)
        r = int(np.argmax(load))
        shards = [shards[r]]

    def make_job(j):
        return dict(kind=L.KIND_EMAGLS2, basis=shDefinition, order=int(order), fs=float(fs), length=int(length), hL=hL, hR=hR, hrir_azi=azi, hrir_zen=zen,
                    mic_radius=radii[j], mic_azi=maz, mic_zen=mzn, sim_order_pad=int(pad_of[j]) if nmics <= 32 else 0)   # (no padding on the path of more than 32 microphones)
    # (the balanced cut makes chunks of up to 32 designs, `max_batch` on average: the library must not cut them again at `max_batch`
    # -- a chunk of 20 would run as 16 + 4, the tail on the slower slab form, and the cost model behind the shards would no longer
    # describe what runs; chunks end where the padded shape changes anyway.  More than 32 microphones: plan by plan.)
    out = _run_job_list(len(radii), shards, make_job, group, 32 if nmics <= 32 else max_batch)
    if _one_share_of is not None:
        return [out[j] for j in shards[0]]
    return out


def emagls_from_atf_subjects(subjects, hrirGridAziZenRad, atfIrs, atfGridAziZenRad, fs, filterLen, fTrans, group=None,
                             max_batch=8):
    """getEMagLsFiltersFromAtf (lib/getEMagLsFiltersFromAtf.m:1) for every HRTF subject of `subjects` = [(hL, hR), ...] on ONE
    ATF set and HRIR grid (BASELINE config 5): the subjects are spread over the ranks, each rank runs its share in batches
    that compute the ATF side once (Batch.shares_atf_side) and sweep all their subjects in one resident launch; one gather.
    Returns [(wMlsL, wMlsR), ...] in the order of `subjects` on rank 0, None elsewhere."""
    from . import _lib as L
    hg = np.asarray(hrirGridAziZenRad, dtype=np.float64)
    ag = np.asarray(atfGridAziZenRad, dtype=np.float64)
    atf = np.asfortranarray(atfIrs, dtype=np.float64)
    hazi, hzen, aazi, azen = (np.ascontiguousarray(v) for v in (hg[:, 0], hg[:, 1], ag[:, 0], ag[:, 1]))
    _, _, world = _pg(group)

    def make_job(j):
        return dict(kind=L.KIND_FROM_ATF, basis="real", order=0, fs=float(fs), length=int(filterLen), hL=np.asfortranarray(subjects[j][0], dtype=np.float64),
                    hR=np.asfortranarray(subjects[j][1], dtype=np.float64), hrir_azi=hazi, hrir_zen=hzen, atf=atf, atf_azi=aazi, atf_zen=azen,
                    f_trans=float(fTrans))
    return _run_job_list(len(subjects), _even_shards(len(subjects), world), make_job, group, max_batch)


def emagls_hrir_sets(subjects, hrirGridAziRad, hrirGridZenRad, micRadius, micGridAziRad, micGridZenRad, order, fs, length,
                     shDefinition="real", kind="emagls", group=None, max_batch=8):
    """getEMagLsFilters (lib/getEMagLsFilters.m:32; kind 'emagls2': getEMagLs2Filters, 'emainch': getEMagLsFiltersEMAinCH with
    micGridZenRad None) for every HRIR set of `subjects` = [(hL, hR), ...] on ONE HRIR grid and ONE array: the loop over
    subjects a user of the reference writes around the call.  The sets are spread over the ranks; each rank runs its share in
    batches that compute the geometry stages once (Batch.share_geometry: SH matrices, array model, every bin's regularised
    inverse) and sweep all their sets in one resident launch (designs with more than 32 channels: one at a time); one gather.
    Same filters as the single calls.  Returns [(wL, wR), ...] in the order of `subjects` on rank 0, None elsewhere."""
    from . import _lib as L
    K = {"emagls": L.KIND_EMAGLS, "emagls2": L.KIND_EMAGLS2, "emainch": L.KIND_EMA_CH}[kind]
    azi = np.asarray(hrirGridAziRad, dtype=np.float64)
    zen = np.asarray(hrirGridZenRad, dtype=np.float64)
    maz = np.asarray(micGridAziRad, dtype=np.float64)
    mzn = None if micGridZenRad is None else np.asarray(micGridZenRad, dtype=np.float64)
    _, _, world = _pg(group)

    def make_job(j):
        return dict(kind=K, basis=shDefinition, order=int(order), fs=float(fs), length=int(length), hL=np.asfortranarray(subjects[j][0], dtype=np.float64),
                    hR=np.asfortranarray(subjects[j][1], dtype=np.float64), hrir_azi=azi, hrir_zen=zen, mic_radius=float(micRadius), mic_azi=maz, mic_zen=mzn)
    return _run_job_list(len(subjects), _even_shards(len(subjects), world), make_job, group, max_batch, share_geometry=True)


def magls_hrir_sets(subjects, hrirGridAziRad, hrirGridZenRad, order, fs, length, shDefinition="real", group=None, max_batch=8):
    """getMagLsFilters (lib/getMagLsFilters.m:30; hrirGridZenRad None: getMagLsFilters2D on a horizontal grid) for every HRIR
    set of `subjects` = [(hL, hR), ...] on ONE grid: spread over the ranks, each rank's share in batches that compute the SH
    side once (Batch.share_geometry) and sweep all their sets in one resident launch (orders 5..7, more than 32 channels: one
    design at a time); one gather.  Same filters as the single calls.  Returns [(wL, wR), ...] in the order of `subjects` on
    rank 0, None elsewhere."""
    from . import _lib as L
    azi = np.asarray(hrirGridAziRad, dtype=np.float64)
    zen = None if hrirGridZenRad is None else np.asarray(hrirGridZenRad, dtype=np.float64)
    K = L.KIND_MAGLS if zen is not None else L.KIND_MAGLS_2D
    _, _, world = _pg(group)

    def make_job(j):
        return dict(kind=K, basis=shDefinition, order=int(order), fs=float(fs), length=int(length), hL=np.asfortranarray(subjects[j][0], dtype=np.float64),
                    hR=np.asfortranarray(subjects[j][1], dtype=np.float64), hrir_azi=azi, hrir_zen=zen)
    return _run_job_list(len(subjects), _even_shards(len(subjects), world), make_job, group, max_batch, share_geometry=True)
