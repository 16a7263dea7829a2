"""The N > 1 path on CPU: job sharding and the single gather, world_size 2 over gloo."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from emagls_amd.batch import run_batch, run_lane_batches, shard_jobs, simulation_order


def _design(job):
    """Stand-in for a filter design (the HIP library needs a GPU): deterministic in the job parameters."""
    r, cplx = job
    rng = np.random.default_rng(int(r * 1e6))
    w = rng.standard_normal((16, 5))
    if cplx:
        w = w + 1j * rng.standard_normal((16, 5))
    return w, -w


def _worker(rank, world, port, cplx, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    radii = np.linspace(0.02, 0.10, 7)
    jobs = [(float(r), cplx) for r in radii]
    costs = [(max(4, int(np.ceil(439.6 * r))) + 1) ** 2 for r in radii]  # (simOrder+1)^2 at 48 kHz
    out = run_batch(jobs, _design, costs)
    # the class-aware runner (whole padded lane batches per rank): 21 radii -> 3 batches of equal cost, pad order = the batch's highest
    radii2 = np.linspace(0.02, 0.10, 21)
    jobs2 = [(float(r), cplx) for r in radii2]
    so = [simulation_order(4, 48000.0, r, raw=True) for r in radii2]
    seen = []

    def batch_fn(bjobs, pad):
        assert 1 <= len(bjobs) <= 32 and pad >= max(simulation_order(4, 48000.0, r, raw=True) for r, _ in bjobs)
        seen.append(len(bjobs))
        return [_design(j) for j in bjobs]
    out2 = run_lane_batches(jobs2, so, batch_fn)
    assert seen and sum(seen) <= 21
    if rank == 0:
        ok = all(np.array_equal(out[j][0], _design(jobs[j])[0]) and np.array_equal(out[j][1], _design(jobs[j])[1])
                 for j in range(len(jobs)))
        ok = ok and len(out2) == 21 and all(np.array_equal(out2[j][0], _design(jobs2[j])[0]) and
                                              np.array_equal(out2[j][1], _design(jobs2[j])[1]) for j in range(21))
        q.put(("ok" if ok else "mismatch", len(out)))
    else:
        assert out is None and out2 is None
    dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("cplx", [False, True])
def test_two_ranks_gather(cplx):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, cplx, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) == ("ok", 7)


def test_shard_jobs_balanced():
    costs = [(n + 1) ** 2 for n in range(9, 45)]
    shards = shard_jobs(costs, 8)
    assert sorted(j for s in shards for j in s) == list(range(len(costs)))
    loads = [sum(costs[j] for j in s) for s in shards]
    assert max(loads) / min(loads) < 1.15
    assert shard_jobs([1.0] * 5, 8)[5:] == [[], [], []]


def test_single_process_batch():
    jobs = [(0.03, False), (0.05, False)]
    out = run_batch(jobs, _design)
    assert len(out) == 2 and np.array_equal(out[1][0], _design(jobs[1])[0])


def test_single_process_lane_batches():
    jobs = [(0.03, False), (0.05, False), (0.031, False)]
    so = [simulation_order(4, 48000.0, r, raw=True) for r, _ in jobs]
    out = run_lane_batches(jobs, so, lambda bj, pad: [_design(j) for j in bj])
    assert len(out) == 3 and all(np.array_equal(out[j][0], _design(jobs[j])[0]) for j in range(3))


def _fake_share(jobs, res, max_batch, share_geometry=False):
    """What batch._run_share does, without the library: the filters of every job written through the addresses `res` hands out
    (column-major len x channels, complex interleaved), exactly as emagls_jobs_run would."""
    import ctypes as C
    if not jobs:
        return
    cplx = jobs[0]["job"][1]
    res.ensure(16, 5, cplx)
    for i, kw in enumerate(jobs):
        if kw["job"][0] < 0:
            raise ValueError("unsupported shape (test)")
        pl, pr = res.ptrs(i, 1)
        wL, wR = _design(kw["job"])
        for w, addr in ((wL, pl[0]), (wR, pr[0])):
            flat = np.asfortranarray(w).ravel(order="F")
            flat = flat.view(np.float64) if cplx else flat
            C.memmove(addr, flat.ctypes.data, flat.nbytes)


def _job_list_worker(rank, world, port, cplx, fail, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from emagls_amd import batch as B
    B._run_share = _fake_share
    jobs = [(0.02 + 0.004 * j, cplx) for j in range(7)]
    if fail:
        jobs[5] = (-1.0, cplx)   # (lands on one rank only)
    shards = B._even_shards(len(jobs), world)
    try:
        out = B._run_job_list(len(jobs), shards, lambda j: {"job": jobs[j]}, None, 3)
    except Exception as e:
        q.put((rank, type(e).__name__))
        dist.destroy_process_group()
        return
    if rank == 0:
        ok = len(out) == 7 and all(np.array_equal(out[j][0], _design(jobs[j])[0]) and np.array_equal(out[j][1], _design(jobs[j])[1])
                                   and out[j][0].shape == (16, 5) for j in range(7))
        q.put((rank, "ok" if ok else "mismatch"))
    else:
        q.put((rank, "none" if out is None else "unexpected"))
    dist.destroy_process_group()


@pytest.mark.parametrize("cplx", [False, True])
def test_job_list_runner_gathers_device_layout_buffers(cplx):
    """batch._run_job_list (the loop behind the four job lists): a share per rank through the library's scheduler, the filters
    written in place into the gather's own buffer, one gather, job order restored on rank 0 -- two gloo ranks, the library replaced
    by a stand-in."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_job_list_worker, args=(r, 2, port, cplx, False, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert dict(q.get(timeout=5) for _ in range(2)) == {0: "ok", 1: "none"}


def test_a_failing_rank_does_not_leave_the_others_in_the_gather():
    """One rank's share raises (an unsupported shape): every rank raises -- the failing one its own exception, the other a
    RuntimeError naming it -- instead of waiting in dist.gather forever."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_job_list_worker, args=(r, 2, port, False, True, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(60)
        assert p.exitcode == 0, "a rank hung or crashed"
    got = dict(q.get(timeout=5) for _ in range(2))
    assert sorted(got.values()) == ["RuntimeError", "ValueError"]


def test_job_list_runner_single_process():
    from emagls_amd import batch as B
    keep = B._run_share
    B._run_share = _fake_share
    try:
        jobs = [(0.02 + 0.004 * j, True) for j in range(5)]
        out = B._run_job_list(5, B._even_shards(5, 1), lambda j: {"job": jobs[j]}, None, 2)
        assert all(np.array_equal(out[j][0], _design(jobs[j])[0]) and np.isfortran(out[j][0]) for j in range(5))
        with pytest.raises(ValueError):
            B._run_job_list(1, [[0]], lambda j: {"job": (-1.0, False)}, None, 2)
    finally:
        B._run_share = keep


def _gpu_worker(rank, world, port, q):
    """Two ranks with REAL designs: both ranks use GPU 0 (a 1-GPU box), the gather runs on gloo.  The job lists north_star names:
    array radii (class-aware lane batches), HRIR sets on one geometry (geometry-sharing batches), FromAtf subjects."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import emagls_amd as E
        from emagls_amd import synth
        from emagls_amd.batch import emagls2_radius_sweep, emagls_from_atf_subjects, emagls_hrir_sets
        rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
        azi, zen = synth.fibonacci_grid(700)
        maz, mzn = synth.em32_grid()
        hL, hR = synth.rigid_sphere_hrirs(azi, zen, taps=64)
        radii = [0.031, 0.0312, 0.0335, 0.047, 0.0471, 0.0472, 0.0473]
        out_r = emagls2_radius_sweep(hL, hR, azi, zen, radii, maz, mzn, 4, 48000.0, 64)
        subjects = [synth.rigid_sphere_hrirs(azi, zen, taps=64, seed=40 + j, head_radius=0.08 + 0.002 * j) for j in range(5)]
        out_s = emagls_hrir_sets(subjects, azi, zen, 0.042, maz, mzn, 4, 48000.0, 128, "complex", max_batch=3)
        atf, aazi, azen = synth.glasses_atfs(natf=600, nmics=5, taps=48)
        hg, ag = np.column_stack([azi, zen]), np.column_stack([aazi, azen])
        out_a = emagls_from_atf_subjects(subjects[:3], hg, atf, ag, 48000.0, 128, 2000.0)
        if rank == 0:
            worst = 0.0
            for j in (0, 3, 6):
                w = E.getEMagLs2Filters(hL, hR, azi, zen, radii[j], maz, mzn, 4, 48000.0, 64, "real")
                worst = max(worst, rel(out_r[j][0], w[0]), rel(out_r[j][1], w[1]))
            for j in range(5):
                w = E.getEMagLsFilters(subjects[j][0], subjects[j][1], azi, zen, 0.042, maz, mzn, 4, 48000.0, 128, "complex")
                worst = max(worst, rel(out_s[j][0], w[0]), rel(out_s[j][1], w[1]))
            for j in range(3):
                w = E.getEMagLsFiltersFromAtf(subjects[j][0], subjects[j][1], hg, atf, ag, 48000.0, 128, 2000.0, verbose=False)
                worst = max(worst, rel(out_a[j][0], w[0]), rel(out_a[j][1], w[1]))
            q.put(("ok" if (len(out_r), len(out_s), len(out_a)) == (7, 5, 3) else "counts", worst))
        else:
            assert out_r is None and out_s is None and out_a is None
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_with_real_designs_on_one_gpu():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gpu_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    tag, worst = q.get(timeout=5)
    print(f"two ranks, three job lists, gathered on rank 0: worst rel vs single calls = {worst:.3e}")
    assert tag == "ok" and worst < 1e-9


def _nccl_worker(port, q):
    """ONE rank on RCCL with EMAGLS_FORCE_COLLECTIVE=1: the job list's filters written device to device into the gather's buffer
    (a torch tensor), the agreement all-reduce, the shape all-reduce and the gather itself all run on `nccl` -- the device branch of
    batch._run_job_list that a box with one GPU can execute."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["EMAGLS_FORCE_COLLECTIVE"] = "1"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        import emagls_amd as E
        from emagls_amd import synth
        from emagls_amd.batch import emagls2_radius_sweep, emagls_hrir_sets
        rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
        azi, zen = synth.fibonacci_grid(700)
        maz, mzn = synth.em32_grid()
        hL, hR = synth.rigid_sphere_hrirs(azi, zen, taps=64)
        subjects = [synth.rigid_sphere_hrirs(azi, zen, taps=64, seed=40 + j, head_radius=0.08 + 0.002 * j) for j in range(4)]
        out_s = emagls_hrir_sets(subjects, azi, zen, 0.042, maz, mzn, 4, 48000.0, 128, "complex", max_batch=3)
        radii = [0.047, 0.0471, 0.0473]
        out_r = emagls2_radius_sweep(hL, hR, azi, zen, radii, maz, mzn, 4, 48000.0, 64)
        worst = 0.0
        for j in range(4):
            w = E.getEMagLsFilters(subjects[j][0], subjects[j][1], azi, zen, 0.042, maz, mzn, 4, 48000.0, 128, "complex")
            worst = max(worst, rel(out_s[j][0], w[0]), rel(out_s[j][1], w[1]))
        for j in range(3):
            w = E.getEMagLs2Filters(hL, hR, azi, zen, radii[j], maz, mzn, 4, 48000.0, 64, "real")
            worst = max(worst, rel(out_r[j][0], w[0]), rel(out_r[j][1], w[1]))
        q.put(("ok", worst))
    except Exception as e:   # pragma: no cover
        q.put(("error: %r" % (e,), 1.0))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_job_lists_through_rccl_on_one_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_nccl_worker, args=(_free_port(), q))
    p.start()
    p.join(600)
    assert p.exitcode == 0
    tag, worst = q.get(timeout=5)
    print(f"one rank, RCCL collectives, device-resident results: worst rel vs single calls = {worst:.3e}")
    assert tag == "ok" and worst < 1e-9
