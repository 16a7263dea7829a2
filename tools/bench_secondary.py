"""Secondary figures of bench.py (same JSON line, key "secondary"): BASELINE config 4 and config 5 on one GPU.

config 4: getEMagLs2Filters, raw 32-microphone em32, 2702 directions, 1024 taps -- one lane batch of 8 array radii of one
          simulation-order class (a batch of the 256-radius job list as emagls_amd.batch.lane_groups forms them), at the
          middle (r = 5 cm, order 22) and at the far end (r = 10 cm, order 44) of the radius range;
config 5: getEMagLsFiltersFromAtf, 16 384 ATF directions x 8 microphones, 2702 HRIR directions, 2048 taps -- one HRTF subject.
Inputs resident in HBM, hipGraph replay where the pipeline captures; wall-clock over `reps` executes after three warm ones."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _grids():
    from emagls_amd import synth
    gpath = os.path.join(ROOT, "tests", "golden", "ref_fixtures.npz")
    if os.path.exists(gpath):
        g = np.load(gpath)
        return g["grid/hrirGridAziRad"], g["grid/hrirGridZenRad"], g["grid/micGridAziRad"], g["grid/micGridZenRad"]
    azi, zen = synth.fibonacci_grid(2702)
    maz, mzn = synth.em32_grid()
    return azi, zen, maz, mzn


def _csrc_now():
    from tools.csrc_hash import csrc_sha16
    return csrc_sha16()


def _roofline(workload, ms_per_execute, compulsory_bytes, note):
    """`roofline` block of a secondary workload: its time per execute (measured here) against the HBM bytes one execute moves
    (PMC passes of the same workload, profiles/secondary_traffic.json -- written by tools/secondary_traffic.py from
    tools/experiments/secondary_prof.sh) and against the bytes it has to move at least (inputs once in, filters once out)."""
    path = os.path.join(ROOT, "profiles", "secondary_traffic.json")
    t, stale = {}, None
    try:
        allw = json.load(open(path))
        t = allw.get(workload, {})
        from tools.csrc_hash import stale as _stale
        stale = _stale(allw.get("csrc_sha16"))     # the PMC passes were taken on other kernel sources than this run's
    except Exception:
        pass
    sec = ms_per_execute * 1e-3
    traffic = t.get("bytes_per_execute")
    out = {"bound": "latency", "ms_per_execute": round(ms_per_execute, 4), "compulsory_bytes": int(compulsory_bytes),
           "compulsory_GBps": round(compulsory_bytes / sec / 1e9, 1), "traffic": traffic,
           "achieved": round(traffic / sec / 1e9, 1) if traffic else None, "peak": 8000.0, "unit": "GB/s",
           "frac": round(traffic / sec / 1e9 / 8000.0, 4) if traffic else None,
           "traffic_over_compulsory": round(traffic / compulsory_bytes, 2) if traffic else None,
           "kernel_time_us_per_execute": t.get("kernel_time_us_per_execute"), "dominant_kernel": t.get("dominant_kernel"), "note": note}
    if traffic is not None:
        out["stale"] = bool(stale)
        out["traffic_source"] = "profiles/secondary_traffic.json (PMC passes on kernel sources %s; this run's: %s)" % (allw.get("csrc_sha16", "unstamped"), _csrc_now())
    return out


def config4(radii, reps=4, roofline_key=None):
    from emagls_amd import Batch, Plan, synth, _lib as L
    azi, zen, maz, mzn = _grids()
    hL, hR = synth.rigid_sphere_hrirs(azi, zen)
    plans = []
    for r in radii:
        p = Plan(L.KIND_EMAGLS2, "real", 4, 48000.0, 1024, hL.shape[0], hL.shape[1], float(r), 32)
        p.set_hrir_grid(azi, zen)
        p.set_mic_grid(maz, mzn)
        p.set_hrirs(hL, hR)
        plans.append(p)
    info = plans[0].info()
    b = Batch(plans)
    lanes = b.lane_mode()
    for _ in range(3):
        b.execute()
    b.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        b.execute()
    b.synchronize()
    dt = (time.perf_counter() - t0) / reps
    b.get_filters()   # (status check of the last execute)
    b.close()
    for p in plans:
        p.close()
    res = {"radii_cm": [round(100 * float(radii[0]), 3), round(100 * float(radii[-1]), 3)], "designs_per_batch": len(radii), "lane_mode": lanes,
           "sim_order": info.sim_order, "orthonormal_route_orders": info.hh_orders, "gram_route_from_bin": info.gram_from,
           "ms_per_batch": round(dt * 1e3, 3), "filter_sets_per_s": round(len(radii) / dt, 1)}
    if roofline_key:
        comp = len(radii) * (2 * hL.size * 8 + 2 * 1024 * 32 * 8)
        res["roofline"] = _roofline(roofline_key, dt * 1e3, comp, "one lane batch of 8 radii alone on the GPU; the resident sweep (1024 taps: 983 dependent "
                                    "bins) is the dominant kernel and is bound by its per-bin exchange, not by bytes")
    return res


def config4_rank_share(world=8, rank=None, reps=4, max_batch=16):
    """BASELINE config 4 as named: 256 radii on 2..10 cm over 8 GPUs.  One rank's full share -- 32 radii as 2 padded lane
    batches of 16 (round 4; 4 of 8 with max_batch = 8: emagls_amd.batch.padded_lane_batches / shard_lane_batches) -- all its
    batches resident and in flight together, full size (1024 taps); the rank is the one the cost model loads most.
    filter_sets_per_s is what ONE GPU of the 8 delivers on the job list."""
    from emagls_amd import Batch, Plan, synth, _lib as L
    from emagls_amd.batch import padded_lane_batches, shard_lane_batches, simulation_order
    azi, zen, maz, mzn = _grids()
    hL, hR = synth.rigid_sphere_hrirs(azi, zen)
    radii = np.linspace(0.02, 0.10, 256)
    so = [simulation_order(4, 48000.0, r, raw=True) for r in radii]
    per_rank, load = shard_lane_batches(padded_lane_batches(so, max_batch), world)
    if rank is None:
        rank = int(np.argmax(load))      # the rank the cost model loads most: its share bounds the job list
    mine = per_rank[rank]
    import ctypes
    import torch
    prev = ctypes.c_int(0)
    L.check(L.load().emagls_set_batch_max(max(max_batch, 8), ctypes.byref(prev)))
    streams = [torch.cuda.Stream() for _ in mine]
    units = []
    for (idx, pad), st in zip(mine, streams):
        plans = []
        for j in idx:
            p = Plan(L.KIND_EMAGLS2, "real", 4, 48000.0, 1024, hL.shape[0], hL.shape[1], float(radii[j]), 32, sim_order_pad=pad)
            p.set_hrir_grid(azi, zen)
            p.set_mic_grid(maz, mzn)
            p.set_hrirs(hL, hR)
            plans.append(p)
        b = Batch(plans)
        b.set_stream(st.cuda_stream)
        units.append((b, plans))
    lanes = [b.lane_mode() for b, _ in units]
    for b, _ in units:
        for _ in range(3):
            b.execute()
        b.synchronize()
    each = []
    for b, _ in units:      # one batch at a time
        t0 = time.perf_counter()
        b.execute()
        b.synchronize()
        each.append(round((time.perf_counter() - t0) * 1e3, 3))
    t0 = time.perf_counter()
    for _ in range(reps):
        for b, _ in units:
            b.execute()
        for b, _ in units:
            b.synchronize()
    dt = (time.perf_counter() - t0) / reps
    n = sum(len(idx) for idx, _ in mine)
    for b, plans in units:
        b.get_filters()
        b.close()
        for p in plans:
            p.close()
    L.check(L.load().emagls_set_batch_max(prev.value, None))
    return {"ranks": world, "rank": rank, "designs": n, "lane_batches": [len(idx) for idx, _ in mine], "pad_orders": [pad for _, pad in mine],
            "lane_mode": lanes, "rank_load_spread": round(max(load) / min(load), 4), "ms_per_batch_alone": each,
            "ms_per_share": round(dt * 1e3, 3), "filter_sets_per_s": round(n / dt, 1)}


def config4_rank_share_runner(world=8, reps=4, max_batch=16):
    """The same share through the PRODUCT's runner: emagls_amd.batch.emagls2_radius_sweep with host arrays (the HRIRs travel over
    PCIe once per design, the filters come back to the host), the library's scheduler (emagls_jobs_run) keeping the rank's chunks
    in flight.  Three situations: the first call of the process (library initialisation, plan creation, eager run), a call on NEW
    radii of the same classes (plans created inside the call from pooled memory, eager run), and the same list again (the chunks'
    plans and batches resident: uploads, hipGraph replays, downloads)."""
    from emagls_amd import synth, _lib as L
    from emagls_amd.batch import emagls2_radius_sweep
    azi, zen, maz, mzn = _grids()
    hL, hR = synth.rigid_sphere_hrirs(azi, zen)
    radii = np.linspace(0.02, 0.10, 256)

    def call(r):
        t0 = time.perf_counter()
        out = emagls2_radius_sweep(hL, hR, azi, zen, r, maz, mzn, 4, 48000.0, 1024, "real", max_batch=max_batch, _one_share_of=world)
        return time.perf_counter() - t0, len(out)
    t_first, n = call(radii)
    t_new = [call(radii + 1e-5 * (k + 1))[0] for k in range(2)]          # (other radii, the same simulation-order classes)
    t_again = [call(radii)[0] for _ in range(reps + 1)][1:]              # (the first list again: its chunks are resident after one repeat)
    L.check(L.load().emagls_cache_clear())
    dt = float(np.median(t_again))
    return {"ranks": world, "designs": n, "max_batch": max_batch, "first_call_s": round(t_first, 4), "new_radii_s": [round(t, 4) for t in t_new],
            "resident_s": [round(t, 4) for t in t_again], "ms_per_share": round(dt * 1e3, 3), "filter_sets_per_s": round(n / dt, 1),
            "filter_sets_per_s_new_radii": round(n / float(np.median(t_new)), 1),
            "note": "emagls_amd.batch.emagls2_radius_sweep (emagls_jobs_run underneath), host arrays in and out; filter_sets_per_s: the list's "
                    "chunks resident; _new_radii: plans created inside the call"}


def em64(reps=2):
    """A 64-capsule array (33..64 channels: the plain S-space path of wide_array.hip): getEMagLs2Filters, 64 microphones,
    r = 4.2 cm, 2702 directions, 1024 taps -- one design at a time."""
    from emagls_amd import Plan, synth, _lib as L
    azi, zen, _, _ = _grids()
    hL, hR = synth.rigid_sphere_hrirs(azi, zen)
    maz, mzn = synth.fibonacci_grid(64)
    p = Plan(L.KIND_EMAGLS2, "real", 4, 48000.0, 1024, hL.shape[0], hL.shape[1], 0.042, 64)
    p.set_hrir_grid(azi, zen)
    p.set_mic_grid(maz, mzn)
    p.set_hrirs(hL, hR)
    for _ in range(2):
        p.execute()
    p.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        p.execute()
        p.synchronize()
    dt = (time.perf_counter() - t0) / reps
    p.get_filters()
    i = p.info()
    p.close()
    out = {"mics": 64, "radius_cm": 4.2, "taps": 1024, "sim_order": i.sim_order, "ms_per_design": round(dt * 1e3, 2), "filter_sets_per_s": round(1.0 / dt, 1),
           "device_GB": round(i.device_bytes / 1e9, 2)}
    # HRIR sets on this one geometry (emagls_design_hrir_sets): two plans alternate and keep G_k, the per-bin factors and Y_reg_inv_k
    # between sets (round 6) -- first call (plans created, geometry stages once per plan) and a repeat
    import emagls_amd as E
    nset = 8
    sets = [synth.rigid_sphere_hrirs(azi, zen, seed=5 + j) for j in range(nset)]
    sL, sR = np.stack([q[0] for q in sets], axis=2), np.stack([q[1] for q in sets], axis=2)
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        E.designHrirSets("emagls2", sL, sR, azi, zen, 0.042, maz, mzn, 4, 48000.0, 1024, "real")
        times.append(time.perf_counter() - t0)
    out["hrir_sets_on_one_geometry"] = {"sets": nset, "first_call_ms_per_set": round(times[0] * 1e3 / nset, 2), "ms_per_set": round(min(times[1:]) * 1e3 / nset, 2),
                                        "filter_sets_per_s": round(nset / min(times[1:]), 1),
                                        "note": "emagls_design_hrir_sets: the sets pass through two plans that keep their geometry stages (same filters as the single designs, bit for bit)"}
    return out


def plain_paths(reps=2):
    """The designs of round 6's widened shapes on the plain (launch-per-bin) paths, one at a time, 2702 directions: getEMagLsFiltersEMAinSH of
    order 6 (49 channels, 20 equatorial microphones, 512 taps: tall Householder kernels), getMagLsFilters of order 15 (256 channels, 512 taps:
    the loop forms of wide.hip), getLsFilters of order 15."""
    from emagls_amd import Plan, synth, _lib as L
    azi, zen, _, _ = _grids()
    hL, hR = synth.rigid_sphere_hrirs(azi, zen)

    def timed(p):
        p.set_hrir_grid(azi, zen)
        p.set_hrirs(hL, hR)
        for _ in range(2):
            p.execute()
        p.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            p.execute()
            p.synchronize()
        dt = (time.perf_counter() - t0) / reps
        p.get_filters()
        p.close()
        return round(dt * 1e3, 2)
    out = {}
    p = Plan(L.KIND_EMA_SH, "real", 6, 48000.0, 512, hL.shape[0], hL.shape[1], 0.05, 20)
    p.set_mic_grid(np.linspace(0, 2 * np.pi, 20, endpoint=False) + 0.1, None)
    out["emainsh_order6_ms_per_design"] = timed(p)
    out["magls_order15_ms_per_design"] = timed(Plan(L.KIND_MAGLS, "real", 15, 48000.0, 512, hL.shape[0], hL.shape[1]))
    out["ls_order15_ms_per_design"] = timed(Plan(L.KIND_LS, "real", 15, 48000.0, 128, hL.shape[0], hL.shape[1]))
    return out


def config5(reps=4, subjects=8):
    """BASELINE config 5: one HRTF subject alone, and the batch of 8 subjects of one ATF set (ATF side computed once, one
    resident sweep launch for all subjects)."""
    from emagls_amd import Batch, Plan, synth, _lib as L
    azi, zen, _, _ = _grids()
    atf, aazi, azen = synth.glasses_atfs(natf=16384, nmics=8, taps=256)

    def mk(j):
        hL, hR = synth.rigid_sphere_hrirs(azi, zen, seed=100 + j, head_radius=0.075 + 0.02 * j / max(subjects - 1, 1))
        p = Plan(L.KIND_FROM_ATF, "real", 0, 48000.0, 2048, hL.shape[0], hL.shape[1], nmics=8, f_trans=2000.0, atf_taps=256, natf=16384)
        p.set_hrir_grid(azi, zen)
        p.set_hrirs(hL, hR)
        p.set_atfs(atf, aazi, azen)
        return p
    p = mk(0)
    for _ in range(3):
        p.execute()
    p.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        p.execute()
        p.synchronize()
    dt = (time.perf_counter() - t0) / reps
    p.get_filters()
    launches = p.info().num_sweep_launches
    p.close()
    out = {"atf_dirs": 16384, "mics": 8, "taps": 2048, "ms_per_subject": round(dt * 1e3, 3), "filter_sets_per_s": round(1.0 / dt, 1),
           "sweep_launches": launches}
    plans = [mk(j) for j in range(subjects)]
    b = Batch(plans)
    for _ in range(3):
        b.execute()
    b.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        b.execute()
        b.synchronize()
    dtb = (time.perf_counter() - t0) / reps
    b.get_filters()
    out["batch"] = {"subjects": subjects, "atf_side_shared": b.shares_atf_side(), "ms_per_batch": round(dtb * 1e3, 3),
                    "ms_per_subject": round(dtb * 1e3 / subjects, 3), "filter_sets_per_s": round(subjects / dtb, 1)}
    hbytes = 2 * 2702 * 512 * 8     # (the synthetic HRIRs of one subject; the ATF set counts once per execute)
    out["roofline"] = _roofline("config5_single", dt * 1e3, atf.size * 8 + hbytes + 2 * 2048 * 8 * 8,
                                "one subject: 1024 dependent bins x 8 channels, bound by the per-bin exchange of the resident sweep")
    out["batch"]["roofline"] = _roofline("config5_batch", dtb * 1e3, atf.size * 8 + subjects * (hbytes + 2 * 2048 * 8 * 8),
                                         "8 subjects of one ATF set in one resident sweep launch")
    b.close()
    for q in plans:
        q.close()
    return out


def config3_hrir_sets(n_batches=3, per_batch=32, rounds=6):
    """BASELINE config 3's design (em32, r = 4.2 cm, N = 4, complex SH, 2702 directions, 512 taps) as a job list of HRIR SETS on
    one geometry -- the loop over subjects around getEMagLsFilters with the same grids and array: batches with
    emagls_batch_set_geometry_sharing run the geometry stages once per batch (32 sets per batch since round 5: one register-resident
    sweep launch per batch; 16 until then: 3.1 k sets/s against 4.5 k).  n_batches batches of per_batch sets in flight,
    every execute recomputes everything (plan 0's geometry included) from the inputs resident in HBM."""
    import ctypes
    import torch
    from emagls_amd import Batch, Plan, synth, _lib as L
    azi, zen, maz, mzn = _grids()
    lib = L.load()
    prev = ctypes.c_int(0)
    L.check(lib.emagls_set_batch_max(max(per_batch, 8), ctypes.byref(prev)))
    units = []
    try:
        for u in range(n_batches):
            plans = []
            for j in range(per_batch):
                hL, hR = synth.rigid_sphere_hrirs(azi, zen, seed=777 + 100 * u + j)
                p = Plan(L.KIND_EMAGLS, "complex", 4, 48000.0, 512, hL.shape[0], hL.shape[1], 0.042, 32)
                p.set_hrir_grid(azi, zen)
                p.set_mic_grid(maz, mzn)
                p.set_hrirs(hL, hR)
                plans.append(p)
            b = Batch(plans)
            b.set_stream(torch.cuda.Stream().cuda_stream)
            b.share_geometry(True)
            units.append((plans, b))
    finally:
        L.check(lib.emagls_set_batch_max(prev.value, None))
    for plans, b in units:
        for _ in range(2):
            b.execute()
        b.synchronize()
    shared = all(b.shares_geometry() for _, b in units)
    t0 = time.perf_counter()
    waves = os.environ.get("EMAGLS_BENCH_WAVES", "1") != "0"   # all batches started together and collected together (bench.py, run_designs)
    for r in range(rounds):
        for plans, b in units:
            if r and not waves:
                b.synchronize()
            b.execute()
        if waves:
            for plans, b in units:
                b.synchronize()
    for plans, b in units:
        b.synchronize()
    dt = time.perf_counter() - t0
    for plans, b in units:
        b.get_filters()   # (status check)
        b.close()
        for p in plans:
            p.close()
    n = rounds * n_batches * per_batch
    return {"hrir_sets": n, "batches_in_flight": n_batches, "sets_per_batch": per_batch, "geometry_shared": shared,
            "ms_per_set": round(dt / n * 1e3, 4), "filter_sets_per_s": round(n / dt, 1),
            "note": "same filters as independent designs (bit-identical to single plans in tests/test_gpu_parity.py); the headline "
                    "figure of this line treats its designs as independent and does NOT use this"}


def config3_host_arrays(njobs=256, nsets=32):
    """BASELINE config 3 as ONE job list of 256 designs with HOST arrays in and out (pageable NumPy memory: 5.5 MB of HRIRs in, 0.4 MB
    of filters out per design, the PCIe-inclusive figure of the scheduler): the first call of the shape in an initialised process --
    after emagls_cache_clear: no plans, no arenas, no graphs, an empty block pool --, and the same list again (chunks resident)."""
    import ctypes as C
    from emagls_amd import synth, _lib as L
    lib = L.load()
    azi, zen, maz, mzn = _grids()
    sets = [tuple(np.asfortranarray(h) for h in synth.rigid_sphere_hrirs(azi, zen, seed=4242 + j)) for j in range(nsets)]
    nsamp, D = sets[0][0].shape
    outs = [(np.zeros((512, 25), dtype=np.complex128, order="F"), np.zeros((512, 25), dtype=np.complex128, order="F")) for _ in range(njobs)]
    desc = L.DesignDesc(L.KIND_EMAGLS, L.BASIS["complex"], 4, 48000.0, 512, nsamp, D, 0.042, 32, 0.0, 0, 0, 0, 0, 0)
    keep = [np.ascontiguousarray(x, dtype=np.float64) for x in (azi, zen, maz, mzn)]
    jobs = (L.Job * njobs)()
    for j in range(njobs):
        jb = jobs[j]
        jb.desc = desc
        jb.hL, jb.hR = C.c_void_p(sets[j % nsets][0].ctypes.data), C.c_void_p(sets[j % nsets][1].ctypes.data)
        jb.hrir_azi, jb.hrir_zen, jb.mic_azi, jb.mic_zen = (C.c_void_p(k.ctypes.data) for k in keep)
        jb.wL, jb.wR = C.c_void_p(outs[j][0].ctypes.data), C.c_void_p(outs[j][1].ctypes.data)
    def five_calls():
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            L.check(lib.emagls_jobs_run(jobs, njobs, 32, 4, 0))
            ts.append(time.perf_counter() - t0)
        return ts
    L.check(lib.emagls_cache_clear())             # nothing resident, the block pool empty: every byte is fresh device memory
    cold = five_calls()
    L.check(lib.emagls_cache_release_designs())   # nothing resident, the library keeps its device memory (a long-running process)
    ts = five_calls()
    L.check(lib.emagls_cache_clear())
    res = float(np.median(ts[2:]))
    return {"designs": njobs, "first_call_s": round(ts[0], 4), "first_call_filter_sets_per_s": round(njobs / ts[0], 1), "second_call_s": round(ts[1], 4),
            "first_call_on_fresh_device_memory_s": round(cold[0], 4),
            "resident_s": [round(t, 4) for t in ts[2:]], "filter_sets_per_s": round(njobs / res, 1),
            "note": "emagls_jobs_run, chunks of 32, four in flight, pageable host arrays; first_call: nothing resident (emagls_cache_release_designs: "
                    "plans and arenas are created, eager runs; the second call captures the graphs), device memory from the library's block pool; "
                    "first_call_on_fresh_device_memory: after emagls_cache_clear, which also empties the pool -- hipMalloc of ~50 GB of new VRAM is then "
                    "part of the call (0.1 s on some boxes, seconds on others); filter_sets_per_s: chunks resident"}


def config2_hrir_sets(n_batches=3, per_batch=16, rounds=6, share=True, diffuse=False):
    """BASELINE config 2's design (getMagLsFilters N = 4, 2702 directions, 512 taps) as a job list of HRIR sets: MagLS plans in
    batches -- one resident sweep launch per batch instead of one per design; share=True: sets on one grid, SH side once per
    batch (not available with the covariance constraint, whose rendering needs every plan's own operands)."""
    import ctypes
    import torch
    from emagls_amd import Batch, Plan, synth, _lib as L
    azi, zen, _, _ = _grids()
    lib = L.load()
    prev = ctypes.c_int(0)
    L.check(lib.emagls_set_batch_max(max(per_batch, 8), ctypes.byref(prev)))
    units = []
    single_ms = None
    try:
        for u in range(n_batches):
            plans = []
            for j in range(per_batch):
                hL, hR = synth.rigid_sphere_hrirs(azi, zen, seed=900 + 100 * u + j)
                p = Plan(L.KIND_MAGLS, "real", 4, 48000.0, 512, hL.shape[0], hL.shape[1], 0.0, 0, diffuseness=diffuse)
                p.set_hrir_grid(azi, zen)
                p.set_hrirs(hL, hR)
                plans.append(p)
            if single_ms is None:    # one design alone, for comparison
                p = plans[0]
                for _ in range(3):
                    p.execute()
                p.synchronize()
                t0 = time.perf_counter()
                for _ in range(8):
                    p.execute()
                    p.synchronize()
                single_ms = (time.perf_counter() - t0) / 8 * 1e3
            b = Batch(plans)
            b.set_stream(torch.cuda.Stream().cuda_stream)
            b.share_geometry(share)
            units.append((plans, b))
    finally:
        L.check(lib.emagls_set_batch_max(prev.value, None))
    for plans, b in units:
        for _ in range(2):
            b.execute()
        b.synchronize()
    shared = all(b.shares_geometry() for _, b in units)
    t0 = time.perf_counter()
    waves = os.environ.get("EMAGLS_BENCH_WAVES", "1") != "0"   # all batches started together and collected together (bench.py, run_designs)
    for r in range(rounds):
        for plans, b in units:
            if r and not waves:
                b.synchronize()
            b.execute()
        if waves:
            for plans, b in units:
                b.synchronize()
    for plans, b in units:
        b.synchronize()
    dt = time.perf_counter() - t0
    for plans, b in units:
        b.get_filters()
        b.close()
        for p in plans:
            p.close()
    n = rounds * n_batches * per_batch
    return {"hrir_sets": n, "batches_in_flight": n_batches, "sets_per_batch": per_batch, "geometry_shared": shared,
            "covariance_constraint": bool(diffuse), "single_design_ms": round(single_ms, 3), "ms_per_set": round(dt / n * 1e3, 4),
            "filter_sets_per_s": round(n / dt, 1)}


def binaural_decode(nsamp=120000, nch=25, length=512, reps=10, kinds=("real", "complex")):
    """north_star item (iii) / SURVEY a13: dependencies/binauralDecode.m:33-42 at the harness's size -- a 120 000-sample SH
    recording x 25 channels through 512-tap filters, both ears -- real and complex SH, buffers resident in HBM
    (emagls_binaural_decode_device: overlap-save on hipFFT).  Algorithmic bytes = signal in + filters in + two ears out;
    achieved = those bytes / time against the 8 TB/s HBM peak (the overlap-save passes move several times as much)."""
    import ctypes as C
    import torch
    from emagls_amd import _lib as L
    lib = L.load()
    rng = np.random.default_rng(5)
    out = {}
    for name, cplx in (("real", False), ("complex", True)):
        if name not in kinds:
            continue
        dt = np.complex128 if cplx else np.float64
        sig = rng.standard_normal((nch, nsamp)) + (1j * rng.standard_normal((nch, nsamp)) if cplx else 0)
        wl = rng.standard_normal((nch, length)) + (1j * rng.standard_normal((nch, length)) if cplx else 0)
        wr = rng.standard_normal((nch, length)) + (1j * rng.standard_normal((nch, length)) if cplx else 0)
        to = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(dt))).cuda()     # [nch][nsamp] row-major == [nsamp x nch] column-major
        d_sig, d_wl, d_wr = to(sig), to(wl), to(wr)
        d_out = torch.zeros((2, nsamp), dtype=torch.float64, device="cuda")
        call = lambda: L.check(lib.emagls_binaural_decode_device(C.c_void_p(d_sig.data_ptr()), int(cplx), nsamp, nch, C.c_void_p(d_wl.data_ptr()),
                                                                 C.c_void_p(d_wr.data_ptr()), int(cplx), length, C.c_void_p(d_out.data_ptr()), None, None))
        for _ in range(3):
            call()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            call()                       # (synchronises its stream before returning)
            ts.append(time.perf_counter() - t0)
        t = float(np.median(ts))
        es = 16 if cplx else 8
        nbytes = es * nsamp * nch + 2 * es * length * nch + 8.0 * nsamp * 2
        out[name] = {"ms": round(t * 1e3, 4), "samples_per_s": round(nsamp / t, 1), "realtime_factor_48k": round(nsamp / 48000.0 / t, 1),
                     "algorithmic_bytes": nbytes, "achieved_GBps": round(nbytes / t / 1e9, 2), "frac_of_hbm_peak": round(nbytes / t / 1e9 / 8000.0, 5)}
        if nsamp >= 1000000:
            out[name]["roofline"] = _roofline("decode_" + name, t * 1e3, nbytes, "overlap-save blocks on wave-private transforms; every sample is part of two "
                                              "segments -- the eight blocks of a workgroup are consecutive, so seven of eight second reads stay in the CU's caches")
            out[name]["roofline"]["bound"] = "hbm"
    out["shape"] = {"samples": nsamp, "channels": nch, "taps": length}
    return out


def binaural_decode_long(nsamp=4800000, nch=25, length=512, reps=3, kinds=("real", "complex")):
    """The same render loop on 100 s of audio (4.8 M samples): launch overheads no longer count, what remains is the traffic of
    the overlap-save passes."""
    out = binaural_decode(nsamp, nch, length, reps, kinds)
    return out


def run():
    out = {}
    for name, radii in (("config4_r5cm", np.linspace(0.0480, 0.0500, 8)), ("config4_r10cm", np.linspace(0.0980, 0.1000, 8))):
        try:
            out[name] = config4(radii, roofline_key=name)
        except Exception as e:
            out[name] = {"error": repr(e)}
    try:
        out["config4_rank_share"] = config4_rank_share()
    except Exception as e:
        out["config4_rank_share"] = {"error": repr(e)}
    try:
        out["config4_rank_share_through_the_runner"] = config4_rank_share_runner()
    except Exception as e:
        out["config4_rank_share_through_the_runner"] = {"error": repr(e)}
    try:
        out["config3_job_list_with_host_arrays"] = config3_host_arrays()
    except Exception as e:
        out["config3_job_list_with_host_arrays"] = {"error": repr(e)}
    try:
        out["config3_hrir_sets_on_one_geometry"] = config3_hrir_sets()
    except Exception as e:
        out["config3_hrir_sets_on_one_geometry"] = {"error": repr(e)}
    try:
        out["config2_magls_hrir_sets"] = {"one_grid_shared": config2_hrir_sets(), "independent": config2_hrir_sets(share=False),
                                          "with_covariance_constraint": config2_hrir_sets(share=False, diffuse=True)}
    except Exception as e:
        out["config2_magls_hrir_sets"] = {"error": repr(e)}
    try:
        out["em64_emagls2"] = em64()
    except Exception as e:
        out["em64_emagls2"] = {"error": repr(e)}
    try:
        out["plain_paths"] = plain_paths()
    except Exception as e:
        out["plain_paths"] = {"error": repr(e)}
    try:
        out["binaural_decode"] = binaural_decode()
    except Exception as e:
        out["binaural_decode"] = {"error": repr(e)}
    try:
        out["binaural_decode_100s"] = binaural_decode_long()
    except Exception as e:
        out["binaural_decode_100s"] = {"error": repr(e)}
    try:
        out["config5"] = config5()
    except Exception as e:
        out["config5"] = {"error": repr(e)}
    return out


if __name__ == "__main__":
    import json
    print(json.dumps(run()))
