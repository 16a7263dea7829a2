"""The library's job scheduler (emagls_jobs_run, include/emagls.h) against the single calls: the loop over HRIR sets / array radii /
subjects that a user of the reference writes around one of its functions (testEMagLs.m:75-95), handed over in one call."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


@pytest.fixture(scope="module")
def thin(grids, hrirs):
    sub = slice(0, 2702, 3)
    return dict(hL=hrirs[0][:, sub], hR=hrirs[1][:, sub], azi=grids["azi"][sub], zen=grids["zen"][sub])


def test_job_list_equals_the_single_calls(grids, thin):
    """A mixed list -- 37 eMagLS designs on their own HRIR sets (chunks of 32 + 5: the first on the register-resident sweep with
    twelve waves per workgroup), 3 eMagLS2 designs on different radii of one simulation-order class, 2 MagLS designs -- through
    emagls_jobs_run with host arrays in and out; every job's filters equal the single call's (the lane batches are bit-identical
    to single plans; the one-shot entry points take the same kernels)."""
    import emagls_amd as E
    from emagls_amd import _lib as L
    from emagls_amd.jobs import JobList
    azi, zen, maz, mzn = thin["azi"], thin["zen"], grids["mic_azi"], grids["mic_zen"]
    rng = np.random.default_rng(7)
    sets = [(thin["hL"] * (1.0 + 0.05 * rng.standard_normal()), thin["hR"] * (1.0 + 0.05 * rng.standard_normal())) for _ in range(37)]
    jl = JobList()
    for hL, hR in sets:
        jl.add(L.KIND_EMAGLS, "complex", 4, 48000.0, 128, hL, hR, azi, zen, mic_radius=0.042, mic_azi=maz, mic_zen=mzn, out_shape=(128, 25, True))
    radii = [0.0470, 0.0471, 0.0473]
    for r in radii:
        jl.add(L.KIND_EMAGLS2, "real", 4, 48000.0, 128, thin["hL"], thin["hR"], azi, zen, mic_radius=r, mic_azi=maz, mic_zen=mzn, sim_order_pad=21,
               out_shape=(128, 32, False))
    for hL, hR in sets[:2]:
        jl.add(L.KIND_MAGLS, "real", 4, 48000.0, 128, hL, hR, azi, zen, out_shape=(128, 25, False))
    jl.run()
    res = jl.results()
    worst = 0.0
    for j in (0, 5, 31, 32, 36):
        w = E.getEMagLsFilters(sets[j][0], sets[j][1], azi, zen, 0.042, maz, mzn, 4, 48000.0, 128, "complex")
        worst = max(worst, rel(res[j][0], w[0]), rel(res[j][1], w[1]))
    for i, r in enumerate(radii):
        w = E.getEMagLs2Filters(thin["hL"], thin["hR"], azi, zen, r, maz, mzn, 4, 48000.0, 128, "real")
        worst = max(worst, rel(res[37 + i][0], w[0]), rel(res[37 + i][1], w[1]))
    for i in range(2):
        w = E.getMagLsFilters(sets[i][0], sets[i][1], azi, zen, 4, 48000.0, 128, "real")
        worst = max(worst, rel(res[40 + i][0], w[0]), rel(res[40 + i][1], w[1]))
    print(f"job list of 42 designs (3 kinds, 5 chunks) vs the single calls: worst rel = {worst:.3e}")
    assert worst < 1e-9
    # the same list again (resident chunks, replayed graphs), with the designs of a chunk sharing their geometry: bit-identical
    jl.run(share_geometry=True)
    res2 = jl.results()
    assert all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) for a, b in zip(res, res2))
    L.check(L.load().emagls_cache_clear())


def test_job_list_reports_a_failing_job(thin, grids):
    """A job the library cannot run (an array design without its microphone grid) fails the call with its message; the jobs before
    it may have run, nothing hangs."""
    from emagls_amd import _lib as L
    from emagls_amd.jobs import JobList
    jl = JobList()
    jl.add(L.KIND_EMAGLS, "real", 4, 48000.0, 128, thin["hL"], thin["hR"], thin["azi"], thin["zen"], mic_radius=0.042, mic_azi=grids["mic_azi"],
           mic_zen=grids["mic_zen"], out_shape=(128, 25, False))
    jl.add(L.KIND_EMAGLS, "real", 4, 48000.0, 128, thin["hL"], thin["hR"], thin["azi"], thin["zen"], mic_radius=0.05, nmics=32, out_shape=(128, 25, False))
    with pytest.raises(L.EmaglsError) as e:
        jl.run()
    assert "microphone grid" in str(e.value)
    L.check(L.load().emagls_cache_clear())


def test_design_out_shape_equals_the_plans(thin, grids):
    """emagls_design_out_shape against emagls_plan_info of a plan of the same descriptor, every kind (the runners of
    emagls_amd/batch.py and the MEX 'jobs' command size their outputs by it)."""
    import ctypes as C
    from emagls_amd import Plan, _lib as L
    lib = L.load()
    for kind, basis, order, nmics, extra in [(L.KIND_LS, "complex", 2, 0, {}), (L.KIND_MAGLS, "real", 3, 0, {}), (L.KIND_MAGLS_2D, "complex", 5, 0, {}),
                                             (L.KIND_EMAGLS, "complex", 4, 32, {}), (L.KIND_EMAGLS2, "complex", 2, 12, {}), (L.KIND_EMAGLS2, "real", 4, 40, {}),
                                             (L.KIND_EMA_CH, "real", 3, 9, {}), (L.KIND_EMA_SH, "complex", 2, 9, {}),
                                             (L.KIND_FROM_ATF, "real", 0, 5, dict(f_trans=2000.0, atf_taps=48, natf=300))]:
        p = Plan(kind, basis, order, 48000.0, 128, 64, 901, 0.042 if nmics and kind != L.KIND_FROM_ATF else 0.0, nmics, **extra)
        i = p.info()
        d = L.DesignDesc(kind, L.BASIS[basis], order, 48000.0, 128, 64, 901, 0.042 if nmics and kind != L.KIND_FROM_ATF else 0.0, nmics,
                         extra.get("f_trans", 0.0), extra.get("atf_taps", 0), extra.get("natf", 0), 0, 0, 0)
        r, c, z = C.c_int64(0), C.c_int64(0), C.c_int(0)
        L.check(lib.emagls_design_out_shape(C.byref(d), C.byref(r), C.byref(c), C.byref(z)))
        assert (r.value, c.value, bool(z.value)) == (i.out_rows, i.out_cols, bool(i.out_is_complex)), (kind, basis)
        p.close()


def test_sweep_gate_counts_launches_until_they_finish(grids, thin, monkeypatch):
    """Register-resident sweeps of different sizes in flight at once (the gate dropped a launch from its count as soon as
    ONE later launch waited for it, so a third launch could start next to a 32-design sweep that had not finished -- both only partly
    resident, spinning until the time-out).  A list of 32 + 12 + 12 + 32 + 12 designs on a small grid, four chunks in flight, the
    in-kernel wait for peers raised to 2 s: a launch that is not resident shows as a failure (status word 1 -> EMAGLS_ERR_HIP after
    the launch-per-bin re-run is refused for chunks above 16) or as seconds of run time, not as a silent fallback."""
    import time
    from emagls_amd import _lib as L
    from emagls_amd.jobs import JobList
    monkeypatch.setenv("EMAGLS_SWEEP_WAIT_MS", "2000")
    azi, zen, maz, mzn = thin["azi"], thin["zen"], grids["mic_azi"], grids["mic_zen"]
    rng = np.random.default_rng(5)
    sizes = [32, 12, 12, 32, 12]
    jl = JobList()
    order = []
    for ci, n in enumerate(sizes):     # chunks end where the shape changes: alternate two filter lengths
        length = 128 if ci % 2 == 0 else 160
        for _ in range(n):
            hL, hR = thin["hL"] * (1.0 + 0.05 * rng.standard_normal()), thin["hR"] * (1.0 + 0.05 * rng.standard_normal())
            jl.add(L.KIND_EMAGLS, "complex", 4, 48000.0, length, hL, hR, azi, zen, mic_radius=0.042, mic_azi=maz, mic_zen=mzn, out_shape=(length, 25, True))
            order.append((hL, hR, length))
    for rep in range(3):     # eager, capture, replay -- every chunk again
        t0 = time.perf_counter()
        jl.run(batch_size=32, in_flight=4)
        dt = time.perf_counter() - t0
        assert dt < 1.5, f"run {rep} took {dt:.2f} s: a sweep launch waited for workgroups that were not resident"
    res = jl.results()
    import emagls_amd as E
    worst = 0.0
    for j in (0, 31, 32, 43, 56, 87, 99):
        hL, hR, length = order[j]
        w = E.getEMagLsFilters(hL, hR, azi, zen, 0.042, maz, mzn, 4, 48000.0, length, "complex")
        worst = max(worst, rel(res[j][0], w[0]), rel(res[j][1], w[1]))
    print(f"sweeps of 32 + 12 + 12 + 32 + 12 designs in flight: worst rel vs the single calls = {worst:.3e}")
    assert worst < 1e-9
    L.check(L.load().emagls_cache_clear())


def test_job_list_over_the_devices_of_one_process(grids, thin):
    """emagls_jobs_run_devices with the one GPU of this box listed twice: the split of emagls_jobs_shard (array radii in padded lane
    batches of equal cost, whole batches per 'device'), two host threads, every job's filters written straight into its own output
    arrays -- the same filters as emagls_jobs_run on one device.  (Several distinct devices cannot be had here: MULTICHIP records.)"""
    from emagls_amd import _lib as L
    from emagls_amd.jobs import JobList
    azi, zen, maz, mzn = thin["azi"], thin["zen"], grids["mic_azi"], grids["mic_zen"]
    radii = np.linspace(0.030, 0.060, 24)      # simulation orders 14 ... 27: several classes, padded lane batches

    def build():
        jl = JobList()
        for r in radii:
            jl.add(L.KIND_EMAGLS2, "real", 4, 48000.0, 128, thin["hL"], thin["hR"], azi, zen, mic_radius=float(r), mic_azi=maz, mic_zen=mzn, out_shape=(128, 32, False))
        rng = np.random.default_rng(3)
        for _ in range(5):
            jl.add(L.KIND_MAGLS, "real", 3, 48000.0, 128, thin["hL"] * (1 + 0.1 * rng.standard_normal()), thin["hR"], azi, zen, out_shape=(128, 16, False))
        return jl
    one, two = build(), build()
    rank, pos, pad = two.shard(2, 8)
    assert set(rank) == {0, 1} and max(pad[:24]) >= 27 and all(p == 0 for p in pad[24:])
    one.run(batch_size=32, in_flight=4)
    two.run(batch_size=32, in_flight=4, devices=[0, 0])
    worst = max(max(rel(a[0], b[0]), rel(a[1], b[1])) for a, b in zip(two.results(), one.results()))
    print(f"29 jobs over devices [0, 0] vs one device: worst rel = {worst:.3e}")
    assert worst < 5e-7     # (a padded design may take another sweep form or route than the unpadded one: the synthesising sweeps' 5e-9 ... 6e-8)
    assert all(np.abs(a[0]).max() > 0 for a in two.results())
    L.check(L.load().emagls_cache_clear())
