"""GPU parity through the C ABI against the CPU oracle, tolerance 1e-6 relative complex error (BASELINE.json north_star), with the
reference's own assertAllClose metrics (verifyEMagLs.m:370-395): batches of array designs (emagls_batch_*, emagls_design_hrir_sets; the loop over HRIR sets / radii around lib/getEMagLsFilters.m:32, testEMagLs.m:75-95): lane and stream mode, geometry sharing, residency.
(Split out of tests/test_gpu_parity.py in round 6 so that `-x` loses less.)"""
import os

import numpy as np
import pytest

from oracle import emagls_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-6


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def report(name, w, o):
    nd, db, adb = O.assert_all_close_metrics(w, o)
    print(f"{name}: norm_diff={nd:.3e} max_dB={db:.3e} max|dB|={adb:.3e}")
    return nd


@pytest.fixture(scope="module")
def thin(grids, hrirs):
    sub = slice(0, 2702, 3)
    return dict(hL=hrirs[0][:, sub], hR=hrirs[1][:, sub], azi=grids["azi"][sub], zen=grids["zen"][sub])


@pytest.mark.parametrize("lanes", ["lanes", "streams"])
def test_batch_of_designs_matches_single_designs(grids, thin, monkeypatch, lanes):
    """Four designs of the same shape (two array radii x two HRIR sets) executed as one batch -- one sweep launch
    per bin for all of them -- give bit-identical filters to four separate designs, also under graph replay."""
    from emagls_amd import Batch, Plan, _lib as L, synth
    if lanes == "streams":  # per-design stages on the plans' own streams, only the sweep launch is shared
        monkeypatch.setenv("EMAGLS_BATCH_LANES", "0")
    hL2, hR2 = synth.rigid_sphere_hrirs(thin["azi"], thin["zen"], seed=99)
    jobs = [(0.042, thin["hL"], thin["hR"]), (0.040, thin["hL"], thin["hR"]), (0.042, hL2, hR2), (0.040, hL2, hR2)]
    plans, singles = [], []
    for r, hL, hR in jobs:
        def mk():
            p = Plan(L.KIND_EMAGLS, "complex", 4, 48000.0, 128, hL.shape[0], hL.shape[1], r, 32)
            p.set_hrir_grid(thin["azi"], thin["zen"])
            p.set_mic_grid(grids["mic_azi"], grids["mic_zen"])
            p.set_hrirs(hL, hR)
            return p
        q = mk()
        q.execute()
        singles.append(q.get_filters())
        q.close()
        plans.append(mk())
    b = Batch(plans)
    first = None
    for it in range(3):  # eager, captured, replayed
        b.execute()
        res = b.get_filters()
        for (wL, wR), (sL, sR) in zip(res, singles):
            # the batch sums the per-workgroup partials in a different (fixed) order than a single design
            assert rel(wL, sL) < 1e-12 and rel(wR, sR) < 1e-12, it
        if first is None:
            first = res
        else:  # deterministic: eager, captured and replayed runs agree to the bit
            for (wL, wR), (fL, fR) in zip(res, first):
                assert np.array_equal(wL, fL) and np.array_equal(wR, fR), it
    oL, oR = O.getEMagLsFilters(jobs[3][1], jobs[3][2], thin["azi"], thin["zen"], 0.040, grids["mic_azi"], grids["mic_zen"], 4,
                                48000.0, 128, "complex")
    assert rel(res[3][0], oL) < TOL and rel(res[3][1], oR) < TOL
    b.close()
    for p in plans:
        p.close()


def test_lane_batch_matches_single_designs(grids, thin):
    """Designs of identical shape (same array radius, hence the same simulation order) are executed in lane mode:
    every launch of the pipeline covers the whole batch.  Each design has its own HRIR grid (rotated), HRIR set and
    microphone grid, so every per-design stage differs between the lanes."""
    from emagls_amd import Batch, Plan, _lib as L, synth
    jobs = []
    for j in range(3):
        azi = np.mod(thin["azi"] + 0.37 * j, 2 * np.pi)
        hL, hR = synth.rigid_sphere_hrirs(azi, thin["zen"], seed=7 + j)
        jobs.append((azi, hL, hR, np.mod(grids["mic_azi"] + 0.2 * j, 2 * np.pi)))

    def mk(job):
        azi, hL, hR, maz = job
        p = Plan(L.KIND_EMAGLS, "complex", 4, 48000.0, 128, hL.shape[0], hL.shape[1], 0.042, 32)
        p.set_hrir_grid(azi, thin["zen"])
        p.set_mic_grid(maz, grids["mic_zen"])
        p.set_hrirs(hL, hR)
        return p

    singles = []
    for job in jobs:
        q = mk(job)
        q.execute()
        singles.append(q.get_filters())
        q.close()
    plans = [mk(job) for job in jobs]
    b = Batch(plans)
    first = None
    for it in range(3):  # eager, captured, replayed
        b.execute()
        res = b.get_filters()
        for (wL, wR), (sL, sR) in zip(res, singles):
            assert rel(wL, sL) < 1e-12 and rel(wR, sR) < 1e-12, it
        if first is None:
            first = res
        else:
            for (wL, wR), (fL, fR) in zip(res, first):
                assert np.array_equal(wL, fL) and np.array_equal(wR, fR), it
    # the stages before the sweep forked onto four streams (emagls_batch_set_streams; what a job list of one chunk runs): eager,
    # captured with the forks, replayed -- the same filters as on one stream to rounding (since round 5 the forked form takes its
    # Gram-route Jacobi bins one by one and the lane groups in warm-started runs of two: not bitwise the same), and bitwise the same
    # from execute to execute
    b.set_streams(4)
    forked = None
    for it in range(3):
        b.execute()
        got = b.get_filters()
        for (wL, wR), (fL, fR) in zip(got, first):
            assert rel(wL, fL) < 1e-12 and rel(wR, fR) < 1e-12, it
        if forked is None:
            forked = got
        else:
            for (wL, wR), (fL, fR) in zip(got, forked):
                assert np.array_equal(wL, fL) and np.array_equal(wR, fR), it
    b.set_streams(1)
    b.execute()
    for (wL, wR), (fL, fR) in zip(b.get_filters(), first):
        assert np.array_equal(wL, fL) and np.array_equal(wR, fR)
    # the plans still work on their own after the batch moved their buffers into its arena
    plans[1].execute()
    wL, wR = plans[1].get_filters()
    assert rel(wL, singles[1][0]) < 1e-12 and rel(wR, singles[1][1]) < 1e-12
    azi, hL, hR, maz = jobs[2]
    oL, oR = O.getEMagLsFilters(hL, hR, azi, thin["zen"], 0.042, maz, grids["mic_zen"], 4, 48000.0, 128, "complex")
    assert rel(res[2][0], oL) < TOL and rel(res[2][1], oR) < TOL
    b.close()
    for p in plans:
        p.close()


def test_gram_matrix_of_a_lane_batch(grids, thin):
    """Lane batches with enough Gram tiles to fill the chip (28 tiles x 6 designs here) take the LDS-staged Gram kernel without a
    K split (gram_lds_kernel): the Gram matrix of every design against NumPy on the design's own conj(Y), its leading block
    copied for the Cholesky factorisation, and the filters against the single designs (which take the K-split kernel)."""
    from emagls_amd import Batch, Plan, _lib as L, synth
    plans, singles = [], []
    for j in range(6):
        azi = np.mod(thin["azi"] + 0.21 * j, 2 * np.pi)
        hL, hR = synth.rigid_sphere_hrirs(azi, thin["zen"], seed=70 + j)
        p = Plan(L.KIND_EMAGLS, "real", 4, 48000.0, 128, hL.shape[0], hL.shape[1], 0.042, 32)
        p.set_hrir_grid(azi, thin["zen"])
        p.set_mic_grid(grids["mic_azi"], grids["mic_zen"])
        p.set_hrirs(hL, hR)
        p.execute()
        singles.append(p.get_filters())
        plans.append(p)
    b = Batch(plans)
    assert b.lane_mode()
    b.execute()
    res = b.get_filters()
    i = plans[0].info()
    S, D = i.num_sh_sim, thin["azi"].size
    ldS = -(-S // 64) * 64
    worst = 0.0
    for j in (0, 3, 5):
        Yc = plans[j].debug("Yc", np.float64).reshape(-1, ldS)[:D, :S]
        Gy = plans[j].debug("Gy", np.float64, (S, S))
        ref = Yc.T @ Yc
        blk = (np.arange(S)[:, None] // 64) <= (np.arange(S)[None, :] // 64)      # the upper block triangle is what is formed
        worst = max(worst, np.abs(Gy - ref)[blk].max() / np.abs(ref).max())
        assert np.all(Gy[~blk] == 0.0)
    print(f"Gram matrix of a 6-design lane batch (LDS-staged kernel) vs NumPy: rel = {worst:.3e}")
    assert worst < 1e-13
    for (wL, wR), (sL, sR) in zip(res, singles):
        assert rel(wL, sL) < 1e-11 and rel(wR, sR) < 1e-11
    b.close()
    for p in plans:
        p.close()


def test_twenty_hrir_sets_on_one_geometry_take_the_register_resident_sweep(grids, thin):
    """Geometry sharing in batches of more than 16 sets (round 5: bench.py's secondary figure runs batches of 32, 4.5 k sets/s against
    3.1 k with 16): the geometry stages once, ONE register-resident sweep launch for all sets -- the same filters as the single designs
    (which take the slab form of the sweep: to rounding) and as the same batch without sharing."""
    import ctypes
    from emagls_amd import Batch, Plan, _lib as L
    lib = L.load()
    rng = np.random.default_rng(77)
    plans, singles = [], []
    for j in range(20):
        hL = thin["hL"] * (1.0 + 0.03 * j) + 1e-3 * rng.standard_normal(thin["hL"].shape)
        hR = thin["hR"] * (1.0 - 0.02 * j) + 1e-3 * rng.standard_normal(thin["hR"].shape)
        p = Plan(L.KIND_EMAGLS, "complex", 4, 48000.0, 128, hL.shape[0], hL.shape[1], grids["mic_radius"], 32)
        p.set_hrir_grid(thin["azi"], thin["zen"])
        p.set_mic_grid(grids["mic_azi"], grids["mic_zen"])
        p.set_hrirs(hL, hR)
        if j in (0, 7, 19):
            p.execute()
            singles.append((j, p.get_filters()))
        plans.append(p)
    prev = ctypes.c_int(0)
    L.check(lib.emagls_set_batch_max(32, ctypes.byref(prev)))
    try:
        b = Batch(plans)
    finally:
        L.check(lib.emagls_set_batch_max(prev.value, None))
    b.execute()
    indep = b.get_filters()
    assert plans[0].info().sweep_form == 3
    b.share_geometry(True)
    b.execute()
    assert b.shares_geometry() and plans[1].info().num_sweep_launches == 1
    shared = b.get_filters()
    worst_s = max(max(rel(shared[j][0], w[0]), rel(shared[j][1], w[1])) for j, w in singles)
    worst_i = max(max(rel(a[0], c[0]), rel(a[1], c[1])) for a, c in zip(shared, indep))
    print(f"20 HRIR sets on one geometry, register-resident sweep: vs single plans {worst_s:.3e}, vs the same batch unshared {worst_i:.3e}")
    assert worst_s < 2e-7 and worst_i < 1e-9
    b.close()
    for p in plans:
        p.close()


@pytest.mark.parametrize("kind", ["emagls", "emagls2", "emainch"])
def test_hrir_sets_on_one_geometry_share_it(grids, thin, kind):
    """Batches of HRIR sets on one geometry (the loop over subjects around getEMagLsFilters with the same grids and array):
    with Batch.share_geometry() the SH matrices, the array model, pwGrid_k and its regularised inverses run once (plan 0) and
    every other plan only runs what its HRIRs enter -- same filters as the single designs; replays are bitwise reproducible;
    a plan whose microphone grid is replaced afterwards makes the batch fall back to independent designs."""
    import ctypes
    from emagls_amd import Batch, Plan, _lib as L
    rng = np.random.default_rng(31)
    K = {"emagls": L.KIND_EMAGLS, "emagls2": L.KIND_EMAGLS2, "emainch": L.KIND_EMA_CH}[kind]
    order, nm = (4, 32) if kind != "emainch" else (3, 9)
    maz = grids["mic_azi"] if kind != "emainch" else np.linspace(0, 2 * np.pi, nm, endpoint=False) + 0.2
    plans, singles = [], []
    n = 6
    for j in range(n):
        hL = thin["hL"] * (1.0 + 0.07 * j) + 1e-3 * rng.standard_normal(thin["hL"].shape)
        hR = thin["hR"] * (1.0 - 0.03 * j) + 1e-3 * rng.standard_normal(thin["hR"].shape)
        p = Plan(K, "complex", order, 48000.0, 128, hL.shape[0], hL.shape[1], grids["mic_radius"], nm)
        p.set_hrir_grid(thin["azi"], thin["zen"])
        p.set_mic_grid(maz, None if kind == "emainch" else grids["mic_zen"])
        p.set_hrirs(hL, hR)
        p.execute()
        singles.append(p.get_filters())
        plans.append(p)
    b = Batch(plans)
    b.execute()
    assert not b.shares_geometry()                        # off by default: independent designs
    indep = b.get_filters()
    b.share_geometry(True)
    outs = []
    for it in range(3):
        b.execute()
        assert b.shares_geometry()
        outs.append(b.get_filters())
    assert plans[1].info().num_sweep_launches == 1
    worst = max(max(rel(o[0], s_[0]), rel(o[1], s_[1])) for o, s_ in zip(outs[0], singles))
    worst_i = max(max(rel(o[0], s_[0]), rel(o[1], s_[1])) for o, s_ in zip(outs[0], indep))
    print(f"{kind}: {n} HRIR sets on one geometry vs single plans: worst rel = {worst:.3e}; vs the same batch unshared {worst_i:.3e}")
    assert worst < 1e-12 and worst_i < 1e-9      # (a lane batch warm-starts its Jacobi runs differently from a single design)
    for it in (1, 2):
        for a, c in zip(outs[0], outs[it]):
            assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1])
    assert rel(singles[0][0], singles[3][0]) > 1e-3
    # one plan leaves the common geometry: the batch runs its designs independently again, with the new array
    plans[2].set_mic_grid(maz + 0.11, None if kind == "emainch" else grids["mic_zen"])
    b.execute()
    assert not b.shares_geometry()
    moved = b.get_filters()
    plans[2].execute()
    ref2 = plans[2].get_filters()
    assert max(rel(moved[2][0], ref2[0]), rel(moved[2][1], ref2[1])) < 1e-9
    assert max(rel(moved[0][0], singles[0][0]), rel(moved[5][1], singles[5][1])) < 1e-9
    assert rel(moved[2][0], singles[2][0]) > 1e-4
    b.close()
    for p in plans:
        p.close()


@pytest.mark.parametrize("kind", ["ls", "magls", "magls2d", "emagls", "emagls2", "emainch", "emainsh"])
def test_design_hrir_sets_in_one_call(grids, thin, kind):
    """emagls_design_hrir_sets: the loop over HRIR sets around a design function as ONE C call (3-D arrays in and out; plans and
    geometry-sharing batches of up to 16 sets inside, kept for the next call).  19 sets = a batch of 16 and a tail batch of 3;
    every set equals its single call; a second call of the same shape reuses the cached plans."""
    import emagls_amd as E
    rng = np.random.default_rng(51)
    nsets = 19
    if kind == "magls2d":
        from emagls_amd import synth
        azi = np.sort(np.mod(np.linspace(0, 2 * np.pi, 300, endpoint=False) + 0.002 * rng.standard_normal(300), 2 * np.pi))
        zen, base = None, synth.rigid_sphere_hrirs(azi, np.full(300, np.pi / 2))
    else:
        azi, zen, base = thin["azi"], thin["zen"], (thin["hL"], thin["hR"])
    hL = np.stack([base[0] * (1 + 0.03 * j) + 1e-3 * rng.standard_normal(base[0].shape) for j in range(nsets)], axis=2)
    hR = np.stack([base[1] * (1 - 0.02 * j) + 1e-3 * rng.standard_normal(base[1].shape) for j in range(nsets)], axis=2)
    order = {"magls2d": 5, "emainch": 3, "emainsh": 2}.get(kind, 4)
    ma = np.linspace(0, 2 * np.pi, 9, endpoint=False) + 0.2 if kind in ("emainch", "emainsh") else grids["mic_azi"]
    mz = None if kind in ("emainch", "emainsh") else grids["mic_zen"]
    if kind == "emainsh":   # (plan by plan, no batches: fewer sets)
        nsets, hL, hR = 6, hL[:, :, :6], hR[:, :, :6]
    kw = dict(order=order, fs=48000.0, len=128, shDefinition="complex")
    if kind in ("emagls", "emagls2", "emainch", "emainsh"):
        kw.update(micRadius=grids["mic_radius"], micGridAziRad=ma, micGridZenRad=mz)
    single = {"ls": lambda a, b: E.getLsFilters(a, b, azi, zen, order, "complex"),
              "magls": lambda a, b: E.getMagLsFilters(a, b, azi, zen, order, 48000.0, 128, "complex"),
              "magls2d": lambda a, b: E.getMagLsFilters2D(a, b, azi, order, 48000.0, 128, "complex"),
              "emagls": lambda a, b: E.getEMagLsFilters(a, b, azi, zen, grids["mic_radius"], ma, mz, order, 48000.0, 128, "complex"),
              "emagls2": lambda a, b: E.getEMagLs2Filters(a, b, azi, zen, grids["mic_radius"], ma, mz, order, 48000.0, 128, "complex"),
              "emainch": lambda a, b: E.getEMagLsFiltersEMAinCH(a, b, azi, zen, grids["mic_radius"], ma, order, 48000.0, 128, "complex"),
              "emainsh": lambda a, b: E.getEMagLsFiltersEMAinSH(a, b, azi, zen, grids["mic_radius"], ma, order, 48000.0, 128, "complex")}[kind]
    for rep in range(2):
        wL, wR = E.designHrirSets(kind, hL, hR, azi, zen, **kw)
        worst = 0.0
        for j in ((0, 7, 15, 16, 18) if nsets == 19 else (0, 3, 4, 5)):
            sL, sR = single(hL[:, :, j], hR[:, :, j])
            assert wL[:, :, j].shape == sL.shape and wL.dtype == sL.dtype
            worst = max(worst, rel(wL[:, :, j], sL), rel(wR[:, :, j], sR))
        print(f"{kind}: {nsets} HRIR sets in one call (pass {rep}): worst rel vs single calls = {worst:.3e}")
        assert worst < 1e-9
    assert rel(wL[:, :, 0], wL[:, :, nsets // 2]) > 1e-3


@pytest.mark.parametrize("kind,order,nmics", [("ls", 6, 0), ("magls", 5, 0), ("emagls2", 4, 40)])
def test_design_hrir_sets_above_32_channels(thin, kind, order, nmics):
    """Designs with more than 32 channels (LS / MagLS orders 5-7, arrays of 33-64 microphones) do not enter batches; the HRIR-set
    job list runs their chunks plan by plan -- the same filters as the single calls (the header's promise; round 3 returned
    EMAGLS_ERR_UNSUPPORTED as soon as nsets > 1).  5 sets = a chunk of four and a tail of one; also through the multi-GPU job
    runner (one process)."""
    import emagls_amd as E
    from emagls_amd import synth
    rng = np.random.default_rng(77)
    nsets = 5
    azi, zen = thin["azi"], thin["zen"]
    hL = np.stack([thin["hL"] * (1 + 0.03 * j) + 1e-3 * rng.standard_normal(thin["hL"].shape) for j in range(nsets)], axis=2)
    hR = np.stack([thin["hR"] * (1 - 0.02 * j) + 1e-3 * rng.standard_normal(thin["hR"].shape) for j in range(nsets)], axis=2)
    kw = dict(order=order, fs=48000.0, len=128, shDefinition="real")
    if kind == "emagls2":
        maz, mzn = synth.fibonacci_grid(nmics)
        kw.update(micRadius=0.042, micGridAziRad=maz, micGridZenRad=mzn)
        single = lambda a, b: E.getEMagLs2Filters(a, b, azi, zen, 0.042, maz, mzn, order, 48000.0, 128, "real")
    elif kind == "magls":
        single = lambda a, b: E.getMagLsFilters(a, b, azi, zen, order, 48000.0, 128, "real")
    else:
        single = lambda a, b: E.getLsFilters(a, b, azi, zen, order, "real")
    wL, wR = E.designHrirSets(kind, hL, hR, azi, zen, **kw)
    worst = 0.0
    for j in range(nsets):
        sL, sR = single(hL[:, :, j], hR[:, :, j])
        assert wL[:, :, j].shape == sL.shape
        worst = max(worst, rel(wL[:, :, j], sL), rel(wR[:, :, j], sR))
    print(f"{kind} order {order} ({wL.shape[1]} channels): {nsets} HRIR sets in one call, chunks run plan by plan: worst rel vs single calls = {worst:.3e}")
    assert worst < 1e-12
    if kind == "magls":
        from emagls_amd.batch import magls_hrir_sets
        res = magls_hrir_sets([(hL[:, :, j], hR[:, :, j]) for j in range(nsets)], azi, zen, order, 48000.0, 128, "real", max_batch=3)
        worst = max(max(rel(res[j][0], wL[:, :, j]), rel(res[j][1], wR[:, :, j])) for j in range(nsets))
        print(f"magls order {order} through emagls_amd.batch.magls_hrir_sets (chunks of 3, plan by plan): worst rel = {worst:.3e}")
        assert worst < 1e-12


def test_design_hrir_sets_alternating_plan_sets(thin):
    """40 sets = two full chunks (which alternate between two sets of plans, the second chunk's upload overlapping the first
    chunk's compute) and a tail of 8: every chunk lands in its place."""
    import emagls_amd as E
    rng = np.random.default_rng(52)
    nsets = 40
    hL = np.stack([thin["hL"] * (1 + 0.01 * j) + 1e-3 * rng.standard_normal(thin["hL"].shape) for j in range(nsets)], axis=2)
    hR = np.stack([thin["hR"] * (1 - 0.01 * j) for j in range(nsets)], axis=2)
    for rep in range(2):
        wL, wR = E.designHrirSets("magls", hL, hR, thin["azi"], thin["zen"], order=3, fs=48000.0, len=128, shDefinition="real")
        worst = 0.0
        for j in (0, 15, 16, 31, 32, 39):
            sL, sR = E.getMagLsFilters(hL[:, :, j], hR[:, :, j], thin["azi"], thin["zen"], 3, 48000.0, 128, "real")
            worst = max(worst, rel(wL[:, :, j], sL), rel(wR[:, :, j], sR))
        print(f"40 HRIR sets in one call (pass {rep}): worst rel vs single calls = {worst:.3e}")
        assert worst < 1e-12
    assert rel(wL[:, :, 3], wL[:, :, 30]) > 1e-3


def test_geometry_sharing_with_twelve_hrir_sets_and_kinds_without_the_option(grids, thin):
    """9-16 HRIR sets share one sweep launch (twin workgroups) on plan 0's operands; a kind without the option (EMAinSH) accepts the switch
    and runs as before."""
    import ctypes
    from emagls_amd import Batch, Plan, _lib as L
    rng = np.random.default_rng(32)
    lib = L.load()
    prev = ctypes.c_int(0)
    L.check(lib.emagls_set_batch_max(16, ctypes.byref(prev)))
    try:
        plans, singles = [], []
        for j in range(12):
            hL = thin["hL"] * (1.0 + 0.05 * j) + 1e-3 * rng.standard_normal(thin["hL"].shape)
            p = Plan(L.KIND_EMAGLS, "real", 4, 48000.0, 128, hL.shape[0], hL.shape[1], grids["mic_radius"], 32)
            p.set_hrir_grid(thin["azi"], thin["zen"])
            p.set_mic_grid(grids["mic_azi"], grids["mic_zen"])
            p.set_hrirs(hL, thin["hR"])
            p.execute()
            singles.append(p.get_filters())
            plans.append(p)
        b = Batch(plans)
    finally:
        L.check(lib.emagls_set_batch_max(prev.value, None))
    b.share_geometry(True)
    b.execute()
    out = b.get_filters()
    assert b.shares_geometry() and plans[0].info().num_sweep_launches == 1
    worst = max(max(rel(o[0], s_[0]), rel(o[1], s_[1])) for o, s_ in zip(out, singles))
    print(f"12 HRIR sets on one geometry vs single plans: worst rel = {worst:.3e}")
    assert worst < 1e-12
    b.close()
    for p in plans:
        p.close()
    mp = []
    ma = np.linspace(0, 2 * np.pi, 9, endpoint=False) + 0.2
    for j in range(3):
        p = Plan(L.KIND_EMA_SH, "real", 2, 48000.0, 128, thin["hL"].shape[0], thin["hL"].shape[1], grids["mic_radius"], 9)
        p.set_hrir_grid(thin["azi"], thin["zen"])
        p.set_mic_grid(ma, None)
        p.set_hrirs(thin["hL"] * (1 + 0.1 * j), thin["hR"])
        p.execute()
        mp.append((p, p.get_filters()))
    b = Batch([p for p, _ in mp])
    b.share_geometry(True)
    b.execute()
    assert not b.shares_geometry()
    for (p, ref), o in zip(mp, b.get_filters()):
        assert rel(o[0], ref[0]) < 1e-10
    b.close()
    for p, _ in mp:
        p.close()


def test_sixteen_design_lane_batch(grids, thin, monkeypatch):
    """9 to 16 designs (opt-in: emagls_set_batch_max(16), the product's default stays 8 and so does the suite's) share one sweep
    launch with two designs per XCD: a batch of 12 designs (different HRIR sets and microphone grids) equals the single designs
    and sweeps with the persistent kernel; replays are bitwise reproducible.  Both forms of the launch: twin workgroups (two
    slabs of one design per CU, the default) and two independent workgroups per CU (EMAGLS_SWEEP_TWIN=0).  Without the opt-in
    a batch holds at most 8."""
    import ctypes
    from emagls_amd import Batch, Plan, _lib as L
    from emagls_amd._lib import EmaglsError
    rng = np.random.default_rng(21)
    plans, singles = [], []
    for j in range(12):
        hL = thin["hL"] * (1.0 + 0.1 * j) + 1e-3 * rng.standard_normal(thin["hL"].shape)
        hR = thin["hR"] * (1.0 - 0.02 * j)
        maz = grids["mic_azi"] + 0.05 * j
        p = Plan(L.KIND_EMAGLS, "complex", 4, 48000.0, 128, hL.shape[0], hL.shape[1], grids["mic_radius"], 32)
        p.set_hrir_grid(thin["azi"], thin["zen"])
        p.set_mic_grid(maz, grids["mic_zen"])
        p.set_hrirs(hL, hR)
        p.execute()
        singles.append(p.get_filters())
        plans.append(p)
    lib = L.load()
    prev = ctypes.c_int(0)
    L.check(lib.emagls_set_batch_max(8, ctypes.byref(prev)))
    with pytest.raises(EmaglsError, match="at most 8 designs"):
        Batch(plans)
    L.check(lib.emagls_set_batch_max(16, None))
    try:
        b = Batch(plans)
    finally:
        L.check(lib.emagls_set_batch_max(prev.value, None))
    outs = []
    for it in range(3):
        b.execute()
        outs.append(b.get_filters())
    assert plans[0].info().num_sweep_launches == 1     # one persistent launch, not the launch-per-bin fallback
    worst = max(max(rel(o[0], s_[0]), rel(o[1], s_[1])) for o, s_ in zip(outs[0], singles))
    print(f"12-design lane batch vs single plans: worst rel = {worst:.3e}")
    assert worst < 1e-12
    for it in (1, 2):
        for a, c in zip(outs[0], outs[it]):
            assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1])
    assert rel(singles[0][0], singles[5][0]) > 1e-3
    monkeypatch.setenv("EMAGLS_SWEEP_TWIN", "0")
    b.execute()
    plain = b.get_filters()
    monkeypatch.delenv("EMAGLS_SWEEP_TWIN")
    assert plans[0].info().num_sweep_launches == 1
    worst = max(max(rel(o[0], s_[0]), rel(o[1], s_[1])) for o, s_ in zip(plain, singles))
    print(f"12-design lane batch, two workgroups per CU, vs single plans: worst rel = {worst:.3e}")
    assert worst < 1e-12
    b.close()
    for p in plans:
        p.close()


def test_gram_tile_on_the_four_block_mfma_shape(grids, thin, monkeypatch):
    """The Gram product of a lane batch on v_mfma_f64_4x4x4_4b (EMAGLS_GRAM_MFMA4=1; gram_chol.hip: measured slower than the
    16 x 16 x 4 kernel in this pipeline, so off by default) against the default kernel: the same filters to rounding."""
    from emagls_amd import Batch, Plan, _lib as L, synth
    res = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("EMAGLS_GRAM_MFMA4", mode)
        plans = []
        for j in range(5):   # (28 tiles x 5 designs: the LDS-staged tile kernels take over from 128 workgroups on)
            azi = np.mod(grids["azi"] + 0.21 * j, 2 * np.pi)   # (the full 2702-point grid: the thin ones take the K-split kernel)
            hL, hR = synth.rigid_sphere_hrirs(azi, grids["zen"], taps=64, seed=11 + j)
            p = Plan(L.KIND_EMAGLS, "real", 4, 48000.0, 128, hL.shape[0], hL.shape[1], 0.042, 32)
            p.set_hrir_grid(azi, grids["zen"])
            p.set_mic_grid(grids["mic_azi"], grids["mic_zen"])
            p.set_hrirs(hL, hR)
            plans.append(p)
        b = Batch(plans)
        b.execute()
        res[mode] = b.get_filters()
        b.close()
        for p in plans:
            p.close()
    worst = max(max(rel(a[0], c[0]), rel(a[1], c[1])) for a, c in zip(res["0"], res["1"]))
    print(f"Gram tile on the 4 x 4 x 4 shape vs the 16 x 16 x 4 kernel: worst rel = {worst:.3e}")
    # (bit-identical, as it turns out: both shapes contract four rows per instruction in the same order; the kernels themselves against
    # a host sum: tests/test_gpu_stages.py::test_gram_tile_kernels_against_a_host_sum)
    assert worst < 1e-9


def test_residency_is_decided_before_the_launch(grids, thin, monkeypatch):
    """A resident sweep needs all its workgroups on the device at once.  Whether they fit is decided BEFORE the launch from the
    runtime's occupancy figure of the kernel variant and the CUs of an XCD (EMAGLS_CU_BUDGET stands in for a CU-masked queue or
    a shared GPU): a design or a batch that cannot be resident takes the launch-per-bin sweep at once -- no wait for peers until
    a time-out (the 0.2 s stall of earlier rounds), same filters."""
    import ctypes
    import time
    from emagls_amd import Batch, Plan, _lib as L

    def plan(j=0):
        p = Plan(L.KIND_EMAGLS, "complex", 4, 48000.0, 128, thin["hL"].shape[0], thin["hL"].shape[1], 0.042, 32)
        p.set_hrir_grid(thin["azi"], thin["zen"])
        p.set_mic_grid(grids["mic_azi"] + 0.05 * j, grids["mic_zen"])
        p.set_hrirs(thin["hL"] * (1.0 + 0.1 * j), thin["hR"])
        return p

    p = plan()
    p.execute()
    ref = p.get_filters()
    assert p.info().sweep_form == 2 and p.info().num_sweep_launches == 1   # resident, operands evaluated in the launch
    p.close()
    # 901 directions = 15 workgroups per design on one XCD: 4 CUs per XCD cannot hold them
    monkeypatch.setenv("EMAGLS_CU_BUDGET", "32")
    p = plan()
    assert p.info().sweep_form == 0
    t0 = time.perf_counter()
    p.execute()
    out = p.get_filters()
    dt = time.perf_counter() - t0
    assert p.info().num_sweep_launches > 1
    p.close()
    print(f"one design without room for a resident sweep: launch per bin from the start, first execute {dt * 1e3:.1f} ms, "
          f"rel vs the resident form = {max(rel(out[0], ref[0]), rel(out[1], ref[1])):.3e}")
    # (no wall-clock bound: the sweep form and the launch count already show that no time-out path was taken)
    assert rel(out[0], ref[0]) < 1e-6 and rel(out[1], ref[1]) < 1e-6
    # 8 CUs per XCD: one design fits; the 12 designs of a batch (two designs per XCD) fit in the register-resident form (8 workgroups
    # per design, three per CU) and do not in the slab form (15 workgroups per design, two per CU)
    monkeypatch.setenv("EMAGLS_CU_BUDGET", "64")
    lib = L.load()
    for reg in ("0", "1"):
        monkeypatch.setenv("EMAGLS_SWEEP_REG", reg)
        plans = [plan(j) for j in range(12)]
        singles = []
        for q in plans:
            assert q.info().sweep_form == 2
            q.execute()
            singles.append(q.get_filters())
        prev = ctypes.c_int(0)
        L.check(lib.emagls_set_batch_max(16, ctypes.byref(prev)))
        try:
            b = Batch(plans)
        finally:
            L.check(lib.emagls_set_batch_max(prev.value, None))
        assert plans[0].info().sweep_form == (3 if reg == "1" else 0)
        t0 = time.perf_counter()
        b.execute()
        outs = b.get_filters()
        dt = time.perf_counter() - t0
        assert (plans[0].info().num_sweep_launches > 1) == (reg == "0")
        worst = max(max(rel(o[0], s_[0]), rel(o[1], s_[1])) for o, s_ in zip(outs, singles))
        print(f"12-design batch on 64 CUs, EMAGLS_SWEEP_REG={reg}: sweep form {plans[0].info().sweep_form}, first execute {dt * 1e3:.1f} ms, "
              f"worst rel vs the single designs = {worst:.3e}")
        assert worst < 1e-6
        b.close()
        for q in plans:
            q.close()


def test_batch_of_designs_with_the_diffuseness_constraint(grids, thin):
    """The constraint's kernel in lane mode (grid.z = design): a batch of eMagLS plans with `diffuseness` equals the one-shot
    calls with applyDiffusenessConst, and differs from the unconstrained design."""
    import emagls_amd as E
    from emagls_amd import Batch, Plan, _lib as L
    sets = [dict(hL=np.ascontiguousarray(thin["hL"]) if j == 0 else np.ascontiguousarray(thin["hL"][:, ::-1]),
                 hR=np.ascontiguousarray(thin["hR"]) if j == 0 else np.ascontiguousarray(thin["hR"][:, ::-1])) for j in range(2)]
    plans = []
    for s_ in sets:
        p = Plan(L.KIND_EMAGLS, "real", 4, 48000.0, 128, thin["hL"].shape[0], thin["hL"].shape[1], grids["mic_radius"], 32, diffuseness=True)
        p.set_hrir_grid(thin["azi"], thin["zen"])
        p.set_mic_grid(grids["mic_azi"], grids["mic_zen"])
        p.set_hrirs(s_["hL"], s_["hR"])
        plans.append(p)
    b = Batch(plans)
    for it in range(3):
        b.execute()
        res = b.get_filters()
    for (wL, wR), s_ in zip(res, sets):
        args = (s_["hL"], s_["hR"], thin["azi"], thin["zen"], grids["mic_radius"], grids["mic_azi"], grids["mic_zen"], 4, 48000.0, 128, "real")
        sL, sR = E.getEMagLsFilters(*args, applyDiffusenessConst=True)
        uL, _ = E.getEMagLsFilters(*args)
        assert rel(wL, sL) < 1e-12 and rel(wR, sR) < 1e-12 and rel(wL, uL) > 1e-3
    b.close()
    for p in plans:
        p.close()
