"""BASELINE config 4 as specified: getEMagLs2Filters on the raw 32-microphone em32, 2702 directions, for array radii
spread over 2..10 cm.  The simulation order max(4, ceil(fs*pi*r/343)) reaches 44 (S = 2025 SH channels) at 10 cm
(dependencies/getSMAIRMatrix.m:95).  GPU path against the oracle and, for the job batching (emagls_amd.batch.shard_jobs /
lane_groups + Batch), lane batches against one-shot designs."""
import numpy as np
import pytest

from oracle import emagls_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-6


@pytest.fixture(scope="module")
def hrirs64(grids):
    """64-tap HRIRs: the oracle needs 0.35 s per bin at simulation order 44, so its live comparisons use 64 solved bins."""
    from emagls_amd import synth
    return synth.rigid_sphere_hrirs(grids["azi"], grids["zen"], taps=64, centre_delay=16.0)


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def test_every_radius_of_the_config4_batch_is_supported(grids):
    """No EMAGLS_ERR_UNSUPPORTED anywhere in linspace(0.02, 0.10, 256) at 1024 taps; the routes of the per-bin factorisation
    keep the orthonormal (Householder) route within its tile: at most 27 orders there, whatever the simulation order."""
    from emagls_amd import Plan, _lib as L
    from emagls_amd.batch import simulation_order
    seen = set()
    for r in np.linspace(0.02, 0.10, 256):
        so = simulation_order(4, 48000.0, r, raw=True)
        if so in seen:
            continue   # one plan per shape class (a plan allocates its buffers: 36 classes instead of 256 radii)
        seen.add(so)
        p = Plan(L.KIND_EMAGLS2, "real", 4, 48000.0, 1024, 128, 2702, float(r), 32)
        i = p.info()
        p.close()
        assert i.sim_order == so and i.num_sh_sim == (so + 1) ** 2
        assert 1 <= i.gram_from < i.num_pos_freqs and i.hh_orders <= 27 and i.hh_orders <= so + 1
    assert min(seen) == 9 and max(seen) == 44 and len(seen) == 36


@pytest.mark.parametrize("radius,length", [(0.10, 64), (0.0875, 64), (0.062, 128)])
def test_emagls2_large_radius_vs_oracle(grids, hrirs64, radius, length):
    """Simulation orders 44 / 39 / 28 (S = 2025 / 1600 / 841): all of them beyond the 768-row tile of the S-space route, served
    by the Gram route on all orders plus the orthonormal route on the orders that are above rounding noise in the low bins."""
    import emagls_amd as E
    hrirs = hrirs64
    args = (hrirs[0], hrirs[1], grids["azi"], grids["zen"], radius, grids["mic_azi"], grids["mic_zen"], 4, 48000.0, length, "real")
    wL, wR = E.getEMagLs2Filters(*args)
    oL, oR = O.getEMagLs2Filters(*args)
    assert wL.shape == (length, 32) and wL.dtype == np.float64
    nd = O.assert_all_close_metrics(np.hstack([wL, wR]), np.hstack([oL, oR]))
    print(f"eMagLS2 r={radius} len={length}: rel L {rel(wL, oL):.3e} R {rel(wR, oR):.3e} norm_diff={nd[0]:.3e} max|dB|={nd[2]:.3e}")
    assert rel(wL, oL) < TOL and rel(wR, oR) < TOL


def test_emagls2_config4_full_size_r10cm(grids, hrirs):
    """One job of the config-4 batch at the far end of the radius range, full size: r = 10 cm, 1024 taps (nfft 2048, 1024 solved
    bins, k_cut 86), simulation order 44.  The oracle needs minutes for it: its output is a stored vector
    (tests/golden/oracle_vectors.npz, written by tests/golden/make_oracle_vectors.py from the same seeded inputs)."""
    import os
    import emagls_amd as E
    vec = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_vectors.npz"))
    args = (hrirs[0], hrirs[1], grids["azi"], grids["zen"], 0.10, grids["mic_azi"], grids["mic_zen"], 4, 48000.0, 1024, "real")
    wL, wR = E.getEMagLs2Filters(*args)
    oL, oR = vec["config4_r100mm_len1024/wL"], vec["config4_r100mm_len1024/wR"]
    assert wL.shape == oL.shape == (1024, 32)
    print(f"eMagLS2 config 4, r = 10 cm, 1024 taps: rel L {rel(wL, oL):.3e} R {rel(wR, oR):.3e}")
    assert rel(wL, oL) < TOL and rel(wR, oR) < TOL


def test_emagls_sh_domain_large_radius(grids, hrirs64):
    """The SH-domain design (getEMagLsFilters, 25 channels, complex basis) on an 8 cm array: simulation order 36."""
    import emagls_amd as E
    hrirs = hrirs64
    args = (hrirs[0], hrirs[1], grids["azi"], grids["zen"], 0.08, grids["mic_azi"], grids["mic_zen"], 4, 48000.0, 64, "complex")
    wL, wR = E.getEMagLsFilters(*args)
    oL, oR = O.getEMagLsFilters(*args)
    print(f"eMagLS r = 8 cm: rel L {rel(wL, oL):.3e} R {rel(wR, oR):.3e}")
    assert rel(wL, oL) < TOL and rel(wR, oR) < TOL


def test_radius_sweep_through_job_batching(grids, hrirs64):
    """shard_jobs + lane_groups + Batch over 16 radii spanning 2..10 cm (eight shape classes, two radii each, so that every
    lane batch holds two designs): each job equals its one-shot design."""
    import emagls_amd as E
    from emagls_amd import Batch, Plan, _lib as L
    from emagls_amd.batch import lane_groups, shard_jobs, simulation_order
    hrirs, length = hrirs64, 64
    base = np.linspace(0.02, 0.10, 8)
    radii = np.sort(np.concatenate([base, base - 4e-4]))
    so = [simulation_order(4, 48000.0, r, raw=True) for r in radii]
    assert len(set(so)) == 8 and max(so) == 44
    shards = shard_jobs([(s + 1) ** 2 for s in so], 2)      # two "ranks", run one after the other on this GPU
    assert sorted(j for s in shards for j in s) == list(range(16))
    results = {}
    for mine in shards:
        for group in lane_groups([so[j] for j in mine]):
            plans = []
            for gi in group:
                r = float(radii[mine[gi]])
                p = Plan(L.KIND_EMAGLS2, "real", 4, 48000.0, length, 64, 2702, r, 32)
                p.set_hrir_grid(grids["azi"], grids["zen"])
                p.set_mic_grid(grids["mic_azi"], grids["mic_zen"])
                p.set_hrirs(hrirs[0], hrirs[1])
                plans.append(p)
            b = Batch(plans)
            assert b.lane_mode()      # radii of one simulation-order class differ in their routes by a bin: unified by the batch
            b.execute()
            b.execute()     # (second execute: hipGraph capture path)
            for gi, w in zip(group, b.get_filters()):
                results[mine[gi]] = w
            b.close()
            for p in plans:
                p.close()
    assert sorted(results) == list(range(16))
    worst = 0.0
    for j in (0, 5, 10, 15):     # one-shot designs of a spread of jobs (each also a different code path: plan of its own)
        wL, wR = E.getEMagLs2Filters(hrirs[0], hrirs[1], grids["azi"], grids["zen"], float(radii[j]), grids["mic_azi"], grids["mic_zen"],
                                     4, 48000.0, length, "real")
        worst = max(worst, rel(results[j][0], wL), rel(results[j][1], wR))
    print(f"radius sweep: lane batches vs one-shot designs, worst rel = {worst:.3e}")
    assert worst < 1e-9
    # and two of them against the oracle (smallest and largest radius)
    for j in (0, 15):
        oL, oR = O.getEMagLs2Filters(hrirs[0], hrirs[1], grids["azi"], grids["zen"], float(radii[j]), grids["mic_azi"], grids["mic_zen"],
                                     4, 48000.0, length, "real")
        assert rel(results[j][0], oL) < TOL and rel(results[j][1], oR) < TOL


def test_run_batch_composes_with_the_real_designs(grids, hrirs64):
    """emagls_amd.batch.run_batch (the multi-GPU job runner; its gather is covered on CPU with two gloo ranks) driving the real
    design function on this GPU: jobs come back in job order and equal direct calls."""
    import emagls_amd as E
    from emagls_amd.batch import run_batch, simulation_order
    radii = [0.06, 0.025, 0.042]

    def design(r):
        return E.getEMagLs2Filters(hrirs64[0], hrirs64[1], grids["azi"], grids["zen"], r, grids["mic_azi"], grids["mic_zen"], 4,
                                   48000.0, 64, "real")
    out = run_batch(radii, design, costs=[(simulation_order(4, 48000.0, r, raw=True) + 1) ** 2 for r in radii])
    assert len(out) == 3
    for (wL, wR), r in zip(out, radii):
        dL, dR = design(r)
        assert wL.shape == (64, 32) and np.array_equal(wL, dL) and np.array_equal(wR, dR)


@pytest.mark.parametrize("max_batch", [16])   # (8 -- four batches per rank, round 4's split -- ran until round 5: 32 s of the suite)
def test_one_ranks_share_of_config4_in_lane_mode(grids, hrirs64, max_batch):
    """BASELINE config 4 as named: 256 radii over 8 ranks.  One rank's full share -- 2 padded lane batches of about 16 designs (cut at
    equal cost since round 5: more designs per batch at low simulation orders, never more than 32) or 4 of about 8 designs of neighbouring simulation-order classes
    (emagls_amd.batch.padded_lane_batches / shard_lane_batches) -- through Batch: every batch runs in LANE mode, every job equals
    its own one-shot design (which is laid out for its own simulation order, no padding), and one job of every batch is compared
    with the oracle."""
    import ctypes
    import emagls_amd as E
    from emagls_amd import Batch, Plan, _lib as L
    from emagls_amd.batch import padded_lane_batches, shard_lane_batches, simulation_order
    hrirs, length = hrirs64, 64
    radii = np.linspace(0.02, 0.10, 256)
    so = [simulation_order(4, 48000.0, r, raw=True) for r in radii]
    per_rank, load = shard_lane_batches(padded_lane_batches(so, max_batch), 8)
    mine = per_rank[5]
    assert len(mine) == 32 // max_batch and all(1 <= len(idx) <= 32 for idx, _ in mine)
    nmine = sum(len(idx) for idx, _ in mine)
    prev = ctypes.c_int(0)
    L.check(L.load().emagls_set_batch_max(max(len(idx) for idx, _ in mine), ctypes.byref(prev)))
    results, padded = {}, 0
    for idx, pad in mine:
        plans = []
        for j in idx:
            p = Plan(L.KIND_EMAGLS2, "real", 4, 48000.0, length, 64, 2702, float(radii[j]), 32, sim_order_pad=pad)
            p.set_hrir_grid(grids["azi"], grids["zen"])
            p.set_mic_grid(grids["mic_azi"], grids["mic_zen"])
            p.set_hrirs(hrirs[0], hrirs[1])
            i = p.info()
            assert i.sim_order == pad and i.sim_order_own == so[j] and i.num_sh_sim == (pad + 1) ** 2
            padded += i.sim_order != i.sim_order_own
            plans.append(p)
        b = Batch(plans)
        assert b.lane_mode()
        b.execute()
        b.execute()     # (second execute: hipGraph capture path)
        b.execute()
        for j, w in zip(idx, b.get_filters()):
            results[j] = w
        b.close()
        for p in plans:
            p.close()
    L.check(L.load().emagls_set_batch_max(prev.value, None))
    assert len(results) == nmine and padded >= 4      # the share really mixes simulation-order classes
    worst = 0.0
    for idx, _ in mine:                             # first and last job of every batch against its one-shot design
        for j in (idx[0], idx[-1]):
            wL, wR = E.getEMagLs2Filters(hrirs[0], hrirs[1], grids["azi"], grids["zen"], float(radii[j]), grids["mic_azi"],
                                         grids["mic_zen"], 4, 48000.0, length, "real")
            worst = max(worst, rel(results[j][0], wL), rel(results[j][1], wR))
    print(f"config 4, one rank's share ({nmine} radii, {len(mine)} lane batches): padded lane batches vs one-shot designs, worst rel = {worst:.3e}")
    assert worst < 1e-8
    worst_o = 0.0
    for idx, pad in mine:                           # one padded job per batch against the oracle
        j = next((j for j in idx if so[j] < pad), idx[0])
        oL, oR = O.getEMagLs2Filters(hrirs[0], hrirs[1], grids["azi"], grids["zen"], float(radii[j]), grids["mic_azi"], grids["mic_zen"],
                                     4, 48000.0, length, "real")
        worst_o = max(worst_o, rel(results[j][0], oL), rel(results[j][1], oR))
    print(f"config 4, one rank's share: {len(mine)} jobs vs oracle, worst rel = {worst_o:.3e}")
    assert worst_o < TOL


def test_sim_order_pad_is_refused_where_it_cannot_apply(grids):
    from emagls_amd import Plan, _lib as L
    with pytest.raises(L.EmaglsError):
        Plan(L.KIND_EMA_SH, "real", 2, 48000.0, 64, 64, 2702, 0.05, 16, sim_order_pad=30)
    with pytest.raises(L.EmaglsError):
        Plan(L.KIND_EMAGLS, "real", 4, 48000.0, 64, 64, 2702, 0.042, 32, custom_basis=True, sim_order_pad=30)
    p = Plan(L.KIND_EMAGLS, "real", 4, 48000.0, 64, 64, 2702, 0.042, 32, sim_order_pad=5)   # below the design's own order: no effect
    assert p.info().sim_order == p.info().sim_order_own == 19
    p.close()


def test_job_lists_of_config4_and_config5(grids, hrirs64):
    """emagls_amd.batch.emagls2_radius_sweep / emagls_from_atf_subjects: the two job lists BASELINE.json names as one call each
    (single process here; the split over ranks and the gather are covered by the gloo tests)."""
    import emagls_amd as E
    from emagls_amd import synth
    from emagls_amd.batch import emagls2_radius_sweep, emagls_from_atf_subjects
    # 10 radii -> one lane batch of 10 laid out for the largest radius' simulation order (batches of 5 + 5 with max_batch = 8): a design
    # padded by seven orders may take the other sweep form than its one-shot twin -- the filters then agree to the 5e-9 ... 6e-8 the
    # two forms differ by, not to the 1e-12 of identical routes
    radii = [0.031, 0.0312, 0.0335, 0.047, 0.0471, 0.0472, 0.0473, 0.0474, 0.0475, 0.0476]
    out = emagls2_radius_sweep(hrirs64[0], hrirs64[1], grids["azi"], grids["zen"], radii, grids["mic_azi"], grids["mic_zen"], 4, 48000.0, 64)
    assert len(out) == len(radii)
    for j in (0, 2, 9):
        wL, wR = E.getEMagLs2Filters(hrirs64[0], hrirs64[1], grids["azi"], grids["zen"], radii[j], grids["mic_azi"], grids["mic_zen"], 4,
                                     48000.0, 64, "real")
        assert rel(out[j][0], wL) < 1e-7 and rel(out[j][1], wR) < 1e-7
    sub = slice(0, 2702, 3)
    azi, zen = grids["azi"][sub], grids["zen"][sub]
    subjects = [synth.rigid_sphere_hrirs(azi, zen, seed=5 + j, head_radius=0.08 + 0.004 * j) for j in range(3)]
    atf, aazi, azen = synth.glasses_atfs(natf=1024, nmics=6, taps=64)
    hg, ag = np.column_stack([azi, zen]), np.column_stack([aazi, azen])
    res = emagls_from_atf_subjects(subjects, hg, atf, ag, 48000.0, 128, 2000.0)
    assert len(res) == 3
    for j in range(3):
        wL, wR = E.getEMagLsFiltersFromAtf(subjects[j][0], subjects[j][1], hg, atf, ag, 48000.0, 128, 2000.0, verbose=False)
        assert rel(res[j][0], wL) < 1e-11 and rel(res[j][1], wR) < 1e-11
    # HRIR sets on one grid and one array (the loop over subjects around getEMagLsFilters): batches that share the geometry stages
    from emagls_amd.batch import emagls_hrir_sets
    subjects5 = [synth.rigid_sphere_hrirs(azi, zen, seed=50 + j, head_radius=0.08 + 0.003 * j) for j in range(5)]
    res = emagls_hrir_sets(subjects5, azi, zen, grids["mic_radius"], grids["mic_azi"], grids["mic_zen"], 4, 48000.0, 128, "complex", max_batch=3)
    assert len(res) == 5     # batches of 3 + 2
    for j in range(5):
        wL, wR = E.getEMagLsFilters(subjects5[j][0], subjects5[j][1], azi, zen, grids["mic_radius"], grids["mic_azi"], grids["mic_zen"], 4,
                                    48000.0, 128, "complex")
        assert res[j][0].dtype == wL.dtype and rel(res[j][0], wL) < 1e-12 and rel(res[j][1], wR) < 1e-12
    # the same loop around getMagLsFilters (BASELINE config 2's design): MagLS plans in batches, one sweep launch per batch
    from emagls_amd.batch import magls_hrir_sets
    res = magls_hrir_sets(subjects5, azi, zen, 4, 48000.0, 128, "real", max_batch=4)
    assert len(res) == 5     # batches of 4 + 1
    for j in range(5):
        wL, wR = E.getMagLsFilters(subjects5[j][0], subjects5[j][1], azi, zen, 4, 48000.0, 128, "real")
        assert res[j][0].dtype == wL.dtype and rel(res[j][0], wL) < 1e-12 and rel(res[j][1], wR) < 1e-12
