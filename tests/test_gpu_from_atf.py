"""GPU parity through the C ABI against the CPU oracle, tolerance 1e-6 relative complex error (BASELINE.json north_star), with the
reference's own assertAllClose metrics (verifyEMagLs.m:370-395): getEMagLsFiltersFromAtf (lib/getEMagLsFiltersFromAtf.m:29-151): BASELINE config 5, grid match, ATF-side sharing, routes.
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


def _match_idx(hg, ag, nmics=2, taps=16):
    """match_idx / match_dev of a FROM_ATF plan (the grid matching runs in the first stage of the design)."""
    from emagls_amd import Plan, _lib as L
    rng = np.random.default_rng(3)
    D, Da = hg.shape[0], ag.shape[0]
    p = Plan(L.KIND_FROM_ATF, "real", 0, 48000.0, 32, nsamp=taps, ndirs=D, nmics=nmics, f_trans=2000.0, atf_taps=taps, natf=Da)
    p.set_hrir_grid(hg[:, 0], hg[:, 1])
    p.set_hrirs(rng.standard_normal((taps, D)), rng.standard_normal((taps, D)))
    p.set_atfs(rng.standard_normal((taps, nmics, Da)), ag[:, 0], ag[:, 1])
    p.execute()
    p.synchronize()
    n = min(D, Da)
    idx = p.debug("match_idx", np.int64)[:n].copy()
    dev = p.debug("match_dev", np.float64)[:n].copy()
    mean = p.info().mean_grid_dev_deg
    p.close()
    return idx, dev, mean


def test_match_idx_bit_exact(grids):
    """Index work must be bit-exact: the nearest-neighbour indices of lib/getEMagLsFiltersFromAtf.m:81-95 on config 5's grids
    (2702 HRIR directions against the 16 384-point ATF lattice), on the reverse case (ATF grid smaller) and on constructed
    exact ties (duplicate ATF directions, mirror-image pairs: MATLAB's min returns the first index, :84)."""
    from emagls_amd import synth
    hg = np.column_stack([grids["azi"], grids["zen"]])
    aazi, azen = synth.fibonacci_grid(16384)
    ag = np.column_stack([aazi, azen])
    smaller, oidx, odev = O.matchGrids(hg, ag)
    idx, dev, mean = _match_idx(hg, ag)
    assert smaller and np.array_equal(idx, oidx)
    assert np.abs(dev - odev).max() < 1e-6 and abs(mean - odev.mean()) < 1e-9   # acos near 1 amplifies the last-bit differences of cos/sin
    # ATF grid smaller: it picks from the HRIR grid
    sazi, szen = synth.fibonacci_grid(700)
    sg = np.column_stack([sazi + 0.01, szen])
    smaller, oidx, odev = O.matchGrids(hg, sg)
    idx, dev, mean = _match_idx(hg, sg)
    assert not smaller and np.array_equal(idx, oidx)
    # equal sizes: the HRIR grid is the "smaller" one (min([a b]) returns the first index, :62)
    smaller, oidx, _ = O.matchGrids(sg, sg[::-1].copy())
    idx, _, _ = _match_idx(sg, sg[::-1].copy())
    assert smaller and np.array_equal(idx, oidx) and np.array_equal(idx, np.arange(700)[::-1])
    # exact ties: every ATF direction appears three times (positions j, j + n, j + 2n) -> the first copy wins;
    # and mirror pairs about azimuth 0 at the equator: (+a) listed before (-a) -> index of (+a)
    n = 257
    bazi, bzen = synth.fibonacci_grid(n)
    tg = np.column_stack([np.tile(bazi, 3), np.tile(bzen, 3)])
    q = np.column_stack([bazi + 1e-3, bzen])[:64]
    smaller, oidx, _ = O.matchGrids(q, tg)
    idx, _, _ = _match_idx(q, tg)
    assert np.array_equal(idx, oidx) and idx.max() < n
    a = np.linspace(0.05, 1.0, 40)
    mg = np.column_stack([np.concatenate([a, -a]), np.full(80, np.pi / 2)])
    qh = np.column_stack([np.zeros(3), np.full(3, np.pi / 2)])
    smaller, oidx, _ = O.matchGrids(qh, mg)
    idx, _, _ = _match_idx(qh, mg)
    assert oidx.tolist() == [0, 0, 0] and np.array_equal(idx, oidx)


def test_from_atf_config5_shape(grids, hrirs):
    """BASELINE config 5 shape: ATF grid of 16 384 directions x 8 microphones, 2048 taps, fTrans 2 kHz, the full
    2702-direction HRIR grid.  Checked against the oracle on the same inputs."""
    import emagls_amd as E
    from emagls_amd import synth
    atf, aazi, azen = synth.glasses_atfs(natf=16384, nmics=8, taps=256)
    hg = np.column_stack([grids["azi"], grids["zen"]])
    ag = np.column_stack([aazi, azen])
    wL, wR = E.getEMagLsFiltersFromAtf(hrirs[0], hrirs[1], hg, atf, ag, 48000.0, 2048, 2000.0, verbose=False)
    # (the oracle needs 30 s for it: its output is a stored vector, tests/golden/make_oracle_vectors.py, same seeded inputs)
    vec = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_vectors.npz"))
    oL, oR = vec["config5_full/wL"], vec["config5_full/wR"]
    assert wL.shape == (2048, 8)
    assert report("FromAtf config5 L", wL, oL) < TOL and report("FromAtf config5 R", wR, oR) < TOL


def test_from_atf_small(thin):
    import emagls_amd as E
    from emagls_amd import synth
    atf, aazi, azen = synth.glasses_atfs(natf=2048, nmics=8, taps=128)
    hg = np.column_stack([thin["azi"], thin["zen"]])
    ag = np.column_stack([aazi, azen])
    wL, wR = E.getEMagLsFiltersFromAtf(thin["hL"], thin["hR"], hg, atf, ag, 48000.0, 256, 2000.0)
    oL, oR, dev = O.getEMagLsFiltersFromAtf(thin["hL"], thin["hR"], hg, atf, ag, 48000.0, 256, 2000.0)
    assert wL.shape == (256, 8)
    assert report("FromAtf L", wL, oL) < TOL and report("FromAtf R", wR, oR) < TOL


@pytest.mark.parametrize("nmics", [40, 64])
def test_from_atf_above_32_microphones(thin, nmics):
    """lib/getEMagLsFiltersFromAtf.m:40 takes any microphone count.  33..64 microphones: the matched ATF matrix of every bin is
    factored by the plain per-bin kernels of wide_array.hip (Householder QR + one-sided Jacobi, Y_reg_inv_k written out), one sweep
    launch per bin (round 3 refused more than 32)."""
    import emagls_amd as E
    from emagls_amd import synth
    atf, aazi, azen = synth.glasses_atfs(natf=1500, nmics=nmics, taps=64)
    hg = np.column_stack([thin["azi"], thin["zen"]])
    ag = np.column_stack([aazi, azen])
    wL, wR = E.getEMagLsFiltersFromAtf(thin["hL"], thin["hR"], hg, atf, ag, 48000.0, 128, 2000.0, verbose=False)
    oL, oR, dev = O.getEMagLsFiltersFromAtf(thin["hL"], thin["hR"], hg, atf, ag, 48000.0, 128, 2000.0)
    assert wL.shape == (128, nmics)
    assert report(f"FromAtf {nmics} microphones L", wL, oL) < TOL and report("R", wR, oR) < TOL


def test_from_atf_subject_list_above_32_microphones(thin):
    """emagls_amd.batch.emagls_from_atf_subjects with a 40-microphone ATF set: the library does not batch such designs, the job list
    runs them plan by plan and returns what single calls return."""
    import emagls_amd as E
    from emagls_amd import synth
    from emagls_amd.batch import emagls_from_atf_subjects
    atf, aazi, azen = synth.glasses_atfs(natf=1200, nmics=40, taps=64)
    hg = np.column_stack([thin["azi"], thin["zen"]])
    ag = np.column_stack([aazi, azen])
    subjects = [(thin["hL"], thin["hR"]), (thin["hR"], thin["hL"])]
    res = emagls_from_atf_subjects(subjects, hg, atf, ag, 48000.0, 128, 2000.0)
    for (hL, hR), (wL, wR) in zip(subjects, res):
        sL, sR = E.getEMagLsFiltersFromAtf(hL, hR, hg, atf, ag, 48000.0, 128, 2000.0, verbose=False)
        assert rel(wL, sL) < 1e-12 and rel(wR, sR) < 1e-12


@pytest.mark.parametrize("natf", [1024, 500])
def test_from_atf_512_taps_wave_prologue(thin, monkeypatch, natf):
    """512-tap FromAtf filters: nfft = 1024, so the HRIR prologue with the integer circshift (lib/getEMagLsFiltersFromAtf.m:43-53)
    runs on the wave-private transforms -- on all HRIR directions (ATF grid the larger one) and on the gathered ones (ATF grid the
    smaller one, FromAtf.m:71-79); same design with EMAGLS_HRIR_FFT_WAVE=0 on the LDS form."""
    import emagls_amd as E
    from emagls_amd import synth
    atf, aazi, azen = synth.glasses_atfs(natf=natf, nmics=8, taps=128)
    hg = np.column_stack([thin["azi"], thin["zen"]])
    ag = np.column_stack([aazi, azen])
    oL, oR, dev = O.getEMagLsFiltersFromAtf(thin["hL"], thin["hR"], hg, atf, ag, 48000.0, 512, 2000.0)
    wL, wR = E.getEMagLsFiltersFromAtf(thin["hL"], thin["hR"], hg, atf, ag, 48000.0, 512, 2000.0, verbose=False)
    assert report("FromAtf 512 taps (wave prologue) L", wL, oL) < TOL and report("R", wR, oR) < TOL
    monkeypatch.setenv("EMAGLS_HRIR_FFT_WAVE", "0")
    vL, vR = E.getEMagLsFiltersFromAtf(thin["hL"], thin["hR"], hg, atf, ag, 48000.0, 512, 2000.0, verbose=False)
    assert report("FromAtf 512 taps (LDS prologue) L", vL, oL) < TOL and rel(vL, wL) < 1e-9 and rel(vR, wR) < 1e-9


def test_from_atf_atf_grid_smaller(thin):
    """ATF grid smaller than the HRIR grid: the HRTFs are gathered instead (FromAtf.m:71-79,91-93)."""
    import emagls_amd as E
    from emagls_amd import synth
    atf, aazi, azen = synth.glasses_atfs(natf=512, nmics=4, taps=64)
    hg = np.column_stack([thin["azi"], thin["zen"]])
    ag = np.column_stack([aazi, azen])
    wL, wR = E.getEMagLsFiltersFromAtf(thin["hL"], thin["hR"], hg, atf, ag, 48000.0, 128, 1500.0, verbose=False)
    oL, oR, dev = O.getEMagLsFiltersFromAtf(thin["hL"], thin["hR"], hg, atf, ag, 48000.0, 128, 1500.0)
    assert report("FromAtf(small ATF grid) L", wL, oL) < TOL and report("FromAtf(small ATF grid) R", wR, oR) < TOL


def _atf_plan(thin, hL, hR, atf, aazi, azen, length=256, f_trans=2000.0):
    from emagls_amd import Plan, _lib as L
    p = Plan(L.KIND_FROM_ATF, "real", 0, 48000.0, length, hL.shape[0], hL.shape[1], nmics=atf.shape[1], f_trans=f_trans,
             atf_taps=atf.shape[0], natf=atf.shape[2])
    p.set_hrir_grid(thin["azi"], thin["zen"])
    p.set_hrirs(hL, hR)
    p.set_atfs(atf, aazi, azen)
    return p


def test_from_atf_runs_on_the_persistent_sweep(thin):
    """One resident sweep launch instead of one launch per bin (938 at config 5), the per-bin factors from the M x M Gram
    matrices of the matched ATF spectra; EMAGLS_SWEEP_PERSIST=0 keeps the launch-per-bin form, same filters."""
    from emagls_amd import synth
    atf, aazi, azen = synth.glasses_atfs(natf=2048, nmics=8, taps=128)
    p = _atf_plan(thin, thin["hL"], thin["hR"], atf, aazi, azen)
    outs = []
    for _ in range(3):   # eager, captured, replayed
        p.execute()
        outs.append(p.get_filters())
    i = p.info()
    p.close()
    assert i.num_sweep_launches == 1 and i.gram_from == 1
    for wL, wR in outs[1:]:
        assert np.array_equal(wL, outs[0][0]) and np.array_equal(wR, outs[0][1])
    hg, ag = np.column_stack([thin["azi"], thin["zen"]]), np.column_stack([aazi, azen])
    oL, oR, dev = O.getEMagLsFiltersFromAtf(thin["hL"], thin["hR"], hg, atf, ag, 48000.0, 256, 2000.0)
    assert report("FromAtf persistent L", outs[0][0], oL) < TOL and report("R", outs[0][1], oR) < TOL


def test_from_atf_batch_of_subjects_shares_the_atf_side(thin):
    """BASELINE config 5's batch: HRTF subjects of ONE ATF set.  The batch computes the ATF side (spectra of the matched ATFs,
    per-bin factors) once and sweeps all subjects in one resident launch; every subject equals its single design and the
    oracle.  A batch whose plans hold different ATF sets is detected (device-side comparison) and runs unshared."""
    from emagls_amd import Batch, synth
    atf, aazi, azen = synth.glasses_atfs(natf=2048, nmics=8, taps=128)
    subjects = [synth.rigid_sphere_hrirs(thin["azi"], thin["zen"], seed=40 + j, head_radius=0.075 + 0.005 * j) for j in range(4)]
    singles = []
    for hL, hR in subjects:
        q = _atf_plan(thin, hL, hR, atf, aazi, azen)
        q.execute()
        singles.append(q.get_filters())
        q.close()
    plans = [_atf_plan(thin, hL, hR, atf, aazi, azen) for hL, hR in subjects]
    b = Batch(plans)
    first = None
    for it in range(3):
        b.execute()
        res = b.get_filters()
        assert b.shares_atf_side()
        for (wL, wR), (sL, sR) in zip(res, singles):
            assert rel(wL, sL) < 1e-12 and rel(wR, sR) < 1e-12, it
        if first is None:
            first = res
        else:
            for (wL, wR), (fL, fR) in zip(res, first):
                assert np.array_equal(wL, fL) and np.array_equal(wR, fR), it
    assert plans[0].info().num_sweep_launches == 1
    hg, ag = np.column_stack([thin["azi"], thin["zen"]]), np.column_stack([aazi, azen])
    oL, oR, dev = O.getEMagLsFiltersFromAtf(subjects[3][0], subjects[3][1], hg, atf, ag, 48000.0, 256, 2000.0)
    assert report("FromAtf batch, subject 3 L", res[3][0], oL) < TOL and report("R", res[3][1], oR) < TOL
    # one subject gets another ATF set: no sharing any more, results still per plan
    atf2 = atf * 1.0
    atf2[:, 3, :] *= 0.5
    plans[2].set_atfs(atf2, aazi, azen)
    b.execute()
    res2 = b.get_filters()
    assert not b.shares_atf_side()
    q = _atf_plan(thin, subjects[2][0], subjects[2][1], atf2, aazi, azen)
    q.execute()
    sL, sR = q.get_filters()
    q.close()
    assert rel(res2[2][0], sL) < 1e-12 and rel(res2[2][1], sR) < 1e-12
    assert rel(res2[1][0], singles[1][0]) < 1e-12 and rel(res2[0][1], singles[0][1]) < 1e-12
    b.close()
    for p in plans:
        p.close()


def test_from_atf_sixteen_microphones(thin):
    """More than 8 ATF microphones (lib/getEMagLsFiltersFromAtf.m:40 takes any count): the Gram route carries up to 32."""
    import emagls_amd as E
    from emagls_amd import synth
    atf, aazi, azen = synth.glasses_atfs(natf=2048, nmics=16, taps=128)
    hg, ag = np.column_stack([thin["azi"], thin["zen"]]), np.column_stack([aazi, azen])
    wL, wR = E.getEMagLsFiltersFromAtf(thin["hL"], thin["hR"], hg, atf, ag, 48000.0, 256, 2000.0, verbose=False)
    oL, oR, dev = O.getEMagLsFiltersFromAtf(thin["hL"], thin["hR"], hg, atf, ag, 48000.0, 256, 2000.0)
    assert wL.shape == (256, 16)
    assert report("FromAtf 16 mics L", wL, oL) < TOL and report("R", wR, oR) < TOL


def test_from_atf_ill_conditioned_atfs_take_the_dense_route(thin):
    """Two nearly identical microphones: cond(atfsMatched(k,:,:)) ~ 1e5 at every bin, beyond what the Gram route is accurate for.
    Its device-side check raises the status flag, the route's start moves behind the offending bins and the design is re-run on
    the dense route (Householder QR + Jacobi SVD of the matched ATF matrix itself): still the oracle's filters."""
    from emagls_amd import synth
    atf, aazi, azen = synth.glasses_atfs(natf=2048, nmics=6, taps=128, noise=0.0)
    rng = np.random.default_rng(3)
    atf[:, 5, :] = atf[:, 4, :] + 1e-5 * rng.standard_normal(atf[:, 4, :].shape)
    p = _atf_plan(thin, thin["hL"], thin["hR"], atf, aazi, azen)
    p.execute()
    wL, wR = p.get_filters()
    i = p.info()
    p.close()
    assert i.gram_from != 1            # the route moved (0: every bin on the dense route)
    hg, ag = np.column_stack([thin["azi"], thin["zen"]]), np.column_stack([aazi, azen])
    oL, oR, dev = O.getEMagLsFiltersFromAtf(thin["hL"], thin["hR"], hg, atf, ag, 48000.0, 256, 2000.0)
    assert report("FromAtf ill-conditioned L", wL, oL) < TOL and report("R", wR, oR) < TOL


def test_from_atf_subjects_in_one_call(thin):
    """emagls_from_atf_hrir_sets: the HRTF subjects of one ATF set as ONE call (BASELINE config 5's job list): the ATF set goes to
    the GPU once, its side is computed once per batch; 5 subjects equal their single calls, twice (the second call reuses the
    cached plans)."""
    import emagls_amd as E
    from emagls_amd import synth
    rng = np.random.default_rng(61)
    azi, zen = thin["azi"], thin["zen"]
    hL = np.stack([thin["hL"] * (1 + 0.04 * j) + 1e-3 * rng.standard_normal(thin["hL"].shape) for j in range(5)], axis=2)
    hR = np.stack([thin["hR"] * (1 - 0.03 * j) for j in range(5)], axis=2)
    atf, aazi, azen = synth.glasses_atfs(natf=700, nmics=6, taps=64)
    hg, ag = np.column_stack([azi, zen]), np.column_stack([aazi + 0.01, azen])
    for rep in range(2):
        wL, wR, dev = E.fromAtfHrirSets(hL, hR, hg, atf, ag, 48000.0, 128, 2000.0)
        assert wL.shape == (128, 6, 5)
        worst = 0.0
        for j in (0, 2, 4):
            sL, sR = E.getEMagLsFiltersFromAtf(hL[:, :, j], hR[:, :, j], hg, atf, ag, 48000.0, 128, 2000.0, verbose=False)
            worst = max(worst, rel(wL[:, :, j], sL), rel(wR[:, :, j], sR))
        assert 0.0 < dev < 20.0       # mean grid deviation in degrees (lib/getEMagLsFiltersFromAtf.m:96)
        print(f"5 FromAtf subjects in one call (pass {rep}): worst rel vs single calls = {worst:.3e}")
        assert worst < 1e-11
    assert rel(wL[:, :, 0], wL[:, :, 3]) > 1e-3
