"""GPU parity through the C ABI against the CPU oracle, tolerance 1e-6 relative complex error (BASELINE.json north_star), with the
reference's own assertAllClose metrics (verifyEMagLs.m:370-395): getEMagLsFilters / getEMagLs2Filters (lib/getEMagLsFilters.m:32-142, lib/getEMagLs2Filters.m:32-135): BASELINE config 3 at full size, config 4 shapes, routes of the per-bin factorisation, custom shFunction, error behaviour.
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


@pytest.mark.parametrize("basis,length", [("real", 128), ("complex", 256)])
def test_emagls_filters_thin(grids, thin, basis, length):
    import emagls_amd as E
    args = (thin["hL"], thin["hR"], thin["azi"], thin["zen"], grids["mic_radius"], grids["mic_azi"], grids["mic_zen"], 4,
            48000.0, length, basis)
    wL, wR = E.getEMagLsFilters(*args)
    oL, oR = O.getEMagLsFilters(*args)
    assert wL.dtype == oL.dtype and wL.shape == (length, 25)
    assert report("eMagLS L " + basis, wL, oL) < TOL and report("eMagLS R " + basis, wR, oR) < TOL


@pytest.mark.parametrize("fn", ["getEMagLsFilters", "getEMagLs2Filters"])
def test_complex_basis_pipelines_agree(grids, thin, monkeypatch, fn):
    """A complex-basis design is served by the real-arithmetic pipeline and a unitary channel transform (W_c = W_r T_N;
    eMagLS2 is basis free).  The complex-arithmetic pipeline (EMAGLS_REAL_INTERNAL=0) must give the same filters, and both
    must match the oracle's complex-basis computation."""
    import emagls_amd as E
    args = (thin["hL"], thin["hR"], thin["azi"], thin["zen"], grids["mic_radius"], grids["mic_azi"], grids["mic_zen"], 4,
            48000.0, 128, "complex")
    rL, rR = getattr(E, fn)(*args)
    monkeypatch.setenv("EMAGLS_REAL_INTERNAL", "0")
    cL, cR = getattr(E, fn)(*args)
    monkeypatch.delenv("EMAGLS_REAL_INTERNAL")
    oL, oR = getattr(O, fn)(*args)
    assert rL.dtype == np.complex128 and cL.dtype == np.complex128
    assert report(fn + " real-internal vs complex pipeline", rL, cL) < 1e-9 and rel(rR, cR) < 1e-9
    assert report(fn + " complex pipeline vs oracle", cL, oL) < TOL and rel(cR, oR) < TOL
    assert report(fn + " real-internal vs oracle", rL, oL) < TOL and rel(rR, oR) < TOL


@pytest.mark.parametrize("basis", ["real", "complex"])
def test_emagls2_filters_thin(grids, thin, basis):
    import emagls_amd as E
    args = (thin["hL"], thin["hR"], thin["azi"], thin["zen"], grids["mic_radius"], grids["mic_azi"], grids["mic_zen"], 4,
            48000.0, 256, basis)
    wL, wR = E.getEMagLs2Filters(*args)
    oL, oR = O.getEMagLs2Filters(*args)
    assert wL.shape == (256, 32)
    assert report("eMagLS2 L " + basis, wL, oL) < TOL and report("eMagLS2 R " + basis, wR, oR) < TOL


@pytest.mark.parametrize("basis", ["real", "complex"])
def test_emagls_tiny_array_ill_conditioned_bins(grids, thin, basis):
    """A 7 mm array: kr stays below 0.4 up to 3 kHz, the high orders vanish and cond(pwGrid) is far above 1e4 in the
    first swept bins.  Those bins cannot use Y_reg_inv = conj(G) conj(M); they take the accurate S-space form
    conj(Q) Z_k (real basis: Q materialised; complex basis: conj(Yc) (Z_k R^-H))."""
    import emagls_amd as A
    from emagls_amd import Plan, _lib as L
    length = 128
    wL, wR = A.getEMagLsFilters(thin["hL"], thin["hR"], thin["azi"], thin["zen"], 0.007, grids["mic_azi"], grids["mic_zen"], 4,
                                48000.0, length, basis)
    oL, oR = O.getEMagLsFilters(thin["hL"], thin["hR"], thin["azi"], thin["zen"], 0.007, grids["mic_azi"], grids["mic_zen"], 4,
                                48000.0, length, basis)
    assert rel(wL, oL) < TOL and rel(wR, oR) < TOL, (rel(wL, oL), rel(wR, oR))
    p = Plan(L.KIND_EMAGLS, basis, 4, 48000.0, length, thin["hL"].shape[0], thin["hL"].shape[1], 0.007, 32)
    p.set_hrir_grid(thin["azi"], thin["zen"])
    p.set_mic_grid(grids["mic_azi"], grids["mic_zen"])
    p.set_hrirs(thin["hL"], thin["hR"])
    p.execute()
    p.synchronize()
    ok = p.debug("cond_ok", np.float64)
    k0 = p.info().k_cut - 1
    assert (ok[k0:] == 0).sum() >= 3, "the test must exercise the ill-conditioned path"
    p.close()


def test_gram_route_fallback(grids, thin, monkeypatch):
    """The well-conditioned swept bins are factorised from the Gram matrix B^H B; which bins qualify is estimated on the
    host from kr.  The Jacobi kernel verifies the estimate and requests a re-run on the Householder route when a bin is
    worse conditioned.  A forced, far too optimistic estimate on a 7 mm array must still give the oracle's filters, for a
    single plan and for a batch."""
    import emagls_amd as A
    from emagls_amd import Batch, Plan, _lib as L
    monkeypatch.setenv("EMAGLS_GRAM_COND_EST", "1e30")
    args = (thin["hL"], thin["hR"], thin["azi"], thin["zen"], 0.007, grids["mic_azi"], grids["mic_zen"], 4, 48000.0, 128, "complex")
    oL, oR = O.getEMagLsFilters(*args)
    wL, wR = A.getEMagLsFilters(*args)
    assert rel(wL, oL) < TOL and rel(wR, oR) < TOL, (rel(wL, oL), rel(wR, oR))
    plans = []
    for j in range(2):
        p = Plan(L.KIND_EMAGLS, "complex", 4, 48000.0, 128, thin["hL"].shape[0], thin["hL"].shape[1], 0.007, 32)
        p.set_hrir_grid(thin["azi"], thin["zen"])
        p.set_mic_grid(grids["mic_azi"], grids["mic_zen"])
        p.set_hrirs(thin["hL"], thin["hR"])
        plans.append(p)
    b = Batch(plans)
    for it in range(2):
        b.execute()
        for bL, bR in b.get_filters():
            assert rel(bL, oL) < TOL and rel(bR, oR) < TOL, it
    b.close()
    for p in plans:
        p.close()


@pytest.mark.parametrize("order", [1, 2, 3])
def test_emagls_low_orders(grids, thin, order):
    """Orders below 4 (4, 9, 16 channels): the persistent sweep loads all 32 slab rows of a bin whatever the channel
    count, so the last bin reads up to 28 rows of padding behind G (regression: the padding once was 8 rows)."""
    import emagls_amd as E
    args = (thin["hL"], thin["hR"], thin["azi"], thin["zen"], grids["mic_radius"], grids["mic_azi"], grids["mic_zen"],
            order, 48000.0, 128, "complex" if order == 2 else "real")
    wL, wR = E.getEMagLsFilters(*args)
    oL, oR = O.getEMagLsFilters(*args)
    assert wL.shape == (128, (order + 1) ** 2)
    assert report(f"eMagLS N={order} L", wL, oL) < TOL and report(f"eMagLS N={order} R", wR, oR) < TOL


def test_emagls_filters_config3_full(grids, hrirs):
    """BASELINE config 3: em32 r = 4.2 cm, N = 4, complex SH, 2702 directions, 512 taps."""
    import emagls_amd as E
    args = (hrirs[0], hrirs[1], grids["azi"], grids["zen"], grids["mic_radius"], grids["mic_azi"], grids["mic_zen"], 4,
            48000.0, 512, "complex")
    wL, wR = E.getEMagLsFilters(*args)
    oL, oR = O.getEMagLsFilters(*args)
    assert report("eMagLS config3 L", wL, oL) < TOL and report("eMagLS config3 R", wR, oR) < TOL


def test_emagls2_filters_config4_shape(grids, hrirs):
    """BASELINE config 4, one job of the radius batch: raw 32-mic em32, 2702 directions, 1024 taps (nfft 2048,
    1024 solved bins, k_cut 86), default real basis.  The oracle needs 110 s for it: its output is a stored vector
    (tests/golden/oracle_vectors.npz, written by tests/golden/make_oracle_vectors.py from the same seeded inputs)."""
    import os
    import emagls_amd as E
    vec = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_vectors.npz"))
    args = (hrirs[0], hrirs[1], grids["azi"], grids["zen"], 0.05, grids["mic_azi"], grids["mic_zen"], 4, 48000.0, 1024, "real")
    wL, wR = E.getEMagLs2Filters(*args)
    oL, oR = vec["config4_r50mm_len1024/wL"], vec["config4_r50mm_len1024/wR"]
    assert wL.shape == (1024, 32) and wL.dtype == np.float64
    assert report("eMagLS2 config4 L", wL, oL) < TOL and report("eMagLS2 config4 R", wR, oR) < TOL


@pytest.mark.parametrize("order,radius,sim_order", [(1, 0.005, 4), (6, 0.005, 4), (1, 0.01, 5), (6, 0.01, 5)])
def test_emagls2_simulation_order_rule(grids, thin, order, radius, sim_order):
    """eMagLS2 simulates at max(4, ceil(fs*pi*r/343)) whatever `order` is (lib/getEMagLs2Filters.m:51-63 leaves params.order
    unset -> dependencies/getSMAIRMatrix.m:39-41); `order` only moves f_cut.  Plan constants and filters against the oracle."""
    import emagls_amd as E
    from emagls_amd import Plan, _lib as L
    p = Plan(L.KIND_EMAGLS2, "real", order, 48000.0, 128, thin["hL"].shape[0], thin["hL"].shape[1], radius, 12)
    info = p.info()
    p.close()
    assert info.sim_order == sim_order and info.num_sh_sim == (sim_order + 1) ** 2
    assert info.k_cut == int(np.ceil(max(1e3, 500 * order) / (24000.0 / 128)))
    args = (thin["hL"], thin["hR"], thin["azi"], thin["zen"], radius, grids["mic_azi"][:12], grids["mic_zen"][:12], order, 48000.0, 128, "real")
    wL, wR = E.getEMagLs2Filters(*args)
    oL, oR = O.getEMagLs2Filters(*args)
    assert report(f"eMagLS2 order {order} r {radius} L", wL, oL) < TOL and report("R", wR, oR) < TOL


def _sn3d_sh(N, dirs, basisType="real"):
    """A custom shFunction as a user of the reference would pass it (lib/getEMagLsFilters.m:32): SN3D-weighted harmonics."""
    Y = O.getSH(N, dirs, basisType)
    w = np.concatenate([np.full(2 * n + 1, 1.0 / np.sqrt(2 * n + 1)) for n in range(N + 1)])
    return Y * w[None, :]


@pytest.mark.parametrize("basis", ["real", "complex"])
def test_custom_sh_function(grids, thin, basis):
    """shFunction handles (lib/getLsFilters.m:27, getMagLsFilters.m:30, getEMagLsFilters.m:32, getEMagLs2Filters.m:32) are
    evaluated on the host side and travel as matrices (emagls_*_with_basis).  Passing the default function that way must give
    the built-in result; a genuinely different basis (SN3D) must give what the oracle computes with the same function."""
    import emagls_amd as E
    hL, hR, azi, zen = thin["hL"], thin["hR"], thin["azi"], thin["zen"]
    mic = (grids["mic_radius"], grids["mic_azi"], grids["mic_zen"])
    for fn, args in (("getLsFilters", (hL, hR, azi, zen, 3)), ("getMagLsFilters", (hL, hR, azi, zen, 3, 48000.0, 128)),
                     ("getEMagLsFilters", (hL, hR, azi, zen) + mic + (3, 48000.0, 128)),
                     ("getEMagLs2Filters", (hL, hR, azi, zen) + mic + (3, 48000.0, 128))):
        bL, bR = getattr(E, fn)(*args, basis)
        cL, cR = getattr(E, fn)(*args, basis, O.getSH)
        # (the built-in array designs take the synthesising sweep, a caller's matrices the materialised operands: the same filters
        # to what the bins' conditioning makes of operands that agree to 1e-15)
        tol_b = 1e-6 if "EMagLs" in fn else 1e-9
        assert cL.dtype == bL.dtype and rel(cL, bL) < tol_b and rel(cR, bR) < tol_b, (fn, rel(cL, bL), rel(cR, bR))
        sL, sR = getattr(E, fn)(*args, basis, _sn3d_sh)
        oL, oR = getattr(O, fn)(*args, basis, shFunction=_sn3d_sh)
        assert report(fn + " SN3D shFunction " + basis, sL, oL) < TOL and rel(sR, oR) < TOL
        if fn != "getEMagLs2Filters":   # (raw-microphone filters do not depend on the basis scaling)
            assert rel(sL, bL) > 1e-3


def test_one_shot_plan_cache(grids, thin):
    """The one-shot entry points reuse the plan of the previous call of the same shape (buffers, captured graphs): results
    must follow the inputs, not the cache, and emagls_cache_clear() must leave the library usable."""
    import emagls_amd as E
    from emagls_amd import _lib as L
    args = lambda h: (h[0], h[1], thin["azi"], thin["zen"], grids["mic_radius"], grids["mic_azi"], grids["mic_zen"], 4, 48000.0, 128, "complex")
    h1 = (thin["hL"], thin["hR"])
    h2 = (thin["hR"][::-1].copy() * 0.5, thin["hL"].copy())
    a1 = E.getEMagLsFilters(*args(h1))
    a2 = E.getEMagLsFilters(*args(h2))          # same shape: served by the cached plan (second execute: graph capture)
    a3 = E.getEMagLsFilters(*args(h1))          # third: graph replay
    a4 = E.getEMagLsFilters(*args(h2))
    assert rel(a3[0], a1[0]) < 1e-12 and rel(a3[1], a1[1]) < 1e-12 and rel(a4[0], a2[0]) < 1e-12
    assert rel(a2[0], a1[0]) > 1e-2
    o2 = O.getEMagLsFilters(*args(h2))
    assert rel(a4[0], o2[0]) < TOL and rel(a4[1], o2[1]) < TOL
    # a different microphone grid under the same shape key
    g2 = (h1[0], h1[1], thin["azi"], thin["zen"], grids["mic_radius"], grids["mic_azi"] + 0.3, grids["mic_zen"], 4, 48000.0, 128, "complex")
    b1 = E.getEMagLsFilters(*g2)
    ob = O.getEMagLsFilters(*g2)
    assert rel(b1[0], ob[0]) < TOL and rel(b1[0], a1[0]) > 1e-3
    L.check(L.load().emagls_cache_clear())
    a5 = E.getEMagLsFilters(*args(h1))
    assert rel(a5[0], a1[0]) < 1e-12


def test_error_behaviour(grids, hrirs):
    """assert(len >= size(hL,1), 'len too short') (lib/getEMagLsFilters.m:42) and friends."""
    import emagls_amd as E
    from emagls_amd._lib import EmaglsError
    with pytest.raises(EmaglsError, match="len too short"):
        E.getEMagLsFilters(hrirs[0], hrirs[1], grids["azi"], grids["zen"], 0.042, grids["mic_azi"], grids["mic_zen"], 4,
                           48000.0, 64)
    with pytest.raises(EmaglsError, match="HRIR len too short"):
        E.getMagLsFilters(hrirs[0], hrirs[1], grids["azi"], grids["zen"], 4, 48000.0, 64)
    with pytest.raises(ValueError):
        E.getLsFilters(hrirs[0], hrirs[1], grids["azi"][:10], grids["zen"], 4)
    # len > nfft = min(2048, 2*len): the reference fails with an index error (lib/getEMagLsFilters.m:135); here: a clean status
    with pytest.raises(EmaglsError, match="index error"):
        E.getEMagLsFilters(hrirs[0], hrirs[1], grids["azi"], grids["zen"], 0.042, grids["mic_azi"], grids["mic_zen"], 4, 48000.0, 4096)
    with pytest.raises(EmaglsError, match="index error"):
        E.getMagLsFilters(hrirs[0], hrirs[1], grids["azi"], grids["zen"], 4, 48000.0, 4096)


@pytest.mark.parametrize("fn,radius", [("getEMagLsFilters", 0.12), ("getEMagLs2Filters", 0.142), ("getEMagLs2Filters", 0.193)])
def test_simulation_orders_above_47(grids, fn, radius):
    """dependencies/getSMAIRMatrix.m:95 takes any array radius; until round 5 the build stopped at simulation order 47 (10.9 cm at
    48 kHz) -- a table size, and one kernel (the Chebyshev conversion of the series, one thread per order in a single wave) that
    was silently wrong from 65 orders on.  Orders 53, 63 and 85 (12 cm, 14.2 cm, 19.3 cm: the em32's layout on a larger sphere) on a
    1500-point grid against the oracle; above 85 the call is refused: the reference's own getSH overflows there (170!)."""
    import emagls_amd as E
    from emagls_amd import synth
    azi, zen = synth.fibonacci_grid(1500)
    hL, hR = synth.rigid_sphere_hrirs(azi, zen, taps=64)
    args = (hL, hR, azi, zen, radius, grids["mic_azi"], grids["mic_zen"], 4, 48000.0, 96, "real")
    wL, wR = getattr(E, fn)(*args)
    # (the oracle needs 20-30 s for each: stored vectors, tests/golden/make_oracle_vectors.py, same seeded inputs)
    vec = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_vectors.npz"))
    key = "order53_emagls" if fn == "getEMagLsFilters" else ("order63_emagls2" if radius < 0.15 else "order85_emagls2")
    oL, oR = vec[key + "/wL"], vec[key + "/wR"]
    assert report(f"{fn} r = {100 * radius:.1f} cm L", wL, oL) < TOL and report("R", wR, oR) < TOL


@pytest.mark.parametrize("length", [100, 150, 300])
def test_filter_lengths_whose_fft_length_is_not_a_power_of_two(grids, thin, length):
    """nfft = min(2048, 2*len) for any even len (lib/getEMagLsFilters.m:44): lengths such as 100, 150, 300 give nfft = 200, 300, 600.
    Those run on direct-DFT kernels (prologue, ATF spectra, epilogue) instead of the LDS FFTs; every design against the oracle."""
    import emagls_amd as E
    from emagls_amd import synth
    hL, hR = thin["hL"][:64], thin["hR"][:64]
    a = (hL, hR, thin["azi"], thin["zen"])
    wL, wR = E.getMagLsFilters(*a, 4, 48000.0, length, "real")
    oL, oR = O.getMagLsFilters(*a, 4, 48000.0, length, "real")
    assert wL.shape == (length, 25)
    assert report(f"MagLS len {length} L", wL, oL) < TOL and report("R", wR, oR) < TOL
    for fn, basis in (("getEMagLsFilters", "complex"), ("getEMagLs2Filters", "real")):
        args = a + (0.042, grids["mic_azi"], grids["mic_zen"], 4, 48000.0, length, basis)
        wL, wR = getattr(E, fn)(*args)
        oL, oR = getattr(O, fn)(*args)
        assert report(f"{fn} {basis} len {length} L", wL, oL) < TOL and report("R", wR, oR) < TOL
    atf, aazi, azen = synth.glasses_atfs(natf=1024, nmics=6, taps=48)
    hg, ag = np.column_stack([thin["azi"], thin["zen"]]), np.column_stack([aazi, azen])
    wL, wR = E.getEMagLsFiltersFromAtf(hL, hR, hg, atf, ag, 48000.0, length, 2000.0, verbose=False)
    oL, oR, dev = O.getEMagLsFiltersFromAtf(hL, hR, hg, atf, ag, 48000.0, length, 2000.0)
    assert report(f"FromAtf len {length} L", wL, oL) < TOL and report("R", wR, oR) < TOL


def test_designs_on_large_hrir_grids():
    """lib/*.m take any number of HRIR directions.  Above 3072 the resident sweep does not hold a design on one XCD and the
    launch-per-bin sweeps take over, their workgroups walking several 64-direction slabs (dense_sweep_nwg) so that the next
    launch can still stage every partial sum; FromAtf above 4096 matched directions stays on the Gram route.  MagLS, eMagLS and
    FromAtf on a 5000-point grid against the oracle."""
    import emagls_amd as E
    from emagls_amd import synth
    D = 5000
    azi, zen = synth.fibonacci_grid(D)
    hL, hR = synth.rigid_sphere_hrirs(azi, zen, taps=64)
    maz, mzn = synth.em32_grid()
    wL, wR = E.getMagLsFilters(hL, hR, azi, zen, 4, 48000.0, 128)
    oL, oR = O.getMagLsFilters(hL, hR, azi, zen, 4, 48000.0, 128)
    assert report("MagLS, 5000 directions L", wL, oL) < TOL and report("R", wR, oR) < TOL
    args = (hL, hR, azi, zen, 0.042, maz, mzn, 4, 48000.0, 128)
    wL, wR = E.getEMagLsFilters(*args)
    oL, oR = O.getEMagLsFilters(*args)
    assert report("eMagLS, 5000 directions L", wL, oL) < TOL and report("R", wR, oR) < TOL
    atf, aazi, azen = synth.glasses_atfs(natf=5300, nmics=8, taps=64)
    hg, ag = np.column_stack([azi, zen]), np.column_stack([aazi, azen])
    wL, wR = E.getEMagLsFiltersFromAtf(hL, hR, hg, atf, ag, 48000.0, 128, 2000.0, verbose=False)
    oL, oR, _ = O.getEMagLsFiltersFromAtf(hL, hR, hg, atf, ag, 48000.0, 128, 2000.0)
    assert report("FromAtf, 5000 matched directions L", wL, oL) < TOL and report("R", wR, oR) < TOL
