"""GPU parity of the render-side neighbours through the C ABI against the CPU oracle: the chain the
reference's harness runs between the microphone recording and the binaural decoder (verifyEMagLs.m:235-262), the 2-D MagLS
design and the two equalisation filters.  Floating point, tolerance 1e-6 relative (BASELINE.json north_star); the
elementwise ones are held to 1e-10."""
import numpy as np
import pytest

from oracle import emagls_oracle as O

pytestmark = pytest.mark.gpu
TOL = 1e-6


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


@pytest.mark.parametrize("kind,kw", [("tikhonov", {}), ("tikhonov", {"regulConst": 1e-3}), ("softlimit", {"noiseGainDb": 20.0}),
                                     ("full", {}), ("none", {})])
def test_get_radial_filter(kind, kw):
    import emagls_amd as E
    params = dict(order=4, fs=48000.0, smaRadius=0.042, arrayType="rigid", irLen=512, oversamplingFactor=2, radialFilter=kind, **kw)
    rad = E.getRadialFilter(params)
    ref = O.getRadialFilter(4, 48000.0, 0.042, irLen=512, oversamplingFactor=2, radialFilter=kind, **kw)
    assert rad.shape == (513, 5)
    nan_ref = np.isnan(ref)
    assert np.array_equal(np.isnan(rad), nan_ref)          # 0/0 and 1/0 at DC for the orders above 0, as the reference computes them
    if kind in ("softlimit", "full"):
        assert nan_ref[0, 1:].all() and not nan_ref[1:].any()
    assert rel(rad[~nan_ref], np.asarray(ref, dtype=complex)[~nan_ref]) < 1e-10
    assert np.all(rad[-1].imag == 0) and np.all(rad[-1].real >= 0)     # Nyquist bin := abs (getRadialFilter.m:68-70)


def test_get_radial_filter_defaults_and_errors():
    import emagls_amd as E
    rad = E.getRadialFilter(order=2, fs=44100.0, smaRadius=0.05, arrayType="rigid")       # irLen 256, oversampling 2, tikhonov 1e-2
    assert rel(rad, O.getRadialFilter(2, 44100.0, 0.05)) < 1e-10 and rad.shape == (257, 3)
    with pytest.raises(ValueError, match="Unkown radialFilter"):
        E.getRadialFilter(order=2, fs=44100.0, smaRadius=0.05, arrayType="rigid", radialFilter="wiener")
    with pytest.raises(NotImplementedError, match="not yet implemented"):
        E.getRadialFilter(order=2, fs=44100.0, smaRadius=0.05, arrayType="rigid", waveModel="pointSource")
    with pytest.raises(KeyError):
        E.getRadialFilter(order=2, fs=44100.0, arrayType="rigid")
    odd = E.getRadialFilter(order=1, fs=48000.0, smaRadius=0.05, arrayType="rigid", irLen=9, oversamplingFactor=1)
    assert odd.shape == (5, 2)       # nfft = 9: linspace(0, fs/2, nfft/2+1) has 5.5 -> 5 points in MATLAB; no Nyquist rule


@pytest.mark.parametrize("nsamp,kind,kw", [(6000, "tikhonov", {}), (300, "tikhonov", {}), (5000, "softlimit", {"noiseGainDb": 15.0})])
def test_apply_radial_filter(nsamp, kind, kw):
    """The harness configuration (verifyEMagLs.m:239-250: irLen = filterLen = 512, oversamplingFactor 1) on seeded noise,
    a signal shorter than nfft (zero padded, :20-22) and the softlimit filter whose DC NaNs are zeroed (:14)."""
    import emagls_amd as E
    rng = np.random.default_rng(7)
    sig = rng.standard_normal((nsamp, 25))
    params = dict(order=4, fs=48000.0, smaRadius=0.042, arrayType="rigid", irLen=512, oversamplingFactor=1, nfft=512,
                  radialFilter=kind, **kw)
    out = E.applyRadialFilter(sig, params)
    ref = O.applyRadialFilter(sig, 4, 48000.0, 0.042, 512, 1, radialFilter=kind, **kw)
    assert out.shape == ref.shape == (max(nsamp, 512) - 256, 25)
    assert rel(out, ref) < 1e-10
    with pytest.raises(ValueError, match="params.nfft"):
        E.applyRadialFilter(sig, dict(params, nfft=1024))


@pytest.mark.parametrize("basis", ["real", "complex"])
def test_encode_sh(grids, basis):
    import emagls_amd as E
    rng = np.random.default_rng(11)
    rec = rng.standard_normal((4000, grids["mic_azi"].size))
    out = E.encodeSH(rec, grids["mic_azi"], grids["mic_zen"], 4, basis)
    ref = O.encodeSH(rec, grids["mic_azi"], grids["mic_zen"], 4, basis)
    assert out.shape == (4000, 25) and out.dtype == ref.dtype
    assert rel(out, ref) < 1e-11
    from emagls_amd._lib import EmaglsError
    with pytest.raises(EmaglsError, match="fewer microphones"):
        E.encodeSH(rec[:, :16], grids["mic_azi"][:16], grids["mic_zen"][:16], 4, basis)


def test_harness_render_chain(grids, thin_hrirs):
    """verifyEMagLs.m:228-262 end to end on the GPU: eMagLS filters, SH encoding, radial filters, binaural decoding --
    against the same chain on the oracle."""
    import emagls_amd as E
    hL, hR, azi, zen = thin_hrirs
    rng = np.random.default_rng(3)
    rec = rng.standard_normal((3000, grids["mic_azi"].size))
    args = (hL, hR, azi, zen, grids["mic_radius"], grids["mic_azi"], grids["mic_zen"], 4, 48000.0, 128, "real")
    wL, wR = E.getEMagLsFilters(*args)
    oL, oR = O.getEMagLsFilters(*args)
    sh = E.encodeSH(rec, grids["mic_azi"], grids["mic_zen"], 4)
    osh = O.encodeSH(rec, grids["mic_azi"], grids["mic_zen"], 4)
    out = E.binauralDecode(sh, 48000.0, wL, wR, 48000.0)
    ref = O.binauralDecode(osh, oL, oR)
    assert rel(out, ref) < TOL
    # the MagLS branch of the harness: radial filters in front of the decoder
    p = dict(order=4, fs=48000.0, smaRadius=grids["mic_radius"], arrayType="rigid", irLen=128, oversamplingFactor=1, nfft=128)
    shf = E.applyRadialFilter(sh, p)
    oshf = O.applyRadialFilter(osh, 4, 48000.0, grids["mic_radius"], 128, 1)
    assert rel(shf, oshf) < 1e-9


@pytest.fixture(scope="module")
def thin_hrirs(grids, hrirs):
    sub = slice(0, 2702, 3)
    return hrirs[0][:, sub], hrirs[1][:, sub], grids["azi"][sub], grids["zen"][sub]


@pytest.fixture(scope="module")
def horizontal_hrirs():
    from emagls_amd import synth
    azi = np.linspace(0.0, 2 * np.pi, 180, endpoint=False)
    hL, hR = synth.rigid_sphere_hrirs(azi, np.full(azi.size, np.pi / 2))
    return hL, hR, azi


@pytest.mark.parametrize("basis,order,length", [("real", 4, 256), ("complex", 4, 256), ("real", 7, 512), ("complex", 15, 128),
                                                ("real", 20, 128), ("complex", 24, 128),   # (orders above 15: more than 32 channels, wide.hip)
                                                ("real", 40, 128), ("complex", 60, 128)])  # (above 31: more than 64 channels, its loop forms)
def test_magls_filters_2d(horizontal_hrirs, basis, order, length):
    import emagls_amd as E
    hL, hR, azi = horizontal_hrirs
    wL, wR = E.getMagLsFilters2D(hL, hR, azi, order, 48000.0, length, basis)
    oL, oR = O.getMagLsFilters2D(hL, hR, azi, order, 48000.0, length, basis)
    assert wL.shape == (length, 2 * order + 1) and wL.dtype == oL.dtype
    assert rel(wL, oL) < TOL and rel(wR, oR) < TOL
    if basis == "complex":       # w_{-m} = conj(w_m): what getChFreqDomainConjugate is there for
        for m in range(1, order + 1):
            assert rel(wL[:, 2 * m - 1], np.conj(wL[:, 2 * m])) < 1e-9


def test_magls_filters_2d_errors(horizontal_hrirs):
    import emagls_amd as E
    from emagls_amd._lib import EmaglsError
    hL, hR, azi = horizontal_hrirs
    with pytest.raises(EmaglsError, match="HRIR len too short"):       # getMagLsFilters2D.m:38
        E.getMagLsFilters2D(hL, hR, azi, 4, 48000.0, 64)
    with pytest.raises(EmaglsError, match="order above 127"):
        E.getMagLsFilters2D(hL, hR, azi, 128, 48000.0, 256)


@pytest.mark.parametrize("radius,order,fs,length", [(0.042, 4, 48000.0, 512), (0.0875, 3, 44100.0, 256), (0.042, 1, 48000.0, 2048)])
def test_spherical_head_filter(radius, order, fs, length):
    import emagls_amd as E
    w, W = E.getMagLsSphericalHeadFilter(radius, order, fs, length)
    ow, oW = O.getMagLsSphericalHeadFilter(radius, order, fs, length)
    assert w.shape == (length, 1) and W.shape == (min(2048, 2 * length), 1)
    assert rel(w[:, 0], ow) < 1e-10 and rel(W[:, 0], oW) < 1e-12


def test_spherical_head_filter_order_above_simulation_order():
    import emagls_amd as E
    from emagls_amd._lib import EmaglsError
    with pytest.raises(EmaglsError, match="index error"):      # bn_Hi(:, 1:order+1) with ceil(fs*pi*r/c) = 2 < order
        E.getMagLsSphericalHeadFilter(0.004, 4, 48000.0, 256)


@pytest.mark.parametrize("basis", ["real", "complex"])
def test_array_diffuse_filter(grids, basis):
    import emagls_amd as E
    args = (grids["mic_radius"], grids["mic_azi"], grids["mic_zen"], 4, 48000.0, 512, basis)
    w = E.getMagLsArrayDiffuseFilter(*args)
    ow = O.getMagLsArrayDiffuseFilter(*args)
    assert w.shape == (512, 1)
    assert rel(w[:, 0], ow) < 1e-9
    # a custom shFunction crosses the boundary as a matrix at the simulation order
    calls = []

    def sh(n, dirs, definition):
        calls.append(n)
        return O.getSH(n, dirs, definition)
    w2 = E.getMagLsArrayDiffuseFilter(*args, shFunction=sh)
    assert calls == [int(np.ceil(48000.0 * np.pi * grids["mic_radius"] / 343.0))]
    assert rel(w2[:, 0], ow) < 1e-9


# --------------------------------------------------------------------------------------------
# diffuseness (covariance) constraint, SURVEY 8(f) rank 1
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fn,basis", [("getMagLsFilters", "real"), ("getMagLsFilters", "complex"), ("getEMagLsFilters", "real"),
                                      ("getEMagLsFilters", "complex"), ("getEMagLs2Filters", "real")])
def test_diffuseness_constraint(grids, thin_hrirs, fn, basis):
    """The three designs with applyDiffusenessConst against the oracle's specification, and the property itself: the filters
    differ from the unconstrained ones (by a few per cent), and per bin by a 2x2 Hermitian ear mixing."""
    import emagls_amd as E
    hL, hR, azi, zen = thin_hrirs
    if fn == "getMagLsFilters":
        args = (hL, hR, azi, zen, 4, 48000.0, 128, basis)
    else:
        args = (hL, hR, azi, zen, grids["mic_radius"], grids["mic_azi"], grids["mic_zen"], 4, 48000.0, 128, basis)
    wL, wR = getattr(E, fn)(*args, applyDiffusenessConst=True)
    oL, oR = getattr(O, fn)(*args, applyDiffusenessConst=True)
    uL, uR = getattr(E, fn)(*args)
    assert wL.dtype == oL.dtype and wL.shape == oL.shape
    assert rel(wL, oL) < TOL and rel(wR, oR) < TOL, (rel(wL, oL), rel(wR, oR))
    assert 1e-3 < rel(wL, uL) < 0.5
    # per-bin structure, as the reference's fixture pairs show it (tests/test_oracle_kats.py::test_diffuseness_*)
    F = lambda w: np.fft.fft(np.vstack([w, np.zeros_like(w)]), axis=0)
    Wl, Wr, Dl, Dr = F(uL), F(uR), F(wL), F(wR)
    for k in (12, 30, 60, 100):
        M, r = O.fit_ear_mixing(Wl[k], Wr[k], Dl[k], Dr[k])
        assert r < 5e-2 and np.abs(M - M.conj().T).max() < 5e-2, (k, r, M)


def test_diffuseness_constraint_errors(grids, thin_hrirs):
    from emagls_amd import Plan, _lib as L
    from emagls_amd._lib import EmaglsError
    with pytest.raises(EmaglsError, match="diffuseness constraint applies"):
        Plan(L.KIND_LS, "real", 4, 48000.0, 128, 128, 900, diffuseness=True)


# --------------------------------------------------------------------------------------------
# helper functions of the reference a caller may use on their own
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("basis", ["real", "complex"])
def test_get_ch(basis):
    import emagls_amd as E
    azi = np.linspace(-3.0, 7.0, 333)
    Y = E.getCH(6, azi, basis)
    Yo = O.getCH(6, azi, basis)
    assert Y.shape == (333, 13) and Y.dtype == Yo.dtype and rel(Y, Yo) < 1e-14


@pytest.mark.parametrize("basis,raw", [("real", False), ("complex", False), ("real", True), ("complex", True)])
def test_get_smair_matrix(grids, basis, raw):
    """The array model the filter designs only use in factored form, materialised (getSMAIRMatrix.m:110-127): against the
    oracle's restatement, including the real(b_n) rule of the last bin."""
    import emagls_amd as E
    params = dict(order=4, fs=48000.0, irLen=256, oversamplingFactor=1, smaRadius=grids["mic_radius"], radialFilter="none",
                  smaDesignAziZenRad=np.column_stack([grids["mic_azi"], grids["mic_zen"]]), shDefinition=basis, returnRawMicSigs=raw)
    sm, p = E.getSMAIRMatrix(params)
    so, simOrder = O.getSMAIRMatrix(4, 48000.0, 256, grids["mic_radius"], params["smaDesignAziZenRad"], basis, returnRawMicSigs=raw)
    assert p["simulationOrder"] == simOrder == 19 and sm.shape == so.shape == (32 if raw else 25, 400, 129)
    assert rel(sm, so) < 1e-11
    assert np.all(sm[:, :, -1].imag == 0) or basis == "complex"


def test_get_smair_matrix_with_radial_filters(grids):
    """radialFilter other than 'none' (getSMAIRMatrix.m:129-138) against the oracle's restatement of those lines: rows of the
    SH-domain model scaled by the radial filter of their order -- and, literally like the reference, scaled TWICE at the
    Nyquist bin (:134 applies BnTi, :136 applies real(BnTi) to the already filtered slice)."""
    import emagls_amd as E
    grid = np.column_stack([grids["mic_azi"], grids["mic_zen"]])
    for kind, kw in (("tikhonov", {}), ("softlimit", {"noiseGainDb": 15.0}), ("full", {})):
        sm1, p1 = E.getSMAIRMatrix(order=3, fs=48000.0, irLen=128, oversamplingFactor=1, smaRadius=0.042, smaDesignAziZenRad=grid,
                                   radialFilter=kind, **kw)
        so, _ = O.getSMAIRMatrix(3, 48000.0, 128, 0.042, grid, "real", radialFilter=kind, noiseGainDb=kw.get("noiseGainDb", 20.0))
        ok = np.isfinite(so)      # ('softlimit' / 'full' are 0/0 or 1/0 at DC for the orders above 0, there like here)
        assert sm1.shape == so.shape and np.array_equal(np.isfinite(sm1), ok) and rel(sm1[ok], so[ok]) < 1e-11, kind
    sm0, _ = E.getSMAIRMatrix(order=3, fs=48000.0, irLen=128, oversamplingFactor=1, smaRadius=0.042, smaDesignAziZenRad=grid,
                               radialFilter="none")
    sm1, _ = E.getSMAIRMatrix(order=3, fs=48000.0, irLen=128, oversamplingFactor=1, smaRadius=0.042, smaDesignAziZenRad=grid,
                               radialFilter="tikhonov")
    rad = O.getRadialFilter(3, 48000.0, 0.042, irLen=128, oversamplingFactor=1)          # [P x order+1]
    n_of_c = np.repeat(np.arange(4), 2 * np.arange(4) + 1)
    scale = rad[:, n_of_c].T[:, None, :].copy()
    scale[:, :, -1] = scale[:, :, -1] ** 2            # the Nyquist quirk, spelled out
    assert rel(sm1, sm0 * scale) < 1e-12


def test_get_smair_matrix_defaults_are_the_references(grids):
    """dependencies/getSMAIRMatrix.m:36-84: order 4, fs 48 kHz, r = 4.2 cm, radialFilter 'regul', noiseGainDb 20,
    oversamplingFactor 4, irLen 2048.  'regul' is not a filter getRadialFilter.m knows (:63-64 -> error), so a call that leaves
    radialFilter unset fails exactly like the reference's unless the raw microphone signals are requested."""
    import emagls_amd as E
    grid = np.column_stack([grids["mic_azi"], grids["mic_zen"]])
    with pytest.raises(ValueError, match='Unkown radialFilter parameter "regul"'):
        E.getSMAIRMatrix(smaDesignAziZenRad=grid, irLen=32)
    sm, p = E.getSMAIRMatrix(smaDesignAziZenRad=grid, irLen=32, returnRawMicSigs=True)
    assert p["oversamplingFactor"] == 4 and p["order"] == 4 and p["fs"] == 48000 and p["smaRadius"] == 0.042 and p["noiseGainDb"] == 20
    assert sm.shape == (32, 400, 4 * 32 // 2 + 1)
    so, _ = O.getSMAIRMatrix(4, 48000.0, 32, 0.042, grid, "real", returnRawMicSigs=True, oversamplingFactor=4)
    assert rel(sm, so) < 1e-11
    with pytest.raises(KeyError):      # the reference would load its t-design file here; the mirror asks for the grid
        E.getSMAIRMatrix(irLen=32, returnRawMicSigs=True)


def test_render_side_with_fft_lengths_that_are_not_powers_of_two():
    """The equalisation filters (nfft = min(2048, 2*len)) and the radial-filter chain (nfft = oversamplingFactor * irLen) at
    lengths such as 120 -> 240 and 100 x 2 -> 200."""
    import emagls_amd as E
    w, Wf = E.getMagLsSphericalHeadFilter(0.0875, 3, 44100.0, 120)
    ow, oW = O.getMagLsSphericalHeadFilter(0.0875, 3, 44100.0, 120)
    assert w.shape == (120, 1) and Wf.shape == (240, 1) and rel(w[:, 0], ow) < 1e-9 and rel(Wf[:, 0], oW) < 1e-9
    rng = np.random.default_rng(2)
    sig = rng.standard_normal((900, 9))
    params = dict(order=2, fs=48000.0, smaRadius=0.042, arrayType="rigid", irLen=100, oversamplingFactor=2, nfft=200, radialFilter="tikhonov")
    out = E.applyRadialFilter(sig, params)
    ref = O.applyRadialFilter(sig, 2, 48000.0, 0.042, 100, 2, radialFilter="tikhonov")
    assert out.shape == ref.shape and rel(out, ref) < 1e-9
