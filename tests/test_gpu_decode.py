"""GPU parity through the C ABI against the CPU oracle, tolerance 1e-6 relative complex error (BASELINE.json north_star), with the
reference's own assertAllClose metrics (verifyEMagLs.m:370-395): binauralDecode (dependencies/binauralDecode.m:33-64).
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


def test_binaural_decode(golden):
    import emagls_amd as E
    rng = np.random.default_rng(5)
    sig = rng.standard_normal((20000, 25))
    wL = golden["real_eMagLS_woDC/wEMlsL"]
    wR = golden["real_eMagLS_woDC/wEMlsR"]
    out = E.binauralDecode(sig, 48000, wL, wR, 48000)
    ref = O.binauralDecode(sig, wL, wR)
    assert out.shape == (20000, 2) and rel(out, ref) < 1e-12
    out2 = E.binauralDecode(sig, 48000, wL, wR, 48000, True)
    ref2 = O.binauralDecode(sig, wL, wR, True)
    assert out2.shape == ref2.shape and rel(out2, ref2) < 1e-12
    # linearity (size-independent property)
    a = E.binauralDecode(2.5 * sig, 48000, wL, wR, 48000)
    assert rel(a, 2.5 * out) < 1e-13


@pytest.mark.parametrize("nsamp,nch,length", [(5000, 25, 512), (777, 4, 64), (100, 9, 256), (3000, 1, 2), (4097, 7, 1000),
                                              (9000, 36, 2048), (2600, 16, 3000), (2000, 3, 257), (600000, 5, 400), (700000, 2, 512),
                                              (1100000, 3, 512)])
def test_binaural_decode_shapes(nsamp, nch, length, monkeypatch):
    """The fused overlap-save kernels (257..512 taps: wave-private 1024-point transforms, one wave per block on long signals
    and eight waves per block on short ones; up to 256 taps, or with EMAGLS_DECODE_WAVE=0: half-wave two-factor transforms;
    up to 2048 taps: LDS transform passes; segments, spectra and products never leave the CU) on ragged shapes --
    odd channel counts (the last transform carries one channel), signals shorter than a block, one channel, two taps, a length
    that is not a power of two -- against the oracle's time-domain sum, and against the hipFFT passes (EMAGLS_DECODE_FUSED=0),
    which also serve the filters above 2048 taps."""
    import emagls_amd as E
    rng = np.random.default_rng(nsamp + nch)
    sig = rng.standard_normal((nsamp, nch))
    wL = rng.standard_normal((length, nch)) * np.exp(-np.arange(length) / (0.3 * length))[:, None]
    wR = rng.standard_normal((length, nch)) * np.exp(-np.arange(length) / (0.3 * length))[:, None]
    out = E.binauralDecode(sig, 48000, wL, wR, 48000)
    ref = O.binauralDecode(sig, wL, wR)
    assert out.shape == ref.shape == (nsamp, 2)
    monkeypatch.setenv("EMAGLS_DECODE_FILTER_FFT", "hipfft")   # the wave form's filter tables from hipFFT spectra instead of its own transform
    assert rel(E.binauralDecode(sig, 48000, wL, wR, 48000), ref) < 1e-12
    monkeypatch.delenv("EMAGLS_DECODE_FILTER_FFT")
    monkeypatch.setenv("EMAGLS_DECODE_WAVE", "0")        # the half-wave two-factor form (what up to 256 taps take anyway)
    half = E.binauralDecode(sig, 48000, wL, wR, 48000)
    monkeypatch.setenv("EMAGLS_DECODE_REGFFT", "0")      # the fused kernel on LDS transform passes (what 513..2048 taps take anyway)
    lds = E.binauralDecode(sig, 48000, wL, wR, 48000)
    monkeypatch.setenv("EMAGLS_DECODE_FUSED", "0")
    plain = E.binauralDecode(sig, 48000, wL, wR, 48000)
    print(f"decode {nsamp} x {nch}, {length} taps: fused vs oracle rel = {rel(out, ref):.3e}, half-wave form {rel(half, ref):.3e}, "
          f"LDS-pass form {rel(lds, ref):.3e}, hipFFT passes {rel(plain, ref):.3e}")
    assert rel(out, ref) < 1e-12 and rel(half, ref) < 1e-12 and rel(lds, ref) < 1e-12 and rel(plain, ref) < 1e-12


def test_binaural_decode_complex(golden):
    """Complex-SH rendering (dependencies/binauralDecode.m:39-42,59-64): complex filters (the reference's own complex eMagLS
    fixture) on a complex-SH signal; the output is the real part of the accumulated products, the discarded imaginary part is
    reported like the reference's warning does."""
    import warnings
    import emagls_amd as E
    rng = np.random.default_rng(11)
    wL = golden["complex_eMagLS_woDC/wEMlsL"]
    wR = golden["complex_eMagLS_woDC/wEMlsR"]
    assert np.iscomplexobj(wL) and wL.shape == (512, 25)
    sig = rng.standard_normal((9000, 25)) + 1j * rng.standard_normal((9000, 25))
    ref = O.binauralDecode(sig, wL, wR)
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        out = E.binauralDecode(sig, 48000, wL, wR, 48000)
    assert out.shape == (9000, 2) and out.dtype == np.float64 and rel(out, ref) < 1e-12
    assert any("discarding imaginary part" in str(w.message) for w in wlist)
    with warnings.catch_warnings(record=True) as wlist2:
        warnings.simplefilter("always")
        out2 = E.binauralDecode(sig, 48000, wL, wR, 48000, True)
    assert rel(out2, O.binauralDecode(sig, wL, wR, True)) < 1e-12
    # the numbers in the warning are the reference's: sum(abs(imag(binauralOut))) AFTER the compensateDelay cut (:53-62)
    full = [sum(O.fftfilt(w[:, c], sig[:, c]) for c in range(25)) for w in (wL, wR)]
    for msgs, skip in ((wlist, 0), (wlist2, 512 // 2 - 1)):
        m_ = [str(w.message) for w in msgs if "discarding imaginary part" in str(w.message)][0]
        assert m_ == "discarding imaginary part with sum of [%.2g, %.2g] in rendering result." % tuple(np.abs(f.imag[skip:]).sum() for f in full)
    # a real signal through complex filters
    sr = rng.standard_normal((4000, 25))
    assert rel(E.binauralDecode(sr, 48000, wL, wR, 48000), O.binauralDecode(sr, wL, wR)) < 1e-12
    # A complex-SH encoded REAL sound field through filters with the symmetry w_{n,-m} = (-1)^m conj(w_{n,m}) renders without an
    # imaginary part.  The reference's complex MagLS fixture has that symmetry (test_oracle_kats); its complex eMagLS fixture
    # does not (DC := real(bin 2) per complex coefficient, lib/getEMagLsFilters.m:110-111) -- which is why the reference warns.
    mL, mR = golden["complex_MagLS_woDC/wMlsL"], golden["complex_MagLS_woDC/wMlsR"]
    N = 4
    T = np.zeros((25, 25), complex)   # Y_c = Y_r T  (tests/test_oracle_kats.py::real_to_complex_T)
    for n in range(N + 1):
        T[n * n + n, n * n + n] = 1
        for m in range(1, n + 1):
            a, b = n * n + n + m, n * n + n - m
            T[a, a] = (-1) ** m / np.sqrt(2); T[b, a] = 1j * (-1) ** m / np.sqrt(2)
            T[a, b] = 1 / np.sqrt(2); T[b, b] = -1j / np.sqrt(2)
    sc = sr @ np.conj(T)              # complex-SH coefficients of the real field with real-SH coefficients sr
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        oc = E.binauralDecode(sc, 48000, mL, mR, 48000)
    full = sum(O.fftfilt(mL[:, c], sc[:, c]) for c in range(25))
    assert np.abs(full.imag).max() < 1e-12 * np.abs(full.real).max() and rel(oc[:, 0], full.real) < 1e-12
    msgs = [str(w.message) for w in wlist if "discarding imaginary part" in str(w.message)]
    assert all(float(x) < 1e-9 for m_ in msgs for x in m_.split("[")[1].split("]")[0].split(","))
