"""CPU oracle: literal FP64 NumPy restatement of the eMagLS reference hot path.

THIS IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import it; the product (emagls_amd/) never does and fails loudly without its HIP library.

Every function cites the reference file:line it restates (paths relative to /root/reference).
The reference is MATLAB and cannot run in this container or on the GPU box (no matlab/octave), so
the oracle is pinned against the reference's own shipped golden filter sets
(resources/HRIR_L2702_512samples_32channels_sh4_*.mat -> tests/golden/ref_fixtures.npz) through
the fixture-internal known-answer tests in tests/test_oracle_kats.py.  Their input
(resources/HRIR_L2702.mat, Zenodo 3928297) is not shipped, therefore END-TO-END PARITY TO 1e-6 IS
UNPINNED: the KATs pin the SH convention, ACN order, Condon-Shortley phase, the real<->complex
relation, getShFreqDomainConjugate's semantics, window end points, delay bookkeeping and (to
discrimination level, ~5 %) the rigid-sphere modal coefficients -- not the tap values.

Third-party arithmetic that is NOT under /root/reference (empty git submodules, .gitmodules:1-9,
pinned SHAs unrecoverable) is restated from the published algorithms:
  * getSH                     polarch/Spherical-Harmonic-Transform  (orthonormal SH, ACN, [azi zen])
  * sphModalCoeffs            polarch/Array-Response-Simulator      (b_n(kr), rigid sphere)
  * getShFreqDomainConjugate  thomasdeppisch/sh-symmetries
and MathWorks built-ins (pinv, svd, grpdelay, hann, fftfilt, median, angle) from their documented
behaviour.  Array layout follows MATLAB: hL is [numSamples x numDirections], filters [len x C].
"""
from __future__ import annotations

import math

import numpy as np
import scipy.special as sps

C_SOUND = 343.0  # dependencies/getSMAIRMatrix.m:86
NFFT_MAX_LEN = 2048  # lib/getEMagLsFilters.m:35
F_CUT_MIN_FREQ = 1e3  # lib/getEMagLsFilters.m:36
SVD_REGUL_CONST = 0.01  # lib/getEMagLsFilters.m:39
SMAIR_DEFAULT_ORDER = 4  # dependencies/getSMAIRMatrix.m:39-41 (params.order when the caller leaves it unset)


# --------------------------------------------------------------------------------------------
# third-party restatements
# --------------------------------------------------------------------------------------------
def getSH(N, dirs, basisType="real"):
    """Orthonormal spherical harmonics, rows = directions, columns = ACN (n^2+n+m).

    Restates polarch getSH (call sites lib/getLsFilters.m:30, lib/getEMagLsFilters.m:68,
    dependencies/getSMAIRMatrix.m:101).  dirs = [azimuth, zenith] in radians.  Convention is the
    one stated in-tree by dependencies/getNnm.m:20-28 + dependencies/getCH.m:22-27:
      complex: Y_n^m = N_n^m P_n^m(cos zen) e^{i m azi}, Condon-Shortley inside P (MATLAB legendre),
               Y_n^{-m} = (-1)^m conj(Y_n^m);
      real:    Condon-Shortley cancelled; m<0 -> sqrt2 N P^{|m|} sin(|m| azi), m>0 -> sqrt2 N P^m cos(m azi).
    """
    dirs = np.asarray(dirs, dtype=np.float64)
    azi = dirs[:, 0]
    zen = dirs[:, 1]
    D = dirs.shape[0]
    x = np.cos(zen)
    cplx = basisType == "complex"
    if not cplx and basisType != "real":
        raise ValueError("basisType must be 'real' or 'complex'")
    Y = np.zeros((D, (N + 1) ** 2), dtype=np.complex128 if cplx else np.float64)
    for n in range(N + 1):
        m = np.arange(0, n + 1)
        # MATLAB legendre(n, x): unnormalised P_n^m with Condon-Shortley phase == scipy lpmv
        Lnm = sps.lpmv(m[:, None], n, x[None, :])  # (n+1) x D
        lg = sps.gammaln(n - m + 1) - sps.gammaln(n + m + 1)
        norm = np.sqrt((2 * n + 1) / (4 * np.pi) * np.exp(lg))  # sqrt((2n+1)(n-m)!/(4pi (n+m)!))
        if cplx:
            Ypos = (norm[:, None] * Lnm) * np.exp(1j * m[:, None] * azi[None, :])
            for mm in range(0, n + 1):
                Y[:, n * n + n + mm] = Ypos[mm]
                if mm > 0:
                    Y[:, n * n + n - mm] = (-1) ** mm * np.conj(Ypos[mm])
        else:
            for mm in range(0, n + 1):
                base = norm[mm] * ((-1) ** mm) * Lnm[mm]  # cancel Condon-Shortley
                if mm == 0:
                    Y[:, n * n + n] = base
                else:
                    Y[:, n * n + n + mm] = base * (math.sqrt(2.0) * np.cos(mm * azi))
                    Y[:, n * n + n - mm] = base * (math.sqrt(2.0) * np.sin(mm * azi))
    return Y


def _sph_besselj(n, x):
    return sps.spherical_jn(n, x)


def _dsph_besselj(n, x):
    return sps.spherical_jn(n, x, derivative=True)


def _sph_hankel2(n, x):
    return sps.spherical_jn(n, x) - 1j * sps.spherical_yn(n, x)


def _dsph_hankel2(n, x):
    return sps.spherical_jn(n, x, derivative=True) - 1j * sps.spherical_yn(n, x, derivative=True)


def sphModalCoeffs(N, kr, arrayType="rigid", dirCoeff=0.0):
    """Modal coefficients b_n(kr), [len(kr) x (N+1)] (polarch sphModalCoeffs; call site
    dependencies/getSMAIRMatrix.m:107).  rigid: 4 pi i^n (j_n - j_n'/h_n^(2)' h_n^(2)),
    open: 4 pi i^n j_n.  kr == 0 -> [4 pi, 0, 0, ...]; NaN (huge orders) -> 0.
    Sign / conjugation / scale are pinned (to ~5 %) by the cross-fixture physics KAT."""
    kr = np.asarray(kr, dtype=np.float64).ravel()
    b = np.zeros((kr.size, N + 1), dtype=np.complex128)
    nz = kr != 0
    x = kr[nz]
    for n in range(N + 1):
        if arrayType == "open":
            t = 4 * np.pi * (1j ** n) * _sph_besselj(n, x)
        elif arrayType == "rigid":
            with np.errstate(all="ignore"):
                jn = _sph_besselj(n, x)
                jnp_ = _dsph_besselj(n, x)
                hn = _sph_hankel2(n, x)
                hnp = _dsph_hankel2(n, x)
                t = 4 * np.pi * (1j ** n) * (jn - (jnp_ / hnp) * hn)
        else:
            raise ValueError("arrayType must be 'rigid' or 'open'")
        b[nz, n] = t
        b[~nz, n] = 4 * np.pi if n == 0 else 0.0
    b[np.isnan(b)] = 0
    return b


def getShFreqDomainConjugate(Wpos):
    """Positive-frequency half [P x C] (complex-SH filters) -> full spectrum [nfft x C],
    nfft = 2(P-1), such that the time-domain filters obey w_{n,-m} = (-1)^m conj(w_{n,m}):
    W(nfft-k,(n,m)) = (-1)^m conj(W(k,(n,-m))), k = 1..P-2 (call sites lib/getMagLsFilters.m:78-79,
    lib/getEMagLsFilters.m:118-119; semantics pinned by the fixture symmetry KAT)."""
    Wpos = np.asarray(Wpos)
    P, C = Wpos.shape
    N = int(round(math.sqrt(C))) - 1
    assert (N + 1) ** 2 == C
    neg = np.empty((P - 2, C), dtype=np.complex128)
    for n in range(N + 1):
        for m in range(-n, n + 1):
            neg[:, n * n + n + m] = (-1) ** m * np.conj(Wpos[1 : P - 1, n * n + n - m])
    return np.vstack([Wpos, neg[::-1]])


# --------------------------------------------------------------------------------------------
# MathWorks built-ins
# --------------------------------------------------------------------------------------------
def pinv(A):
    """MATLAB pinv: SVD, tol = max(size(A)) * eps(norm(A))."""
    U, s, Vh = np.linalg.svd(A, full_matrices=False)
    tol = max(A.shape) * np.spacing(s[0])
    r = int(np.sum(s > tol))
    return (Vh[:r].conj().T / s[:r]) @ U[:, :r].conj().T


def hann(N):
    """MATLAB hann(N) (symmetric): 0.5 (1 - cos(2 pi n/(N-1)))."""
    if N == 1:
        return np.ones(1)
    n = np.arange(N)
    return 0.5 * (1 - np.cos(2 * np.pi * n / (N - 1)))


def grpdelay_fir(b, nfft_half_plus1):
    """grpdelay(b,1,f,fs) for FIR b at f = linspace(0, fs/2, P): Re{sum n b[n] z^-n / sum b[n] z^-n},
    bins with |den| < 10 eps -> 0 (call site lib/getEMagLsFilters.m:74-75)."""
    b = np.asarray(b, dtype=np.float64).ravel()
    P = nfft_half_plus1
    w = np.pi * np.arange(P) / (P - 1)
    n = np.arange(b.size)
    E = np.exp(-1j * np.outer(w, n))
    den = E @ b
    num = E @ (n * b)
    bad = np.abs(den) < 10 * np.finfo(float).eps
    num[bad] = 0
    den[bad] = 1
    return np.real(num / den)


def fftfilt(b, x):
    """fftfilt(b,x): first len(x) samples of the linear convolution, per column
    (call site dependencies/binauralDecode.m:40-41)."""
    b = np.asarray(b)
    x = np.asarray(x)
    n = x.shape[0]
    L = n + b.shape[0] - 1
    nf = 1 << (L - 1).bit_length()
    y = np.fft.ifft(np.fft.fft(b, nf, axis=0) * np.fft.fft(x, nf, axis=0), axis=0)[:n]
    if np.isrealobj(b) and np.isrealobj(x):
        y = y.real
    return y


# --------------------------------------------------------------------------------------------
# dependencies/*.m
# --------------------------------------------------------------------------------------------
def sh_repToOrder(v):
    """dependencies/sh_repToOrder.m:15-20: (n+1) per-order weights -> (n+1)^2 ACN channels."""
    v = np.asarray(v)
    n = v.shape[0] - 1
    out = np.zeros(((n + 1) ** 2,) + v.shape[1:], dtype=v.dtype)
    for nn in range(n + 1):
        out[nn * nn : (nn + 1) ** 2] = v[nn]
    return out


def getFadeWindow(irLen, relFadeLen=0.15):
    """dependencies/getFadeWindow.m:9-16.  MATLAB round() = half away from zero."""
    nf = int(math.floor(relFadeLen * irLen + 0.5))
    h = hann(2 * nf)
    return np.concatenate([h[:nf], np.ones(irLen - 2 * nf), h[nf:]])


def applySubsampleDelay(sig, delay_samples):
    """dependencies/applySubsampleDelay.m:10-18.  sig [n x ...]; delay scalar or broadcastable
    over the trailing dims."""
    sig = np.asarray(sig)
    n = sig.shape[0]
    omega = np.linspace(0, 0.5, n // 2 + 1)
    delay = np.asarray(delay_samples, dtype=np.float64)
    shp = (omega.size,) + (1,) * (sig.ndim - 1)
    e = np.exp(-1j * 2 * np.pi * omega.reshape(shp) * delay)
    e = np.broadcast_to(e, (omega.size,) + np.broadcast_shapes(e.shape[1:], sig.shape[1:])).copy()
    e[-1] = np.real(e[-1])
    e = np.concatenate([e, np.conj(e[-2:0:-1])], axis=0)
    return np.fft.ifft(np.fft.fft(sig, axis=0) * e, axis=0)


def simulation_order(order, fs, radius):
    """dependencies/getSMAIRMatrix.m:95"""
    return max(int(order), int(math.ceil(fs * math.pi * radius / C_SOUND)))


def emagls2_simulation_order(fs, radius):
    """Simulation order of getEMagLs2Filters: params.order is left unset there (lib/getEMagLs2Filters.m:51-63), hence 4."""
    return simulation_order(SMAIR_DEFAULT_ORDER, fs, radius)


def getSMAIRMatrix(order, fs, irLen, smaRadius, smaDesignAziZenRad, shDefinition="real",
                   returnRawMicSigs=False, arrayType="rigid", shFunction=None, oversamplingFactor=1, radialFilter="none",
                   regulConst=1e-2, noiseGainDb=20.0):
    """dependencies/getSMAIRMatrix.m:86-141, plane-wave model.  The keyword defaults here are what the five entry points pass
    (oversamplingFactor=1, radialFilter='none': lib/getEMagLsFilters.m:51-63), NOT the reference's struct defaults
    ('regul' -- which getRadialFilter.m:63-64 rejects -- and 4, getSMAIRMatrix.m:50-51,70-71).
    Returns [C x S x P] like the reference."""
    nfft = oversamplingFactor * irLen
    assert nfft % 2 == 0
    f = np.linspace(0, fs / 2, nfft // 2 + 1)
    simOrder = simulation_order(order, fs, smaRadius)
    S = (simOrder + 1) ** 2
    nOut = (order + 1) ** 2
    P = f.size
    M = smaDesignAziZenRad.shape[0]
    Y_Hi = (shFunction or getSH)(simOrder, smaDesignAziZenRad, shDefinition)   # params.shFunction (:101)
    Y_Lo_pinv = pinv(Y_Hi[:, :nOut])
    bnAll = -sphModalCoeffs(simOrder, 2 * np.pi * f / C_SOUND * smaRadius, arrayType).T  # (simOrder+1) x P
    rows = M if returnRawMicSigs else nOut
    out = np.zeros((rows, S, P), dtype=np.complex128)
    for k in range(P):
        Bn = sh_repToOrder(bnAll[:, k])
        if k == P - 1:
            Bn = np.real(Bn)  # :115-117
        pM = Y_Hi * Bn[None, :]
        out[:, :, k] = pM if returnRawMicSigs else Y_Lo_pinv @ pM
    if not returnRawMicSigs and str(radialFilter).lower() != "none":   # :129-138
        radFilts = getRadialFilter(order, fs, smaRadius, irLen=irLen, oversamplingFactor=oversamplingFactor, radialFilter=str(radialFilter).lower(),
                                   arrayType=arrayType, regulConst=regulConst, noiseGainDb=noiseGainDb).T
        for k in range(P):
            BnTi = sh_repToOrder(radFilts[:, k])
            out[:, :, k] = BnTi[:, None] * out[:, :, k]
            if k == P - 1:   # :135-137: the (real part of the) filter once more, on the already filtered slice
                out[:, :, k] = np.real(BnTi)[:, None] * out[:, :, k]
    return out, simOrder


# --------------------------------------------------------------------------------------------
# lib/*.m
# --------------------------------------------------------------------------------------------
def getLsFilters(hL, hR, aziRad, zenRad, order, shDefinition="real", shFunction=None):
    """lib/getLsFilters.m:30-34"""
    Y_conj = (shFunction or getSH)(order, np.column_stack([aziRad, zenRad]), shDefinition).conj().T
    Y_pinv = pinv(Y_conj)
    return hL @ Y_pinv, hR @ Y_pinv


def _design_consts(fs, length, f_trans):
    nfft = min(NFFT_MAX_LEN, 2 * length)
    f = np.linspace(0, fs / 2, nfft // 2 + 1)
    k_cut = int(math.ceil(f_trans / f[1]))  # 1-based MATLAB index
    return nfft, f, f.size, k_cut


def _pad(h, nfft):
    out = np.zeros((nfft, h.shape[1]), dtype=h.dtype)
    out[: h.shape[0]] = h
    return out


def _grp_delay(h_padded, P):
    return float(np.median(grpdelay_fir(h_padded.sum(axis=1), P)))


def _finish(W_l, W_r, P, nfft, length, is_real_basis, delayL, delayR, integer_shift=False, conj_fn=None):
    """DC rule, spectrum completion, ifft, shift, truncate, fade
    (lib/getEMagLsFilters.m:110-142; lib/getEMagLsFiltersFromAtf.m:125-151)."""
    out = []
    for W, dly in ((W_l, delayL), (W_r, delayR)):
        if is_real_basis:
            Wf = np.vstack([W[:P], np.conj(W[P - 2 : 0 : -1])])
        else:
            Wf = (conj_fn or getShFreqDomainConjugate)(W[:P])
        w = np.fft.ifft(Wf, axis=0)
        if integer_shift:
            w = np.roll(w, int(dly), axis=0)
        else:
            w = applySubsampleDelay(w, dly)
        n_shift = nfft // 2
        w = w[n_shift - length // 2 : n_shift + length // 2]
        w = w * getFadeWindow(length)[:, None]
        out.append(w)
    return out


def getMagLsFilters(hL, hR, aziRad, zenRad, order, fs, length, shDefinition="real", shFunction=None, _Y=None, _conj_fn=None,
                    applyDiffusenessConst=False):
    """lib/getMagLsFilters.m:30-98"""
    assert length >= hL.shape[0], "HRIR len too short"
    nfft, f, P, k_cut = _design_consts(fs, length, max(F_CUT_MIN_FREQ, 500 * order))
    if _Y is None:
        _Y = (shFunction or getSH)(order, np.column_stack([aziRad, zenRad]), shDefinition)
    Y_conj = _Y.conj().T
    Y_pinv = pinv(Y_conj)
    is_real = np.isrealobj(Y_conj)
    hL = _pad(hL, nfft)
    hR = _pad(hR, nfft)
    grpD = (_grp_delay(hL, P), _grp_delay(hR, P))
    W = []
    Hs = []
    for h, g in ((hL, grpD[0]), (hR, grpD[1])):
        h = applySubsampleDelay(h, -g)
        w_LS = h @ Y_pinv
        H = np.fft.fft(h, axis=0)
        Hs.append(H)
        Wm = np.fft.fft(w_LS, axis=0).astype(np.complex128)
        for k in range(k_cut, P + 1):  # 1-based
            phi = np.angle(Wm[k - 2] @ Y_conj)
            t = np.abs(H[k - 1]) * np.exp(1j * phi)
            if k == P:
                t = np.real(t)
            Wm[k - 1] = t @ Y_pinv
        W.append(Wm)
    if applyDiffusenessConst:   # (MagLS keeps its least-squares DC bin: bins 2..P as for the array variants)
        W[0], W[1] = applyDiffusenessConstraint(W[0], W[1], Hs[0], Hs[1], lambda k: Y_conj, P)
    n_shift = nfft // 2
    wL, wR = _finish(W[0], W[1], P, nfft, length, is_real, n_shift, n_shift + (grpD[1] - grpD[0]), conj_fn=_conj_fn)
    if is_real:
        wL, wR = wL.real, wR.real
    return wL, wR


def _emagls_core(HL, HR, pwGrid_of_k, P, k_cut, C, collect=None, diffuseness=False):
    """The per-bin loop shared by lib/getEMagLsFilters.m:85-106, lib/getEMagLs2Filters.m:85-105 and
    lib/getEMagLsFiltersFromAtf.m:100-120.  HL/HR are [>=P x D]; pwGrid_of_k(k) -> [C x D] (k 1-based)."""
    W_l = np.zeros((P, C), dtype=np.complex128)
    W_r = np.zeros((P, C), dtype=np.complex128)
    for k in range(2, P + 1):
        pw = pwGrid_of_k(k)
        U, s, Vh = np.linalg.svd(pw.T, full_matrices=False)  # svd(pwGrid.','econ','vector')
        s_reg = 1.0 / np.maximum(s, SVD_REGUL_CONST * s.max())
        Y_reg_inv = np.conj(U) @ (s_reg[:, None] * Vh.conj())  # conj(U) * (s .* V.')
        if collect is not None:
            collect(k, pw, s, Y_reg_inv)
        if k < k_cut:
            W_l[k - 1] = HL[k - 1] @ Y_reg_inv
            W_r[k - 1] = HR[k - 1] @ Y_reg_inv
        else:
            phi_l = np.angle(W_l[k - 2] @ pw)
            phi_r = np.angle(W_r[k - 2] @ pw)
            tl = np.abs(HL[k - 1]) * np.exp(1j * phi_l)
            tr = np.abs(HR[k - 1]) * np.exp(1j * phi_r)
            if k == P:
                tl, tr = np.real(tl), np.real(tr)
            W_l[k - 1] = tl @ Y_reg_inv
            W_r[k - 1] = tr @ Y_reg_inv
    if diffuseness:
        W_l, W_r = applyDiffusenessConstraint(W_l, W_r, HL, HR, pwGrid_of_k, P)
    W_l[0] = np.real(W_l[1])  # :110-111
    W_r[0] = np.real(W_r[1])
    return W_l, W_r


def _hrir_prologue(hL, hR, nfft, P):
    """lib/getEMagLsFilters.m:72-81"""
    hL = _pad(hL, nfft)
    hR = _pad(hR, nfft)
    gL = _grp_delay(hL, P)
    gR = _grp_delay(hR, P)
    HL = np.fft.fft(applySubsampleDelay(hL, -gL), axis=0)
    HR = np.fft.fft(applySubsampleDelay(hR, -gR), axis=0)
    return HL, HR, gL, gR


def _matmul(A, B):
    """A @ B for a strided complex A (a frequency page of smairMat) and a real or complex B, through BLAS: NumPy multiplies
    non-contiguous or mixed real/complex operands with a scalar loop (3 s instead of 0.05 s per bin at simulation order 44)."""
    A = np.ascontiguousarray(A)
    if np.iscomplexobj(A) and np.isrealobj(B):
        return (A.real @ B) + 1j * (A.imag @ B)
    return A @ B


def _emagls_generic(hL, hR, aziRad, zenRad, micRadius, micAzi, micZen, order, fs, length,
                    shDefinition, raw, collect=None, shFunction=None, applyDiffusenessConst=False):
    assert length >= hL.shape[0], "len too short"
    nfft, f, P, k_cut = _design_consts(fs, length, max(F_CUT_MIN_FREQ, 500 * order))
    # lib/getEMagLs2Filters.m:51-63 never sets params.order, so dependencies/getSMAIRMatrix.m:39-41 defaults it to 4:
    # the eMagLS2 simulation order is max(4, ceil(fs*pi*r/343)) whatever `order` is (`order` only sets f_cut, :47).
    # lib/getEMagLsFilters.m:51 (and both EMA variants, :54 / :52) do pass params.order = order.
    smair_order = SMAIR_DEFAULT_ORDER if raw else order
    smair, simOrder = getSMAIRMatrix(smair_order, fs, nfft, micRadius, np.column_stack([micAzi, micZen]),
                                     shDefinition, returnRawMicSigs=raw, shFunction=shFunction)
    Y_Hi_conj = (shFunction or getSH)(simOrder, np.column_stack([aziRad, zenRad]), shDefinition).conj().T   # :68
    HL, HR, gL, gR = _hrir_prologue(hL, hR, nfft, P)
    C = smair.shape[0]
    W_l, W_r = _emagls_core(HL, HR, lambda k: _matmul(smair[:, :, k - 1], Y_Hi_conj), P, k_cut, C, collect,
                            diffuseness=applyDiffusenessConst)
    is_real = np.isrealobj(Y_Hi_conj) or raw  # eMagLS2 always mirrors (lib/getEMagLs2Filters.m:113-114)
    n_shift = nfft // 2
    wL, wR = _finish(W_l, W_r, P, nfft, length, is_real, n_shift, n_shift + gR - gL)
    if np.isrealobj(Y_Hi_conj):
        wL, wR = wL.real, wR.real
    return wL, wR


def getEMagLsFilters(hL, hR, aziRad, zenRad, micRadius, micAzi, micZen, order, fs, length,
                     shDefinition="real", collect=None, shFunction=None, applyDiffusenessConst=False):
    """lib/getEMagLsFilters.m:32-142 (applyDiffusenessConst: the removed option, see applyDiffusenessConstraint)"""
    return _emagls_generic(hL, hR, aziRad, zenRad, micRadius, micAzi, micZen, order, fs, length,
                           shDefinition, raw=False, collect=collect, shFunction=shFunction, applyDiffusenessConst=applyDiffusenessConst)


def getEMagLs2Filters(hL, hR, aziRad, zenRad, micRadius, micAzi, micZen, order, fs, length,
                      shDefinition="real", collect=None, shFunction=None, applyDiffusenessConst=False):
    """lib/getEMagLs2Filters.m:32-135"""
    return _emagls_generic(hL, hR, aziRad, zenRad, micRadius, micAzi, micZen, order, fs, length,
                           shDefinition, raw=True, collect=collect, shFunction=shFunction, applyDiffusenessConst=applyDiffusenessConst)



# --------------------------------------------------------------------------------------------
# Equatorial microphone arrays in circular harmonics
# --------------------------------------------------------------------------------------------
def getCH(N, aziRad, basisType="real"):
    """dependencies/getCH.m:17-28: circular harmonics [numDirs x 2N+1], ordered [C_0, C_-1, C_1, ..., C_-N, C_N];
    real: sqrt(2) sin(m phi) / sqrt(2) cos(m phi); complex: exp(-/+ 1i m phi)."""
    azi = np.asarray(aziRad, dtype=float).reshape(-1)
    Y = np.zeros((azi.size, 2 * N + 1), dtype=np.complex128 if basisType == "complex" else np.float64)
    Y[:, 0] = 1.0
    for nn in range(1, N + 1):
        if basisType == "real":
            Y[:, 2 * nn - 1] = math.sqrt(2.0) * np.sin(nn * azi)
            Y[:, 2 * nn] = math.sqrt(2.0) * np.cos(nn * azi)
        else:
            Y[:, 2 * nn - 1] = np.exp(-1j * nn * azi)
            Y[:, 2 * nn] = np.exp(1j * nn * azi)
    return Y


def getChFreqDomainConjugate(Wpos):
    """Positive-frequency half [P x 2N+1] (complex-CH filters) -> full spectrum [nfft x 2N+1] such that the time-domain
    filters obey w_{-m} = conj(w_m) (C_{-m} = conj(C_m), no Condon-Shortley sign): W(nfft-k, m) = conj(W(k, -m)).
    Third-party (thomasdeppisch/sh-symmetries, un-vendored; call site lib/getEMagLsFiltersEMAinCH.m:125-126):
    restated from the symmetry of the basis, no fixture pins it."""
    Wpos = np.asarray(Wpos)
    P, C = Wpos.shape
    N = (C - 1) // 2
    neg = np.empty((P - 2, C), dtype=np.complex128)
    neg[:, 0] = np.conj(Wpos[1 : P - 1, 0])
    for nn in range(1, N + 1):
        neg[:, 2 * nn - 1] = np.conj(Wpos[1 : P - 1, 2 * nn])
        neg[:, 2 * nn] = np.conj(Wpos[1 : P - 1, 2 * nn - 1])
    return np.vstack([Wpos, neg[::-1]])


def getEMagLsFiltersEMAinCH(hL, hR, aziRad, zenRad, micRadius, micAzi, order, fs, length, shDefinition="real"):
    """lib/getEMagLsFiltersEMAinCH.m:32-151: eMagLS for an equatorial array, output in circular harmonics.
    pwGrid_CH = pinv(CH(order, micAzi)) * (pMics(:,:,k) * Y_hor^H)  (:66-75), same per-bin loop (:93-113), DC rule
    (:117-118), Hermitian mirror or the CH conjugate rule (:121-127), shift / truncate / fade (:134-147).
    No fixture pins this function: parity unpinned."""
    assert length >= hL.shape[0], "len too short"
    nfft, f, P, k_cut = _design_consts(fs, length, max(F_CUT_MIN_FREQ, 500 * order))
    micAzi = np.asarray(micAzi, dtype=float).reshape(-1)
    micGrid = np.column_stack([micAzi, np.full(micAzi.size, np.pi / 2)])
    smair, simOrder = getSMAIRMatrix(order, fs, nfft, micRadius, micGrid, shDefinition, returnRawMicSigs=True)
    Y_hor_conj = getSH(simOrder, np.column_stack([aziRad, zenRad]), shDefinition).conj().T
    Y_CH_Mic_pinv = pinv(getCH(order, micAzi, shDefinition))
    HL, HR, gL, gR = _hrir_prologue(hL, hR, nfft, P)
    C = 2 * order + 1
    W_l, W_r = _emagls_core(HL, HR, lambda k: Y_CH_Mic_pinv @ (smair[:, :, k - 1] @ Y_hor_conj), P, k_cut, C)
    is_real = np.isrealobj(Y_hor_conj)
    n_shift = nfft // 2
    out = []
    for W, dly in ((W_l, n_shift), (W_r, n_shift + gR - gL)):
        Wf = np.vstack([W[:P], np.conj(W[P - 2 : 0 : -1])]) if is_real else getChFreqDomainConjugate(W[:P])
        w = applySubsampleDelay(np.fft.ifft(Wf, axis=0), dly)
        w = w[n_shift - length // 2 : n_shift + length // 2] * getFadeWindow(length)[:, None]
        out.append(w.real if is_real else w)
    return out[0], out[1]


# --------------------------------------------------------------------------------------------
# Render-side neighbours (GPU path: a later round)
# --------------------------------------------------------------------------------------------
def getRadialFilter(order, fs, smaRadius, irLen=256, oversamplingFactor=2, radialFilter="tikhonov", arrayType="rigid",
                    regulConst=1e-2, noiseGainDb=None):
    """dependencies/getRadialFilter.m:25-71 (plane-wave model): [nfft/2+1 x order+1] with nfft = oversamplingFactor*irLen."""
    nfft = oversamplingFactor * irLen
    f = np.linspace(0, fs / 2, nfft // 2 + 1)
    if radialFilter == "none":
        return np.ones((nfft // 2 + 1, order + 1))
    bn = sphModalCoeffs(order, 2 * np.pi * f / C_SOUND * smaRadius, arrayType)[:, : order + 1]
    with np.errstate(all="ignore"):
        if radialFilter == "tikhonov":
            rad = np.conj(bn) / (np.conj(bn) * bn + regulConst)
        elif radialFilter == "softlimit":
            g = 10 ** (noiseGainDb / 20)
            rad = 2 * g / np.pi * np.abs(bn) / bn * np.arctan(np.pi / (2 * g * np.abs(bn)))
        elif radialFilter == "full":
            rad = 1.0 / bn
        else:
            raise ValueError("unknown radialFilter")
    if nfft % 2 == 0:
        rad[-1] = np.abs(rad[-1])  # Nyquist bin (:68-70)
    return rad


def applyRadialFilter(inSig, order, fs, smaRadius, irLen, oversamplingFactor=1, **kw):
    """dependencies/applyRadialFilter.m:9-31 with params.nfft = oversamplingFactor * irLen as the harness sets it
    (verifyEMagLs.m:239-250): inSig [numSamples x (order+1)^2] -> filtered, with the filter delay nfft/2 removed."""
    nfft = oversamplingFactor * irLen
    rad = getRadialFilter(order, fs, smaRadius, irLen=irLen, oversamplingFactor=oversamplingFactor, **kw)
    rad = np.where(np.isnan(rad), 0, rad)
    ir = np.fft.ifft(np.vstack([rad, np.conj(rad[-2:0:-1])]), axis=0)
    ir = applySubsampleDelay(ir, nfft / 2)
    ir = ir * getFadeWindow(nfft, 0.05)[:, None]
    sig = np.asarray(inSig, dtype=np.float64)
    if sig.shape[0] < nfft:
        sig = np.vstack([sig, np.zeros((nfft - sig.shape[0], sig.shape[1]))])
    full = np.column_stack([sh_repToOrder(ir[t]) for t in range(ir.shape[0])]).T   # [nfft x (order+1)^2]
    out = np.column_stack([fftfilt(full[:, c].real, sig[:, c]) for c in range(sig.shape[1])])
    return out[nfft // 2:]


# --------------------------------------------------------------------------------------------
# Render-side neighbours: 2-D MagLS, SH encoding, equalisation filters
# --------------------------------------------------------------------------------------------
def getMagLsFilters2D(hLHor, hRHor, horHrirGridAziRad, order, fs, length, chDefinition="real"):
    """lib/getMagLsFilters2D.m:33-103: the loop of getMagLsFilters with Y_conj = getCH(order, azi)' (:49; 2*order+1 channels,
    the numHarmonics of :47 is unused) and getChFreqDomainConjugate for a complex basis (:82-83)."""
    return getMagLsFilters(hLHor, hRHor, None, None, order, fs, length, chDefinition,
                           _Y=getCH(order, horHrirGridAziRad, chDefinition), _conj_fn=getChFreqDomainConjugate)


def encodeSH(smaRecording, micGridAziRad, micGridZenRad, order, shDefinition="real"):
    """verifyEMagLs.m:235-236: E = getSH(order, micGrid, shDefinition).'; shRecording = smaRecording * pinv(E)."""
    E = getSH(order, np.column_stack([micGridAziRad, micGridZenRad]), shDefinition).T
    return np.asarray(smaRecording, dtype=np.float64) @ pinv(E)


def _eq_consts(micRadius, fs, length):
    nfft = min(NFFT_MAX_LEN, 2 * length)
    f = np.linspace(0, fs / 2, nfft // 2 + 1)
    kr = 2 * np.pi * f / C_SOUND * micRadius
    return nfft, kr, int(math.ceil(fs * math.pi * micRadius / C_SOUND))


def _eq_finish(W, nfft, length):
    """:51-67 of lib/getMagLsSphericalHeadFilter.m (same lines in lib/getMagLsArrayDiffuseFilter.m:68-85)"""
    P = nfft // 2 + 1
    Wf = np.concatenate([W[:P], np.conj(W[P - 2:0:-1])])
    w = np.fft.ifft(Wf)
    n_shift = nfft // 2
    w = applySubsampleDelay(w[:, None], n_shift)[:, 0]
    w = w[n_shift - length // 2:n_shift + length // 2]
    return np.real(w * getFadeWindow(length)), Wf


def _df(bn_expanded):
    """rms(abs(x), 2) * sqrt(size(x, 2)) / (4*pi)"""
    a = np.abs(bn_expanded)
    return np.sqrt(np.mean(a * a, axis=1)) * math.sqrt(bn_expanded.shape[1]) / (4 * math.pi)


def getMagLsSphericalHeadFilter(micRadius, order, fs, length):
    """lib/getMagLsSphericalHeadFilter.m:23-67 -> (wShf [len], W_Shf [nfft])."""
    nfft, kr, simOrder = _eq_consts(micRadius, fs, length)
    bn_Hi = sphModalCoeffs(simOrder, kr, "rigid")
    if order > simOrder:
        raise IndexError("bn_Hi(:, 1:order+1): index exceeds the simulation order")
    bn_Lo = bn_Hi[:, :order + 1]
    hi = np.array([sh_repToOrder(r) for r in bn_Hi])
    lo = np.array([sh_repToOrder(r) for r in bn_Lo])
    W = 1.0 / (_df(hi) / _df(lo))
    w, Wf = _eq_finish(W.astype(np.complex128), nfft, length)
    return w, np.real(Wf)


def getMagLsArrayDiffuseFilter(micRadius, micGridAziRad, micGridZenRad, order, fs, length, shDefinition="real", shFunction=None):
    """lib/getMagLsArrayDiffuseFilter.m:33-85"""
    sh = shFunction or getSH
    nfft, kr, simOrder = _eq_consts(micRadius, fs, length)
    hi = np.array([sh_repToOrder(r) for r in sphModalCoeffs(simOrder, kr, "rigid")])
    grid = np.column_stack([micGridAziRad, micGridZenRad])
    Y_Hi_conj = sh(simOrder, grid, shDefinition).conj().T
    bn_Lo = (hi @ Y_Hi_conj) @ sh(order, grid, shDefinition)
    hi_df = _df(hi)
    lo_df = _df(bn_Lo)
    lo_df = lo_df / lo_df[0]
    W_Alias = hi_df / lo_df
    _, W_Shf = getMagLsSphericalHeadFilter(micRadius, order, fs, length)
    W = W_Shf[:W_Alias.shape[0]] * W_Alias
    return _eq_finish(W.astype(np.complex128), nfft, length)[0]


# --------------------------------------------------------------------------------------------
# Diffuseness (covariance) constraint -- SURVEY 8(f) rank 1.  NOT in the reference snapshot (CHANGELOG.md:10-12 removed it;
# verifyEMagLs.m:137-145 hard-defaults applyDiffusenessConst to false); only the *_wDC fixtures survive.  Specified here from
# Zaunschirm, Schoerkhuber, Hoeldrich, "Binaural rendering of Ambisonic signals by head-related impulse response time
# alignment and a diffuseness constraint", JASA 143(6), 2018, sec. IV.C, and pinned STRUCTURALLY against the fixture pairs
# (tools/probe_dc_fixtures.py, tests/test_oracle_kats.py::test_diffuseness_*): per bin the two ears' filters are mixed by a
# 2x2 matrix, W_dc(k,:,[l r]) = W(k,:,[l r]) M(k), that makes the diffuse-field covariance of the rendered ear signals equal
# the one of the HRTF set.  Among all M with M^H Rhat M = R the paper takes the one closest to the identity (min ||Hhat M -
# Hhat||_F); its stationarity condition Rhat M (I + Lambda) = Rhat with a Hermitian multiplier makes M Hermitian, i.e. M is
# the unique Hermitian positive definite solution of  M Rhat M = R  (the fixtures' M(k) are Hermitian to 1e-4 between 1 and
# 20 kHz).  What the fixtures cannot pin without the HRIR set: the exact definition of R for MagLS (cross term) and
# the numerics of the nearly singular bins below ~200 Hz, where the reference's own filters deviate from Hermitian by 1e-2.
# Parity unpinned.
# --------------------------------------------------------------------------------------------
def _sqrtm_hpd(A):
    w, V = np.linalg.eigh(A)
    return (V * np.sqrt(np.maximum(w, 0.0))) @ V.conj().T


def diffuseness_mixing(Rhat, R):
    """The Hermitian positive definite M with M Rhat M = R (2x2 ear covariances):
    M = Rhat^-1/2 (Rhat^1/2 R Rhat^1/2)^1/2 Rhat^-1/2."""
    s = _sqrtm_hpd(Rhat)
    si = np.linalg.inv(s)
    return si @ _sqrtm_hpd(s @ R @ s) @ si


def ear_covariance(hl, hr):
    """[2 x 2] covariance of two ear responses over the directions of the grid: E_d[conj(h_i) h_j] (equal weights, like every
    other sum over the HRIR grid in the reference)."""
    H = np.column_stack([hl, hr])
    return H.conj().T @ H / H.shape[0]


def applyDiffusenessConstraint(W_l, W_r, HL, HR, pwGrid_of_k, P):
    """Bins 2..P (1-based; bin 1 is set from bin 2 afterwards, lib/getEMagLsFilters.m:110-111): rendered HRTFs
    Hhat_e = W_e(k,:) pwGrid_k over the HRIR grid, R from the time-aligned HRTFs HL/HR, W(k,:,[l r]) <- W(k,:,[l r]) M(k)."""
    W_l, W_r = W_l.copy(), W_r.copy()
    for k in range(2, P + 1):
        pw = pwGrid_of_k(k)
        M = diffuseness_mixing(ear_covariance(W_l[k - 1] @ pw, W_r[k - 1] @ pw), ear_covariance(HL[k - 1], HR[k - 1]))
        wl, wr = W_l[k - 1].copy(), W_r[k - 1].copy()
        W_l[k - 1] = wl * M[0, 0] + wr * M[1, 0]
        W_r[k - 1] = wl * M[0, 1] + wr * M[1, 1]
    return W_l, W_r


def fit_ear_mixing(Wl, Wr, Dl, Dr):
    """Least-squares 2x2 M with [Dl Dr] = [Wl Wr] M for one bin (columns = ears, rows = channels) and its relative residual."""
    A = np.column_stack([Wl, Wr])
    B = np.column_stack([Dl, Dr])
    M = np.linalg.lstsq(A, B, rcond=None)[0]
    return M, float(np.linalg.norm(A @ M - B) / max(np.linalg.norm(B), 1e-300))


# --------------------------------------------------------------------------------------------
# Equatorial microphone arrays in spherical harmonics (second half; GPU path: next round)
# --------------------------------------------------------------------------------------------
def getNnm(N, zenRad, harmonicsDef="real"):
    """dependencies/getNnm.m:10-31: the SHs without their azimuth factor, [(N+1)^2].  MATLAB's legendre() and
    scipy.special.lpmv() both carry the Condon-Shortley phase."""
    from scipy.special import lpmv
    out = np.zeros((N + 1) ** 2)
    x = math.cos(zenRad)
    for nn in range(N + 1):
        for mm in range(-nn, nn + 1):
            am = abs(mm)
            Pn = float(lpmv(am, nn, x))
            if harmonicsDef == "complex":
                Pnm = (-1) ** am * math.factorial(nn - am) / math.factorial(nn + am) * Pn if mm < 0 else Pn
                v = math.sqrt((2 * nn + 1) * math.factorial(nn - mm) / (4 * math.pi * math.factorial(nn + mm))) * Pnm
            else:
                v = (-1) ** mm * math.sqrt((2 * nn + 1) * math.factorial(nn - am) / (4 * math.pi * math.factorial(nn + am))) * Pn
            out[nn * nn + nn + mm] = v
    return out


def getChToShExpansionMatrix(order, harmonicsDef="real"):
    """dependencies/getChToShExpansionMatrix.m:11-18: [(N+1)^2 x 2N+1], circular -> equatorial spherical harmonics."""
    J = np.zeros(((order + 1) ** 2, 2 * order + 1))
    Nnm = getNnm(order, math.pi / 2, harmonicsDef)
    for n in range(order + 1):
        for m in range(-n, n + 1):
            J[n * n + n + m, 2 * abs(m) - (1 if m < 0 else 0)] = Nnm[n * n + n + m]
    return J


def shRotationForElevation(aziRad, zenRad, order, shDefinition="real"):
    """The matrix lib/getEMagLsFiltersEMAinSH.m:96-98 multiplies a ROW of SH coefficients with: rotate to the front,
    tilt to the elevation pi/2 - zen, rotate back to the azimuth -- i.e. the rotation about the horizontal axis
    perpendicular to the azimuth that carries the direction (azi, pi/2) to (azi, zen).
    Third party and un-vendored (polarch euler2rotationMatrix + getSHrotMtx): restated from this stated intent, NOT from the
    library's code, so its sign conventions are pinned only by the physics test in tests/test_oracle_kats.py
    (the coefficients of a horizontal plane wave become those of the elevated one).  Parity unpinned.
    With f'(x) = f(R^-1 x) and Y_i(R^-1 x) = sum_j D_ij Y_j(x) the coefficient row transforms as c' = c D; D is obtained by
    least squares on a point set that resolves order N exactly."""
    a = float(aziRad)
    alpha = math.pi / 2 - float(zenRad)                      # elevation
    ax = np.array([math.sin(a), -math.cos(a), 0.0])         # right-hand rotation about it lifts (azi, pi/2) upwards
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + math.sin(alpha) * K + (1 - math.cos(alpha)) * (K @ K)
    n = 4 * (order + 1) ** 2 + 8
    i = np.arange(n) + 0.5
    zen = np.arccos(1 - 2 * i / n)
    azi = np.mod(np.pi * (1 + 5 ** 0.5) * i, 2 * np.pi)
    X = np.column_stack([np.sin(zen) * np.cos(azi), np.sin(zen) * np.sin(azi), np.cos(zen)])
    Xr = X @ R                                              # rows: R^-1 x_p = R^T x_p
    azr, znr = np.arctan2(Xr[:, 1], Xr[:, 0]), np.arccos(np.clip(Xr[:, 2], -1, 1))
    A = getSH(order, np.column_stack([azr, znr]), shDefinition)     # Y_i(R^-1 x_p)
    B = getSH(order, np.column_stack([azi, zen]), shDefinition)     # Y_j(x_p)
    return (pinv(B) @ A).T


def getEMagLsFiltersEMAinSH(hL, hR, aziRad, zenRad, micRadius, micAzi, order, fs, length, shDefinition="real"):
    """lib/getEMagLsFiltersEMAinSH.m:32-180: eMagLS for an equatorial array with the output in spherical harmonics
    (3-DOF head rotation).  Horizontal plane waves on the array (:66-69), circular -> spherical harmonics without radial
    filters (:77-82), per-direction SH rotation to the HRIR elevation (:85-100), the common per-bin loop (:118-139).
    See shRotationForElevation for the one un-vendored convention.  Parity unpinned."""
    assert length >= hL.shape[0], "len too short"
    nfft, f, P, k_cut = _design_consts(fs, length, max(F_CUT_MIN_FREQ, 500 * order))
    aziRad = np.asarray(aziRad, dtype=float).reshape(-1)
    zenRad = np.asarray(zenRad, dtype=float).reshape(-1)
    micAzi = np.asarray(micAzi, dtype=float).reshape(-1)
    micGrid = np.column_stack([micAzi, np.full(micAzi.size, np.pi / 2)])
    ema, simOrder = getSMAIRMatrix(order, fs, nfft, micRadius, micGrid, shDefinition, returnRawMicSigs=True)
    Y_hor_conj = getSH(simOrder, np.column_stack([aziRad, np.full(aziRad.size, np.pi / 2)]), shDefinition).conj().T
    C = (order + 1) ** 2
    D = aziRad.size
    F = pinv(getCH(order, micAzi, shDefinition).T) @ getChToShExpansionMatrix(order, shDefinition).T   # [M x C]
    rot = [None if zenRad[d] == np.pi / 2 else shRotationForElevation(aziRad[d], zenRad[d], order, shDefinition) for d in range(D)]
    is_real = np.isrealobj(Y_hor_conj)

    def pw_of_k(k):
        emaDir = ema[:, :, k - 1] @ Y_hor_conj            # [M x D]
        sh = emaDir.T @ F                                  # row d: emaIrDir(k, :, d) * pinv(YCh.') * J.'
        for d in range(D):
            if rot[d] is not None:
                sh[d] = sh[d] @ rot[d]
        return sh.T                                        # pwGridAll(:, :, k)  [C x D]

    HL, HR, gL, gR = _hrir_prologue(hL, hR, nfft, P)
    W_l, W_r = _emagls_core(HL, HR, pw_of_k, P, k_cut, C)
    n_shift = nfft // 2
    wL, wR = _finish(W_l, W_r, P, nfft, length, is_real, n_shift, n_shift + gR - gL)
    return (wL.real, wR.real) if is_real else (wL, wR)


def _sph2cart_unit(aziZen):
    azi, zen = aziZen[:, 0], aziZen[:, 1]
    ele = np.pi / 2 - zen
    return np.column_stack([np.cos(ele) * np.cos(azi), np.cos(ele) * np.sin(azi), np.sin(ele)])


def matchGrids(hrirGridAziZenRad, atfGridAziZenRad):
    """lib/getEMagLsFiltersFromAtf.m:56-95: the smaller grid picks, per point, its nearest neighbour in the larger one
    (Euclidean distance of the unit vectors, MATLAB min(): FIRST index on exact ties, :84); equal sizes -> the HRIR grid
    counts as the smaller one (:62, min([a b]) returns the first).  Returns (hrir_grid_is_smaller, idx (0-based),
    angular deviation in degrees per point (:85-93))."""
    hc = _sph2cart_unit(np.asarray(hrirGridAziZenRad, dtype=np.float64))
    ac = _sph2cart_unit(np.asarray(atfGridAziZenRad, dtype=np.float64))
    hrir_smaller = hc.shape[0] <= ac.shape[0]
    dirC, matchC = (hc, ac) if hrir_smaller else (ac, hc)
    D = dirC.shape[0]
    idx = np.empty(D, dtype=np.int64)
    dev = np.empty(D)
    for ii in range(D):
        d = np.sqrt(np.sum((matchC - dirC[ii]) ** 2, axis=1))
        idx[ii] = int(np.argmin(d))   # first minimum, like MATLAB's min
        dev[ii] = np.degrees(np.arccos(np.clip(dirC[ii] @ matchC[idx[ii]], -1, 1)))
    return hrir_smaller, idx, dev


def getEMagLsFiltersFromAtf(hL, hR, hrirGridAziZenRad, atfIrs, atfGridAziZenRad, fs, filterLen, fTrans):
    """lib/getEMagLsFiltersFromAtf.m:29-151.  atfIrs [taps x numMics x numAtfDirs].
    Returns (wL, wR, meanGridDeviationDeg) -- the third value is what :96 prints."""
    assert filterLen >= hL.shape[0], "len too short"
    nfft, f, P, kTrans = _design_consts(fs, filterLen, fTrans)
    M = atfIrs.shape[1]
    hL = _pad(hL, nfft)
    hR = _pad(hR, nfft)
    gL = _grp_delay(hL, P)
    gR = _grp_delay(hR, P)
    rnd = lambda v: int(math.floor(abs(v) + 0.5)) * (1 if v >= 0 else -1)  # MATLAB round
    HL = np.fft.fft(np.roll(hL, -rnd(gL), axis=0), axis=0)
    HR = np.fft.fft(np.roll(hR, -rnd(gR), axis=0), axis=0)
    atfs = np.fft.fft(atfIrs, nfft, axis=0)
    hrir_smaller, idx, dev = matchGrids(hrirGridAziZenRad, atfGridAziZenRad)
    if hrir_smaller:
        HLm, HRm = HL, HR
        atfM = atfs[:P][:, :, idx]  # P x M x D
    else:
        HLm, HRm = HL[:P][:, idx], HR[:P][:, idx]
        atfM = atfs[:P]
    W_l, W_r = _emagls_core(HLm, HRm, lambda k: atfM[k - 1], P, kTrans, M)
    n_shift = int(math.floor(nfft / 2 + 0.5))
    wL, wR = _finish(W_l, W_r, P, nfft, filterLen, True, n_shift, n_shift, integer_shift=True)
    return wL.real, wR.real, float(dev.mean())


def binauralDecode(sig, wL, wR, compensateDelay=False):
    """dependencies/binauralDecode.m:33-64 (core loop; resample / rotation / extra convolution are
    out of scope).  sig [numSamples x C], filters [len x C] -> [numSamples x 2] real."""
    n, C = sig.shape
    left = np.zeros(n, dtype=np.complex128)
    right = np.zeros(n, dtype=np.complex128)
    for ii in range(C):
        left += fftfilt(wL[:, ii], sig[:, ii])
        right += fftfilt(wR[:, ii], sig[:, ii])
    out = np.column_stack([left, right])
    if compensateDelay:
        d = wL.shape[0] // 2
        out = out[d - 1 :]
    return out.real


# --------------------------------------------------------------------------------------------
# harness rule
# --------------------------------------------------------------------------------------------
def assert_all_close_metrics(x1, x2):
    """verifyEMagLs.m:370-395.  Returns (norm_diff, signed max dB over bins>=1 of |fft(x1)/fft(x2)|,
    max |dB|)."""
    x1 = np.asarray(x1)
    x2 = np.asarray(x2)
    norm_diff = float(np.max(np.abs(x1 - x2)) / max(np.max(np.abs(x1)), np.max(np.abs(x2))))
    with np.errstate(all="ignore"):
        r = np.abs(np.fft.fft(x1, axis=0) / np.fft.fft(x2, axis=0))[1:]
        db = 20 * np.log10(r)
    db = db[np.isfinite(db)]
    return norm_diff, float(db.max()), float(np.abs(db).max())
