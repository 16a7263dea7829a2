"""Host-side mirror of the reference's MATLAB entry points (same names, argument order and error
behaviour), calling the C ABI in include/emagls.h.  NumPy arrays in MATLAB layout:
hL/hR are [numSamples x numDirections]; filters come back [len x numChannels].

    getLsFilters            lib/getLsFilters.m:1-2
    getMagLsFilters         lib/getMagLsFilters.m:1-2
    getEMagLsFilters        lib/getEMagLsFilters.m:1-2
    getEMagLs2Filters       lib/getEMagLs2Filters.m:1-2
    getEMagLsFiltersEMAinCH lib/getEMagLsFiltersEMAinCH.m:1-2
    getEMagLsFiltersEMAinSH lib/getEMagLsFiltersEMAinSH.m:1-2
    getEMagLsFiltersFromAtf lib/getEMagLsFiltersFromAtf.m:1
    binauralDecode          dependencies/binauralDecode.m:1-2
    getMagLsFilters2D       lib/getMagLsFilters2D.m:1
    getRadialFilter         dependencies/getRadialFilter.m:1   (params struct -> dict or keywords)
    applyRadialFilter       dependencies/applyRadialFilter.m:1
    encodeSH                verifyEMagLs.m:235-236             (smaRecording * pinv(getSH(order, micGrid).'))
    getMagLsSphericalHeadFilter  lib/getMagLsSphericalHeadFilter.m:1
    getMagLsArrayDiffuseFilter   lib/getMagLsArrayDiffuseFilter.m:1
    getSH / sphModalCoeffs  the un-vendored third-party functions the above call
    getCH / getSMAIRMatrix  dependencies/getCH.m:1, dependencies/getSMAIRMatrix.m:1 (the array model materialised)

A custom `shFunction` (a callable with getSH's signature: shFunction(N, [azi zen], shDefinition) -> [dirs x (N+1)^2]) cannot
cross the C ABI as a handle: it is evaluated here, at the simulation order the library reports, and its matrices go through the
emagls_*_with_basis entry points -- exactly what the MEX wrappers do with a MATLAB function handle.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L


def _f(a):
    """MATLAB-layout (column-major) float64 copy and its pointer."""
    a = np.asfortranarray(np.asarray(a, dtype=np.float64))
    return a, a.ctypes.data_as(C.c_void_p)


def _vec(a, n=None, name="vector"):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).ravel())
    if n is not None and a.size != n:
        raise ValueError("%s must have %d elements, got %d" % (name, n, a.size))
    return a, a.ctypes.data_as(C.c_void_p)


def _out(rows, cols, cplx):
    w = np.zeros((rows, cols), dtype=np.complex128 if cplx else np.float64, order="F")
    return w, w.ctypes.data_as(C.c_void_p)


def _basis(shDefinition, shFunction=None):
    if shDefinition is None or shDefinition == "":
        shDefinition = "real"
    if shDefinition not in L.BASIS:
        raise ValueError("shDefinition must be 'real' or 'complex'")
    return L.BASIS[shDefinition], shDefinition == "complex"


def _hrirs(hL, hR):
    hL, pL = _f(hL)
    hR, pR = _f(hR)
    if hL.ndim != 2 or hL.shape != hR.shape:
        raise ValueError("hL and hR must be [numSamples x numDirections] arrays of equal shape")
    return hL, hR, pL, pR


def getSH(N, dirs, basisType="real"):
    dirs = np.asarray(dirs, dtype=np.float64)
    D = dirs.shape[0]
    azi, pa = _vec(dirs[:, 0])
    zen, pz = _vec(dirs[:, 1])
    b, cplx = _basis(basisType, None)
    Y, pY = _out(D, (N + 1) ** 2, cplx)
    L.check(L.load().emagls_sh_basis(int(N), D, pa, pz, b, pY))
    return Y


def sphModalCoeffs(N, kr, arrayType="rigid", dirCoeff=0.0):
    if arrayType != "rigid":
        raise NotImplementedError("only the rigid-sphere model is on the eMagLS path (lib/getEMagLsFilters.m:38)")
    kr, pk = _vec(kr)
    b, pb = _out(kr.size, N + 1, True)
    L.check(L.load().emagls_modal_bn(int(N), kr.size, pk, pb))
    return b


def getCH(N, aziRad, basisType="real"):
    """dependencies/getCH.m:1: circular harmonics [numDirs x 2N+1], ordered [C_0, C_-1, C_1, ..., C_-N, C_N]."""
    azi, pa = _vec(aziRad)
    b, cplx = _basis(basisType)
    Y, pY = _out(azi.size, 2 * int(N) + 1, cplx)
    L.check(L.load().emagls_ch_basis(int(N), azi.size, pa, b, pY))
    return Y


# the reference's own struct defaults (dependencies/getSMAIRMatrix.m:36-84); regulConst: getRadialFilter.m:49-51
_SMAIR_DEFAULTS = {"order": 4, "fs": 48000, "smaRadius": 0.042, "arrayType": "rigid", "radialFilter": "regul", "sourceDist": 2,
                   "dirCoeff": 0, "waveModel": "planeWave", "noiseGainDb": 20, "oversamplingFactor": 4, "irLen": 2048,
                   "returnRawMicSigs": False, "shDefinition": "real", "regulConst": 1e-2}


def getSMAIRMatrix(params=None, **kw):
    """dependencies/getSMAIRMatrix.m:1: the array model itself, [numShsOut | numMics x numShsSimulation x numFreqs] complex.
    `params` is the reference's struct as a dict (order, fs, irLen, oversamplingFactor, smaRadius, smaDesignAziZenRad,
    shDefinition, returnRawMicSigs, radialFilter, ...; plane-wave model, rigid sphere, built-in getSH).  Fields left out take
    the reference's defaults (:36-84) -- including radialFilter 'regul', which getRadialFilter.m:63-64 rejects: like there, a
    call that wants the SH-domain model must name its radial filter ('none', 'tikhonov', 'softlimit', 'full').  The one field
    without a default here is smaDesignAziZenRad (the reference loads a t-design file that is not part of its repository).
    Returns (smairMat, params) like the reference, with params['simulationOrder'] added."""
    p = dict(_SMAIR_DEFAULTS)
    p.update(params or {})
    p.update(kw)
    if "smaDesignAziZenRad" not in p:
        raise KeyError("params.smaDesignAziZenRad is required")
    if str(p["waveModel"]).lower() != "planewave" or p["arrayType"] != "rigid" or p["dirCoeff"] != 0:
        raise NotImplementedError("only the plane-wave model of a rigid sphere is built in (what the filter designs use)")
    if p.get("shFunction") is not None:
        raise NotImplementedError("getSMAIRMatrix with a custom shFunction is not supported")
    kind = str(p["radialFilter"]).lower()
    if p["returnRawMicSigs"]:
        kind = "none"          # (:124-126: the radial filter is never looked at for raw microphone signals)
    elif kind not in L.RADIAL:
        raise ValueError('Unkown radialFilter parameter "%s".' % p["radialFilter"])   # getRadialFilter.m:64, spelling kept
    grid = np.asarray(p["smaDesignAziZenRad"], dtype=np.float64)
    azi, pa = _vec(grid[:, 0])
    zen, pz = _vec(grid[:, 1])
    b, _ = _basis(p["shDefinition"])
    nfft = int(p["oversamplingFactor"]) * int(p["irLen"])
    order, M = int(p["order"]), azi.size
    so = int(max(order, np.ceil(float(p["fs"]) * np.pi * float(p["smaRadius"]) / 343.0)))
    rows = M if p["returnRawMicSigs"] else (order + 1) ** 2
    out = np.zeros((rows, (so + 1) ** 2, nfft // 2 + 1), dtype=np.complex128, order="F")
    sim = C.c_int(0)
    L.check(L.load().emagls_get_smair_matrix(order, float(p["fs"]), int(p["irLen"]), int(p["oversamplingFactor"]), float(p["smaRadius"]),
                                             pa, pz, M, b, 1 if p["returnRawMicSigs"] else 0, L.RADIAL[kind], float(p["regulConst"]),
                                             float(p["noiseGainDb"]), out.ctypes.data_as(C.c_void_p), C.byref(sim)))
    assert sim.value == so
    p["simulationOrder"] = so
    return out, p


def _sh_matrix(shFunction, n, azi, zen, shDefinition, cplx, rows):
    """Evaluate a custom shFunction and check its result: [rows x (n+1)^2], real or complex like the basis."""
    Y = np.asarray(shFunction(int(n), np.column_stack([azi, zen]), shDefinition if shDefinition else "real"))
    if Y.shape != (rows, (n + 1) ** 2):
        raise ValueError("shFunction returned %s, expected (%d, %d)" % (Y.shape, rows, (n + 1) ** 2))
    if np.iscomplexobj(Y) != cplx:
        raise ValueError("shFunction returned a %s matrix for shDefinition=%r" % ("complex" if np.iscomplexobj(Y) else "real", shDefinition))
    Y = np.asfortranarray(Y, dtype=np.complex128 if cplx else np.float64)
    return Y, Y.ctypes.data_as(C.c_void_p)


def getLsFilters(hL, hR, hrirGridAziRad, hrirGridZenRad, order, shDefinition="real", shFunction=None):
    b, cplx = _basis(shDefinition, shFunction)
    hL, hR, pL, pR = _hrirs(hL, hR)
    n, D = hL.shape
    azi, pa = _vec(hrirGridAziRad, D, "hrirGridAziRad")
    zen, pz = _vec(hrirGridZenRad, D, "hrirGridZenRad")
    wL, pwL = _out(n, (order + 1) ** 2, cplx)
    wR, pwR = _out(n, (order + 1) ** 2, cplx)
    if shFunction is not None:
        Y, pY = _sh_matrix(shFunction, order, azi, zen, shDefinition, cplx, D)
        L.check(L.load().emagls_get_ls_filters_with_basis(pL, pR, n, D, pY, int(order), b, pwL, pwR))
        return wL, wR
    L.check(L.load().emagls_get_ls_filters(pL, pR, n, D, pa, pz, int(order), b, pwL, pwR))
    return wL, wR


def _no_handle_with_dc(shFunction, applyDiffusenessConst):
    if applyDiffusenessConst and shFunction is not None:
        raise NotImplementedError("applyDiffusenessConst with a custom shFunction is not supported")


def getMagLsFilters(hL, hR, hrirGridAziRad, hrirGridZenRad, order, fs, len, shDefinition="real", shFunction=None,
                    applyDiffusenessConst=False):
    """lib/getMagLsFilters.m:1-2.  applyDiffusenessConst: the option the reference removed (its stale docstring :4 still lists
    it); specification: DESIGN.md section 7."""
    _no_handle_with_dc(shFunction, applyDiffusenessConst)
    b, cplx = _basis(shDefinition, shFunction)
    hL, hR, pL, pR = _hrirs(hL, hR)
    n, D = hL.shape
    azi, pa = _vec(hrirGridAziRad, D, "hrirGridAziRad")
    zen, pz = _vec(hrirGridZenRad, D, "hrirGridZenRad")
    wL, pwL = _out(int(len), (order + 1) ** 2, cplx)
    wR, pwR = _out(int(len), (order + 1) ** 2, cplx)
    if shFunction is not None:
        Y, pY = _sh_matrix(shFunction, order, azi, zen, shDefinition, cplx, D)
        L.check(L.load().emagls_get_magls_filters_with_basis(pL, pR, n, D, pY, int(order), float(fs), int(len), b, pwL, pwR))
        return wL, wR
    if applyDiffusenessConst:
        L.check(L.load().emagls_get_magls_filters_dc(pL, pR, n, D, pa, pz, int(order), float(fs), int(len), 1, b, pwL, pwR))
        return wL, wR
    L.check(L.load().emagls_get_magls_filters(pL, pR, n, D, pa, pz, int(order), float(fs), int(len), b, pwL, pwR))
    return wL, wR


def _sma(fn_name, raw, hL, hR, azi, zen, micRadius, micAzi, micZen, order, fs, len, shDefinition, shFunction, dc=False):
    _no_handle_with_dc(shFunction, dc)
    b, cplx = _basis(shDefinition, shFunction)
    hL, hR, pL, pR = _hrirs(hL, hR)
    n, D = hL.shape
    azi, pa = _vec(azi, D, "hrirGridAziRad")
    zen, pz = _vec(zen, D, "hrirGridZenRad")
    micAzi, pma = _vec(micAzi)
    micZen, pmz = _vec(micZen, micAzi.size, "micGridZenRad")
    M = micAzi.size
    C_ = M if raw else (order + 1) ** 2
    wL, pwL = _out(int(len), C_, cplx)
    wR, pwR = _out(int(len), C_, cplx)
    if shFunction is not None:
        # lib/getEMagLsFilters.m:68 and dependencies/getSMAIRMatrix.m:101 call the handle at the simulation order
        so = L.load().emagls_simulation_order(L.KIND_EMAGLS2 if raw else L.KIND_EMAGLS, int(order), float(fs), float(micRadius))
        Yh, pYh = _sh_matrix(shFunction, so, azi, zen, shDefinition, cplx, D)
        Ym, pYm = _sh_matrix(shFunction, so, micAzi, micZen, shDefinition, cplx, M)
        fn = getattr(L.load(), fn_name + "_with_basis")
        L.check(fn(pL, pR, n, D, pYh, float(micRadius), pYm, M, int(order), float(fs), int(len), b, pwL, pwR))
        return wL, wR
    if dc:
        fn = getattr(L.load(), fn_name + "_dc")
        L.check(fn(pL, pR, n, D, pa, pz, float(micRadius), pma, pmz, M, int(order), float(fs), int(len), 1, b, pwL, pwR))
        return wL, wR
    fn = getattr(L.load(), fn_name)
    L.check(fn(pL, pR, n, D, pa, pz, float(micRadius), pma, pmz, M, int(order), float(fs), int(len), b, pwL, pwR))
    return wL, wR


def getEMagLsFilters(hL, hR, hrirGridAziRad, hrirGridZenRad, micRadius, micGridAziRad, micGridZenRad, order, fs, len,
                     shDefinition="real", shFunction=None, applyDiffusenessConst=False):
    return _sma("emagls_get_emagls_filters", False, hL, hR, hrirGridAziRad, hrirGridZenRad, micRadius, micGridAziRad,
                micGridZenRad, order, fs, len, shDefinition, shFunction, applyDiffusenessConst)


def getEMagLs2Filters(hL, hR, hrirGridAziRad, hrirGridZenRad, micRadius, micGridAziRad, micGridZenRad, order, fs, len,
                      shDefinition="real", shFunction=None, applyDiffusenessConst=False):
    return _sma("emagls_get_emagls2_filters", True, hL, hR, hrirGridAziRad, hrirGridZenRad, micRadius, micGridAziRad,
                micGridZenRad, order, fs, len, shDefinition, shFunction, applyDiffusenessConst)


def designHrirSets(kind, hL, hR, hrirGridAziRad, hrirGridZenRad=None, micRadius=0.0, micGridAziRad=None, micGridZenRad=None, order=4,
                   fs=48000.0, len=512, shDefinition="real"):
    """The loop over HRIR sets around one of the reference's design functions, as ONE call (emagls_design_hrir_sets): hL, hR
    [numSamples x numDirections x numSets] on one grid (and one array); kind in 'ls', 'magls', 'magls2d', 'emagls', 'emagls2',
    'emainch', 'emainsh'.  Returns wL, wR [len x channels x numSets] -- the filters nsets single calls return (lib/getLsFilters.m:30,
    getMagLsFilters.m:30, getMagLsFilters2D.m:1, getEMagLsFilters.m:32, getEMagLs2Filters.m:32, getEMagLsFiltersEMAinCH.m:32)."""
    kinds = {"ls": L.KIND_LS, "magls": L.KIND_MAGLS, "magls2d": L.KIND_MAGLS_2D, "emagls": L.KIND_EMAGLS, "emagls2": L.KIND_EMAGLS2,
             "emainch": L.KIND_EMA_CH, "emainsh": L.KIND_EMA_SH}
    if kind not in kinds:
        raise ValueError("kind must be one of %s" % sorted(kinds))
    b, cplx = _basis(shDefinition)
    hL = np.asfortranarray(hL, dtype=np.float64)
    hR = np.asfortranarray(hR, dtype=np.float64)
    if hL.ndim != 3 or hL.shape != hR.shape:
        raise ValueError("hL / hR must be equal-shaped [numSamples x numDirections x numSets] arrays")
    n, D, nsets = hL.shape
    azi, pa = _vec(hrirGridAziRad, D, "hrirGridAziRad")
    zen, pz = (None, None) if hrirGridZenRad is None else _vec(hrirGridZenRad, D, "hrirGridZenRad")
    arr = kind in ("emagls", "emagls2", "emainch", "emainsh")
    micAzi, pma = _vec(micGridAziRad) if arr else (None, None)
    micZen, pmz = (None, None) if (not arr or micGridZenRad is None) else _vec(micGridZenRad, micAzi.size, "micGridZenRad")
    N = int(order)
    C_ = {"ls": (N + 1) ** 2, "magls": (N + 1) ** 2, "magls2d": 2 * N + 1, "emagls": (N + 1) ** 2, "emainch": 2 * N + 1, "emainsh": (N + 1) ** 2}.get(kind)
    if kind == "emagls2":
        C_ = micAzi.size
    rows = n if kind == "ls" else int(len)
    dt = np.complex128 if cplx else np.float64      # (complex SH definition: complex filters for every kind, eMagLS2 included)
    wL = np.zeros((rows, C_, nsets), dtype=dt, order="F")
    wR = np.zeros((rows, C_, nsets), dtype=dt, order="F")
    L.check(L.load().emagls_design_hrir_sets(kinds[kind], hL.ctypes.data, hR.ctypes.data, n, D, nsets, pa, pz, float(micRadius), pma, pmz,
                                             micAzi.size if arr else 0, N, float(fs), int(len), b, wL.ctypes.data, wR.ctypes.data))
    return wL, wR


def fromAtfHrirSets(hL, hR, hrirGridAziZenRad, atfIrs, atfGridAziZenRad, fs, filterLen, fTrans):
    """getEMagLsFiltersFromAtf (lib/getEMagLsFiltersFromAtf.m:1) for every HRIR set (subject) of hL, hR [numSamples x
    numDirections x numSets] on ONE ATF set, as one call (emagls_from_atf_hrir_sets): the ATF set is uploaded once and its side
    computed once per batch of up to 16 subjects.  Returns wL, wR [filterLen x numMics x numSets], meanGridDevDeg."""
    hL = np.asfortranarray(hL, dtype=np.float64)
    hR = np.asfortranarray(hR, dtype=np.float64)
    if hL.ndim != 3 or hL.shape != hR.shape:
        raise ValueError("hL / hR must be equal-shaped [numSamples x numDirections x numSets] arrays")
    n, D, nsets = hL.shape
    hg = np.asarray(hrirGridAziZenRad, dtype=np.float64)
    ag = np.asarray(atfGridAziZenRad, dtype=np.float64)
    atf = np.asfortranarray(atfIrs, dtype=np.float64)
    taps, M, Da = atf.shape
    azi, pa = _vec(hg[:, 0], D, "hrirGridAziZenRad(:,1)")
    zen, pz = _vec(hg[:, 1], D, "hrirGridAziZenRad(:,2)")
    aazi, paa = _vec(ag[:, 0], Da, "atfGridAziZenRad(:,1)")
    azen, paz = _vec(ag[:, 1], Da, "atfGridAziZenRad(:,2)")
    wL = np.zeros((int(filterLen), M, nsets), dtype=np.float64, order="F")
    wR = np.zeros((int(filterLen), M, nsets), dtype=np.float64, order="F")
    dev = C.c_double(0.0)
    L.check(L.load().emagls_from_atf_hrir_sets(hL.ctypes.data, hR.ctypes.data, n, D, nsets, pa, pz, atf.ctypes.data, taps, M, Da, paa, paz, float(fs),
                                               int(filterLen), float(fTrans), wL.ctypes.data, wR.ctypes.data, C.byref(dev)))
    return wL, wR, dev.value


def getEMagLsFiltersEMAinCH(hL, hR, hrirGridAziRad, hrirGridZenRad, micRadius, micGridAziRad, order, fs, len,
                            shDefinition="real", shFunction=None, chFunction=None):
    """lib/getEMagLsFiltersEMAinCH.m:1-2: eMagLS filters in circular harmonics for an equatorial microphone array;
    returns [len x (2*order+1)] per ear, channels ordered [C_0, C_-1, C_1, ..., C_-N, C_N] (dependencies/getCH.m)."""
    if chFunction is not None or shFunction is not None:
        raise NotImplementedError("custom chFunction / shFunction handles are not supported for the EMA variant; the defaults "
                                  "@getCH / @getSH are built in")
    b, cplx = _basis(shDefinition)
    hL, hR, pL, pR = _hrirs(hL, hR)
    n, D = hL.shape
    azi, pa = _vec(hrirGridAziRad, D, "hrirGridAziRad")
    zen, pz = _vec(hrirGridZenRad, D, "hrirGridZenRad")
    micAzi, pma = _vec(micGridAziRad)
    C_ = 2 * int(order) + 1
    wL, pwL = _out(int(len), C_, cplx)
    wR, pwR = _out(int(len), C_, cplx)
    L.check(L.load().emagls_get_emagls_filters_ema_in_ch(pL, pR, n, D, pa, pz, float(micRadius), pma, micAzi.size, int(order),
                                                         float(fs), int(len), b, pwL, pwR))
    return wL, wR


def getEMagLsFiltersEMAinSH(hL, hR, hrirGridAziRad, hrirGridZenRad, micRadius, micGridAziRad, order, fs, len,
                            shDefinition="real", shFunction=None, chFunction=None):
    """lib/getEMagLsFiltersEMAinSH.m:1-2: eMagLS filters in spherical harmonics for an equatorial microphone array (the
    horizontal sound field is expanded from circular to spherical harmonics and rotated to every HRIR direction's elevation);
    returns [len x (order+1)^2] per ear."""
    if chFunction is not None or shFunction is not None:
        raise NotImplementedError("custom chFunction / shFunction handles are not supported for the EMA variants; the defaults "
                                  "@getCH / @getSH are built in")
    b, cplx = _basis(shDefinition)
    hL, hR, pL, pR = _hrirs(hL, hR)
    n, D = hL.shape
    azi, pa = _vec(hrirGridAziRad, D, "hrirGridAziRad")
    zen, pz = _vec(hrirGridZenRad, D, "hrirGridZenRad")
    micAzi, pma = _vec(micGridAziRad)
    C_ = (int(order) + 1) ** 2
    wL, pwL = _out(int(len), C_, cplx)
    wR, pwR = _out(int(len), C_, cplx)
    L.check(L.load().emagls_get_emagls_filters_ema_in_sh(pL, pR, n, D, pa, pz, float(micRadius), pma, micAzi.size, int(order),
                                                         float(fs), int(len), b, pwL, pwR))
    return wL, wR


def getEMagLsFiltersFromAtf(hL, hR, hrirGridAziZenRad, atfIrs, atfGridAziZenRad, fs, filterLen, fTrans, verbose=True):
    hL, hR, pL, pR = _hrirs(hL, hR)
    n, D = hL.shape
    hg = np.asarray(hrirGridAziZenRad, dtype=np.float64)
    ag = np.asarray(atfGridAziZenRad, dtype=np.float64)
    azi, pa = _vec(hg[:, 0], D, "hrirGridAziZenRad")
    zen, pz = _vec(hg[:, 1], D, "hrirGridAziZenRad")
    atf, patf = _f(atfIrs)
    if atf.ndim != 3:
        raise ValueError("atfIrs must be [taps x numMics x numDirections]")
    taps, M, Da = atf.shape
    aazi, paa = _vec(ag[:, 0], Da, "atfGridAziZenRad")
    azen, paz = _vec(ag[:, 1], Da, "atfGridAziZenRad")
    wL, pwL = _out(int(filterLen), M, False)
    wR, pwR = _out(int(filterLen), M, False)
    dev = C.c_double(0.0)
    L.check(L.load().emagls_get_emagls_filters_from_atf(pL, pR, n, D, pa, pz, patf, taps, M, Da, paa, paz, float(fs),
                                                        int(filterLen), float(fTrans), pwL, pwR, C.byref(dev)))
    if verbose:  # the reference prints this line (lib/getEMagLsFiltersFromAtf.m:96)
        print("Matching HRTF and ATF grids, average grid deviation: %.5g deg" % dev.value)
    return wL, wR


def binauralDecode(sig, inFs, decodingFilterLeft, decodingFilterRight, decodingFilterFs, compensateDelay=False,
                   signal=None, signalFs=None, horRotAngleRad=None):
    """dependencies/binauralDecode.m:1-2.  Real or complex (complex-SH) signals and filters; the output is real: the reference
    forces it and warns with the absolute sum of the discarded imaginary part (:59-64), and so does this function."""
    import warnings
    if decodingFilterFs != inFs or signal is not None or (horRotAngleRad not in (None, 0)):
        raise NotImplementedError("resampling, rotation and the extra convolution are outside the accelerated path")
    in_c = np.iscomplexobj(sig)
    w_c = np.iscomplexobj(decodingFilterLeft) or np.iscomplexobj(decodingFilterRight)

    def arr(a, cplx):
        a = np.asfortranarray(np.asarray(a, dtype=np.complex128 if cplx else np.float64))
        return a, a.ctypes.data_as(C.c_void_p)

    sig, ps = arr(sig, in_c)
    wL, pwL = arr(decodingFilterLeft, w_c)
    wR, pwR = arr(decodingFilterRight, w_c)
    n, Cc = sig.shape
    ln = wL.shape[0]
    if wL.shape != wR.shape or wL.shape[1] != Cc:
        raise ValueError("filters must be [len x numChannels] matching the signal's channel count")
    skip = (ln // 2 - 1) if (compensateDelay and ln // 2 > 0) else 0
    out, po = _out(max(n - skip, 0), 2, False)
    if not (in_c or w_c):
        L.check(L.load().emagls_binaural_decode(ps, n, Cc, pwL, pwR, ln, 1 if compensateDelay else 0, po))
        return out
    im = (C.c_double * 2)(0.0, 0.0)
    L.check(L.load().emagls_binaural_decode_complex(ps, 1 if in_c else 0, n, Cc, pwL, pwR, 1 if w_c else 0, ln,
                                                    1 if compensateDelay else 0, po, im))
    # binauralDecode.m:59-63: `if ~isreal(binauralOut)` -- whenever the accumulated result is a complex array, which it is as
    # soon as a signal or a filter is complex (MATLAB only drops an all-zero imaginary part at the end of an arithmetic
    # operation; a sum that happens to be exactly real is the one case in which the reference stays silent, and so do we)
    if im[0] != 0.0 or im[1] != 0.0:
        warnings.warn("discarding imaginary part with sum of [%.2g, %.2g] in rendering result." % (im[0], im[1]))
    return out


# --------------------------------------------------------------------------------------------
# render-side neighbours
# --------------------------------------------------------------------------------------------
def getMagLsFilters2D(hLHor, hRHor, horHrirGridAziRad, order, fs, len, chDefinition="real"):
    """lib/getMagLsFilters2D.m:1: MagLS filters in circular harmonics for a horizontal HRIR set; [len x (2*order+1)] per ear,
    channels [C_0, C_-1, C_1, ..., C_-N, C_N]."""
    b, cplx = _basis(chDefinition)
    hL, hR, pL, pR = _hrirs(hLHor, hRHor)
    n, D = hL.shape
    azi, pa = _vec(horHrirGridAziRad, D, "horHrirGridAziRad")
    wL, pwL = _out(int(len), 2 * int(order) + 1, cplx)
    wR, pwR = _out(int(len), 2 * int(order) + 1, cplx)
    L.check(L.load().emagls_get_magls_filters_2d(pL, pR, n, D, pa, int(order), float(fs), int(len), b, pwL, pwR))
    return wL, wR


_RADIAL_DEFAULTS = {"radialFilter": "tikhonov", "waveModel": "planeWave", "oversamplingFactor": 2, "irLen": 256, "dirCoeff": 0,
                    "regulConst": 1e-2}   # dependencies/getRadialFilter.m:27-41,58-60


def _radial_params(params, kw):
    p = dict(_RADIAL_DEFAULTS)
    p.update(params or {})
    p.update(kw)
    for k in ("order", "fs", "smaRadius", "arrayType"):
        if k not in p:
            raise KeyError("params.%s is required" % k)       # MATLAB: reference to non-existent field
    kind = str(p["radialFilter"]).lower()
    if kind != "none" and str(p["waveModel"]).lower() == "pointsource":
        raise NotImplementedError('WaveModel parameter "%s" not yet implemented.' % p["waveModel"])    # :50-52
    if kind not in L.RADIAL:
        raise ValueError('Unkown radialFilter parameter "%s".' % p["radialFilter"])                   # :76
    if kind != "none" and (p["arrayType"] != "rigid" or p["dirCoeff"] != 0):
        raise NotImplementedError("only the rigid-sphere model is built in (the harness's arrayType, verifyEMagLs.m:245)")
    if kind == "softlimit" and "noiseGainDb" not in p:
        raise KeyError("params.noiseGainDb is required for the softlimit filter")
    return p, L.RADIAL[kind]


def getRadialFilter(params=None, **kw):
    """dependencies/getRadialFilter.m:1: radFilts [nfft/2+1 x order+1] complex, nfft = oversamplingFactor * irLen.  `params`
    is the reference's struct as a dict (order, fs, smaRadius, arrayType, irLen, oversamplingFactor, radialFilter, regulConst,
    noiseGainDb); keywords override it."""
    p, kind = _radial_params(params, kw)
    nfft = int(p["oversamplingFactor"]) * int(p["irLen"])
    rad, pr = _out(nfft // 2 + 1, int(p["order"]) + 1, True)
    L.check(L.load().emagls_get_radial_filter(int(p["order"]), float(p["fs"]), float(p["smaRadius"]), int(p["irLen"]),
                                              int(p["oversamplingFactor"]), kind, float(p["regulConst"]),
                                              float(p.get("noiseGainDb", float("nan"))), pr))
    return rad


def applyRadialFilter(inSig, params=None, **kw):
    """dependencies/applyRadialFilter.m:1: inSig [numSamples x (order+1)^2] filtered per SH order with the radial-filter
    impulse responses, the delay nfft/2 removed.  params.nfft must be oversamplingFactor * irLen (verifyEMagLs.m:250)."""
    p, kind = _radial_params(params, kw)
    nfft = int(p["oversamplingFactor"]) * int(p["irLen"])
    if "nfft" not in p:
        raise KeyError("params.nfft is required")
    if int(p["nfft"]) != nfft:
        raise ValueError("params.nfft must equal oversamplingFactor * irLen (the reference's arrays do not conform otherwise)")
    sig, ps = _f(inSig)
    C_ = (int(p["order"]) + 1) ** 2
    if sig.ndim != 2 or sig.shape[1] != C_:
        raise ValueError("inSig must be [numSamples x (order+1)^2]")
    lib = L.load()
    if sig.shape[0] < nfft:
        print("applyRadialFilter: short signal, applying zero padding!")      # :21
    rows = lib.emagls_apply_radial_filter_rows(sig.shape[0], int(p["irLen"]), int(p["oversamplingFactor"]))
    out, po = _out(rows, C_, False)
    L.check(lib.emagls_apply_radial_filter(ps, sig.shape[0], int(p["order"]), float(p["fs"]), float(p["smaRadius"]), int(p["irLen"]),
                                           int(p["oversamplingFactor"]), kind, float(p["regulConst"]),
                                           float(p.get("noiseGainDb", float("nan"))), po))
    return out


def encodeSH(smaRecording, micGridAziRad, micGridZenRad, order, shDefinition="real"):
    """verifyEMagLs.m:235-236: shRecording = smaRecording * pinv(getSH(order, micGrid, shDefinition).')."""
    b, cplx = _basis(shDefinition)
    sig, ps = _f(smaRecording)
    if sig.ndim != 2:
        raise ValueError("smaRecording must be [numSamples x numMics]")
    n, M = sig.shape
    azi, pa = _vec(micGridAziRad, M, "micGridAziRad")
    zen, pz = _vec(micGridZenRad, M, "micGridZenRad")
    out, po = _out(n, (int(order) + 1) ** 2, cplx)
    L.check(L.load().emagls_sh_encode(ps, n, M, pa, pz, int(order), b, po))
    return out


def getMagLsSphericalHeadFilter(micRadius, order, fs, len):
    """lib/getMagLsSphericalHeadFilter.m:1: (wShf [len x 1], W_Shf [nfft x 1])."""
    lib = L.load()
    w, pw = _out(int(len), 1, False)
    W, pW = _out(int(lib.emagls_eq_filter_nfft(int(len))), 1, False)
    L.check(lib.emagls_get_magls_spherical_head_filter(float(micRadius), int(order), float(fs), int(len), pw, pW))
    return w, W


def getMagLsArrayDiffuseFilter(micRadius, micGridAziRad, micGridZenRad, order, fs, len, shDefinition="real", shFunction=None):
    """lib/getMagLsArrayDiffuseFilter.m:1: wAdf [len x 1].  A custom shFunction is evaluated here at the simulation order
    ceil(fs*pi*micRadius/343) (:38) and handed over as a matrix; its low-order matrix is taken as the leading columns."""
    import math
    b, cplx = _basis(shDefinition)
    azi, pa = _vec(micGridAziRad)
    zen, pz = _vec(micGridZenRad, azi.size, "micGridZenRad")
    w, pw = _out(int(len), 1, False)
    pY = None
    if shFunction is not None:
        sim = int(math.ceil(float(fs) * math.pi * float(micRadius) / 343.0))
        Y, pY = _sh_matrix(shFunction, sim, azi, zen, shDefinition, cplx, azi.size)
    L.check(L.load().emagls_get_magls_array_diffuse_filter(float(micRadius), pa, pz, azi.size, int(order), float(fs), int(len), b,
                                                           pY, pw))
    return w
