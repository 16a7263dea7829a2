"""The MATLAB gateway mex/emagls_mex.cpp, compiled against a test stand-in for mex.h (tests/mexstub/: MATLAB is not in the
image) and driven from Python: the command dispatch, the argument marshalling (column-major arrays, interleaved complex,
3-D ATF arrays, strings, logicals), the output allocation and the error forwarding are the gateway's own code; the stand-in
only supplies the mx* / mex* functions it calls.  What this cannot show is that MATLAB's real mex.h agrees with the stand-in."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "mexstub")
LIBDIR = os.path.join(ROOT, "emagls_amd", "lib")


@pytest.fixture(scope="module")
def mex():
    if not os.path.exists(os.path.join(LIBDIR, "libemagls.so")):
        pytest.skip("libemagls.so is not built")
    out = os.path.join(STUB, "_build", "libmexharness.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    srcs = [os.path.join(ROOT, "mex", "emagls_mex.cpp"), os.path.join(STUB, "mexstub.cpp")]
    if not os.path.exists(out) or any(os.path.getmtime(s) > os.path.getmtime(out) for s in srcs + [os.path.join(STUB, "mex.h")]):
        cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-shared", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + STUB] + srcs + \
              ["-L" + LIBDIR, "-lemagls", "-Wl,-rpath," + LIBDIR, "-o", out]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
    import torch  # noqa: F401  (first: the library then shares torch's HIP runtime, as in emagls_amd/_lib.py)
    h = C.CDLL(out)
    h.stub_array.restype = C.c_void_p
    h.stub_array.argtypes = [C.c_int, C.POINTER(C.c_size_t), C.c_void_p, C.c_int]
    h.stub_string.restype = C.c_void_p
    h.stub_string.argtypes = [C.c_char_p]
    h.stub_logical.restype = C.c_void_p
    h.stub_logical.argtypes = [C.c_int]
    h.stub_free.argtypes = [C.c_void_p]
    h.stub_ndim.argtypes = [C.c_void_p]
    h.stub_dims.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    h.stub_is_complex.argtypes = [C.c_void_p]
    h.stub_data.restype = C.c_void_p
    h.stub_data.argtypes = [C.c_void_p]
    h.stub_call.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]
    h.stub_struct.restype = C.c_void_p
    h.stub_struct.argtypes = [C.c_size_t, C.c_int, C.POINTER(C.c_char_p)]
    h.stub_struct_set.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
    h.stub_is_cell.argtypes = [C.c_void_p]
    h.stub_cell_get.restype = C.c_void_p
    h.stub_cell_get.argtypes = [C.c_void_p, C.c_size_t]

    class MexCallError(RuntimeError):
        pass

    def to_mx(v):
        if isinstance(v, list) and v and isinstance(v[0], dict):     # a 1 x n struct array (fields: the union of the keys; missing = [])
            names = sorted({k for d in v for k in d})
            arr = (C.c_char_p * len(names))(*[k.encode() for k in names])
            st = h.stub_struct(len(v), len(names), arr)
            for i, d in enumerate(v):
                for f, k in enumerate(names):
                    if k in d and d[k] is not None:
                        h.stub_struct_set(st, i, f, to_mx(d[k]))
            return st
        if isinstance(v, str):
            return h.stub_string(v.encode())
        if isinstance(v, (bool, np.bool_)):
            return h.stub_logical(int(v))
        a = np.asarray(v)
        a = np.asfortranarray(a.astype(np.complex128 if np.iscomplexobj(a) else np.float64))
        if a.ndim < 2:
            a = a.reshape((1, 1) if a.ndim == 0 else (-1, 1), order="F")
        dims = (C.c_size_t * a.ndim)(*a.shape)
        return h.stub_array(a.ndim, dims, a.ctypes.data_as(C.c_void_p), int(np.iscomplexobj(a)))

    def from_mx(p):
        if h.stub_is_cell(p):
            nd = h.stub_ndim(p)
            dims = (C.c_size_t * nd)()
            h.stub_dims(p, dims)
            m, n = int(dims[0]), int(dims[1])
            return [[from_mx(h.stub_cell_get(p, c * m + r)) for c in range(n)] for r in range(m)]    # column-major cells -> rows of a list
        nd = h.stub_ndim(p)
        dims = (C.c_size_t * nd)()
        h.stub_dims(p, dims)
        shape = tuple(int(d) for d in dims)
        n = int(np.prod(shape))
        cplx = bool(h.stub_is_complex(p))
        raw = np.ctypeslib.as_array(C.cast(h.stub_data(p), C.POINTER(C.c_double)), shape=(n * (2 if cplx else 1),)).copy()
        return (raw.view(np.complex128) if cplx else raw).reshape(shape, order="F")

    def call(nlhs, *args):
        """[out1, ...] = emagls_mex(args...)"""
        ins = [to_mx(a) for a in args]
        prhs = (C.c_void_p * len(ins))(*ins)
        plhs = (C.c_void_p * max(nlhs, 1))()
        err = C.create_string_buffer(2048)
        rc = h.stub_call(nlhs, plhs, len(ins), prhs, err, len(err))
        for p in ins:
            h.stub_free(p)
        if rc:
            raise MexCallError(err.value.decode())
        outs = [from_mx(plhs[i]) for i in range(nlhs)]
        for i in range(nlhs):
            h.stub_free(plhs[i])
        return outs

    call.Error = MexCallError
    return call


def test_gateway_compiles_and_dispatches(mex):
    """No GPU needed: the command string, the simulation-order query the wrappers use for caller-evaluated shFunction handles
    (mex/getEMagLsFilters.m), and the gateway's own argument errors."""
    from oracle import emagls_oracle as O
    assert mex(1, "simorder", "emagls", 4, 48000.0, 0.042)[0].item() == O.simulation_order(4, 48000.0, 0.042) == 19
    assert mex(1, "simorder", "emagls2", 1, 48000.0, 0.1)[0].item() == O.emagls2_simulation_order(48000.0, 0.1) == 44
    with pytest.raises(mex.Error, match="first argument must be a command string"):
        mex(0, 3.0)
    with pytest.raises(mex.Error, match="not enough input arguments"):
        mex(2, "nothing", np.zeros((4, 4)), np.zeros((4, 4)))
    with pytest.raises(mex.Error, match="unknown command 'nothing'"):
        mex(2, "nothing", *([np.zeros((4, 4))] * 12))
    with pytest.raises(mex.Error, match="decode needs"):
        mex(1, "decode", np.zeros((4, 4)))
    # 'jobs': the struct array is checked field by field before anything reaches the GPU
    with pytest.raises(mex.Error, match="jobs needs a struct array"):
        mex(1, "jobs", np.zeros((4, 4)))
    h = np.zeros((8, 30))
    ok = dict(kind="magls", hL=h, hR=h, hrirGridAziRad=np.zeros(30), hrirGridZenRad=np.zeros(30), order=1, fs=48000.0, len=16, shDefinition="real")
    with pytest.raises(mex.Error, match=r"jobs\(2\).kind: unknown design kind 'nothing'"):
        mex(1, "jobs", [ok, dict(ok, kind="nothing")])
    with pytest.raises(mex.Error, match=r"jobs\(1\).order is missing"):
        mex(1, "jobs", [{k: v for k, v in ok.items() if k != "order"}])
    with pytest.raises(mex.Error, match=r"jobs\(1\).hrirGridZenRad must have 30 elements"):
        mex(1, "jobs", [dict(ok, hrirGridZenRad=np.zeros(7))])
    with pytest.raises(mex.Error, match=r"jobs\(1\).micRadius is missing"):
        mex(1, "jobs", [dict(ok, kind="emagls2")])


@pytest.fixture(scope="module")
def thin(grids, hrirs):
    sub = slice(0, 2702, 3)
    return dict(hL=hrirs[0][:, sub], hR=hrirs[1][:, sub], azi=grids["azi"][sub], zen=grids["zen"][sub])


@pytest.mark.gpu
def test_gateway_design_calls_match_the_python_binding(mex, grids, thin):
    """'ls' / 'magls' / 'emagls' (both bases) / 'fromatf' (3-D array) / 'decode' (real, complex, compensateDelay) through the
    gateway equal the ctypes binding's results bit for bit: same library, same buffers, the marshalling is what differs."""
    import emagls_amd as E
    from emagls_amd import synth
    hL, hR, azi, zen = thin["hL"], thin["hR"], thin["azi"], thin["zen"]
    wL, wR = mex(2, "ls", hL, hR, azi, zen, 3, "real")
    eL, eR = E.getLsFilters(hL, hR, azi, zen, 3, "real")
    assert wL.shape == eL.shape and np.array_equal(wL, eL) and np.array_equal(wR, eR)
    for basis in ("real", "complex"):
        wL, wR = mex(2, "emagls", hL, hR, azi, zen, grids["mic_radius"], grids["mic_azi"], grids["mic_zen"], 4, 48000.0, 128, basis)
        eL, eR = E.getEMagLsFilters(hL, hR, azi, zen, grids["mic_radius"], grids["mic_azi"], grids["mic_zen"], 4, 48000.0, 128, basis)
        assert wL.dtype == eL.dtype and wL.shape == eL.shape == (128, 25) and np.array_equal(wL, eL) and np.array_equal(wR, eR)
    wL, wR = mex(2, "magls", hL, hR, azi, zen, 2, 48000.0, 128, "complex")
    eL, eR = E.getMagLsFilters(hL, hR, azi, zen, 2, 48000.0, 128, "complex")
    assert np.array_equal(wL, eL) and np.array_equal(wR, eR)
    with pytest.raises(mex.Error, match="shDefinition must be 'real' or 'complex'"):
        mex(2, "ls", hL, hR, azi, zen, 3, "imaginary")
    with pytest.raises(mex.Error, match="eMagLS:native.*len too short"):      # the library's message, forwarded
        mex(2, "magls", hL, hR, azi, zen, 2, 48000.0, 16, "real")
    atf, aazi, azen = synth.glasses_atfs(natf=300, nmics=5, taps=48, fs=48000.0)
    hg, ag = np.column_stack([azi, zen]), np.column_stack([aazi + 0.01, azen])
    wL, wR = mex(2, "fromatf", hL, hR, hg, atf, ag, 48000.0, 128, 2000.0)
    eL, eR = E.getEMagLsFiltersFromAtf(hL, hR, hg, atf, ag, 48000.0, 128, 2000.0, verbose=False)[:2]
    assert wL.shape == (128, 5) and np.array_equal(wL, eL) and np.array_equal(wR, eR)
    # 'sets': 3-D HRIR arrays, a loop over HRIR sets in one call (MagLS and eMagLS2; [] for the arguments a kind does not have)
    h3L = np.stack([hL * (1 + 0.05 * j) for j in range(5)], axis=2)
    h3R = np.stack([hR * (1 - 0.03 * j) for j in range(5)], axis=2)
    empty = np.zeros((0, 0))
    sL, sR = mex(2, "sets", "magls", h3L, h3R, azi, zen, empty, empty, empty, 3, 48000.0, 128, "complex")
    assert sL.shape == (128, 16, 5) and np.iscomplexobj(sL)
    for j in (0, 4):
        eL, eR = E.getMagLsFilters(h3L[:, :, j], h3R[:, :, j], azi, zen, 3, 48000.0, 128, "complex")
        assert np.array_equal(sL[:, :, j], eL) and np.array_equal(sR[:, :, j], eR)
    sL, sR = mex(2, "sets", "emagls2", h3L, h3R, azi, zen, grids["mic_radius"], grids["mic_azi"], grids["mic_zen"], 4, 48000.0, 128, "real")
    assert sL.shape == (128, 32, 5) and not np.iscomplexobj(sL)
    eL, eR = E.getEMagLs2Filters(h3L[:, :, 3], h3R[:, :, 3], azi, zen, grids["mic_radius"], grids["mic_azi"], grids["mic_zen"], 4, 48000.0, 128, "real")
    assert np.array_equal(sL[:, :, 3], eL) and np.array_equal(sR[:, :, 3], eR)
    aL, aR = mex(2, "fromatfsets", h3L, h3R, hg, atf, ag, 48000.0, 128, 2000.0)
    assert aL.shape == (128, 5, 5)
    eL, eR = E.getEMagLsFiltersFromAtf(h3L[:, :, 2], h3R[:, :, 2], hg, atf, ag, 48000.0, 128, 2000.0, verbose=False)[:2]
    assert np.array_equal(aL[:, :, 2], eL) and np.array_equal(aR[:, :, 2], eR)
    with pytest.raises(mex.Error, match="unknown design kind"):
        mex(2, "sets", "fromatf", h3L, h3R, azi, zen, empty, empty, empty, 3, 48000.0, 128, "real")
    rng = np.random.default_rng(3)
    sig = rng.standard_normal((3000, 25))
    fL, fR = E.getEMagLsFilters(hL, hR, azi, zen, grids["mic_radius"], grids["mic_azi"], grids["mic_zen"], 4, 48000.0, 128, "real")
    assert np.array_equal(mex(1, "decode", sig, fL, fR, False)[0], E.binauralDecode(sig, 48000, fL, fR, 48000))
    assert np.array_equal(mex(1, "decode", sig, fL, fR, True)[0], E.binauralDecode(sig, 48000, fL, fR, 48000, True))
    cL, cR = E.getEMagLsFilters(hL, hR, azi, zen, grids["mic_radius"], grids["mic_azi"], grids["mic_zen"], 4, 48000.0, 128, "complex")
    sc = sig + 1j * rng.standard_normal(sig.shape)
    out, imag = mex(2, "decode", sc, cL, cR, False)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ref = E.binauralDecode(sc, 48000, cL, cR, 48000)
    assert np.array_equal(out, ref) and imag.shape == (1, 2) and np.all(imag > 0)


@pytest.mark.gpu
def test_job_list_through_the_gateway(mex, grids, thin):
    """emagls_mex('jobs', struct array) -- mex/designJobs.m -- against emagls_amd.jobs.JobList on the same list: 11 eMagLS designs on
    their own HRIR sets (one chunk on the register-resident sweep), 3 eMagLS2 designs on different array radii, 2 MagLS designs, one
    FromAtf design; the gateway marshals the struct fields into emagls_job records and allocates the cell array of outputs, the
    scheduler behind both is the same: bit for bit."""
    from emagls_amd import _lib as L, synth
    from emagls_amd.jobs import JobList
    hL, hR, azi, zen = thin["hL"], thin["hR"], thin["azi"], thin["zen"]
    maz, mzn, r = grids["mic_azi"], grids["mic_zen"], grids["mic_radius"]
    rng = np.random.default_rng(11)
    sets = [(hL * (1.0 + 0.05 * rng.standard_normal()), hR * (1.0 + 0.05 * rng.standard_normal())) for _ in range(11)]
    atf, aazi, azen = synth.glasses_atfs(natf=300, nmics=5, taps=48, fs=48000.0)
    common = dict(hrirGridAziRad=azi, hrirGridZenRad=zen, fs=48000.0, len=128)
    jobs = [dict(common, kind="emagls", hL=a, hR=b, micRadius=r, micGridAziRad=maz, micGridZenRad=mzn, order=4, shDefinition="complex") for a, b in sets]
    radii = [0.0470, 0.0471, 0.0473]
    jobs += [dict(common, kind="emagls2", hL=hL, hR=hR, micRadius=x, micGridAziRad=maz, micGridZenRad=mzn, order=4, shDefinition="real", simOrderPad=21) for x in radii]
    jobs += [dict(common, kind="magls", hL=a, hR=b, order=3, shDefinition="real") for a, b in sets[:2]]
    jobs += [dict(common, kind="fromatf", hL=hL, hR=hR, atfIrs=atf, atfGridAziRad=aazi + 0.01, atfGridZenRad=azen, fTrans=2000.0)]
    W = mex(1, "jobs", jobs, 32, 4, False)[0]
    assert len(W) == len(jobs) and all(len(row) == 2 for row in W)
    jl = JobList()
    for a, b in sets:
        jl.add(L.KIND_EMAGLS, "complex", 4, 48000.0, 128, a, b, azi, zen, mic_radius=r, mic_azi=maz, mic_zen=mzn, out_shape=(128, 25, True))
    for x in radii:
        jl.add(L.KIND_EMAGLS2, "real", 4, 48000.0, 128, hL, hR, azi, zen, mic_radius=x, mic_azi=maz, mic_zen=mzn, sim_order_pad=21, out_shape=(128, 32, False))
    for a, b in sets[:2]:
        jl.add(L.KIND_MAGLS, "real", 3, 48000.0, 128, a, b, azi, zen, out_shape=(128, 16, False))
    jl.add(L.KIND_FROM_ATF, "real", 0, 48000.0, 128, hL, hR, azi, zen, atf=atf, atf_azi=aazi + 0.01, atf_zen=azen, f_trans=2000.0, out_shape=(128, 5, False))
    jl.run(batch_size=32, in_flight=4)
    for j, ((gl, gr), (el, er)) in enumerate(zip(W, jl.results())):
        assert gl.shape == el.shape and gl.dtype == el.dtype, j
        assert np.array_equal(gl, el) and np.array_equal(gr, er), j
    assert np.abs(W[0][0]).max() > 0 and np.abs(W[-1][1]).max() > 0
    # HRIR sets on one geometry: the flag reaches the scheduler (the same filters: the geometry stages run once, on plan 0's grids)
    W2 = mex(1, "jobs", jobs[:11], 32, 4, True)[0]
    dev = max(np.abs(a[e] - b[e]).max() / np.abs(a[e]).max() for a, b in zip(W[:11], W2) for e in range(2))
    assert dev < 1e-11, dev
    with pytest.raises(mex.Error, match="eMagLS:native.*len too short"):     # the library's message, forwarded
        mex(1, "jobs", [dict(jobs[0], len=16)])
    # the same list over "two" devices of this process (the one GPU listed twice): emagls_jobs_run_devices behind the sixth argument
    W3 = mex(1, "jobs", jobs, 32, 4, False, np.array([0.0, 0.0]))[0]
    dev3 = max(np.abs(a[e] - b[e]).max() / np.abs(a[e]).max() for a, b in zip(W, W3) for e in range(2))
    assert dev3 < 5e-7, dev3   # (the radius jobs run in padded lane batches there)
    with pytest.raises(mex.Error, match="eMagLS:native.*no such device"):
        mex(1, "jobs", jobs[:2], 32, 4, False, np.array([0.0, 99.0]))
    L.check(L.load().emagls_cache_clear())
