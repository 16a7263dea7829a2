// MEX gateway for the MI355X eMagLS library (include/emagls.h).  One gateway, dispatched on a
// command string, so that the MATLAB wrappers in this directory keep the reference's file names and
// signatures (lib/getLsFilters.m:1-2, lib/getMagLsFilters.m:1-2, lib/getEMagLsFilters.m:1-2,
// lib/getEMagLs2Filters.m:1-2, lib/getEMagLsFiltersFromAtf.m:1, lib/getEMagLsFiltersEMAinCH.m:1-2, lib/getMagLsFilters2D.m:1,
// lib/getMagLsSphericalHeadFilter.m:1, lib/getMagLsArrayDiffuseFilter.m:1, dependencies/getRadialFilter.m:1,
// dependencies/applyRadialFilter.m:1, dependencies/binauralDecode.m:1-2).
//
// Build on a machine that has MATLAB (R2018a+, interleaved complex) and ROCm:
//     mex -R2018a emagls_mex.cpp -I../include -L../emagls_amd/lib -lemagls
// The build image has neither mex.h nor MATLAB: there the file is compiled against a test stand-in for mex.h and driven
// by tests/test_mex_gateway.py (tests/mexstub/).  It is a thin adapter: argument checks, pointer hand-over, mxArray
// allocation, error forwarding.
#include <cctype>
#include <cstring>
#include <string>
#include <vector>

#include "emagls.h"
#include "mex.h"

namespace {

void fail(int rc) { mexErrMsgIdAndTxt("eMagLS:native", "%s (code %d)", emagls_last_error(), rc); }

const double* dbl(const mxArray* a, const char* what) {
    if (!mxIsDouble(a) || mxIsComplex(a)) mexErrMsgIdAndTxt("eMagLS:arg", "%s must be a real double array", what);
    return mxGetDoubles(a);
}
int basis_of(const mxArray* a) {
    if (!a || mxIsEmpty(a)) return EMAGLS_BASIS_REAL;  // default 'real' (lib/getEMagLsFilters.m:33)
    char buf[16] = {0};
    mxGetString(a, buf, sizeof buf);
    if (!std::strcmp(buf, "real")) return EMAGLS_BASIS_REAL;
    if (!std::strcmp(buf, "complex")) return EMAGLS_BASIS_COMPLEX;
    mexErrMsgIdAndTxt("eMagLS:arg", "shDefinition must be 'real' or 'complex'");
    return 0;
}
mxArray* out_matrix(mwSize rows, mwSize cols, int basis) {
    return mxCreateDoubleMatrix(rows, cols, basis == EMAGLS_BASIS_COMPLEX ? mxCOMPLEX : mxREAL);
}
void* out_ptr(mxArray* a) { return mxIsComplex(a) ? (void*)mxGetComplexDoubles(a) : (void*)mxGetDoubles(a); }
const void* in_ptr(const mxArray* a) { return mxIsComplex(a) ? (const void*)mxGetComplexDoubles(a) : (const void*)mxGetDoubles(a); }
// a caller-evaluated SH matrix must be real / complex like the basis asked for
const void* basis_matrix(const mxArray* a, int basis, mwSize rows, mwSize cols, const char* what) {
    if (!mxIsDouble(a) || mxGetM(a) != rows || mxGetN(a) != cols) mexErrMsgIdAndTxt("eMagLS:arg", "%s must be a %d x %d double matrix", what, (int)rows, (int)cols);
    if ((basis == EMAGLS_BASIS_COMPLEX) != (bool)mxIsComplex(a)) mexErrMsgIdAndTxt("eMagLS:arg", "%s does not match shDefinition", what);
    return in_ptr(a);
}
int radial_type(const mxArray* a) {
    char buf[16] = {0};
    mxGetString(a, buf, sizeof buf);
    for (char* q = buf; *q; ++q) *q = (char)tolower(*q);     // strcmpi / lower() in getRadialFilter.m:44,56
    if (!std::strcmp(buf, "tikhonov")) return EMAGLS_RADIAL_TIKHONOV;
    if (!std::strcmp(buf, "softlimit")) return EMAGLS_RADIAL_SOFTLIMIT;
    if (!std::strcmp(buf, "full")) return EMAGLS_RADIAL_FULL;
    if (!std::strcmp(buf, "none")) return EMAGLS_RADIAL_NONE;
    mexErrMsgIdAndTxt("eMagLS:arg", "Unkown radialFilter parameter \"%s\".", buf);
    return 0;
}
// the plans the one-shot entry points cache (device buffers, captured graphs) are released when the MEX file is cleared
void at_exit() { emagls_cache_clear(); }

}  // namespace

// emagls_mex('ls',      hL, hR, azi, zen, order, shDefinition)
// emagls_mex('magls',   hL, hR, azi, zen, order, fs, len, shDefinition)
// emagls_mex('emagls',  hL, hR, azi, zen, micRadius, micAzi, micZen, order, fs, len, shDefinition)
// emagls_mex('emagls2', ... same ...)
// emagls_mex('fromatf', hL, hR, hrirGridAziZen, atfIrs, atfGridAziZen, fs, filterLen, fTrans)
// emagls_mex('emainch' | 'emainsh', hL, hR, azi, zen, micRadius, micAzi, order, fs, len, shDefinition)
// emagls_mex('magls_dc' | 'emagls_dc' | 'emagls2_dc', <the arguments of 'magls' / 'emagls' / 'emagls2'>, applyDiffusenessConst)
// emagls_mex('decode',  in, wL, wR, compensateDelay)            real or complex in / filters; [out, imagAbsSum] = ...
// emagls_mex('sets', kind, hL, hR, azi, zen, micRadius, micAzi, micZen, order, fs, len, shDefinition)   3-D hL / hR: a loop over HRIR sets in one call
// emagls_mex('fromatfsets', hL, hR, hrirGridAziZen, atfIrs, atfGridAziZen, fs, filterLen, fTrans)      3-D hL / hR: the subjects of one ATF set
// emagls_mex('jobs', jobs[, batchSize, inFlight, shareGeometry, devices])   struct array of independent designs (any kinds, radii, HRIR sets): W = {wL, wR} per job
// caller-evaluated shFunction handles (the wrappers evaluate them at emagls_mex('simorder', kind, order, fs, micRadius)):
// emagls_mex('ls_y', hL, hR, Yhrir, order, shDefinition)        emagls_mex('magls_y', hL, hR, Yhrir, order, fs, len, shDefinition)
// emagls_mex('emagls_y' | 'emagls2_y', hL, hR, Yhrir, micRadius, Ymic, order, fs, len, shDefinition)
// render side:
// emagls_mex('magls2d', hLHor, hRHor, azi, order, fs, len, chDefinition)
// emagls_mex('radial', order, fs, smaRadius, irLen, oversamplingFactor, radialFilter, regulConst, noiseGainDb)
// emagls_mex('applyradial', inSig, order, fs, smaRadius, irLen, oversamplingFactor, radialFilter, regulConst, noiseGainDb)
// emagls_mex('encode', smaRecording, micAzi, micZen, order, shDefinition)
// emagls_mex('ch', N, aziRad, basisType)        emagls_mex('smair', order, fs, irLen, oversamplingFactor, smaRadius, micAzi, micZen, ...)
// emagls_mex('shf', micRadius, order, fs, len)                  [wShf, W_Shf] = ...
// emagls_mex('adf', micRadius, micAzi, micZen, order, fs, len, shDefinition[, Yhi])
void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
    static bool registered = false;
    if (!registered) { mexAtExit(at_exit); registered = true; }
    if (nrhs < 1 || !mxIsChar(prhs[0])) mexErrMsgIdAndTxt("eMagLS:arg", "first argument must be a command string");
    char cmd[16] = {0};
    mxGetString(prhs[0], cmd, sizeof cmd);
    const std::string c(cmd);
    if (c == "decode") {
        if (nrhs < 4) mexErrMsgIdAndTxt("eMagLS:arg", "decode needs (in, wL, wR[, compensateDelay])");
        const mwSize n = mxGetM(prhs[1]), ch = mxGetN(prhs[1]), len = mxGetM(prhs[2]);
        const int comp = nrhs > 4 && mxIsLogicalScalarTrue(prhs[4]);
        const mwSize nout = comp ? n - (len / 2 > 0 ? len / 2 - 1 : 0) : n;
        plhs[0] = mxCreateDoubleMatrix(nout, 2, mxREAL);
        const bool ic = mxIsComplex(prhs[1]), wc = mxIsComplex(prhs[2]);
        if (wc != (bool)mxIsComplex(prhs[3])) mexErrMsgIdAndTxt("eMagLS:arg", "the two decoding filters must both be real or both complex");
        int rc;
        if (!ic && !wc) {
            rc = emagls_binaural_decode(dbl(prhs[1], "in"), n, ch, dbl(prhs[2], "wL"), dbl(prhs[3], "wR"), len, comp, mxGetDoubles(plhs[0]));
        } else {   // dependencies/binauralDecode.m:39-42,59-64: complex products accumulated, the real part kept
            double imag_sum[2] = {0, 0};
            rc = emagls_binaural_decode_complex(in_ptr(prhs[1]), ic, n, ch, in_ptr(prhs[2]), in_ptr(prhs[3]), wc, len, comp,
                                                mxGetDoubles(plhs[0]), imag_sum);
            if (nlhs > 1) { plhs[1] = mxCreateDoubleMatrix(1, 2, mxREAL); mxGetDoubles(plhs[1])[0] = imag_sum[0]; mxGetDoubles(plhs[1])[1] = imag_sum[1]; }
        }
        if (rc) fail(rc);
        return;
    }
    if (c == "sets") {
        // [wL, wR] = emagls_mex('sets', kind, hL, hR, azi, zen, micRadius, micAzi, micZen, order, fs, len, shDefinition)
        // hL, hR [numSamples x numDirections x numSets]; zen / micAzi / micZen may be [] where the single call has no such argument.
        // The loop over HRIR sets around lib/getLsFilters.m:30, getMagLsFilters.m:30, getMagLsFilters2D.m:1, getEMagLsFilters.m:32,
        // getEMagLs2Filters.m:32, getEMagLsFiltersEMAinCH.m:32 in one call (emagls_design_hrir_sets).
        if (nrhs < 13) mexErrMsgIdAndTxt("eMagLS:arg", "sets needs (kind, hL, hR, azi, zen, micRadius, micAzi, micZen, order, fs, len, shDefinition)");
        char kb[16] = {0};
        mxGetString(prhs[1], kb, sizeof kb);
        const std::string ks(kb);
        int kind;
        if (ks == "ls") kind = EMAGLS_KIND_LS; else if (ks == "magls") kind = EMAGLS_KIND_MAGLS; else if (ks == "magls2d") kind = EMAGLS_KIND_MAGLS_2D;
        else if (ks == "emagls") kind = EMAGLS_KIND_EMAGLS; else if (ks == "emagls2") kind = EMAGLS_KIND_EMAGLS2;
        else if (ks == "emainch") kind = EMAGLS_KIND_EMA_CH; else if (ks == "emainsh") kind = EMAGLS_KIND_EMA_SH;
        else { mexErrMsgIdAndTxt("eMagLS:arg", "unknown design kind '%s'", kb); return; }
        const double* hL = dbl(prhs[2], "hL");
        const double* hR = dbl(prhs[3], "hR");
        const mwSize nd = mxGetNumberOfDimensions(prhs[2]);
        const mwSize* hd = mxGetDimensions(prhs[2]);
        const mwSize nsamp = hd[0], ndirs = hd[1], nsets = nd > 2 ? hd[2] : 1;
        if (mxGetNumberOfElements(prhs[3]) != mxGetNumberOfElements(prhs[2])) mexErrMsgIdAndTxt("eMagLS:arg", "hL and hR must have the same size");
        auto opt = [&](int i, const char* what) -> const double* { return mxIsEmpty(prhs[i]) ? nullptr : dbl(prhs[i], what); };
        const double* azi = dbl(prhs[4], "hrirGridAziRad");
        const double* zen = opt(5, "hrirGridZenRad");
        const double r = mxIsEmpty(prhs[6]) ? 0.0 : mxGetScalar(prhs[6]);
        const double* mazi = opt(7, "micGridAziRad");
        const double* mzen = opt(8, "micGridZenRad");
        const mwSize nmics = mazi ? mxGetNumberOfElements(prhs[7]) : 0;
        const int order = (int)mxGetScalar(prhs[9]);
        const double fs = mxIsEmpty(prhs[10]) ? 48000.0 : mxGetScalar(prhs[10]);
        const mwSize len = kind == EMAGLS_KIND_LS ? nsamp : (mwSize)mxGetScalar(prhs[11]);
        const int basis = basis_of(prhs[12]);
        mwSize C;
        switch (kind) {
            case EMAGLS_KIND_MAGLS_2D: case EMAGLS_KIND_EMA_CH: C = (mwSize)(2 * order + 1); break;
            case EMAGLS_KIND_EMAGLS2: C = nmics; break;
            default: C = (mwSize)((order + 1) * (order + 1));
        }
        const mwSize od[3] = {len, C, nsets};
        plhs[0] = mxCreateNumericArray(3, od, mxDOUBLE_CLASS, basis == EMAGLS_BASIS_COMPLEX ? mxCOMPLEX : mxREAL);
        mxArray* wR = mxCreateNumericArray(3, od, mxDOUBLE_CLASS, basis == EMAGLS_BASIS_COMPLEX ? mxCOMPLEX : mxREAL);
        const int rc = emagls_design_hrir_sets(kind, hL, hR, nsamp, ndirs, nsets, azi, zen, r, mazi, mzen, nmics, order, fs, len, basis,
                                               out_ptr(plhs[0]), out_ptr(wR));
        if (nlhs > 1) plhs[1] = wR; else mxDestroyArray(wR);
        if (rc) fail(rc);
        return;
    }
    if (c == "jobs") {
        // W = emagls_mex('jobs', jobs, batchSize, inFlight, shareGeometry, devices)
        // The loop a user of the reference writes around one of its design functions (testEMagLs.m:75-95: array radii; HRIR sets;
        // testEMagLsFromAtfs.m:72-73: subjects) handed over in ONE call: the library's scheduler (emagls_jobs_run) cuts the list into
        // chunks of one shape, runs each as a lane batch and keeps several chunks in flight.  `jobs` is a struct array, one element per
        // design, with the reference's argument names as fields:
        //   kind ('ls' | 'magls' | 'magls2d' | 'emagls' | 'emagls2' | 'emainch' | 'emainsh' | 'fromatf'), hL, hR, hrirGridAziRad,
        //   hrirGridZenRad, order, fs, len, shDefinition, micRadius, micGridAziRad, micGridZenRad (array designs),
        //   atfIrs, atfGridAziRad, atfGridZenRad, fTrans (fromatf), applyDiffusenessConst, simOrderPad (optional)
        // Returns an n x 2 cell array {wL, wR} with the filters each single call would return.
        if (nrhs < 2 || !mxIsStruct(prhs[1])) mexErrMsgIdAndTxt("eMagLS:arg", "jobs needs a struct array of designs");
        const mxArray* J = prhs[1];
        const mwSize n = mxGetNumberOfElements(J);
        auto num = [&](int i, double dflt) { return nrhs > i && !mxIsEmpty(prhs[i]) ? mxGetScalar(prhs[i]) : dflt; };
        const int batch_size = (int)num(2, 0), in_flight = (int)num(3, 0);
        const int share = nrhs > 4 && (mxIsLogicalScalarTrue(prhs[4]) || (!mxIsEmpty(prhs[4]) && mxGetScalar(prhs[4]) != 0));
        plhs[0] = mxCreateCellMatrix(n, 2);
        std::vector<emagls_job> jobs(n);
        for (mwSize i = 0; i < n; ++i) {
            auto fld = [&](const char* name) -> const mxArray* { const mxArray* a = mxGetField(J, i, name); return (a && !mxIsEmpty(a)) ? a : nullptr; };
            auto req = [&](const char* name) -> const mxArray* {
                const mxArray* a = fld(name);
                if (!a) mexErrMsgIdAndTxt("eMagLS:arg", "jobs(%d).%s is missing", (int)i + 1, name);
                return a;
            };
            auto vec = [&](const char* name, mwSize want) -> const double* {
                const mxArray* a = fld(name);
                if (!a) return nullptr;
                if (mxGetNumberOfElements(a) != want) mexErrMsgIdAndTxt("eMagLS:arg", "jobs(%d).%s must have %d elements", (int)i + 1, name, (int)want);
                return dbl(a, name);
            };
            char kb[16] = {0};
            mxGetString(req("kind"), kb, sizeof kb);
            const std::string ks(kb);
            emagls_job& jb = jobs[i];
            std::memset(&jb, 0, sizeof jb);
            emagls_design_desc& d = jb.desc;
            if (ks == "ls") d.kind = EMAGLS_KIND_LS; else if (ks == "magls") d.kind = EMAGLS_KIND_MAGLS; else if (ks == "magls2d") d.kind = EMAGLS_KIND_MAGLS_2D;
            else if (ks == "emagls") d.kind = EMAGLS_KIND_EMAGLS; else if (ks == "emagls2") d.kind = EMAGLS_KIND_EMAGLS2;
            else if (ks == "emainch") d.kind = EMAGLS_KIND_EMA_CH; else if (ks == "emainsh") d.kind = EMAGLS_KIND_EMA_SH;
            else if (ks == "fromatf") d.kind = EMAGLS_KIND_FROM_ATF;
            else mexErrMsgIdAndTxt("eMagLS:arg", "jobs(%d).kind: unknown design kind '%s'", (int)i + 1, kb);
            const mxArray* hL = req("hL");
            const mxArray* hR = req("hR");
            if (mxGetM(hL) != mxGetM(hR) || mxGetN(hL) != mxGetN(hR)) mexErrMsgIdAndTxt("eMagLS:arg", "jobs(%d): hL and hR must have the same size", (int)i + 1);
            d.nsamp = (int64_t)mxGetM(hL); d.ndirs = (int64_t)mxGetN(hL);
            jb.hL = dbl(hL, "hL"); jb.hR = dbl(hR, "hR");
            d.basis = basis_of(fld("shDefinition"));
            const bool atf = d.kind == EMAGLS_KIND_FROM_ATF;
            d.order = atf ? 0 : (int)mxGetScalar(req("order"));
            d.fs = fld("fs") ? mxGetScalar(fld("fs")) : 48000.0;
            d.len = d.kind == EMAGLS_KIND_LS ? d.nsamp : (int64_t)mxGetScalar(req("len"));
            jb.hrir_azi = vec("hrirGridAziRad", (mwSize)d.ndirs);
            jb.hrir_zen = vec("hrirGridZenRad", (mwSize)d.ndirs);
            if (!jb.hrir_azi) mexErrMsgIdAndTxt("eMagLS:arg", "jobs(%d).hrirGridAziRad is missing", (int)i + 1);
            const bool arr = d.kind == EMAGLS_KIND_EMAGLS || d.kind == EMAGLS_KIND_EMAGLS2 || d.kind == EMAGLS_KIND_EMA_CH || d.kind == EMAGLS_KIND_EMA_SH;
            if (arr) {
                d.mic_radius = mxGetScalar(req("micRadius"));
                d.nmics = (int64_t)mxGetNumberOfElements(req("micGridAziRad"));
                jb.mic_azi = vec("micGridAziRad", (mwSize)d.nmics);
                jb.mic_zen = vec("micGridZenRad", (mwSize)d.nmics);
            }
            if (atf) {
                const mxArray* a = req("atfIrs");                      // [taps x mics x dirs] (lib/getEMagLsFiltersFromAtf.m:11)
                const mwSize* ad = mxGetDimensions(a);
                d.atf_taps = (int64_t)ad[0]; d.nmics = (int64_t)ad[1]; d.natf = mxGetNumberOfDimensions(a) > 2 ? (int64_t)ad[2] : 1;
                jb.atf = dbl(a, "atfIrs");
                jb.atf_azi = vec("atfGridAziRad", (mwSize)d.natf);
                jb.atf_zen = vec("atfGridZenRad", (mwSize)d.natf);
                d.f_trans = mxGetScalar(req("fTrans"));
                d.basis = EMAGLS_BASIS_REAL;
            }
            if (const mxArray* a = fld("applyDiffusenessConst")) d.diffuseness = mxIsLogicalScalarTrue(a) || mxGetScalar(a) != 0;
            if (const mxArray* a = fld("simOrderPad")) d.sim_order_pad = (int)mxGetScalar(a);
            int64_t rows = 0, cols = 0;
            int cplx = 0;
            const int rc = emagls_design_out_shape(&d, &rows, &cols, &cplx);
            if (rc) fail(rc);
            mxArray* wl = mxCreateDoubleMatrix((mwSize)rows, (mwSize)cols, cplx ? mxCOMPLEX : mxREAL);
            mxArray* wr = mxCreateDoubleMatrix((mwSize)rows, (mwSize)cols, cplx ? mxCOMPLEX : mxREAL);
            mxSetCell(plhs[0], i, wl);           // column-major n x 2: wL in column 1, wR in column 2
            mxSetCell(plhs[0], n + i, wr);
            jb.wL = out_ptr(wl); jb.wR = out_ptr(wr);
        }
        // devices (optional, 6th argument): HIP ordinals of the GPUs of this MATLAB process to split the list over -- one host thread per
        // device, every device writes its jobs' filters straight into the output arrays (emagls_jobs_run_devices; no gather)
        std::vector<int> devices;
        if (nrhs > 5 && !mxIsEmpty(prhs[5])) {
            const double* dv = dbl(prhs[5], "devices");
            for (mwSize i = 0; i < mxGetNumberOfElements(prhs[5]); ++i) devices.push_back((int)dv[i]);
        }
        const int fl = share ? EMAGLS_JOBS_SHARE_GEOMETRY : 0;
        const int rc = !n ? 0 : devices.empty() ? emagls_jobs_run(jobs.data(), (int64_t)n, batch_size, in_flight, fl)
                                                 : emagls_jobs_run_devices(jobs.data(), (int64_t)n, devices.data(), (int)devices.size(), batch_size, in_flight, fl);
        if (rc) fail(rc);
        return;
    }
    if (c == "fromatfsets") {
        // [wL, wR] = emagls_mex('fromatfsets', hL, hR, hrirGridAziZenRad, atfIrs, atfGridAziZenRad, fs, filterLen, fTrans)
        // hL, hR [numSamples x numDirections x numSubjects]: lib/getEMagLsFiltersFromAtf.m:1 in a loop over subjects, one call
        if (nrhs < 9) mexErrMsgIdAndTxt("eMagLS:arg", "not enough input arguments");
        const double* hL = dbl(prhs[1], "hL");
        const double* hR = dbl(prhs[2], "hR");
        const mwSize* hd = mxGetDimensions(prhs[1]);
        const mwSize nsamp = hd[0], ndirs = hd[1], nsets = mxGetNumberOfDimensions(prhs[1]) > 2 ? hd[2] : 1;
        if (mxGetNumberOfElements(prhs[2]) != mxGetNumberOfElements(prhs[1])) mexErrMsgIdAndTxt("eMagLS:arg", "hL and hR must have the same size");
        const double* hg = dbl(prhs[3], "hrirGridAziZenRad");
        const mwSize* ad = mxGetDimensions(prhs[4]);
        const mwSize taps = ad[0], mics = ad[1], natf = mxGetNumberOfDimensions(prhs[4]) > 2 ? ad[2] : 1;
        const double* ag = dbl(prhs[5], "atfGridAziZenRad");
        const mwSize len = (mwSize)mxGetScalar(prhs[7]);
        const mwSize od[3] = {len, mics, nsets};
        plhs[0] = mxCreateNumericArray(3, od, mxDOUBLE_CLASS, mxREAL);
        mxArray* wR = mxCreateNumericArray(3, od, mxDOUBLE_CLASS, mxREAL);
        double dev = 0.0;
        const int rc = emagls_from_atf_hrir_sets(hL, hR, nsamp, ndirs, nsets, hg, hg + ndirs, dbl(prhs[4], "atfIrs"), taps, mics, natf, ag, ag + natf,
                                                 mxGetScalar(prhs[6]), len, mxGetScalar(prhs[8]), mxGetDoubles(plhs[0]), mxGetDoubles(wR), &dev);
        if (nlhs > 1) plhs[1] = wR; else mxDestroyArray(wR);
        if (rc) fail(rc);
        mexPrintf("Matching HRTF and ATF grids, average grid deviation: %g deg\n", dev);  // FromAtf.m:96
        return;
    }
    if (c == "simorder") {   // kind: 'emagls' | 'emagls2'
        char kind[16] = {0};
        mxGetString(prhs[1], kind, sizeof kind);
        plhs[0] = mxCreateDoubleScalar(emagls_simulation_order(!std::strcmp(kind, "emagls2") ? EMAGLS_KIND_EMAGLS2 : EMAGLS_KIND_EMAGLS,
                                                               (int)mxGetScalar(prhs[2]), mxGetScalar(prhs[3]), mxGetScalar(prhs[4])));
        return;
    }
    if (c == "radial" || c == "applyradial") {
        const int o = c == "radial" ? 1 : 2;     // index of `order`
        if (nrhs < o + 8) mexErrMsgIdAndTxt("eMagLS:arg", "not enough input arguments");
        const int order = (int)mxGetScalar(prhs[o]);
        const double fs = mxGetScalar(prhs[o + 1]), r = mxGetScalar(prhs[o + 2]);
        const mwSize irLen = (mwSize)mxGetScalar(prhs[o + 3]);
        const int ovs = (int)mxGetScalar(prhs[o + 4]), type = radial_type(prhs[o + 5]);
        const double regul = mxGetScalar(prhs[o + 6]), gain = mxGetScalar(prhs[o + 7]);
        int rc;
        if (c == "radial") {
            plhs[0] = mxCreateDoubleMatrix((irLen * ovs) / 2 + 1, order + 1, mxCOMPLEX);
            rc = emagls_get_radial_filter(order, fs, r, irLen, ovs, type, regul, gain, mxGetComplexDoubles(plhs[0]));
        } else {
            const mwSize n = mxGetM(prhs[1]);
            plhs[0] = mxCreateDoubleMatrix(emagls_apply_radial_filter_rows(n, irLen, ovs), mxGetN(prhs[1]), mxREAL);
            if (mxGetN(prhs[1]) != (mwSize)((order + 1) * (order + 1))) mexErrMsgIdAndTxt("eMagLS:arg", "inSig must have (order+1)^2 columns");
            rc = emagls_apply_radial_filter(dbl(prhs[1], "inSig"), n, order, fs, r, irLen, ovs, type, regul, gain, mxGetDoubles(plhs[0]));
        }
        if (rc) fail(rc);
        return;
    }
    if (c == "ch") {   // getCH(N, aziRad, basisType)
        const int order = (int)mxGetScalar(prhs[1]), basis = basis_of(nrhs > 3 ? prhs[3] : nullptr);
        const mwSize nd = mxGetNumberOfElements(prhs[2]);
        plhs[0] = out_matrix(nd, 2 * order + 1, basis);
        int rc = emagls_ch_basis(order, nd, dbl(prhs[2], "aziRad"), basis, out_ptr(plhs[0]));
        if (rc) fail(rc);
        return;
    }
    if (c == "smair") {   // (order, fs, irLen, oversamplingFactor, smaRadius, micAzi, micZen, shDefinition, returnRawMicSigs, radialFilter, regulConst, noiseGainDb)
        if (nrhs < 13) mexErrMsgIdAndTxt("eMagLS:arg", "not enough input arguments");
        const int order = (int)mxGetScalar(prhs[1]);
        const double fs = mxGetScalar(prhs[2]), r = mxGetScalar(prhs[5]);
        const mwSize irLen = (mwSize)mxGetScalar(prhs[3]), M = mxGetNumberOfElements(prhs[6]);
        const int ovs = (int)mxGetScalar(prhs[4]), basis = basis_of(prhs[8]);
        const int raw = mxIsLogicalScalarTrue(prhs[9]) || mxGetScalar(prhs[9]) != 0;
        const int so = emagls_simulation_order(EMAGLS_KIND_EMAGLS, order, fs, r);
        const mwSize dims[3] = {raw ? M : (mwSize)((order + 1) * (order + 1)), (mwSize)((so + 1) * (so + 1)), (irLen * ovs) / 2 + 1};
        plhs[0] = mxCreateNumericArray(3, dims, mxDOUBLE_CLASS, mxCOMPLEX);
        int rc = emagls_get_smair_matrix(order, fs, irLen, ovs, r, dbl(prhs[6], "micAzi"), dbl(prhs[7], "micZen"), M, basis, raw,
                                         radial_type(prhs[10]), mxGetScalar(prhs[11]), mxGetScalar(prhs[12]), mxGetComplexDoubles(plhs[0]), nullptr);
        if (rc) fail(rc);
        return;
    }
    if (c == "encode") {
        if (nrhs < 5) mexErrMsgIdAndTxt("eMagLS:arg", "not enough input arguments");
        const mwSize n = mxGetM(prhs[1]), M = mxGetN(prhs[1]);
        const int order = (int)mxGetScalar(prhs[4]), basis = basis_of(nrhs > 5 ? prhs[5] : nullptr);
        plhs[0] = out_matrix(n, (order + 1) * (order + 1), basis);
        int rc = emagls_sh_encode(dbl(prhs[1], "smaRecording"), n, M, dbl(prhs[2], "micAzi"), dbl(prhs[3], "micZen"), order, basis, out_ptr(plhs[0]));
        if (rc) fail(rc);
        return;
    }
    if (c == "shf") {
        const mwSize len = (mwSize)mxGetScalar(prhs[4]);
        plhs[0] = mxCreateDoubleMatrix(len, 1, mxREAL);
        mxArray* W = mxCreateDoubleMatrix(emagls_eq_filter_nfft(len), 1, mxREAL);
        int rc = emagls_get_magls_spherical_head_filter(mxGetScalar(prhs[1]), (int)mxGetScalar(prhs[2]), mxGetScalar(prhs[3]), len,
                                                        mxGetDoubles(plhs[0]), mxGetDoubles(W));
        if (nlhs > 1) plhs[1] = W; else mxDestroyArray(W);
        if (rc) fail(rc);
        return;
    }
    if (c == "adf") {
        if (nrhs < 8) mexErrMsgIdAndTxt("eMagLS:arg", "not enough input arguments");
        const mwSize M = mxGetNumberOfElements(prhs[2]), len = (mwSize)mxGetScalar(prhs[6]);
        const int basis = basis_of(prhs[7]);
        plhs[0] = mxCreateDoubleMatrix(len, 1, mxREAL);
        int rc = emagls_get_magls_array_diffuse_filter(mxGetScalar(prhs[1]), dbl(prhs[2], "micAzi"), dbl(prhs[3], "micZen"), M,
                                                       (int)mxGetScalar(prhs[4]), mxGetScalar(prhs[5]), len, basis,
                                                       nrhs > 8 ? in_ptr(prhs[8]) : nullptr, mxGetDoubles(plhs[0]));
        if (rc) fail(rc);
        return;
    }
    if (nrhs < 6) mexErrMsgIdAndTxt("eMagLS:arg", "not enough input arguments");
    if (c == "magls_dc" || c == "emagls_dc" || c == "emagls2_dc") {   // the removed applyDiffusenessConst option (include/emagls.h)
        const bool ml = c == "magls_dc", raw = c == "emagls2_dc";
        const int o = ml ? 5 : 8;                    // index of `order`
        if (nrhs < o + 5) mexErrMsgIdAndTxt("eMagLS:arg", "not enough input arguments");
        const mwSize ns = mxGetM(prhs[1]), nd = mxGetN(prhs[1]);
        const int order = (int)mxGetScalar(prhs[o]);
        const double fs = mxGetScalar(prhs[o + 1]);
        const mwSize len = (mwSize)mxGetScalar(prhs[o + 2]);
        const int basis = basis_of(prhs[o + 3]);
        const int dc = mxIsLogicalScalarTrue(prhs[o + 4]) || mxGetScalar(prhs[o + 4]) != 0;
        const mwSize nmics = ml ? 0 : mxGetNumberOfElements(prhs[6]);
        const mwSize C = raw ? nmics : (mwSize)((order + 1) * (order + 1));
        plhs[0] = out_matrix(len, C, basis);
        plhs[1] = out_matrix(len, C, basis);
        int rc;
        if (ml)
            rc = emagls_get_magls_filters_dc(dbl(prhs[1], "hL"), dbl(prhs[2], "hR"), ns, nd, dbl(prhs[3], "azi"), dbl(prhs[4], "zen"), order, fs,
                                             len, dc, basis, out_ptr(plhs[0]), out_ptr(plhs[1]));
        else
            rc = (raw ? emagls_get_emagls2_filters_dc : emagls_get_emagls_filters_dc)(
                dbl(prhs[1], "hL"), dbl(prhs[2], "hR"), ns, nd, dbl(prhs[3], "azi"), dbl(prhs[4], "zen"), mxGetScalar(prhs[5]),
                dbl(prhs[6], "micAzi"), dbl(prhs[7], "micZen"), nmics, order, fs, len, dc, basis, out_ptr(plhs[0]), out_ptr(plhs[1]));
        if (rc) fail(rc);
        return;
    }
    const double* hL = dbl(prhs[1], "hL");
    const double* hR = dbl(prhs[2], "hR");
    const mwSize nsamp = mxGetM(prhs[1]), ndirs = mxGetN(prhs[1]);
    int rc = 0;
    if (c == "ls") {
        const int order = (int)mxGetScalar(prhs[5]);
        const int basis = basis_of(nrhs > 6 ? prhs[6] : nullptr);
        const mwSize C = (order + 1) * (order + 1);
        plhs[0] = out_matrix(nsamp, C, basis);
        plhs[1] = out_matrix(nsamp, C, basis);
        rc = emagls_get_ls_filters(hL, hR, nsamp, ndirs, dbl(prhs[3], "azi"), dbl(prhs[4], "zen"), order, basis,
                                   out_ptr(plhs[0]), out_ptr(plhs[1]));
    } else if (c == "magls") {
        const int order = (int)mxGetScalar(prhs[5]);
        const double fs = mxGetScalar(prhs[6]);
        const mwSize len = (mwSize)mxGetScalar(prhs[7]);
        const int basis = basis_of(nrhs > 8 ? prhs[8] : nullptr);
        const mwSize C = (order + 1) * (order + 1);
        plhs[0] = out_matrix(len, C, basis);
        plhs[1] = out_matrix(len, C, basis);
        rc = emagls_get_magls_filters(hL, hR, nsamp, ndirs, dbl(prhs[3], "azi"), dbl(prhs[4], "zen"), order, fs, len, basis,
                                      out_ptr(plhs[0]), out_ptr(plhs[1]));
    } else if (c == "emagls" || c == "emagls2") {
        if (nrhs < 11) mexErrMsgIdAndTxt("eMagLS:arg", "not enough input arguments");
        const double r = mxGetScalar(prhs[5]);
        const mwSize nmics = mxGetNumberOfElements(prhs[6]);
        const int order = (int)mxGetScalar(prhs[8]);
        const double fs = mxGetScalar(prhs[9]);
        const mwSize len = (mwSize)mxGetScalar(prhs[10]);
        const int basis = basis_of(nrhs > 11 ? prhs[11] : nullptr);
        const bool raw = c == "emagls2";
        const mwSize C = raw ? nmics : (mwSize)((order + 1) * (order + 1));
        plhs[0] = out_matrix(len, C, basis);
        plhs[1] = out_matrix(len, C, basis);
        rc = (raw ? emagls_get_emagls2_filters : emagls_get_emagls_filters)(
            hL, hR, nsamp, ndirs, dbl(prhs[3], "azi"), dbl(prhs[4], "zen"), r, dbl(prhs[6], "micAzi"), dbl(prhs[7], "micZen"),
            nmics, order, fs, len, basis, out_ptr(plhs[0]), out_ptr(plhs[1]));
    } else if (c == "magls2d") {   // getMagLsFilters2D(hLHor, hRHor, horHrirGridAziRad, order, fs, len, chDefinition)
        if (nrhs < 7) mexErrMsgIdAndTxt("eMagLS:arg", "not enough input arguments");
        const int order = (int)mxGetScalar(prhs[4]);
        const double fs = mxGetScalar(prhs[5]);
        const mwSize len = (mwSize)mxGetScalar(prhs[6]);
        const int basis = basis_of(nrhs > 7 ? prhs[7] : nullptr);
        plhs[0] = out_matrix(len, 2 * order + 1, basis);
        plhs[1] = out_matrix(len, 2 * order + 1, basis);
        rc = emagls_get_magls_filters_2d(hL, hR, nsamp, ndirs, dbl(prhs[3], "azi"), order, fs, len, basis, out_ptr(plhs[0]), out_ptr(plhs[1]));
    } else if (c == "ls_y" || c == "magls_y") {
        const bool ls = c == "ls_y";
        const int order = (int)mxGetScalar(prhs[4]);
        const double fs = ls ? 0.0 : mxGetScalar(prhs[5]);
        const mwSize len = ls ? nsamp : (mwSize)mxGetScalar(prhs[6]);
        const int basis = basis_of(nrhs > (ls ? 5 : 7) ? prhs[ls ? 5 : 7] : nullptr);
        const mwSize C = (order + 1) * (order + 1);
        const void* Y = basis_matrix(prhs[3], basis, ndirs, C, "shFunction(order, hrirGrid)");
        plhs[0] = out_matrix(len, C, basis);
        plhs[1] = out_matrix(len, C, basis);
        rc = ls ? emagls_get_ls_filters_with_basis(hL, hR, nsamp, ndirs, Y, order, basis, out_ptr(plhs[0]), out_ptr(plhs[1]))
                : emagls_get_magls_filters_with_basis(hL, hR, nsamp, ndirs, Y, order, fs, len, basis, out_ptr(plhs[0]), out_ptr(plhs[1]));
    } else if (c == "emagls_y" || c == "emagls2_y") {
        if (nrhs < 10) mexErrMsgIdAndTxt("eMagLS:arg", "not enough input arguments");
        const bool raw = c == "emagls2_y";
        const double r = mxGetScalar(prhs[4]);
        const mwSize nmics = mxGetM(prhs[5]);
        const int order = (int)mxGetScalar(prhs[6]);
        const double fs = mxGetScalar(prhs[7]);
        const mwSize len = (mwSize)mxGetScalar(prhs[8]);
        const int basis = basis_of(nrhs > 9 ? prhs[9] : nullptr);
        const int so = emagls_simulation_order(raw ? EMAGLS_KIND_EMAGLS2 : EMAGLS_KIND_EMAGLS, order, fs, r);
        const mwSize S = (mwSize)(so + 1) * (so + 1), C = raw ? nmics : (mwSize)((order + 1) * (order + 1));
        const void* Yh = basis_matrix(prhs[3], basis, ndirs, S, "shFunction(simulationOrder, hrirGrid)");
        const void* Ym = basis_matrix(prhs[5], basis, nmics, S, "shFunction(simulationOrder, micGrid)");
        plhs[0] = out_matrix(len, C, basis);
        plhs[1] = out_matrix(len, C, basis);
        rc = (raw ? emagls_get_emagls2_filters_with_basis : emagls_get_emagls_filters_with_basis)(
            hL, hR, nsamp, ndirs, Yh, r, Ym, nmics, order, fs, len, basis, out_ptr(plhs[0]), out_ptr(plhs[1]));
    } else if (c == "emainch" || c == "emainsh") {   // getEMagLsFiltersEMAinCH(hL, hR, azi, zen, micRadius, micGridAziRad, order, fs, len, shDefinition)
        if (nrhs < 10) mexErrMsgIdAndTxt("eMagLS:arg", "not enough input arguments");
        const double r = mxGetScalar(prhs[5]);
        const mwSize nmics = mxGetNumberOfElements(prhs[6]);
        const int order = (int)mxGetScalar(prhs[7]);
        const double fs = mxGetScalar(prhs[8]);
        const mwSize len = (mwSize)mxGetScalar(prhs[9]);
        const int basis = basis_of(nrhs > 10 ? prhs[10] : nullptr);
        const bool sh = c == "emainsh";   // getEMagLsFiltersEMAinSH: same arguments, (order+1)^2 spherical-harmonic channels
        const mwSize C = sh ? (mwSize)((order + 1) * (order + 1)) : (mwSize)(2 * order + 1);
        plhs[0] = out_matrix(len, C, basis);
        plhs[1] = out_matrix(len, C, basis);
        rc = (sh ? emagls_get_emagls_filters_ema_in_sh : emagls_get_emagls_filters_ema_in_ch)(
            hL, hR, nsamp, ndirs, dbl(prhs[3], "azi"), dbl(prhs[4], "zen"), r, dbl(prhs[6], "micAzi"), nmics, order, fs, len, basis,
            out_ptr(plhs[0]), out_ptr(plhs[1]));
    } else if (c == "fromatf") {
        if (nrhs < 9) mexErrMsgIdAndTxt("eMagLS:arg", "not enough input arguments");
        const double* hg = dbl(prhs[3], "hrirGridAziZenRad");   // [ndirs x 2], column-major: azi then zen
        const mwSize* ad = mxGetDimensions(prhs[4]);             // [taps x mics x dirs]
        const mwSize taps = ad[0], mics = ad[1], natf = mxGetNumberOfDimensions(prhs[4]) > 2 ? ad[2] : 1;
        const double* ag = dbl(prhs[5], "atfGridAziZenRad");
        const double fs = mxGetScalar(prhs[6]);
        const mwSize len = (mwSize)mxGetScalar(prhs[7]);
        const double ftrans = mxGetScalar(prhs[8]);
        plhs[0] = mxCreateDoubleMatrix(len, mics, mxREAL);
        plhs[1] = mxCreateDoubleMatrix(len, mics, mxREAL);
        double dev = 0.0;
        rc = emagls_get_emagls_filters_from_atf(hL, hR, nsamp, ndirs, hg, hg + ndirs, dbl(prhs[4], "atfIrs"), taps, mics, natf,
                                                ag, ag + natf, fs, len, ftrans, mxGetDoubles(plhs[0]), mxGetDoubles(plhs[1]), &dev);
        if (!rc) mexPrintf("Matching HRTF and ATF grids, average grid deviation: %g deg\n", dev);  // FromAtf.m:96
    } else {
        mexErrMsgIdAndTxt("eMagLS:arg", "unknown command '%s'", cmd);
    }
    if (rc) fail(rc);
}
