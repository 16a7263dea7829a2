// MEX gateway for the MI355X eMagLS library (include/emagls.h).  One gateway, dispatched on a
// command string, so that the MATLAB wrappers in this directory keep the reference's file names and
// signatures (lib/getLsFilters.m:1-2, lib/getMagLsFilters.m:1-2, lib/getEMagLsFilters.m:1-2,
// lib/getEMagLs2Filters.m:1-2, lib/getEMagLsFiltersFromAtf.m:1, dependencies/binauralDecode.m:1-2).
//
// Build on a machine that has MATLAB (R2018a+, interleaved complex) and ROCm:
//     mex -R2018a emagls_mex.cpp -I../include -L../emagls_amd/lib -lemagls
// This container has neither mex.h nor MATLAB, so the file is compiled nowhere here; it is a thin
// adapter: argument checks, pointer hand-over, mxArray allocation, error forwarding.
#include <cstring>
#include <string>

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

}  // namespace

// emagls_mex('ls',      hL, hR, azi, zen, order, shDefinition)
// emagls_mex('magls',   hL, hR, azi, zen, order, fs, len, shDefinition)
// emagls_mex('emagls',  hL, hR, azi, zen, micRadius, micAzi, micZen, order, fs, len, shDefinition)
// emagls_mex('emagls2', ... same ...)
// emagls_mex('fromatf', hL, hR, hrirGridAziZen, atfIrs, atfGridAziZen, fs, filterLen, fTrans)
// emagls_mex('decode',  in, wL, wR, compensateDelay)
void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
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
        int rc = emagls_binaural_decode(dbl(prhs[1], "in"), n, ch, dbl(prhs[2], "wL"), dbl(prhs[3], "wR"), len, comp,
                                        mxGetDoubles(plhs[0]));
        if (rc) fail(rc);
        return;
    }
    if (nrhs < 6) mexErrMsgIdAndTxt("eMagLS:arg", "not enough input arguments");
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
    } else if (c == "emainch") {   // getEMagLsFiltersEMAinCH(hL, hR, azi, zen, micRadius, micGridAziRad, order, fs, len, shDefinition)
        if (nrhs < 10) mexErrMsgIdAndTxt("eMagLS:arg", "not enough input arguments");
        const double r = mxGetScalar(prhs[5]);
        const mwSize nmics = mxGetNumberOfElements(prhs[6]);
        const int order = (int)mxGetScalar(prhs[7]);
        const double fs = mxGetScalar(prhs[8]);
        const mwSize len = (mwSize)mxGetScalar(prhs[9]);
        const int basis = basis_of(nrhs > 10 ? prhs[10] : nullptr);
        const mwSize C = (mwSize)(2 * order + 1);
        plhs[0] = out_matrix(len, C, basis);
        plhs[1] = out_matrix(len, C, basis);
        rc = emagls_get_emagls_filters_ema_in_ch(hL, hR, nsamp, ndirs, dbl(prhs[3], "azi"), dbl(prhs[4], "zen"), r,
                                                 dbl(prhs[6], "micAzi"), nmics, order, fs, len, basis, out_ptr(plhs[0]),
                                                 out_ptr(plhs[1]));
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
