// C ABI of the render-side neighbours (include/emagls.h, "render side"): radial filters, SH encoding and the two optional
// equalisation filters.  Host arrays in, host arrays out; every array operation runs in the kernels of render.hip,
// modal.hip, fft.hip (the IR epilogue), factor.hip (pinv) and decode.hip (overlap-save).  No CPU fallback.
#include <vector>

#include "../../include/emagls.h"
#include "kernels.hpp"

using namespace emagls;

namespace {

constexpr double C_SOUND = 343.0;     // dependencies/getRadialFilter.m:31, lib/getMagLsSphericalHeadFilter.m:26
constexpr int NFFT_MAX_LEN = 2048;    // lib/getMagLsSphericalHeadFilter.m:23

// device scratch of one call, freed on every exit path
struct Scratch {
    std::vector<void*> ptrs;
    hipStream_t st = nullptr;
    Scratch() { st = pool_stream_take(); }
    ~Scratch() {
        for (void* p : ptrs) hipFree(p);
        pool_stream_give(st);
    }
    template <typename T = void> T* get(size_t bytes, bool zero = false) {
        void* p = nullptr;
        HIP_CHECK(hipMalloc(&p, bytes ? bytes : 16));
        ptrs.push_back(p);
        if (zero) HIP_CHECK(hipMemsetAsync(p, 0, bytes, st));
        return reinterpret_cast<T*>(p);
    }
    template <typename T> T* put(const T* host, size_t count) {
        T* p = get<T>(sizeof(T) * count);
        HIP_CHECK(hipMemcpyAsync(p, host, sizeof(T) * count, hipMemcpyHostToDevice, st));
        return p;
    }
    void sync() { HIP_CHECK(hipStreamSynchronize(st)); }
};

bool is_pow2(int64_t v) { return v > 0 && (v & (v - 1)) == 0; }

// b_n(kr) of a rigid sphere on the bin grid f = linspace(0, fs/2, P):  [P][nOrd]
cplx* modal_on_bins(Scratch& s, int order, int P, double fs, double radius) {
    cplx* bn = s.get<cplx>(sizeof(cplx) * (size_t)P * (order + 1));
    const double kr_step = 2.0 * kPi * ((fs / 2.0) / (double)(P - 1)) / C_SOUND * radius;
    launch_modal_bn(order, P, nullptr, kr_step, 1.0, bn, order + 1, 1, s.st);
    return bn;
}

void check_radial_args(int order, double fs, double radius, int64_t ir_len, int oversampling, int type, double noise_gain_db) {
    if (order < 0) throw Error(EMAGLS_ERR_ARG, "negative SH order");
    if (!(fs > 0) || !(radius > 0)) throw Error(EMAGLS_ERR_ARG, "fs and smaRadius must be positive");
    if (ir_len < 1 || oversampling < 1) throw Error(EMAGLS_ERR_ARG, "irLen and oversamplingFactor must be positive");
    if (type < EMAGLS_RADIAL_TIKHONOV || type > EMAGLS_RADIAL_NONE) throw Error(EMAGLS_ERR_ARG, "unknown radialFilter");
    if (type == EMAGLS_RADIAL_SOFTLIMIT && !std::isfinite(noise_gain_db)) throw Error(EMAGLS_ERR_ARG, "softlimit needs noiseGainDb");
    if ((ir_len * oversampling) / 2 + 1 < 2) throw Error(EMAGLS_ERR_ARG, "nfft must be at least 2");
}

// the radial filters on the device: [k][n] with NaNs zeroed (for the IR) and/or [n][P] for the caller
void radial_on_device(Scratch& s, int order, double fs, double radius, int64_t nfft, int type, double regul, double noise_gain_db,
                      cplx* out_kn, cplx* out_cm, bool zero_nan = true) {
    const int P = (int)(nfft / 2 + 1);
    cplx* bn = modal_on_bins(s, order, P, fs, radius);
    const double g = type == EMAGLS_RADIAL_SOFTLIMIT ? pow(10.0, noise_gain_db / 20.0) : 1.0;
    launch_radial_filter(bn, order + 1, P, type, regul, g, nfft % 2 == 0, zero_nan, out_kn, out_cm, s.st);
}

}  // namespace

extern "C" {

int emagls_get_radial_filter(int order, double fs, double sma_radius, int64_t ir_len, int oversampling, int filter_type,
                             double regul_const, double noise_gain_db, void* rad) {
    return guarded_call([&] {
        if (!rad) throw Error(EMAGLS_ERR_ARG, "null pointer");
        check_radial_args(order, fs, sma_radius, ir_len, oversampling, filter_type, noise_gain_db);
        const int64_t nfft = ir_len * oversampling;
        const int P = (int)(nfft / 2 + 1);
        Scratch s;
        cplx* d_rad = s.get<cplx>(sizeof(cplx) * (size_t)P * (order + 1));
        radial_on_device(s, order, fs, sma_radius, nfft, filter_type, regul_const, noise_gain_db, nullptr, d_rad);
        HIP_CHECK(hipMemcpyAsync(rad, d_rad, sizeof(cplx) * (size_t)P * (order + 1), hipMemcpyDeviceToHost, s.st));
        s.sync();
    });
}

int64_t emagls_apply_radial_filter_rows(int64_t nsamp, int64_t ir_len, int oversampling) {
    const int64_t nfft = ir_len * oversampling;
    return std::max(nsamp, nfft) - nfft / 2;
}

int emagls_apply_radial_filter(const double* sig, int64_t nsamp, int order, double fs, double sma_radius, int64_t ir_len,
                               int oversampling, int filter_type, double regul_const, double noise_gain_db, double* out) {
    return guarded_call([&] {
        if (!sig || !out) throw Error(EMAGLS_ERR_ARG, "null pointer");
        if (nsamp < 0) throw Error(EMAGLS_ERR_ARG, "invalid shape");
        check_radial_args(order, fs, sma_radius, ir_len, oversampling, filter_type, noise_gain_db);
        const int64_t nfft = ir_len * oversampling;
        if (nfft % 2 || nfft < 8 || nfft > 4096)
            throw Error(EMAGLS_ERR_UNSUPPORTED, "radial-filter FFT length must be even and in [8, 4096] in this build");
        const int P = (int)(nfft / 2 + 1), nOrd = order + 1, C = nOrd * nOrd;
        const int64_t n = std::max(nsamp, nfft);     // applyRadialFilter.m:20-22: shorter signals are zero-padded to nfft
        Scratch s;
        cplx* d_rad = s.get<cplx>(sizeof(cplx) * (size_t)P * nOrd);
        radial_on_device(s, order, fs, sma_radius, nfft, filter_type, regul_const, noise_gain_db, d_rad, nullptr);
        // ir = ifft(mirrored spectrum), delayed by nfft/2, faded over 5 % at either end (applyRadialFilter.m:15-18)
        cplx* tw = s.get<cplx>(sizeof(cplx) * (size_t)nfft);
        double* d_ir = s.get<double>(sizeof(double) * (size_t)nOrd * nfft);
        launch_twiddles((int)nfft, tw, s.st);
        launch_filter_epilogue(d_rad, nOrd, (int)nfft, (int)nfft, tw, nullptr, 0, 0, 0, 0, d_ir, nullptr, s.st, 1, 0.05);
        const double* d_sig = s.put(sig, (size_t)std::max<int64_t>(nsamp, 1) * C);
        double* d_out = s.get<double>(sizeof(double) * (size_t)(n - nfft / 2) * C);
        filter_channels_by_order(d_sig, nsamp, n, C, d_ir, nOrd, nfft, nfft / 2, d_out, s.st);
        HIP_CHECK(hipMemcpyAsync(out, d_out, sizeof(double) * (size_t)(n - nfft / 2) * C, hipMemcpyDeviceToHost, s.st));
        s.sync();
    });
}

int emagls_sh_encode(const double* sig, int64_t nsamp, int64_t nmics, const double* mic_azi, const double* mic_zen, int order,
                     int basis, void* out) {
    return guarded_call([&] {
        if (!sig || !mic_azi || !mic_zen || !out) throw Error(EMAGLS_ERR_ARG, "null pointer");
        if (nsamp < 0 || nmics < 1 || order < 0) throw Error(EMAGLS_ERR_ARG, "invalid shape");
        if (basis != EMAGLS_BASIS_REAL && basis != EMAGLS_BASIS_COMPLEX) throw Error(EMAGLS_ERR_ARG, "shDefinition must be 'real' or 'complex'");
        const bool cb = basis == EMAGLS_BASIS_COMPLEX;
        const int nOut = (order + 1) * (order + 1), M = (int)nmics;
        if (nOut > 32) throw Error(EMAGLS_ERR_UNSUPPORTED, "SH order above 4 is not supported for the encoder in this build");
        if (M < nOut) throw Error(EMAGLS_ERR_UNSUPPORTED, "fewer microphones than SH channels");
        if (nsamp == 0) return;
        const int ldM = (int)(ceil_div(M, 64) * 64);
        Scratch s;
        const double* d_azi = s.put(mic_azi, (size_t)M);
        const double* d_zen = s.put(mic_zen, (size_t)M);
        double* tab = s.get<double>(sizeof(double) * sh_coeff_count(order));
        void* Ycm = s.get(esz(cb) * (size_t)nOut * M);
        cplx* Yc = s.get<cplx>(sizeof(cplx) * (size_t)nOut * ldM, true);
        cplx* Z = s.get<cplx>(sizeof(cplx) * (size_t)nOut * ldM, true);
        cplx* V = s.get<cplx>(sizeof(cplx) * (size_t)nOut * ldM, true);
        double* tau = s.get<double>(sizeof(double) * nOut);
        cplx* R2 = s.get<cplx>(sizeof(cplx) * (size_t)nOut * nOut);
        cplx* Nw = s.get<cplx>(sizeof(cplx) * (size_t)nOut * nOut);
        launch_sh_coeff(order, tab, s.st);
        launch_sh_basis(order, M, d_azi, d_zen, tab, cb, Ycm, M, s.st);
        launch_widen(Ycm, M, cb, Yc, ldM, nOut, M, false, false, s.st);
        // pinv(Y_mic) with MATLAB's tolerance max(size) eps(norm)   (verifyEMagLs.m:235: pinv(E), E = Y_mic.')
        FactorArgs a{};
        a.S = M; a.C = nOut; a.ldS = ldM; a.kb0 = 0; a.P = 2;
        a.Xd = Yc; a.xd_stride = 0;
        a.reg_mode = 1; a.tol_dim = (double)std::max(M, nOut);
        a.Z = Z; a.Vws = V; a.tauw = tau; a.R2w = R2; a.Nw = Nw;
        launch_factor(a, 1, true, s.st);
        const double* d_sig = s.put(sig, (size_t)nsamp * M);
        const size_t out_bytes = esz(cb) * (size_t)nsamp * nOut;
        void* d_out = s.get(out_bytes);
        launch_sh_encode(d_sig, nsamp, M, Z, ldM, nOut, cb, d_out, s.st);
        HIP_CHECK(hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, s.st));
        s.sync();
    });
}

// common part of the two equalisation filters: nfft, the modal coefficients and the two diffuse-field responses
namespace {
struct EqSetup {
    int nfft, P, simOrder;
    cplx* bn;
    double *df_hi, *df_lo;
};
EqSetup eq_setup(Scratch& s, double radius, int order, double fs, int64_t len) {
    if (!(radius > 0) || !(fs > 0) || order < 0 || len < 2) throw Error(EMAGLS_ERR_ARG, "invalid argument");
    if (len % 2) throw Error(EMAGLS_ERR_ARG, "filter length must be even");
    EqSetup e{};
    e.nfft = (int)std::min<int64_t>(NFFT_MAX_LEN, 2 * len);
    if (e.nfft < 8) throw Error(EMAGLS_ERR_UNSUPPORTED, "filter length below 4 is not supported");
    if (len > e.nfft) throw Error(EMAGLS_ERR_ARG, "len exceeds the oversampled FFT length min(2048, 2*len): the reference fails with an index error");
    e.P = e.nfft / 2 + 1;
    e.simOrder = (int)std::ceil(fs * kPi * radius / C_SOUND);     // (no max(order, .) here: SphericalHeadFilter.m:31)
    if (e.simOrder < order)
        throw Error(EMAGLS_ERR_ARG, "order exceeds the simulation order ceil(fs*pi*r/c): the reference fails with an index error (bn_Hi(:, 1:order+1))");
    if (e.simOrder > 95) throw Error(EMAGLS_ERR_UNSUPPORTED, "simulation order above 95 is not supported in this build");
    e.bn = modal_on_bins(s, e.simOrder, e.P, fs, radius);
    e.df_hi = s.get<double>(sizeof(double) * e.P);
    e.df_lo = s.get<double>(sizeof(double) * e.P);
    launch_diffuse_field(e.bn, e.simOrder + 1, order + 1, e.P, e.df_hi, e.df_lo, s.st);
    return e;
}
// W [P] -> linear-phase-like, truncated and faded taps (SphericalHeadFilter.m:51-67)
void eq_taps(Scratch& s, const EqSetup& e, const cplx* W, int64_t len, double* host_out) {
    cplx* tw = s.get<cplx>(sizeof(cplx) * (size_t)e.nfft);
    double* d_w = s.get<double>(sizeof(double) * (size_t)len);
    launch_twiddles(e.nfft, tw, s.st);
    launch_filter_epilogue(W, 1, e.nfft, (int)len, tw, nullptr, 0, 0, 0, 0, d_w, nullptr, s.st, 1, 0.15);
    HIP_CHECK(hipMemcpyAsync(host_out, d_w, sizeof(double) * (size_t)len, hipMemcpyDeviceToHost, s.st));
}
}  // namespace

int emagls_get_magls_spherical_head_filter(double mic_radius, int order, double fs, int64_t len, double* w_shf, double* W_shf) {
    return guarded_call([&] {
        if (!w_shf) throw Error(EMAGLS_ERR_ARG, "null pointer");
        Scratch s;
        EqSetup e = eq_setup(s, mic_radius, order, fs, len);
        cplx* W = s.get<cplx>(sizeof(cplx) * e.P);
        double* Wfull = W_shf ? s.get<double>(sizeof(double) * e.nfft) : nullptr;
        launch_eq_spectrum(e.df_hi, e.df_lo, nullptr, e.P, 0, W, Wfull, s.st);
        eq_taps(s, e, W, len, w_shf);
        if (W_shf) HIP_CHECK(hipMemcpyAsync(W_shf, Wfull, sizeof(double) * e.nfft, hipMemcpyDeviceToHost, s.st));
        s.sync();
    });
}

int64_t emagls_eq_filter_nfft(int64_t len) { return std::min<int64_t>(NFFT_MAX_LEN, 2 * len); }

int emagls_get_magls_array_diffuse_filter(double mic_radius, const double* mic_azi, const double* mic_zen, int64_t nmics, int order,
                                          double fs, int64_t len, int basis, const void* Y_hi, double* w_adf) {
    return guarded_call([&] {
        if (!w_adf || (!Y_hi && (!mic_azi || !mic_zen))) throw Error(EMAGLS_ERR_ARG, "null pointer");
        if (nmics < 1) throw Error(EMAGLS_ERR_ARG, "invalid shape");
        if (basis != EMAGLS_BASIS_REAL && basis != EMAGLS_BASIS_COMPLEX) throw Error(EMAGLS_ERR_ARG, "shDefinition must be 'real' or 'complex'");
        const bool cb = basis == EMAGLS_BASIS_COMPLEX;
        Scratch s;
        EqSetup e = eq_setup(s, mic_radius, order, fs, len);
        const int M = (int)nmics, S = (e.simOrder + 1) * (e.simOrder + 1), nOut = (order + 1) * (order + 1);
        void* Y = nullptr;
        if (Y_hi) {   // a caller-supplied shFunction, evaluated at the simulation order: [nmics x S] column-major
            Y = s.get(esz(cb) * (size_t)S * M);
            HIP_CHECK(hipMemcpyAsync(Y, Y_hi, esz(cb) * (size_t)S * M, hipMemcpyHostToDevice, s.st));
        } else {
            const double* d_azi = s.put(mic_azi, (size_t)M);
            const double* d_zen = s.put(mic_zen, (size_t)M);
            double* tab = s.get<double>(sizeof(double) * sh_coeff_count(e.simOrder));
            Y = s.get(esz(cb) * (size_t)S * M);
            launch_sh_coeff(e.simOrder, tab, s.st);
            launch_sh_basis(e.simOrder, M, d_azi, d_zen, tab, cb, Y, M, s.st);
        }
        double* df_arr = s.get<double>(sizeof(double) * e.P);
        launch_array_diffuse(e.bn, e.simOrder + 1, Y, cb, M, S, M, nOut, e.P, df_arr, s.st);
        cplx* W = s.get<cplx>(sizeof(cplx) * e.P);
        launch_eq_spectrum(e.df_hi, e.df_lo, df_arr, e.P, 1, W, nullptr, s.st);
        eq_taps(s, e, W, len, w_adf);
        s.sync();
    });
}

int emagls_ch_basis(int order, int64_t ndirs, const double* azi, int basis, void* Y) {
    return guarded_call([&] {
        if (!azi || !Y) throw Error(EMAGLS_ERR_ARG, "null pointer");
        if (order < 0 || ndirs < 1) throw Error(EMAGLS_ERR_ARG, "invalid shape");
        if (basis != EMAGLS_BASIS_REAL && basis != EMAGLS_BASIS_COMPLEX) throw Error(EMAGLS_ERR_ARG, "basisType must be 'real' or 'complex'");
        const bool cb = basis == EMAGLS_BASIS_COMPLEX;
        Scratch s;
        const double* d_azi = s.put(azi, (size_t)ndirs);
        const size_t bytes = esz(cb) * (size_t)(2 * order + 1) * ndirs;
        void* d_Y = s.get(bytes);
        launch_ch_basis(order, (int)ndirs, d_azi, cb, d_Y, (int)ndirs, s.st, !cb);   // [channel][direction] == column-major [ndirs x 2N+1]
        HIP_CHECK(hipMemcpyAsync(Y, d_Y, bytes, hipMemcpyDeviceToHost, s.st));
        s.sync();
    });
}

int emagls_get_smair_matrix(int order, double fs, int64_t ir_len, int oversampling, double sma_radius, const double* mic_azi,
                            const double* mic_zen, int64_t nmics, int basis, int return_raw_mic_sigs, int radial_filter_type,
                            double regul_const, double noise_gain_db, void* smair, int* sim_order) {
    return guarded_call([&] {
        if (!mic_azi || !mic_zen || !smair) throw Error(EMAGLS_ERR_ARG, "null pointer");
        if (order < 0 || nmics < 1 || !(fs > 0) || !(sma_radius > 0) || ir_len < 1 || oversampling < 1) throw Error(EMAGLS_ERR_ARG, "invalid argument");
        if (basis != EMAGLS_BASIS_REAL && basis != EMAGLS_BASIS_COMPLEX) throw Error(EMAGLS_ERR_ARG, "shDefinition must be 'real' or 'complex'");
        if (radial_filter_type < EMAGLS_RADIAL_TIKHONOV || radial_filter_type > EMAGLS_RADIAL_NONE) throw Error(EMAGLS_ERR_ARG, "unknown radialFilter");
        const int64_t nfft = ir_len * oversampling;
        if (nfft % 2) throw Error(EMAGLS_ERR_ARG, "nfft must be even");      // getSMAIRMatrix.m:88
        const bool cb = basis == EMAGLS_BASIS_COMPLEX, raw = return_raw_mic_sigs != 0;
        const int P = (int)(nfft / 2 + 1), M = (int)nmics;
        const int simOrder = std::max(order, (int)std::ceil(fs * kPi * sma_radius / C_SOUND));     // :95
        if (sim_order) *sim_order = simOrder;
        const int S = (simOrder + 1) * (simOrder + 1), nOut = (order + 1) * (order + 1), rows = raw ? M : nOut;
        if (!raw && nOut > 32) throw Error(EMAGLS_ERR_UNSUPPORTED, "SH order above 4 is not supported in this build");
        if (!raw && M < nOut) throw Error(EMAGLS_ERR_UNSUPPORTED, "fewer microphones than SH channels");
        if (simOrder > 95) throw Error(EMAGLS_ERR_UNSUPPORTED, "simulation order above 95 is not supported in this build");
        const int ldM = (int)(ceil_div(M, 64) * 64), ldS = (int)(ceil_div(S, 64) * 64);
        Scratch s;
        const double* d_azi = s.put(mic_azi, (size_t)M);
        const double* d_zen = s.put(mic_zen, (size_t)M);
        double* tab = s.get<double>(sizeof(double) * sh_coeff_count(simOrder));
        void* Ycm = s.get(esz(cb) * (size_t)S * M);                       // [S][M]
        void* Yrm = s.get(esz(cb) * (size_t)ldM * ldS, true);             // [M][ldS]
        launch_sh_coeff(simOrder, tab, s.st);
        launch_sh_basis(simOrder, M, d_azi, d_zen, tab, cb, Ycm, M, s.st);
        launch_transpose_conj(Ycm, M, S, M, Yrm, M, ldS, cb, false, s.st);
        const void* E = Yrm;
        if (!raw) {   // E = pinv(Y_Hi(:, 1:numShsOut)) Y_Hi   (:102, :119-121)
            cplx* Yc = s.get<cplx>(sizeof(cplx) * (size_t)nOut * ldM, true);
            cplx* Z = s.get<cplx>(sizeof(cplx) * (size_t)nOut * ldM, true);
            cplx* V = s.get<cplx>(sizeof(cplx) * (size_t)nOut * ldM, true);
            double* tau = s.get<double>(sizeof(double) * nOut);
            cplx* R2 = s.get<cplx>(sizeof(cplx) * (size_t)nOut * nOut);
            cplx* Nw = s.get<cplx>(sizeof(cplx) * (size_t)nOut * nOut);
            launch_widen(Ycm, M, cb, Yc, ldM, nOut, M, false, false, s.st);
            FactorArgs a{};
            a.S = M; a.C = nOut; a.ldS = ldM; a.kb0 = 0; a.P = 2;
            a.Xd = Yc; a.xd_stride = 0;
            a.reg_mode = 1; a.tol_dim = (double)std::max(M, nOut);
            a.Z = Z; a.Vws = V; a.tauw = tau; a.R2w = R2; a.Nw = Nw;
            launch_factor(a, 1, true, s.st);
            void* Em = s.get(esz(cb) * (size_t)nOut * ldS, true);
            launch_small_gemm(Z, ldM, true, Yrm, ldS, cb, Em, ldS, cb, nOut, S, M, s.st);
            E = Em;
        }
        cplx* bn = s.get<cplx>(sizeof(cplx) * (size_t)P * (simOrder + 1));
        const double kr_step = 2.0 * kPi * ((fs / 2.0) / (double)(P - 1)) / C_SOUND * sma_radius;
        launch_modal_bn(simOrder, P, nullptr, kr_step, -1.0, bn, simOrder + 1, 1, s.st);            // bnAll = -sphModalCoeffs(...)  (:107)
        cplx* rad = nullptr;
        if (!raw && radial_filter_type != EMAGLS_RADIAL_NONE) {
            if (radial_filter_type == EMAGLS_RADIAL_SOFTLIMIT && !std::isfinite(noise_gain_db)) throw Error(EMAGLS_ERR_ARG, "softlimit needs noiseGainDb");
            rad = s.get<cplx>(sizeof(cplx) * (size_t)P * (order + 1));
            radial_on_device(s, order, fs, sma_radius, nfft, radial_filter_type, regul_const, noise_gain_db, rad, nullptr, /*zero_nan=*/false);
        }
        const size_t bytes = sizeof(cplx) * (size_t)rows * S * P;
        cplx* d_out = s.get<cplx>(bytes);
        launch_smair(E, cb, ldS, bn, simOrder + 1, rad, order + 1, rows, S, P, d_out, s.st);
        HIP_CHECK(hipMemcpyAsync(smair, d_out, bytes, hipMemcpyDeviceToHost, s.st));
        s.sync();
    });
}

}  // extern "C"
