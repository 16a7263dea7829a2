// binauralDecode core loop (dependencies/binauralDecode.m:33-42): out(:,ear) = sum_c fftfilt(w_ear(:,c), in(:,c)).
// Overlap-save with hipFFT: the C channel spectra of a block are multiplied with the filter spectra and
// accumulated in the frequency domain, so each ear needs ONE inverse transform per block instead of C.
#include <hipfft/hipfft.h>

#include "kernels.hpp"

namespace emagls {

static void fft_check(hipfftResult r, const char* what) {
    if (r != HIPFFT_SUCCESS) {
        char buf[256];
        snprintf(buf, sizeof buf, "hipFFT error %d in %s", (int)r, what);
        throw Error(3, buf);
    }
}

// seg[c][b][i] = in[c*n + b*B - (len-1) + i]   (zero outside [0, n))
__global__ void ols_pack_kernel(const double* __restrict__ in, int64_t n, int C, int64_t nblocks, int Nf, int64_t B,
                                int64_t len, double* __restrict__ seg) {
    const int64_t total = (int64_t)C * nblocks * Nf;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = idx % Nf, cb = idx / Nf, b = cb % nblocks, c = cb / nblocks;
        const int64_t src = b * B - (len - 1) + i;
        seg[idx] = (src >= 0 && src < n) ? in[c * n + src] : 0.0;
    }
}
// wpad[e][c][i] = w_e[c*len + i] for i < len else 0
__global__ void ols_padfilt_kernel(const double* __restrict__ wL, const double* __restrict__ wR, int C, int64_t len, int Nf,
                                   double* __restrict__ wpad) {
    const int64_t total = (int64_t)2 * C * Nf;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = idx % Nf, ec = idx / Nf, c = ec % C, e = ec / C;
        const double* w = e ? wR : wL;
        wpad[idx] = (i < len) ? w[c * len + i] : 0.0;
    }
}
// Yf[e][b][k] = sum_c Xf[c][b][k] Wf[e][c][k]
__global__ void ols_mac_kernel(const cplx* __restrict__ Xf, const cplx* __restrict__ Wf, int C, int64_t nblocks, int Pf,
                               cplx* __restrict__ Yf) {
    const int64_t total = (int64_t)2 * nblocks * Pf;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = idx % Pf, eb = idx / Pf, b = eb % nblocks, e = eb / nblocks;
        cplx acc = mk(0, 0);
        for (int c = 0; c < C; ++c) cfma(acc, Xf[((int64_t)c * nblocks + b) * Pf + k], Wf[((int64_t)e * C + c) * Pf + k]);
        Yf[idx] = acc;
    }
}
// out[e*n + b*B + i] = y[e][b][len-1+i] / Nf
__global__ void ols_unpack_kernel(const double* __restrict__ y, int64_t n, int64_t nblocks, int Nf, int64_t B, int64_t len,
                                  double* __restrict__ out) {
    const int64_t total = (int64_t)2 * n;
    const double scale = 1.0 / (double)Nf;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = idx % n, e = idx / n;
        const int64_t b = t / B, i = t % B;
        out[idx] = y[((int64_t)e * nblocks + b) * Nf + (len - 1) + i] * scale;
    }
}

void binaural_decode_real(const double* sig, int64_t n, int C, const double* wL, const double* wR, int64_t len,
                          double* out, hipStream_t st) {
    if (n <= 0) return;
    int Nf = 1024;
    while (Nf < 4 * len) Nf <<= 1;
    const int64_t B = Nf - (len - 1);
    const int64_t nblocks = ceil_div(n, B);
    const int Pf = Nf / 2 + 1;
    double *seg = nullptr, *wpad = nullptr, *y = nullptr;
    cplx *Xf = nullptr, *Wf = nullptr, *Yf = nullptr;
    hipfftHandle pf = 0, pw = 0, pi = 0;
    auto cleanup = [&]() {
        if (pf) hipfftDestroy(pf);
        if (pw) hipfftDestroy(pw);
        if (pi) hipfftDestroy(pi);
        hipFree(seg); hipFree(wpad); hipFree(y); hipFree(Xf); hipFree(Wf); hipFree(Yf);
    };
    try {
        HIP_CHECK(hipMalloc(&seg, sizeof(double) * C * nblocks * Nf));
        HIP_CHECK(hipMalloc(&wpad, sizeof(double) * 2 * C * Nf));
        HIP_CHECK(hipMalloc(&y, sizeof(double) * 2 * nblocks * Nf));
        HIP_CHECK(hipMalloc(&Xf, sizeof(cplx) * C * nblocks * Pf));
        HIP_CHECK(hipMalloc(&Wf, sizeof(cplx) * 2 * C * Pf));
        HIP_CHECK(hipMalloc(&Yf, sizeof(cplx) * 2 * nblocks * Pf));
        int nn[1] = {Nf};
        fft_check(hipfftPlanMany(&pf, 1, nn, nullptr, 1, Nf, nullptr, 1, Pf, HIPFFT_D2Z, (int)(C * nblocks)), "plan D2Z signal");
        fft_check(hipfftPlanMany(&pw, 1, nn, nullptr, 1, Nf, nullptr, 1, Pf, HIPFFT_D2Z, 2 * C), "plan D2Z filters");
        fft_check(hipfftPlanMany(&pi, 1, nn, nullptr, 1, Pf, nullptr, 1, Nf, HIPFFT_Z2D, (int)(2 * nblocks)), "plan Z2D");
        fft_check(hipfftSetStream(pf, st), "set stream");
        fft_check(hipfftSetStream(pw, st), "set stream");
        fft_check(hipfftSetStream(pi, st), "set stream");
        ols_pack_kernel<<<2048, 256, 0, st>>>(sig, n, C, nblocks, Nf, B, len, seg);
        KERNEL_CHECK();
        ols_padfilt_kernel<<<256, 256, 0, st>>>(wL, wR, C, len, Nf, wpad);
        KERNEL_CHECK();
        fft_check(hipfftExecD2Z(pf, seg, (hipfftDoubleComplex*)Xf), "exec D2Z signal");
        fft_check(hipfftExecD2Z(pw, wpad, (hipfftDoubleComplex*)Wf), "exec D2Z filters");
        ols_mac_kernel<<<2048, 256, 0, st>>>(Xf, Wf, C, nblocks, Pf, Yf);
        KERNEL_CHECK();
        fft_check(hipfftExecZ2D(pi, (hipfftDoubleComplex*)Yf, y), "exec Z2D");
        ols_unpack_kernel<<<2048, 256, 0, st>>>(y, n, nblocks, Nf, B, len, out);
        KERNEL_CHECK();
        HIP_CHECK(hipStreamSynchronize(st));
    } catch (...) {
        cleanup();
        throw;
    }
    cleanup();
}

}  // namespace emagls
