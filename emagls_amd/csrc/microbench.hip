// Measured FP64 peaks of the device this library runs on: the denominators of the roofline figures that bench.py
// reports (the MI355X FP64 peak is not in the local guides; measure it with a v_mfma_f64_16x16x4_f64
// loop on every CU and use the measured figure).
#include "kernels.hpp"

namespace emagls {

typedef double double4_t __attribute__((ext_vector_type(4)));

// every wave: `iters` rounds of 8 independent accumulator tiles (8 x 2048 flop per round)
__global__ void __launch_bounds__(256) mfma_f64_peak_kernel(int iters, double* __restrict__ sink) {
    double4_t acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = double4_t{0.0, 0.0, 0.0, 0.0};
    const double a = 1.0 + 1e-9 * (double)(threadIdx.x & 63), b = 1.0 - 1e-9 * (double)(threadIdx.x & 15);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) sink[0] = s;   // keeps the loop alive; never true
}

// every lane: `iters` rounds of 16 independent FMA chains (16 x 2 flop per lane and round)
__global__ void __launch_bounds__(256) fma_f64_peak_kernel(int iters, double* __restrict__ sink) {
    double acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 1e-3 * (double)(i + 1);
    const double a = 1.0 - 1e-12 * (double)(threadIdx.x & 63), b = 1e-13;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = fma(acc[i], a, b);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
    if (s == 12345.678) sink[0] = s;
}

// returns the best of `reps` timings in TFLOP/s; which = 0: MFMA (v_mfma_f64_16x16x4_f64), 1: vector FMA (v_fma_f64)
double measure_fp64_peak(int which, int reps) {
    int dev = 0, cus = 0;
    HIP_CHECK(hipGetDevice(&dev));
    HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    double* sink = nullptr;
    HIP_CHECK(hipMalloc(&sink, 64));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    const int blocks = cus * 8;          // 8 x 4 waves per CU: two waves per SIMD cover the dependent-issue latency
    const int iters = which == 0 ? 4096 : 16384;
    const double flop = which == 0 ? (double)blocks * 4 * iters * 8 * 2048.0 : (double)blocks * 256 * iters * 16 * 2.0;
    double best = 0.0;
    for (int r = 0; r < reps + 1; ++r) {
        HIP_CHECK(hipEventRecord(e0, nullptr));
        if (which == 0) mfma_f64_peak_kernel<<<blocks, 256, 0, nullptr>>>(iters, sink);
        else fma_f64_peak_kernel<<<blocks, 256, 0, nullptr>>>(iters, sink);
        HIP_CHECK(hipEventRecord(e1, nullptr));
        HIP_CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (r > 0 && ms > 0.f) best = std::max(best, flop / (ms * 1e-3) / 1e12);   // (first launch: module load)
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(sink);
    return best;
}

}  // namespace emagls
