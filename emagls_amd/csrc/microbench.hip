// Measured FP64 peaks of the device this library runs on: the denominators of the roofline figures that bench.py
// reports (the MI355X FP64 peak is not in the local guides; measure it with a v_mfma_f64_16x16x4_f64
// loop on every CU and use the measured figure).
#include "kernels.hpp"

namespace emagls {

typedef double double4_t __attribute__((ext_vector_type(4)));

// every wave: `iters` rounds of NACC independent accumulator tiles (NACC x 2048 flop per round)
template <int NACC>
__global__ void __launch_bounds__(256) mfma_f64_peak_kernel(int iters, double* __restrict__ sink) {
    const long long c0 = clock64(), w0 = wall_clock64();
    double4_t acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = double4_t{0.0, 0.0, 0.0, 0.0};
    const double a = 1.0 + 1e-9 * (double)(threadIdx.x & 63), b = 1.0 - 1e-9 * (double)(threadIdx.x & 15);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) sink[0] = s;   // keeps the loop alive; never true
    if (blockIdx.x == 0 && threadIdx.x == 0) {   // shader cycles and 100 MHz wall ticks of this wave: the clock the loop ran at
        sink[1] = (double)(clock64() - c0);
        sink[2] = (double)(wall_clock64() - w0);
    }
}

// the same loop on v_mfma_f64_4x4x4_4b (four independent 4 x 4 x 4 blocks per instruction, 512 flop): on gfx950 THIS shape reaches
// the nominal rate of the FP64 matrix pipe -- 75 TFLOP/s of 78.6 -- while the 16 x 16 x 4 shape sustains 49 (62 %) whatever the number
// of waves, accumulators or operand registers (tools/experiments/mfma_peak.hip, round 5)
template <int NACC>
__global__ void __launch_bounds__(256) mfma_f64_4x4_peak_kernel(int iters, double* __restrict__ sink) {
    const long long c0 = clock64(), w0 = wall_clock64();
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0;
    const double a = 1.0 + 1e-9 * (double)(threadIdx.x & 63), b = 1.0 - 1e-9 * (double)(threadIdx.x & 15);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    if (s == 12345.678) sink[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sink[1] = (double)(clock64() - c0);
        sink[2] = (double)(wall_clock64() - w0);
    }
}

// every lane: `iters` rounds of 16 independent FMA chains (16 x 2 flop per lane and round)
__global__ void __launch_bounds__(256) fma_f64_peak_kernel(int iters, double* __restrict__ sink) {
    const long long c0 = clock64(), w0 = wall_clock64();
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
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sink[1] = (double)(clock64() - c0);
        sink[2] = (double)(wall_clock64() - w0);
    }
}

// returns the best of `reps` timings in TFLOP/s; which = 0: MFMA (v_mfma_f64_16x16x4_f64), 1: vector FMA (v_fma_f64), 2: MFMA
// (v_mfma_f64_4x4x4_4b_f64).
// burst: a launch of <= 1 ms (the chip has no time to settle at its sustained power state) instead of ~10 ms; *mhz (optional)
// receives the shader clock the measured loop ran at (s_memtime cycles over the 100 MHz wall counter, one wave).
double measure_fp64_peak(int which, int reps, bool burst, double* mhz) {
    int dev = 0, cus = 0;
    HIP_CHECK(hipGetDevice(&dev));
    HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    double* sink = nullptr;
    HIP_CHECK(hipMalloc(&sink, 64));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_CHECK(hipEventCreate(&e0));
    HIP_CHECK(hipEventCreate(&e1));
    // shapes tried (the best counts): workgroups of 4 waves per CU x independent accumulator tiles per wave (MFMA), 8 per CU (FMA)
    const int shapes[][2] = {{8, 8}, {4, 8}, {8, 4}, {4, 16}, {2, 16}, {8, 16}};
    const int nshapes = which == 0 ? 6 : 1;
    if (which == 2) {   // 16 accumulators per wave, 4 waves per CU upwards: one shape (the rate does not depend on it)
        const int blocks = cus * 4, iters = 65536 / (burst ? 16 : 1);
        const double flop = (double)blocks * 4 * iters * 16 * 512.0;
        double best4 = 0.0;
        for (int r = 0; r < reps + 1; ++r) {
            HIP_CHECK(hipEventRecord(e0, nullptr));
            mfma_f64_4x4_peak_kernel<16><<<blocks, 256, 0, nullptr>>>(iters, sink);
            HIP_CHECK(hipEventRecord(e1, nullptr));
            HIP_CHECK(hipEventSynchronize(e1));
            float ms = 0.f;
            HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0 && ms > 0.f && flop / (ms * 1e-3) / 1e12 > best4) {
                best4 = flop / (ms * 1e-3) / 1e12;
                if (mhz) {
                    double h[3] = {0, 0, 0};
                    HIP_CHECK(hipMemcpy(h, sink, sizeof h, hipMemcpyDeviceToHost));
                    *mhz = h[2] > 0 ? h[1] / (h[2] * 0.01) : 0.0;
                }
            }
        }
        hipEventDestroy(e0);
        hipEventDestroy(e1);
        hipFree(sink);
        return best4;
    }
    double best = 0.0;
    for (int sh = 0; sh < nshapes; ++sh) {
        const int blocks = cus * shapes[sh][0], nacc = shapes[sh][1];
        const int iters = (which == 0 ? 4096 * 8 / nacc * 8 / shapes[sh][0] : 16384) / (burst ? 16 : 1);
        const double flop = which == 0 ? (double)blocks * 4 * iters * nacc * 2048.0 : (double)blocks * 256 * iters * 16 * 2.0;
        for (int r = 0; r < reps + 1; ++r) {
            HIP_CHECK(hipEventRecord(e0, nullptr));
            if (which == 0) {
                if (nacc == 4) mfma_f64_peak_kernel<4><<<blocks, 256, 0, nullptr>>>(iters, sink);
                else if (nacc == 8) mfma_f64_peak_kernel<8><<<blocks, 256, 0, nullptr>>>(iters, sink);
                else mfma_f64_peak_kernel<16><<<blocks, 256, 0, nullptr>>>(iters, sink);
            } else {
                fma_f64_peak_kernel<<<blocks, 256, 0, nullptr>>>(iters, sink);
            }
            HIP_CHECK(hipEventRecord(e1, nullptr));
            HIP_CHECK(hipEventSynchronize(e1));
            float ms = 0.f;
            HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (r > 0 && ms > 0.f) {   // (first launch: module load)
                const double tf = flop / (ms * 1e-3) / 1e12;
                if (tf > best) {
                    best = tf;
                    if (mhz) {
                        double h[3] = {0, 0, 0};
                        HIP_CHECK(hipMemcpy(h, sink, sizeof h, hipMemcpyDeviceToHost));
                        *mhz = h[2] > 0 ? h[1] / (h[2] * 0.01) : 0.0;
                    }
                }
            }
        }
    }
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    hipFree(sink);
    return best;
}

}  // namespace emagls
