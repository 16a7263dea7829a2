// FP64 matrix-pipe peak of the device: loops of independent v_mfma_f64 tiles in the shapes a kernel can issue them.
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int NACC, bool DISTINCT>
__global__ void __launch_bounds__(256) k16(int iters, double* sink) {
    double4_t acc[NACC];
    double a[NACC], b[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) { acc[i] = double4_t{0, 0, 0, 0}; a[i] = 1.0 + 1e-9 * (threadIdx.x + i); b[i] = 1.0 - 1e-9 * (threadIdx.x + 2 * i); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(DISTINCT ? a[i] : a[0], DISTINCT ? b[i] : b[0], acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) sink[0] = s;
}
template <int NACC>
__global__ void __launch_bounds__(256) k4(int iters, double* sink) {
    double acc[NACC];
    double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
    if (s == 12345.678) sink[0] = s;
}

template <typename F> double timeit(F launch, double flop) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    double best = 0;
    for (int r = 0; r < 4; ++r) {
        hipEventRecord(e0, nullptr);
        launch();
        hipEventRecord(e1, nullptr);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (r > 0 && flop / (ms * 1e-3) / 1e12 > best) best = flop / (ms * 1e-3) / 1e12;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return best;
}

int main() {
    int cus = 0;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    double* sink = nullptr;
    CK(hipMalloc(&sink, 64));
    const int iters = 20000;
    for (int wgs = 1; wgs <= 8; wgs *= 2) {   // workgroups of 4 waves per CU: 1, 2, 4, 8 waves per SIMD
        const int blocks = cus * wgs;
        const double f16 = (double)blocks * 4 * iters * 2048.0, f4 = (double)blocks * 4 * iters * 512.0;
        printf("%d waves/SIMD: 16x16x4 same operands  x4 %6.2f  x8 %6.2f   distinct operands x4 %6.2f x8 %6.2f   4x4x4_4b x4 %6.2f x8 %6.2f x16 %6.2f TFLOP/s\n", wgs,
               timeit([&] { k16<4, false><<<blocks, 256>>>(iters, sink); }, f16 * 4), timeit([&] { k16<8, false><<<blocks, 256>>>(iters, sink); }, f16 * 8),
               timeit([&] { k16<4, true><<<blocks, 256>>>(iters, sink); }, f16 * 4), timeit([&] { k16<8, true><<<blocks, 256>>>(iters, sink); }, f16 * 8),
               timeit([&] { k4<4><<<blocks, 256>>>(iters, sink); }, f4 * 4), timeit([&] { k4<8><<<blocks, 256>>>(iters, sink); }, f4 * 8),
               timeit([&] { k4<16><<<blocks, 256>>>(iters, sink); }, f4 * 16));
    }
    CK(hipFree(sink));
    return 0;
}
