// FP64 vector issue rate of ONE wave per SIMD against several (how much of the 128 flop/clk/CU a single wave reaches):
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/fp64_rate tools/experiments/fp64_rate.hip && gpurun_out/fp64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int ILP, int KIND>
__global__ void __launch_bounds__(256) k(int iters, double* sink) {
    double acc[ILP], p[ILP];
    for (int i = 0; i < ILP; ++i) { acc[i] = 1e-3 * (i + 1); p[i] = 1.0 + 1e-9 * i; }
    const double a = 1.0 - 1e-12 * (threadIdx.x & 63), b = 1e-13, x = 0.3 + 1e-9 * (threadIdx.x & 7);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
            if (KIND == 0) acc[i] = fma(acc[i], a, b);                     // independent fma chains
            else { acc[i] = fma(b, p[i], acc[i]); p[i] = fma(x, p[i], -(a * p[i])); }   // fma + (mul, fma): the synth pattern, 3 ops
        }
    }
    double s = 0;
    for (int i = 0; i < ILP; ++i) s += acc[i] + p[i];
    if (s == 12345.678) sink[0] = s;
}
template <int ILP, int KIND> void run(int cus, int bpc, int wg) {
    double* sink; hipMalloc(&sink, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    float best = 1e9;
    for (int r = 0; r < 4; ++r) {
        hipEventRecord(e0);
        k<ILP, KIND><<<cus * bpc, wg>>>(iters, sink);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (r && ms < best) best = ms;
    }
    const double ops = (double)cus * bpc * (wg / 64) * iters * ILP * (KIND == 0 ? 1 : 3);   // wave instructions
    const double per_simd_clk = ops / (cus * 4.0) / (best * 1e-3 * 2.4e9);   // wave-instr per SIMD per cycle at 2.4 GHz
    printf("kind %d ILP %2d  wg %4d x %d per CU (%4.1f waves/SIMD): %.3f ms  %.1f TFLOP/s-equiv  %.2f cycles per wave-instr per SIMD\n", KIND, ILP, wg, bpc,
           bpc * wg / 256.0, best, ops * 64 * 2 / (best * 1e-3) / 1e12, 1.0 / per_simd_clk);
    hipFree(sink);
}
int main() {
    int cus = 0; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    printf("CUs %d\n", cus);
    run<16, 0>(cus, 1, 256); run<16, 0>(cus, 2, 256); run<16, 0>(cus, 4, 256); run<16, 0>(cus, 8, 256);
    run<4, 0>(cus, 1, 256); run<8, 0>(cus, 1, 256);
    run<4, 1>(cus, 1, 256); run<4, 1>(cus, 2, 256); run<4, 1>(cus, 4, 256); run<8, 1>(cus, 1, 256); run<8, 1>(cus, 2, 256);
    return 0;
}
