// operand layout of v_mfma_f64_4x4x4_4b_f64 found by experiment: A one-hot in lane s, B[l] = l + 1
//   -> the lanes of D that become non-zero hold D_b[i_s][j], their values name the lane that holds B_b[k_s][j]
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(double* out) {
    const int l = threadIdx.x;
    for (int s = 0; s < 64; ++s) {
        const double a = l == s ? 1.0 : 0.0, b = (double)(l + 1);
        out[s * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
    }
}
int main() {
    double* d; hipMalloc(&d, 64 * 64 * 8);
    probe<<<1, 64>>>(d);
    static double h[64 * 64];
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int s = 0; s < 64; ++s) {
        printf("A lane %2d ->", s);
        for (int l = 0; l < 64; ++l) if (h[s * 64 + l] != 0.0) printf("  D lane %2d = B lane %2d", l, (int)h[s * 64 + l] - 1);
        printf("\n");
    }
    return 0;
}
