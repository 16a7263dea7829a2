// Shared device/host helpers for the eMagLS HIP library (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>

namespace emagls {

// ---------------------------------------------------------------------------------------------
// error handling: every HIP failure becomes a C++ exception; the C ABI turns it into a status code
// ---------------------------------------------------------------------------------------------
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

inline void hip_check(hipError_t e, const char* what, const char* file, int line) {
    if (e != hipSuccess) {
        char buf[512];
        snprintf(buf, sizeof buf, "HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
        throw Error(3, buf);
    }
}
#define HIP_CHECK(x) ::emagls::hip_check((x), #x, __FILE__, __LINE__)
#define KERNEL_CHECK() ::emagls::hip_check(hipGetLastError(), "kernel launch", __FILE__, __LINE__)

// ---------------------------------------------------------------------------------------------
// complex<double> as a plain 16-byte struct (layout == MATLAB interleaved complex / numpy complex128)
// ---------------------------------------------------------------------------------------------
struct __attribute__((aligned(16))) cplx {
    double x, y;
};

__host__ __device__ __forceinline__ cplx mk(double r, double i = 0.0) { return cplx{r, i}; }
__host__ __device__ __forceinline__ cplx operator+(cplx a, cplx b) { return {a.x + b.x, a.y + b.y}; }
__host__ __device__ __forceinline__ cplx operator-(cplx a, cplx b) { return {a.x - b.x, a.y - b.y}; }
__host__ __device__ __forceinline__ cplx operator-(cplx a) { return {-a.x, -a.y}; }
__host__ __device__ __forceinline__ cplx operator*(cplx a, cplx b) {
    return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x};
}
__host__ __device__ __forceinline__ cplx operator*(double a, cplx b) { return {a * b.x, a * b.y}; }
__host__ __device__ __forceinline__ cplx operator*(cplx a, double b) { return {a.x * b, a.y * b}; }
__host__ __device__ __forceinline__ cplx& operator+=(cplx& a, cplx b) { a.x += b.x; a.y += b.y; return a; }
__host__ __device__ __forceinline__ cplx& operator-=(cplx& a, cplx b) { a.x -= b.x; a.y -= b.y; return a; }
__host__ __device__ __forceinline__ cplx conj(cplx a) { return {a.x, -a.y}; }
__host__ __device__ __forceinline__ double conj(double a) { return a; }
__host__ __device__ __forceinline__ double norm2(cplx a) { return a.x * a.x + a.y * a.y; }
__host__ __device__ __forceinline__ double norm2(double a) { return a * a; }
__host__ __device__ __forceinline__ double cabs(cplx a) { return hypot(a.x, a.y); }
// acc += a*b  (fused multiply-adds)
__host__ __device__ __forceinline__ void cfma(cplx& acc, cplx a, cplx b) {
    acc.x = fma(a.x, b.x, acc.x); acc.x = fma(-a.y, b.y, acc.x);
    acc.y = fma(a.x, b.y, acc.y); acc.y = fma(a.y, b.x, acc.y);
}
__host__ __device__ __forceinline__ void cfma(cplx& acc, double a, cplx b) {
    acc.x = fma(a, b.x, acc.x); acc.y = fma(a, b.y, acc.y);
}
__host__ __device__ __forceinline__ void cfma(cplx& acc, cplx a, double b) {
    acc.x = fma(a.x, b, acc.x); acc.y = fma(a.y, b, acc.y);
}
__host__ __device__ __forceinline__ void cfma(double& acc, double a, double b) { acc = fma(a, b, acc); }
// acc += conj(a)*b
__host__ __device__ __forceinline__ void cfma_conj(cplx& acc, cplx a, cplx b) {
    acc.x = fma(a.x, b.x, acc.x); acc.x = fma(a.y, b.y, acc.x);
    acc.y = fma(a.x, b.y, acc.y); acc.y = fma(-a.y, b.x, acc.y);
}
__host__ __device__ __forceinline__ void cfma_conj(cplx& acc, double a, cplx b) { cfma(acc, a, b); }
__host__ __device__ __forceinline__ void cfma_conj(double& acc, double a, double b) { acc = fma(a, b, acc); }
__host__ __device__ __forceinline__ cplx cdiv(cplx a, cplx b) {
    // Smith's algorithm (no spurious overflow)
    if (fabs(b.x) >= fabs(b.y)) {
        double r = b.y / b.x, d = b.x + b.y * r;
        return {(a.x + a.y * r) / d, (a.y - a.x * r) / d};
    }
    double r = b.x / b.y, d = b.x * r + b.y;
    return {(a.x * r + a.y) / d, (a.y * r - a.x) / d};
}
__host__ __device__ __forceinline__ cplx to_cplx(double a) { return {a, 0.0}; }
__host__ __device__ __forceinline__ cplx to_cplx(cplx a) { return a; }
template <typename T> __host__ __device__ __forceinline__ T zero_of();
template <> __host__ __device__ __forceinline__ double zero_of<double>() { return 0.0; }
template <> __host__ __device__ __forceinline__ cplx zero_of<cplx>() { return {0.0, 0.0}; }

// ---------------------------------------------------------------------------------------------
// wave64 helpers
// ---------------------------------------------------------------------------------------------
#ifdef __HIPCC__
__device__ __forceinline__ double shfl_xor_d(double v, int m) { return __shfl_xor(v, m, 64); }
__device__ __forceinline__ cplx shfl_xor_c(cplx v, int m) { return {__shfl_xor(v.x, m, 64), __shfl_xor(v.y, m, 64)}; }
__device__ __forceinline__ double shfl_d(double v, int src) { return __shfl(v, src, 64); }

// all-reduce (sum) over aligned groups of W consecutive lanes, W power of two <= 64
template <int W> __device__ __forceinline__ double group_sum(double v) {
#pragma unroll
    for (int m = W / 2; m >= 1; m >>= 1) v += shfl_xor_d(v, m);
    return v;
}
template <int W> __device__ __forceinline__ cplx group_sum(cplx v) {
#pragma unroll
    for (int m = W / 2; m >= 1; m >>= 1) { v.x += shfl_xor_d(v.x, m); v.y += shfl_xor_d(v.y, m); }
    return v;
}
template <int W> __device__ __forceinline__ double group_max(double v) {
#pragma unroll
    for (int m = W / 2; m >= 1; m >>= 1) v = fmax(v, shfl_xor_d(v, m));
    return v;
}
#endif

constexpr double kPi = 3.14159265358979323846;

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace emagls
