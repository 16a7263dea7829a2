// Shared device/host helpers for the eMagLS HIP library (gfx950 / MI355X only).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <mutex>
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
        (void)hipGetLastError();   // clear the sticky error: a later KERNEL_CHECK must not report this failure as its own
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
// ---- batches: one launch covers `n` designs of identical shape whose buffers sit `stride` bytes apart
// (grid.z enumerates the designs; every kernel offsets its pointer arguments by blockIdx.z * stride)
struct BatchCtx { int n = 1; size_t stride = 0; };
BatchCtx& batch_ctx();  // thread-local launch context (sh_basis.hip); {1, 0} outside emagls_batch_execute
struct BatchScope {
    BatchCtx saved;
    BatchScope(int n, size_t stride) : saved(batch_ctx()) { batch_ctx() = BatchCtx{n, stride}; }
    ~BatchScope() { batch_ctx() = saved; }
};

#ifdef __HIPCC__
inline dim3 bgrid(dim3 g) { g.z = (unsigned)batch_ctx().n; return g; }
// EMAGLS_XCD_RUNS=0: the kernels that take an XCD-aware tile order (xcd_run_index) keep the dispatch order
inline int xcd_runs_enabled() { static const int on = [] { const char* e = getenv("EMAGLS_XCD_RUNS"); return (e && e[0] == '0') ? 0 : 1; }(); return on; }
// (byte arithmetic on the pointer itself: a round trip through an integer hides the address space from the compiler and
// every access through the result becomes a flat_* instruction, which also counts on lgkmcnt and so couples with LDS waits)
template <typename T> __device__ __forceinline__ T* boffz(T* p, size_t stride, unsigned z) {
    typedef typename std::conditional<std::is_const<T>::value, const char, char>::type byte_t;
    return p ? reinterpret_cast<T*>(reinterpret_cast<byte_t*>(p) + (size_t)z * stride) : p;
}
template <typename T> __device__ __forceinline__ T* boff(T* p, size_t stride) { return boffz(p, stride, blockIdx.z); }
// XCD-aware workgroup -> (lane, tile) for kernels whose tiles of ONE lane share operands (round 6).  Workgroups go to the eight XCDs in
// turn in dispatch order (x fastest, then y, then z), so the plain mapping deals the tiles of a lane over all eight L2s and every tile
// fetches the shared operands itself.  Here XCD x takes a contiguous run of the lane-major (lane, tile) list: the tiles of a lane start
// together on ONE XCD, walk their operands at the same pace, and what one tile fetched is an L2 hit for the others.  tile = x + gridDim.x * y
// of the remapped workgroup.  (A bijection of the grid for any lane count: XCD x owns ceil((W - x) / 8) list items.)
__device__ __forceinline__ void xcd_run_index(unsigned& tile, unsigned& lane) {
    const unsigned nt = gridDim.x * gridDim.y, W = nt * gridDim.z, L = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned x = L & 7u, slot = L >> 3, q = W >> 3, r = W & 7u;
    const unsigned g = x * q + (x < r ? x : r) + slot;
    lane = g / nt;
    tile = g - lane * nt;
}
// the integer round trip (flat_* accesses): kept for the factor kernels, whose QR kernel the compiler schedules worse
// with global_* accesses under its 128-VGPR cap (805 -> 901 us per 8-design launch)
template <typename T> __device__ __forceinline__ T* boff_flat(T* p, size_t stride, unsigned z) {
    return p ? reinterpret_cast<T*>(reinterpret_cast<uintptr_t>(p) + (size_t)z * stride) : p;
}
// streaming (write-once, read by a LATER kernel from HBM anyway) stores: nontemporal, so that a result of hundreds of MB does
// not evict what other kernels keep in L2 / MALL.  EMAGLS_NT_STORES is defined by the build; -DEMAGLS_NT_STORES=0 restores plain stores.
#ifndef EMAGLS_NT_STORES
#define EMAGLS_NT_STORES 1
#endif
typedef double nt_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void stream_store(double* p, double v) {
#if EMAGLS_NT_STORES
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
__device__ __forceinline__ void stream_store(cplx* p, cplx v) {
#if EMAGLS_NT_STORES
    nt_d2 w = {v.x, v.y};
    __builtin_nontemporal_store(w, reinterpret_cast<nt_d2*>(p));
#else
    *p = v;
#endif
}
__device__ __forceinline__ double shfl_xor_d(double v, int m) { return __shfl_xor(v, m, 64); }
__device__ __forceinline__ cplx shfl_xor_c(cplx v, int m) { return {__shfl_xor(v.x, m, 64), __shfl_xor(v.y, m, 64)}; }
__device__ __forceinline__ double shfl_d(double v, int src) { return __shfl(v, src, 64); }

// ---- fast FP64 reciprocal / reciprocal square root: hardware seed + Newton steps (~1 ulp), far
// cheaper than the IEEE division / sqrt expansions on the latency-bound paths (Jacobi, sweep)
__device__ __forceinline__ double fast_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
#pragma unroll
    for (int i = 0; i < 2; ++i) y = fma(fma(-x, y, 1.0), y, y);
    return y;
}
__device__ __forceinline__ double fast_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const double e = fma(-x * y, y, 1.0);        // 1 - x y^2
        y = fma(y * e, fma(0.375, e, 0.5), y);       // y (1 + e/2 + 3 e^2/8)
    }
    return y;
}

// ---- DPP lane exchange (no LDS crossbar round trip): v from another lane of the same 16-lane row
template <int CTRL> __device__ __forceinline__ double dpp_d(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
constexpr int DPP_XOR1 = 0xB1;         // quad_perm [1,0,3,2]
constexpr int DPP_XOR2 = 0x4E;         // quad_perm [2,3,0,1]
constexpr int DPP_HALF_MIRROR = 0x141; // lane i <-> 7-i within each 8 lanes
constexpr int DPP_ROW_MIRROR = 0x140;  // lane i <-> 15-i within each 16 lanes

// all-reduce (sum) over aligned groups of W consecutive lanes, W power of two <= 64.
// Steps inside a 16-lane row use DPP; the 32/64-lane steps use ds_bpermute.
template <int W> __device__ __forceinline__ double group_sum(double v) {
    if (W >= 2) v += dpp_d<DPP_XOR1>(v);
    if (W >= 4) v += dpp_d<DPP_XOR2>(v);
    if (W >= 8) v += dpp_d<DPP_HALF_MIRROR>(v);   // both quads already hold their sums
    if (W >= 16) v += dpp_d<DPP_ROW_MIRROR>(v);
    if (W >= 32) v += shfl_xor_d(v, 16);
    if (W >= 64) v += shfl_xor_d(v, 32);
    return v;
}
// full-wave all-reduce: DPP inside the four 16-lane rows, then four scalar lane reads (no LDS crossbar)
__device__ __forceinline__ double wave_sum(double v) {
    v += dpp_d<DPP_XOR1>(v);
    v += dpp_d<DPP_XOR2>(v);
    v += dpp_d<DPP_HALF_MIRROR>(v);
    v += dpp_d<DPP_ROW_MIRROR>(v);
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const double r0 = __hiloint2double(__builtin_amdgcn_readlane(hi, 0), __builtin_amdgcn_readlane(lo, 0));
    const double r1 = __hiloint2double(__builtin_amdgcn_readlane(hi, 16), __builtin_amdgcn_readlane(lo, 16));
    const double r2 = __hiloint2double(__builtin_amdgcn_readlane(hi, 32), __builtin_amdgcn_readlane(lo, 32));
    const double r3 = __hiloint2double(__builtin_amdgcn_readlane(hi, 48), __builtin_amdgcn_readlane(lo, 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ cplx wave_sum(cplx v) { return {wave_sum(v.x), wave_sum(v.y)}; }

template <int W> __device__ __forceinline__ cplx group_sum(cplx v) {
    return {group_sum<W>(v.x), group_sum<W>(v.y)};
}
template <int W> __device__ __forceinline__ double group_max(double v) {
#pragma unroll
    for (int m = W / 2; m >= 1; m >>= 1) v = fmax(v, shfl_xor_d(v, m));
    return v;
}
#endif

// first() is true once per device (of the calling thread's current device): hipFuncSetAttribute and friends are per-device
// state, a process-wide `static bool` would leave every device but the first without them
struct PerDeviceOnce {
    std::mutex mu;
    uint64_t seen[4] = {0, 0, 0, 0};
    bool first() {
        int dev = 0;
        HIP_CHECK(hipGetDevice(&dev));
        std::lock_guard<std::mutex> lk(mu);
        uint64_t& w = seen[(dev >> 6) & 3];
        const uint64_t bit = 1ull << (dev & 63);
        if (w & bit) return false;
        w |= bit;
        return true;
    }
};
// run a scope on a given device and restore the caller's current device afterwards (plans and batches remember theirs)
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev) {
        if (dev < 0) return;
        HIP_CHECK(hipGetDevice(&prev));
        if (prev != dev) { HIP_CHECK(hipSetDevice(dev)); switched = true; }
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

constexpr double kPi = 3.14159265358979323846;

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace emagls
