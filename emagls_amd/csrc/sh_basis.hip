// Spherical-harmonic basis assembly (replaces polarch getSH; call sites lib/getLsFilters.m:30,
// lib/getMagLsFilters.m:47, lib/getEMagLsFilters.m:68, dependencies/getSMAIRMatrix.m:101).
//
// One thread per direction.  Fully-normalised associated-Legendre three-term recurrence in n for
// fixed m, seeded from the sectoral term, normalisation folded into the recurrence coefficients (no
// factorials, so orders > 85 do not overflow as the reference's factorial form does).  The
// recurrence coefficients are wave-uniform and come from a small table (scalar loads).  Output is
// MATLAB column-major [ (N+1)^2 ][ D ]: consecutive lanes = consecutive directions, so every store
// instruction of a wave writes 512 contiguous bytes (real) / 1 KiB (complex) -- the kernel is an
// HBM-write stream: 8*D*S (real) or 16*D*S (complex) algorithmic bytes, 16*D bytes read.
#include "kernels.hpp"

namespace emagls {

BatchCtx& batch_ctx() {
    static thread_local BatchCtx ctx;
    return ctx;
}

// table layout: A[n*(N+1)+m], B[n*(N+1)+m] for n >= m+2;  C1[m] = sqrt(2m+3);  CM[m] = sqrt((2m+1)/(2m))
__global__ void sh_coeff_kernel(int N, double* __restrict__ tab, size_t bstride) {
    tab = boff(tab, bstride);
    const int stride = (N + 1) * (N + 1);
    double* A = tab;
    double* B = tab + stride;
    double* C1 = tab + 2 * stride;
    double* CM = C1 + (N + 1);
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < stride; idx += gridDim.x * blockDim.x) {
        int n = idx / (N + 1), m = idx % (N + 1);
        double a = 0.0, b = 0.0;
        if (n >= m + 2) {
            double nn = (double)n, mm = (double)m;
            a = sqrt((4.0 * nn * nn - 1.0) / (nn * nn - mm * mm));
            b = sqrt(((nn - 1.0) * (nn - 1.0) - mm * mm) / (4.0 * (nn - 1.0) * (nn - 1.0) - 1.0));
        }
        A[idx] = a;
        B[idx] = b;
        if (n == 0) {
            C1[m] = sqrt(2.0 * m + 3.0);
            CM[m] = (m == 0) ? 0.0 : sqrt((2.0 * m + 1.0) / (2.0 * m));
        }
    }
}

template <bool COMPLEX>
__global__ void __launch_bounds__(256) sh_basis_kernel(int N, int64_t D, const double* __restrict__ azi,
                                                       const double* __restrict__ zen,
                                                       const double* __restrict__ tab, double* __restrict__ Y,
                                                       int64_t ld, size_t bstride) {
    azi = boff(azi, bstride); zen = boff(zen, bstride); tab = boff(tab, bstride); Y = boff(Y, bstride);
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    const int stride = (N + 1) * (N + 1);
    const double* A = tab;
    const double* B = tab + stride;
    const double* C1 = tab + 2 * stride;
    const double* CM = C1 + (N + 1);
    const double phi = azi[d];
    // MATLAB legendre() sees only x = cos(zenith); sin is sqrt(1-x^2) >= 0 (matters at the poles)
    const double x = cos(zen[d]);
    const double s = sqrt(fmax(0.0, 1.0 - x * x));
    const double SQ2 = 1.4142135623730951;
    double pmm = 0.28209479177387814;  // sqrt(1/(4 pi))
    for (int m = 0; m <= N; ++m) {
        if (m > 0) pmm *= CM[m] * s;
        double sn = 0.0, cs = 1.0;
        if (m > 0) sincos((double)m * phi, &sn, &cs);
        const double sign = (m & 1) ? -1.0 : 1.0;
        double p0 = pmm, p1 = 0.0;
        for (int n = m; n <= N; ++n) {
            double p;
            if (n == m) {
                p = p0;
            } else if (n == m + 1) {
                p1 = C1[m] * x * p0;
                p = p1;
            } else {
                p = A[n * (N + 1) + m] * (x * p1 - B[n * (N + 1) + m] * p0);
                p0 = p1;
                p1 = p;
            }
            const int64_t base = (int64_t)n * n + n;
            if (COMPLEX) {
                cplx* Yc = reinterpret_cast<cplx*>(Y);
                if (m == 0) {
                    stream_store(Yc + base * ld + d, mk(p, 0.0));
                } else {
                    stream_store(Yc + (base + m) * ld + d, mk(sign * p * cs, sign * p * sn));  // Condon-Shortley
                    stream_store(Yc + (base - m) * ld + d, mk(p * cs, -p * sn));               // (-1)^m conj(Y_n^m)
                }
            } else {
                if (m == 0) {
                    stream_store(Y + base * ld + d, p);
                } else {
                    stream_store(Y + (base + m) * ld + d, SQ2 * p * cs);
                    stream_store(Y + (base - m) * ld + d, SQ2 * p * sn);
                }
            }
        }
    }
}

// Yt[d][s] = conj(Y[s][d]) (or plain transpose when conj_it == 0), rows d in [D, Dpad) zero-filled.
template <typename T>
__global__ void __launch_bounds__(256) transpose_conj_kernel(const T* __restrict__ Y, int64_t D, int64_t S, int64_t ldY,
                                                             T* __restrict__ Yt, int64_t Dpad, int64_t ldYt,
                                                             int conj_it, size_t bstride) {
    Y = boff(Y, bstride); Yt = boff(Yt, bstride);
    __shared__ T tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const int64_t d0 = (int64_t)blockIdx.x * 32, s0 = (int64_t)blockIdx.y * 32;
    for (int r = ty; r < 32; r += 8) {
        int64_t s = s0 + r, d = d0 + tx;
        T v = zero_of<T>();
        if (s < S && d < D) v = Y[s * ldY + d];
        tile[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        int64_t d = d0 + r, s = s0 + tx;
        if (d < Dpad && s < S) {
            T v = tile[tx][r];
            Yt[d * ldYt + s] = conj_it ? conj(v) : v;
        }
    }
}

// zero fill with a plain kernel (hipMemsetAsync nodes misbehave under graph replay on older runtimes)
__global__ void zero_fill_kernel(unsigned long long* __restrict__ p, int64_t n8, size_t bstride) {
    p = boff(p, bstride);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0ull;
}
void launch_zero(void* p, size_t bytes, hipStream_t st) {
    const int64_t n8 = (int64_t)((bytes + 7) / 8);  // (every buffer is allocated in multiples of 16 bytes)
    if (n8 == 0) return;
    const unsigned grid = (unsigned)std::min<int64_t>(2048, ceil_div(n8, 256));
    zero_fill_kernel<<<bgrid(grid), 256, 0, st>>>((unsigned long long*)p, n8, batch_ctx().stride);
    KERNEL_CHECK();
}

// dst_z = src for z < n lanes, dst_z = src + (z + 1) * stride bytes: a lane batch's plan 0 hands operands that are the same for
// every lane to the other lanes' own slots (16-byte words; every buffer is allocated in multiples of 16 bytes)
__global__ void broadcast_lanes_kernel(const uint4* __restrict__ src, int64_t n16, size_t stride) {
    uint4* dst = reinterpret_cast<uint4*>(reinterpret_cast<char*>(const_cast<uint4*>(src)) + ((size_t)blockIdx.z + 1) * stride);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
void launch_broadcast_lanes(const void* src, size_t bytes, size_t stride, int nlanes, hipStream_t st) {
    const int64_t n16 = (int64_t)((bytes + 15) / 16);
    if (n16 == 0 || nlanes <= 0) return;
    const unsigned gx = (unsigned)std::min<int64_t>(1024, ceil_div(n16, 256));
    broadcast_lanes_kernel<<<dim3(gx, 1, (unsigned)nlanes), 256, 0, st>>>((const uint4*)src, n16, stride);
    KERNEL_CHECK();
}

// a lane batch's results (wL / wR of every lane, at the arena stride) into the caller's device buffers: one launch instead of
// two copies per design (blockIdx.y = design, blockIdx.z = ear)
__global__ void scatter_lanes_kernel(const uint4* __restrict__ srcL, const uint4* __restrict__ srcR, size_t stride, int64_t n16, LanePtrs dst) {
    const int j = blockIdx.y, e = blockIdx.z;
    const uint4* src = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(e ? srcR : srcL) + (size_t)j * stride);
    uint4* d = reinterpret_cast<uint4*>(dst.p[2 * j + e]);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) d[i] = src[i];
}
void launch_scatter_lanes(const void* srcL, const void* srcR, size_t stride, size_t bytes, int n, const LanePtrs& dst, hipStream_t st) {
    const int64_t n16 = (int64_t)(bytes / 16);
    if (n16 == 0 || n <= 0) return;
    const unsigned gx = (unsigned)std::min<int64_t>(64, ceil_div(n16, 256));
    scatter_lanes_kernel<<<dim3(gx, (unsigned)n, 2), 256, 0, st>>>((const uint4*)srcL, (const uint4*)srcR, stride, n16, dst);
    KERNEL_CHECK();
}

// the inputs of a lane batch (hL / hR of every design) from the caller's device buffers into the plans' own: ONE launch instead of two
// copies per design (40 copies of 2.8 MB each took 0.37 ms of a 7.6 ms job list at config 3 -- 0.3 TB/s, the copies' own launch
// gaps -- and everything else waited behind them); blockIdx.y = buffer
__global__ void __launch_bounds__(256) gather_buffers_kernel(LanePtrs src, LanePtrs dst, int64_t n8) {
    const int j = blockIdx.y;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x, i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (((reinterpret_cast<uintptr_t>(src.p[j]) | reinterpret_cast<uintptr_t>(dst.p[j])) & 15) == 0) {
        const uint4* s = reinterpret_cast<const uint4*>(src.p[j]);
        uint4* d = reinterpret_cast<uint4*>(dst.p[j]);
        const int64_t n16 = n8 >> 1;
        for (int64_t i = i0; i < n16; i += stride) d[i] = s[i];
        if ((n8 & 1) && i0 == 0) reinterpret_cast<unsigned long long*>(dst.p[j])[n8 - 1] = reinterpret_cast<const unsigned long long*>(src.p[j])[n8 - 1];
    } else {
        const unsigned long long* s = reinterpret_cast<const unsigned long long*>(src.p[j]);
        unsigned long long* d = reinterpret_cast<unsigned long long*>(dst.p[j]);
        for (int64_t i = i0; i < n8; i += stride) d[i] = s[i];
    }
}
void launch_gather_buffers(const LanePtrs& src, const LanePtrs& dst, int nbuf, size_t bytes, hipStream_t st) {
    const int64_t n8 = (int64_t)(bytes / 8);
    if (n8 == 0 || nbuf <= 0) return;
    const unsigned gx = (unsigned)std::min<int64_t>(96, ceil_div(n8 / 2 + 1, 256));
    gather_buffers_kernel<<<dim3(gx, (unsigned)nbuf), 256, 0, st>>>(src, dst, n8);
    KERNEL_CHECK();
}

// the same for buffers of different sizes (a plan's ~70 buffers on their way into a batch's arena: one launch per plan instead of one
// hipMemcpyAsync per buffer -- 924 calls per chunk of 14 plans were 9 ms of host time); every buffer starts 16-byte aligned and is
// allocated in multiples of 16 bytes
__global__ void __launch_bounds__(256) move_buffers_kernel(BufferMoves m) {
    const int j = blockIdx.y;
    const int64_t n16 = (int64_t)((m.bytes[j] + 15) / 16);
    const uint4* s = reinterpret_cast<const uint4*>(m.src[j]);
    uint4* d = reinterpret_cast<uint4*>(m.dst[j]);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) d[i] = s[i];
}
void launch_move_buffers(const BufferMoves& m, hipStream_t st) {
    if (m.n <= 0) return;
    size_t big = 0;
    for (int j = 0; j < m.n; ++j) big = std::max(big, m.bytes[j]);
    // (four 16-byte words per thread on the largest buffer, up to 2048 workgroups per buffer: with 64 the ~0.3 GB of a plan moved at
    // 0.4 TB/s; the blocks of a small buffer beyond its end return at once)
    const unsigned gx = (unsigned)std::max<int64_t>(1, std::min<int64_t>(2048, ceil_div((int64_t)(big / 16 + 1), 1024)));
    move_buffers_kernel<<<dim3(gx, (unsigned)m.n), 256, 0, st>>>(m);
    KERNEL_CHECK();
}

// *differ = 1 when two device buffers differ in any 8-byte word (a batch of FromAtf subjects checks that its plans really hold
// the same ATF set and grids before it computes the ATF side once for all of them)
__global__ void compare_words_kernel(const unsigned long long* __restrict__ a, const unsigned long long* __restrict__ b, int64_t n8, int* differ) {
    bool d = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (int64_t)gridDim.x * blockDim.x) d = d || a[i] != b[i];
    if (d) atomicExch(differ, 1);
}
void launch_compare_words(const void* a, const void* b, size_t bytes, int* differ, hipStream_t st) {
    const int64_t n8 = (int64_t)(bytes / 8);
    if (n8 == 0) return;
    const unsigned grid = (unsigned)std::min<int64_t>(4096, ceil_div(n8, 256));
    compare_words_kernel<<<grid, 256, 0, st>>>((const unsigned long long*)a, (const unsigned long long*)b, n8, differ);
    KERNEL_CHECK();
}

// circular harmonics of an equatorial array (dependencies/getCH.m:17-28), written as the complex [channel][mic] matrix
// that the pinv factorisation takes:  out[c * ld + m] = C_c(azi_m),  channels [C_0, C_-1, C_1, ..., C_-N, C_N]
// (out_real: the real basis as doubles, the layout the HRIR-side pipeline keeps a real basis in)
__global__ void ch_basis_kernel(int N, int M, const double* __restrict__ azi, int cplx_basis, int out_real, void* __restrict__ out_,
                                int ld, size_t bstride) {
    azi = boff(azi, bstride); out_ = boff(out_, bstride);
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int C = 2 * N + 1;
    if (idx >= C * M) return;
    const int c = idx / M, m = idx % M;
    cplx v = mk(1.0, 0.0);
    if (c > 0) {
        const int nn = (c + 1) / 2;
        const bool neg = (c & 1) != 0;   // c = 2 nn - 1: C_-nn, c = 2 nn: C_+nn
        double sn, cs;
        sincos((double)nn * azi[m], &sn, &cs);
        if (cplx_basis) v = neg ? mk(cs, -sn) : mk(cs, sn);
        else v = mk(1.4142135623730951 * (neg ? sn : cs), 0.0);
    }
    if (out_real) reinterpret_cast<double*>(out_)[(size_t)c * ld + m] = v.x;
    else reinterpret_cast<cplx*>(out_)[(size_t)c * ld + m] = v;
}
void launch_ch_basis(int N, int M, const double* azi, bool cplx_basis, void* out, int ld, hipStream_t st, bool out_real) {
    ch_basis_kernel<<<bgrid(((2 * N + 1) * M + 255) / 256), 256, 0, st>>>(N, M, azi, cplx_basis ? 1 : 0, out_real ? 1 : 0, out, ld,
                                                                          batch_ctx().stride);
    KERNEL_CHECK();
}

// Y_c(n,+m) = (-1)^m (Y_r(n,+m) + i Y_r(n,-m)) / sqrt2,  Y_c(n,-m) = (Y_r(n,+m) - i Y_r(n,-m)) / sqrt2  (sh_basis_kernel above),
// so a row of coefficients transforms the same way: one thread per (row, n, m >= 0), both members of a pair in one thread.
__global__ void sh_rows_to_complex_kernel(cplx* __restrict__ W, int C, int nrows, int order, size_t bstride) {
    W = boff(W, bstride);
    const int S = (order + 1) * (order + 1);
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)nrows * S) return;
    const int r = (int)(idx / S), a = (int)(idx % S);
    int n = 0;
    while ((n + 1) * (n + 1) <= a) ++n;
    const int m = a - n * n - n;
    if (m <= 0) return;   // m = 0 unchanged; the pair is handled by its +m member
    cplx* w = W + (int64_t)r * C;
    const cplx p = w[n * n + n + m], q = w[n * n + n - m];
    const double r2 = 0.70710678118654752440, sg = (m & 1) ? -r2 : r2;
    w[n * n + n + m] = mk(sg * (p.x - q.y), sg * (p.y + q.x));   // (-1)^m (p + i q) / sqrt2
    w[n * n + n - m] = mk(r2 * (p.x + q.y), r2 * (p.y - q.x));   // (p - i q) / sqrt2
}
void launch_sh_rows_to_complex(void* W, int C, int nrows, int order, hipStream_t st) {
    const int64_t n = (int64_t)nrows * (order + 1) * (order + 1);
    if (n <= 0) return;
    sh_rows_to_complex_kernel<<<bgrid((unsigned)ceil_div(n, 256)), 256, 0, st>>>((cplx*)W, C, nrows, order, batch_ctx().stride);
    KERNEL_CHECK();
}

void launch_sh_coeff(int N, double* tab, hipStream_t st) {
    sh_coeff_kernel<<<bgrid(8), 256, 0, st>>>(N, tab, batch_ctx().stride);
    KERNEL_CHECK();
}

void launch_sh_basis(int N, int64_t D, const double* azi, const double* zen, const double* tab, bool cplx_basis,
                     void* Y, int64_t ld, hipStream_t st) {
    if (D <= 0) return;
    dim3 grid((unsigned)ceil_div(D, 256));
    if (cplx_basis)
        sh_basis_kernel<true><<<bgrid(grid), 256, 0, st>>>(N, D, azi, zen, tab, (double*)Y, ld, batch_ctx().stride);
    else
        sh_basis_kernel<false><<<bgrid(grid), 256, 0, st>>>(N, D, azi, zen, tab, (double*)Y, ld, batch_ctx().stride);
    KERNEL_CHECK();
}

void launch_transpose_conj(const void* Y, int64_t D, int64_t S, int64_t ldY, void* Yt, int64_t Dpad, int64_t ldYt,
                           bool is_cplx, bool conj_it, hipStream_t st) {
    dim3 grid((unsigned)ceil_div(Dpad, 32), (unsigned)ceil_div(S, 32));
    if (is_cplx)
        transpose_conj_kernel<cplx><<<bgrid(grid), 256, 0, st>>>((const cplx*)Y, D, S, ldY, (cplx*)Yt, Dpad, ldYt, conj_it, batch_ctx().stride);
    else
        transpose_conj_kernel<double><<<bgrid(grid), 256, 0, st>>>((const double*)Y, D, S, ldY, (double*)Yt, Dpad, ldYt,
                                                           conj_it, batch_ctx().stride);
    KERNEL_CHECK();
}

}  // namespace emagls
