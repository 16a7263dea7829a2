// LS / MagLS above 32 channels (SH orders 5..15, CH orders 16..127: up to 256 channels): a plain path for the shapes the tuned
// kernels (32-channel register tiles, 32-row LDS slabs) do not cover.  lib/getMagLsFilters.m:45-48 takes any order; order 5-7
// decoders are ordinary use, and up to 64 channels the kernels below keep a workgroup's operands in registers; 65..256 channels
// (round 6: gram_inverse_big_*, sweep_wide_loop_kernel) walk them in loops.
//
//   pinv(Y_conj) = Y conj(M),  M = (Y^T conj(Y))^-1 = R^-1 R^-H  from the Cholesky factor R of the Gram matrix -- valid where
//   MATLAB's pinv drops no singular value (tolerance max(size) eps(norm)); the kernel certifies cond(Gy) <= ||Gy||_F ||M||_F
//   < 1e8 (cond(Y) < 1e4: relative error eps cond(Y)^2 < 2e-8) and raises status word 5 otherwise;
//   the sweep (lib/getMagLsFilters.m:64-72) as one launch per bin: a workgroup sums the partial sums of the previous bin,
//   forms p and t = |H| p / |p| for its 64 directions and its partial sums of t pinv(Y_conj).
#include "kernels.hpp"

namespace emagls {

namespace {

constexpr int WD_SMAX = 64;     // the register-staged kernels
constexpr int WD_BIG = 256;     // the loop forms

// M = R^-1 R^-H for upper-triangular R (S x S, row major), as complex [S][S]; status[5] = 1 when the certificate fails
template <typename T>
__global__ void __launch_bounds__(256) gram_inverse_kernel(const T* __restrict__ R, int S, cplx* __restrict__ M, int* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    cplx* Ri = reinterpret_cast<cplx*>(dyn);            // [S][S + 1]  R^-1 (upper triangular)
    __shared__ double red[256];
    const int tid = threadIdx.x, ld = S + 1;
    for (int idx = tid; idx < S * ld; idx += 256) Ri[idx] = mk(0.0, 0.0);
    __syncthreads();
    if (tid < S) {   // column j of the inverse by back substitution:  sum_k R[i][k] X[k][j] = delta_ij
        const int j = tid;
        for (int i = j; i >= 0; --i) {
            cplx acc = mk(i == j ? 1.0 : 0.0, 0.0);
            for (int k = i + 1; k <= j; ++k) { cplx t = mk(0.0, 0.0); cfma(t, to_cplx(R[(size_t)i * S + k]), Ri[k * ld + j]); acc = acc - t; }
            Ri[i * ld + j] = cdiv(acc, to_cplx(R[(size_t)i * S + i]));
        }
    }
    __syncthreads();
    double fm = 0.0, fg = 0.0;
    for (int idx = tid; idx < S * S; idx += 256) {
        const int i = idx / S, j = idx % S;
        cplx acc = mk(0.0, 0.0), g = mk(0.0, 0.0);
        for (int k = (i > j ? i : j); k < S; ++k) cfma(acc, Ri[i * ld + k], conj(Ri[j * ld + k]));
        for (int k = 0; k <= (i < j ? i : j); ++k) cfma_conj(g, to_cplx(R[(size_t)k * S + i]), to_cplx(R[(size_t)k * S + j]));   // Gy = R^H R
        M[idx] = acc;
        fm += norm2(acc); fg += norm2(g);
    }
    red[tid] = fm; __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) { if (tid < s2) red[tid] += red[tid + s2]; __syncthreads(); }
    fm = red[0]; __syncthreads();
    red[tid] = fg; __syncthreads();
    for (int s2 = 128; s2 > 0; s2 >>= 1) { if (tid < s2) red[tid] += red[tid + s2]; __syncthreads(); }
    fg = red[0];
    if (tid == 0 && !(fm * fg < 1e16)) atomicExch(status + 5, 1);   // (||Gy||_F ||M||_F)^2 >= (1e8)^2, or not finite
}

// Ypinv[c][d] = sum_s Ycm[s][d] conj(M[s][c])
template <typename T>
__global__ void __launch_bounds__(256) ypinv_gram_kernel(const T* __restrict__ Ycm, int64_t ldD, const cplx* __restrict__ M, int S, int D,
                                                         T* __restrict__ Ypinv) {
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    if (d >= D) return;
    cplx acc = mk(0.0, 0.0);
    for (int s = 0; s < S; ++s) cfma(acc, to_cplx(Ycm[(int64_t)s * ldD + d]), conj(M[(size_t)s * S + c]));
    if constexpr (sizeof(T) == sizeof(double)) Ypinv[(int64_t)c * ldD + d] = acc.x; else Ypinv[(int64_t)c * ldD + d] = acc;
}

// one bin of the sweep: X = Y_conj [c][ldD], Z = pinv(Y_conj) [c][ldD] (both fixed over the bins), 64 directions per workgroup.
// Every operand of the launch is requested before the first use (the three loops of the first form loaded one element per
// iteration: 22 + 32 + 32 dependent round trips to L2 per bin, 30 us per launch).
template <typename TX>
__global__ void __launch_bounds__(256) sweep_wide_kernel(DenseSweepArgs a, int kb) {
    __shared__ __attribute__((aligned(16))) cplx Wp[2][WD_SMAX];
    __shared__ __attribute__((aligned(16))) cplx ts[2][64];
    const int C = a.C, nWG = a.nWG, tid = threadIdx.x;
    const cplx* Wprev = a.Wpart + (int64_t)((kb - 1) & 1) * nWG * 2 * C;
    cplx* Wout = a.Wpart + (int64_t)(kb & 1) * nWG * 2 * C;
    const TX* X = reinterpret_cast<const TX*>(a.X) + (int64_t)kb * a.x_stride;   // (strides 0: MagLS, one operand pair for every bin)
    const TX* Z = reinterpret_cast<const TX*>(a.Zd) + (int64_t)kb * a.z_stride;
    const int64_t d0 = (int64_t)blockIdx.x * 64, na = a.P - a.kabs0;
    const bool first = kb == a.kfirst;
    constexpr int HC = WD_SMAX / 2;
    // ---- requests: the previous launch's partial sums first (they gate the chain), then this slab's operands
    const int pair1 = tid >> 1, half1 = tid & 1;             // phases 1 and 3: (ear, channel) x 2 lanes
    constexpr int NG = sizeof(TX) == sizeof(double) ? 24 : 12;   // staged partial sums per thread; more workgroups: a loop behind them
    cplx gw[NG];
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        const int w = half1 + 2 * i;
        gw[i] = (!first && pair1 < 2 * C && w < nWG) ? Wprev[(int64_t)pair1 * nWG + w] : mk(0.0, 0.0);
    }
    const int dd2 = tid & 63, e2 = (tid >> 6) & 1, half2 = tid >> 7;   // phase 2: (direction, ear) x 2 lanes over the channels
    const int64_t d2 = d0 + dd2;
    TX xr[HC];
#pragma unroll
    for (int i = 0; i < HC; ++i) {
        const int c = half2 + 2 * i;
        xr[i] = (d2 < a.D && c < C) ? X[(int64_t)c * a.ldD + d2] : zero_of<TX>();
    }
    const double habs = (d2 < a.D && half2 == 0) ? a.Habs[((int64_t)e2 * na + (kb - a.kabs0)) * a.ldH + d2] : 0.0;
    TX zr[32];                                               // phase 3: this thread pair's (ear, channel) row of the slab (2 C <= 128 pairs)
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const int64_t d = d0 + half1 + 2 * j;
        zr[j] = (pair1 < 2 * C && d < a.D) ? Z[(int64_t)(pair1 % C) * a.ldD + d] : zero_of<TX>();
    }
    // ---- W(kb-1,:): the least-squares row for the first swept bin, the sum of the workgroups' partial sums afterwards
    if (pair1 < 2 * C) {
        const int e = pair1 / C, c = pair1 % C;
        cplx acc = mk(0.0, 0.0);
        if (first) { if (half1 == 0) acc = a.W[((int64_t)e * a.P + (kb - 1)) * C + c]; }
        else {
            cplx a0 = mk(0.0, 0.0), a1 = mk(0.0, 0.0);
#pragma unroll
            for (int i = 0; i < NG; i += 2) { a0 += gw[i]; a1 += gw[i + 1]; }
            acc = a0 + a1;
            for (int w = half1 + 2 * NG; w < nWG; w += 2) acc += Wprev[(int64_t)pair1 * nWG + w];
        }
        acc = group_sum<2>(acc);
        if (half1 == 0) {
            Wp[e][c] = acc;
            if (blockIdx.x == 0 && !first) a.W[((int64_t)e * a.P + (kb - 1)) * C + c] = acc;
        }
    }
    __syncthreads();
    // ---- p = W(kb-1,:) Y_conj, t = |H| exp(i angle(p)) (Nyquist: real part)
    {
        cplx p = mk(0.0, 0.0), q = mk(0.0, 0.0);
#pragma unroll
        for (int i = 0; i < HC; i += 2) {
            if (half2 + 2 * i < C) cfma(p, Wp[e2][half2 + 2 * i], to_cplx(xr[i]));
            if (half2 + 2 * i + 2 < C) cfma(q, Wp[e2][half2 + 2 * i + 2], to_cplx(xr[i + 1]));
        }
        p += q;
        if (half2 == 1) ts[e2][dd2] = p;
        __syncthreads();
        cplx t = mk(0.0, 0.0);
        if (half2 == 0) {
            p += ts[e2][dd2];
            if (d2 < a.D) {
                const double a2 = norm2(p);
                t = mk(habs, 0.0);                                // angle(0) = 0
                if (a2 > 0.0) { const double ia = habs / sqrt(a2); t = mk(p.x * ia, p.y * ia); }
                if (kb == a.P - 1) t.y = 0.0;
            }
        }
        __syncthreads();   // (every wave has read the half sums)
        if (half2 == 0) ts[e2][dd2] = t;
    }
    __syncthreads();
    // ---- this workgroup's partial sums of t pinv(Y_conj)
    if (pair1 < 2 * C) {
        const int e = pair1 / C;
        cplx a0 = mk(0.0, 0.0), a1 = mk(0.0, 0.0);
#pragma unroll
        for (int j = 0; j < 32; j += 2) {
            cfma(a0, ts[e][half1 + 2 * j], to_cplx(zr[j]));
            cfma(a1, ts[e][half1 + 2 * j + 2], to_cplx(zr[j + 1]));
        }
        cplx acc = group_sum<2>(a0 + a1);
        if (half1 == 0) Wout[(int64_t)pair1 * nWG + blockIdx.x] = acc;
    }
}


// ---- 65..256 channels.  R^-1 in global memory (S x S complex would be 1 MB of LDS at S = 256): thread j solves column j by back
// substitution (Ri [k][j]: neighbouring threads read neighbouring columns), then M = R^-1 R^-H one row per workgroup with the two
// Frobenius norms of the certificate summed by atomics, then the certificate.
template <typename T>
__global__ void __launch_bounds__(WD_BIG) gram_inverse_big_solve_kernel(const T* __restrict__ R, int S, cplx* __restrict__ Ri, double* __restrict__ nrm) {
    const int j = threadIdx.x;
    if (j < 2) nrm[j] = 0.0;
    if (j >= S) return;
    for (int i = S - 1; i > j; --i) Ri[(size_t)i * S + j] = mk(0.0, 0.0);
    for (int i = j; i >= 0; --i) {
        cplx acc = mk(i == j ? 1.0 : 0.0, 0.0);
        const T* Rrow = R + (size_t)i * S;
        cplx t0 = mk(0.0, 0.0), t1 = mk(0.0, 0.0);
        int k = i + 1;
        for (; k + 1 <= j; k += 2) {
            cfma(t0, to_cplx(Rrow[k]), Ri[(size_t)k * S + j]);
            cfma(t1, to_cplx(Rrow[k + 1]), Ri[(size_t)(k + 1) * S + j]);
        }
        if (k <= j) cfma(t0, to_cplx(Rrow[k]), Ri[(size_t)k * S + j]);
        acc = acc - (t0 + t1);
        Ri[(size_t)i * S + j] = cdiv(acc, to_cplx(Rrow[i]));
    }
}
template <typename T>
__global__ void __launch_bounds__(WD_BIG) gram_inverse_big_rows_kernel(const T* __restrict__ R, const cplx* __restrict__ Ri, int S, cplx* __restrict__ M,
                                                                       double* __restrict__ nrm) {
    __shared__ double red[2][WD_BIG];
    const int i = blockIdx.x, j = threadIdx.x;
    double fm = 0.0, fg = 0.0;
    if (j < S) {
        cplx acc = mk(0.0, 0.0), g = mk(0.0, 0.0);
        for (int k = (i > j ? i : j); k < S; ++k) cfma(acc, Ri[(size_t)i * S + k], conj(Ri[(size_t)j * S + k]));
        for (int k = 0; k <= (i < j ? i : j); ++k) cfma_conj(g, to_cplx(R[(size_t)k * S + i]), to_cplx(R[(size_t)k * S + j]));   // Gy = R^H R
        M[(size_t)i * S + j] = acc;
        fm = norm2(acc); fg = norm2(g);
    }
    red[0][j] = fm; red[1][j] = fg;
    __syncthreads();
    for (int s2 = WD_BIG / 2; s2 > 0; s2 >>= 1) {
        if (j < s2) { red[0][j] += red[0][j + s2]; red[1][j] += red[1][j + s2]; }
        __syncthreads();
    }
    if (j == 0) { atomicAdd(nrm, red[0][0]); atomicAdd(nrm + 1, red[1][0]); }
}
__global__ void gram_inverse_big_check_kernel(const double* __restrict__ nrm, int* __restrict__ status) {
    if (!(nrm[0] * nrm[1] < 1e16)) atomicExch(status + 5, 1);   // (||Gy||_F ||M||_F)^2 >= (1e8)^2, or not finite
}

// one bin of the sweep for 65..256 channels: the phases of sweep_wide_kernel with loops over the channels (64 directions per
// workgroup; 2 C <= 512 (ear, channel) pairs, up to two per thread)
template <typename TX>
__global__ void __launch_bounds__(256) sweep_wide_loop_kernel(DenseSweepArgs a, int kb) {
    __shared__ __attribute__((aligned(16))) cplx Wp[2][WD_BIG];
    __shared__ __attribute__((aligned(16))) cplx ts[2][64];
    __shared__ __attribute__((aligned(16))) cplx ph[2][2][64];
    const int C = a.C, nWG = a.nWG, tid = threadIdx.x;
    const cplx* Wprev = a.Wpart + (int64_t)((kb - 1) & 1) * nWG * 2 * C;
    cplx* Wout = a.Wpart + (int64_t)(kb & 1) * nWG * 2 * C;
    const TX* X = reinterpret_cast<const TX*>(a.X) + (int64_t)kb * a.x_stride;
    const TX* Z = reinterpret_cast<const TX*>(a.Zd) + (int64_t)kb * a.z_stride;
    const int64_t d0 = (int64_t)blockIdx.x * 64, na = a.P - a.kabs0;
    const bool first = kb == a.kfirst;
    // ---- W(kb-1,:): the least-squares row for the first swept bin, the sum of the workgroups' partial sums afterwards
    for (int pair = tid; pair < 2 * C; pair += 256) {
        const int e = pair / C, c = pair % C;
        cplx acc = mk(0.0, 0.0);
        if (first) acc = a.W[((int64_t)e * a.P + (kb - 1)) * C + c];
        else {
            cplx a0 = mk(0.0, 0.0), a1 = mk(0.0, 0.0);
            const cplx* src = Wprev + (int64_t)pair * nWG;
            int w = 0;
            for (; w + 1 < nWG; w += 2) { a0 += src[w]; a1 += src[w + 1]; }
            if (w < nWG) a0 += src[w];
            acc = a0 + a1;
        }
        Wp[e][c] = acc;
        if (blockIdx.x == 0 && !first) a.W[((int64_t)e * a.P + (kb - 1)) * C + c] = acc;
    }
    __syncthreads();
    // ---- p = W(kb-1,:) Y_conj: (direction, ear) x two halves of the channels;  t = |H| exp(i angle(p)) (Nyquist: real part)
    {
        const int dd = tid & 63, e = (tid >> 6) & 1, half = tid >> 7;
        const int64_t d = d0 + dd;
        cplx p0 = mk(0.0, 0.0), p1 = mk(0.0, 0.0);
        if (d < a.D) {
            int c = half;
            for (; c + 2 < C; c += 4) {
                cfma(p0, Wp[e][c], to_cplx(X[(int64_t)c * a.ldD + d]));
                cfma(p1, Wp[e][c + 2], to_cplx(X[(int64_t)(c + 2) * a.ldD + d]));
            }
            if (c < C) cfma(p0, Wp[e][c], to_cplx(X[(int64_t)c * a.ldD + d]));
        }
        ph[half][e][dd] = p0 + p1;
        __syncthreads();
        if (half == 0) {
            const cplx p = ph[0][e][dd] + ph[1][e][dd];
            cplx t = mk(0.0, 0.0);
            if (d < a.D) {
                const double habs = a.Habs[((int64_t)e * na + (kb - a.kabs0)) * a.ldH + d];
                const double a2 = norm2(p);
                t = mk(habs, 0.0);                                // angle(0) = 0
                if (a2 > 0.0) { const double ia = habs / sqrt(a2); t = mk(p.x * ia, p.y * ia); }
                if (kb == a.P - 1) t.y = 0.0;
            }
            ts[e][dd] = t;
        }
    }
    __syncthreads();
    // ---- this workgroup's partial sums of t pinv(Y_conj)
    for (int pair = tid; pair < 2 * C; pair += 256) {
        const int e = pair / C, c = pair % C;
        const TX* z = Z + (int64_t)c * a.ldD + d0;
        const int n = (int)(a.D - d0 < 64 ? a.D - d0 : 64);
        cplx a0 = mk(0.0, 0.0), a1 = mk(0.0, 0.0);
        int j = 0;
        for (; j + 1 < n; j += 2) { cfma(a0, ts[e][j], to_cplx(z[j])); cfma(a1, ts[e][j + 1], to_cplx(z[j + 1])); }
        if (j < n) cfma(a0, ts[e][j], to_cplx(z[j]));
        Wout[(int64_t)pair * nWG + blockIdx.x] = a0 + a1;
    }
}

__global__ void __launch_bounds__(256) sweep_wide_finalize_kernel(const cplx* __restrict__ Wpart, cplx* __restrict__ W, int nWG, int C, int P,
                                                                  int kb_last) {
    const cplx* Wprev = Wpart + (int64_t)(kb_last & 1) * nWG * 2 * C;
    for (int pair = threadIdx.x; pair < 2 * C; pair += 256) {
        cplx acc = mk(0.0, 0.0);
        for (int w = 0; w < nWG; ++w) acc += Wprev[(int64_t)pair * nWG + w];
        W[((int64_t)(pair / C) * P + kb_last) * C + pair % C] = acc;
    }
}

}  // namespace

// work: S x S complex values + two doubles (only read and written for S > 64)
void launch_gram_inverse(const void* R, int S, bool is_cplx, void* M, int* status, hipStream_t st, void* work) {
    if (S > WD_BIG) throw Error(2, "more than 256 channels is not supported");
    if (S > WD_SMAX) {
        if (!work) throw Error(2, "gram inverse: workspace missing");
        cplx* Ri = (cplx*)work;
        double* nrm = reinterpret_cast<double*>(Ri + (size_t)S * S);
        if (is_cplx) {
            gram_inverse_big_solve_kernel<cplx><<<1, WD_BIG, 0, st>>>((const cplx*)R, S, Ri, nrm);
            gram_inverse_big_rows_kernel<cplx><<<S, WD_BIG, 0, st>>>((const cplx*)R, Ri, S, (cplx*)M, nrm);
        } else {
            gram_inverse_big_solve_kernel<double><<<1, WD_BIG, 0, st>>>((const double*)R, S, Ri, nrm);
            gram_inverse_big_rows_kernel<double><<<S, WD_BIG, 0, st>>>((const double*)R, Ri, S, (cplx*)M, nrm);
        }
        gram_inverse_big_check_kernel<<<1, 1, 0, st>>>(nrm, status);
        KERNEL_CHECK();
        return;
    }
    const size_t dyn = sizeof(cplx) * (size_t)S * (S + 1);
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
        HIP_CHECK(hipFuncSetAttribute((const void*)gram_inverse_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
        HIP_CHECK(hipFuncSetAttribute((const void*)gram_inverse_kernel<cplx>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024));
    }
    if (is_cplx) gram_inverse_kernel<cplx><<<1, 256, dyn, st>>>((const cplx*)R, S, (cplx*)M, status);
    else gram_inverse_kernel<double><<<1, 256, dyn, st>>>((const double*)R, S, (cplx*)M, status);
    KERNEL_CHECK();
}
void launch_ypinv_gram(const void* Ycm, int64_t ldD, bool is_cplx, const void* M, int S, int D, void* Ypinv, hipStream_t st) {
    dim3 grid((unsigned)ceil_div(D, 256), S);
    if (is_cplx) ypinv_gram_kernel<cplx><<<grid, 256, 0, st>>>((const cplx*)Ycm, ldD, (const cplx*)M, S, D, (cplx*)Ypinv);
    else ypinv_gram_kernel<double><<<grid, 256, 0, st>>>((const double*)Ycm, ldD, (const cplx*)M, S, D, (double*)Ypinv);
    KERNEL_CHECK();
}
void launch_sweep_wide(const DenseSweepArgs& a, int kb, bool x_cplx, hipStream_t st) {
    if (a.C > WD_BIG) throw Error(2, "wide sweep: more than 256 channels");
    if (a.C > WD_SMAX) {
        if (x_cplx) sweep_wide_loop_kernel<cplx><<<a.nWG, 256, 0, st>>>(a, kb);
        else sweep_wide_loop_kernel<double><<<a.nWG, 256, 0, st>>>(a, kb);
        KERNEL_CHECK();
        return;
    }
    if (x_cplx) sweep_wide_kernel<cplx><<<a.nWG, 256, 0, st>>>(a, kb);
    else sweep_wide_kernel<double><<<a.nWG, 256, 0, st>>>(a, kb);
    KERNEL_CHECK();
}
void launch_sweep_wide_finalize(const void* Wpart, void* W, int nWG, int C, int P, int kb_last, hipStream_t st) {
    sweep_wide_finalize_kernel<<<1, 256, 0, st>>>((const cplx*)Wpart, (cplx*)W, nWG, C, P, kb_last);
    KERNEL_CHECK();
}

}  // namespace emagls
