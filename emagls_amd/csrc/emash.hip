// getEMagLsFiltersEMAinSH (lib/getEMagLsFiltersEMAinSH.m:66-143): eMagLS for an equatorial microphone array with the output in
// spherical harmonics.  The reference builds, per bin and direction d,
//     pwGrid_k(:,d) = Rot_d^T (J pinv(CH(micAzi))) Y_mic diag(b_n(k)) conj(Y_hor(d,:))^T          (:66-100)
// (plane waves from the HORIZONTAL projection of the HRIR grid on the array, circular -> spherical harmonics without radial
// filters, then a rotation of the coefficient row to the direction's elevation).  Frequency enters through b_n only, so
//     G_k = pwGrid_k.' = sum_n b_n(k) QT'_n,     QT'_n(d,:) = ( conj(Y_hor)(d,blk_n) E0(:,blk_n)^T ) Rot_d,   E0 = J pinv(CH) Y_mic
// -- the order terms of the eMagLS pipeline (qt_kernel) with every direction's row rotated once.  pwGrid_k is well conditioned
// at every bin here (no radial terms: cond <= 4e2 for 2..9 cm arrays), so all bins take the Gram route, with the Gram matrices
// formed from G_k itself (the rotation destroys the common S-space the GEMM over order pairs needs).
//
// Rot_d is polarch's getSHrotMtx(euler2rotationMatrix(-azi, zen - pi/2, azi, 'zyz'), N, basis) (:96-98), un-vendored: restated
// from its stated intent (rotate the direction (azi, pi/2) up to (azi, zen)) exactly like the CPU restatement does, as the
// least-squares solution of Y_i(R^-1 x_p) = sum_j Rot_ij Y_j(x_p) on a point set that resolves order N.
#include "kernels.hpp"

namespace emagls {

// E0[c][s] = Nnm[c] Ech[ch(c)][s]: the circular-to-spherical expansion J (dependencies/getChToShExpansionMatrix.m:11-18) applied
// to the rows of pinv(CH) Y_mic.  Nnm[c] = Y_c(phi_c, pi/2) / CH_ch(c)(phi_c) is read off our own SH basis evaluated at one
// azimuth per channel (Ypts[c][c], launch_sh_basis on C points), so that it carries that basis' normalisation and phase.
template <typename T>
__global__ void __launch_bounds__(256) ema_sh_e0_kernel(const T* __restrict__ Ech, int ldS, const T* __restrict__ Ypts, int C, int S,
                                                        int cplx_basis, T* __restrict__ E0, size_t bstride) {
    Ech = boff(Ech, bstride); Ypts = boff(Ypts, bstride); E0 = boff(E0, bstride);
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= C * S) return;
    const int c = idx / S, s = idx - c * S;
    int n = 0;
    while ((n + 1) * (n + 1) <= c) ++n;
    const int m = c - n * n - n;
    const int am = m < 0 ? -m : m;
    const int ch = m == 0 ? 0 : 2 * am - (m < 0 ? 1 : 0);
    const double chval = (cplx_basis || m == 0) ? 1.0 : 1.4142135623730951;
    const T nnm = Ypts[(size_t)c * C + c] * (1.0 / chval);
    E0[(size_t)c * ldS + s] = nnm * Ech[(size_t)ch * ldS + s];
}
void launch_ema_sh_e0(const void* Ech, int ldS, const void* Ypts, int C, int S, bool cb, void* E0, hipStream_t st) {
    const unsigned g = (unsigned)ceil_div((int64_t)C * S, 256);
    if (cb) ema_sh_e0_kernel<cplx><<<bgrid(g), 256, 0, st>>>((const cplx*)Ech, ldS, (const cplx*)Ypts, C, S, 1, (cplx*)E0, batch_ctx().stride);
    else ema_sh_e0_kernel<double><<<bgrid(g), 256, 0, st>>>((const double*)Ech, ldS, (const double*)Ypts, C, S, 0, (double*)E0, batch_ctx().stride);
    KERNEL_CHECK();
}

// Evaluation points of the rotation fit: the fixed set x_p (a Fibonacci spiral of npts points) and, for every direction d,
// R_d^-1 x_p, R_d = the right-hand rotation by the elevation pi/2 - zen_d about the horizontal axis (sin azi, -cos azi, 0).
// Output as (azimuth, zenith) lists: [d * npts + p] for d < D, the unrotated set at d == D.
__global__ void __launch_bounds__(256) rot_points_kernel(const double* __restrict__ azi, const double* __restrict__ zen, int D, int npts,
                                                         double* __restrict__ azr, double* __restrict__ znr, size_t bstride) {
    azi = boff(azi, bstride); zen = boff(zen, bstride); azr = boff(azr, bstride); znr = boff(znr, bstride);
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)(D + 1) * npts) return;
    const int d = (int)(idx / npts), pt = (int)(idx - (int64_t)d * npts);
    const double i = (double)pt + 0.5;
    const double z = acos(1.0 - 2.0 * i / (double)npts);
    const double a = fmod(kPi * (1.0 + sqrt(5.0)) * i, 2.0 * kPi);
    double sz, cz, sa, ca;
    sincos(z, &sz, &cz);
    sincos(a, &sa, &ca);
    double x[3] = {sz * ca, sz * sa, cz};
    if (d < D) {
        const double alpha = kPi / 2.0 - zen[d];
        double sd, cd, sal, cal;
        sincos(azi[d], &sd, &cd);
        sincos(alpha, &sal, &cal);
        const double ax[3] = {sd, -cd, 0.0};
        // R^-1 x = x cos(alpha) - (ax x x) sin(alpha) + ax (ax . x)(1 - cos(alpha))      (Rodrigues, angle -alpha)
        const double cr[3] = {ax[1] * x[2] - ax[2] * x[1], ax[2] * x[0] - ax[0] * x[2], ax[0] * x[1] - ax[1] * x[0]};
        const double dt = ax[0] * x[0] + ax[1] * x[1];
        for (int j = 0; j < 3; ++j) x[j] = x[j] * cal - cr[j] * sal + ax[j] * dt * (1.0 - cal);
    }
    azr[idx] = atan2(x[1], x[0]);
    znr[idx] = acos(fmin(fmax(x[2], -1.0), 1.0));
}
void launch_rot_points(const double* azi, const double* zen, int D, int npts, double* azr, double* znr, hipStream_t st) {
    rot_points_kernel<<<bgrid((unsigned)ceil_div((int64_t)(D + 1) * npts, 256)), 256, 0, st>>>(azi, zen, D, npts, azr, znr, batch_ctx().stride);
    KERNEL_CHECK();
}

// Rot_d[i][j] = sum_p pinvB[j][p] A[i][d npts + p]   (the transpose of pinv(B) A_d), identity where zen_d == pi/2 exactly
// (the reference skips those directions, :92).  A is the SH matrix of all rotated points [C][ldA], Z = pinv(B) as [C][ldP]
// complex (the factorisation's output).  One workgroup per direction.
template <typename T> __device__ __forceinline__ T from_c(cplx v);
template <> __device__ __forceinline__ double from_c<double>(cplx v) { return v.x; }
template <> __device__ __forceinline__ cplx from_c<cplx>(cplx v) { return v; }
template <typename T>
__global__ void __launch_bounds__(256) rot_from_points_kernel(const T* __restrict__ A, int64_t ldA, const cplx* __restrict__ Z, int ldP, int C,
                                                              int npts, const double* __restrict__ zen, T* __restrict__ Rot, size_t bstride) {
    A = boff(A, bstride); Z = boff(Z, bstride); zen = boff(zen, bstride); Rot = boff(Rot, bstride);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* a_s = reinterpret_cast<T*>(smem);   // [C][npts]
    T* z_s = a_s + (size_t)C * npts;       // [C][npts]
    const int d = blockIdx.x;
    T* out = Rot + (size_t)d * C * C;
    if (zen[d] == 1.5707963267948966) {
        for (int idx = threadIdx.x; idx < C * C; idx += blockDim.x) out[idx] = from_c<T>(mk((idx / C) == (idx % C) ? 1.0 : 0.0, 0.0));
        return;
    }
    for (int idx = threadIdx.x; idx < C * npts; idx += blockDim.x) {
        const int c = idx / npts, p = idx - c * npts;
        a_s[idx] = A[(int64_t)c * ldA + (int64_t)d * npts + p];
        z_s[idx] = from_c<T>(Z[(size_t)c * ldP + p]);
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < C * C; idx += blockDim.x) {
        const int i = idx / C, j = idx - i * C;
        T acc = from_c<T>(mk(0.0, 0.0));
        for (int p = 0; p < npts; ++p) cfma(acc, z_s[j * npts + p], a_s[i * npts + p]);
        out[idx] = acc;
    }
}
// The same for more than 32 channels (orders 5..7: the full C x npts operands of a direction no longer fit the LDS): one workgroup per
// (direction, order l), only the (2l+1) x (2l+1) diagonal block of Rot_d -- all that qt_rotate_kernel reads; the rest of Rot stays zero.
template <typename T>
__global__ void __launch_bounds__(256) rot_blocks_from_points_kernel(const T* __restrict__ A, int64_t ldA, const cplx* __restrict__ Z, int ldP, int C,
                                                                     int npts, const double* __restrict__ zen, T* __restrict__ Rot, size_t bstride) {
    A = boff(A, bstride); Z = boff(Z, bstride); zen = boff(zen, bstride); Rot = boff(Rot, bstride);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int d = blockIdx.x, l = blockIdx.y, B = 2 * l + 1, c0 = l * l;
    T* a_s = reinterpret_cast<T*>(smem);   // [B][npts]
    T* z_s = a_s + (size_t)B * npts;       // [B][npts]
    T* out = Rot + (size_t)d * C * C;
    if (zen[d] == 1.5707963267948966) {
        for (int idx = threadIdx.x; idx < B * B; idx += blockDim.x) out[(size_t)(c0 + idx / B) * C + c0 + idx % B] = from_c<T>(mk((idx / B) == (idx % B) ? 1.0 : 0.0, 0.0));
        return;
    }
    for (int idx = threadIdx.x; idx < B * npts; idx += blockDim.x) {
        const int c = c0 + idx / npts, p = idx % npts;
        a_s[idx] = A[(int64_t)c * ldA + (int64_t)d * npts + p];
        z_s[idx] = from_c<T>(Z[(size_t)c * ldP + p]);
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < B * B; idx += blockDim.x) {
        const int i = idx / B, j = idx - i * B;
        T acc = from_c<T>(mk(0.0, 0.0));
        for (int p = 0; p < npts; ++p) cfma(acc, z_s[j * npts + p], a_s[i * npts + p]);
        out[(size_t)(c0 + i) * C + c0 + j] = acc;
    }
}
void launch_rot_from_points(const void* A, int64_t ldA, const void* Z, int ldP, int C, int npts, const double* zen, int D, bool cb, void* Rot,
                            hipStream_t st) {
    if (C > 32) {
        int N = 0;
        while ((N + 1) * (N + 1) < C) ++N;
        const size_t smb = (size_t)2 * (2 * N + 1) * npts * (cb ? sizeof(cplx) : sizeof(double));
        static PerDeviceOnce attr_blocks;
        if (attr_blocks.first()) {
            HIP_CHECK(hipFuncSetAttribute((const void*)rot_blocks_from_points_kernel<cplx>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            HIP_CHECK(hipFuncSetAttribute((const void*)rot_blocks_from_points_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
        launch_zero(Rot, (cb ? sizeof(cplx) : sizeof(double)) * (size_t)D * C * C, st);
        const dim3 grid((unsigned)D, (unsigned)(N + 1));
        if (cb) rot_blocks_from_points_kernel<cplx><<<bgrid(grid), 256, smb, st>>>((const cplx*)A, ldA, (const cplx*)Z, ldP, C, npts, zen, (cplx*)Rot, batch_ctx().stride);
        else rot_blocks_from_points_kernel<double><<<bgrid(grid), 256, smb, st>>>((const double*)A, ldA, (const cplx*)Z, ldP, C, npts, zen, (double*)Rot, batch_ctx().stride);
        KERNEL_CHECK();
        return;
    }
    const size_t sm = (size_t)2 * C * npts * (cb ? sizeof(cplx) : sizeof(double));
    static PerDeviceOnce attr_once;   // (function attributes are per device)
    if (attr_once.first()) {
        HIP_CHECK(hipFuncSetAttribute((const void*)rot_from_points_kernel<cplx>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_CHECK(hipFuncSetAttribute((const void*)rot_from_points_kernel<double>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    if (cb) rot_from_points_kernel<cplx><<<bgrid(D), 256, sm, st>>>((const cplx*)A, ldA, (const cplx*)Z, ldP, C, npts, zen, (cplx*)Rot, batch_ctx().stride);
    else rot_from_points_kernel<double><<<bgrid(D), 256, sm, st>>>((const double*)A, ldA, (const cplx*)Z, ldP, C, npts, zen, (double*)Rot, batch_ctx().stride);
    KERNEL_CHECK();
}

// QT[n][:, d] <- QT[n][:, d] Rot_d, in place.  Rot_d is block diagonal per SH order l (the fit's off-block entries are rounding
// noise): one thread per direction walks the blocks, keeps a block in registers and applies it to all simulation orders n.
template <typename T, int L>
__device__ __forceinline__ void rotate_block(T* __restrict__ QT, int64_t ldD, int nOrd, int C, int d, const T* __restrict__ rot) {
    constexpr int B = 2 * L + 1, c0 = L * L;
    T blk[B][B];
#pragma unroll
    for (int i = 0; i < B; ++i)
#pragma unroll
        for (int j = 0; j < B; ++j) blk[i][j] = rot[(size_t)(c0 + i) * C + (c0 + j)];
    for (int n = 0; n < nOrd; ++n) {
        T* q = QT + ((int64_t)n * C + c0) * ldD + d;
        T v[B], o[B];
#pragma unroll
        for (int i = 0; i < B; ++i) v[i] = q[(int64_t)i * ldD];
#pragma unroll
        for (int j = 0; j < B; ++j) {
            T acc = v[0] * blk[0][j];
#pragma unroll
            for (int i = 1; i < B; ++i) cfma(acc, v[i], blk[i][j]);
            o[j] = acc;
        }
#pragma unroll
        for (int j = 0; j < B; ++j) q[(int64_t)j * ldD] = o[j];
    }
}
// orders 5..7: the block (up to 15 x 15) stays in memory (a thread's own block, read through L2), the output row in registers
template <typename T, int L>
__device__ __forceinline__ void rotate_block_mem(T* __restrict__ QT, int64_t ldD, int nOrd, int C, int d, const T* __restrict__ rot) {
    constexpr int B = 2 * L + 1, c0 = L * L;
    for (int n = 0; n < nOrd; ++n) {
        T* q = QT + ((int64_t)n * C + c0) * ldD + d;
        T o[B];
#pragma unroll
        for (int j = 0; j < B; ++j) o[j] = from_c<T>(mk(0.0, 0.0));
#pragma unroll 1
        for (int i = 0; i < B; ++i) {
            const T vi = q[(int64_t)i * ldD];
            const T* r = rot + (size_t)(c0 + i) * C + c0;
#pragma unroll
            for (int j = 0; j < B; ++j) cfma(o[j], vi, r[j]);
        }
#pragma unroll
        for (int j = 0; j < B; ++j) q[(int64_t)j * ldD] = o[j];
    }
}
template <typename T>
__global__ void __launch_bounds__(128) qt_rotate_kernel(T* __restrict__ QT, int64_t ldD, int nOrd, int C, int N, int D, const T* __restrict__ Rot,
                                                        size_t bstride) {
    QT = boff(QT, bstride); Rot = boff(Rot, bstride);
    const int d = blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    const T* rot = Rot + (size_t)d * C * C;
    // (order 0 is rotation invariant)
    if (N >= 1) rotate_block<T, 1>(QT, ldD, nOrd, C, d, rot);
    if (N >= 2) rotate_block<T, 2>(QT, ldD, nOrd, C, d, rot);
    if (N >= 3) rotate_block<T, 3>(QT, ldD, nOrd, C, d, rot);
    if (N >= 4) rotate_block<T, 4>(QT, ldD, nOrd, C, d, rot);
    if (N >= 5) rotate_block_mem<T, 5>(QT, ldD, nOrd, C, d, rot);
    if (N >= 6) rotate_block_mem<T, 6>(QT, ldD, nOrd, C, d, rot);
    if (N >= 7) rotate_block_mem<T, 7>(QT, ldD, nOrd, C, d, rot);
}
void launch_qt_rotate(void* QT, int64_t ldD, int nOrd, int C, int N, int D, const void* Rot, bool cb, hipStream_t st) {
    if (N > 7) throw Error(2, "EMAinSH: SH order above 7 is not supported in this build");
    const unsigned g = (unsigned)ceil_div(D, 128);
    if (cb) qt_rotate_kernel<cplx><<<bgrid(g), 128, 0, st>>>((cplx*)QT, ldD, nOrd, C, N, D, (const cplx*)Rot, batch_ctx().stride);
    else qt_rotate_kernel<double><<<bgrid(g), 128, 0, st>>>((double*)QT, ldD, nOrd, C, N, D, (const double*)Rot, batch_ctx().stride);
    KERNEL_CHECK();
}

// Gram matrices straight from the direction-space operands: A_k = G_k^H G_k (C x C Hermitian) for the bins [kb0, kb0 + nbins),
// written in the packed real form gram_solve_kernel reads:  P[c C + c'] = c <= c' ? Re A[c][c'] : Im A[c'][c].
// One workgroup per bin; 64 directions at a time through LDS.
constexpr int GFG_TD = 64;
__global__ void __launch_bounds__(256) gram_from_g_kernel(const cplx* __restrict__ G, int64_t g_stride, int64_t ldD, int D, int C, int kb0,
                                                          int g0, double* __restrict__ Apk, int ldK, size_t bstride) {
    G = boff(G, bstride); Apk = boff(Apk, bstride);
    __shared__ cplx g_s[32][GFG_TD + 1];
    const int kb = kb0 + blockIdx.x;
    const cplx* Gk = G + (int64_t)(kb - g0) * g_stride;
    constexpr int NP = 4;   // pairs per thread: 32 * 32 / 256
    cplx acc[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) acc[q] = mk(0.0, 0.0);
    for (int d0 = 0; d0 < D; d0 += GFG_TD) {
        __syncthreads();
        for (int idx = threadIdx.x; idx < C * GFG_TD; idx += 256) {
            const int c = idx / GFG_TD, t = idx - c * GFG_TD;
            g_s[c][t] = (d0 + t < D) ? Gk[(int64_t)c * ldD + d0 + t] : mk(0.0, 0.0);
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int idx = threadIdx.x + q * 256;
            const int c = idx / C, c2 = idx - c * C;
            if (idx < C * C && c <= c2) {
                cplx a = acc[q];
                for (int t = 0; t < GFG_TD; ++t) cfma_conj(a, g_s[c][t], g_s[c2][t]);
                acc[q] = a;
            }
        }
    }
    double* Pk = Apk + (int64_t)blockIdx.x * ldK;   // (rows are relative to kb0, like gram_solve_kernel reads them)
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int idx = threadIdx.x + q * 256;
        const int c = idx / C, c2 = idx - c * C;
        if (idx < C * C && c <= c2) {
            Pk[c * C + c2] = acc[q].x;
            if (c < c2) Pk[c2 * C + c] = acc[q].y;
        }
    }
}
void launch_gram_from_g(const void* G, int64_t g_stride, int64_t ldD, int D, int C, int kb0, int nbins, int g0, double* Apk, int ldK,
                        hipStream_t st) {
    if (nbins <= 0) return;
    if (C > 32) throw Error(2, "EMAinSH: more than 32 channels is not supported");
    gram_from_g_kernel<<<bgrid(nbins), 256, 0, st>>>((const cplx*)G, g_stride, ldD, D, C, kb0, g0, Apk, ldK, batch_ctx().stride);
    KERNEL_CHECK();
}

}  // namespace emagls
