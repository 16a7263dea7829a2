// Gram route of the per-bin regularised inverse: every well-conditioned bin of an array design, batched as ONE
// FP64-MFMA GEMM over the bins instead of an S x C assembly per bin.
//
// Reference (lib/getEMagLsFilters.m:87-90): pwGrid = smairMat(:,:,k) * Y_Hi_conj; [U,s,V] = svd(pwGrid.','econ');
// s = 1 ./ max(s, 0.01*max(s)); Y_reg_inv = conj(U) * (s .* V.').
// With P_k = pwGrid.' = conj(Y) diag(b_n(k)) E^T (D x C; E = the C x S array matrix of dependencies/getSMAIRMatrix.m:101-121):
//     A_k = P_k^H P_k = conj(E) diag(conj(b)) Gy diag(b) E^T = sum_{n,n'} conj(b_n(k)) b_n'(k) K_nn'
//     Gy = Y^T conj(Y)  (S x S Gram matrix of the HRIR-grid SH matrix, gram_mfma_kernel),
//     K_nn' = conj(E_n) Gy[n,n'] E_n'^T  (C x C, frequency independent; K_n'n = K_nn'^H)
// so  A_k = sum_n |b_n|^2 K_nn + sum_{n<n'} ( Re(beta) (K + K^H) + Im(beta) i (K - K^H) ),  beta = conj(b_n) b_n':
// a real GEMM  [bins x nOrd^2] . [nOrd^2 x C^2]  on Hermitian matrices packed into C^2 reals.  From A_k = V S^2 V^H the
// sweep's operand M_k = V diag(s_reg / s) V^H follows (and Y_reg_inv_k = conj(P_k) conj(M_k)); where nothing is clipped,
// i.e. cond(P_k) <= 100, M_k = A_k^-1 (gram_solve_kernel), otherwise the Jacobi kernel of factor.hip takes the bin.
// Error eps cond(P_k)^2: the host starts the route where the modal-strength estimate promises cond < 3e2 and the device
// verifies it (factor.hip); the ill-conditioned low bins keep the orthonormal S-space route (Householder QR + Jacobi SVD).
//
// Packed Hermitian X (C x C) -> C^2 reals:  P[c C + c'] = Re X[c][c'] for c <= c',  Im X[c'][c] for c > c'.
#include "kernels.hpp"

namespace emagls {

typedef double double4_t __attribute__((ext_vector_type(4)));

namespace {

template <typename T> __device__ __forceinline__ T gy_at(const T* __restrict__ Gy, int S, int s, int j) {
    // the Gram kernel fills the upper block triangle (64 x 64 tiles) only: element (s, j) with s > j comes from (j, s)
    return s <= j ? Gy[(int64_t)s * S + j] : conj(Gy[(int64_t)j * S + s]);
}

// F[n][c][s] = sum_{j in order block n} Gy(s, j) E[c][j]   for s < (n+1)^2   (the rows the pairs n' <= n need)
// One workgroup = order n x a tile of 256 rows s; a thread keeps the sums of ALL channels of its row in registers and reads every
// Gy element once (E's block of the order sits in LDS, broadcast reads).  The earlier form -- a workgroup per (order, channel) --
// re-read the order's block column of Gy once per channel: 10.9 GB of traffic and 2.8 ms per 8-design launch at simulation
// order 44 (config 4, r = 10 cm: S = 2025), 41 times the matrix.
constexpr int GE_CMAX = 32;
template <typename T>
__global__ void __launch_bounds__(256) gy_times_e_kernel(const T* __restrict__ Gy, const T* __restrict__ E, int S, int ldE,
                                                         T* __restrict__ F, int64_t ldF, int C, size_t bstride) {
    Gy = boff(Gy, bstride); E = boff(E, bstride); F = boff(F, bstride);
    extern __shared__ __attribute__((aligned(16))) char dyn_ge[];
    T* es = reinterpret_cast<T*>(dyn_ge);   // [C][nb + 1]
    const int n = blockIdx.x;
    const int jb = n * n, je = min(S, (n + 1) * (n + 1)), nb = je - jb;
    if ((int)blockIdx.y * 256 >= je) return;
    for (int idx = threadIdx.x; idx < C * nb; idx += 256) {
        const int c = idx / nb, j = idx % nb;
        es[c * (nb + 1) + j] = E[(int64_t)c * ldE + jb + j];
    }
    __syncthreads();
    const int s = blockIdx.y * 256 + threadIdx.x;
    if (s >= je) return;
    T acc[GE_CMAX];
#pragma unroll
    for (int c = 0; c < GE_CMAX; ++c) acc[c] = zero_of<T>();
    for (int j = 0; j < nb; ++j) {
        const T g = gy_at(Gy, S, s, jb + j);
#pragma unroll
        for (int c = 0; c < GE_CMAX; ++c)
            if (c < C) cfma(acc[c], g, es[c * (nb + 1) + j]);
    }
#pragma unroll
    for (int c = 0; c < GE_CMAX; ++c)
        if (c < C) F[((int64_t)n * C + c) * ldF + s] = acc[c];
}

__device__ __forceinline__ double pk_re(double v) { return v; }
__device__ __forceinline__ double pk_re(cplx v) { return v.x; }
__device__ __forceinline__ double pk_im(double) { return 0.0; }
__device__ __forceinline__ double pk_im(cplx v) { return v.y; }

// one workgroup per order pair (n <= n'):  K = conj(E_n) F_n'[block n]  (C x C, staged in LDS), folded into the packed rows
//   n == n' : row n                  <- (K + K^H) / 2
//   n <  n' : rows nOrd + 2 q, + 1   <- K + K^H,  i (K - K^H)         q = index of the pair in lexicographic order
template <typename T>
__global__ void __launch_bounds__(256) kfold_kernel(const T* __restrict__ E, int ldE, const T* __restrict__ F, int64_t ldF, int S, int C,
                                                    int nOrd, double* __restrict__ Kmat, int ldK, size_t bstride) {
    E = boff(E, bstride); F = boff(F, bstride); Kmat = boff(Kmat, bstride);
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    T* Ks = reinterpret_cast<T*>(dyn);   // [C][C + 1]
    int t = blockIdx.x, n = 0;
    while (t >= nOrd - n) { t -= nOrd - n; ++n; }
    const int n2 = n + t;
    const int sb = n * n, se = min(S, (n + 1) * (n + 1));
    for (int idx = threadIdx.x; idx < C * C; idx += blockDim.x) {
        const int c = idx / C, c2 = idx % C;
        const T* e = E + (int64_t)c * ldE;
        const T* f = F + ((int64_t)n2 * C + c2) * ldF;
        T a0 = zero_of<T>(), a1 = zero_of<T>();
        int s = sb;
        for (; s + 1 < se; s += 2) { cfma_conj(a0, e[s], f[s]); cfma_conj(a1, e[s + 1], f[s + 1]); }
        if (s < se) cfma_conj(a0, e[s], f[s]);
        Ks[c * (C + 1) + c2] = a0 + a1;
    }
    __syncthreads();
    // pair index q of (n, n2), n < n2, in lexicographic order
    const int q = n * nOrd - n * (n + 1) / 2 + (n2 - n - 1);
    double* r0 = Kmat + (int64_t)(n == n2 ? n : nOrd + 2 * q) * ldK;
    double* r1 = r0 + ldK;
    for (int idx = threadIdx.x; idx < C * C; idx += blockDim.x) {
        const int c = idx / C, c2 = idx % C;
        const int a = min(c, c2), b = max(c, c2);          // the upper-triangle element this slot describes
        const T kab = Ks[a * (C + 1) + b], kba = Ks[b * (C + 1) + a];
        // X = K + K^H : X[a][b] = K[a][b] + conj(K[b][a]);   Z = i (K - K^H) : Z[a][b] = i (K[a][b] - conj(K[b][a]))
        const double xr = pk_re(kab) + pk_re(kba), xi = pk_im(kab) - pk_im(kba);
        const double zr = -(pk_im(kab) + pk_im(kba)), zi = pk_re(kab) - pk_re(kba);
        if (n == n2) r0[idx] = 0.5 * (c <= c2 ? xr : xi);
        else { r0[idx] = c <= c2 ? xr : xi; r1[idx] = c <= c2 ? zr : zi; }
    }
}

// coefficients of the GEMM, K-major:  Cf[row][bin - kb0]  (rows as in kfold_kernel); Nyquist bin: real(b_n).
// thread = (bin, first order n): lanes are consecutive bins, so every row is written in contiguous runs
__global__ void __launch_bounds__(128) gram_coef_kernel(const cplx* __restrict__ bn, int nOrd, int P, int kb0, int nbins, double* __restrict__ Cf,
                                                        int ldC, size_t bstride) {
    bn = boff(bn, bstride); Cf = boff(Cf, bstride);
    const int bi = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y;
    if (bi >= nbins) return;
    const int kb = kb0 + bi;
    const cplx* b = bn + (int64_t)kb * nOrd;
    const bool nyq = kb == P - 1;
    cplx x = b[n];
    if (nyq) x.y = 0.0;
    Cf[(int64_t)n * ldC + bi] = norm2(x);
    int q = n * nOrd - n * (n + 1) / 2;   // index of the pair (n, n + 1) in lexicographic order
    for (int m = n + 1; m < nOrd; ++m, ++q) {
        cplx y = b[m];
        if (nyq) y.y = 0.0;
        cplx be = mk(0, 0);
        cfma_conj(be, x, y);   // conj(b_n) b_m
        Cf[(int64_t)(nOrd + 2 * q) * ldC + bi] = be.x;
        Cf[(int64_t)(nOrd + 2 * q + 1) * ldC + bi] = be.y;
    }
}

// C[m][n] = sum_k A[k][m] B[k][n]  (both operands K-major: a wave's 16 lanes of an MFMA operand read 128 contiguous bytes).
// Workgroup = 4 waves = 64 x 64 tile, each wave 2 x 2 tiles of v_mfma_f64_16x16x4_f64.  A and B are padded with zeros to
// multiples of 64 columns and K to a multiple of 4 rows, so the loop carries no bounds test.
// f64 fragment layout: A[i = l & 15][k = l >> 4], B[k = l >> 4][j = l & 15]; C/D: col = l & 15, row = (l >> 4) + 4 reg.
// Split-K: blockIdx.x = m tile + mtiles * split; split s covers the rows [s K, (s+1) K) and writes the partial product at
// Cm + s * cstride (the caller sums the partials in a fixed order).
__global__ void __launch_bounds__(256) gemm_tn_f64_kernel(const double* __restrict__ A, int lda, const double* __restrict__ B, int ldb, int K,
                                                          double* __restrict__ Cm, int ldc, int M, int N, int mtiles, int64_t cstride,
                                                          size_t bstride, int xcd_runs) {
    // (the tiles of a lane share both operands: XCD-aware order, xcd_run_index)
    unsigned zl = blockIdx.z, bx = blockIdx.x, by = blockIdx.y;
    if (xcd_runs) { unsigned tl; xcd_run_index(tl, zl); by = tl / gridDim.x; bx = tl - by * gridDim.x; }
    A = boffz(A, bstride, zl); B = boffz(B, bstride, zl); Cm = boffz(Cm, bstride, zl);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int split = (int)bx / mtiles, mt = (int)bx - split * mtiles;
    A += (int64_t)split * K * lda; B += (int64_t)split * K * ldb; Cm += (int64_t)split * cstride;
    const int m0 = mt * 64 + (wave >> 1) * 32, n0 = (int)by * 64 + (wave & 1) * 32;
    const int ii = lane & 15, kk = lane >> 4;
    double4_t acc[2][2];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc[x][y] = double4_t{0.0, 0.0, 0.0, 0.0};
    const double* pa = A + (int64_t)kk * lda + m0 + ii;
    const double* pb = B + (int64_t)kk * ldb + n0 + ii;
#pragma unroll 4
    for (int k = 0; k < K; k += 4) {
        const double a0 = pa[0], a1 = pa[16], b0 = pb[0], b1 = pb[16];
        pa += 4 * (int64_t)lda; pb += 4 * (int64_t)ldb;
        acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gm = m0 + 16 * x + kk + 4 * r, gn = n0 + 16 * y + ii;
                if (gm < M && gn < N) Cm[(int64_t)gm * ldc + gn] = acc[x][y][r];
            }
}

// wave-synchronous LDS hand-over inside one wave (no workgroup barrier)
__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Direct route: one WAVE per bin inverts the Hermitian positive definite A_k (C <= 32, padded to 32 with the identity) in
// registers by C Gauss-Jordan sweep steps without pivoting.  Lane l holds column k = l & 31, rows i = (l >> 5) + 2 t.
// A swept matrix stays Hermitian up to signs: row j is conj(column j), negated at the already swept positions, so only the
// pivot column travels through LDS.  Certificate that the reference's 1 % clipping does nothing:
// cond(A) <= ||A||_F ||A^-1||_F <= 1 / reg_c^2.  Bins that fail it get their full A written to R2w for the Jacobi kernel.
constexpr int GS_C = 32;
__global__ void __launch_bounds__(256) gram_solve_kernel(const double* __restrict__ Apk, int ldA, int C, int kb0, int nbins, double reg_c,
                                                         cplx* __restrict__ Mw, cplx* __restrict__ R2w, double* __restrict__ sv,
                                                         int* __restrict__ route, int* __restrict__ sweeps_out, size_t bstride) {
    Apk = boff(Apk, bstride); Mw = boff(Mw, bstride); R2w = boff(R2w, bstride); sv = boff(sv, bstride); route = boff(route, bstride);
    sweeps_out = boff(sweeps_out, bstride);
    __shared__ __attribute__((aligned(16))) cplx colbuf[4][GS_C];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int bi = blockIdx.x * 4 + wave;
    if (bi >= nbins) return;   // (wave-uniform; no workgroup barrier below)
    const int kb = kb0 + bi;
    const double* P = Apk + (int64_t)bi * ldA;
    const int k = lane & 31, ih = lane >> 5;
    cplx x[16];
    double fa = 0.0;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const int i = ih + 2 * t;
        cplx v = mk(i == k ? 1.0 : 0.0, 0.0);
        if (i < C && k < C) {
            const int a = min(i, k), b = max(i, k);
            const double re = P[a * C + b], im = a < b ? P[b * C + a] : 0.0;
            v = mk(re, i <= k ? im : -im);   // X[i][k]: upper element as stored, lower element conjugated
            fa += norm2(v);
        }
        x[t] = v;
    }
    fa = wave_sum(fa);
    cplx* col = colbuf[wave];
    bool bad = false;
    // Round 5: the steps are unrolled (j is a compile-time constant inside a step), so that row j's register x[j >> 1] is named
    // directly, and the three special cases of a sweep step -- pivot, row j, column j -- are folded into the operands of ONE
    // update form x <- alpha x - c_i r_k (column j: alpha = -1 / p, r = 0, its x IS c_i; elsewhere alpha = 1, r = row_k / p)
    // instead of three selects per element: 8.3 k vector instructions per bin before, a third of that now (profiles/r05_pmc.md).
#pragma unroll
    for (int j = 0; j < GS_C; ++j) {
        if (j >= C) continue;   // (uniform)
        const bool colj = k == j;
        if (colj) {
#pragma unroll
            for (int t = 0; t < 16; ++t) col[ih + 2 * t] = x[t];
        }
        wave_sync_lds();
        const cplx pj = col[j];
        bad = bad || !(pj.x > 0.0);
        const double ip = fast_rcp(pj.x > 0.0 ? pj.x : 1.0);
        // row j from column j: A[j][k] = conj(A[k][j]) for k not yet swept, -conj(A[k][j]) for k < j
        const cplx ck = col[k];
        const cplx rowk = k < j ? mk(-ck.x, ck.y) : mk(ck.x, -ck.y);
        const cplx rkf = mk(rowk.x * ip, rowk.y * ip);
        const double alpha = colj ? -ip : 1.0;
        const cplx rk = colj ? mk(0.0, 0.0) : rkf;
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const cplx ci = col[ih + 2 * t];
            cplx v;
            v.x = fma(ci.y, rk.y, fma(-ci.x, rk.x, x[t].x * alpha));
            v.y = fma(-ci.y, rk.x, fma(-ci.x, rk.y, x[t].y * alpha));
            x[t] = v;
        }
        // row j itself: the pivot's reciprocal on the diagonal, row_k / p elsewhere
        if (ih == (j & 1)) x[j >> 1] = colj ? mk(ip, 0.0) : rkf;
        wave_sync_lds();   // the column buffer is rewritten in the next step
    }
    double fm = 0.0;
#pragma unroll
    for (int t = 0; t < 16; ++t) { const int i = ih + 2 * t; if (i < C && k < C) fm += norm2(x[t]); }
    fm = wave_sum(fm);
    const double thr = 1.0 / (reg_c * reg_c);
    const bool direct = !__builtin_amdgcn_ballot_w64(bad) && fa > 0.0 && fa * fm <= thr * thr;   // (||A||_F ||A^-1||_F)^2
    if (direct) {
        cplx* M = Mw + (int64_t)(kb - 1) * C * C;     // (the factor stage stores bin kb at slot kb - 1)
#pragma unroll
        for (int t = 0; t < 16; ++t) { const int i = ih + 2 * t; if (i < C && k < C) M[i * C + k] = x[t]; }
        // bounds instead of singular values: s_max <= ||A||_F^(1/2), s_min >= ||A^-1||_F^(-1/2)
        if (sv && lane < C) sv[(int64_t)kb * C + lane] = lane == 0 ? sqrt(sqrt(fa)) : 1.0 / sqrt(sqrt(fm));
        if (lane == 0) { route[kb] = 2; if (sweeps_out) sweeps_out[kb] = 0; }
    } else {
        cplx* A = R2w + (int64_t)(kb - 1) * C * C;    // full Hermitian matrix, row major: the Jacobi kernel's Gram form
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            const int i = ih + 2 * t;
            if (i < C && k < C) {
                const int a = min(i, k), b = max(i, k);
                const double re = P[a * C + b], im = a < b ? P[b * C + a] : 0.0;
                A[i * C + k] = mk(re, i <= k ? im : -im);
            }
        }
        if (lane == 0) route[kb] = 1;
    }
}

// least-squares bins on the Gram route:  W(k,:) = H(k,:) Y_reg_inv_k = (H(k,:) conj(G_k)) conj(M_k)
// one workgroup (16 waves) per bin; a wave takes the channels c = wave, wave + 16, ..., its lanes stride over the directions
// (1 KB loads; the few bins of this kind leave the chip empty, so a bin's channels run side by side)
__global__ void __launch_bounds__(1024) ls_gram_kernel(const cplx* __restrict__ Hc, int64_t ldH, int n_c, const cplx* __restrict__ G, int64_t g_stride,
                                                      int64_t ldD, const cplx* __restrict__ Mw, int D, int C, int P, int kb0, cplx* __restrict__ W,
                                                      size_t bstride) {
    Hc = boff(Hc, bstride); G = boff(G, bstride); Mw = boff(Mw, bstride); W = boff(W, bstride);
    __shared__ __attribute__((aligned(16))) cplx vt[64];
    const int kb = kb0 + blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const cplx* h0 = Hc + ((int64_t)0 * n_c + kb) * ldH;
    const cplx* h1 = Hc + ((int64_t)1 * n_c + kb) * ldH;
    for (int c = wave; c < C; c += 16) {
        const cplx* g = G + (int64_t)kb * g_stride + (int64_t)c * ldD;
        cplx a0 = mk(0, 0), a1 = mk(0, 0), b0 = mk(0, 0), b1 = mk(0, 0);
        int d = lane;
        for (; d + 64 < D; d += 128) {
            const cplx ga = conj(g[d]), gb = conj(g[d + 64]);
            cfma(a0, h0[d], ga); cfma(a1, h1[d], ga);
            cfma(b0, h0[d + 64], gb); cfma(b1, h1[d + 64], gb);
        }
        if (d < D) { const cplx ga = conj(g[d]); cfma(a0, h0[d], ga); cfma(a1, h1[d], ga); }
        const cplx s0 = wave_sum(a0 + b0), s1 = wave_sum(a1 + b1);
        if (lane == 0) { vt[c] = s0; vt[32 + c] = s1; }
    }
    __syncthreads();
    for (int pair = threadIdx.x; pair < 2 * C; pair += blockDim.x) {
        const int e = pair / C, c = pair % C;
        const cplx* M = Mw + (int64_t)(kb - 1) * C * C;
        cplx acc = mk(0, 0);
        for (int cc = 0; cc < C; ++cc) cfma(acc, vt[32 * e + cc], conj(M[cc * C + c]));
        W[((int64_t)e * P + kb) * C + c] = acc;
    }
}

}  // namespace

int gram_kmat_rows(int nOrd) { return nOrd * nOrd; }

// F workspace: [nOrd][C][ldF];  Kmat: [round_up(nOrd^2, 4)][ldK] with ldK = round_up(C^2, 64), zero filled once at plan set-up
void launch_gram_kmat(const void* Gy, const void* E, int S, int ldE, int C, int nOrd, bool is_cplx, void* F, int64_t ldF, double* Kmat, int ldK,
                      hipStream_t st) {
    const int npairs = nOrd * (nOrd + 1) / 2;
    if (C > GE_CMAX) throw Error(2, "gram route: more than 32 channels");
    if (is_cplx) {
        gy_times_e_kernel<cplx><<<bgrid(dim3(nOrd, (unsigned)ceil_div(S, 256))), 256, sizeof(cplx) * C * (2 * nOrd), st>>>((const cplx*)Gy, (const cplx*)E, S, ldE, (cplx*)F, ldF, C, batch_ctx().stride);
        KERNEL_CHECK();
        kfold_kernel<cplx><<<bgrid(npairs), 256, sizeof(cplx) * C * (C + 1), st>>>((const cplx*)E, ldE, (const cplx*)F, ldF, S, C, nOrd, Kmat, ldK, batch_ctx().stride);
    } else {
        gy_times_e_kernel<double><<<bgrid(dim3(nOrd, (unsigned)ceil_div(S, 256))), 256, sizeof(double) * C * (2 * nOrd), st>>>((const double*)Gy, (const double*)E, S, ldE, (double*)F, ldF, C, batch_ctx().stride);
        KERNEL_CHECK();
        kfold_kernel<double><<<bgrid(npairs), 256, sizeof(double) * C * (C + 1), st>>>((const double*)E, ldE, (const double*)F, ldF, S, C, nOrd, Kmat, ldK, batch_ctx().stride);
    }
    KERNEL_CHECK();
}

// A_k (packed) for the bins [kb0, kb0 + nbins):  Apk[bin][C^2] = Cf^T Kmat
void launch_gram_gemm(const void* bn, int nOrd, int P, int kb0, int nbins, double* Cf, int ldC, const double* Kmat, int ldK, int C, double* Apk,
                      int ldA, hipStream_t st) {
    if (nbins <= 0) return;
    gram_coef_kernel<<<bgrid(dim3((nbins + 127) / 128, nOrd)), 128, 0, st>>>((const cplx*)bn, nOrd, P, kb0, nbins, Cf, ldC, batch_ctx().stride);
    KERNEL_CHECK();
    const int K = (nOrd * nOrd + 3) / 4 * 4;
    const int mtiles = (nbins + 63) / 64;
    gemm_tn_f64_kernel<<<bgrid(dim3(mtiles, (C * C + 63) / 64)), 256, 0, st>>>(Cf, ldC, Kmat, ldK, K, Apk, ldA, nbins, C * C, mtiles, 0,
                                                                               batch_ctx().stride, xcd_runs_enabled());
    KERNEL_CHECK();
}

// ---- least-squares rows H conj(Yc) on the matrix pipe (the same GEMM kernel, split over the directions)
namespace {
constexpr int HYM_KS = 16;   // K splits: 2702 directions -> slices of 172, summed in a fixed order
// out[r][s] = conj(P[r][s]), P = sum over splits; Pw[split][2 r + re/im of H][s (x2: re/im of Yc when complex)]
__global__ void __launch_bounds__(256) hy_combine_kernel(const double* __restrict__ Pw, int ldP, int64_t pstride, int nrows, int S, int y_cplx,
                                                         cplx* __restrict__ out, int ldS, size_t bstride) {
    Pw = boff(Pw, bstride); out = boff(out, bstride);
    const int r = blockIdx.y;
    const int sc = blockIdx.x * 256 + threadIdx.x;
    if (sc >= S) return;
    double hr_yr = 0.0, hi_yr = 0.0, hr_yi = 0.0, hi_yi = 0.0;
    for (int ks = 0; ks < HYM_KS; ++ks) {
        const double* p0 = Pw + (int64_t)ks * pstride + (int64_t)(2 * r) * ldP;
        const double* p1 = p0 + ldP;
        if (y_cplx) { hr_yr += p0[2 * sc]; hr_yi += p0[2 * sc + 1]; hi_yr += p1[2 * sc]; hi_yi += p1[2 * sc + 1]; }
        else { hr_yr += p0[sc]; hi_yr += p1[sc]; }
    }
    // P = (Hr + i Hi)(Yr - i Yi) = (Hr Yr + Hi Yi) + i (Hi Yr - Hr Yi);  out = conj(P)
    out[(int64_t)r * ldS + sc] = mk(hr_yr + hi_yi, -(hi_yr - hr_yi));
}
}  // namespace

static inline int hym_kslice(int D) { return (int)(ceil_div(ceil_div(D, HYM_KS), 4) * 4); }
size_t hy_mfma_workspace_doubles(int n_c, int S, bool y_cplx) {
    const int M = 4 * n_c, N = y_cplx ? 2 * S : S;
    return (size_t)HYM_KS * (size_t)(ceil_div(M, 64) * 64) * (size_t)(ceil_div(N, 64) * 64);
}
void launch_hy_conj_mfma(const double* HcT, int ldT, int n_c, const void* Yc, int64_t ldY, bool y_cplx, int D, int S, double* Pw, void* out,
                         int ldS, hipStream_t st) {
    if (n_c <= 0) return;
    const int M = 4 * n_c, N = y_cplx ? 2 * S : S;            // rows: (ear, bin, re/im); columns: s (re/im interleaved when complex)
    const int ldb = (int)(y_cplx ? 2 * ldY : ldY);
    const int ldP = (int)(ceil_div(N, 64) * 64), mtiles = (M + 63) / 64;
    const int64_t pstride = (int64_t)(ceil_div(M, 64) * 64) * ldP;
    const int kc = hym_kslice(D);                            // rows D .. HYM_KS kc of both operands are zero (padded buffers)
    gemm_tn_f64_kernel<<<bgrid(dim3(mtiles * HYM_KS, (N + 63) / 64)), 256, 0, st>>>(HcT, ldT, (const double*)Yc, ldb, kc, Pw, ldP, M, N, mtiles,
                                                                                    pstride, batch_ctx().stride, xcd_runs_enabled());
    KERNEL_CHECK();
    hy_combine_kernel<<<bgrid(dim3((unsigned)ceil_div(S, 256), 2 * n_c)), 256, 0, st>>>(Pw, ldP, pstride, 2 * n_c, S, y_cplx ? 1 : 0, (cplx*)out, ldS,
                                                                                      batch_ctx().stride);
    KERNEL_CHECK();
}
int hy_mfma_kpad(int D) { return HYM_KS * hym_kslice(D); }

void launch_gram_solve(const double* Apk, int ldA, int C, int kb0, int nbins, double reg_c, void* Mw, void* R2w, double* sv, int* route,
                       int* sweeps_out, hipStream_t st) {
    if (nbins <= 0) return;
    if (C > GS_C) throw Error(2, "gram route: more than 32 channels");
    gram_solve_kernel<<<bgrid((nbins + 3) / 4), 256, 0, st>>>(Apk, ldA, C, kb0, nbins, reg_c, (cplx*)Mw, (cplx*)R2w, sv, route, sweeps_out,
                                                            batch_ctx().stride);
    KERNEL_CHECK();
}

void launch_ls_gram(const void* Hc, int64_t ldH, int n_c, const void* G, int64_t g_stride, int64_t ldD, const void* Mw, int D, int C, int P,
                    int kb_lo, int kb_hi, void* W, hipStream_t st) {
    if (kb_hi <= kb_lo) return;
    if (2 * C > 64) throw Error(2, "gram route: more than 32 channels");
    ls_gram_kernel<<<bgrid(kb_hi - kb_lo), 1024, 0, st>>>((const cplx*)Hc, ldH, n_c, (const cplx*)G, g_stride, ldD, (const cplx*)Mw, D, C, P, kb_lo,
                                                        (cplx*)W, batch_ctx().stride);
    KERNEL_CHECK();
}

}  // namespace emagls
