// Per-frequency-bin regularised inverse, batched over bins (one workgroup per bin).
//
// Reference (lib/getEMagLsFilters.m:87-90, same in getEMagLs2Filters.m:86-89, getEMagLsFiltersFromAtf.m:101-104):
//     pwGrid = smairMat(:,:,k) * Y_Hi_conj;  [U,s,V] = svd(pwGrid.','econ');
//     s = 1 ./ max(s, 0.01*max(s));          Y_reg_inv = conj(U) * (s .* V.');
//
// Here pwGrid.' = Q B_k with Q orthonormal (gram_chol.hip), B_k = R diag(b_n(k)) E^T = sum_n b_n(k) T_n
// (S x C), so  Y_reg_inv = conj(Q) Z_k  with  Z_k = conj(U_B) diag(s_reg) V^T  (S x C), U_B S V^H = B_k.
// The clipped singular values receive the LARGEST weight (100/s_max), so U_B must be orthonormal to
// working precision even where s_min/s_max ~ 1e-13: a Gram/normal-equation solve cannot do that.
// Per bin:  Householder QR of B_k (registers, column-per-lane-group)  ->  one-sided Jacobi SVD of
// R2^H (LDS, converges in <= 10 sweeps)  ->  N = U2 diag(s_reg) V^H  ->  Z_k = conj(Q2 [N; 0]).
//
// Thread layout: tid = c * NCH + ch; lane group (NCH = 32 or 64 consecutive lanes) owns column c,
// lane ch owns rows s = ch + NCH*i.  All reductions are inside one wave.
#include "kernels.hpp"

namespace emagls {

// batches: shift every pointer of the argument block to design z
__device__ __forceinline__ void batch_offset(FactorArgs& a, size_t bstride, unsigned z) {
    a.Tn = boff_flat(a.Tn, bstride, z); a.bn = boff_flat(a.bn, bstride, z); a.Xd = boff_flat(a.Xd, bstride, z);
    a.Z = boff_flat(a.Z, bstride, z); a.Vws = boff_flat(a.Vws, bstride, z); a.sv = boff_flat(a.sv, bstride, z);
    a.Hq = boff_flat(a.Hq, bstride, z); a.cond_ok = boff_flat(a.cond_ok, bstride, z); a.W = boff_flat(a.W, bstride, z); a.sweeps_out = boff_flat(a.sweeps_out, bstride, z);
    a.route = boff_flat(a.route, bstride, z); a.status = boff_flat(a.status, bstride, z);
    a.tauw = boff_flat(a.tauw, bstride, z); a.R2w = boff_flat(a.R2w, bstride, z); a.Nw = boff_flat(a.Nw, bstride, z); a.Mw = boff_flat(a.Mw, bstride, z);
}

__device__ __forceinline__ void batch_offset(FactorArgs& a, size_t bstride) { batch_offset(a, bstride, blockIdx.z); }

constexpr int CPMAX = 32;  // max (even-padded) column count

// =============================================================================================
// kernel 1: assemble B_k, Householder QR.  Leaves v_j in Vws, tau_j in tauw, R2 (upper) in R2w.
// =============================================================================================
template <typename TT, int NCH, int RPT, int MAXT>
__global__ void __launch_bounds__(MAXT) factor_qr_kernel(FactorArgs a, size_t bstride) {
    // (tried: flattened index split as (design = L mod n, bin = L / n) so that a design's workgroups share one XCD and its
    // T_n stays in that L2: no measurable change, 859 vs 869 us per 8-design launch with +-40 us between launches)
    const unsigned bl_z = blockIdx.z, bl_x = blockIdx.x;
    batch_offset(a, bstride, bl_z);
    __shared__ __attribute__((aligned(16))) cplx bns[96];
    __shared__ cplx alpha_s[CPMAX];
    __shared__ double tau_s[CPMAX];
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    cplx* vbuf = reinterpret_cast<cplx*>(dyn);  // [2][ldS]

    const int tid = threadIdx.x;
    const int c = tid / NCH, ch = tid % NCH;
    const int S = a.S, C = a.C, ldS = a.ldS;
    const int kb = a.kb0 + (int)bl_x;
    const bool active = c < C;

    cplx B[RPT];
    // ------------------------------------------------------------------ 1. assemble / load B_k
    if (a.Tn) {
        for (int n = tid; n < a.nOrders; n += blockDim.x) {
            cplx b = a.bn[(int64_t)kb * a.nOrders + n];
            if (kb == a.P - 1) b.y = 0.0;  // Nyquist: real(Bn)  (dependencies/getSMAIRMatrix.m:115-117)
            bns[n] = b;
        }
        __syncthreads();
        const TT* Tn = reinterpret_cast<const TT*>(a.Tn);
        // (tried: a uniform start order per row group with four independent masked loads in flight per pass instead of
        // this load -> wait -> FMA loop: 960 vs 870 us per 8-design launch -- the kernel sits at its 128-VGPR cap, the wider
        // loop spills, and lanes masked by their own start order fetch fewer lines of the 1.6 MB T_n stream per workgroup)
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int s = ch + NCH * i;
            cplx acc = mk(0, 0);
            if (active && s < S) {
                int n0 = (int)sqrt((double)s);  // first order whose block reaches row s: (n0+1)^2 > s
                while ((n0 + 1) * (n0 + 1) <= s) ++n0;
                while (n0 > 0 && n0 * n0 > s) --n0;
                for (int n = n0; n < a.nOrders; ++n) cfma(acc, bns[n], Tn[((int64_t)n * C + c) * ldS + s]);
            }
            B[i] = acc;
        }
    } else {
        const cplx* X = a.Xd + (int64_t)kb * a.xd_stride;
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int s = ch + NCH * i;
            B[i] = (active && s < S) ? X[(int64_t)c * ldS + s] : mk(0, 0);
        }
    }
    // ------------------------------------------------------------------ 1b. Gram route
    // Well above k_cut the columns of B_k are well conditioned (cond < ~3e2, decided by the host from kr): the Jacobi
    // kernel can work on A = B^H B directly (error eps cond^2, here < 1e-11) and the 25 sequential reflector steps,
    // their workspace traffic and the back-transform disappear.  A is formed from LDS row chunks in 2 x 2 tiles.
    if (a.gram_from > 0 && kb >= a.gram_from) {
        constexpr int GR = 4 * NCH;                    // rows per chunk (each lane contributes 4 of its rows)
        cplx* Bs = vbuf + 2 * (size_t)ldS;             // [CPMAX][GR + 1]
        constexpr int GLD = GR + 1;
        for (int idx = tid; idx < CPMAX * GLD; idx += blockDim.x) Bs[idx] = mk(0, 0);
        const int nblk1 = (C + 1) / 2, ntile = nblk1 * (nblk1 + 1) / 2;
        const int part = tid & 7, tstep = blockDim.x >> 3;
        constexpr int TS = 2;   // tile slots per thread: (16 * 17 / 2 = 136 tiles) * 8 lanes <= 2 * 832 threads
        int tbi[TS], tbj[TS];
        bool tvalid[TS];
        cplx g00[TS], g01[TS], g10[TS], g11[TS];
#pragma unroll
        for (int u = 0; u < TS; ++u) {
            int t = (tid >> 3) + u * tstep, bi = 0;
            tvalid[u] = t < ntile;
            while (bi < nblk1 && t >= nblk1 - bi) { t -= nblk1 - bi; ++bi; }
            tbi[u] = tvalid[u] ? bi : 0;
            tbj[u] = tvalid[u] ? bi + t : 0;
            g00[u] = g01[u] = g10[u] = g11[u] = mk(0, 0);
        }
        __syncthreads();
        for (int g = 0; g * 4 < RPT; ++g) {
            if (active) {
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) {
                    const int i = 4 * g + ii;
                    cplx v = mk(0, 0);
#pragma unroll
                    for (int i2 = 0; i2 < RPT; ++i2) if (i2 == i) v = B[i2];   // (static register indexing)
                    Bs[c * GLD + ch + NCH * ii] = v;
                }
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < TS; ++u) {
                if (!tvalid[u]) continue;
                const cplx* a0 = Bs + (2 * tbi[u]) * GLD, *a1 = a0 + GLD, *b0 = Bs + (2 * tbj[u]) * GLD, *b1 = b0 + GLD;
#pragma unroll 4
                for (int rr = 0; rr < GR / 8; ++rr) {
                    const int r = rr * 8 + part;
                    const cplx x0 = a0[r], x1 = a1[r], y0 = b0[r], y1 = b1[r];
                    cfma_conj(g00[u], x0, y0); cfma_conj(g01[u], x0, y1); cfma_conj(g10[u], x1, y0); cfma_conj(g11[u], x1, y1);
                }
            }
            __syncthreads();
        }
        cplx* A = a.R2w + (int64_t)bl_x * C * C;   // full Hermitian matrix, row major
        auto put = [&](int r, int cc, cplx v) {
            if (r < C && cc < C) { A[(int64_t)r * C + cc] = v; A[(int64_t)cc * C + r] = conj(v); }
        };
#pragma unroll
        for (int u = 0; u < TS; ++u) {
            const cplx s00 = group_sum<8>(g00[u]), s01 = group_sum<8>(g01[u]), s10 = group_sum<8>(g10[u]), s11 = group_sum<8>(g11[u]);
            if (tvalid[u] && part == 0) {
                const int r0 = 2 * tbi[u], c0 = 2 * tbj[u];
                put(r0, c0, s00); put(r0, c0 + 1, s01); put(r0 + 1, c0, s10); put(r0 + 1, c0 + 1, s11);
            }
        }
        if (tid == 0) a.route[kb] = 1;
        return;
    }
    // ------------------------------------------------------------------ 2. Householder QR
    // One barrier per column: while the lane groups c > j apply H_j, the group of column j+1 goes on to form v_{j+1}
    // from its freshly updated column (look-ahead), so the pivot computation never leaves the other groups waiting.
    // The pivot arithmetic (norm, phase, tau) uses the reciprocal / rsqrt seeds + Newton steps instead of the
    // library sqrt, hypot and divisions: it sits on the critical path of all 25 steps.
    cplx* Vw = a.Vws + (int64_t)bl_x * C * ldS;
    auto make_reflector = [&](int j) {  // executed by the lane group c == j on its own column
        cplx* vb = vbuf + (size_t)(j & 1) * ldS;
        double n2 = 0.0;
        cplx x0 = mk(0, 0);
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int s = ch + NCH * i;
            if (s >= j && s < S) n2 += norm2(B[i]);
            if (s == j) x0 = B[i];
        }
        n2 = group_sum<NCH>(n2);
        x0 = group_sum<NCH>(x0);
        const bool nz = n2 > 0.0;
        const double nrm = nz ? n2 * fast_rsqrt(n2) : 0.0;
        const double a2 = norm2(x0);
        const double iax0 = a2 > 0.0 ? fast_rsqrt(a2) : 0.0;   // 1 / |x0|
        const double ax0 = a2 * iax0;
        cplx alpha = mk(0, 0);
        double tau = 0.0;
        if (nz) {
            alpha = (a2 > 0.0) ? mk(-x0.x * iax0 * nrm, -x0.y * iax0 * nrm) : mk(-nrm, 0.0);
            tau = fast_rcp(nrm * (nrm + ax0));  // 2 / ||v||^2, ||v||^2 = 2 nrm (nrm + |x0|)
        }
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int s = ch + NCH * i;
            if (s == j) B[i] = B[i] - alpha;  // v0 = x0 - alpha
            if (s >= j && s < S) {
                const cplx v = nz ? B[i] : mk(0, 0);
                vb[s] = v;
                Vw[(int64_t)j * ldS + s] = v;
            }
        }
        if (ch == 0) { alpha_s[j] = alpha; tau_s[j] = tau; }
    };
    if (c == 0) make_reflector(0);
    __syncthreads();
    for (int j = 0; j < C; ++j) {
        const cplx* vb = vbuf + (size_t)(j & 1) * ldS;
        if (active && c > j) {
            const double tau = tau_s[j];
            cplx w = mk(0, 0);
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int s = ch + NCH * i;
                if (s >= j && s < S) cfma_conj(w, vb[s], B[i]);
            }
            w = group_sum<NCH>(w);
            w = w * tau;
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int s = ch + NCH * i;
                if (s >= j && s < S) { cplx p = w * vb[s]; B[i] -= p; }
            }
            if (c == j + 1) make_reflector(j + 1);  // writes the other half of vbuf
        }
        __syncthreads();
    }
    // ------------------------------------------------------------------ 3. hand R2 and tau to the SVD kernel
    if (active) {
        cplx* R2 = a.R2w + (int64_t)bl_x * C * C;
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int s = ch + NCH * i;
            if (s < c) R2[(int64_t)s * C + c] = B[i];
        }
        if (ch == 0) {
            R2[(int64_t)c * C + c] = alpha_s[c];
            a.tauw[(int64_t)bl_x * C + c] = tau_s[c];
        }
    }
}

// =============================================================================================
// kernel 2: one-sided Jacobi SVD of X = R2^H (C x C) in LDS, 256 threads = 16 column pairs x 16 lanes.
//   R2 = Vx Sigma Ux^H;  N = Vx diag(g) Xrot^H with g = s_reg / s   (U2 diag(s_reg) V^H)
// =============================================================================================
__global__ void __launch_bounds__(256) factor_jacobi_kernel(FactorArgs a, size_t bstride) {
    batch_offset(a, bstride);
    __shared__ __attribute__((aligned(16))) cplx Xs[CPMAX][CPMAX + 1];  // Xs[col][row]
    __shared__ __attribute__((aligned(16))) cplx Vs[CPMAX][CPMAX + 1];
    __shared__ double g_s[CPMAX];
    __shared__ double w_s[CPMAX];
    __shared__ double sig_s[CPMAX];
    __shared__ __attribute__((aligned(16))) cplx Ws[2][CPMAX + 1];   // direct route: pivot column and row of a sweep step
    __shared__ double fro_s[2];
    __shared__ int chol_bad;
    const int tid = threadIdx.x;
    const int C = a.C;
    const int Cp = (C + 1) & ~1;
    bool have_v = false;   // Vs holds the rotations of an earlier bin of this run
    // a workgroup walks `jrun` consecutive bins: neighbouring bins have nearly the same singular vectors, so the
    // rotations accumulated for one bin are the starting point of the next (X = R2^H V_prev is already almost
    // orthogonal by columns) and the sweeps drop from ~9 to ~3.  jrun = 1 keeps the bins independent.
    const bool solo = (int)blockIdx.x < a.jsplit;   // (workgroup-uniform)
    const int jrun = solo ? 1 : (a.jrun > 0 ? a.jrun : 1);
    const int bi0 = solo ? (int)blockIdx.x : a.jsplit + ((int)blockIdx.x - a.jsplit) * jrun;
    for (int t = 0; t < jrun; ++t) {
    const int bi = bi0 + t;   // bin slot (workspaces are indexed by it)
    if (bi >= a.nbins) break;
    const int kb = a.kb0 + bi;
    const cplx* R2 = a.R2w + (int64_t)bi * C * C;
    const bool gram = a.route && a.route[kb] != 0;   // R2 holds A = B^H B (full): X = A, Xrot = V Lambda
    // ---- direct route.  The reference clips the singular values at reg_c s_max (1 %).  Where cond(B) <= 1/reg_c nothing
    // is clipped and M = V diag(1/s^2) V^H = (B^H B)^-1: an in-place inverse of the C x C Gram matrix instead of an SVD
    // (all swept bins of BASELINE config 3 qualify: cond 54 at k_cut, < 2 above 5 kHz).  The certificate
    // cond(A) <= ||A||_F ||A^-1||_F <= 1/reg_c^2 is sufficient and rigorous; bins that fail it take the Jacobi route.
    if (gram && a.reg_mode == 0 && a.Mw) {
        if (tid < 2) fro_s[tid] = 0.0;
        if (tid == 0) chol_bad = 0;
        __syncthreads();
        double fa = 0.0;
        for (int idx = tid; idx < CPMAX * CPMAX; idx += 256) {
            const int col = idx / CPMAX, row = idx % CPMAX;
            cplx v = mk(0, 0);
            if (row < C && col < C) v = R2[(int64_t)row * C + col];
            Xs[col][row] = v;   // A[row][col]
            fa += norm2(v);
        }
        fa = wave_sum(fa);
        if ((tid & 63) == 0) atomicAdd(&fro_s[0], fa);
        __syncthreads();
        // in-place inversion of the Hermitian positive definite A by C sweep (Gauss-Jordan) steps without pivoting: every
        // step is one rank-1 update of the whole C x C matrix spread over the 256 threads, two barriers per step
        // (a Cholesky factorisation + triangular inverse has three times the sequential depth at this size)
        cplx* colj = &Ws[0][0];          // column j and row j of the current matrix
        cplx* rowj = &Ws[1][0];
        for (int j = 0; j < C; ++j) {
            const double piv = Xs[j][j].x;
            if (!(piv > 0.0)) { if (tid == 0) chol_bad = 1; }
            const double ip = fast_rcp(piv > 0.0 ? piv : 1.0);
            if (tid < C) { colj[tid] = Xs[j][tid]; rowj[tid] = Xs[tid][j]; }   // A[tid][j], A[j][tid]
            __syncthreads();
            for (int idx = tid; idx < C * C; idx += 256) {
                const int k = idx / C, i = idx - k * C;   // element A[i][k] = Xs[k][i]
                cplx v;
                if (i == j && k == j) v = mk(ip, 0.0);
                else if (i == j) v = mk(rowj[k].x * ip, rowj[k].y * ip);
                else if (k == j) v = mk(-colj[i].x * ip, -colj[i].y * ip);
                else { cplx pr = mk(0, 0); cfma(pr, colj[i], rowj[k]); v = Xs[k][i] - mk(pr.x * ip, pr.y * ip); }
                Xs[k][i] = v;
            }
            __syncthreads();
        }
        // M = A^-1 (now in Xs); its Frobenius norm for the certificate
        cplx* M = a.Mw + (int64_t)bi * C * C;
        cplx mloc[(CPMAX * CPMAX + 255) / 256];
        double fm = 0.0;
#pragma unroll
        for (int u = 0; u < (CPMAX * CPMAX + 255) / 256; ++u) {
            const int idx = tid + 256 * u, aa = idx / C, bb = idx % C;
            cplx acc = mk(0, 0);
            if (idx < C * C) acc = Xs[bb][aa];   // M[aa][bb]
            mloc[u] = acc;
            fm += norm2(acc);
        }
        fm = wave_sum(fm);
        if ((tid & 63) == 0) atomicAdd(&fro_s[1], fm);
        __syncthreads();
        const double thr = 1.0 / (a.reg_c * a.reg_c);
        const bool direct = !chol_bad && fro_s[0] * fro_s[1] <= thr * thr && fro_s[0] > 0.0;   // (||A||_F ||A^-1||_F)^2
        if (direct) {
#pragma unroll
            for (int u = 0; u < (CPMAX * CPMAX + 255) / 256; ++u) {
                const int idx = tid + 256 * u;
                if (idx < C * C) M[idx] = mloc[u];
            }
            if (a.sv && tid < C) {
                // bounds instead of singular values: s_max <= ||A||_F^(1/2), s_min >= ||A^-1||_F^(-1/2)
                a.sv[(int64_t)kb * C + tid] = (tid == 0) ? sqrt(sqrt(fro_s[0])) : 1.0 / sqrt(sqrt(fro_s[1]));
            }
            if (tid == 0) { a.route[kb] = 2; if (a.sweeps_out) a.sweeps_out[kb] = 0; }
            __syncthreads();
            continue;   // next bin of the run
        }
        __syncthreads();
    }
    if (!have_v) {
        for (int idx = tid; idx < CPMAX * CPMAX; idx += 256) {
            const int col = idx / CPMAX, row = idx % CPMAX;  // X[row][col] = conj(R2[col][row]) for col <= row
            cplx v = mk(0, 0);
            if (gram) { if (row < C && col < C) v = R2[(int64_t)row * C + col]; }
            else if (row < C && col <= row) v = conj(R2[(int64_t)col * C + row]);
            Xs[col][row] = v;
            Vs[col][row] = (col == row && col < C) ? mk(1, 0) : mk(0, 0);
        }
    } else {
        // X = R2^H V_prev :  X[row][col] = sum_{m <= row} conj(R2[m][row]) V[m][col]   (Gram form: X = A V_prev)
        // The factor is staged in Xs itself (Ts[m][row]); every thread keeps its four results in registers until all
        // reads are done, so no third C x C buffer is needed (LDS decides how many bins run per CU).
        for (int idx = tid; idx < CPMAX * CPMAX; idx += 256) {
            const int m = idx / CPMAX, row = idx % CPMAX;
            cplx v = mk(0, 0);
            if (gram) { if (row < C && m < C) v = R2[(int64_t)row * C + m]; }
            else if (row < C && m <= row) v = conj(R2[(int64_t)m * C + row]);
            Xs[m][row] = v;
        }
        __syncthreads();
        cplx xnew[CPMAX * CPMAX / 256];
#pragma unroll
        for (int u = 0; u < CPMAX * CPMAX / 256; ++u) {
            const int idx = tid + 256 * u, col = idx / CPMAX, row = idx % CPMAX;
            cplx acc = mk(0, 0);
            if (row < C && col < C)
                for (int m = 0; m < C; ++m) cfma(acc, Xs[m][row], Vs[col][m]);   // (zero above the diagonal in the R2 form)
            xnew[u] = acc;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < CPMAX * CPMAX / 256; ++u) {
            const int idx = tid + 256 * u;
            Xs[idx / CPMAX][idx % CPMAX] = xnew[u];
        }
    }
    __syncthreads();
    {
        constexpr int GL = 16;                // lanes per pair
        constexpr int RL = CPMAX / GL;        // rows per lane (2)
        const int npairs = Cp / 2;
        const int pi = tid / GL, gl = tid % GL;
        const bool jactive = pi < npairs;
        int sweeps = 0;
        for (; sweeps < 60; ++sweeps) {
            int rotated = 0;
            for (int r = 0; r < Cp - 1; ++r) {
                if (jactive) {
                    int p, q;
                    if (pi == 0) { p = Cp - 1; q = r; }
                    else { p = (r + pi) % (Cp - 1); q = (r - pi + (Cp - 1)) % (Cp - 1); }
                    cplx xp[RL], xq[RL], vp[RL], vq[RL];
                    double al = 0.0, be = 0.0;
                    cplx ga = mk(0, 0);
#pragma unroll
                    for (int t = 0; t < RL; ++t) {
                        const int row = gl + GL * t;
                        xp[t] = Xs[p][row];
                        xq[t] = Xs[q][row];
                        vp[t] = Vs[p][row];  // fetched with X: the rotation below then has no LDS latency left
                        vq[t] = Vs[q][row];
                        al += norm2(xp[t]);
                        be += norm2(xq[t]);
                        cfma_conj(ga, xp[t], xq[t]);
                    }
                    al = group_sum<GL>(al);
                    be = group_sum<GL>(be);
                    ga = group_sum<GL>(ga);
                    const double ag2 = norm2(ga);
                    // rotate when |gamma| > eps sqrt(alpha beta)
                    if (ag2 > (2.220446049250313e-16 * 2.220446049250313e-16) * (al * be) && ag2 > 0.0) {
                        const double iag = fast_rsqrt(ag2);           // 1/|gamma|
                        const double zeta = 0.5 * (be - al) * iag;
                        const double z2 = fma(zeta, zeta, 1.0);
                        const double sq = z2 * fast_rsqrt(z2);        // sqrt(1 + zeta^2)
                        const double t_ = (zeta == 0.0) ? 1.0 : copysign(fast_rcp(fabs(zeta) + sq), zeta);
                        const double cs = fast_rsqrt(fma(t_, t_, 1.0)), sn = cs * t_;
                        const cplx ph = mk(ga.x * iag, ga.y * iag);
                        const cplx sph = mk(sn * ph.x, sn * ph.y);          // s * ph
                        const cplx spc = mk(sn * ph.x, -sn * ph.y);         // s * conj(ph)
#pragma unroll
                        for (int t = 0; t < RL; ++t) {
                            const int row = gl + GL * t;
                            Xs[p][row] = cs * xp[t] - spc * xq[t];
                            Xs[q][row] = sph * xp[t] + cs * xq[t];
                            Vs[p][row] = cs * vp[t] - spc * vq[t];
                            Vs[q][row] = sph * vp[t] + cs * vq[t];
                        }
                        rotated = 1;
                    }
                }
                __syncthreads();
            }
            if (!__syncthreads_or(rotated)) { ++sweeps; break; }
        }
        if (a.sweeps_out && tid == 0) a.sweeps_out[kb] = sweeps;
        have_v = true;
    }
    // ---- singular values, regularisation weights
    if (tid < CPMAX) {
        double n2 = 0.0;
        for (int row = 0; row < CPMAX; ++row) n2 += norm2(Xs[tid][row]);
        sig_s[tid] = gram ? sqrt(sqrt(n2)) : sqrt(n2);   // Gram form: the column norms are the eigenvalues s^2
    }
    __syncthreads();
    if (tid < CPMAX) {
        double smax = 0.0;
        for (int i = 0; i < C; ++i) smax = fmax(smax, sig_s[i]);
        const double s = sig_s[tid];
        double g = 0.0;
        if (tid < C && s > 0.0) {
            if (a.reg_mode == 0) {
                g = 1.0 / (fmax(s, a.reg_c * smax) * s);  // s_reg / s,  s_reg = 1/max(s, c smax)
            } else {
                int ex;
                frexp(smax, &ex);                                      // smax = m 2^ex, m in [0.5,1)
                const double tol = a.tol_dim * ldexp(1.0, ex - 53);   // max(size) * eps(smax)  (MATLAB pinv)
                g = (s > tol) ? 1.0 / (s * s) : 0.0;
            }
        }
        g_s[tid] = g;
        // weights of M = V diag(g) V^H with V = Xrot / sigma  (Gram form: V = Xrot / s^2)
        w_s[tid] = (s > 0.0) ? (gram ? g / ((s * s) * (s * s)) : g / (s * s)) : 0.0;
        if (gram && tid == 0) {
            double smin = INFINITY;
            for (int i = 0; i < C; ++i) smin = fmin(smin, sig_s[i]);
            if (!(smax <= 3.0e3 * smin) && a.status) atomicExch(a.status + 2, 1);  // the kr estimate was too optimistic
        }
        if (a.sv && tid < C) a.sv[(int64_t)kb * C + tid] = s;
    }
    __syncthreads();
    // N[a][b] = sum_i Vx[a][i] g_i conj(Xrot[b][i])
    cplx* N = a.Nw + (int64_t)bi * C * C;
    if (!gram)  // (the Gram form has no back-transform)
    for (int idx = tid; idx < C * C; idx += 256) {
        const int aa = idx / C, bb = idx % C;
        cplx acc = mk(0, 0);
        for (int i = 0; i < C; ++i) {
            const cplx t = g_s[i] * Vs[i][aa];
            cfma(acc, t, conj(Xs[i][bb]));
        }
        N[idx] = acc;
    }
    // M = V diag(g) V^H with V = Xrot / sigma:  M[a][b] = sum_i Xrot[a][i] (g_i / sigma_i^2) conj(Xrot[b][i])
    if (a.Mw) {
        cplx* M = a.Mw + (int64_t)bi * C * C;
        for (int idx = tid; idx < C * C; idx += 256) {
            const int aa = idx / C, bb = idx % C;
            cplx acc = mk(0, 0);
            for (int i = 0; i < C; ++i) {
                const cplx t = w_s[i] * Xs[i][aa];
                cfma(acc, t, conj(Xs[i][bb]));
            }
            M[idx] = acc;
        }
    }
    __syncthreads();  // the next bin of the run rewrites Xs
    }
}

// =============================================================================================
// kernel 3: Z_k = conj(Q2 [N; 0]) by applying the stored reflectors backwards; least-squares bins.
// =============================================================================================
template <int NCH, int RPT, int MAXT>
__global__ void __launch_bounds__(MAXT) factor_back_kernel(FactorArgs a, size_t bstride) {
    batch_offset(a, bstride);
    const int tid = threadIdx.x;
    const int c = tid / NCH, ch = tid % NCH;
    const int S = a.S, C = a.C, ldS = a.ldS;
    const int kb = a.kb0 + blockIdx.x;
    if (c >= C) return;
    // Z_k feeds the least-squares bins and the ill-conditioned swept bins only (the sweep uses G_k and M_k)
    if (a.cond_ok && kb >= a.ls_end && a.cond_ok[kb] != 0.0) return;
    const cplx* Vw = a.Vws + (int64_t)blockIdx.x * C * ldS;
    const cplx* N = a.Nw + (int64_t)blockIdx.x * C * C;
    const double* tauw = a.tauw + (int64_t)blockIdx.x * C;
    cplx B[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int s = ch + NCH * i;
        B[i] = (s < C) ? N[(int64_t)s * C + c] : mk(0, 0);
    }
    for (int j = C - 1; j >= 0; --j) {
        const double tau = tauw[j];
        const cplx* vj = Vw + (int64_t)j * ldS;
        if constexpr (RPT <= 16) {
            cplx v[RPT];
            cplx w = mk(0, 0);
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int s = ch + NCH * i;
                v[i] = (s >= j && s < S) ? vj[s] : mk(0, 0);
                cfma_conj(w, v[i], B[i]);
            }
            w = group_sum<NCH>(w);
            w = w * tau;
#pragma unroll
            for (int i = 0; i < RPT; ++i) { cplx p = w * v[i]; B[i] -= p; }
        } else {
            // tall problems: re-read v (L1/L2 resident) in the update pass instead of holding RPT more registers
            cplx w = mk(0, 0);
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int s = ch + NCH * i;
                if (s >= j && s < S) cfma_conj(w, vj[s], B[i]);
            }
            w = group_sum<NCH>(w);
            w = w * tau;
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int s = ch + NCH * i;
                if (s >= j && s < S) { cplx p = w * vj[s]; B[i] -= p; }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int s = ch + NCH * i;
        if (s < S) a.Z[((int64_t)kb * C + c) * ldS + s] = conj(B[i]);
    }
    if (a.Hq && kb < a.ls_end) {
        for (int e = 0; e < 2; ++e) {
            const cplx* hq = a.Hq + e * a.hq_estride + (int64_t)kb * a.ldHq;
            cplx w = mk(0, 0);
#pragma unroll
            for (int i = 0; i < RPT; ++i) {
                const int s = ch + NCH * i;
                if (s < S) { if (a.hq_conj) cfma(w, hq[s], B[i]); else cfma(w, hq[s], conj(B[i])); }
            }
            if (a.hq_conj) w = conj(w);
            w = group_sum<NCH>(w);
            if (ch == 0) a.W[((int64_t)e * a.P + kb) * C + c] = w;
        }
    }
}

// ---------------------------------------------------------------------------------------------
template <typename TT, int NCH, int RPT, int MAXT>
static void launch_one(const FactorArgs& a, int nbins, hipStream_t st, int phases) {
    const int threads = (int)(ceil_div((int64_t)NCH * a.C, 64) * 64);
    if (threads > MAXT) throw Error(2, "factor: too many channels for this row count");
    const size_t dyn = ((size_t)2 * a.ldS + (a.gram_from > 0 ? (size_t)CPMAX * (4 * NCH + 1) : 0)) * sizeof(cplx);
    if (a.nOrders > 96) throw Error(2, "factor: simulation order above 95 is not supported");
    FactorArgs aq = a;
    if (dyn > 140 * 1024) {  // no room for the Gram route's row chunks next to the reflector buffers
        aq.gram_from = 0;
    }
    static bool attr_set = false;
    if (!attr_set) {
        HIP_CHECK(hipFuncSetAttribute((const void*)factor_qr_kernel<TT, NCH, RPT, MAXT>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
        attr_set = true;
    }
    if (phases & 1) {
        factor_qr_kernel<TT, NCH, RPT, MAXT><<<bgrid(nbins), threads, aq.gram_from > 0 ? dyn : (size_t)2 * a.ldS * sizeof(cplx), st>>>(aq, batch_ctx().stride);
        KERNEL_CHECK();
        {
            FactorArgs aj = a;
            aj.nbins = nbins;
            const int jr = aj.jrun > 0 ? aj.jrun : 1;
            aj.jsplit = (jr > 1 && aj.gram_from > aj.kb0) ? std::min(nbins, aj.gram_from - aj.kb0) : 0;
            factor_jacobi_kernel<<<bgrid(aj.jsplit + (nbins - aj.jsplit + jr - 1) / jr), 256, 0, st>>>(aj, batch_ctx().stride);
        }
        KERNEL_CHECK();
    }
    if (phases & 2) {
        factor_back_kernel<NCH, RPT, MAXT><<<bgrid(nbins), threads, 0, st>>>(a, batch_ctx().stride);
        KERNEL_CHECK();
    }
}

template <typename TT>
static void dispatch(const FactorArgs& a, int nbins, hipStream_t st, int phases) {
    const int S = a.S, C = a.C;
    if (C > CPMAX) throw Error(2, "factor: more than 32 output channels is not supported in this build");
    if (S < C) throw Error(2, "factor: fewer rows than channels (under-determined array model) is not supported");
    if (!a.tauw || !a.R2w || !a.Nw || !a.Vws) throw Error(2, "factor: workspaces missing");
    if (C * 32 <= 1024) {
        if (S <= 32 * 1) return launch_one<TT, 32, 1, 1024>(a, nbins, st, phases);
        if (S <= 32 * 4) return launch_one<TT, 32, 4, 1024>(a, nbins, st, phases);
        if (S <= 32 * 8) return launch_one<TT, 32, 8, 1024>(a, nbins, st, phases);
        if (S <= 32 * 13) return launch_one<TT, 32, 13, 1024>(a, nbins, st, phases);
        if (S <= 32 * 16) return launch_one<TT, 32, 16, 1024>(a, nbins, st, phases);
        if (S <= 32 * 24) return launch_one<TT, 32, 24, 1024>(a, nbins, st, phases);
    }
    if (C * 64 <= 512) {
        if (S <= 64 * 24) return launch_one<TT, 64, 24, 512>(a, nbins, st, phases);
        if (S <= 64 * 43) return launch_one<TT, 64, 43, 512>(a, nbins, st, phases);
        if (S <= 64 * 64) return launch_one<TT, 64, 64, 512>(a, nbins, st, phases);
    }
    throw Error(2, "factor: problem shape (rows x channels) not supported in this build");
}

// phases: 1 = QR + Jacobi, 2 = back-transform (+ least-squares bins), 3 = both
void launch_factor(const FactorArgs& a0, int nbins, bool tn_cplx, hipStream_t st, int phases) {
    if (nbins <= 0) return;
    const FactorArgs& a = a0;
    if (a.Tn && !tn_cplx) dispatch<double>(a, nbins, st, phases); else dispatch<cplx>(a, nbins, st, phases);
}

}  // namespace emagls
