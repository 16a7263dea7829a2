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
        const int bstr = a.bn_stride > 0 ? a.bn_stride : a.nOrders;
        for (int n = tid; n < a.nOrders; n += blockDim.x) {
            cplx b = a.bn[(int64_t)kb * bstr + n];
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
__device__ __forceinline__ void factor_jacobi_body(FactorArgs a, const int blk, size_t bstride) {
    batch_offset(a, bstride);
    __shared__ __attribute__((aligned(16))) cplx Xs[CPMAX][CPMAX + 1];  // Xs[col][row]
    __shared__ __attribute__((aligned(16))) cplx Vs[CPMAX][CPMAX + 1];
    __shared__ double g_s[CPMAX];
    __shared__ double w_s[CPMAX];
    __shared__ double sig_s[CPMAX];
    const int tid = threadIdx.x;
    const int C = a.C;
    const int Cp = (C + 1) & ~1;
    bool have_v = false;   // Vs holds the rotations of an earlier bin of this run
    // a workgroup walks `jrun` consecutive bins: neighbouring bins have nearly the same singular vectors, so the
    // rotations accumulated for one bin are the starting point of the next (X = R2^H V_prev is already almost
    // orthogonal by columns) and the sweeps drop from ~9 to ~3.  jrun = 1 keeps the bins independent.
    const bool solo = blk < a.jsplit;   // (workgroup-uniform)
    const int jrun = solo ? 1 : (a.jrun > 0 ? a.jrun : 1);
    const int bi0 = solo ? blk : a.jsplit + (blk - a.jsplit) * jrun;
    for (int t = 0; t < jrun; ++t) {
    const int bi = bi0 + t;   // bin slot (workspaces are indexed by it)
    if (bi >= a.nbins) break;
    const int kb = a.kb0 + bi;
    const cplx* R2 = a.R2w + (int64_t)bi * C * C;
    // Gram-route bins: gram_solve_kernel (gramroute.hip) inverted the well-conditioned ones directly (route 2: nothing left to
    // do) and left A = B^H B (full Hermitian) in R2w for the others (route 1): X = A, Xrot = V Lambda
    if (a.route && a.route[kb] == 2) continue;   // (workgroup-uniform)
    const bool gram = a.route && a.route[kb] != 0;
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
                        // Another sweep is needed only after a rotation that was not already tiny: Jacobi converges quadratically,
                        // so a sweep whose largest |gamma| / sqrt(alpha beta) is below 1e-8 leaves off-diagonal terms of ~1e-16 --
                        // the verification sweep that would follow (25 rounds without a rotation) is skipped.
                        if (ag2 > 1.0e-16 * (al * be)) rotated = 1;
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
            if (!(smax <= a.cond_limit * smin) && a.status) {   // the kr estimate was too optimistic: the host moves the start of the route
                atomicExch(a.status + 2, 1);
                atomicMax(a.status + 3, kb);
            }
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
__global__ void __launch_bounds__(256) factor_jacobi_kernel(FactorArgs a, size_t bstride) { factor_jacobi_body(a, (int)blockIdx.x, bstride); }
// two independent sets of bins in one launch (the Gram-route bins whose 1 % clipping is active and the Householder-route bins of
// one design: each set is as long as its slowest bin's sweeps, so two launches on one stream cost twice that)
__global__ void __launch_bounds__(256) factor_jacobi_pair_kernel(FactorArgs a, FactorArgs b, int na, size_t bstride) {
    if ((int)blockIdx.x < na) factor_jacobi_body(a, (int)blockIdx.x, bstride);
    else factor_jacobi_body(b, (int)blockIdx.x - na, bstride);
}

// =============================================================================================
// kernel 3: Z_k = conj(Q2 [N; 0]) by applying the stored reflectors backwards; least-squares bins.
// =============================================================================================
// Round 5: the reflector of a step comes from LDS.  Every thread of the workgroup needs all of v_j (the C columns of Z_k are updated
// with the same vector), and the form of rounds 1-4 -- each thread loads its rows of v_j from global memory, holds them in registers
// next to its rows of B and was capped at 72 registers so that two workgroups share a CU -- spilled 25 registers inside the loop:
// 430 MB of scratch writes per 16-design launch, 14 us per step, 354 us per launch (profiles/r05_pmc.md).  Now v_j is staged once per
// step (zero outside rows j..S-1, so the passes need no row test), v_(j-1) is requested before step j's arithmetic and stored after
// it, and both passes read the vector from LDS: registers hold B only.
template <int NCH, int RPT, int MAXT>
__global__ void __launch_bounds__(MAXT) factor_back_kernel(FactorArgs a, size_t bstride) {
    batch_offset(a, bstride);
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    cplx* vbuf = reinterpret_cast<cplx*>(dyn);   // [2][ldv]
    __shared__ double tau_s[CPMAX];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int c = tid / NCH, ch = tid % NCH;
    const int S = a.S, C = a.C, ldS = a.ldS;
    const int ldv = max(ldS, NCH * RPT);         // (the passes read NCH * RPT rows: zeros beyond S)
    const int kb = a.kb0 + blockIdx.x;
    // Z_k feeds the least-squares bins and the ill-conditioned swept bins only (the sweep uses G_k and M_k)
    if (a.cond_ok && kb >= a.ls_end && a.cond_ok[kb] != 0.0) return;   // (uniform over the workgroup)
    const bool active = c < C;
    const cplx* Vw = a.Vws + (int64_t)blockIdx.x * C * ldS;
    const cplx* N = a.Nw + (int64_t)blockIdx.x * C * C;
    if (tid < C) tau_s[tid] = a.tauw[(int64_t)blockIdx.x * C + tid];
    cplx B[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int s = ch + NCH * i;
        B[i] = (active && s < C) ? N[(int64_t)s * C + c] : mk(0, 0);
    }
    // staged elements per thread and step through registers (requested a step ahead); rows beyond NST * threads -- few channels on
    // many rows -- are staged by a plain loop after the step's arithmetic
    constexpr int NST = (NCH * RPT + MAXT - 1) / MAXT < 4 ? (NCH * RPT + MAXT - 1) / MAXT : 4;
    auto stage_load = [&](int j, cplx (&r)[NST]) __attribute__((always_inline)) {
        const cplx* vj = Vw + (int64_t)j * ldS;
#pragma unroll
        for (int q = 0; q < NST; ++q) {
            const int s = tid + nthr * q;
            r[q] = (s >= j && s < S) ? vj[s] : mk(0, 0);
        }
    };
    auto stage_store = [&](int j, const cplx (&r)[NST]) __attribute__((always_inline)) {
        cplx* dst = vbuf + (size_t)(j & 1) * ldv;
#pragma unroll
        for (int q = 0; q < NST; ++q) {
            const int s = tid + nthr * q;
            if (s < ldv) dst[s] = r[q];
        }
    };
    // rows beyond NST * threads (tall problems on few threads): staged by a plain loop
    auto stage_rest = [&](int j) __attribute__((always_inline)) {
        const cplx* vj = Vw + (int64_t)j * ldS;
        cplx* dst = vbuf + (size_t)(j & 1) * ldv;
        for (int s = tid + nthr * NST; s < ldv; s += nthr) dst[s] = (s >= j && s < S) ? vj[s] : mk(0, 0);
    };
    {
        cplx r[NST];
        stage_load(C - 1, r);
        stage_store(C - 1, r);
        stage_rest(C - 1);
    }
    for (int j = C - 1; j >= 0; --j) {
        __syncthreads();   // v_j is in its buffer; the other buffer's readers (step j + 1) are done
        cplx r[NST];
        if (j > 0) stage_load(j - 1, r);
        const cplx* v = vbuf + (size_t)(j & 1) * ldv + ch;
        cplx w = mk(0, 0);
        // (a few rows at a time: the scheduler would otherwise request all RPT values of the vector at once, next to the RPT rows of B)
        constexpr int CH = RPT > 8 ? 4 : 8;
#pragma unroll
        for (int i0 = 0; i0 < RPT; i0 += CH) {
#pragma unroll
            for (int i = i0; i < (i0 + CH < RPT ? i0 + CH : RPT); ++i) cfma_conj(w, v[NCH * i], B[i]);
            if constexpr (RPT > 8) __builtin_amdgcn_sched_barrier(0);
        }
        w = group_sum<NCH>(w);
        w = w * tau_s[j];
        // (tall problems: the update pass reads the vector from LDS again instead of holding RPT more values next to B)
        if constexpr (RPT > 8) asm volatile("" ::: "memory");
#pragma unroll
        for (int i0 = 0; i0 < RPT; i0 += CH) {
#pragma unroll
            for (int i = i0; i < (i0 + CH < RPT ? i0 + CH : RPT); ++i) { cplx p = w * v[NCH * i]; B[i] -= p; }
            if constexpr (RPT > 8) __builtin_amdgcn_sched_barrier(0);
        }
        if (j > 0) { stage_store(j - 1, r); stage_rest(j - 1); }
    }
    if (!active) return;
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

// The form of rounds 1-4 (every thread reads its rows of v_j from global memory): kept for the tall problems (more than 256 rows:
// config 4's large radii), where the rows of B alone fill the registers a 1024-thread workgroup may have and the compiler spills
// less around global loads than around the staged form's LDS reads (EMAGLS_BACK_LDS=1 takes the staged form there too).
template <int NCH, int RPT, int MAXT>
__global__ void __launch_bounds__(MAXT)
__attribute__((amdgpu_waves_per_eu(4, 8))) factor_back_global_kernel(FactorArgs a, size_t bstride) {
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
    const size_t dyn = (size_t)2 * a.ldS * sizeof(cplx);
    if (a.nOrders > 96) throw Error(2, "factor: simulation order above 95 is not supported");
    if (phases & 1) {
        factor_qr_kernel<TT, NCH, RPT, MAXT><<<bgrid(nbins), threads, dyn, st>>>(a, batch_ctx().stride);
        KERNEL_CHECK();
        if (!(phases & 8)) {   // (8: the caller launches the Jacobi step itself, launch_factor_jacobi_pair)
            FactorArgs aj = a;
            aj.nbins = nbins;
            aj.jsplit = nbins;   // one workgroup per bin: the full SVDs of the ill-conditioned bins are the long pole
            aj.jrun = 1;
            factor_jacobi_kernel<<<bgrid(nbins), 256, 0, st>>>(aj, batch_ctx().stride);
            KERNEL_CHECK();
        }
    }
    if (phases & 2) {
        static const bool lds_tall = [] { const char* e = getenv("EMAGLS_BACK_LDS"); return e && e[0] == '1'; }();
        if (RPT <= 8 || lds_tall) {
            static PerDeviceOnce back_once;
            if (back_once.first())
                HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(factor_back_kernel<NCH, RPT, MAXT>), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
            const size_t dyn_b = (size_t)2 * std::max(a.ldS, NCH * RPT) * sizeof(cplx);
            factor_back_kernel<NCH, RPT, MAXT><<<bgrid(nbins), threads, dyn_b, st>>>(a, batch_ctx().stride);
        } else {
            factor_back_global_kernel<NCH, RPT, MAXT><<<bgrid(nbins), threads, 0, st>>>(a, batch_ctx().stride);
        }
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

void launch_factor_jacobi_gram(const FactorArgs& a, int nbins, hipStream_t st) {
    if (nbins <= 0) return;
    if (a.C > CPMAX) throw Error(2, "factor: more than 32 output channels is not supported in this build");
    FactorArgs aj = a;
    aj.nbins = nbins;
    const int jr = aj.jrun > 0 ? aj.jrun : 1;
    aj.jsplit = 0;   // runs of jr neighbouring bins per workgroup (warm start); bins that took the direct route are skipped
    factor_jacobi_kernel<<<bgrid((nbins + jr - 1) / jr), 256, 0, st>>>(aj, batch_ctx().stride);
    KERNEL_CHECK();
}

// The Jacobi steps of the Gram-route bins (launch_factor_jacobi_gram's arguments) and of the Householder-route bins (launch_factor's,
// after its QR with phases = 1 | 8) as one launch.
void launch_factor_jacobi_pair(const FactorArgs& gram, int nb_gram, const FactorArgs& hh, int nb_hh, hipStream_t st) {
    if (gram.C > CPMAX || hh.C > CPMAX) throw Error(2, "factor: more than 32 output channels is not supported in this build");
    FactorArgs ag = gram, ah = hh;
    ag.nbins = nb_gram;
    const int jr = ag.jrun > 0 ? ag.jrun : 1;
    ag.jsplit = 0;
    const int na = (nb_gram + jr - 1) / jr;
    ah.nbins = nb_hh; ah.jsplit = nb_hh; ah.jrun = 1;
    factor_jacobi_pair_kernel<<<bgrid(na + nb_hh), 256, 0, st>>>(ag, ah, na, batch_ctx().stride);
    KERNEL_CHECK();
}

// phases: 1 = QR + Jacobi (| 8: QR only), 2 = back-transform (+ least-squares bins), 3 = both
void launch_factor(const FactorArgs& a0, int nbins, bool tn_cplx, hipStream_t st, int phases) {
    if (nbins <= 0) return;
    const FactorArgs& a = a0;
    if (a.Tn && !tn_cplx) dispatch<double>(a, nbins, st, phases); else dispatch<cplx>(a, nbins, st, phases);
}

}  // namespace emagls
