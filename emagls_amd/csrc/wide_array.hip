// eMagLS / eMagLS2 with 33..64 output channels or microphones (e.g. a 64-capsule array): a plain path for the widths the tuned
// per-bin kernels (32-column register tiles, factor.hip / gramroute.hip / sweep_persist.hip) do not hold.
//
// Reference per bin (lib/getEMagLs2Filters.m:84-99): pwGrid = smairMat(:,:,k) * Y_Hi_conj; [U,s,V] = svd(pwGrid.','econ');
// s = 1 ./ max(s, 0.01*max(s)); Y_reg_inv = conj(U) * (s .* V.'), then least squares below k_cut and the phase recurrence above.
//
// Every bin takes the orthonormal S-space route of factor.hip, in global memory instead of registers:
//   pwGrid.' = Q B_k,  Q = conj(Y) R^-1 (Cholesky-QR of the HRIR-grid SH matrix),  B_k = R diag(b_n(k)) E^T = sum_n b_n(k) T_n (S x C)
//   B_k = Q2 R2                      Householder QR, one workgroup per bin (wa_qr_kernel)
//   R2^H J = A (columns orthogonal)  one-sided Jacobi in LDS on the ROWS of R2 -- the form that is accurate for the row-graded
//                                    B_k, s_min / s_max down to 1e-13 (wa_jacobi_kernel);  s_j = |a_j|
//   N = conj(J) diag(s_reg / s) A^T  (C x C),  Z_k = conj(Q2 [conj(N); 0])  (back-transform, wa_back_kernel)
//   Y_reg_inv_k = conj(Q) Z_k        (D x C, materialised for every bin: wa_yri_kernel)
// and the sweep runs one launch per bin on G_k = pwGrid_k.' (dspace.hip) and Y_reg_inv_k (wide.hip: sweep_wide_kernel).
// Slow next to the 32-channel pipeline (tens of milliseconds per design), exact in the same sense.
#include "kernels.hpp"

namespace emagls {

namespace {

constexpr int WA_CMAX = 64;

__device__ __forceinline__ double block_sum(double v, double* red) {   // all threads of the workgroup receive the sum
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < nw; ++w) t += red[w];
    return t;
}

// B[kb][c][s] = sum_n b_n(kb) Tn[n][c][s]   (T_n is zero for s >= (n+1)^2; Nyquist: real(b_n), getSMAIRMatrix.m:115-117)
__global__ void __launch_bounds__(256) wa_assemble_kernel(const double* __restrict__ Tn, const cplx* __restrict__ bn, int nOrd, int S, int C, int ldS,
                                                          int P, int kb0, cplx* __restrict__ B) {
    const int kb = kb0 + blockIdx.x, c = blockIdx.y;
    __shared__ cplx bs[96];
    for (int n = threadIdx.x; n < nOrd; n += 256) {
        cplx b = bn[(int64_t)kb * nOrd + n];
        if (kb == P - 1) b.y = 0.0;
        bs[n] = b;
    }
    __syncthreads();
    cplx* out = B + ((int64_t)blockIdx.x * C + c) * ldS;
    for (int s = threadIdx.x; s < ldS; s += 256) {
        cplx acc = mk(0.0, 0.0);
        if (s < S) {
            int n0 = (int)sqrt((double)s);
            while ((n0 + 1) * (n0 + 1) <= s) ++n0;
            while (n0 > 0 && n0 * n0 > s) --n0;
            for (int n = n0; n < nOrd; ++n) cfma(acc, bs[n], Tn[((int64_t)n * C + c) * ldS + s]);
        }
        out[s] = acc;
    }
}

// Householder QR of the S x C matrix of one bin (columns B[c][0..S)), in place: R2 (upper, row major [i][k]) to R2w, the
// reflector vectors v_j (unnormalised, entries j..S-1) to Vw[c = j], tau_j = 2 / |v_j|^2.
__global__ void __launch_bounds__(512) wa_qr_kernel(cplx* __restrict__ B, cplx* __restrict__ Vw, int S, int C, int ldS, double* __restrict__ tauw,
                                                    cplx* __restrict__ R2w) {
    __shared__ double red[8];
    __shared__ cplx s_alpha;
    __shared__ double s_tau;
    cplx* Bk = B + (int64_t)blockIdx.x * C * ldS;
    cplx* Vk = Vw + (int64_t)blockIdx.x * C * ldS;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int j = 0; j < C; ++j) {
        cplx* aj = Bk + (int64_t)j * ldS;
        double n2 = 0.0;
        for (int s = j + tid; s < S; s += 512) n2 += norm2(aj[s]);
        n2 = block_sum(n2, red);
        if (tid == 0) {
            const cplx x0 = aj[j];
            const double nx = sqrt(n2), ax = cabs(x0);
            cplx alpha = mk(-nx, 0.0);
            if (ax > 0.0) alpha = mk(-x0.x / ax * nx, -x0.y / ax * nx);
            const double nv2 = 2.0 * nx * (nx + ax);          // |x - alpha e_1|^2
            s_alpha = alpha;
            s_tau = nv2 > 0.0 ? 2.0 / nv2 : 0.0;
        }
        __syncthreads();
        const cplx alpha = s_alpha;
        const double tau = s_tau;
        cplx* vj = Vk + (int64_t)j * ldS;
        for (int s = j + tid; s < S; s += 512) {
            cplx v = aj[s];
            if (s == j) v = v - alpha;
            vj[s] = v;
        }
        if (tid == 0) tauw[(int64_t)blockIdx.x * C + j] = tau;
        __syncthreads();   // v_j is complete (global memory, this workgroup)
        // columns k > j: a_k -= tau v (v^H a_k); a wave per column
        for (int k = j + 1 + wave; k < C; k += 8) {
            cplx* ak = Bk + (int64_t)k * ldS;
            cplx w = mk(0.0, 0.0);
            for (int s = j + lane; s < S; s += 64) cfma_conj(w, vj[s], ak[s]);
            w = wave_sum(w);
            w = mk(w.x * tau, w.y * tau);
            for (int s = j + lane; s < S; s += 64) { cplx t = mk(0.0, 0.0); cfma(t, vj[s], w); ak[s] = ak[s] - t; }
        }
        // column j itself: alpha on the diagonal, zeros below
        for (int s = j + tid; s < S; s += 512) aj[s] = s == j ? alpha : mk(0.0, 0.0);
        __syncthreads();
    }
    cplx* R2 = R2w + (int64_t)blockIdx.x * C * C;
    for (int idx = tid; idx < C * C; idx += 512) {
        const int i = idx / C, k = idx % C;
        R2[idx] = (i <= k && i < S) ? Bk[(int64_t)k * ldS + i] : mk(0.0, 0.0);
    }
}

// One-sided Jacobi on M = R2^H (C x C, columns of M = conjugated rows of R2): M J = A with orthogonal columns a_j = s_j u_j.
// 32 disjoint column pairs per round (round-robin tournament, C - 1 rounds per sweep), 32 lanes per pair.
// Output N = conj(J) diag(s_reg / s) A^T (C x C, row major) and the singular values.
// flag2: another sweep follows one in which some rotation had |gamma|^2 > flag2 alpha beta (1e-28: the rule of rounds 3-5)
__global__ void __launch_bounds__(1024) wa_jacobi_kernel(const cplx* __restrict__ R2w, int C, double reg_c, cplx* __restrict__ Nw, double* __restrict__ sv,
                                                         int* __restrict__ sweeps_out, double flag2) {
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    const int Cp = (C + 1) & ~1, ld = Cp + 1;            // even column count (a zero column pads an odd C)
    cplx* A = reinterpret_cast<cplx*>(dyn);               // [Cp][ld]  A[col][row]
    cplx* J = A + (size_t)Cp * ld;                        // [Cp][ld]  J[col][row]
    __shared__ int s_rot;
    __shared__ double s_s[WA_CMAX + 2];
    const cplx* R2 = R2w + (int64_t)blockIdx.x * C * C;
    const int tid = threadIdx.x;
    for (int idx = tid; idx < Cp * ld; idx += 1024) { A[idx] = mk(0.0, 0.0); J[idx] = mk(0.0, 0.0); }
    __syncthreads();
    for (int idx = tid; idx < C * C; idx += 1024) {
        const int col = idx / C, row = idx % C;          // M[row][col] = conj(R2[col][row])
        A[col * ld + row] = conj(R2[col * C + row]);
    }
    for (int i = tid; i < Cp; i += 1024) J[i * ld + i] = mk(1.0, 0.0);
    __syncthreads();
    const int pr = tid >> 5, l32 = tid & 31;             // pair index within the round, lane within the pair
    const int npairs = Cp / 2;
    int sweeps = 0;
    for (int sw = 0; sw < 60; ++sw) {
        if (tid == 0) s_rot = 0;
        __syncthreads();
        for (int rnd = 0; rnd < Cp - 1; ++rnd) {
            if (pr < npairs) {
                // positions pr and Cp - 1 - pr of the tournament order: position 0 is fixed, the others rotate
                auto player = [&](int pos) { return pos == 0 ? 0 : 1 + ((pos - 1 - rnd) % (Cp - 1) + (Cp - 1)) % (Cp - 1); };
                int p = player(pr), q = player(Cp - 1 - pr);
                if (p > q) { const int t = p; p = q; q = t; }
                cplx* ap = A + p * ld; cplx* aq = A + q * ld;
                cplx* jp = J + p * ld; cplx* jq = J + q * ld;
                // (Cp <= 64: at most two rows per lane; J's rows are requested with A's, so the rotation has no LDS round trip left)
                const int r0 = l32, r1 = l32 + 32;
                const bool one = r0 < Cp, two = r1 < Cp;
                const cplx x0 = one ? ap[r0] : mk(0.0, 0.0), y0 = one ? aq[r0] : mk(0.0, 0.0);
                const cplx u0 = one ? jp[r0] : mk(0.0, 0.0), v0 = one ? jq[r0] : mk(0.0, 0.0);
                const cplx x1 = two ? ap[r1] : mk(0.0, 0.0), y1 = two ? aq[r1] : mk(0.0, 0.0);
                const cplx u1 = two ? jp[r1] : mk(0.0, 0.0), v1 = two ? jq[r1] : mk(0.0, 0.0);
                double al = norm2(x0) + norm2(x1), be = norm2(y0) + norm2(y1);
                cplx ga = mk(0.0, 0.0);
                cfma_conj(ga, x0, y0); cfma_conj(ga, x1, y1);
                al = group_sum<32>(al); be = group_sum<32>(be); ga = group_sum<32>(ga);
                const double ag2 = norm2(ga), ab = al * be;
                // rotate when |gamma| > 1e-15 sqrt(alpha beta)  (reciprocals and roots from the hardware seeds + Newton steps, as in
                // factor_jacobi_body: the IEEE expansions of the seven divisions and roots were the longest part of a round)
                if (ag2 > 0.0 && ag2 > 1e-30 * ab) {
                    const double iag = fast_rsqrt(ag2);           // 1 / |gamma|
                    const cplx ph = mk(ga.x * iag, ga.y * iag);
                    const double zeta = 0.5 * (be - al) * iag;
                    const double z2 = fma(zeta, zeta, 1.0);
                    const double sq = z2 * fast_rsqrt(z2);        // sqrt(1 + zeta^2)
                    const double t = (zeta >= 0.0 ? 1.0 : -1.0) * fast_rcp(fabs(zeta) + sq);
                    const double c = fast_rsqrt(fma(t, t, 1.0)), s = c * t;
                    const cplx sph = mk(s * ph.x, s * ph.y), sphc = mk(s * ph.x, -s * ph.y);   // s e^{i phi}, s e^{-i phi}
                    auto rot = [&](const cplx& x, const cplx& y, cplx& nx, cplx& ny) __attribute__((always_inline)) {
                        cplx t1 = mk(0.0, 0.0), t2 = mk(0.0, 0.0);
                        cfma(t1, sphc, y); cfma(t2, sph, x);
                        nx = mk(c * x.x, c * x.y) - t1; ny = t2 + mk(c * y.x, c * y.y);
                    };
                    cplx nx, ny;
                    if (one) {
                        rot(x0, y0, nx, ny); ap[r0] = nx; aq[r0] = ny;
                        rot(u0, v0, nx, ny); jp[r0] = nx; jq[r0] = ny;
                    }
                    if (two) {
                        rot(x1, y1, nx, ny); ap[r1] = nx; aq[r1] = ny;
                        rot(u1, v1, nx, ny); jp[r1] = nx; jq[r1] = ny;
                    }
                    if (l32 == 0 && ag2 > flag2 * ab) s_rot = 1;
                }
            }
            __syncthreads();
        }
        sweeps = sw + 1;
        const int again = s_rot;
        __syncthreads();
        if (!again) break;
    }
    // singular values and their regularised inverses
    for (int j = tid >> 5; j < Cp; j += 32) {
        double n2 = 0.0;
        for (int r = l32; r < Cp; r += 32) n2 += norm2(A[j * ld + r]);
        n2 = group_sum<32>(n2);
        if (l32 == 0) s_s[j] = sqrt(n2);
    }
    __syncthreads();
    double smax = 0.0;
    for (int j = 0; j < C; ++j) smax = fmax(smax, s_s[j]);
    // (a padded zero column ends as a column of the null space: it is simply not used below -- only the C real columns count,
    //  and for an odd C the pad column never mixes with the others because its dot products are exactly zero)
    cplx* N = Nw + (int64_t)blockIdx.x * C * C;
    for (int idx = tid; idx < C * C; idx += 1024) {
        const int i = idx / C, k = idx % C;   // N[i][k] = sum_j conj(J[i][j]) (s_reg_j / s_j) A[k][j]   (J[row i][col j] = J[j*ld + i])
        cplx acc = mk(0.0, 0.0);
        for (int j = 0; j < Cp; ++j) {
            const double sj = s_s[j];
            if (!(sj > 0.0)) continue;
            const double w = 1.0 / (fmax(sj, reg_c * smax) * sj);
            const cplx jc = conj(J[j * ld + i]), a = A[j * ld + k];
            cplx t = mk(0.0, 0.0);
            cfma(t, jc, a);
            acc.x += w * t.x; acc.y += w * t.y;
        }
        N[idx] = acc;
    }
    if (sv) for (int j = tid; j < C; j += 1024) sv[(int64_t)blockIdx.x * C + j] = s_s[j];
    if (sweeps_out && tid == 0) sweeps_out[blockIdx.x] = sweeps;
}

// Z_k = conj(Q2 [conj(N); 0]): the reflectors applied in reverse to X = [conj(N); 0] (S x C), then the conjugate.  Z[kb][c][s].
__global__ void __launch_bounds__(512) wa_back_kernel(const cplx* __restrict__ Vw, const double* __restrict__ tauw, const cplx* __restrict__ Nw, int S, int C,
                                                      int ldS, cplx* __restrict__ Z) {
    const cplx* Vk = Vw + (int64_t)blockIdx.x * C * ldS;
    const cplx* N = Nw + (int64_t)blockIdx.x * C * C;
    cplx* Zk = Z + (int64_t)blockIdx.x * C * ldS;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    for (int idx = tid; idx < C * ldS; idx += 512) {
        const int c = idx / ldS, s = idx % ldS;
        Zk[idx] = (s < C && s < S) ? conj(N[s * C + c]) : mk(0.0, 0.0);   // X[s][c] = conj(N[s][c])
    }
    __syncthreads();
    for (int j = C - 1; j >= 0; --j) {
        const cplx* vj = Vk + (int64_t)j * ldS;
        const double tau = tauw[(int64_t)blockIdx.x * C + j];
        for (int k = wave; k < C; k += 8) {
            cplx* xk = Zk + (int64_t)k * ldS;
            cplx w = mk(0.0, 0.0);
            for (int s = j + lane; s < S; s += 64) cfma_conj(w, vj[s], xk[s]);
            w = wave_sum(w);
            w = mk(w.x * tau, w.y * tau);
            for (int s = j + lane; s < S; s += 64) { cplx t = mk(0.0, 0.0); cfma(t, vj[s], w); xk[s] = xk[s] - t; }
        }
        __syncthreads();
    }
    for (int idx = tid; idx < C * ldS; idx += 512) Zk[idx] = conj(Zk[idx]);
}

// ---------------------------------------------------------------------------------------------
// Register-resident forms of the two Householder kernels for S <= 64 NR rows (NR = 7: simulation order <= 20).  The forms above
// walk every column through L2 for every reflector (two dependent passes per column and reflector: 5.7 and 11.8 ms for the 513
// bins of the 64-capsule design); here a wave keeps its four columns in registers, lane l holding the rows l, l + 64, ...
//   wa_back_reg_kernel   columns are independent: no barrier at all; two workgroups of 8 waves per bin (32 columns each)
//   wa_qr_reg_kernel     16 waves = 64 columns; the owner of column j forms v_j and hands it to the others through LDS
//                        (two buffers: one barrier per reflector)
// ---------------------------------------------------------------------------------------------
template <int NR>
__global__ void __launch_bounds__(512) wa_back_reg_kernel(const cplx* __restrict__ Vw, const double* __restrict__ tauw, const cplx* __restrict__ Nw, int S,
                                                          int C, int ldS, cplx* __restrict__ Z) {
    const cplx* Vk = Vw + (int64_t)blockIdx.x * C * ldS;
    const cplx* N = Nw + (int64_t)blockIdx.x * C * C;
    cplx* Zk = Z + (int64_t)blockIdx.x * C * ldS;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int kbase = 32 * blockIdx.y + wave;      // this wave's columns: kbase + 8 q
    cplx x[4][NR];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = kbase + 8 * q;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int s = lane + 64 * r;
            x[q][r] = (k < C && s < C && s < S) ? conj(N[s * C + k]) : mk(0.0, 0.0);   // X[s][k] = conj(N[s][k])
        }
    }
    for (int j = C - 1; j >= 0; --j) {
        const cplx* vj = Vk + (int64_t)j * ldS;
        const double tau = tauw[(int64_t)blockIdx.x * C + j];
        cplx v[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int s = lane + 64 * r;
            v[r] = (s >= j && s < S) ? vj[s] : mk(0.0, 0.0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            cplx w = mk(0.0, 0.0);
#pragma unroll
            for (int r = 0; r < NR; ++r) cfma_conj(w, v[r], x[q][r]);
            w = wave_sum(w);
            w = mk(-w.x * tau, -w.y * tau);
#pragma unroll
            for (int r = 0; r < NR; ++r) cfma(x[q][r], v[r], w);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = kbase + 8 * q;
        if (k < C) {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int s = lane + 64 * r;
                if (s < ldS) Zk[(int64_t)k * ldS + s] = conj(x[q][r]);
            }
        }
    }
}

template <int NR>
__global__ void __launch_bounds__(1024) wa_qr_reg_kernel(const cplx* __restrict__ B, cplx* __restrict__ Vw, int S, int C, int ldS,
                                                         double* __restrict__ tauw, cplx* __restrict__ R2w) {
    __shared__ __attribute__((aligned(16))) cplx vbuf[2][64 * NR];
    __shared__ double s_tau[2];
    const cplx* Bk = B + (int64_t)blockIdx.x * C * ldS;
    cplx* Vk = Vw + (int64_t)blockIdx.x * C * ldS;
    cplx* R2 = R2w + (int64_t)blockIdx.x * C * C;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    cplx x[4][NR];                                  // columns wave + 16 q
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = wave + 16 * q;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int s = lane + 64 * r;
            x[q][r] = (k < C && s < S) ? Bk[(int64_t)k * ldS + s] : mk(0.0, 0.0);
        }
    }
    for (int j = 0; j < C; ++j) {
        const int buf = j & 1;
        if (wave == (j & 15)) {   // the owner: |a_j(j:)|, alpha, v_j = a_j(j:) - alpha e_1, tau_j = 2 / |v_j|^2
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q == (j >> 4)) {
                    double n2 = 0.0;
                    cplx x0 = mk(0.0, 0.0);
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        const int s = lane + 64 * r;
                        if (s >= j) n2 += norm2(x[q][r]);
                        if (s == j) x0 = x[q][r];
                    }
                    n2 = wave_sum(n2);
                    x0 = wave_sum(x0);               // (one lane holds it, the others zero)
                    const double nx = sqrt(n2), ax = cabs(x0);
                    cplx alpha = mk(-nx, 0.0);
                    if (ax > 0.0) alpha = mk(-x0.x / ax * nx, -x0.y / ax * nx);
                    const double nv2 = 2.0 * nx * (nx + ax);          // |x - alpha e_1|^2
                    const double tau = nv2 > 0.0 ? 2.0 / nv2 : 0.0;
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        const int s = lane + 64 * r;
                        cplx v = s >= j ? x[q][r] : mk(0.0, 0.0);
                        if (s == j) v = v - alpha;
                        vbuf[buf][s] = v;
                        if (s >= j && s < S) Vk[(int64_t)j * ldS + s] = v;
                        if (s == j) x[q][r] = alpha; else if (s > j) x[q][r] = mk(0.0, 0.0);
                    }
                    if (lane == 0) { s_tau[buf] = tau; tauw[(int64_t)blockIdx.x * C + j] = tau; }
                }
            }
        }
        __syncthreads();
        const double tau = s_tau[buf];
        // columns k > j: a_k -= tau v (v^H a_k)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = wave + 16 * q;
            if (k > j && k < C) {
                cplx w = mk(0.0, 0.0);
#pragma unroll
                for (int r = 0; r < NR; ++r) cfma_conj(w, vbuf[buf][lane + 64 * r], x[q][r]);
                w = wave_sum(w);
                w = mk(-w.x * tau, -w.y * tau);
#pragma unroll
                for (int r = 0; r < NR; ++r) cfma(x[q][r], vbuf[buf][lane + 64 * r], w);
            }
        }
    }
    // R2 (upper, row major [i][k]); the rest of the square zero
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = wave + 16 * q;
        if (k < C) {
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int i = lane + 64 * r;
                if (i < C) R2[i * C + k] = (i <= k && i < S) ? x[q][r] : mk(0.0, 0.0);
            }
        }
    }
}

// Yri[kb][c][d] = sum_s conj(Q[d][s]) Z[kb][c][s]   (Q real: the real-arithmetic pipeline).  A workgroup: 64 directions x all
// channels of one bin, S walked in chunks of 32 through LDS.
__global__ void __launch_bounds__(256) wa_yri_kernel(const double* __restrict__ Q, int64_t ldQ, const cplx* __restrict__ Z, int S, int C, int ldS, int D,
                                                     int64_t ldD, cplx* __restrict__ Yri) {
    __shared__ double qs[32][65];
    __shared__ cplx zs[WA_CMAX][33];
    const int kbi = blockIdx.y, d0 = blockIdx.x * 64;
    const cplx* Zk = Z + (int64_t)kbi * C * ldS;
    const int tid = threadIdx.x, dl = tid & 63, cg = tid >> 6;     // channels cg, cg + 4, ...
    cplx acc[WA_CMAX / 4];
#pragma unroll
    for (int i = 0; i < WA_CMAX / 4; ++i) acc[i] = mk(0.0, 0.0);
    for (int s0 = 0; s0 < S; s0 += 32) {
        __syncthreads();
        for (int idx = tid; idx < 32 * 64; idx += 256) {
            const int dd = idx >> 5, ss = idx & 31;
            qs[ss][dd] = (d0 + dd < D && s0 + ss < S) ? Q[(int64_t)(d0 + dd) * ldQ + s0 + ss] : 0.0;
        }
        for (int idx = tid; idx < C * 32; idx += 256) {
            const int c = idx >> 5, ss = idx & 31;
            zs[c][ss] = (s0 + ss < S) ? Zk[(int64_t)c * ldS + s0 + ss] : mk(0.0, 0.0);
        }
        __syncthreads();
#pragma unroll 4
        for (int ss = 0; ss < 32; ++ss) {
            const double q = qs[ss][dl];
#pragma unroll
            for (int i = 0; i < WA_CMAX / 4; ++i) {
                const int c = cg + 4 * i;
                if (c < C) cfma(acc[i], q, zs[c][ss]);
            }
        }
    }
    if (d0 + dl < D)
#pragma unroll
        for (int i = 0; i < WA_CMAX / 4; ++i) {
            const int c = cg + 4 * i;
            if (c < C) Yri[((int64_t)kbi * C + c) * ldD + d0 + dl] = acc[i];
        }
}

// The same product on the FP64 matrix cores (v_mfma_f64_16x16x4): the complex rows of Z_k as 2 C real rows against the real Q.
// Workgroup = 4 waves = 128 real rows (all channels) x 64 directions of one bin, S walked in chunks of 32 through LDS (both
// operands K-major there: As[k][row], Bs[k][direction]); a wave holds 32 rows x 64 directions = 2 x 4 tiles.  A 16-row tile
// carries 8 channels with the rows ordered so that a lane's four result registers are (re, im) of channel kk and (re, im) of
// channel 4 + kk: row 4 reg + kk  <->  channel 4 (reg >> 1) + kk, part reg & 1 -- the results leave as whole complex numbers.
// (the scalar form above reads one LDS operand per FMA pair: 7.8 TFLOP/s, 18 ms for the 64-capsule design)
constexpr int WY_KC = 32, WY_LDA = 132, WY_LDB = 68;
typedef double wa_double4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) wa_yri_mfma_kernel(const double* __restrict__ Q, int64_t ldQ, const cplx* __restrict__ Z, int S, int C, int ldS,
                                                          int D, int64_t ldD, cplx* __restrict__ Yri) {
    __shared__ __attribute__((aligned(16))) double As[WY_KC][WY_LDA];
    __shared__ __attribute__((aligned(16))) double Bs[WY_KC][WY_LDB];
    const int kbi = blockIdx.y, d0 = blockIdx.x * 64;
    const cplx* Zk = Z + (int64_t)kbi * C * ldS;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, ii = lane & 15, kk = lane >> 4;
    wa_double4 acc[2][4];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] = wa_double4{0.0, 0.0, 0.0, 0.0};
    // the next chunk of both operands is requested before the matrix instructions of the current one and stored behind them (round 6:
    // load -> barrier -> 64 MFMAs -> barrier per chunk left the pipe idle for a global round trip in every chunk)
    constexpr int NA = WA_CMAX * WY_KC / 256, NB = 64 * WY_KC / 256;
    cplx za[NA];
    double qb[NB];
    auto fetch = [&](int s0) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int idx = tid + 256 * u, c = idx >> 5, ss = idx & 31;
            za[u] = (c < C && s0 + ss < S) ? Zk[(int64_t)c * ldS + s0 + ss] : mk(0.0, 0.0);
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int idx = tid + 256 * u, j = idx >> 5, ss = idx & 31;
            qb[u] = (d0 + j < D && s0 + ss < S) ? Q[(int64_t)(d0 + j) * ldQ + s0 + ss] : 0.0;
        }
    };
    fetch(0);
    for (int s0 = 0; s0 < S; s0 += WY_KC) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int idx = tid + 256 * u, c = idx >> 5, ss = idx & 31;
            const int row = 16 * (c >> 3) + 8 * ((c & 7) >> 2) + (c & 3);   // the (re) row of channel c; its (im) row is 4 further
            As[ss][row] = za[u].x;
            As[ss][row + 4] = za[u].y;
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int idx = tid + 256 * u, j = idx >> 5, ss = idx & 31;
            Bs[ss][j] = qb[u];
        }
        __syncthreads();
        if (s0 + WY_KC < S) fetch(s0 + WY_KC);
#pragma unroll
        for (int k0 = 0; k0 < WY_KC; k0 += 4) {
            const double a0 = As[k0 + kk][32 * wave + ii], a1 = As[k0 + kk][32 * wave + 16 + ii];
            double b[4];
#pragma unroll
            for (int y = 0; y < 4; ++y) b[y] = Bs[k0 + kk][16 * y + ii];
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                acc[0][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b[y], acc[0][y], 0, 0, 0);
                acc[1][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b[y], acc[1][y], 0, 0, 0);
            }
        }
    }
    // C / D layout: column = lane & 15, row = (lane >> 4) + 4 reg
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const int c0 = 8 * (2 * wave + x) + kk;
#pragma unroll
        for (int y = 0; y < 4; ++y) {
            const int d = d0 + 16 * y + ii;
            if (d < D) {
                if (c0 < C) Yri[((int64_t)kbi * C + c0) * ldD + d] = mk(acc[x][y][0], acc[x][y][1]);
                if (c0 + 4 < C) Yri[((int64_t)kbi * C + c0 + 4) * ldD + d] = mk(acc[x][y][2], acc[x][y][3]);
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Tall problems (more rows than the register forms above hold: EMAinSH orders 5..7 and matched ATF matrices factor D-long columns --
// 2702 rows --, arrays at large radii S > 448; round 6).  The plain forms walk every column twice per reflector with one load in flight
// per lane: 14.5 ms (QR) and 30.3 ms (back-transform) for the 256 bins of an order-6 EMAinSH design.  Here the reflector of a step
// lies in LDS (zero outside its rows, so no row tests), and a column is loaded ONCE per step with all its loads in flight:
//   wa_back_tall_kernel  the columns of X are independent: a wave keeps ONE column in registers through all reflectors (NRT rows per
//                        lane), a workgroup = 8 columns of a bin; v_{j-1} is requested before step j's arithmetic
//   wa_qr_tall_kernel    a wave takes the trailing columns of a step in turn: whole column -> registers, dot, update, store
// ---------------------------------------------------------------------------------------------
constexpr int WA_TALL = 48;     // rows per lane: up to 3072 rows
template <int NRT>
__global__ void __launch_bounds__(512) wa_back_tall_kernel(const cplx* __restrict__ Vw, const double* __restrict__ tauw, const cplx* __restrict__ Nw, int S,
                                                           int C, int ldS, cplx* __restrict__ Z) {
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    cplx* vs = reinterpret_cast<cplx*>(dyn);              // [64 NRT]
    __shared__ double tau_s[WA_CMAX];
    const cplx* Vk = Vw + (int64_t)blockIdx.x * C * ldS;
    const cplx* N = Nw + (int64_t)blockIdx.x * C * C;
    cplx* Zk = Z + (int64_t)blockIdx.x * C * ldS;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c = 8 * blockIdx.y + wave;
    const bool active = c < C;
    if (tid < C) tau_s[tid] = tauw[(int64_t)blockIdx.x * C + tid];
    cplx x[NRT];
#pragma unroll
    for (int i = 0; i < NRT; ++i) {
        const int s = lane + 64 * i;
        x[i] = (active && s < C && s < S) ? conj(N[s * C + c]) : mk(0.0, 0.0);   // X[s][c] = conj(N[s][c])
    }
    constexpr int NST = 64 * NRT / 512;   // staged values per thread and step
    cplx r[NST];
    auto fetch = [&](int j) __attribute__((always_inline)) {
        const cplx* vj = Vk + (int64_t)j * ldS;
#pragma unroll
        for (int q = 0; q < NST; ++q) {
            const int s = tid + 512 * q;
            r[q] = (s >= j && s < S) ? vj[s] : mk(0.0, 0.0);
        }
    };
    fetch(C - 1);
    for (int j = C - 1; j >= 0; --j) {
        __syncthreads();   // (the readers of step j + 1 are done)
#pragma unroll
        for (int q = 0; q < NST; ++q) vs[tid + 512 * q] = r[q];
        __syncthreads();
        if (j > 0) fetch(j - 1);
        // (four rows of the reflector at a time: the scheduler would otherwise request all NRT values next to the NRT rows of x and spill)
        cplx w0 = mk(0.0, 0.0), w1 = mk(0.0, 0.0);
#pragma unroll
        for (int i0 = 0; i0 < NRT; i0 += 4) {
#pragma unroll
            for (int i = i0; i < i0 + 4; i += 2) { cfma_conj(w0, vs[lane + 64 * i], x[i]); cfma_conj(w1, vs[lane + 64 * (i + 1)], x[i + 1]); }
            __builtin_amdgcn_sched_barrier(0);
        }
        cplx w = wave_sum(w0 + w1);
        const double tau = tau_s[j];
        w = mk(w.x * tau, w.y * tau);
#pragma unroll
        for (int i0 = 0; i0 < NRT; i0 += 4) {
#pragma unroll
            for (int i = i0; i < i0 + 4; ++i) { cplx t = mk(0.0, 0.0); cfma(t, vs[lane + 64 * i], w); x[i] = x[i] - t; }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (!active) return;
#pragma unroll
    for (int i = 0; i < NRT; ++i) {
        const int s = lane + 64 * i;
        if (s < ldS) Zk[(int64_t)c * ldS + s] = s < S ? conj(x[i]) : mk(0.0, 0.0);
    }
}

template <int NRT>
__global__ void __launch_bounds__(512) wa_qr_tall_kernel(cplx* __restrict__ B, cplx* __restrict__ Vw, int S, int C, int ldS, double* __restrict__ tauw,
                                                         cplx* __restrict__ R2w) {
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    cplx* vs = reinterpret_cast<cplx*>(dyn);              // [64 NRT]  v_j, zero outside rows j .. S-1
    __shared__ double red[8];
    __shared__ cplx s_alpha;
    __shared__ double s_tau;
    cplx* Bk = B + (int64_t)blockIdx.x * C * ldS;
    cplx* Vk = Vw + (int64_t)blockIdx.x * C * ldS;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    constexpr int NST = 64 * NRT / 512;
    for (int j = 0; j < C; ++j) {
        cplx* aj = Bk + (int64_t)j * ldS;
        // column j (as the earlier steps left it) into LDS, its norm on the way
        cplx r[NST];
        double n2 = 0.0;
#pragma unroll
        for (int q = 0; q < NST; ++q) {
            const int s = tid + 512 * q;
            r[q] = (s >= j && s < S) ? aj[s] : mk(0.0, 0.0);
            n2 += norm2(r[q]);
        }
#pragma unroll
        for (int q = 0; q < NST; ++q) vs[tid + 512 * q] = r[q];
        n2 = block_sum(n2, red);   // (contains the barriers that complete vs)
        if (tid == 0) {
            const cplx x0 = vs[j];
            const double nx = sqrt(n2), ax = cabs(x0);
            cplx alpha = mk(-nx, 0.0);
            if (ax > 0.0) alpha = mk(-x0.x / ax * nx, -x0.y / ax * nx);
            const double nv2 = 2.0 * nx * (nx + ax);          // |x - alpha e_1|^2
            s_alpha = alpha;
            s_tau = nv2 > 0.0 ? 2.0 / nv2 : 0.0;
            vs[j] = x0 - alpha;
            tauw[(int64_t)blockIdx.x * C + j] = s_tau;
        }
        __syncthreads();
        const cplx alpha = s_alpha;
        const double tau = s_tau;
        // v_j to memory (the back-transform reads it), column j itself: alpha on the diagonal, zeros below
        cplx* vj = Vk + (int64_t)j * ldS;
#pragma unroll
        for (int q = 0; q < NST; ++q) {
            const int s = tid + 512 * q;
            if (s >= j && s < S) { vj[s] = vs[s]; aj[s] = s == j ? alpha : mk(0.0, 0.0); }
        }
        // columns k > j: a_k -= tau v (v^H a_k); a wave per column, the column once through registers
        for (int k = j + 1 + wave; k < C; k += 8) {
            cplx* ak = Bk + (int64_t)k * ldS;
            cplx a[NRT];
#pragma unroll
            for (int i = 0; i < NRT; ++i) {
                const int s = lane + 64 * i;
                a[i] = (s >= j && s < S) ? ak[s] : mk(0.0, 0.0);
            }
            __builtin_amdgcn_sched_barrier(0);
            cplx w0 = mk(0.0, 0.0), w1 = mk(0.0, 0.0);
#pragma unroll
            for (int i0 = 0; i0 < NRT; i0 += 8) {
#pragma unroll
                for (int i = i0; i < i0 + 8; i += 2) { cfma_conj(w0, vs[lane + 64 * i], a[i]); cfma_conj(w1, vs[lane + 64 * (i + 1)], a[i + 1]); }
                __builtin_amdgcn_sched_barrier(0);
            }
            cplx w = wave_sum(w0 + w1);
            w = mk(w.x * tau, w.y * tau);
#pragma unroll
            for (int i0 = 0; i0 < NRT; i0 += 8) {
#pragma unroll
                for (int i = i0; i < i0 + 8; ++i) {
                    const int s = lane + 64 * i;
                    if (s >= j && s < S) { cplx t = mk(0.0, 0.0); cfma(t, vs[s], w); ak[s] = a[i] - t; }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();   // (vs is rewritten by the next step; the trailing columns are complete in memory for this workgroup)
    }
    cplx* R2 = R2w + (int64_t)blockIdx.x * C * C;
    for (int idx = tid; idx < C * C; idx += 512) {
        const int i = idx / C, k = idx % C;
        R2[idx] = (i <= k && i < S) ? Bk[(int64_t)k * ldS + i] : mk(0.0, 0.0);
    }
}

// least-squares rows: W[e][kb][c] = sum_d Hc[e][kb][d] Yri[kb][c][d]   (kb < n_c)
__global__ void __launch_bounds__(256) wa_ls_kernel(const cplx* __restrict__ Hc, int64_t ldH, int n_c, const cplx* __restrict__ Yri, int64_t ldD, int D, int C,
                                                    int P, int kb_first, cplx* __restrict__ W) {
    const int kb = kb_first + blockIdx.x, e = blockIdx.y;
    const cplx* h = Hc + ((int64_t)e * n_c + kb) * ldH;
    const cplx* Y = Yri + (int64_t)(kb - kb_first) * C * ldD;
    const int part = threadIdx.x & 7;
    for (int c = threadIdx.x >> 3; c < C; c += 32) {
        cplx acc = mk(0.0, 0.0);
        for (int d = part; d < D; d += 8) cfma(acc, h[d], Y[(int64_t)c * ldD + d]);
        acc = group_sum<8>(acc);
        if (part == 0) W[((int64_t)e * P + kb) * C + c] = acc;
    }
}

// E = pinv(Y_lo) Y_mic with pinv(Y_lo) = (Y_lo^T Y_lo)^-1 Y_lo^T (real basis, full column rank certified by the caller):
//   Ag[c][c'] = sum_m Ycm[c][m] Ycm[c'][m]   (Ycm = [S][M], rows = SH channels)
__global__ void wa_lo_gram_kernel(const double* __restrict__ Ycm, int M, int nOut, double* __restrict__ Ag) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nOut * nOut) return;
    const int c = idx / nOut, c2 = idx % nOut;
    double acc = 0.0;
    for (int m = 0; m < M; ++m) acc = fma(Ycm[(int64_t)c * M + m], Ycm[(int64_t)c2 * M + m], acc);
    Ag[idx] = acc;
}
//   E[c][s] = sum_c' Minv[c][c'] sum_m Ycm[c'][m] Ycm[s][m]
__global__ void wa_e_kernel(const double* __restrict__ Ycm, int M, int nOut, int S, const cplx* __restrict__ Minv, double* __restrict__ E, int ldE) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
    if (s >= S) return;
    double acc = 0.0;
    for (int c2 = 0; c2 < nOut; ++c2) {
        double g = 0.0;
        for (int m = 0; m < M; ++m) g = fma(Ycm[(int64_t)c2 * M + m], Ycm[(int64_t)s * M + m], g);
        acc = fma(Minv[(int64_t)c * nOut + c2].x, g, acc);
    }
    E[(int64_t)c * ldE + s] = acc;
}

}  // namespace

void launch_wa_assemble(const void* Tn, const void* bn, int nOrd, int S, int C, int ldS, int P, int kb0, int nbins, void* B, hipStream_t st) {
    if (nbins <= 0) return;
    if (nOrd > 96) throw Error(2, "wide array path: simulation order above 95");
    wa_assemble_kernel<<<dim3(nbins, C), 256, 0, st>>>((const double*)Tn, (const cplx*)bn, nOrd, S, C, ldS, P, kb0, (cplx*)B);
    KERNEL_CHECK();
}
void launch_wa_factor(void* B, void* Vw, int S, int C, int ldS, int nbins, double reg_c, double* tauw, void* R2w, void* Nw, double* sv, int* sweeps,
                      void* Z, hipStream_t st) {
    if (nbins <= 0) return;
    if (C > WA_CMAX) throw Error(2, "wide array path: at most 64 channels");
    // (fewer simulated SH channels than microphones: pwGrid has rank S < C, and the reference's clipped inverse then carries 100 / s_max
    // times left singular vectors that LAPACK's rounding alone determines -- its own result changes with the SVD driver.  The path
    // for up to 32 channels returns the part that is determined; this one refuses.)
    if (S < C) throw Error(2, "33..64 channels with fewer simulated SH channels than channels (rank-deficient array model: the reference's clipped inverse is rounding noise there) is not supported");
    const char* e_reg = getenv("EMAGLS_WA_REG");   // =0: the forms that walk the columns through L2
    const bool reg7 = ldS <= 64 * 7 && !(e_reg && e_reg[0] == '0');
    const char* e_tall = getenv("EMAGLS_WA_TALL");   // =0: the plain forms for the tall problems
    const bool tall = !reg7 && S <= 64 * WA_TALL && !(e_tall && e_tall[0] == '0');
    const size_t dyn_tall = sizeof(cplx) * 64 * WA_TALL;
    if (tall) {
        static PerDeviceOnce tall_once;
        if (tall_once.first()) {
            HIP_CHECK(hipFuncSetAttribute((const void*)wa_qr_tall_kernel<WA_TALL>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
            HIP_CHECK(hipFuncSetAttribute((const void*)wa_back_tall_kernel<WA_TALL>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        }
    }
    if (reg7) wa_qr_reg_kernel<7><<<nbins, 1024, 0, st>>>((const cplx*)B, (cplx*)Vw, S, C, ldS, tauw, (cplx*)R2w);
    else if (tall) wa_qr_tall_kernel<WA_TALL><<<nbins, 512, dyn_tall, st>>>((cplx*)B, (cplx*)Vw, S, C, ldS, tauw, (cplx*)R2w);
    else wa_qr_kernel<<<nbins, 512, 0, st>>>((cplx*)B, (cplx*)Vw, S, C, ldS, tauw, (cplx*)R2w);
    KERNEL_CHECK();
    const int Cp = (C + 1) & ~1;
    const size_t dyn = sizeof(cplx) * (size_t)2 * Cp * (Cp + 1);
    static PerDeviceOnce attr_once;
    if (attr_once.first()) HIP_CHECK(hipFuncSetAttribute((const void*)wa_jacobi_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    static const double flag2 = [] { const char* e = getenv("EMAGLS_WA_JACOBI_FLAG"); const double f = e ? atof(e) : 1e-14; return f * f; }();
    wa_jacobi_kernel<<<nbins, 1024, dyn, st>>>((const cplx*)R2w, C, reg_c, (cplx*)Nw, sv, sweeps, flag2);
    KERNEL_CHECK();
    if (reg7) wa_back_reg_kernel<7><<<dim3(nbins, (unsigned)ceil_div(C, 32)), 512, 0, st>>>((const cplx*)Vw, tauw, (const cplx*)Nw, S, C, ldS, (cplx*)Z);
    else if (tall) wa_back_tall_kernel<WA_TALL><<<dim3(nbins, (unsigned)ceil_div(C, 8)), 512, dyn_tall, st>>>((const cplx*)Vw, tauw, (const cplx*)Nw, S, C, ldS, (cplx*)Z);
    else wa_back_kernel<<<nbins, 512, 0, st>>>((const cplx*)Vw, tauw, (const cplx*)Nw, S, C, ldS, (cplx*)Z);
    KERNEL_CHECK();
}
void launch_wa_yri(const void* Q, int64_t ldQ, const void* Z, int S, int C, int ldS, int D, int64_t ldD, int nbins, void* Yri, hipStream_t st) {
    if (nbins <= 0) return;
    const char* e = getenv("EMAGLS_WA_YRI_MFMA");   // =0: the scalar form
    if (e && e[0] == '0') wa_yri_kernel<<<dim3((unsigned)ceil_div(D, 64), nbins), 256, 0, st>>>((const double*)Q, ldQ, (const cplx*)Z, S, C, ldS, D, ldD, (cplx*)Yri);
    else wa_yri_mfma_kernel<<<dim3((unsigned)ceil_div(D, 64), nbins), 256, 0, st>>>((const double*)Q, ldQ, (const cplx*)Z, S, C, ldS, D, ldD, (cplx*)Yri);
    KERNEL_CHECK();
}
void launch_wa_ls(const void* Hc, int64_t ldH, int n_c, const void* Yri, int64_t ldD, int D, int C, int P, int kb_first, int kb_end, void* W, hipStream_t st) {
    if (kb_end <= kb_first) return;
    wa_ls_kernel<<<dim3(kb_end - kb_first, 2), 256, 0, st>>>((const cplx*)Hc, ldH, n_c, (const cplx*)Yri, ldD, D, C, P, kb_first, (cplx*)W);
    KERNEL_CHECK();
}
void launch_wa_lo_gram(const void* Ycm, int M, int nOut, double* Ag, hipStream_t st) {
    wa_lo_gram_kernel<<<(unsigned)ceil_div(nOut * nOut, 256), 256, 0, st>>>((const double*)Ycm, M, nOut, Ag);
    KERNEL_CHECK();
}
void launch_wa_e(const void* Ycm, int M, int nOut, int S, const void* Minv, void* E, int ldE, hipStream_t st) {
    wa_e_kernel<<<dim3((unsigned)ceil_div(S, 256), nOut), 256, 0, st>>>((const double*)Ycm, M, nOut, S, (const cplx*)Minv, (double*)E, ldE);
    KERNEL_CHECK();
}

}  // namespace emagls
