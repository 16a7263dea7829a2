// The sequential magnitude-least-squares phase sweep (one launch per frequency bin) and the
// small dense products around it.
//
// Reference (lib/getEMagLsFilters.m:95-103):
//     phi = angle(W(k-1,:) * pwGrid);   W(k,:) = (abs(H(k,:)) .* exp(1i*phi)) * Y_reg_inv;
// With pwGrid.' = Q B_k and Y_reg_inv = conj(Q) Z_k:
//     z = W(k-1,:) B_k^T (1xS) ; p = Q z ; t = |H| p/|p| ; u = t conj(Q) (1xS) ; W(k,:) = u Z_k.
// exp(1i*angle(p)) is evaluated as p/|p| (1 when p == 0, as angle(0) = 0); Nyquist takes real(t).
// Each workgroup owns a slab of directions; its partial W(k,:) goes to a [nWG][2][C] buffer that the
// next launch sums first (the launch boundary is the grid-wide barrier and the release/acquire).
#include "kernels.hpp"

namespace emagls {

constexpr int SW_NT = 512;  // threads per workgroup (8 waves)


__device__ __forceinline__ cplx unit_phase_times(double h, cplx p, bool nyquist) {
    const double ap = cabs(p);
    cplx t = (ap > 0.0) ? mk(h * (p.x / ap), h * (p.y / ap)) : mk(h, 0.0);
    if (nyquist) t.y = 0.0;
    return t;
}

// sum the previous launch's partials into Wp[e*C+c] (LDS); workgroup 0 publishes W(kb-1,:)
__device__ __forceinline__ void gather_prev(cplx* Wp, const cplx* Wpart_prev, cplx* W, int nWG, int C, int P, int kb,
                                            bool first) {
    const int tid = threadIdx.x;
    const int pair = tid >> 3, part = tid & 7;
    if (pair < 2 * C) {
        const int e = pair / C, c = pair % C;
        cplx acc = mk(0, 0);
        if (first) {
            if (part == 0) acc = W[((int64_t)e * P + (kb - 1)) * C + c];
        } else {
            for (int w = part; w < nWG; w += 8) acc += Wpart_prev[((int64_t)w * 2 + e) * C + c];
        }
        acc = group_sum<8>(acc);
        if (part == 0) {
            Wp[pair] = acc;
            if (blockIdx.x == 0 && !first) W[((int64_t)e * P + (kb - 1)) * C + c] = acc;
        }
    }
}

template <typename TQ, int RS>
__global__ void __launch_bounds__(SW_NT) sweep_factored_kernel(SweepArgs a, int kb) {
    __shared__ __attribute__((aligned(16))) cplx Wp[64];
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    cplx* zs = reinterpret_cast<cplx*>(dyn);  // [2][ldS]
    cplx* us = zs + 2 * a.ldS;                // [2][ldS]
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int S = a.S, C = a.C, ldS = a.ldS;
    const bool nyq = (kb == a.P - 1);
    const cplx* Wprev = a.Wpart + (int64_t)((kb - 1) & 1) * a.nWG * 2 * C;
    cplx* Wout = a.Wpart + (int64_t)(kb & 1) * a.nWG * 2 * C;

    gather_prev(Wp, Wprev, a.W, a.nWG, C, a.P, kb, kb == a.kfirst);
    __syncthreads();
    // z[e][s] = sum_c W(k-1)[e][c] B_k[c][s]
    {
        const cplx* Bk = a.Bk + (int64_t)kb * C * ldS;
        for (int s = tid; s < S; s += SW_NT) {
            cplx z0 = mk(0, 0), z1 = mk(0, 0);
            for (int c = 0; c < C; ++c) {
                const cplx b = Bk[(int64_t)c * ldS + s];
                cfma(z0, Wp[c], b);
                cfma(z1, Wp[C + c], b);
            }
            zs[s] = z0;
            zs[ldS + s] = z1;
        }
    }
    __syncthreads();
    cplx z0[RS], z1[RS], u0[RS], u1[RS];
#pragma unroll
    for (int i = 0; i < RS; ++i) {
        const int s = lane + 64 * i;
        z0[i] = (s < S) ? zs[s] : mk(0, 0);
        z1[i] = (s < S) ? zs[ldS + s] : mk(0, 0);
        u0[i] = mk(0, 0);
        u1[i] = mk(0, 0);
    }
    const TQ* Q = reinterpret_cast<const TQ*>(a.Q);
    const int64_t d0 = (int64_t)blockIdx.x * a.dpw;
    const int64_t na = a.P - a.kabs0;
    const double* HaL = a.Habs + ((int64_t)0 * na + (kb - a.kabs0)) * a.ldD;
    const double* HaR = a.Habs + ((int64_t)1 * na + (kb - a.kabs0)) * a.ldD;
    for (int dd = wave; dd < a.dpw; dd += SW_NT / 64) {
        const int64_t d = d0 + dd;
        if (d >= a.D) break;
        TQ q[RS];
        cplx p0 = mk(0, 0), p1 = mk(0, 0);
#pragma unroll
        for (int i = 0; i < RS; ++i) {
            const int s = lane + 64 * i;
            q[i] = (s < S) ? Q[d * a.ldQ + s] : zero_of<TQ>();
            cfma(p0, z0[i], q[i]);
            cfma(p1, z1[i], q[i]);
        }
        p0 = group_sum<64>(p0);
        p1 = group_sum<64>(p1);
        const cplx t0 = unit_phase_times(HaL[d], p0, nyq);
        const cplx t1 = unit_phase_times(HaR[d], p1, nyq);
#pragma unroll
        for (int i = 0; i < RS; ++i) {
            const TQ qc = conj(q[i]);
            cfma(u0[i], t0, qc);
            cfma(u1[i], t1, qc);
        }
    }
    // deterministic cross-wave accumulation of u into LDS
    for (int w = 0; w < SW_NT / 64; ++w) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < RS; ++i) {
                const int s = lane + 64 * i;
                if (s < S) {
                    if (w == 0) { us[s] = u0[i]; us[ldS + s] = u1[i]; }
                    else { us[s] += u0[i]; us[ldS + s] += u1[i]; }
                }
            }
        }
        __syncthreads();
    }
    // partial W(k,:) = u Z_k
#pragma unroll
    for (int i = 0; i < RS; ++i) {
        const int s = lane + 64 * i;
        u0[i] = (s < S) ? us[s] : mk(0, 0);
        u1[i] = (s < S) ? us[ldS + s] : mk(0, 0);
    }
    const cplx* Zk = a.Z + (int64_t)kb * C * ldS;
    for (int c = wave; c < C; c += SW_NT / 64) {
        cplx w0 = mk(0, 0), w1 = mk(0, 0);
#pragma unroll
        for (int i = 0; i < RS; ++i) {
            const int s = lane + 64 * i;
            const cplx zv = (s < S) ? Zk[(int64_t)c * ldS + s] : mk(0, 0);
            cfma(w0, u0[i], zv);
            cfma(w1, u1[i], zv);
        }
        w0 = group_sum<64>(w0);
        w1 = group_sum<64>(w1);
        if (lane == 0) {
            Wout[((int64_t)blockIdx.x * 2 + 0) * C + c] = w0;
            Wout[((int64_t)blockIdx.x * 2 + 1) * C + c] = w1;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// dense sweep: pwGrid_k given directly as X[kb][c][d] and Y_reg_inv_k as Zd[kb][c][d]
// (strides may be 0: plain MagLS uses the fixed Y_conj / Y_pinv, lib/getMagLsFilters.m:64-72)
// ---------------------------------------------------------------------------------------------

constexpr int DS_NT = 256;
constexpr int DS_DPW = 64;

template <typename TX>
__global__ void __launch_bounds__(DS_NT) sweep_dense_kernel(DenseSweepArgs a, int kb) {
    __shared__ __attribute__((aligned(16))) cplx Wp[64];
    __shared__ __attribute__((aligned(16))) cplx ts[2][DS_DPW];
    const int tid = threadIdx.x;
    const int C = a.C;
    const bool nyq = (kb == a.P - 1);
    const cplx* Wprev = a.Wpart + (int64_t)((kb - 1) & 1) * a.nWG * 2 * C;
    cplx* Wout = a.Wpart + (int64_t)(kb & 1) * a.nWG * 2 * C;
    gather_prev(Wp, Wprev, a.W, a.nWG, C, a.P, kb, kb == a.kfirst);
    __syncthreads();
    const TX* X = reinterpret_cast<const TX*>(a.X) + (int64_t)kb * a.x_stride;
    const TX* Zd = reinterpret_cast<const TX*>(a.Zd) + (int64_t)kb * a.z_stride;
    const int64_t d0 = (int64_t)blockIdx.x * DS_DPW;
    const int64_t na = a.P - a.kabs0;
    if (tid < 2 * DS_DPW) {
        const int e = tid / DS_DPW, dd = tid % DS_DPW;
        const int64_t d = d0 + dd;
        cplx t = mk(0, 0);
        if (d < a.D) {
            cplx p = mk(0, 0);
            for (int c = 0; c < C; ++c) cfma(p, Wp[e * C + c], X[(int64_t)c * a.ldD + d]);
            t = unit_phase_times(a.Habs[((int64_t)e * na + (kb - a.kabs0)) * a.ldH + d], p, nyq);
        }
        ts[e][dd] = t;
    }
    __syncthreads();
    // partial W(k,:)[e][c] = sum_{d in slab} t[e][d] Zd[c][d] ; 4 lanes per (e,c)
    const int pair = tid >> 2, part = tid & 3;
    if (pair < 2 * C) {
        const int e = pair / C, c = pair % C;
        cplx acc = mk(0, 0);
        for (int dd = part; dd < DS_DPW; dd += 4) {
            const int64_t d = d0 + dd;
            if (d < a.D) cfma(acc, ts[e][dd], Zd[(int64_t)c * a.ldD + d]);
        }
        acc = group_sum<4>(acc);
        if (part == 0) Wout[((int64_t)blockIdx.x * 2 + e) * C + c] = acc;
    }
}

// after the last swept bin: W(P-1,:) = sum of partials
__global__ void __launch_bounds__(SW_NT) sweep_finalize_kernel(const cplx* __restrict__ Wpart, cplx* __restrict__ W,
                                                               int nWG, int C, int P, int kb_last) {
    __shared__ cplx Wp[64];
    const cplx* Wprev = Wpart + (int64_t)(kb_last & 1) * nWG * 2 * C;
    gather_prev(Wp, Wprev, W, nWG, C, P, kb_last + 1, false);
}

// ---------------------------------------------------------------------------------------------
// Hq[e][kb][s] = sum_d Hc[e][kb][d] conj(Q[d][s])   for the least-squares bins
// ---------------------------------------------------------------------------------------------
template <typename TQ>
__global__ void __launch_bounds__(256) hq_kernel(const cplx* __restrict__ Hc, int64_t ldD, int n_c, const TQ* __restrict__ Q,
                                                 int64_t ldQ, int D, int S, int kb_lo, cplx* __restrict__ Hq, int ldS) {
    const int kb = kb_lo + blockIdx.x, e = blockIdx.y;
    const cplx* h = Hc + ((int64_t)e * n_c + kb) * ldD;
    for (int s = blockIdx.z * blockDim.x + threadIdx.x; s < S; s += gridDim.z * blockDim.x) {
        cplx acc = mk(0, 0);
        for (int d = 0; d < D; ++d) cfma(acc, h[d], conj(Q[(int64_t)d * ldQ + s]));
        Hq[((int64_t)e * n_c + kb) * ldS + s] = acc;
    }
}

// Ypinv[c][d] = sum_s conj(Q[d][s]) Zb[c][s]      (pinv(Y_conj) = conj(Q) Z_B, lib/getLsFilters.m:31)
template <typename TQ>
__global__ void __launch_bounds__(256) ypinv_kernel(const TQ* __restrict__ Q, int64_t ldQ, const cplx* __restrict__ Zb, int ldS,
                                                    int D, int S, int C, TQ* __restrict__ Ypinv, int64_t ldD) {
    const int64_t d = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= D) return;
    for (int c = 0; c < C; ++c) {
        cplx acc = mk(0, 0);
        for (int s = 0; s < S; ++s) cfma(acc, conj(Q[d * ldQ + s]), Zb[(int64_t)c * ldS + s]);
        if constexpr (sizeof(TQ) == sizeof(double)) Ypinv[(int64_t)c * ldD + d] = acc.x; else Ypinv[(int64_t)c * ldD + d] = acc;
    }
}

// W[e][kb][c] = sum_d Hc[e][kb][d] Zfix[c][d]      (bins below the cut of plain MagLS)
template <typename TZ>
__global__ void __launch_bounds__(256) ls_apply_kernel(const cplx* __restrict__ Hc, int64_t ldH, int n_c, const TZ* __restrict__ Zf,
                                                       int64_t ldD, int D, int C, int P, cplx* __restrict__ W) {
    const int kb = blockIdx.x, e = blockIdx.y;
    const int tid = threadIdx.x;
    const cplx* h = Hc + ((int64_t)e * n_c + kb) * ldH;
    const int pair = tid >> 3, part = tid & 7;  // 32 channels x 8 lanes
    if (pair < C) {
        cplx acc = mk(0, 0);
        for (int d = part; d < D; d += 8) cfma(acc, h[d], Zf[(int64_t)pair * ldD + d]);
        acc = group_sum<8>(acc);
        if (part == 0) W[((int64_t)e * P + kb) * C + pair] = acc;
    }
}

// LS filters: w[e][c*L + n] = sum_d h_e[d*L + n] Ypinv[c][d]   (lib/getLsFilters.m:33-34)
template <typename TZ>
__global__ void __launch_bounds__(256) ls_filters_kernel(const double* __restrict__ hL, const double* __restrict__ hR, int64_t L,
                                                         int D, const TZ* __restrict__ Yp, int64_t ldD, TZ* __restrict__ wL,
                                                         TZ* __restrict__ wR) {
    const int c = blockIdx.x, e = blockIdx.y;
    const double* h = e ? hR : hL;
    TZ* w = e ? wR : wL;
    for (int64_t n = blockIdx.z * blockDim.x + threadIdx.x; n < L; n += (int64_t)gridDim.z * blockDim.x) {
        TZ acc = zero_of<TZ>();
        for (int d = 0; d < D; ++d) cfma(acc, Yp[(int64_t)c * ldD + d], h[(int64_t)d * L + n]);
        w[(int64_t)c * L + n] = acc;
    }
}

// elementwise conj copy (X = conj(Y) for the plain MagLS sweep) and real->complex widening
template <typename T>
__global__ void conj_copy_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = conj(in[i]);
}
template <typename T>
__global__ void widen_kernel(const T* __restrict__ in, int64_t ldi, cplx* __restrict__ out, int64_t ldo, int rows, int cols,
                             int transpose, int upper_only) {
    // out[r][c] = in[r][c] (or in[c][r] when transpose) as complex; rows x cols of OUT
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (int64_t)rows * cols;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(idx / cols), c = (int)(idx % cols);
        const int ir = transpose ? c : r, ic = transpose ? r : c;
        cplx v = to_cplx(in[(int64_t)ir * ldi + ic]);
        if (upper_only && ir > ic) v = mk(0, 0);
        out[(int64_t)r * ldo + c] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
template <typename TQ>
static void sweep_factored_dispatch(const SweepArgs& a, int kb, hipStream_t st) {
    const size_t dyn = (size_t)4 * a.ldS * sizeof(cplx);
    const int rs = (a.S + 63) / 64;
#define EMAGLS_SWEEP_CASE(R)                                                                         \
    sweep_factored_kernel<TQ, R><<<a.nWG, SW_NT, dyn, st>>>(a, kb)
    if (rs <= 2) EMAGLS_SWEEP_CASE(2);
    else if (rs <= 4) EMAGLS_SWEEP_CASE(4);
    else if (rs <= 7) EMAGLS_SWEEP_CASE(7);
    else if (rs <= 8) EMAGLS_SWEEP_CASE(8);
    else if (rs <= 12) EMAGLS_SWEEP_CASE(12);
    else throw Error(2, "sweep: more than 768 SH channels is not supported in this build");
#undef EMAGLS_SWEEP_CASE
    KERNEL_CHECK();
}

void launch_sweep_factored(const SweepArgs& a, int kb, bool q_cplx, hipStream_t st) {
    if (q_cplx) sweep_factored_dispatch<cplx>(a, kb, st); else sweep_factored_dispatch<double>(a, kb, st);
}

void launch_sweep_dense(const DenseSweepArgs& a, int kb, bool x_cplx, hipStream_t st) {
    if (x_cplx) sweep_dense_kernel<cplx><<<a.nWG, DS_NT, 0, st>>>(a, kb);
    else sweep_dense_kernel<double><<<a.nWG, DS_NT, 0, st>>>(a, kb);
    KERNEL_CHECK();
}
int dense_sweep_nwg(int D) { return (D + DS_DPW - 1) / DS_DPW; }

void launch_sweep_finalize(const void* Wpart, void* W, int nWG, int C, int P, int kb_last, hipStream_t st) {
    sweep_finalize_kernel<<<1, SW_NT, 0, st>>>((const cplx*)Wpart, (cplx*)W, nWG, C, P, kb_last);
    KERNEL_CHECK();
}

void launch_hq(const void* Hc, int64_t ldD, int n_c, const void* Q, int64_t ldQ, bool q_cplx, int D, int S, int kb_lo,
               int kb_hi, void* Hq, int ldS, hipStream_t st) {
    if (kb_hi <= kb_lo) return;
    dim3 grid(kb_hi - kb_lo, 2, (unsigned)ceil_div(S, 256));
    if (q_cplx) hq_kernel<cplx><<<grid, 256, 0, st>>>((const cplx*)Hc, ldD, n_c, (const cplx*)Q, ldQ, D, S, kb_lo, (cplx*)Hq, ldS);
    else hq_kernel<double><<<grid, 256, 0, st>>>((const cplx*)Hc, ldD, n_c, (const double*)Q, ldQ, D, S, kb_lo, (cplx*)Hq, ldS);
    KERNEL_CHECK();
}

void launch_ypinv(const void* Q, int64_t ldQ, bool q_cplx, const void* Zb, int ldS, int D, int S, int C, void* Ypinv,
                  int64_t ldD, hipStream_t st) {
    const unsigned grid = (unsigned)ceil_div(D, 256);
    if (q_cplx) ypinv_kernel<cplx><<<grid, 256, 0, st>>>((const cplx*)Q, ldQ, (const cplx*)Zb, ldS, D, S, C, (cplx*)Ypinv, ldD);
    else ypinv_kernel<double><<<grid, 256, 0, st>>>((const double*)Q, ldQ, (const cplx*)Zb, ldS, D, S, C, (double*)Ypinv, ldD);
    KERNEL_CHECK();
}

void launch_ls_apply(const void* Hc, int64_t ldH, int n_c, const void* Zf, bool z_cplx, int64_t ldD, int D, int C, int P,
                     int kb_lo, int kb_hi, void* W, hipStream_t st) {
    if (kb_hi <= kb_lo) return;
    if (kb_lo != 0) throw Error(2, "ls_apply: bins must start at 0");
    dim3 grid(kb_hi, 2);
    if (z_cplx) ls_apply_kernel<cplx><<<grid, 256, 0, st>>>((const cplx*)Hc, ldH, n_c, (const cplx*)Zf, ldD, D, C, P, (cplx*)W);
    else ls_apply_kernel<double><<<grid, 256, 0, st>>>((const cplx*)Hc, ldH, n_c, (const double*)Zf, ldD, D, C, P, (cplx*)W);
    KERNEL_CHECK();
}

void launch_ls_filters(const double* hL, const double* hR, int64_t L, int D, const void* Yp, bool cplx_basis, int64_t ldD,
                       int C, void* wL, void* wR, hipStream_t st) {
    dim3 grid(C, 2, (unsigned)ceil_div(L, 256));
    if (cplx_basis) ls_filters_kernel<cplx><<<grid, 256, 0, st>>>(hL, hR, L, D, (const cplx*)Yp, ldD, (cplx*)wL, (cplx*)wR);
    else ls_filters_kernel<double><<<grid, 256, 0, st>>>(hL, hR, L, D, (const double*)Yp, ldD, (double*)wL, (double*)wR);
    KERNEL_CHECK();
}

void launch_conj_copy(const void* in, void* out, int64_t n, bool is_cplx, hipStream_t st) {
    if (is_cplx) conj_copy_kernel<cplx><<<1024, 256, 0, st>>>((const cplx*)in, (cplx*)out, n);
    else conj_copy_kernel<double><<<1024, 256, 0, st>>>((const double*)in, (double*)out, n);
    KERNEL_CHECK();
}

void launch_widen(const void* in, int64_t ldi, bool in_cplx, void* out, int64_t ldo, int rows, int cols, bool transpose,
                  bool upper_only, hipStream_t st) {
    const unsigned grid = (unsigned)std::min<int64_t>(1024, ceil_div((int64_t)rows * cols, 256));
    if (in_cplx) widen_kernel<cplx><<<grid, 256, 0, st>>>((const cplx*)in, ldi, (cplx*)out, ldo, rows, cols, transpose, upper_only);
    else widen_kernel<double><<<grid, 256, 0, st>>>((const double*)in, ldi, (cplx*)out, ldo, rows, cols, transpose, upper_only);
    KERNEL_CHECK();
}

}  // namespace emagls
