// The sequential magnitude-least-squares phase sweep in its launch-per-bin forms and the small dense products
// around it.  (The array designs sweep with the persistent kernel of sweep_persist.hip; the kernels here serve
// MagLS / FromAtf, shapes the persistent kernel does not cover, and EMAGLS_SWEEP_PERSIST=0.)
//
// Reference (lib/getEMagLsFilters.m:95-103):
//     phi = angle(W(k-1,:) * pwGrid);   W(k,:) = (abs(H(k,:)) .* exp(1i*phi)) * Y_reg_inv;
// exp(1i*angle(p)) is evaluated as p/|p| (1 when p == 0, as angle(0) = 0); Nyquist takes real(t).
// Each workgroup owns a slab of directions; its partial W(k,:) goes to a [pair][nWG] buffer that the
// next launch sums first (the launch boundary is the grid-wide barrier and the release/acquire).
#include "kernels.hpp"

namespace emagls {

constexpr int SW_NT = 512;  // threads per workgroup (8 waves)


__device__ __forceinline__ cplx unit_phase_times(double h, cplx p, bool nyquist) {
    // |H| exp(1i*angle(p)) = |H| p/|p|;  angle(0) = 0 -> 1.  p is O(1e-3..1e3): |p|^2 cannot over/underflow
    const double a2 = norm2(p);
    cplx t = mk(h, 0.0);
    if (a2 > 0.0) {
        const double ia = h * fast_rsqrt(a2);
        t = mk(p.x * ia, p.y * ia);
    }
    if (nyquist) t.y = 0.0;
    return t;
}

// Partial sums live in Wpart[parity][pair][nWG] (pair = e*C + c): every (ear, channel) pair's nWG values are
// contiguous, so one wave sums a pair with a few coalesced loads per lane and one wave reduction.
// Sums the previous launch's partials into Wp[pair] (LDS); workgroup 0 publishes W(kb-1,:).
__device__ __forceinline__ void gather_prev(cplx* Wp, const cplx* Wpart_prev, cplx* W, int nWG, int C, int P, int kb,
                                            bool first) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nwaves = blockDim.x >> 6;
    for (int pair = wave; pair < 2 * C; pair += nwaves) {
        const int e = pair / C, c = pair % C;
        cplx acc = mk(0, 0);
        if (first) {
            if (lane == 0) acc = W[((int64_t)e * P + (kb - 1)) * C + c];
        } else {
            const cplx* src = Wpart_prev + (int64_t)pair * nWG;
            cplx v0 = mk(0, 0), v1 = mk(0, 0), v2 = mk(0, 0), v3 = mk(0, 0);
            if (lane < nWG) v0 = src[lane];
            if (lane + 64 < nWG) v1 = src[lane + 64];
            if (lane + 128 < nWG) v2 = src[lane + 128];
            if (lane + 192 < nWG) v3 = src[lane + 192];
            acc = (v0 + v1) + (v2 + v3);
            for (int w = lane + 256; w < nWG; w += 64) acc += src[w];
        }
        acc = group_sum<64>(acc);
        if (lane == 0) {
            Wp[pair] = acc;
            if (blockIdx.x == 0 && !first) W[((int64_t)e * P + (kb - 1)) * C + c] = acc;
        }
    }
}

constexpr int SW_CMAX = 32;

// ---------------------------------------------------------------------------------------------
// dense sweep: pwGrid_k given directly as X[kb][c][d] and Y_reg_inv_k as Zd[kb][c][d]
// (strides may be 0: plain MagLS uses the fixed Y_conj / Y_pinv, lib/getMagLsFilters.m:64-72)
// ---------------------------------------------------------------------------------------------

constexpr int DS_NT = 256;
constexpr int DS_DPW = 64;

template <typename TX>
__device__ __forceinline__ void sweep_dense_body(const DenseSweepArgs& a, int kb, cplx* Wp, cplx (*ts)[DS_DPW], cplx* stage) {
    const int tid = threadIdx.x;
    const int C = a.C, nWG = a.nWG;
    const bool nyq = (kb == a.P - 1);
    const bool first = (kb == a.kfirst);
    const cplx* Wprev = a.Wpart + (int64_t)((kb - 1) & 1) * nWG * 2 * C;
    cplx* Wout = a.Wpart + (int64_t)(kb & 1) * nWG * 2 * C;
    const TX* X = reinterpret_cast<const TX*>(a.X) + (int64_t)kb * a.x_stride;
    const TX* Zd = reinterpret_cast<const TX*>(a.Zd) + (int64_t)kb * a.z_stride;
    // grids of more than 64 x 4096 / (2 C) directions: a workgroup walks `spw` slabs of 64 directions, so that the partial sums
    // the next launch stages stay within 16 per thread (dense_sweep_nwg)
    const int nslab = (a.D + DS_DPW - 1) / DS_DPW, spw = (nslab + nWG - 1) / nWG;
    int64_t d0 = (int64_t)blockIdx.x * spw * DS_DPW;
    const int64_t na = a.P - a.kabs0;
    // ---- 0. every load of the launch goes out up front; the previous launch's partial sums first
    //         (they gate the chain and memory returns in order), then this slab's operands
    constexpr int NGV = 16;  // staged partials per thread: 2C*nWG <= 64*64
    const int npart = 2 * C * nWG;
    cplx gv[NGV];
#pragma unroll
    for (int i = 0; i < NGV; ++i) {
        const int f = tid + DS_NT * i;
        gv[i] = (!first && f < npart) ? Wprev[f] : mk(0, 0);
    }
    const int e_ = tid / DS_DPW, dd_ = tid % DS_DPW;       // phase-1 role: (ear, direction)
    const int pair2 = tid >> 2, part2 = tid & 3;             // phase-2 role: (ear, channel) x 4 lanes
    constexpr int NZ = DS_DPW / 4;
    TX xr[SW_CMAX];
    TX zr[NZ];
    double habs;
    bool p1;
    auto load_slab = [&]() __attribute__((always_inline)) {
        const int64_t d_ = d0 + dd_;
        p1 = tid < 2 * DS_DPW && d_ < a.D;
#pragma unroll
        for (int c = 0; c < SW_CMAX; ++c) xr[c] = (p1 && c < C) ? X[(int64_t)c * a.ldD + d_] : zero_of<TX>();
        habs = p1 ? a.Habs[((int64_t)e_ * na + (kb - a.kabs0)) * a.ldH + d_] : 0.0;
#pragma unroll
        for (int j = 0; j < NZ; ++j) {
            const int64_t d = d0 + part2 + 4 * j;
            zr[j] = (pair2 < 2 * C && d < a.D) ? Zd[(int64_t)(pair2 % C) * a.ldD + d] : zero_of<TX>();
        }
    };
    load_slab();
    // ---- 1. W(k-1) = sum of the staged partials
#pragma unroll
    for (int i = 0; i < NGV; ++i) {
        const int f = tid + DS_NT * i;
        if (f < npart) stage[f + f / nWG] = gv[i];
    }
    __syncthreads();
    for (int pair = tid >> 2; pair < 2 * C; pair += DS_NT >> 2) {
        const int part = tid & 3;
        const int e = pair / C, c = pair % C;
        cplx acc = mk(0, 0);
        if (first) {
            if (part == 0) acc = a.W[((int64_t)e * a.P + (kb - 1)) * C + c];
        } else {
            const cplx* row = stage + (size_t)pair * (nWG + 1);
            cplx a0 = mk(0, 0), a1 = mk(0, 0);
            for (int w = part; w < nWG; w += 8) {
                a0 += row[w];
                if (w + 4 < nWG) a1 += row[w + 4];
            }
            acc = a0 + a1;
        }
        acc = group_sum<4>(acc);
        if (part == 0) {
            Wp[pair] = acc;
            if (blockIdx.x == 0 && !first) a.W[((int64_t)e * a.P + (kb - 1)) * C + c] = acc;
        }
    }
    __syncthreads();
    cplx acc3 = mk(0, 0);
    for (int sl = 0; sl < spw; ++sl) {
        if (sl > 0) {
            __syncthreads();   // (the previous slab's t has been consumed)
            d0 += DS_DPW;
            load_slab();
        }
        // ---- 2. p = W(k-1) pwGrid ;  t = |H| p/|p|
        if (tid < 2 * DS_DPW) {
            cplx t = mk(0, 0);
            if (p1) {
                cplx pa = mk(0, 0), pb = mk(0, 0);
#pragma unroll
                for (int c = 0; c < SW_CMAX; c += 2) {
                    if (c < C) cfma(pa, Wp[e_ * C + c], xr[c]);
                    if (c + 1 < C) cfma(pb, Wp[e_ * C + c + 1], xr[c + 1]);
                }
                t = unit_phase_times(habs, pa + pb, nyq);
            }
            ts[e_][dd_] = t;
        }
        __syncthreads();
        // ---- 3. partial W(k,:)[e][c] = sum_{d in slab} t[e][d] Y_reg_inv[d][c] ; 4 lanes per (e,c)
        if (pair2 < 2 * C) {
            const int e = pair2 / C;
#pragma unroll
            for (int j = 0; j < NZ; ++j) cfma(acc3, ts[e][part2 + 4 * j], zr[j]);
        }
    }
    if (pair2 < 2 * C) {
        const int e = pair2 / C, c = pair2 % C;
        acc3 = group_sum<4>(acc3);
        if (part2 == 0) Wout[((int64_t)e * C + c) * nWG + blockIdx.x] = acc3;
    }
}

template <typename TX>
__global__ void __launch_bounds__(DS_NT) sweep_dense_kernel(DenseSweepArgs a, int kb) {
    __shared__ __attribute__((aligned(16))) cplx Wp[64];
    __shared__ __attribute__((aligned(16))) cplx ts[2][DS_DPW];
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    sweep_dense_body<TX>(a, kb, Wp, ts, reinterpret_cast<cplx*>(dyn));
}

// ---------------------------------------------------------------------------------------------
// Halved sweep: one direction-space operand per bin.  With Y_reg_inv_k = conj(G_k) conj(M_k):
//   launch kb:  W(kb-1,:) = (sum of the previous launch's partials) conj(M_{kb-1})        [C x C, every workgroup]
//               p = W(kb-1,:) G_kb^T ; t = |H| p/|p| ; partial v = t conj(G_kb) over this workgroup's slab
// The G slab is fetched once into LDS and used for both products, so a workgroup reads 800 B x nWG of
// partials + 25.6 KB of G + 10 KB of M instead of two 25.6 KB slabs.  Ill-conditioned bins (cond_ok == 0, rare:
// tiny arrays) take their accurate Y_reg_inv slab from memory instead and skip the M product.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void sweep_half_body(const HalfSweepArgs& a, int kb, char* dyn) {
    __shared__ __attribute__((aligned(16))) cplx Wp[64];
    __shared__ __attribute__((aligned(16))) cplx vt[64];
    __shared__ __attribute__((aligned(16))) cplx ts[2][DS_DPW];
    const int tid = threadIdx.x;
    const int C = a.C, nWG = a.nWG;
    cplx* xs = reinterpret_cast<cplx*>(dyn);                  // [C][DS_DPW+1]   G slab
    cplx* ms = xs + (size_t)C * (DS_DPW + 1);                 // [C][C]          M of the previous bin
    cplx* stage = ms + (size_t)C * C;                         // [2C][nWG+1]     partial sums of the previous launch
    const bool nyq = (kb == a.P - 1);
    const bool first = (kb == a.kfirst);
    const cplx* Wprev = a.Wpart + (int64_t)((kb - 1) & 1) * nWG * 2 * C;
    cplx* Wout = a.Wpart + (int64_t)(kb & 1) * nWG * 2 * C;
    const cplx* X = a.G + (int64_t)kb * a.g_stride;
    const int nslab = (a.D + DS_DPW - 1) / DS_DPW, spw = (nslab + nWG - 1) / nWG;   // (slabs per workgroup: sweep_dense_body)
    int64_t d0 = (int64_t)blockIdx.x * spw * DS_DPW;
    const int64_t na = a.P - a.kabs0;
    // ---- 0. all loads up front; the previous launch's partial sums first (they gate the chain)
    constexpr int NGV = 16;
    const int npart = 2 * C * nWG;
    cplx gv[NGV];
#pragma unroll
    for (int i = 0; i < NGV; ++i) {
        const int f = tid + DS_NT * i;
        gv[i] = (!first && f < npart) ? Wprev[f] : mk(0, 0);
    }
    constexpr int NXV = (SW_CMAX * DS_DPW) / DS_NT;  // 8 slab elements per thread
    cplx xv[NXV];
    const int e_ = tid / DS_DPW, dd_ = tid % DS_DPW;
    double habs;
    bool p1;
    auto load_slab = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NXV; ++i) {
            const int f = tid + DS_NT * i, c = f / DS_DPW, dd = f % DS_DPW;
            xv[i] = (c < C && d0 + dd < a.D) ? X[(int64_t)c * a.ldD + d0 + dd] : mk(0, 0);
        }
        const int64_t d_ = d0 + dd_;
        p1 = tid < 2 * DS_DPW && d_ < a.D;
        habs = p1 ? a.Habs[((int64_t)e_ * na + (kb - a.kabs0)) * a.ldH + d_] : 0.0;
    };
    auto stage_slab = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NXV; ++i) {
            const int f = tid + DS_NT * i, c = f / DS_DPW, dd = f % DS_DPW;
            if (c < C) xs[(size_t)c * (DS_DPW + 1) + dd] = xv[i];
        }
    };
    load_slab();
    constexpr int NMV = (SW_CMAX * SW_CMAX) / DS_NT;  // 4 elements of M per thread
    const cplx* M = a.Mw + (int64_t)(kb - 1) * C * C;
    cplx mv[NMV];
#pragma unroll
    for (int i = 0; i < NMV; ++i) {
        const int f = tid + DS_NT * i;
        mv[i] = (!first && f < C * C) ? M[f] : mk(0, 0);
    }
    const bool prev_ok = first ? true : (a.cond_ok[kb - 1] != 0.0);
    const bool cur_ok = a.cond_ok[kb] != 0.0;
    // ---- 1. stage everything in LDS
#pragma unroll
    for (int i = 0; i < NGV; ++i) {
        const int f = tid + DS_NT * i;
        if (f < npart) stage[f + f / nWG] = gv[i];
    }
    stage_slab();
#pragma unroll
    for (int i = 0; i < NMV; ++i) {
        const int f = tid + DS_NT * i;
        if (f < C * C) ms[f] = mv[i];
    }
    __syncthreads();
    // ---- 2. v_total[pair] = sum over workgroups
    for (int pair = tid >> 2; pair < 2 * C; pair += DS_NT >> 2) {
        const int part = tid & 3;
        cplx acc = mk(0, 0);
        if (first) {
            if (part == 0) acc = a.W[((int64_t)(pair / C) * a.P + (kb - 1)) * C + pair % C];
        } else {
            const cplx* row = stage + (size_t)pair * (nWG + 1);
            cplx a0 = mk(0, 0), a1 = mk(0, 0);
            for (int w = part; w < nWG; w += 8) {
                a0 += row[w];
                if (w + 4 < nWG) a1 += row[w + 4];
            }
            acc = a0 + a1;
        }
        acc = group_sum<4>(acc);
        if (part == 0) vt[pair] = acc;
    }
    __syncthreads();
    // ---- 3. W(kb-1,:) = v_total conj(M_{kb-1})   (identity for the first swept bin and for ill-conditioned bins)
    for (int pair = tid >> 2; pair < 2 * C; pair += DS_NT >> 2) {
        const int part = tid & 3;
        const int e = pair / C, c = pair % C;
        cplx acc = mk(0, 0);
        if (first || !prev_ok) {
            if (part == 0) acc = vt[pair];
        } else {
            for (int cc = part; cc < C; cc += 4) cfma(acc, vt[e * C + cc], conj(ms[cc * C + c]));
        }
        acc = group_sum<4>(acc);
        if (part == 0) {
            Wp[pair] = acc;
            if (blockIdx.x == 0 && !first) a.W[((int64_t)e * a.P + (kb - 1)) * C + c] = acc;
        }
    }
    __syncthreads();
    cplx a0 = mk(0, 0), a1 = mk(0, 0);
    for (int sl = 0; sl < spw; ++sl) {
        if (sl > 0) {
            __syncthreads();   // (the previous slab and its t have been consumed)
            d0 += DS_DPW;
            load_slab();
            stage_slab();
            __syncthreads();
        }
        // ---- 4. p = W(kb-1,:) pwGrid ;  t = |H| p/|p|
        if (tid < 2 * DS_DPW) {
            cplx t = mk(0, 0);
            if (p1) {
                cplx pa = mk(0, 0), pb = mk(0, 0);
                int c = 0;
                for (; c + 1 < C; c += 2) {
                    cfma(pa, Wp[e_ * C + c], xs[(size_t)c * (DS_DPW + 1) + dd_]);
                    cfma(pb, Wp[e_ * C + c + 1], xs[(size_t)(c + 1) * (DS_DPW + 1) + dd_]);
                }
                if (c < C) cfma(pa, Wp[e_ * C + c], xs[(size_t)c * (DS_DPW + 1) + dd_]);
                t = unit_phase_times(habs, pa + pb, nyq);
            }
            ts[e_][dd_] = t;
        }
        __syncthreads();
        // ---- 5. partial of this slab: v = t conj(G)  (or t Y_reg_inv for an ill-conditioned bin); 4 lanes per (e,c)
        const int pair = tid >> 2, part = tid & 3;
        if (pair < 2 * C) {
            const int e = pair / C, c = pair % C;
            if (cur_ok) {
                const cplx* xrow = xs + (size_t)c * (DS_DPW + 1);
#pragma unroll
                for (int j = 0; j < DS_DPW / 4; j += 2) {
                    cfma(a0, ts[e][part + 4 * j], conj(xrow[part + 4 * j]));
                    cfma(a1, ts[e][part + 4 * (j + 1)], conj(xrow[part + 4 * (j + 1)]));
                }
            } else {
                const cplx* Y = a.Yri + (int64_t)kb * a.g_stride + (int64_t)c * a.ldD;
                for (int dd = part; dd < DS_DPW; dd += 4)
                    if (d0 + dd < a.D) cfma(a0, ts[e][dd], Y[d0 + dd]);
            }
        }
    }
    {
        const int pair = tid >> 2, part = tid & 3;
        if (pair < 2 * C) {
            const int e = pair / C, c = pair % C;
            cplx acc = group_sum<4>(a0 + a1);
            if (part == 0) Wout[((int64_t)e * C + c) * nWG + blockIdx.x] = acc;
        }
    }
}

__global__ void __launch_bounds__(DS_NT) sweep_half_kernel(HalfSweepMulti m, int kb) {
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    sweep_half_body(m.a[blockIdx.y], kb, dyn);
}

// after the last swept bin: W(P-1,:) = (sum of partials) conj(M_{P-1})
__global__ void __launch_bounds__(256) sweep_half_finalize_kernel(HalfSweepMulti m, int kb_last) {
    const HalfSweepArgs& a = m.a[blockIdx.x];
    __shared__ cplx vt[64];
    const int C = a.C, nWG = a.nWG;
    const cplx* Wprev = a.Wpart + (int64_t)(kb_last & 1) * nWG * 2 * C;
    const int tid = threadIdx.x;
    for (int pair = tid >> 2; pair < 2 * C; pair += 64) {
        const int part = tid & 3;
        cplx acc = mk(0, 0);
        for (int w = part; w < nWG; w += 4) acc += Wprev[(int64_t)pair * nWG + w];
        acc = group_sum<4>(acc);
        if (part == 0) vt[pair] = acc;
    }
    __syncthreads();
    const bool ok = a.cond_ok[kb_last] != 0.0;
    const cplx* M = a.Mw + (int64_t)kb_last * C * C;
    for (int pair = tid; pair < 2 * C; pair += 256) {
        const int e = pair / C, c = pair % C;
        cplx acc = mk(0, 0);
        if (ok) { for (int cc = 0; cc < C; ++cc) cfma(acc, vt[e * C + cc], conj(M[cc * C + c])); }
        else acc = vt[pair];
        a.W[((int64_t)e * a.P + kb_last) * C + c] = acc;
    }
}

void launch_sweep_half(const HalfSweepMulti& m, int kb, hipStream_t st) {
    const HalfSweepArgs& a = m.a[0];
    if (2 * a.C * a.nWG > 16 * DS_NT || a.C > SW_CMAX) throw Error(2, "halved sweep: shape not supported");
    const size_t dyn = sizeof(cplx) * ((size_t)a.C * (DS_DPW + 1) + (size_t)a.C * a.C + (size_t)2 * a.C * (a.nWG + 1));
    sweep_half_kernel<<<dim3(a.nWG, m.n), DS_NT, dyn, st>>>(m, kb);
    KERNEL_CHECK();
}
void launch_sweep_half_finalize(const HalfSweepMulti& m, int kb_last, hipStream_t st) {
    sweep_half_finalize_kernel<<<m.n, 256, 0, st>>>(m, kb_last);
    KERNEL_CHECK();
}

// after the last swept bin: W(P-1,:) = sum of partials
__global__ void __launch_bounds__(SW_NT) sweep_finalize_kernel(const cplx* __restrict__ Wpart, cplx* __restrict__ W,
                                                               int nWG, int C, int P, int kb_last) {
    __shared__ cplx Wp[64];
    const cplx* Wprev = Wpart + (int64_t)(kb_last & 1) * nWG * 2 * C;
    gather_prev(Wp, Wprev, W, nWG, C, P, kb_last + 1, false);
}

// Least-squares rows without Q:  conj(H conj(Yc)) as a tiled product  P[r][s] = sum_d H[r][d] conj(Yc[d][s])
// (r = (ear, bin) rows of Hc, 2 n_c of them).  Workgroup tile 96 rows x 32 columns x one K slice of the directions,
// thread tile 3 x 4: the operands of a 32-direction chunk sit in LDS and every LDS read feeds 3-4 complex FMAs.
// The K slices are summed in a fixed order by hy_reduce_kernel (bitwise reproducible), which also conjugates.
constexpr int HY_RT = 96, HY_CT = 32, HY_DC = 32, HY_KS = 8;

template <typename TY>
__global__ void __launch_bounds__(256) hy_partial_kernel(const cplx* __restrict__ Hc, int64_t ldD, int nrows, const TY* __restrict__ Yc,
                                                         int64_t ldY, int D, int S, cplx* __restrict__ Pw, int ldS, size_t bstride) {
    Hc = boff(Hc, bstride); Yc = boff(Yc, bstride); Pw = boff(Pw, bstride);
    __shared__ __attribute__((aligned(16))) cplx hs[HY_RT][HY_DC + 1];
    __shared__ __attribute__((aligned(16))) TY ys[HY_DC][HY_CT + 1];   // real basis: real tile, 2 FMAs per product
    const int tid = threadIdx.x, cg = tid & 7, rg = tid >> 3;
    const int s0 = blockIdx.x * HY_CT;
    const int rt = blockIdx.y / HY_KS, ks = blockIdx.y % HY_KS;
    const int r0 = rt * HY_RT;
    const int dper = ((D + HY_KS - 1) / HY_KS + HY_DC - 1) / HY_DC * HY_DC;
    const int dbeg = ks * dper, dend = min(D, dbeg + dper);
    cplx acc[3][4];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mk(0, 0);
    for (int d0 = dbeg; d0 < dend; d0 += HY_DC) {
#pragma unroll
        for (int i = 0; i < (HY_RT * HY_DC) / 256; ++i) {
            const int idx = tid + 256 * i, r = idx / HY_DC, dd = idx % HY_DC;
            hs[r][dd] = (r0 + r < nrows && d0 + dd < dend) ? Hc[(int64_t)(r0 + r) * ldD + d0 + dd] : mk(0, 0);
        }
#pragma unroll
        for (int i = 0; i < (HY_DC * HY_CT) / 256; ++i) {
            const int idx = tid + 256 * i, dd = idx / HY_CT, c = idx % HY_CT;
            ys[dd][c] = (d0 + dd < dend && s0 + c < S) ? Yc[(int64_t)(d0 + dd) * ldY + s0 + c] : zero_of<TY>();
        }
        __syncthreads();
#pragma unroll 4
        for (int dd = 0; dd < HY_DC; ++dd) {
            TY y[4];
            cplx h[3];
#pragma unroll
            for (int j = 0; j < 4; ++j) y[j] = conj(ys[dd][4 * cg + j]);
#pragma unroll
            for (int i = 0; i < 3; ++i) h[i] = hs[rg + 32 * i][dd];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) cfma(acc[i][j], h[i], y[j]);
        }
        __syncthreads();
    }
    cplx* P = Pw + (int64_t)ks * nrows * ldS;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int r = r0 + rg + 32 * i;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int sc = s0 + 4 * cg + j;
            if (r < nrows && sc < S) P[(int64_t)r * ldS + sc] = acc[i][j];
        }
    }
}
__global__ void __launch_bounds__(256) hy_reduce_kernel(const cplx* __restrict__ Pw, int nrows, int S, int ldS, cplx* __restrict__ out,
                                                        size_t bstride) {
    Pw = boff(Pw, bstride); out = boff(out, bstride);
    const int r = blockIdx.y;
    const int sc = blockIdx.x * 256 + threadIdx.x;
    if (sc >= S) return;
    cplx acc = mk(0, 0);
#pragma unroll
    for (int ks = 0; ks < HY_KS; ++ks) acc += Pw[((int64_t)ks * nrows + r) * ldS + sc];
    out[(int64_t)r * ldS + sc] = conj(acc);
}
// out[r][s] = conj( sum_d Hc[r][d] conj(Yc[d][s]) ), r < nrows;  Pw: workspace [HY_KS][nrows][ldS]
void launch_hy_conj(const void* Hc, int64_t ldD, int nrows, const void* Yc, int64_t ldY, bool y_cplx, int D, int S, void* Pw, void* out,
                    int ldS, hipStream_t st) {
    if (nrows <= 0) return;
    const dim3 grid((unsigned)ceil_div(S, HY_CT), (unsigned)(ceil_div(nrows, HY_RT) * HY_KS));
    if (y_cplx)
        hy_partial_kernel<cplx><<<bgrid(grid), 256, 0, st>>>((const cplx*)Hc, ldD, nrows, (const cplx*)Yc, ldY, D, S, (cplx*)Pw, ldS,
                                                             batch_ctx().stride);
    else
        hy_partial_kernel<double><<<bgrid(grid), 256, 0, st>>>((const cplx*)Hc, ldD, nrows, (const double*)Yc, ldY, D, S, (cplx*)Pw, ldS,
                                                               batch_ctx().stride);
    KERNEL_CHECK();
    hy_reduce_kernel<<<bgrid(dim3((unsigned)ceil_div(S, 256), nrows)), 256, 0, st>>>((const cplx*)Pw, nrows, S, ldS, (cplx*)out,
                                                                                   batch_ctx().stride);
    KERNEL_CHECK();
}
size_t hy_workspace_elems(int nrows, int ldS) { return (size_t)HY_KS * nrows * ldS; }

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
    const int part = tid & 7;  // 32 channels x 8 lanes per pass
    for (int pair = tid >> 3; pair < C; pair += 32) {
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
__global__ void conj_copy_kernel(const T* __restrict__ in, T* __restrict__ out, int64_t n, size_t bstride) {
    in = boff(in, bstride); out = boff(out, bstride);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = conj(in[i]);
}
template <typename T>
__global__ void widen_kernel(const T* __restrict__ in, int64_t ldi, cplx* __restrict__ out, int64_t ldo, int rows, int cols,
                             int transpose, int upper_only, size_t bstride) {
    in = boff(in, bstride); out = boff(out, bstride);
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
void launch_sweep_dense(const DenseSweepArgs& a, int kb, bool x_cplx, hipStream_t st) {
    if (2 * a.C * a.nWG > 16 * DS_NT) throw Error(2, "dense sweep: too many workgroup partials");
    const size_t dyn = sizeof(cplx) * (size_t)2 * a.C * (a.nWG + 1);
    if (x_cplx) sweep_dense_kernel<cplx><<<a.nWG, DS_NT, dyn, st>>>(a, kb);
    else sweep_dense_kernel<double><<<a.nWG, DS_NT, dyn, st>>>(a, kb);
    KERNEL_CHECK();
}
// workgroups of the launch-per-bin sweeps: one per slab of 64 directions while the next launch can stage all their partial sums
// (2 C nWG <= 16 per thread), several slabs per workgroup on larger grids
int dense_sweep_nwg(int D, int C) {
    const int nslab = (D + DS_DPW - 1) / DS_DPW, cap = std::max(1, 16 * DS_NT / (2 * std::max(C, 1)));
    if (C > SW_CMAX) return nslab;   // (sweep_wide_kernel: one slab per workgroup, the partial sums are read in a loop)
    const int spw = (nslab + cap - 1) / cap;
    return (nslab + spw - 1) / spw;
}

void launch_sweep_finalize(const void* Wpart, void* W, int nWG, int C, int P, int kb_last, hipStream_t st) {
    sweep_finalize_kernel<<<1, SW_NT, 0, st>>>((const cplx*)Wpart, (cplx*)W, nWG, C, P, kb_last);
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
    if (is_cplx) conj_copy_kernel<cplx><<<bgrid(1024), 256, 0, st>>>((const cplx*)in, (cplx*)out, n, batch_ctx().stride);
    else conj_copy_kernel<double><<<bgrid(1024), 256, 0, st>>>((const double*)in, (double*)out, n, batch_ctx().stride);
    KERNEL_CHECK();
}

void launch_widen(const void* in, int64_t ldi, bool in_cplx, void* out, int64_t ldo, int rows, int cols, bool transpose,
                  bool upper_only, hipStream_t st) {
    const unsigned grid = (unsigned)std::min<int64_t>(1024, ceil_div((int64_t)rows * cols, 256));
    if (in_cplx) widen_kernel<cplx><<<bgrid(grid), 256, 0, st>>>((const cplx*)in, ldi, (cplx*)out, ldo, rows, cols, transpose, upper_only, batch_ctx().stride);
    else widen_kernel<double><<<bgrid(grid), 256, 0, st>>>((const double*)in, ldi, (cplx*)out, ldo, rows, cols, transpose, upper_only, batch_ctx().stride);
    KERNEL_CHECK();
}

}  // namespace emagls
