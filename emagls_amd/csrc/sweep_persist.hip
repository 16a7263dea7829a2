// Persistent form of the halved MagLS phase sweep: ONE launch walks all swept bins of up to 8 designs.
//
// Reference recurrence (lib/getEMagLsFilters.m:95-103): W(k,:) depends on W(k-1,:), so the bins are a chain of
// P - k_cut dependent steps.  The launch-per-bin kernels (sweep.hip) pay a kernel boundary per step: cold L2s
// and >= 4.5 us.  Here the workgroups of one design stay resident and exchange their partial sums through
// memory inside the launch:
//
//   * a design's workgroups are the blocks b with b % 8 == design (the dispatcher is observed to place block b
//     on XCD b % 8, so a design's exchange stays inside one XCD's L2; this is a speed assumption only);
//   * the partial W(k,:) of a workgroup (2C complex numbers) is published as 8-byte {payload32, tag32} granules,
//     each written by ONE relaxed agent-scope store (sc1, write-through) and read by relaxed agent-scope loads
//     (sc1, bypass L1): a granule is valid as soon as its tag equals the bin number, so no fence, flag or
//     barrier is needed and the protocol does not depend on where the workgroups run;
//   * two granule slots (bin parity) suffice: a workgroup can only publish bin k+2 after it has read every
//     other workgroup's bin k+1, which those publish only after they finished reading bin k;
//   * the operands of the next bin (G slab, M, |H|) do not depend on the chain; they are fetched into registers
//     right after a bin's exchange completed and are consumed one bin later;
//   * a workgroup that waits longer than the spin limit sets a sticky abort flag and everybody leaves:
//     a missing peer (not co-resident, killed) produces an error return, never a hang.
#include "kernels.hpp"

namespace emagls {

namespace {

constexpr int PS_NT = 256;
constexpr int PS_CMAX = 32;
constexpr unsigned PS_SPIN_LIMIT = 1u << 21;

__device__ __forceinline__ cplx unit_phase(double h, cplx p, bool nyquist) {
    const double a2 = norm2(p);
    cplx t = mk(h, 0.0);
    if (a2 > 0.0) {
        const double ia = h * fast_rsqrt(a2);
        t = mk(p.x * ia, p.y * ia);
    }
    if (nyquist) t.y = 0.0;
    return t;
}

__device__ __forceinline__ void ll_store(unsigned long long* dst, double v, unsigned tag) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    const unsigned long long hi = (unsigned long long)tag << 32;
    __hip_atomic_store(dst, hi | (bits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(dst + 1, hi | (bits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// DPW: directions per workgroup; NV: granule pairs (doubles) gathered per thread and pass
template <int DPW, int NV>
__global__ void __launch_bounds__(PS_NT) sweep_persist_kernel(HalfSweepMulti m, int nWG) {
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    __shared__ __attribute__((aligned(16))) cplx Wp[64];
    __shared__ __attribute__((aligned(16))) cplx vt[64];
    __shared__ __attribute__((aligned(16))) cplx ts[2][DPW];
    __shared__ int s_abort;
    const int design = blockIdx.x & 7, member = blockIdx.x >> 3;
    if (design >= m.n || member >= nWG) return;
    const HalfSweepArgs& a = m.a[design];
    const int tid = threadIdx.x;
    const int C = a.C;
    cplx* xs = reinterpret_cast<cplx*>(dyn);           // [C][DPW+1]     G slab of the current bin
    cplx* ms = xs + (size_t)C * (DPW + 1);             // [C][C]         M of the previous bin
    cplx* stage = ms + (size_t)C * C;                  // [2C][nWG+1]    partial sums of the previous bin
    double* stage_d = reinterpret_cast<double*>(stage);
    const int64_t d0 = (int64_t)member * DPW;
    const int64_t na = a.P - a.kabs0;
    const int ndbl = nWG * 4 * C;                      // doubles in one granule slot
    unsigned long long* ll = a.ll;
    if (tid == 0) s_abort = 0;

    constexpr int NXV = (PS_CMAX * DPW) / PS_NT;
    constexpr int NMV = (PS_CMAX * PS_CMAX) / PS_NT;
    cplx xv[NXV], mv[NMV];
    double habs = 0.0;
    const int e_ = tid / DPW, dd_ = tid % DPW;
    const int64_t d_ = d0 + dd_;
    const bool p1 = tid < 2 * DPW && d_ < a.D;
    // operands of bin kb: G_kb slab, M_{kb-1}, |H_kb|
    auto fetch = [&](int kb) {
        const bool have_g = kb < a.P;
        const cplx* X = a.G + (int64_t)kb * a.g_stride;
#pragma unroll
        for (int i = 0; i < NXV; ++i) {
            const int f = tid + PS_NT * i, c = f / DPW, dd = f % DPW;
            xv[i] = (have_g && c < C && d0 + dd < a.D) ? X[(int64_t)c * a.ldD + d0 + dd] : mk(0, 0);
        }
        const cplx* M = a.Mw + (int64_t)(kb - 1) * C * C;
#pragma unroll
        for (int i = 0; i < NMV; ++i) {
            const int f = tid + PS_NT * i;
            mv[i] = (kb > a.kfirst && f < C * C) ? M[f] : mk(0, 0);
        }
        habs = (have_g && p1) ? a.Habs[((int64_t)e_ * na + (kb - a.kabs0)) * a.ldH + d_] : 0.0;
    };
    fetch(a.kfirst);
    __syncthreads();

    for (int kb = a.kfirst; kb <= a.P; ++kb) {
        const bool first = (kb == a.kfirst);
        const bool last = (kb == a.P);  // only W(P-1,:) is left to form
        const bool nyq = (kb == a.P - 1);
        // ---- A. gather the partial sums of bin kb-1 (granules tagged kb-1) into LDS
        if (!first) {
            const unsigned tag = (unsigned)(kb - 1);
            const unsigned long long* src = ll + (size_t)((kb - 1) & 1) * 2 * ndbl;
            unsigned long long w0[NV], w1[NV];
            for (int base = 0; base < ndbl; base += NV * PS_NT) {
                unsigned spins = 0;
                bool ok;
                do {
#pragma unroll
                    for (int i = 0; i < NV; ++i) {
                        const int f = base + tid + PS_NT * i;
                        if (f < ndbl) {
                            w0[i] = __hip_atomic_load(src + 2 * (size_t)f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            w1[i] = __hip_atomic_load(src + 2 * (size_t)f + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    }
                    ok = true;
#pragma unroll
                    for (int i = 0; i < NV; ++i) {
                        const int f = base + tid + PS_NT * i;
                        if (f < ndbl) ok = ok && (unsigned)(w0[i] >> 32) == tag && (unsigned)(w1[i] >> 32) == tag;
                    }
                    if (!ok) {
                        __builtin_amdgcn_s_sleep(1);
                        ++spins;
                        if ((spins & 255u) == 0 &&
                            __hip_atomic_load(a.abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)
                            spins = PS_SPIN_LIMIT;
                        if (spins >= PS_SPIN_LIMIT) {
                            __hip_atomic_store(a.abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            s_abort = 1;
                            break;
                        }
                    }
                } while (!ok);
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const int f = base + tid + PS_NT * i;
                    if (f < ndbl) {
                        const int wg = f / (4 * C), rem = f - wg * (4 * C);
                        const unsigned long long bits = (w0[i] & 0xffffffffull) | (w1[i] << 32);
                        stage_d[((size_t)(rem >> 1) * (nWG + 1) + wg) * 2 + (rem & 1)] = __longlong_as_double((long long)bits);
                    }
                }
            }
        }
        // ---- B. stage this bin's operands, then issue the next bin's loads
#pragma unroll
        for (int i = 0; i < NXV; ++i) {
            const int f = tid + PS_NT * i, c = f / DPW, dd = f % DPW;
            if (c < C) xs[(size_t)c * (DPW + 1) + dd] = xv[i];
        }
#pragma unroll
        for (int i = 0; i < NMV; ++i) {
            const int f = tid + PS_NT * i;
            if (f < C * C) ms[f] = mv[i];
        }
        const double habs_cur = habs;
        const bool prev_ok = first ? true : (a.cond_ok[kb - 1] != 0.0);
        const bool cur_ok = last ? true : (a.cond_ok[kb] != 0.0);
        __syncthreads();
        if (s_abort) break;
        if (!last) fetch(kb + 1);
        // ---- C. v_total = sum over workgroups;  W(kb-1,:) = v_total conj(M_{kb-1})
        for (int pair = tid >> 2; pair < 2 * C; pair += PS_NT >> 2) {
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
        for (int pair = tid >> 2; pair < 2 * C; pair += PS_NT >> 2) {
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
                if (member == 0 && !first) a.W[((int64_t)e * a.P + (kb - 1)) * C + c] = acc;
            }
        }
        if (last) break;
        __syncthreads();
        // ---- D. p = W(kb-1,:) pwGrid ;  t = |H| p/|p|
        if (tid < 2 * DPW) {
            cplx t = mk(0, 0);
            if (p1) {
                cplx pa = mk(0, 0), pb = mk(0, 0);
                int c = 0;
                for (; c + 1 < C; c += 2) {
                    cfma(pa, Wp[e_ * C + c], xs[(size_t)c * (DPW + 1) + dd_]);
                    cfma(pb, Wp[e_ * C + c + 1], xs[(size_t)(c + 1) * (DPW + 1) + dd_]);
                }
                if (c < C) cfma(pa, Wp[e_ * C + c], xs[(size_t)c * (DPW + 1) + dd_]);
                t = unit_phase(habs_cur, pa + pb, nyq);
            }
            ts[e_][dd_] = t;
        }
        __syncthreads();
        // ---- E. this slab's partial v = t conj(G) (or t Y_reg_inv for an ill-conditioned bin), published as granules
        {
            const int pair = tid >> 2, part = tid & 3;
            if (pair < 2 * C) {
                const int e = pair / C, c = pair % C;
                cplx a0 = mk(0, 0), a1 = mk(0, 0);
                if (cur_ok) {
                    const cplx* xrow = xs + (size_t)c * (DPW + 1);
#pragma unroll
                    for (int j = 0; j < DPW / 4; j += 2) {
                        cfma(a0, ts[e][part + 4 * j], conj(xrow[part + 4 * j]));
                        cfma(a1, ts[e][part + 4 * (j + 1)], conj(xrow[part + 4 * (j + 1)]));
                    }
                } else {
                    const cplx* Y = a.Yri + (int64_t)kb * a.g_stride + (int64_t)c * a.ldD;
                    for (int dd = part; dd < DPW; dd += 4)
                        if (d0 + dd < a.D) cfma(a0, ts[e][dd], Y[d0 + dd]);
                }
                const cplx acc = group_sum<4>(a0 + a1);
                if (part == 0) {
                    unsigned long long* dst = ll + (size_t)(kb & 1) * 2 * ndbl + 2 * ((size_t)member * 4 * C + 2 * pair);
                    ll_store(dst, acc.x, (unsigned)kb);
                    ll_store(dst + 2, acc.y, (unsigned)kb);
                }
            }
        }
        __syncthreads();  // xs / ts / stage are rewritten by the next bin
    }
}

template <int DPW, int NV>
void launch_one(const HalfSweepMulti& m, int nWG, size_t dyn, hipStream_t st) {
    auto kern = sweep_persist_kernel<DPW, NV>;
    HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn));
    kern<<<dim3(8 * nWG), PS_NT, dyn, st>>>(m, nWG);
    KERNEL_CHECK();
}

template <int DPW>
void launch_dpw(const HalfSweepMulti& m, int nWG, hipStream_t st) {
    const HalfSweepArgs& a = m.a[0];
    const size_t dyn = sizeof(cplx) * ((size_t)a.C * (DPW + 1) + (size_t)a.C * a.C + (size_t)2 * a.C * (nWG + 1));
    if (dyn > 150 * 1024) throw Error(2, "persistent sweep: shape not supported");
    const int per_thread = ceil_div(nWG * 4 * a.C, PS_NT);
    if (per_thread <= 12) launch_one<DPW, 12>(m, nWG, dyn, st);
    else if (per_thread <= 17) launch_one<DPW, 17>(m, nWG, dyn, st);
    else launch_one<DPW, 32>(m, nWG, dyn, st);  // more than 32 per thread: several passes
}

}  // namespace

int persist_sweep_dpw(int D) {
    static const int forced = [] {
        const char* e = getenv("EMAGLS_PERSIST_DPW");
        return e ? atoi(e) : 0;
    }();
    if (forced == 64 || forced == 96 || forced == 128) return forced;
    (void)D;
    return 64;
}
int persist_sweep_nwg(int D) { return ceil_div(D, persist_sweep_dpw(D)); }
size_t persist_sweep_ll_bytes(int D, int C) { return sizeof(unsigned long long) * 2 * 2 * (size_t)persist_sweep_nwg(D) * 4 * C; }

void launch_sweep_persist(const HalfSweepMulti& m, hipStream_t st) {
    const HalfSweepArgs& a = m.a[0];
    if (a.C > PS_CMAX || m.n > 8) throw Error(2, "persistent sweep: shape not supported");
    const int dpw = persist_sweep_dpw(a.D), nWG = ceil_div(a.D, dpw);
    switch (dpw) {
        case 64: launch_dpw<64>(m, nWG, st); break;
        case 96: launch_dpw<96>(m, nWG, st); break;
        default: launch_dpw<128>(m, nWG, st); break;
    }
}

}  // namespace emagls
