// Resident MagLS phase sweep with OPERAND SYNTHESIS: the slab of pwGrid_k.' that a workgroup needs in bin k is evaluated
// inside the launch, between two uses of the slab buffer, instead of being written to HBM by an earlier kernel and read back.
//
// Reference (lib/getEMagLsFilters.m:87-103, dependencies/getSMAIRMatrix.m:101-121): pwGrid_k = E diag(b_n(k)) Y^H with
// E = Y_mic (raw microphones) or pinv(Y_lo) Y_mic.  For the built-in real orthonormal SH basis the sum over the degrees of one
// order collapses by the addition theorem,
//     sum_m Y_nm(d) Y_nm(mic j) = (2n + 1) / (4 pi) P_n(x_dj),     x_dj = cos(angle between HRIR direction d and microphone j),
// so the raw-microphone operand is a polynomial in ONE scalar per (direction, microphone) pair,
//     g_k[d][j] = sum_n b_n(k) (2n + 1)/(4 pi) P_n(x_dj) = sum_m bsc[k][m] T_m(x_dj),
// evaluated on the Chebyshev basis (T_{m+1} = 2x T_m - T_{m-1}: one fused operation per term and pair, two more for the complex
// sum; the Legendre recurrence needs a fourth) after synth_coeff_kernel has converted every bin's series.  What a workgroup keeps is x_dj of its 96 directions x 32 microphones
// (12 values per producer thread, in registers) -- 24 KB -- where the materialised order terms QT_n would be 384 KB.  The
// SH-domain designs run the SAME chain in the microphone domain: G_k = g_k Pm^T (Pm = pinv(Y_lo), real) gives
//     p = W(k-1,:) pwGrid = (W Pm) g^T,   t conj(G) = (t conj(g)) Pm^T,
// so with u = t conj(g) (what the workgroups exchange) the chain is  w' = u conj(Mt_{k-1}),  Mt = Pm^T M Pm  (32 x 32, formed per
// bin by synth_mt_kernel), and the filters' rows W(k,:) = (u Pm^T) conj(M_k) follow after the launch (synth_rows_kernel) from the
// totals u(k) the chain stores.  HBM traffic of the sweep: M~ and |H| (34 KB per bin and design instead of 1.1 MB).
//
// Workgroup = 512 threads: waves 0-3 run the chain exactly as sweep_persist_kernel does (M phase, p phase, partial phase, wave 3
// the exchange), waves 4-7 are PRODUCERS: between barrier B4 of bin k-1 (the slab buffer is free) and B2 of bin k (the p
// phase reads it) they evaluate g_k into the buffer -- that window is the chain's exchange wait plus its M phase (2.3 us), in
// which the vector ALUs have nothing else to do; the 60 FP64 operations per pair and bin (at 20 orders) hide there.  The producers take part in all four barriers of a bin; b_n rows travel through a two-slot LDS
// ring that one producer wave refills a bin ahead.
#include "kernels.hpp"
#include "persist_common.hpp"
#include "synth_common.hpp"

namespace emagls {

namespace {

constexpr int SY_NT = 512, SY_CHAIN = 256, SY_COMM0 = 192;
constexpr int SY_MLD = 36;      // row stride of M~ in LDS (16 dwords mod 64)

// NI: quarters of microphone rows (4 NI rows: 8, 16 or 32); PS_DPW directions per workgroup
template <int PS_DPW, int NI>
__global__ void __launch_bounds__(SY_NT) sweep_synth_kernel(HalfSweepMulti m, int nWG) {
    constexpr int ROWS = 4 * NI;
    constexpr int XLD = PS_DPW + 4;     // row stride of the slab (16 dwords mod 64: conflict-free quarter-wave reads)
    constexpr int PUNR = PS_DPW == 96 ? 3 : 4;
    constexpr int NLM = (PS_CMAX * PS_CMAX) / 256;
    constexpr int NPAIR = PS_DPW * ROWS / 256;                  // pairs per producer thread
    constexpr int GS = NPAIR % 3 == 0 ? 3 : (NPAIR % 4 == 0 ? 4 : 2);
    constexpr int NGRP = NPAIR / GS;
    static_assert(NPAIR * 256 == PS_DPW * ROWS && NGRP * GS == NPAIR, "producer layout");
    __shared__ __attribute__((aligned(16))) cplx vt[64];          // totals of the previous bin, [ear][32] zero padded
    __shared__ __attribute__((aligned(16))) cplx Wp[64];          // w'(kb-1,:), same layout
    __shared__ __attribute__((aligned(16))) cplx ts[2][PS_DPW];   // t per ear and direction
    __shared__ __attribute__((aligned(16))) cplx ring[2][SY_NORD]; // scaled modal terms of two bins
    __shared__ int s_abort, s_local;
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    cplx* xs = reinterpret_cast<cplx*>(dyn);                              // [ROWS][XLD]   g_kb slab
    cplx* ms = xs + (size_t)ROWS * XLD;                                   // [ROWS][MLD]   M~_{kb-1}
    double* hs_all = reinterpret_cast<double*>(ms + (size_t)ROWS * SY_MLD);  // [2][2][DPW]  |H_kb| (two buffers)
    // block -> (design, member) as in sweep_persist_kernel (design-major within an XCD for 9-16 designs)
    const int two = m.n > 8 ? 1 : 0, rest = blockIdx.x >> 3;
    const int design = (blockIdx.x & 7) + 8 * ((two && rest >= nWG) ? 1 : 0), member = (two && rest >= nWG) ? rest - nWG : rest;
    if (design >= m.n || member >= nWG) return;
    const HalfSweepArgs& a = m.a[design];
    const int wtid = threadIdx.x;
    const bool producer = wtid >= SY_CHAIN;
    const int tid = wtid & (SY_CHAIN - 1), lane = tid & 63;
    const bool comm = !producer && tid >= SY_COMM0;
    const int C = a.C, P = a.P, npairs = 2 * a.C;     // C: microphones (the chain's channels)
    const int64_t d0 = (int64_t)member * PS_DPW;
    const int64_t na = P - a.kabs0;
    const int nord_pad = a.nord_pad;
    u64* part_ll = a.ll;
    u64* tot_ll = a.ll + (size_t)2 * nWG * 4 * npairs;
    const int nd2 = 2 * npairs;
    u64* xcc_ll = tot_ll + (size_t)2 * 2 * nd2;
    if (wtid < 64) { vt[wtid] = mk(0, 0); Wp[wtid] = mk(0, 0); }
    if (wtid == 0) { s_abort = 0; s_local = 0; }
    {
        const size_t ncplx = (size_t)ROWS * XLD + (size_t)ROWS * SY_MLD + PS_DPW * 2;
        for (size_t i = wtid; i < ncplx; i += SY_NT) xs[i] = mk(0, 0);
    }
    // the chain's waves issue no more than M~ (16 KB) and |H| (1.5 KB) per bin: their polls never wait behind a slab
    // (the producers above the waves of other batches' kernels that share the CU: the chain waits for them at B1 / B2)
    // Two designs share an XCD's CUs in launches of 9-16 designs, and the SIMDs serve the OLDER waves first: the design whose
    // workgroups were placed first ran at its solo speed (5.7 us per bin) and the other on what was left (8.0 us, hop 1 4.3 us:
    // profiles/r04_sweep_timing.md) -- the launch lasts as long as the slower one (2.63 / 3.42 ms).  synth_prio 5 (default) tilts the
    // chains' issue priorities the other way (second design 3, first design 2; producers 0): 2.97 / 3.20 ms.  Tilting the producers
    // too (3: 3.22 / 2.88 ms) overshoots, tilting only the producers (4) does nothing.
    const bool second = two && rest >= nWG;
    if (a.synth_prio == 3) {
        if (!producer) { if (second || !two) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2); }
        else if (second) __builtin_amdgcn_s_setprio(1);
    } else if (a.synth_prio == 4) {   // (experiment: only the producers tilted)
        if (!producer) __builtin_amdgcn_s_setprio(3); else if (second) __builtin_amdgcn_s_setprio(1);
    } else if (a.synth_prio == 5) {   // (experiment: only the chains tilted)
        if (!producer) { if (second || !two) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(2); }
    } else {
        if (!producer) __builtin_amdgcn_s_setprio(3); else if (a.synth_prio == 1) __builtin_amdgcn_s_setprio(1); else if (a.synth_prio == 2) __builtin_amdgcn_s_setprio(2);
    }

    if (producer) {
        // ================================ producers ================================
        // Units: the design's microphones in the chain's row order (a.smap: row -> microphone) are npr antipodal pairs (rows 2u,
        // 2u + 1) followed by nsg single microphones.  Unit slot q = tid + 256 i: unit u = q / DPW, direction dd = q % DPW
        // (consecutive lanes write consecutive elements of a row).
        const int* smap = a.smap;
        const int npr = smap[32], nsg = smap[33];
        const int total = (npr + nsg) * PS_DPW;
        const int ngrp_act = ((total + 255) / 256 + GS - 1) / GS;   // groups of unit slots in use (design-uniform)
        double x[NPAIR];
        const cplx* bsc = a.bsc;
        cplx brow = mk(0, 0);
        const bool rloader = tid < nord_pad;   // (lanes of wave 4)
        auto unit_row = [&](int u) { return u < npr ? 2 * u : 2 * npr + (u - npr); };
        __syncthreads();  // LDS is zeroed
        // twice the cosine of the angle between direction and (first) microphone, once per unit.  Evaluated in a rolled loop
        // through the (still unused) slab buffer: twelve inlined sincos expansions side by side would cost the kernel its
        // register budget
        {
            double* scratch = reinterpret_cast<double*>(xs);
#pragma unroll 1
            for (int i = 0; i < NPAIR; ++i) {
                const int q = tid + 256 * i;
                if (q < total) {
                    const int u = q / PS_DPW, dd = q % PS_DPW;
                    const int64_t dg = d0 + dd < a.D ? d0 + dd : a.D - 1;
                    const int jm = smap[unit_row(u)];
                    double sd, cd, sm, cm;
                    sincos(a.dir_zen[dg], &sd, &cd);
                    sincos(a.mic_zen ? a.mic_zen[jm] : 1.5707963267948966, &sm, &cm);
                    const double v = fma(sd * sm, cos(a.dir_azi[dg] - a.mic_azi[jm]), cd * cm);
                    scratch[q] = 2.0 * fmin(1.0, fmax(-1.0, v));   // (2x: the factor of the Chebyshev recurrence)
                }
            }
#pragma unroll
            for (int i = 0; i < NPAIR; ++i) {
                const int q = tid + 256 * i;
                x[i] = q < total ? scratch[q] : 0.0;
                if (q < total) scratch[q] = 0.0;
            }
        }
        if (rloader) ring[a.kfirst & 1][tid] = ldc(bsc + (int64_t)a.kfirst * nord_pad + tid);
        __syncthreads();  // (the chain's start-up exchange of XCC ids); ring row kfirst is in place
        // groups evaluated before barrier B1 (the exchange wait), the rest between B1 and B2 (the chain's M phase)
        const int ng1 = a.synth_split > 0 ? (ngrp_act * a.synth_split + 99) / 100 : ngrp_act;
        auto run_groups = [&](const cplx* bs, int g_lo, int g_hi) __attribute__((always_inline)) {
#pragma unroll
            for (int g = 0; g < NGRP; ++g) {
                if (g >= g_lo && g < g_hi) {
                    double xg[GS];
                    cplx accE[GS], accO[GS];
#pragma unroll
                    for (int i = 0; i < GS; ++i) xg[i] = x[g * GS + i];
                    synth_group<GS>(xg, accE, accO, bs, nord_pad);
#pragma unroll
                    for (int i = 0; i < GS; ++i) {
                        const int q = tid + 256 * (g * GS + i);
                        if (q < total) {
                            const int u = q / PS_DPW, dd = q % PS_DPW, o = unit_row(u) * XLD + dd;
                            xs[o] = accE[i] + accO[i];
                            if (u < npr) xs[o + XLD] = accE[i] - accO[i];
                        }
                    }
                }
            }
        };
        for (int kb = a.kfirst; kb <= P; ++kb) {
            const bool last = (kb == P);
            const cplx* bs = ring[kb & 1];
            if (!last) {
                if (rloader) brow = ldc(bsc + (int64_t)(kb + 1 < P ? kb + 1 : P - 1) * nord_pad + tid);
                run_groups(bs, 0, ng1);
            }
            __syncthreads();   // B1
            if (s_abort || last) break;
            run_groups(bs, ng1, ngrp_act);
            __syncthreads();  // B2: the slab of bin kb is complete
            if (rloader) ring[(kb + 1) & 1][tid] = brow;
            __syncthreads();  // B3
            __syncthreads();  // B4
        }
        return;
    }

    // ================================ chain ================================
    const int part = tid & 3;
    const int pair = tid >> 2;
    const bool pvalid = pair < npairs;
    const int e = pvalid ? pair / C : 0, c = pvalid ? pair % C : 0;
    // |H| of the next bin: thread lt < 2 DPW takes value lt (fetched after B1, staged after B4)
    const bool hloader = tid < 2 * PS_DPW;
    const int lt = hloader ? tid : 0;
    double hReg = 0.0;
    const int hoff = (int)((int64_t)((lt / PS_DPW) & 1) * na * a.ldH + (d0 + lt % PS_DPW < a.D ? d0 + lt % PS_DPW : a.D - 1));
    auto fetch_h = [&](int kb) __attribute__((always_inline)) {
        const int kbg = kb < P ? kb : P - 1;
        hReg = a.Habs[(int64_t)(kbg - a.kabs0) * a.ldH + hoff];
    };
    cplx mReg[NLM];
    auto fetch_m = [&](int kb, cplx (&mL)[NLM]) __attribute__((always_inline)) {
        const int kbm = kb - 1 > a.kfirst ? kb - 1 : a.kfirst;
        const cplx* M = a.Mw + (int64_t)kbm * C * C;
#pragma unroll
        for (int i = 0; i < NLM; ++i) mL[i] = ldc(M + tid + 256 * i);   // (the buffer is padded by 1024 elements)
    };
    auto stage_m = [&](const cplx (&mL)[NLM]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NLM; ++i) {
            const int f = tid + 256 * i;
            if (f < C * C) ms[(f / C) * SY_MLD + f % C] = mL[i];
        }
    };
    __syncthreads();  // LDS is zeroed
    if (hloader) { fetch_h(a.kfirst); hs_all[(size_t)(a.kfirst & 1) * 2 * PS_DPW + lt] = hReg; }
    fetch_m(a.kfirst, mReg);
    stage_m(mReg);
    if (comm) {  // do all workgroups of this design share an XCD?
        const unsigned xcc = read_xcc_id();
        const unsigned tag0 = 0x58434300u;  // 'XCC'
        if (lane == 0) ll_put(xcc_ll + member, ((u64)tag0 << 32) | xcc, false);
        u64 w = 0;
        const bool alive = ll_wait([&] {
            if (lane >= nWG) return true;
            w = ll_load(xcc_ll + lane);
            return ll_ok(w, tag0);
        }, a.abort_flag, a.wait_ticks);
        const bool same = lane >= nWG || (unsigned)w == xcc;
        if (lane == 0) {
            s_local = alive && __builtin_amdgcn_ballot_w64(!same) == 0 && a.force_global == 0;
            if (!alive) s_abort = 1;
            if (a.timing && member == 1) a.timing[15] = s_local;
        }
    }
    __syncthreads();
    const bool local = s_local != 0;

#define PSTAMP(i) do { if (a.timing && member == 1 && kb < P) a.timing[(int64_t)kb * 16 + (i)] = (long long)wall_clock64(); } while (0)
    for (int kb = a.kfirst; kb <= P; ++kb) {
        const bool first = (kb == a.kfirst);
        const bool last = (kb == P);  // only the totals of bin P-1 are left to store
        const bool nyq = (kb == P - 1);
        // ================= communication wave: the totals of bin kb-1 into vt =================
        if (comm) {
            if (lane == 0) PSTAMP(0);
            if (first) {   // W(kfirst-1,:) Pm from the least-squares bins (synth_init_kernel)
                if (lane < npairs) vt[(lane / C) * PS_CMAX + lane % C] = a.Winit[(lane / C) * PS_CMAX + lane % C];
            } else {
                const unsigned tag = (unsigned)(kb - 1);
                const int slot = (kb - 1) & 1;
                bool alive = true;
                unsigned spins1 = 0;
                // hop 1 (reduce-scatter): the pairs this workgroup owns -- q, q + nWG, q + 2 nWG in ONE pass (the microphone-domain
                // chain exchanges 64 pairs: with two per pass the first workgroups of a design needed a second pass, and every
                // workgroup waits for them in hop 2)
                for (int q = member; q < npairs && alive; q += 3 * nWG) {
                    const int q2 = q + nWG, q3 = q + 2 * nWG;
                    const bool twoq = q2 < npairs, threeq = q3 < npairs;
                    const u64* src = part_ll + (((size_t)slot * npairs + q) * nWG + lane) * 4;   // lane = producing workgroup
                    const u64* src2 = src + (size_t)nWG * nWG * 4;
                    const u64* src3 = src2 + (size_t)nWG * nWG * 4;
                    u64 w0 = 0, w1 = 0, w2 = 0, w3 = 0, u0 = 0, u1 = 0, u2 = 0, u3 = 0, t0 = 0, t1 = 0, t2 = 0, t3 = 0;
                    alive = ll_wait([&] {
                        if (lane >= nWG) return true;
                        w0 = ll_load(src); w1 = ll_load(src + 2); w2 = ll_load(src + 1); w3 = ll_load(src + 3);
                        bool ok = true;
                        if (twoq) {
                            u0 = ll_load(src2); u1 = ll_load(src2 + 2); u2 = ll_load(src2 + 1); u3 = ll_load(src2 + 3);
                            ok = ll_ok(u0, tag) && ll_ok(u1, tag) && ll_ok(u2, tag) && ll_ok(u3, tag);
                        }
                        if (threeq) {
                            t0 = ll_load(src3); t1 = ll_load(src3 + 2); t2 = ll_load(src3 + 1); t3 = ll_load(src3 + 3);
                            ok = ok && ll_ok(t0, tag) && ll_ok(t1, tag) && ll_ok(t2, tag) && ll_ok(t3, tag);
                        }
                        return ok && ll_ok(w0, tag) && ll_ok(w1, tag) && ll_ok(w2, tag) && ll_ok(w3, tag);
                    }, a.abort_flag, a.wait_ticks, &spins1, (a.timing && member == 1 && lane == 0 && q == member) ? &a.timing[(int64_t)kb * 16 + 6] : nullptr);
                    const double re = wave_sum(lane < nWG ? ll_value(w0, w1) : 0.0);
                    const double im = wave_sum(lane < nWG ? ll_value(w2, w3) : 0.0);
                    double re2 = 0.0, im2 = 0.0, re3 = 0.0, im3 = 0.0;
                    if (twoq) {
                        re2 = wave_sum(lane < nWG ? ll_value(u0, u1) : 0.0);
                        im2 = wave_sum(lane < nWG ? ll_value(u2, u3) : 0.0);
                    }
                    if (threeq) {
                        re3 = wave_sum(lane < nWG ? ll_value(t0, t1) : 0.0);
                        im3 = wave_sum(lane < nWG ? ll_value(t2, t3) : 0.0);
                    }
                    if (lane == 0) {
                        u64* dst = tot_ll + (size_t)slot * 2 * nd2 + 2 * q;
                        ll_store(dst, dst + nd2, re, tag, local);
                        ll_store(dst + 1, dst + nd2 + 1, im, tag, local);
                        if (twoq) {
                            ll_store(dst + 2 * nWG, dst + 2 * nWG + nd2, re2, tag, local);
                            ll_store(dst + 2 * nWG + 1, dst + 2 * nWG + nd2 + 1, im2, tag, local);
                        }
                        if (threeq) {
                            ll_store(dst + 4 * nWG, dst + 4 * nWG + nd2, re3, tag, local);
                            ll_store(dst + 4 * nWG + 1, dst + 4 * nWG + nd2 + 1, im3, tag, local);
                        }
                    }
                }
                if (lane == 0) PSTAMP(1);
                if (a.timing && member == 1 && lane == 0) a.timing[(int64_t)kb * 16 + 9] = spins1;
                // hop 2 (all-gather): every total; lane l takes the doubles l and l + 64
                if (alive) {
                    const u64* src = tot_ll + (size_t)slot * 2 * nd2;
                    const int x0 = lane, x1 = lane + 64;
                    u64 w0 = 0, w1 = 0, w2 = 0, w3 = 0;
                    alive = ll_wait([&] {
                        bool ok = true;
                        if (x0 < nd2) { w0 = ll_load(src + x0); w1 = ll_load(src + nd2 + x0); ok = ll_ok(w0, tag) && ll_ok(w1, tag); }
                        if (x1 < nd2) { w2 = ll_load(src + x1); w3 = ll_load(src + nd2 + x1); ok = ok && ll_ok(w2, tag) && ll_ok(w3, tag); }
                        return ok;
                    }, a.abort_flag, a.wait_ticks);
                    double* vd = reinterpret_cast<double*>(vt);
                    // double x = 2 (e C + c) + re/im  ->  padded slot 2 (32 e + c) + re/im
                    if (x0 < nd2) vd[x0 + ((x0 >> 1) >= C ? 2 * (PS_CMAX - C) : 0)] = ll_value(w0, w1);
                    if (x1 < nd2) vd[x1 + ((x1 >> 1) >= C ? 2 * (PS_CMAX - C) : 0)] = ll_value(w2, w3);
                }
                if (!alive && lane == 0) s_abort = 1;
            }
            if (lane == 0) PSTAMP(2);
        }
        const double* hs = hs_all + (size_t)(kb & 1) * 2 * PS_DPW;
        __syncthreads();  // B1: vt is complete
        if (s_abort) break;
        if (hloader) fetch_h(kb + 1);
        // ---- w'(kb-1,:) = u_total conj(M~_{kb-1})  (the start value as it stands); the totals themselves go to memory: the
        // filters' rows are formed from them after the launch
        if (pvalid) {
            cplx acc = mk(0, 0);
            if (first) {
                if (part == 0) acc = vt[e * PS_CMAX + c];
            } else {
#pragma unroll
                for (int i = 0; i < NI; ++i)  // vt and ms are 0 beyond C
                    cfma(acc, vt[e * PS_CMAX + part + 4 * i], conj(ms[(part + 4 * i) * SY_MLD + c]));
            }
            acc = group_sum<4>(acc);
            if (part == 0) {
                Wp[e * PS_CMAX + c] = acc;
                if (member == 0 && !first) a.U[((int64_t)e * P + (kb - 1)) * PS_CMAX + c] = vt[e * PS_CMAX + c];
            }
        }
        if (last) break;
        fetch_m(kb + 1, mReg);
        __syncthreads();  // B2: Wp is complete, and so is the slab of bin kb (producers)
        // ---- p = w'(kb-1,:) g^T ;  t = |H| p/|p|
        if (wtid == 0) PSTAMP(3);
        if (tid < 2 * PS_DPW) {
            const int dA = tid >> 2, dB = dA + PS_DPW / 2;
            cplx pA0 = mk(0, 0), pA1 = mk(0, 0), pB0 = mk(0, 0), pB1 = mk(0, 0);   // p[direction][ear]
#pragma unroll
            for (int i = 0; i < NI; ++i) {  // Wp and xs are 0 beyond C
                const cplx w0 = Wp[part + 4 * i], w1 = Wp[PS_CMAX + part + 4 * i];
                const cplx gA = xs[(part + 4 * i) * XLD + dA], gB = xs[(part + 4 * i) * XLD + dB];
                cfma(pA0, w0, gA); cfma(pA1, w1, gA); cfma(pB0, w0, gB); cfma(pB1, w1, gB);
            }
            pA0 = group_sum<4>(pA0); pA1 = group_sum<4>(pA1); pB0 = group_sum<4>(pB0); pB1 = group_sum<4>(pB1);
            if (part < 2) {   // lane part = ear
                ts[part][dA] = (d0 + dA < a.D) ? unit_phase(hs[part * PS_DPW + dA], part ? pA1 : pA0, nyq) : mk(0, 0);
                ts[part][dB] = (d0 + dB < a.D) ? unit_phase(hs[part * PS_DPW + dB], part ? pB1 : pB0, nyq) : mk(0, 0);
            }
        }
        __syncthreads();  // B3: ts is complete
        // ---- this slab's partial u = t conj(g), published as granules
        if (wtid == 0) PSTAMP(4);
        {
            const int cp = tid >> 4, ep = tid & 15;
            const int c0 = 2 * cp, c1 = c0 + 1;
            if (c0 < C) {
                cplx v00 = mk(0, 0), v01 = mk(0, 0), v10 = mk(0, 0), v11 = mk(0, 0);  // v[ear][channel]
                const cplx* x0 = xs + c0 * XLD, *x1 = xs + (c1 < C ? c1 : c0) * XLD;
#pragma unroll PUNR
                for (int j = 0; j < PS_DPW / 16; ++j) {
                    const int dd = ep + 16 * j;
                    const cplx t0 = ts[0][dd], t1 = ts[1][dd], g0 = conj(x0[dd]), g1 = conj(x1[dd]);
                    cfma(v00, t0, g0); cfma(v01, t0, g1); cfma(v10, t1, g0); cfma(v11, t1, g1);
                }
                v00 = group_sum<16>(v00); v01 = group_sum<16>(v01); v10 = group_sum<16>(v10); v11 = group_sum<16>(v11);
                const int ee = ep >> 3, ch = (ep >> 2) & 1, wi = ep & 3;
                const cplx acc = ee ? (ch ? v11 : v10) : (ch ? v01 : v00);
                const int cc = c0 + ch;
                if (cc < C) {
                    const u64 bits = (u64)__double_as_longlong((wi & 1) ? acc.y : acc.x);
                    const u64 word = ((u64)(unsigned)kb << 32) | ((wi & 2) ? (bits >> 32) : (bits & 0xffffffffull));
                    u64* dst = part_ll + (((size_t)(kb & 1) * npairs + (ee * C + cc)) * nWG + member) * 4 + wi;
                    ll_put(dst, word, local);
                }
            }
        }
        if (wtid == 0) PSTAMP(5);
        __syncthreads();  // B4: the slab buffer is free (the producers start on bin kb + 1), M~ and |H| of bin kb + 1 are staged
        if (hloader) hs_all[(size_t)((kb + 1) & 1) * 2 * PS_DPW + lt] = hReg;
        stage_m(mReg);
    }
#undef PSTAMP
}

// Chebyshev coefficients of the bin's Legendre series:  sum_n beta_n P_n(x) = sum_m bsc[m] T_m(x),  beta_n = b_n(kb) (2n + 1) / (4 pi),
//   bsc[m] = sum_{n >= m, n - m even} c_mn beta_n,   c_mn = (2 - delta_m0) lambda((n - m) / 2) lambda((n + m) / 2),
//   lambda(j) = prod_{i < j} (i + 1/2) / (i + 1)     (all c_mn in (0, 1]: the conversion amplifies nothing).
// Nyquist row real (getSMAIRMatrix.m:116); zero beyond the design's orders up to the even row length.
__global__ void __launch_bounds__(128) synth_coeff_kernel(const cplx* __restrict__ bn, int nOrd, int nord_pad, int P, cplx* __restrict__ bsc, size_t bstride) {
    bn = boff(bn, bstride); bsc = boff(bsc, bstride);
    __shared__ double lam[2 * SY_NORD];
    __shared__ cplx beta[SY_NORD];
    const int kb = blockIdx.x, m = threadIdx.x;
    if (m == 0) { double l = 1.0; for (int j = 0; j < 2 * SY_NORD; ++j) { lam[j] = l; l *= ((double)j + 0.5) / ((double)j + 1.0); } }
    if (m < nOrd) {
        cplx v = bn[(int64_t)kb * nOrd + m];
        if (kb == P - 1) v.y = 0.0;
        const double sc = (double)(2 * m + 1) / (4.0 * kPi);
        beta[m] = mk(v.x * sc, v.y * sc);
    }
    __syncthreads();
    if (m >= nord_pad) return;
    cplx acc = mk(0.0, 0.0);
    for (int n = m; n < nOrd; n += 2) {
        const double c = (m == 0 ? 1.0 : 2.0) * lam[(n - m) / 2] * lam[(n + m) / 2];
        acc.x = fma(c, beta[n].x, acc.x); acc.y = fma(c, beta[n].y, acc.y);
    }
    bsc[(int64_t)kb * nord_pad + m] = acc;
}

// Pm[c][r] (C_out x 32, zero padded): column r is the microphone smap[r] of the chain's row order -- from the complex copy of
// pinv(Y_lo) (ld ldZ), or of the identity for raw microphones
__global__ void __launch_bounds__(256) synth_pm_kernel(const cplx* __restrict__ Zlo, int ldZ, int nOut, int M, const int* __restrict__ smap, double* __restrict__ Pm,
                                                       size_t bstride) {
    Zlo = boff(Zlo, bstride); Pm = boff(Pm, bstride); smap = boff(smap, bstride);
    for (int idx = threadIdx.x; idx < PS_CMAX * PS_CMAX; idx += 256) {
        const int c = idx / PS_CMAX, r = idx % PS_CMAX;
        double v = 0.0;
        if (c < nOut && r < M) { const int j = smap[r]; v = Zlo ? Zlo[(int64_t)c * ldZ + j].x : (c == j ? 1.0 : 0.0); }
        Pm[idx] = v;
    }
}

// one workgroup per bin kb in [k0, P): Mt[kb] = Pm^T M_kb Pm  (M x M, row-major like M_kb);  kb == k0 - 1 (blockIdx 0): the chain's
// start value Winit[e][j] = sum_c W[e][k0-1][c] Pm[c][j]
__global__ void __launch_bounds__(256) synth_mt_kernel(const cplx* __restrict__ Mw, const double* __restrict__ Pm, int nOut, int M, int k0, int P,
                                                       const cplx* __restrict__ W, cplx* __restrict__ Mt, cplx* __restrict__ Winit, size_t bstride) {
    Mw = boff(Mw, bstride); Pm = boff(Pm, bstride); W = boff(W, bstride); Mt = boff(Mt, bstride); Winit = boff(Winit, bstride);
    __shared__ double pm[PS_CMAX][PS_CMAX + 1];
    __shared__ cplx tmp[PS_CMAX][PS_CMAX + 1];   // M Pm: [c'][j]
    const int tid = threadIdx.x;
    for (int idx = tid; idx < PS_CMAX * PS_CMAX; idx += 256) pm[idx / PS_CMAX][idx % PS_CMAX] = Pm[idx];
    __syncthreads();
    if (blockIdx.x == 0) {
        if (tid < 2 * PS_CMAX) {
            const int e = tid / PS_CMAX, j = tid % PS_CMAX;
            cplx acc = mk(0, 0);
            if (j < M) for (int c = 0; c < nOut; ++c) cfma(acc, W[((int64_t)e * P + (k0 - 1)) * nOut + c], pm[c][j]);
            Winit[tid] = acc;
        }
        return;
    }
    const int kb = k0 + (int)blockIdx.x - 1;
    const cplx* Mk = Mw + (int64_t)(kb - 1) * nOut * nOut;   // (the factor stage stores bin kb at slot kb - 1)
    for (int idx = tid; idx < nOut * M; idx += 256) {
        const int cp = idx / M, j = idx % M;
        cplx acc = mk(0, 0);
        for (int c = 0; c < nOut; ++c) cfma(acc, Mk[(int64_t)cp * nOut + c], pm[c][j]);
        tmp[cp][j] = acc;
    }
    __syncthreads();
    cplx* out = Mt + (int64_t)(kb - 1) * M * M;
    for (int idx = tid; idx < M * M; idx += 256) {
        const int j = idx / M, j2 = idx % M;
        cplx acc = mk(0, 0);
        for (int cp = 0; cp < nOut; ++cp) cfma(acc, tmp[cp][j2], pm[cp][j]);
        out[idx] = acc;
    }
}

// the chain's start value of one HRIR set on another plan's geometry (geometry-sharing batches): Winit[e][j] = sum_c W[e][k0-1][c] Pm[c][j]
__global__ void __launch_bounds__(64) synth_winit_kernel(const cplx* __restrict__ W, const double* __restrict__ Pm, int nOut, int M, int k0, int P,
                                                         cplx* __restrict__ Winit, size_t bstride, size_t gstride) {
    W = boff(W, bstride); Winit = boff(Winit, bstride); Pm = boff(Pm, gstride);
    const int tid = threadIdx.x, e = tid / PS_CMAX, j = tid % PS_CMAX;
    cplx acc = mk(0, 0);
    if (j < M) for (int c = 0; c < nOut; ++c) cfma(acc, W[((int64_t)e * P + (k0 - 1)) * nOut + c], Pm[c * PS_CMAX + j]);
    Winit[tid] = acc;
}

// Least-squares bins on the Gram route (lib/getEMagLsFilters.m:94 for the bins below k_cut that the Gram route serves):
// W(k,:) = H(k,:) Y_reg_inv_k = ((H(k,:) conj(g_k)) Pm^T) conj(M_k) -- the same form as the swept bins' rows with
// u(k) = H(k,:) conj(g_k), so these bins need no materialised operand either.  One workgroup = one unit (a microphone or an
// antipodal pair) x one chunk of 1024 directions; a thread keeps 2 cos of its four directions and walks the bins; the partial
// sums over a chunk's directions go to Upart[chunk][e][kb][row] (summed in chunk order by synth_rows_kernel: reproducible).
constexpr int SL_DIRS = 4;   // directions per thread
__global__ void __launch_bounds__(256) synth_ls_kernel(const cplx* __restrict__ Hc, int64_t ldH, int n_c, const cplx* __restrict__ bsc, int nord_pad,
                                                       const double* __restrict__ dir_azi, const double* __restrict__ dir_zen,
                                                       const double* __restrict__ mic_azi, const double* __restrict__ mic_zen, const int* __restrict__ smap,
                                                       int D, int P, int kb_lo, int kb_hi, cplx* __restrict__ Upart, size_t bstride, size_t gstride, int xcd_runs) {
    // (every (unit, chunk) workgroup of a lane reads the lane's spectra of the least-squares bins: XCD-aware order, xcd_run_index)
    unsigned zl = blockIdx.z, bxu = blockIdx.x, byu = blockIdx.y;
    if (xcd_runs) { unsigned tl; xcd_run_index(tl, zl); byu = tl / gridDim.x; bxu = tl - byu * gridDim.x; }
    Hc = boffz(Hc, bstride, zl); Upart = boffz(Upart, bstride, zl); dir_azi = boffz(dir_azi, bstride, zl); dir_zen = boffz(dir_zen, bstride, zl);
    mic_azi = boffz(mic_azi, bstride, zl); mic_zen = boffz(mic_zen, bstride, zl); smap = boffz(smap, bstride, zl); bsc = boffz(bsc, gstride, zl);
    __shared__ __attribute__((aligned(16))) cplx bs[SY_NORD];
    __shared__ cplx red[4][4];
    const int npr = smap[32], nsg = smap[33];
    const int u = (int)bxu, chunk = (int)byu, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (u >= npr + nsg) return;
    const int row = u < npr ? 2 * u : 2 * npr + (u - npr);
    const bool paired = u < npr;
    const int jm = smap[row];
    double sm, cm;
    sincos(mic_zen[jm], &sm, &cm);
    const double maz = mic_azi[jm];
    double x2[SL_DIRS];
    int dd[SL_DIRS];
#pragma unroll 1
    for (int i = 0; i < SL_DIRS; ++i) {
        const int d = chunk * 256 * SL_DIRS + i * 256 + tid;
        dd[i] = d < D ? d : -1;
        const int dc = d < D ? d : D - 1;
        double sd, cd;
        sincos(dir_zen[dc], &sd, &cd);
        const double v = fma(sd * sm, cos(dir_azi[dc] - maz), cd * cm);
        x2[i] = 2.0 * fmin(1.0, fmax(-1.0, v));
    }
    for (int kb = kb_lo; kb < kb_hi; ++kb) {
        __syncthreads();   // (the previous bin's readers of bs / red are done)
        if (tid < nord_pad) bs[tid] = bsc[(int64_t)kb * nord_pad + tid];
        __syncthreads();
        cplx accE[SL_DIRS], accO[SL_DIRS];
        synth_group<SL_DIRS>(x2, accE, accO, bs, nord_pad);
        cplx a0 = mk(0, 0), a1 = mk(0, 0), b0 = mk(0, 0), b1 = mk(0, 0);   // [row A / row B][ear]
#pragma unroll
        for (int i = 0; i < SL_DIRS; ++i) {
            if (dd[i] >= 0) {
                const cplx h0 = Hc[((int64_t)0 * n_c + kb) * ldH + dd[i]], h1 = Hc[((int64_t)1 * n_c + kb) * ldH + dd[i]];
                const cplx gA = conj(accE[i] + accO[i]), gB = conj(accE[i] - accO[i]);
                cfma(a0, h0, gA); cfma(a1, h1, gA); cfma(b0, h0, gB); cfma(b1, h1, gB);
            }
        }
        a0 = wave_sum(a0); a1 = wave_sum(a1); b0 = wave_sum(b0); b1 = wave_sum(b1);
        if (lane == 0) { red[wave][0] = a0; red[wave][1] = a1; red[wave][2] = b0; red[wave][3] = b1; }
        __syncthreads();
        if (tid < 4) {
            const cplx t = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
            const int e = tid & 1, r = row + (tid >> 1);
            if (tid < 2 || paired) Upart[(((int64_t)chunk * 2 + e) * P + kb) * PS_CMAX + r] = t;
        }
    }
}

// filters' rows from the chain's totals: W[e][kb][c] = sum_c' (sum_j U[e][kb][j] Pm[c'][j]) conj(M_kb[c'][c]),  kb in [k0, P)
__global__ void __launch_bounds__(64) synth_rows_kernel(const cplx* __restrict__ U, int nchunks, const double* __restrict__ Pm, const cplx* __restrict__ Mw, int nOut, int M,
                                                        int k0, int P, cplx* __restrict__ W, size_t bstride, size_t gstride) {
    U = boff(U, bstride); Pm = boff(Pm, gstride); Mw = boff(Mw, gstride); W = boff(W, bstride);   // (gstride 0: one geometry for all lanes)
    __shared__ cplx v[2][PS_CMAX];
    __shared__ cplx us[2][PS_CMAX];
    const int kb = k0 + (int)blockIdx.x, tid = threadIdx.x;
    const int e = tid / PS_CMAX, cq = tid % PS_CMAX;
    {   // u(kb): the chain's totals (one chunk), or the sum of the direction chunks' partial sums in chunk order (least-squares bins)
        cplx t = mk(0, 0);
        for (int ch = 0; ch < nchunks; ++ch) t += U[(((int64_t)ch * 2 + e) * P + kb) * PS_CMAX + cq];
        us[e][cq] = cq < M ? t : mk(0, 0);
    }
    __syncthreads();
    cplx acc = mk(0, 0);
    if (cq < nOut) for (int j = 0; j < M; ++j) cfma(acc, us[e][j], Pm[cq * PS_CMAX + j]);
    v[e][cq] = acc;
    __syncthreads();
    if (cq < nOut) {
        const cplx* Mk = Mw + (int64_t)(kb - 1) * nOut * nOut;
        cplx w = mk(0, 0);
        for (int cp = 0; cp < nOut; ++cp) cfma(w, v[e][cp], conj(Mk[(int64_t)cp * nOut + cq]));
        W[((int64_t)e * P + kb) * nOut + cq] = w;
    }
}

}  // namespace

bool synth_sweep_supported(int D, int nmics, int nOrd) {
    return nmics >= 2 && nmics <= PS_CMAX && persist_sweep_nwg(D) <= 32 && nOrd <= SY_NORD;
}
int synth_nord_pad(int nOrd) { return (nOrd + 1) & ~1; }

void launch_synth_prepare(const void* bn, int nOrd, int P, void* bsc, const void* Zlo, int ldZ, int nOut, int M, const int* smap, double* Pm, hipStream_t st) {
    const int np = synth_nord_pad(nOrd);
    synth_coeff_kernel<<<bgrid((unsigned)P), np <= 64 ? 64 : 128, 0, st>>>((const cplx*)bn, nOrd, np, P, (cplx*)bsc, batch_ctx().stride);   // (thread = order)
    KERNEL_CHECK();
    synth_pm_kernel<<<bgrid(1), 256, 0, st>>>((const cplx*)Zlo, ldZ, nOut, M, smap, Pm, batch_ctx().stride);
    KERNEL_CHECK();
}
void launch_synth_mt(const void* Mw, const double* Pm, int nOut, int M, int k0, int P, const void* W, void* Mt, void* Winit, hipStream_t st) {
    if (k0 >= P) return;
    synth_mt_kernel<<<bgrid((unsigned)(P - k0 + 1)), 256, 0, st>>>((const cplx*)Mw, Pm, nOut, M, k0, P, (const cplx*)W, (cplx*)Mt, (cplx*)Winit, batch_ctx().stride);
    KERNEL_CHECK();
}
void launch_synth_winit(const void* W, const void* Pm, int nOut, int M, int k0, int P, void* Winit, hipStream_t st, bool shared_geometry) {
    if (k0 >= P) return;
    synth_winit_kernel<<<bgrid(1), 64, 0, st>>>((const cplx*)W, (const double*)Pm, nOut, M, k0, P, (cplx*)Winit, batch_ctx().stride,
                                                shared_geometry ? 0 : batch_ctx().stride);
    KERNEL_CHECK();
}
int synth_ls_chunks(int D) { return (int)ceil_div(D, 256 * SL_DIRS); }
void launch_synth_ls(const void* Hc, int64_t ldH, int n_c, const void* bsc, int nord_pad, const double* dir_azi, const double* dir_zen, const double* mic_azi,
                     const double* mic_zen, const int* smap, int D, int M, int P, int kb_lo, int kb_hi, void* Upart, hipStream_t st, bool shared_geometry) {
    if (kb_hi <= kb_lo) return;
    synth_ls_kernel<<<bgrid(dim3((unsigned)M, (unsigned)synth_ls_chunks(D))), 256, 0, st>>>((const cplx*)Hc, ldH, n_c, (const cplx*)bsc, nord_pad, dir_azi, dir_zen,
                                                                                          mic_azi, mic_zen, smap, D, P, kb_lo, kb_hi, (cplx*)Upart, batch_ctx().stride,
                                                                                          shared_geometry ? 0 : batch_ctx().stride, xcd_runs_enabled());
    KERNEL_CHECK();
}
void launch_synth_rows(const void* U, int nchunks, const void* Pm, const void* Mw, int nOut, int M, int k_lo, int k_hi, int P, void* W, hipStream_t st,
                       bool shared_geometry) {
    if (k_lo >= k_hi) return;
    synth_rows_kernel<<<bgrid((unsigned)(k_hi - k_lo)), 64, 0, st>>>((const cplx*)U, nchunks, (const double*)Pm, (const cplx*)Mw, nOut, M, k_lo, P, (cplx*)W,
                                                                batch_ctx().stride, shared_geometry ? 0 : batch_ctx().stride);
    KERNEL_CHECK();
}

static void synth_set_attributes() {
    static PerDeviceOnce attr_once;   // (function attributes are per device)
    if (attr_once.first()) {
#define EMAGLS_SY_ATTR(D, N) HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(sweep_synth_kernel<D, N>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024))
        EMAGLS_SY_ATTR(64, 2); EMAGLS_SY_ATTR(64, 4); EMAGLS_SY_ATTR(64, 8);
        EMAGLS_SY_ATTR(96, 2); EMAGLS_SY_ATTR(96, 4); EMAGLS_SY_ATTR(96, 8);
#undef EMAGLS_SY_ATTR
    }
}
static int synth_ni(int M) { return M <= 8 ? 2 : (M <= 16 ? 4 : 8); }
static size_t synth_dyn_bytes(int dpw, int ni) {
    const size_t rows = 4 * (size_t)ni;
    return sizeof(cplx) * (rows * (dpw + 4) + rows * SY_MLD + (size_t)2 * dpw);
}
static const void* synth_kernel_ptr(int dpw, int ni) {
#define EMAGLS_SY_PTR(D, N) reinterpret_cast<const void*>(sweep_synth_kernel<D, N>)
    if (dpw == 64) return ni == 2 ? EMAGLS_SY_PTR(64, 2) : ni == 4 ? EMAGLS_SY_PTR(64, 4) : EMAGLS_SY_PTR(64, 8);
    return ni == 2 ? EMAGLS_SY_PTR(96, 2) : ni == 4 ? EMAGLS_SY_PTR(96, 4) : EMAGLS_SY_PTR(96, 8);
#undef EMAGLS_SY_PTR
}
// residency of the synthesising sweep, decided before the launch like persist_sweep_fits
bool synth_sweep_fits(int D, int nmics, int nOrd, int ndesigns) {
    if (!synth_sweep_supported(D, nmics, nOrd) || ndesigns < 1 || ndesigns > SWEEP_MULTI_MAX) return false;
    synth_set_attributes();
    const int dpw = persist_sweep_dpw(D), ni = synth_ni(nmics);
    int occ = 0;
    HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, synth_kernel_ptr(dpw, ni), SY_NT, synth_dyn_bytes(dpw, ni)));
    const int per_xcd = (ndesigns > 8 ? 2 : 1) * persist_sweep_nwg(D);
    return per_xcd <= occ * (sweep_cu_budget() / 8);
}

void launch_sweep_synth(const HalfSweepMulti& m, hipStream_t st) {
    const HalfSweepArgs& a = m.a[0];
    const int nWG = persist_sweep_nwg(a.D);
    if (!synth_sweep_supported(a.D, a.C, a.nord_pad) || m.n > SWEEP_MULTI_MAX) throw Error(2, "synthesising sweep: shape not supported");
    const unsigned nblocks = 8u * (unsigned)nWG * (m.n > 8 ? 2u : 1u);
    const int dpw = persist_sweep_dpw(a.D);
    const int ni = synth_ni(a.C);
    const size_t dyn = synth_dyn_bytes(dpw, ni);
    synth_set_attributes();
#define EMAGLS_SY_GO(D, N) sweep_synth_kernel<D, N><<<dim3(nblocks), SY_NT, dyn, st>>>(m, nWG)
    if (dpw == 64) { if (ni == 2) EMAGLS_SY_GO(64, 2); else if (ni == 4) EMAGLS_SY_GO(64, 4); else EMAGLS_SY_GO(64, 8); }
    else { if (ni == 2) EMAGLS_SY_GO(96, 2); else if (ni == 4) EMAGLS_SY_GO(96, 4); else EMAGLS_SY_GO(96, 8); }
#undef EMAGLS_SY_GO
    KERNEL_CHECK();
}

}  // namespace emagls
