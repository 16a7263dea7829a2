// Persistent form of the halved MagLS phase sweep: ONE launch walks all swept bins of up to 16 designs.
//
// Reference recurrence (lib/getEMagLsFilters.m:95-103): W(k,:) depends on W(k-1,:), so the bins are a chain of
// P - k_cut dependent steps.  The launch-per-bin kernels (sweep.hip) pay a kernel boundary per step: cold L2s
// and >= 4.5 us.  Here the workgroups of one design stay resident and all-reduce their partial sums through
// memory inside the launch, in two small hops (reduce-scatter, all-gather):
//
//   * a design's workgroups are the blocks b with b % 8 == design (the dispatcher is observed to place block b
//     on XCD b % 8; this is a speed assumption only, the protocol is placement independent);
//   * every number that crosses workgroups travels as two 8-byte {payload32, tag32} granules, each written by ONE
//     relaxed agent-scope store (sc1, write-through) and read by relaxed agent-scope loads (sc1, bypass L1).  A
//     granule is valid as soon as its tag equals the bin number: no fence, flag or barrier is needed;
//   * hop 1: a workgroup publishes its partial W(k,:) (2C complex numbers); the pair (ear, channel) q is owned
//     by workgroup q % nWG, whose communication wave collects the nWG partials of q (one per lane), sums them
//     with a fixed-order wave reduction (bitwise reproducible) and publishes the total;
//     hop 2: every communication wave collects the 2C totals (1.6 KB) into LDS.
//     (A single hop in which every workgroup reads every partial costs nWG x 1.6 KB per workgroup and bin:
//     measured 8.9 us per bin for 43 workgroups, against 7.4 us for the launch-per-bin kernel.)
//   * at start-up the workgroups of a design exchange their XCC ids with that (placement independent) protocol.
//     If all of them run on ONE XCD - the observed placement - the granules are afterwards written with plain
//     stores: the line stays in the XCD's shared L2, where the peers' sc1 loads (which bypass only L1) find it, so
//     a hop costs an L2 round trip instead of a fabric round trip.  Otherwise the sc1 stores stay: a wrong
//     placement guess is slower, never wrong;
//   * two granule slots (bin parity) suffice: a workgroup publishes bin k+2 only after it has read the totals of
//     bin k+1, which exist only after every owner has read the partials of bin k+1, i.e. is done with bin k;
//   * waves 0-3 compute, wave 4 communicates: the polls never sit behind the operand loads in a wave's in-order
//     memory queue.  The operands of the next bin (G slab in two register layouts, M, |H|) do not depend on the
//     chain; the compute waves fetch them into registers at the end of a bin;
//   * a communication wave that waits longer than the spin limit sets a sticky abort flag and everybody leaves:
//     a missing peer (not co-resident, killed) produces an error return, never a hang.
#include "kernels.hpp"
#include "persist_common.hpp"

namespace emagls {

namespace {

// Thread roles (PS_NT = 256 threads, 4 waves -- one per SIMD, so two workgroups always fit a CU side by side):
//   tid < 2 DPW          p phase: (direction pair, channel quarter); the same waves load: they request the next bin's
//                        operands when a bin starts / when its partial phase starts (never while the communication wave
//                        polls) and move them into the LDS buffer once the bin's partial phase is over (barrier B4)
//   tid < 256            M and partial phases: (pair, quarter); fetch and stage M
//   tid >= 192 (wave 3)  communication wave: besides its share of the M and partial phases it runs the exchange (it issues
//                        no operand loads of G, so its polls never queue behind an HBM miss in the wave's in-order memory queue)
// One LDS buffer (77 KB) and four waves per slab.
// NH = 1: one slab per workgroup (256 threads); a launch sweeps up to 8 designs, one per XCD.
// NH = 2 ("twin" workgroups, 512 threads, 9-16 designs per launch, two per XCD): a workgroup carries TWO slabs of the same
//   design -- threads 0-255 the slab 2 member, threads 256-511 the slab 2 member + 1 -- and publishes two partials per pair,
//   so the exchange sees 2 nWG producers as before; the halves share the communication wave, M, vt and Wp.  Two
//   independent 256-thread workgroups of DIFFERENT designs on one CU (the round-3 form of 16-design launches) cost
//   +0.5..0.9 us in hop 1 and +0.5 us in hop 2 per bin (5.7-6.0 us against 4.36 us, tools/sweep_timing.py 16): a CU's vector
//   memory pipeline returns in order across its waves, so one design's polls queued behind the other design's operand
//   loads (HBM misses), whose phases are not aligned with its own.  The two slabs of a twin move in lock step: their loads
//   are issued after B1 / B3 and have drained when the exchange starts, exactly as with one slab per CU.
constexpr int PS_NT = 256, PS_COMM0 = 192;
constexpr int PS_PMAX = 1040;   // bins whose conditioning flags fit the kernel's LDS table
constexpr int PS_MLD = 36;  // row stride of M in LDS (16 dwords mod 64)

// NI: channel quarters that are swept in the M and p phases (channels part + 4 i, i < NI): ceil(C / 4), so that an 8-microphone
// design (FromAtf) does a quarter of the multiply-adds of a 32-channel one instead of multiplying zeros
// (tried: amdgpu_waves_per_eu(3) / (4), i.e. 168 / 128 instead of ~200 VGPRs, so that kernels of other batches with up to four
// waves per SIMD fit next to a twin workgroup: 21-40 / 160-179 spilled registers on the chain, 5.6 instead of 4.4 us per bin with
// 8 designs and 8.2 instead of 6.1 with 16 -- 1455 sets/s at 20 steps, 1900 in long runs, 3100 for HRIR-set batches: rejected)
// (tried at the end of round 3: W(kb-1,:) = v conj(M) formed by the communication wave itself right after hop 2 -- 2 C lanes, one
// output each, four independent chains of C / 4 complex multiply-adds from LDS, `ms` published by an LDS arrival counter --
// so that the M phase and the barrier B2 go away: bit-identical filters, 2.35 against 2.07 ms per 8-design sweep and 3.35
// against 2.99 ms per 16-design sweep, i.e. +0.6 us per bin.  One wave issues the 2 x 28 LDS reads and 112 FP64 FMAs that the
// M phase spreads over four SIMDs, and that costs more than the two barriers it saves: rejected)
template <int PS_DPW, int NI, int NH>
__global__ void __launch_bounds__(PS_NT * NH) sweep_persist_kernel(HalfSweepMulti m, int nWG) {
    constexpr int NTT = PS_NT * NH;     // threads of the workgroup
    constexpr int ROWS = 4 * NI;        // channel rows kept in LDS (the quarters that are swept; rows beyond C stay zero)
    constexpr int XLD = PS_DPW + 4;     // row stride of the G slab (16 dwords mod 64: conflict-free quarter-wave reads)
    constexpr int PUNR = PS_DPW == 96 ? 3 : 4;
    constexpr int PS_NL = 2 * PS_DPW;   // loader threads = the p-phase threads
    constexpr int RG = 8, CH = RG * PS_DPW / PS_NL, NG = PS_CMAX / RG;  // G: NG row groups x CH chunks per loader thread
    static_assert(CH * PS_NL == RG * PS_DPW && NG * RG == PS_CMAX, "loader layout");
    constexpr int NLM = (PS_CMAX * PS_CMAX) / 256;                      // M: loads per M-phase thread
    static_assert(2 * PS_DPW <= PS_COMM0 && PS_DPW >= 64, "role layout");
    __shared__ __attribute__((aligned(16))) cplx vt[64];          // totals of the previous bin, [ear][32] zero padded
    __shared__ __attribute__((aligned(16))) cplx Wp[64];          // W(kb-1,:), same layout
    __shared__ __attribute__((aligned(16))) cplx ts_all[NH][2][PS_DPW];   // t per slab, ear and direction
    __shared__ int s_abort, s_local;
    __shared__ unsigned char s_ok[PS_PMAX];    // cond_ok of every bin (the designs stop at nfft 2048: P <= 1025; checked at launch)
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    // operands of the current bin; rows beyond C stay zero
    cplx* xs0 = reinterpret_cast<cplx*>(dyn);                           // [NH][ROWS][XLD]   G_kb slabs
    cplx* ms = xs0 + (size_t)NH * ROWS * XLD;                           // [ROWS][MLD]   M_{kb-1}
    double* hs0 = reinterpret_cast<double*>(ms + (size_t)ROWS * PS_MLD);  // [NH][2][2][DPW]  |H_kb| (two buffers per slab: 3 KB)
    // up to 8 designs: block b serves design b & 7 (the dispatcher is observed to place block b on XCD b % 8: one design per
    // XCD); 9 to 16 designs: designs j and j + 8 share XCD j, two workgroups per CU
    // Design-major within an XCD: the nWG blocks of design j come before those of design j + 8.  Workgroups are placed in block
    // order; when other kernels hold part of the CUs and only some of an XCD's 2 nWG blocks find room, the resident ones then
    // always include a COMPLETE design, which runs, finishes and makes room for the other.  (Interleaved, half of each design
    // could become resident and both would wait for peers that the other's workgroups keep out: the stall-until-time-out that
    // kept batches above 8 designs opt-in in round 2.)
    const int two = m.n > 8 ? 1 : 0, rest = blockIdx.x >> 3;
    const int design = (blockIdx.x & 7) + 8 * ((two && rest >= nWG) ? 1 : 0), member = (two && rest >= nWG) ? rest - nWG : rest;
    if (design >= m.n || member >= nWG) return;
    const HalfSweepArgs& a = m.a[design];
    if (a.skip_flag && __hip_atomic_load(a.skip_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return;   // (uniform per design)
    // The chain is latency bound and its waves sleep most of the time; kernels of other batches share the CU.  Highest issue
    // priority for the chain's waves: when they have work they get the next slot, at no cost to the others while they wait.
    __builtin_amdgcn_s_setprio(3);
    // wtid: thread of the workgroup; tid: thread of its slab (half); the roles below are per slab
    const int wtid = threadIdx.x, half = NH == 1 ? 0 : (wtid >> 8), tid = wtid & (PS_NT - 1), lane = tid & 63;
    const bool comm = half == 0 && tid >= PS_COMM0;
    const bool loader = tid < PS_NL;
    const int C = a.C, P = a.P, npairs = 2 * a.C;
    const int fm = a.fetch_mode;
    const int nProd = NH * nWG;                        // producers of partials (slabs) of this design
    const int slab = NH * member + half;
    const int64_t d0 = (int64_t)slab * PS_DPW;
    cplx* xs = xs0 + (size_t)half * ROWS * XLD;
    double* hs_all = hs0 + (size_t)half * 4 * PS_DPW;
    cplx (*ts)[PS_DPW] = ts_all[half];
    const int64_t na = P - a.kabs0;
    // totals: granule words of a double x (x = 2 pair + re/im): low halves lo[x], high halves hi[x], each array contiguous
    u64* part_ll = a.ll;                                    // [2][2C][nProd][re lo, im lo, re hi, im hi]: an owner reads nProd x 32 contiguous bytes
    u64* tot_ll = a.ll + (size_t)2 * nProd * 4 * npairs;    // [2][lo 4C | hi 4C]
    const int nd2 = 2 * npairs;                             // doubles per workgroup and bin
    u64* xcc_ll = tot_ll + (size_t)2 * 2 * nd2;             // [nWG] start-up exchange of the XCC ids
    if (wtid < 64) { vt[wtid] = mk(0, 0); Wp[wtid] = mk(0, 0); }
    if (wtid == 0) { s_abort = 0; s_local = 0; }
    // (the per-bin flags are read in every bin of the chain: once from memory, then from LDS)
    for (int i = wtid; i < PS_PMAX; i += NTT) s_ok[i] = (i < P) ? (a.cond_ok[i] != 0.0 ? 1 : 0) : 1;
    {
        const size_t ncplx = (size_t)NH * ROWS * XLD + (size_t)ROWS * PS_MLD + NH * PS_DPW * 2;  // hs: 4 DPW doubles per slab
        for (size_t i = wtid; i < ncplx; i += NTT) xs0[i] = mk(0, 0);
    }
    // roles
    const int part = tid & 3;
    const int pair = tid >> 2;                        // M / partial phases: (pair, quarter)
    const bool pvalid = half == 0 && pair < npairs;   // (npairs <= 64; the M phase runs in the first slab's threads)
    const int e = pvalid ? pair / C : 0, c = pvalid ? pair % C : 0;
    // ---- loaders: operands of bin kb (the G_kb slab, |H_kb|; M_{kb-1} by all threads) from memory into registers ...
    // The loads are unconditional at clamped / padded addresses (a branch or a select around them makes the wave wait on
    // the spot); what lies beyond C or the bin range is finite or never stored, directions beyond D repeat D-1.
    // RG rows of the slab are CH chunks of 2 DPW consecutive elements.
    const int lt = loader ? tid : 0;
    cplx gReg[NG * CH];
    double hReg = 0.0;   // |H|: 2 DPW values, loader thread lt takes value lt
    int goff[CH], xoff[CH], gcc[CH];
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        const int g = lt + PS_NL * j, cc = g / PS_DPW, dd = g % PS_DPW;
        const int64_t dg = d0 + dd < a.D ? d0 + dd : a.D - 1;
        goff[j] = (int)(cc * a.ldD + dg);
        xoff[j] = cc * XLD + dd;
        gcc[j] = cc;
    }
    const int hoff = (int)((int64_t)((lt / PS_DPW) & 1) * na * a.ldH + (d0 + lt % PS_DPW < a.D ? d0 + lt % PS_DPW : a.D - 1));
    // (two halves: the first is issued while the waves are in the M phase and the second while they are in the partial
    // phase, so that the issue time -- 26 KB per CU through a 64 B/clk address path -- never holds up a barrier of the chain)
    auto fetch_g = [&](int kb, int half, cplx (&gL)[NG * CH], double& hL) __attribute__((always_inline)) {
        const int kbg = kb < P ? kb : P - 1;
        const cplx* X = a.G + (int64_t)kbg * a.g_stride;
        constexpr int H0 = (NG * CH) / 2;
        if (half == 0) {
#pragma unroll
            for (int i = 0; i < H0; ++i) {
                const int r = i / CH, j = i % CH;
                gL[i] = ldc(X + goff[j] + r * RG * a.ldD);
            }
        } else {
#pragma unroll
            for (int i = H0; i < NG * CH; ++i) {
                const int r = i / CH, j = i % CH;
                gL[i] = ldc(X + goff[j] + r * RG * a.ldD);
            }
            hL = a.Habs[(int64_t)(kbg - a.kabs0) * a.ldH + hoff];
        }
    };
    auto stage_g = [&](int kb, const cplx (&gL)[NG * CH], double hL) __attribute__((always_inline)) {
        double* hs = hs_all + (size_t)(kb & 1) * 2 * PS_DPW;
#pragma unroll
        for (int i = 0; i < NG * CH; ++i) {
            const int r = i / CH, j = i % CH;
            if (RG * r + gcc[j] < C) xs[xoff[j] + r * RG * XLD] = gL[i];
        }
        hs[lt] = hL;
    };
    // M (C x C) is fetched by the threads of the M phase right after they used the previous one
    cplx mReg[NLM];
    auto fetch_m = [&](int kb, cplx (&mL)[NLM]) __attribute__((always_inline)) {
        const int kbm = kb - 1 > a.kfirst ? kb - 1 : a.kfirst;
        const cplx* M = a.Mw + (int64_t)kbm * C * C;
#pragma unroll
        for (int i = 0; i < NLM; ++i) mL[i] = ldc(M + tid + 256 * i);   // (Mw is padded by 1024 elements)
    };
    auto stage_m = [&](const cplx (&mL)[NLM]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NLM; ++i) {
            const int f = tid + 256 * i;
            if (f < C * C) ms[(f / C) * PS_MLD + f % C] = mL[i];
        }
    };
    __syncthreads();  // LDS is zeroed
    if (loader) {
        fetch_g(a.kfirst, 0, gReg, hReg);
        fetch_g(a.kfirst, 1, gReg, hReg);
        stage_g(a.kfirst, gReg, hReg);
        if (fm == 3) { fetch_g(a.kfirst + 1, 0, gReg, hReg); fetch_g(a.kfirst + 1, 1, gReg, hReg); }
    }
    if (half == 0) {
        fetch_m(a.kfirst, mReg);
        stage_m(mReg);
        if (fm == 3) fetch_m(a.kfirst + 1, mReg);
    }
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
    // Four barriers per bin (B1: vt complete, B2: Wp complete, B3: ts complete, B4: the slab buffer is free); every branch
    // around a barrier is workgroup-uniform.
    for (int kb = a.kfirst; kb <= P; ++kb) {
        const bool first = (kb == a.kfirst);
        const bool last = (kb == P);  // only W(P-1,:) is left to form
        const bool nyq = (kb == P - 1);
        // ================= communication wave: the totals of bin kb-1 into vt =================
        if (comm) {
            if (lane == 0) PSTAMP(0);
            if (lane == 0 && a.timing && member == 1 && kb < P) a.timing[(int64_t)kb * 16 + 10] = (long long)clock64();   // shader clock: MHz under load
            if (first) {
                if (lane < npairs) vt[(lane / C) * PS_CMAX + lane % C] = a.W[((int64_t)(lane / C) * P + (kb - 1)) * C + lane % C];
            } else {
                const unsigned tag = (unsigned)(kb - 1);
                const int slot = (kb - 1) & 1;
                bool alive = true;
                unsigned spins1 = 0;
                // hop 1 (reduce-scatter): the pairs this workgroup owns.  Twin workgroups own the pairs of both their slabs
                // (pair q belongs to slab q % nProd): lanes 0-31 collect those of slab 2 member, lanes 32-63 those of slab
                // 2 member + 1, two pairs each -- four pairs in the one pass the single-slab form needs for two.
                if constexpr (NH == 2) {
                    const int grp = lane >> 5, w = lane & 31;
                    for (int qb = slab + grp; qb - grp < npairs && alive; qb += 2 * nProd) {
                        const int q = qb, q2 = q + nProd;
                        const bool act = w < nProd && q < npairs, twoq = act && q2 < npairs;
                        const u64* src = part_ll + (((size_t)slot * npairs + (act ? q : 0)) * nProd + (act ? w : 0)) * 4;
                        const u64* src2 = twoq ? src + (size_t)nProd * nProd * 4 : src;
                        u64 w0 = 0, w1 = 0, w2 = 0, w3 = 0, u0 = 0, u1 = 0, u2 = 0, u3 = 0;
                        alive = ll_wait([&] {
                            w0 = ll_load(src); w1 = ll_load(src + 2); w2 = ll_load(src + 1); w3 = ll_load(src + 3);
                            u0 = ll_load(src2); u1 = ll_load(src2 + 2); u2 = ll_load(src2 + 1); u3 = ll_load(src2 + 3);
                            if (!act) return true;
                            bool ok = ll_ok(w0, tag) && ll_ok(w1, tag) && ll_ok(w2, tag) && ll_ok(w3, tag);
                            if (twoq) ok = ok && ll_ok(u0, tag) && ll_ok(u1, tag) && ll_ok(u2, tag) && ll_ok(u3, tag);
                            return ok;
                        }, a.abort_flag, a.wait_ticks, &spins1, (a.timing && member == 1 && lane == 0 && qb == slab) ? &a.timing[(int64_t)kb * 16 + 6] : nullptr);
                        const double re = group_sum<32>(act ? ll_value(w0, w1) : 0.0);
                        const double im = group_sum<32>(act ? ll_value(w2, w3) : 0.0);
                        const double re2 = group_sum<32>(twoq ? ll_value(u0, u1) : 0.0);
                        const double im2 = group_sum<32>(twoq ? ll_value(u2, u3) : 0.0);
                        if (w == 0 && q < npairs) {
                            u64* dst = tot_ll + (size_t)slot * 2 * nd2 + 2 * q;
                            ll_store(dst, dst + nd2, re, tag, local);
                            ll_store(dst + 1, dst + nd2 + 1, im, tag, local);
                            if (q2 < npairs) {
                                ll_store(dst + 2 * nProd, dst + 2 * nProd + nd2, re2, tag, local);
                                ll_store(dst + 2 * nProd + 1, dst + 2 * nProd + nd2 + 1, im2, tag, local);
                            }
                        }
                    }
                } else
                for (int q = member; q < npairs && alive; q += 2 * nWG) {
                    const int q2 = q + nWG;
                    const bool twoq = q2 < npairs;
                    const u64* src = part_ll + (((size_t)slot * npairs + q) * nWG + lane) * 4;   // lane = producing workgroup
                    const u64* src2 = src + (size_t)nWG * nWG * 4;
                    u64 w0 = 0, w1 = 0, w2 = 0, w3 = 0, u0 = 0, u1 = 0, u2 = 0, u3 = 0;
                    alive = ll_wait([&] {
                        if (lane >= nWG) return true;
                        w0 = ll_load(src); w1 = ll_load(src + 2); w2 = ll_load(src + 1); w3 = ll_load(src + 3);
                        bool ok = true;
                        if (twoq) {
                            u0 = ll_load(src2); u1 = ll_load(src2 + 2); u2 = ll_load(src2 + 1); u3 = ll_load(src2 + 3);
                            ok = ll_ok(u0, tag) && ll_ok(u1, tag) && ll_ok(u2, tag) && ll_ok(u3, tag);
                        }
                        return ok && ll_ok(w0, tag) && ll_ok(w1, tag) && ll_ok(w2, tag) && ll_ok(w3, tag);
                    }, a.abort_flag, a.wait_ticks, &spins1, (a.timing && member == 1 && lane == 0 && q == member) ? &a.timing[(int64_t)kb * 16 + 6] : nullptr);
                    const double re = wave_sum(lane < nWG ? ll_value(w0, w1) : 0.0);
                    const double im = wave_sum(lane < nWG ? ll_value(w2, w3) : 0.0);
                    double re2 = 0.0, im2 = 0.0;
                    if (twoq) {
                        re2 = wave_sum(lane < nWG ? ll_value(u0, u1) : 0.0);
                        im2 = wave_sum(lane < nWG ? ll_value(u2, u3) : 0.0);
                    }
                    if (lane == 0) {
                        u64* dst = tot_ll + (size_t)slot * 2 * nd2 + 2 * q;
                        ll_store(dst, dst + nd2, re, tag, local);
                        ll_store(dst + 1, dst + nd2 + 1, im, tag, local);
                        if (twoq) {
                            ll_store(dst + 2 * nWG, dst + 2 * nWG + nd2, re2, tag, local);
                            ll_store(dst + 2 * nWG + 1, dst + 2 * nWG + nd2 + 1, im2, tag, local);
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
        const bool prev_ok = first ? true : (s_ok[kb - 1] != 0);
        const bool cur_ok = last ? true : (s_ok[kb] != 0);
        const double* hs = hs_all + (size_t)(kb & 1) * 2 * PS_DPW;
        __syncthreads();  // B1: vt is complete (and this bin's operands are staged)
        if (s_abort) break;
        // Where the next bin's operands are requested (a.fetch_mode).  A CU's vector memory pipeline returns in order across its
        // waves: loads that miss to HBM (1.2-1.5 us) delay every later poll of the communication wave behind them, so they must
        // have drained when the exchange starts -- requesting them after B4 (mode 3, a whole bin period of lead and no issue
        // time on the chain) costs +0.8 us per bin in hop 1 (measured).  Issued here (and after B3 / B2) they drain during the
        // compute phases, at the price of their issue time (~0.15 us per half slab) on the chain.
        if (loader && fm != 3 && fm != 4) { fetch_g(kb + 1, 0, gReg, hReg); if (fm == 2) fetch_g(kb + 1, 1, gReg, hReg); }
        // ---- W(kb-1,:) = v_total conj(M_{kb-1})  (identity for the first swept bin and after an ill-conditioned bin)
        if (pvalid) {
            cplx acc = mk(0, 0);
            if (first || !prev_ok) {
                if (part == 0) acc = vt[e * PS_CMAX + c];
            } else {
#pragma unroll
                for (int i = 0; i < NI; ++i)  // vt and ms are 0 beyond C
                    cfma(acc, vt[e * PS_CMAX + part + 4 * i], conj(ms[(part + 4 * i) * PS_MLD + c]));
            }
            acc = group_sum<4>(acc);
            if (part == 0) {
                Wp[e * PS_CMAX + c] = acc;
                if (member == 0 && !first) a.W[((int64_t)e * P + (kb - 1)) * C + c] = acc;
            }
        }
        if (last) break;
        if (fm != 3 && half == 0) fetch_m(kb + 1, mReg);
        __syncthreads();  // B2: Wp is complete
        if (loader && fm == 1) fetch_g(kb + 1, 1, gReg, hReg);
        // ---- p = W(kb-1,:) pwGrid ;  t = |H| p/|p|
        // thread = (direction pair (dA, dA + DPW/2), channel quarter): a W value read from LDS feeds two directions
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
        if (loader && fm == 0) fetch_g(kb + 1, 1, gReg, hReg);
        if (loader && fm == 4) { fetch_g(kb + 1, 0, gReg, hReg); fetch_g(kb + 1, 1, gReg, hReg); }   // (experiment: everything after B3)
        // ---- this slab's partial v = t conj(G) (or t Y_reg_inv for an ill-conditioned bin), published as granules
        if (wtid == 0) PSTAMP(4);
        // thread = (channel pair cp, 16 direction slices): every t and every G element it reads from LDS feeds two
        // complex FMAs (2 ears x 2 channels), half the LDS traffic of one (ear, channel) pair per thread
        {
            const int cp = tid >> 4, ep = tid & 15;
            const int c0 = 2 * cp, c1 = c0 + 1;
            if (c0 < C) {
                cplx v00 = mk(0, 0), v01 = mk(0, 0), v10 = mk(0, 0), v11 = mk(0, 0);  // v[ear][channel]
                if (cur_ok) {
                    const cplx* x0 = xs + c0 * XLD, *x1 = xs + (c1 < C ? c1 : c0) * XLD;
#pragma unroll PUNR
                    for (int j = 0; j < PS_DPW / 16; ++j) {
                        const int dd = ep + 16 * j;
                        const cplx t0 = ts[0][dd], t1 = ts[1][dd], g0 = conj(x0[dd]), g1 = conj(x1[dd]);
                        cfma(v00, t0, g0); cfma(v01, t0, g1); cfma(v10, t1, g0); cfma(v11, t1, g1);
                    }
                } else {
                    const cplx* Y0 = a.Yri + (int64_t)kb * a.g_stride + (int64_t)c0 * a.ldD;
                    const cplx* Y1 = a.Yri + (int64_t)kb * a.g_stride + (int64_t)(c1 < C ? c1 : c0) * a.ldD;
                    for (int dd = ep; dd < PS_DPW; dd += 16)
                        if (d0 + dd < a.D) {
                            const cplx t0 = ts[0][dd], t1 = ts[1][dd], g0 = Y0[d0 + dd], g1 = Y1[d0 + dd];
                            cfma(v00, t0, g0); cfma(v01, t0, g1); cfma(v10, t1, g0); cfma(v11, t1, g1);
                        }
                }
                v00 = group_sum<16>(v00); v01 = group_sum<16>(v01); v10 = group_sum<16>(v10); v11 = group_sum<16>(v11);
                // all 16 lanes hold the four sums: lane ep stores word ep & 3 of the pair (ear ep >> 3, channel c0 + ((ep >> 2) & 1))
                const int ee = ep >> 3, ch = (ep >> 2) & 1, wi = ep & 3;
                const cplx acc = ee ? (ch ? v11 : v10) : (ch ? v01 : v00);
                const int cc = c0 + ch;
                if (cc < C) {
                    const u64 bits = (u64)__double_as_longlong((wi & 1) ? acc.y : acc.x);
                    const u64 word = ((u64)(unsigned)kb << 32) | ((wi & 2) ? (bits >> 32) : (bits & 0xffffffffull));
                    u64* dst = part_ll + (((size_t)(kb & 1) * npairs + (ee * C + cc)) * nProd + slab) * 4 + wi;
                    ll_put(dst, word, local);
                }
            }
        }
        if (wtid == 0) PSTAMP(5);
        // the slab buffer is free once every wave has left the partial phase (M was last read before B2): refill both while
        // everybody waits for the exchange
        __syncthreads();  // B4
        // Stage the operands of bin kb + 1 (requested one bin period ago: they have arrived) and request those of bin kb + 2 into
        // the same registers.  This is the exchange wait of the compute waves: nothing here delays the chain, and the
        // communication wave (wave 3, no G loads of its own) polls meanwhile.
        if (loader) {
            stage_g(kb + 1, gReg, hReg);
            if (fm == 3) { fetch_g(kb + 2, 0, gReg, hReg); fetch_g(kb + 2, 1, gReg, hReg); }
        }
        if (half == 0) {
            stage_m(mReg);
            if (fm == 3) fetch_m(kb + 2, mReg);
        }
    }
#undef PSTAMP
}

}  // namespace

int persist_sweep_dpw(int D) { return D <= 32 * 64 ? 64 : 96; }
int persist_sweep_nwg(int D) { return (int)ceil_div(D, persist_sweep_dpw(D)); }
bool persist_sweep_supported(int D, int C) { return C <= PS_CMAX && persist_sweep_nwg(D) <= 32; }
// (sized for the twin form as well: 2 ceil(nWG / 2) producers)
size_t persist_sweep_ll_bytes(int D, int C) {
    const size_t nprod = 2 * (size_t)ceil_div(persist_sweep_nwg(D), 2);
    return sizeof(u64) * ((size_t)2 * nprod * 8 * C + (size_t)2 * 8 * C + 64);
}

// EMAGLS_SWEEP_TWIN=0: launches of 9-16 designs as two independent 256-thread workgroups per CU (the earlier form)
// (read at every launch: a test switches forms inside one process)
static bool sweep_twin_enabled() { const char* e = getenv("EMAGLS_SWEEP_TWIN"); return !(e && e[0] == '0'); }
static void persist_set_attributes() {
    static PerDeviceOnce attr_once;   // (function attributes are per device)
    if (attr_once.first()) {
#define EMAGLS_PS_ATTR(D, N) do { \
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(sweep_persist_kernel<D, N, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); \
        HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(sweep_persist_kernel<D, N, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024)); } while (0)
        EMAGLS_PS_ATTR(64, 2); EMAGLS_PS_ATTR(64, 4); EMAGLS_PS_ATTR(64, 7); EMAGLS_PS_ATTR(64, 8);
        EMAGLS_PS_ATTR(96, 2); EMAGLS_PS_ATTR(96, 4); EMAGLS_PS_ATTR(96, 7); EMAGLS_PS_ATTR(96, 8);
#undef EMAGLS_PS_ATTR
    }
}
static const void* persist_kernel_ptr(int dpw, int ni, int nh) {
#define EMAGLS_PS_PTR(D, N) (nh == 2 ? reinterpret_cast<const void*>(sweep_persist_kernel<D, N, 2>) : reinterpret_cast<const void*>(sweep_persist_kernel<D, N, 1>))
    if (dpw == 64) return ni == 2 ? EMAGLS_PS_PTR(64, 2) : ni == 4 ? EMAGLS_PS_PTR(64, 4) : ni == 7 ? EMAGLS_PS_PTR(64, 7) : EMAGLS_PS_PTR(64, 8);
    return ni == 2 ? EMAGLS_PS_PTR(96, 2) : ni == 4 ? EMAGLS_PS_PTR(96, 4) : ni == 7 ? EMAGLS_PS_PTR(96, 7) : EMAGLS_PS_PTR(96, 8);
#undef EMAGLS_PS_PTR
}
static int persist_ni(int C) { return C <= 8 ? 2 : (C <= 16 ? 4 : (C <= 28 ? 7 : 8)); }
static size_t persist_dyn_bytes(int dpw, int ni, int nh) {
    const size_t rows = 4 * (size_t)ni;   // (only the swept channel quarters take LDS: 28 rows for the 25 channels of order 4)
    return sizeof(cplx) * ((size_t)nh * rows * (dpw + 4) + rows * PS_MLD + (size_t)nh * 2 * dpw);
}
// CUs the resident kernels may count on: the device's, or EMAGLS_CU_BUDGET (a CU-masked queue, a shared GPU; read at every call)
int sweep_cu_budget() {
    int dev = 0, n = 0;
    HIP_CHECK(hipGetDevice(&dev));
    HIP_CHECK(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
    if (const char* e = getenv("EMAGLS_CU_BUDGET")) { const int b = atoi(e); if (b > 0 && b < n) n = b; }
    return n;
}
// Can `ndesigns` designs' workgroups of the resident sweep all be on the device at once?  The runtime's occupancy figure for the
// kernel variant (registers, LDS, waves per CU) times the CUs of one XCD against the workgroups a design pair places there --
// decided BEFORE the launch: a sweep whose workgroups cannot all become resident would wait for its peers until the time-out.
bool persist_sweep_fits(int D, int C, int ndesigns) {
    if (!persist_sweep_supported(D, C) || ndesigns < 1 || ndesigns > SWEEP_MULTI_MAX) return false;
    persist_set_attributes();
    const bool twin = ndesigns > 8 && sweep_twin_enabled();
    const int nh = twin ? 2 : 1, dpw = persist_sweep_dpw(D), ni = persist_ni(C);
    const int nWG = (int)ceil_div(persist_sweep_nwg(D), nh);
    int occ = 0;
    HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, persist_kernel_ptr(dpw, ni, nh), PS_NT * nh, persist_dyn_bytes(dpw, ni, nh)));
    const int per_xcd = (ndesigns > 8 ? 2 : 1) * nWG;
    return per_xcd <= occ * (sweep_cu_budget() / 8);
}

void launch_sweep_persist(const HalfSweepMulti& m, hipStream_t st) {
    const HalfSweepArgs& a = m.a[0];
    const int nSlab = persist_sweep_nwg(a.D);
    if (!persist_sweep_supported(a.D, a.C) || m.n > SWEEP_MULTI_MAX || a.P > PS_PMAX) throw Error(2, "persistent sweep: shape not supported");
    const bool twin = m.n > 8 && sweep_twin_enabled();
    const int nh = twin ? 2 : 1;
    const int nWG = (int)ceil_div(nSlab, nh);
    const unsigned nblocks = 8u * (unsigned)nWG * (m.n > 8 ? 2u : 1u);
    const int dpw = persist_sweep_dpw(a.D);
    const int ni = persist_ni(a.C);
    const size_t dyn = persist_dyn_bytes(dpw, ni, nh);
    persist_set_attributes();
#define EMAGLS_PS_GO(D, N) do { if (twin) sweep_persist_kernel<D, N, 2><<<dim3(nblocks), 2 * PS_NT, dyn, st>>>(m, nWG); \
                                else sweep_persist_kernel<D, N, 1><<<dim3(nblocks), PS_NT, dyn, st>>>(m, nWG); } while (0)
    if (dpw == 64) { if (ni == 2) EMAGLS_PS_GO(64, 2); else if (ni == 4) EMAGLS_PS_GO(64, 4); else if (ni == 7) EMAGLS_PS_GO(64, 7); else EMAGLS_PS_GO(64, 8); }
    else { if (ni == 2) EMAGLS_PS_GO(96, 2); else if (ni == 4) EMAGLS_PS_GO(96, 4); else if (ni == 7) EMAGLS_PS_GO(96, 7); else EMAGLS_PS_GO(96, 8); }
#undef EMAGLS_PS_GO
    KERNEL_CHECK();
}

}  // namespace emagls
