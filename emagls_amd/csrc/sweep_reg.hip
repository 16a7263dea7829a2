// Resident MagLS phase sweep, REGISTER-RESIDENT form: the waves that run the chain also evaluate the operand, and nothing of a
// bin's slab of pwGrid_k.' ever lies in LDS.
//
// Reference (lib/getEMagLsFilters.m:87-103): per bin  p = W(k-1,:) pwGrid_k,  t = |H| p/|p|,  W(k,:) = t Y_reg_inv_k -- a
// recurrence over the bins of one design.  sweep_synth.hip (round 4) evaluates pwGrid_k.' inside the launch from
//     g_k[d][j] = sum_m bsc[k][m] T_m(x_dj),    x_dj = cos(angle between HRIR direction d and microphone j)
// (Legendre addition theorem, Chebyshev basis, antipodal microphones as ONE unit: g(x) = E + O, g(-x) = E - O), but keeps the
// slab of 96 directions x 32 microphones in LDS between producer waves and chain waves: 80 KB and 127 registers x 512 threads
// per workgroup, two workgroups per CU, 29 CUs per design, four barriers and a two-hop exchange per bin; its vector ALUs
// were busy 24 % of the time while the launch held 232 CUs (profiles/r04_pmc.md).
//
// Here a LANE owns ONE direction and all NU units of it: x_du (17 values for the em32) and the bin's E_u, O_u (68 values) live in
// registers.  Per bin a wave
//   * p phase:   p_e[d] = sum_u (wE_e[u] E_u + wO_e[u] O_u)  with  wE = w'[row A] + w'[row B], wO = w'[row A] - w'[row B]  read as
//                LDS broadcasts -- no cross-lane step at all;
//   * t = |H| p/|p| in the lane;
//   * partial:   t_e conj(E_u), t_e conj(O_u) (136 values per lane) summed over the wave's 64 directions by a halving
//                reduction: v_permlane32_swap / v_permlane16_swap exchange half of the values with the lane 32 / 16 away (one
//                instruction per word, no select), two DPP steps inside the rows, an all-reduce inside the quads -- 9 values
//                per lane are left, each total owned by one quad;
//   * the four waves' partials are summed through 4.6 KB of LDS, turned into microphone rows (u = uE +- uO) and published as tagged
//     granules; every workgroup of the design reads ALL partials of the bin (one hop: 11 workgroups x 2 KB at 2702 directions) and
//     sums them in workgroup order;
//   * the SAME waves then evaluate E, O of the next bin (1020 fused operations per lane at 20 orders) while the partials of the
//     other workgroups arrive: the exchange latency hides behind arithmetic of the same wave instead of idling a CU.
// 256 threads and 36 KB of LDS per workgroup, three barriers per bin, ceil(D / 256) workgroups per design (11 instead of 29), two
// workgroups per CU: up to 40 designs of 2702 directions per launch (5 per XCD) instead of 16.
//
// Everything around the chain is sweep_synth.hip's: synth_coeff_kernel (bsc), synth_mt_kernel (Mt = Pm^T M Pm, the chain's
// start value), synth_rows_kernel (the filters' rows from the stored totals), the row order of the microphones (smap).
#include "kernels.hpp"
#include "persist_common.hpp"
#include "synth_common.hpp"

#include <algorithm>
#include <map>
#include <mutex>

namespace emagls {

namespace {

constexpr int RG_MLD = 36;      // row stride of M~ in LDS (16 dwords mod 64)

// ---- halving steps of the wave reduction.  Each takes two values per lane and returns ONE: half of the lanes get the sum of `a`
// over the lane pair, the other half the sum of `b`.
// lanes 0-31: a[l] + a[l + 32]; lanes 32-63: b[l - 32] + b[l]
__device__ __forceinline__ double halve32(double a, double b) {
    const unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
    const unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
    const auto lo = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);   // lanes 32-63 of the first <-> lanes 0-31 of the second
    const auto hi = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
// even rows (of 16 lanes): a[l] + a[l + 16]; odd rows: b[l - 16] + b[l]
__device__ __forceinline__ double halve16(double a, double b) {
    const unsigned alo = (unsigned)__double2loint(a), ahi = (unsigned)__double2hiint(a);
    const unsigned blo = (unsigned)__double2loint(b), bhi = (unsigned)__double2hiint(b);
    const auto lo = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);   // odd rows of the first <-> even rows of the second
    const auto hi = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    return __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
}
// inside a row: lanes with keep_a get a[l] + a[perm(l)], the others b[l] + b[perm(l)]; perm (a DPP control, its own inverse) maps the
// keep_a lanes onto the others
template <int CTRL> __device__ __forceinline__ double halve_row(double a, double b, bool keep_a) {
    const double x = keep_a ? a : b, y = keep_a ? b : a;
    return x + dpp_d<CTRL>(y);
}

// E_u, O_u of GS unit slots accumulated in place (synth_group's schedule: two terms per pass, the coefficients of the next pass
// requested before this pass's arithmetic); xcol: the lane's column of 2 cos values, one row of NT per slot
template <int U0, int GS, int NUL, int NT>
__device__ __forceinline__ void synth_units(const double* xcol, cplx (&E)[NUL], cplx (&O)[NUL], const cplx* bs, int nord_pad) {
    double x2[GS], pa[GS], pb[GS];
#pragma unroll
    for (int i = 0; i < GS; ++i) x2[i] = xcol[(U0 + i) * NT];
#pragma unroll
    for (int i = 0; i < GS; ++i) { pa[i] = 1.0; pb[i] = 0.5 * x2[i]; E[U0 + i] = mk(0.0, 0.0); O[U0 + i] = mk(0.0, 0.0); }
    const unsigned bs0 = lds_addr(bs);
    d2_t b0, b1, c0, c1;
    auto request = [&](int n, d2_t& q0, d2_t& q1) __attribute__((always_inline)) {
        const int nn = n < nord_pad ? n : nord_pad - 2;   // (the last pass re-reads its own coefficients)
        lds_read16_async(q0, bs0 + 16 * nn, pa[0]); lds_read16_async(q1, bs0 + 16 * nn + 16, pa[0]);
    };
    auto pass = [&](const d2_t& q0, const d2_t& q1) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < GS; ++i) {
            E[U0 + i].x = fma(q0.x, pa[i], E[U0 + i].x); E[U0 + i].y = fma(q0.y, pa[i], E[U0 + i].y);
            O[U0 + i].x = fma(q1.x, pb[i], O[U0 + i].x); O[U0 + i].y = fma(q1.y, pb[i], O[U0 + i].y);
            pa[i] = fma(x2[i], pb[i], -pa[i]);          // T_{m+2}
            pb[i] = fma(x2[i], pa[i], -pb[i]);          // T_{m+3}
        }
    };
    request(0, b0, b1);
    lds_wait2(b0, b1, pa[0]);
    for (int n = 0; n < nord_pad; n += 4) {
        request(n + 2, c0, c1);
        pass(b0, b1);
        lds_wait2(c0, c1, pb[GS - 1]);
        if (n + 2 >= nord_pad) break;
        request(n + 4, b0, b1);
        pass(c0, c1);
        lds_wait2(b0, b1, pb[GS - 1]);
    }
}

// (a value the compiler must recompute from here on in every iteration: thread-derived indices and addresses of the loop's sections
// otherwise stay in registers for the whole loop next to the operand)
__device__ __forceinline__ int launder(int v) { asm volatile("" : "+v"(v)); return v; }

// pointers that arrive through the argument block in memory: the compiler cannot see that they are global and would emit flat_*
// accesses (which count on the LDS counter as well)
#define RG_GLOBAL(T, p) ((T __attribute__((address_space(1)))*)(p))
typedef const double __attribute__((address_space(1)))* gcd_t;
typedef const cplx __attribute__((address_space(1)))* gcc_t;
typedef u64 __attribute__((address_space(1)))* gu64_t;
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ cplx ldg(gcc_t p) { return mk(p->x, p->y); }
// granules in global memory (the helpers of persist_common.hpp take generic pointers)
__device__ __forceinline__ u64 gll_load(gu64_t p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// (`local`: workgroup-scope stores for traffic BETWEEN workgroups of one XCD -- formally a data race; it relies on the per-CU vector cache
// of gfx942 / gfx950 being write-through, so that the store reaches the XCD's L2 at once.  Any other architecture: EMAGLS_PERSIST_GLOBAL=1.)
__device__ __forceinline__ void gll_put(gu64_t dst, u64 word, bool local) {
    if (local) __hip_atomic_store(dst, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(dst, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// both granules of a double in ONE request that misses the CU's own cache (each 8-byte granule carries its own tag, so the two
// halves need not be read atomically together); the value is defined by rg_wait_loads
__device__ __forceinline__ void gll_load16_async(u32x4_t& out, gu64_t p) { asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(out) : "v"(p) : "memory"); }

constexpr int RG_POLL = 12;   // partials a thread requests at once (22 workgroups at 2702 directions and 4 waves: 11 per thread, one round trip)
__device__ __forceinline__ void rg_wait_loads(u32x4_t (&g)[RG_POLL]) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5]), "+v"(g[6]), "+v"(g[7]), "+v"(g[8]), "+v"(g[9]),
                 "+v"(g[10]), "+v"(g[11]) :: "memory");
}
constexpr int RG_NEX = 4 * PS_CMAX;   // doubles a workgroup publishes per bin: [ear][row (32, zero beyond the microphones)][re / im]
constexpr int RG_SVC = 256;           // threads that serve the workgroup: exchange, M phase, M~ staging (waves 0-3)

// NUL: unit slots per lane (a direction's units are split between the two waves of a pair: 2 NUL >= units); NW: waves per workgroup
template <int NUL, int NW>
__global__ void __launch_bounds__(64 * NW) sweep_reg_kernel(const HalfSweepArgs* __restrict__ args, int n, int nWG, int spread, int prio) {
    constexpr int NT = 64 * NW, DPW = 32 * NW, NB = NW / 2;   // threads, directions and 64-direction blocks of a workgroup
    constexpr int NU2 = 2 * NUL;            // unit slots of a direction
    constexpr int NCH = (NUL + 1) / 2;      // chunks of two slots in the wave reduction
    constexpr int NVW = 16 * NCH;           // values of a wave's partial: [slot][E / O][ear][re / im], padded to whole chunks
    constexpr int GA = (NUL + 1) / 2, GB = NUL - GA;   // slot groups of the synthesis (9: 5, 4)
    static_assert(GB >= 1 && NW % 2 == 0 && NT >= RG_SVC, "layout");
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    cplx* ring = reinterpret_cast<cplx*>(dyn);                       // [2][SY_NORD]  Chebyshev coefficients of two bins
    cplx* ms = ring + 2 * SY_NORD;                                   // [32][MLD]     M~_{kb-1}
    cplx* WE = ms + PS_CMAX * RG_MLD;                                // [2][NU2]      w'[row A] + w'[row B]
    cplx* WO = WE + 2 * NU2;                                         // [2][NU2]      w'[row A] - w'[row B]
    double* vt = reinterpret_cast<double*>(WO + 2 * NU2);            // [2][2][32][2] totals of the previous bin in microphone rows: the sums over the even and over the odd workgroups
    double* wpart = vt + 2 * RG_NEX;                                 // [NW][NVW]     the waves' partials
    double* pp = wpart + NW * NVW;                                   // [NW][4][64]   a wave's share of p (its half of the units): [ear][re / im][lane]
    double* xs = pp + NW * 256;                                      // [NUL][NT]     2 cos(direction, unit): a lane reads its own column
    __shared__ int s_abort, s_local;

    // block -> (XCD, slot) -> (design, member): design j lives on XCD j % 8, the (j / 8)-th design there.  (One design's workgroups
    // after the other's: taking the slots in turn was tried -- with workgroups that share CUs unevenly BOTH designs then run at the
    // pace of their slowest workgroups, 9.9 us per bin each instead of 7.2 / 10.4.)
    // spread != 0: a design's workgroups are consecutive blocks, i.e. dealt round over ALL XCDs -- for design counts that leave
    // CUs idle when every design must stay inside one XCD (20 designs: three per XCD on four of them, ten waves per CU on 27 of 32
    // CUs, against eight waves on 220 of 256); the granules then travel through memory (the check below finds the XCDs differ)
    int design, member;
    if (spread) {
        design = (int)blockIdx.x / nWG;
        member = (int)blockIdx.x - design * nWG;
    } else {
        const int xcd = blockIdx.x & 7, rest = blockIdx.x >> 3;
        const int sub = rest / nWG;
        member = rest - sub * nWG;
        design = xcd + 8 * sub;
    }
    if (design >= n) return;
    // the chain's waves ahead of whatever shares their SIMDs (the stages of other chunks; a lone chunk's own orthonormal route, which runs
    // next to its sweep since round 6): priority 3 of 0..3.  EMAGLS_REG_PRIO=0 (argument `prio`): the default priority.
    if (prio) __builtin_amdgcn_s_setprio(3);
    const HalfSweepArgs& a = args[design];
    const int tid = threadIdx.x;
    const int C = a.C, P = a.P, D = a.D, kfirst = a.kfirst, kabs0 = a.kabs0;   // C: microphones (the chain's channels)
    const int64_t ldH = a.ldH;
    const int nord_pad = a.nord_pad;
    const gcd_t Habs = RG_GLOBAL(const double, a.Habs);
    const gcc_t Mw = RG_GLOBAL(const cplx, a.Mw);
    const gcc_t bsc = RG_GLOBAL(const cplx, a.bsc);
    const gcd_t Winit = RG_GLOBAL(const double, a.Winit);
    cplx __attribute__((address_space(1)))* const Uout = RG_GLOBAL(cplx, a.U);
    long long* const timing = a.timing;
    int* const abort_flag = a.abort_flag;
    const long long wait_ticks = a.wait_ticks;
    const int* smap = a.smap;
    const int npr = smap[32], nsg = smap[33], nun = npr + nsg;
    const gu64_t part_ll = RG_GLOBAL(u64, a.ll);                     // [2][nWG][RG_NEX][2]
    const gu64_t xcc_ll = part_ll + (size_t)2 * nWG * RG_NEX * 2;    // [nWG]
    if (tid == 0) { s_abort = 0; s_local = 0; }
    const int d0 = member * DPW;
    // thread -> direction of the workgroup: wave pair q = wave / 2 takes the directions 64 q .. 64 q + 63 (lane = direction), wave
    // parity h the unit slots h NUL .. h NUL + NUL - 1
    auto dir_of = [&](int t) __attribute__((always_inline)) { return d0 + 64 * (t >> 7) + (t & 63); };

    // ---- twice the cosine of the angle between the lane's direction and the (first) microphone of each of its units, kept in LDS (a
    // lane reads back only its own column, a few values at a time)
    {
        const size_t ncplx = (size_t)2 * SY_NORD + PS_CMAX * RG_MLD + 4 * NU2;
        for (size_t i = tid; i < ncplx; i += NT) ring[i] = mk(0, 0);
        for (int i = tid; i < 2 * RG_NEX + NW * NVW + NW * 256; i += NT) vt[i] = 0.0;
        const int dgi = dir_of(tid) < D ? dir_of(tid) : D - 1;
        double sd, cd;
        sincos(a.dir_zen[dgi], &sd, &cd);
        const double daz = a.dir_azi[dgi];
#pragma unroll 1
        for (int i = 0; i < NUL; ++i) {
            const int u = ((tid >> 6) & 1) * NUL + i;
            double v = 0.0;
            if (u < nun) {
                const int jm = smap[u < npr ? 2 * u : 2 * npr + (u - npr)];
                double sm, cm;
                sincos(a.mic_zen ? a.mic_zen[jm] : 1.5707963267948966, &sm, &cm);
                v = fma(sd * sm, cos(daz - a.mic_azi[jm]), cd * cm);
                v = 2.0 * fmin(1.0, fmax(-1.0, v));   // (2x: the factor of the Chebyshev recurrence)
            }
            xs[i * NT + tid] = v;
        }
    }
    __syncthreads();   // LDS is initialised

    // ---- per-bin operands from memory
    if (tid < nord_pad) {   // (lanes of wave 0)
        ring[(kfirst & 1) * SY_NORD + tid] = ldg(bsc + (int64_t)kfirst * nord_pad + tid);
        const int k1 = kfirst + 1 < P ? kfirst + 1 : P - 1;
        ring[((kfirst + 1) & 1) * SY_NORD + tid] = ldg(bsc + (int64_t)k1 * nord_pad + tid);
    }
    // M~ (C x C in memory, bin kbm at a.Mw + kbm C C: the factor stage stores bin kb at slot kb - 1 and the pointer is shifted):
    // service thread = (row, four columns).  (Reads beyond a row's C columns stay inside the buffer -- it is padded by 1024 elements --
    // and are dropped when the row is staged.)
    constexpr int NLM = 4;
    cplx mReg[NLM];
    auto fetch_m = [&](int kbm, int t) __attribute__((always_inline)) {
        const int r = t >> 3, c0 = (t & 7) * 4;
        const gcc_t M = Mw + (int64_t)kbm * C * C + r * C + c0;
#pragma unroll
        for (int i = 0; i < NLM; ++i) mReg[i] = ldg(M + i);
    };
    auto stage_m = [&](int t) __attribute__((always_inline)) {
        const int r = t >> 3, c0 = (t & 7) * 4;
#pragma unroll
        for (int i = 0; i < NLM; ++i) ms[r * RG_MLD + c0 + i] = (r < C && c0 + i < C) ? mReg[i] : mk(0, 0);
    };
    double hCur[2], hNext[2] = {0.0, 0.0};
    auto fetch_h = [&](int kb, double (&h)[2], int t) __attribute__((always_inline)) {
        const int kbg = (kb < P ? kb : P - 1) - kabs0;
        const int dgi = dir_of(t) < D ? dir_of(t) : D - 1;
        h[0] = (Habs + (int64_t)kbg * ldH)[dgi];
        h[1] = (Habs + ((int64_t)(P - kabs0) + kbg) * ldH)[dgi];
    };
    fetch_h(kfirst, hCur, tid);
    if (tid < RG_SVC) { fetch_m(kfirst, tid); stage_m(tid); }

    // ---- do all workgroups of this design share an XCD?  (then the granules stay in its L2)
    if (tid < 64) {
        const int lane = tid;
        const unsigned xcc = read_xcc_id();
        const unsigned tag0 = 0x58434300u;  // 'XCC'
        if (lane == 0) gll_put(xcc_ll + member, ((u64)tag0 << 32) | xcc, false);
        bool alive = true, same = true;
        for (int base = 0; base < nWG && alive; base += 64) {
            u64 w = 0;
            alive = ll_wait([&] {
                if (base + lane >= nWG) return true;
                w = gll_load(xcc_ll + base + lane);
                return ll_ok(w, tag0);
            }, abort_flag, wait_ticks);
            same = same && __builtin_amdgcn_ballot_w64(base + lane < nWG && (unsigned)w != xcc) == 0;
        }
        if (lane == 0) {
            s_local = alive && same && a.force_global == 0;
            if (!alive) s_abort = 1;
            if (timing && member == 1) timing[15] = s_local;
        }
    }
    __syncthreads();   // ring rows, M~, s_local
    const bool local = s_local != 0;
    if (s_abort) return;

    // ---- operand of one bin: E_u, O_u of the lane's direction, the wave's half of the units
    const bool last_slot_used = ((tid >> 6) & 1) * NUL + NUL - 1 < nun || GB < 2;   // (wave-uniform)
    cplx E[NUL], O[NUL];
#pragma unroll
    for (int u = 0; u < NUL; ++u) { E[u] = mk(0, 0); O[u] = mk(0, 0); }

#define RSTAMP(i) do { if (timing && member == 1 && tid == 0 && kb < P) timing[(int64_t)kb * 16 + (i)] = (long long)wall_clock64(); } while (0)
    for (int kb = kfirst; kb <= P; ++kb) {
        const bool first = (kb == kfirst);
        const bool last = (kb == P);   // only the totals of bin P-1 are left to store
        const bool nyq = (kb == P - 1);
        // ---- the operand of this bin, while the other workgroups' partials of the previous bin arrive (the only place the
        // synthesis is inlined: the loop is rotated so that it leads the iteration)
        RSTAMP(7);
        if (!last) {
            const cplx* bs = ring + (kb & 1) * SY_NORD;
            const double* xcol = xs + launder(tid);
            synth_units<0, GA, NUL, NT>(xcol, E, O, bs, nord_pad);
            __builtin_amdgcn_sched_barrier(0);
            // (the em32's 17 units: the second wave of a pair holds 8 -- its last slot stays zero and is skipped here, in the p phase's
            // weights (zero) and in the wave reduction)
            if (last_slot_used) synth_units<GA, GB, NUL, NT>(xcol, E, O, bs, nord_pad);
            else synth_units<GA, GB - 1, NUL, NT>(xcol, E, O, bs, nord_pad);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ================= totals of bin kb-1: every workgroup's partial, summed in workgroup order =================
        RSTAMP(0);
        if (launder(tid) < RG_SVC) {
            // service thread = (half of the workgroups, ear, row, re / im): the partials of the workgroups half, half + 2, ... of one
            // double, requested together (a loop over 22 sources with one request in flight took 3 us per bin)
            const int t = launder(tid);
            const int j = t & (RG_NEX - 1), half = t >> 7;
            if (first) {   // W(kfirst-1,:) Pm from the least-squares bins (synth_mt_kernel / synth_winit_kernel)
                vt[t] = half == 0 ? Winit[j] : 0.0;
            } else {
                const unsigned tag = (unsigned)(kb - 1);
                const gu64_t src = part_ll + ((size_t)((kb - 1) & 1) * nWG * RG_NEX + j) * 2;
                double sum = 0.0;
                const bool alive = ll_wait([&] {
                    bool ok = true;
                    double s = 0.0;
                    for (int base = half; base < nWG; base += 2 * RG_POLL) {
                        u32x4_t g[RG_POLL];
#pragma unroll
                        for (int i = 0; i < RG_POLL; ++i) {
                            const int w = base + 2 * i < nWG ? base + 2 * i : half;   // (beyond the last source: the first one again, not added)
                            gll_load16_async(g[i], src + (size_t)w * RG_NEX * 2);
                        }
                        rg_wait_loads(g);
#pragma unroll
                        for (int i = 0; i < RG_POLL; ++i) {
                            ok = ok && g[i].y == tag && g[i].w == tag;
                            if (base + 2 * i < nWG) s += __hiloint2double((int)g[i].z, (int)g[i].x);
                        }
                    }
                    sum = s;
                    return ok;
                }, abort_flag, wait_ticks);
                vt[t] = sum;
                if (!alive && (t & 63) == 0) s_abort = 1;
            }
        }
        RSTAMP(2);
        __syncthreads();  // B1: vt is complete
        if (s_abort) break;
        // ---- w'(kb-1,:) = u_total conj(M~_{kb-1})  (the start value as it stands), as the p phase's unit weights; the totals themselves
        // go to memory: the filters' rows are formed from them after the launch
        if (launder(tid) < RG_SVC) {
            const int t = launder(tid);
            const int part = t & 3, pair = t >> 2, e = pair >> 5, c = pair & 31;
            const cplx* vc = reinterpret_cast<const cplx*>(vt) + e * PS_CMAX;   // (+ RG_NEX / 2: the odd workgroups' sum)
            cplx acc = mk(0, 0);
            if (first) {
                if (part == 0) acc = vc[c];
            } else {
#pragma unroll
                for (int i = 0; i < PS_CMAX / 4; ++i)  // vt and ms are 0 beyond C
                    cfma(acc, vc[part + 4 * i] + vc[RG_NEX / 2 + part + 4 * i], conj(ms[(part + 4 * i) * RG_MLD + c]));
            }
            acc = group_sum<4>(acc);
            const cplx other = shfl_xor_c(acc, 4);   // the row c ^ 1
            if (part == 0 && c < C) {
                if (c < 2 * npr) {
                    if ((c & 1) == 0) { WE[e * NU2 + (c >> 1)] = acc + other; WO[e * NU2 + (c >> 1)] = acc - other; }
                } else {
                    WE[e * NU2 + npr + (c - 2 * npr)] = acc; WO[e * NU2 + npr + (c - 2 * npr)] = acc;
                }
                if (member == 0 && !first) {
                    cplx __attribute__((address_space(1)))* dst = Uout + ((int64_t)e * P + (kb - 1)) * PS_CMAX + c;
                    const cplx v = vc[c] + vc[RG_NEX / 2 + c];
                    dst->x = v.x; dst->y = v.y;
                }
            }
        }
        if (last) break;
        __syncthreads();  // B2: the weights are complete; M~ may be replaced
        RSTAMP(3);
        const bool svc = launder(tid) < RG_SVC;
        if (svc) fetch_m(kb, launder(tid));      // M~ of the next iteration: requested here, staged behind the p phase (which covers the round trip)
        // ---- p = w'(kb-1,:) g^T: this wave's units, then the sum with the partner wave's;  t = |H| p/|p|
        cplx t0, t1;
        {
            const int t = launder(tid);
            const int wave = t >> 6, lane = t & 63;
            const cplx* we = WE + (wave & 1) * NUL, *wo = WO + (wave & 1) * NUL;
            cplx p0 = mk(0, 0), p1 = mk(0, 0);
#pragma unroll
            for (int u = 0; u < NUL; ++u) {
                if (u == NUL - 1 && !last_slot_used) continue;
                const cplx we0 = we[u], wo0 = wo[u], we1 = we[NU2 + u], wo1 = wo[NU2 + u];
                cfma(p0, we0, E[u]); cfma(p0, wo0, O[u]);
                cfma(p1, we1, E[u]); cfma(p1, wo1, O[u]);
            }
            double* mine = pp + wave * 256 + lane;
            mine[0] = p0.x; mine[64] = p0.y; mine[128] = p1.x; mine[192] = p1.y;
            if (svc) stage_m(t);
            __syncthreads();  // Bp: the partner wave's share of p
            RSTAMP(8);
            const double* theirs = pp + (wave ^ 1) * 256 + lane;
            p0.x += theirs[0]; p0.y += theirs[64]; p1.x += theirs[128]; p1.y += theirs[192];
            const bool dvalid = dir_of(t) < D;
            t0 = dvalid ? unit_phase(hCur[0], p0, nyq) : mk(0, 0);
            t1 = dvalid ? unit_phase(hCur[1], p1, nyq) : mk(0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- this wave's partial  t conj(E_u), t conj(O_u)  summed over its 64 directions
        {
            const int t = launder(tid);
            const int lane = t & 63;
            const bool keep8 = (lane & 8) == 0, keep4 = (lane & 4) == 0;
            // two unit slots (sixteen values: slot, E / O, ear, re / im) at a time: four halving steps leave ONE register whose lane l
            // holds the sum over sixteen lanes of value 16 c + (l >> 2), the all-reduce inside the quad completes it
            double r[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                if (NUL % 2 == 1 && c == NCH - 1 && !last_slot_used) { r[c] = 0.0; continue; }   // (a chunk of the unused slot alone)
                double v[16];   // value j = 8 (slot & 1) + 4 eo + 2 e + ri
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    if (2 * c + s < NUL) {
                        const cplx gE = E[2 * c + s < NUL ? 2 * c + s : 0], gO = O[2 * c + s < NUL ? 2 * c + s : 0];
                        v[8 * s + 0] = fma(t0.x, gE.x, t0.y * gE.y); v[8 * s + 1] = fma(t0.y, gE.x, -(t0.x * gE.y));
                        v[8 * s + 2] = fma(t1.x, gE.x, t1.y * gE.y); v[8 * s + 3] = fma(t1.y, gE.x, -(t1.x * gE.y));
                        v[8 * s + 4] = fma(t0.x, gO.x, t0.y * gO.y); v[8 * s + 5] = fma(t0.y, gO.x, -(t0.x * gO.y));
                        v[8 * s + 6] = fma(t1.x, gO.x, t1.y * gO.y); v[8 * s + 7] = fma(t1.y, gO.x, -(t1.x * gO.y));
                    } else {
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[8 * s + j] = 0.0;
                    }
                }
                double h1[8], h2[4], h3[2];
#pragma unroll
                for (int j = 0; j < 8; ++j) h1[j] = halve32(v[j], v[j + 8]);                         // lanes >= 32: the odd slot
#pragma unroll
                for (int j = 0; j < 4; ++j) h2[j] = halve16(h1[j], h1[j + 4]);                       // odd rows: O
#pragma unroll
                for (int j = 0; j < 2; ++j) h3[j] = halve_row<DPP_ROW_MIRROR>(h2[j], h2[j + 2], keep8);   // bit 3: ear 1
                double x = halve_row<DPP_HALF_MIRROR>(h3[0], h3[1], keep4);                          // bit 2: imaginary part
                x += dpp_d<DPP_XOR1>(x);
                x += dpp_d<DPP_XOR2>(x);
                r[c] = x;
                __builtin_amdgcn_sched_barrier(0);   // (one chunk at a time: the scheduler otherwise starts them all)
            }
            if ((lane & 3) == 0) {
                double* wp = wpart + (t >> 6) * NVW + (lane >> 2);
#pragma unroll
                for (int c = 0; c < NCH; ++c) wp[16 * c] = r[c];
            }
        }
        // (the operand registers are free now: what the next iterations need from memory is requested here)
        {
            const int t = launder(tid);
            fetch_h(kb + 1, hNext, t);
            if (t < nord_pad) ring[(kb & 1) * SY_NORD + t] = ldg(bsc + (int64_t)(kb + 2 < P ? kb + 2 : P - 1) * nord_pad + t);   // row kb + 2 (row kb was read at the top of this iteration, before B1)
        }
        __syncthreads();  // B3: the waves' partials, M~ and the ring row are in place
        RSTAMP(4);
        // ---- the workgroup's partial in microphone rows, published as granules
        if (launder(tid) < RG_NEX) {
            const int t = launder(tid);
            const int xe = t >> 6, xr = (t >> 1) & 31, xri = t & 1;
            double v = 0.0;
            if (xr < C) {
                const int xu = xr < 2 * npr ? (xr >> 1) : npr + (xr - 2 * npr);
                const bool xneg = xr < 2 * npr && (xr & 1);
                const int h = xu >= NUL ? 1 : 0;
                const double* wE = wpart + h * NVW + 8 * (xu - h * NUL) + 2 * xe + xri;   // wave h of block 0; + 4: the O value
                double sE = 0.0, sO = 0.0;
#pragma unroll
                for (int q = 0; q < NB; ++q) { sE += wE[2 * q * NVW]; sO += wE[2 * q * NVW + 4]; }
                v = xneg ? sE - sO : sE + sO;
            }
            const gu64_t dst = part_ll + (((size_t)(kb & 1) * nWG + member) * RG_NEX + t) * 2;
            const u64 bits = (u64)__double_as_longlong(v), tg = (u64)(unsigned)kb << 32;
            gll_put(dst, tg | (bits & 0xffffffffull), local);
            gll_put(dst + 1, tg | (bits >> 32), local);
        }
        RSTAMP(5);
        hCur[0] = hNext[0]; hCur[1] = hNext[1];
    }
#undef RSTAMP
}

// the wave reduction of the kernel above on its own: in [64 lanes][16 values] -> out [16] (lane l ends up with the sum over all lanes of
// value l >> 2)
__global__ void __launch_bounds__(64) reg_reduce_selftest_kernel(const double* __restrict__ in, double* __restrict__ out) {
    const int lane = threadIdx.x;
    const double* v = in + lane * 16;
    const bool keep8 = (lane & 8) == 0, keep4 = (lane & 4) == 0;
    double h1[8], h2[4], h3[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) h1[j] = halve32(v[j], v[j + 8]);
#pragma unroll
    for (int j = 0; j < 4; ++j) h2[j] = halve16(h1[j], h1[j + 4]);
#pragma unroll
    for (int j = 0; j < 2; ++j) h3[j] = halve_row<DPP_ROW_MIRROR>(h2[j], h2[j + 2], keep8);
    double x = halve_row<DPP_HALF_MIRROR>(h3[0], h3[1], keep4);
    x += dpp_d<DPP_XOR1>(x);
    x += dpp_d<DPP_XOR2>(x);
    if ((lane & 3) == 0) out[lane >> 2] = x;
}

// argument blocks into device memory, stream-ordered (no host buffer has to outlive the call)
__global__ void __launch_bounds__(64) store_args_kernel(HalfSweepMulti m, HalfSweepArgs* dst) {
    const int nw = (int)(sizeof(HalfSweepArgs) / 8);
    for (int j = 0; j < m.n; ++j) {
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(&m.a[j]);
        unsigned long long* d = reinterpret_cast<unsigned long long*>(dst + j);
        for (int i = threadIdx.x; i < nw; i += 64) d[i] = src[i];
    }
}
static_assert(sizeof(HalfSweepArgs) % 8 == 0, "argument block in whole words");

constexpr int RG_NUL = 9;   // unit slots per wave: designs of up to 18 units (the em32: 15 antipodal pairs + 2 single capsules)
size_t reg_dyn_bytes(int nul, int nw) {
    const size_t nu2 = 2 * (size_t)nul, nvw = 16 * (((size_t)nul + 1) / 2);
    return sizeof(cplx) * ((size_t)2 * SY_NORD + PS_CMAX * RG_MLD + 4 * nu2) +
           sizeof(double) * (2 * RG_NEX + (size_t)nw * nvw + (size_t)nw * 256 + (size_t)nul * 64 * nw);
}
constexpr int RG_WAVES[] = {4, 6, 8, 10, 12};   // waves per workgroup of the instantiations
const void* reg_kernel_ptr(int nw) {
    switch (nw) {
        case 4: return reinterpret_cast<const void*>(sweep_reg_kernel<RG_NUL, 4>);
        case 6: return reinterpret_cast<const void*>(sweep_reg_kernel<RG_NUL, 6>);
        case 8: return reinterpret_cast<const void*>(sweep_reg_kernel<RG_NUL, 8>);
        case 10: return reinterpret_cast<const void*>(sweep_reg_kernel<RG_NUL, 10>);
        default: return reinterpret_cast<const void*>(sweep_reg_kernel<RG_NUL, 12>);
    }
}
void reg_set_attributes() {
    static PerDeviceOnce attr_once;   // (function attributes are per device)
    if (attr_once.first())
        for (int nw : RG_WAVES) HIP_CHECK(hipFuncSetAttribute(reg_kernel_ptr(nw), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
}
int reg_nwg(int D, int nw) { return (int)ceil_div(D, 32 * nw); }
// workgroups of `nw` waves the runtime can keep on one CU (registers, LDS, wave slots of THIS build of the kernel), once per device:
// the residency decisions below count on three 4-wave workgroups per CU and on one of every larger form
int reg_occupancy(int nw) {
    static std::mutex mu;
    static std::map<std::pair<int, int>, int> cache;
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find({dev, nw});
    if (it != cache.end()) return it->second;
    reg_set_attributes();
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reg_kernel_ptr(nw), 64 * nw, reg_dyn_bytes(RG_NUL, nw)) != hipSuccess) { (void)hipGetLastError(); nb = 0; }
    cache[{dev, nw}] = nb;
    return nb;
}
// thirds of a CU a workgroup of nw waves is counted as: four waves share a CU with two more of their kind, larger workgroups take it
int reg_wg_cost(int nw) { return nw == 4 ? 1 : 3; }

}  // namespace

int reg_sweep_max_units() { return 2 * RG_NUL; }
bool reg_sweep_supported(int D, int nmics, int nunits, int nOrd) {
    return nmics >= 2 && nmics <= PS_CMAX && nunits >= 1 && nunits <= 2 * RG_NUL && D >= 1 && reg_nwg(D, 12) <= 32 && nOrd <= SY_NORD;
}
size_t reg_sweep_ll_bytes(int D, int nmics) {
    const size_t nwg = (size_t)reg_nwg(D, 4);   // (the form with the most workgroups)
    (void)nmics;
    return sizeof(u64) * ((size_t)2 * nwg * RG_NEX * 2 + nwg + 64);
}
// the gate's unit: thirds of a CU per XCD (sweep_reg.hip workgroups of four waves are a third, larger ones a whole CU)
int reg_sweep_slots_per_xcd() { return 3 * (sweep_cu_budget() / 8); }
// Waves per workgroup for a launch of n designs: the smallest workgroups (the most CUs per design, the shortest bins) with which
// every workgroup has a CU of its own -- workgroups that share CUs unevenly run at the pace of the slowest -- and, when even
// twelve waves do not get there, four waves with up to three workgroups per CU.  0: the launch cannot be resident.
int reg_sweep_pick_waves(int D, int ndesigns) {
    if (ndesigns < 1 || ndesigns > REG_SWEEP_MAX) return 0;
    const int nsub = (int)ceil_div(ndesigns, 8), cus = sweep_cu_budget() / 8;
    if (const char* e = getenv("EMAGLS_REG_WAVES")) {   // (experiments: 4 ... 12 whenever it fits)
        const int nw = atoi(e);
        if ((nw == 4 || nw == 6 || nw == 8 || nw == 10 || nw == 12) && nsub * reg_nwg(D, nw) * reg_wg_cost(nw) <= 3 * cus && reg_occupancy(nw) >= (nw == 4 ? 3 : 1)) return nw;
    }
    for (int nw : RG_WAVES) if (nsub * reg_nwg(D, nw) <= cus && reg_occupancy(nw) >= 1) return nw;
    return (nsub * reg_nwg(D, 4) <= 3 * cus && reg_occupancy(4) >= 3) ? 4 : 0;
}
// Waves per workgroup when the designs of a launch are spread over all XCDs (0: not spread): taken when it gives every workgroup a
// CU of its own with FEWER waves than the XCD-local layout needs (EMAGLS_REG_SPREAD=0: never, =1: whenever it fits).
int reg_sweep_spread_waves(int D, int ndesigns) {
    const char* es = getenv("EMAGLS_REG_SPREAD");   // (read at every launch: a test switches layouts inside one process)
    const int mode = es ? atoi(es) : 2;
    if (mode == 0 || ndesigns < 1 || ndesigns > REG_SWEEP_MAX || getenv("EMAGLS_REG_WAVES")) return 0;
    const int cus = sweep_cu_budget(), local = reg_sweep_pick_waves(D, ndesigns);
    for (int nw : RG_WAVES)
        if (ndesigns * reg_nwg(D, nw) <= cus && reg_occupancy(nw) >= 1) return (mode == 1 || local == 0 || nw < local) ? nw : 0;
    return 0;
}
int reg_sweep_gate_cost(int D, int ndesigns) {
    if (const int ns = reg_sweep_spread_waves(D, ndesigns)) return (int)ceil_div((int64_t)ndesigns * reg_nwg(D, ns), 8) * reg_wg_cost(ns);
    const int nw = reg_sweep_pick_waves(D, ndesigns);
    return nw ? (int)ceil_div(ndesigns, 8) * reg_nwg(D, nw) * reg_wg_cost(nw) : 0;
}
int reg_sweep_capacity(int D) {
    int n = 0;
    for (int k = 8; k <= REG_SWEEP_MAX; k += 8) if (reg_sweep_pick_waves(D, k)) n = k;
    return n;
}
bool reg_sweep_fits(int D, int nmics, int nunits, int nOrd, int ndesigns) {
    return reg_sweep_supported(D, nmics, nunits, nOrd) && reg_sweep_pick_waves(D, ndesigns) != 0;
}

// args: `n` argument blocks in device memory
void launch_sweep_reg(const HalfSweepArgs* args_dev, const HalfSweepArgs& a0, int n, hipStream_t st) {
    const int ns = reg_sweep_spread_waves(a0.D, n);
    const int nw = ns ? ns : reg_sweep_pick_waves(a0.D, n);
    if (!nw) throw Error(2, "register-resident sweep: the launch cannot be resident");
    const int nWG = reg_nwg(a0.D, nw), spread = ns ? 1 : 0;
    const unsigned nblocks = spread ? (unsigned)(n * nWG) : 8u * (unsigned)nWG * (unsigned)ceil_div(n, 8);
    const size_t dyn = reg_dyn_bytes(RG_NUL, nw);
    reg_set_attributes();
    static const int prio = [] { const char* e = getenv("EMAGLS_REG_PRIO"); return e ? atoi(e) : 1; }();
    switch (nw) {
        case 4: sweep_reg_kernel<RG_NUL, 4><<<dim3(nblocks), 256, dyn, st>>>(args_dev, n, nWG, spread, prio); break;
        case 6: sweep_reg_kernel<RG_NUL, 6><<<dim3(nblocks), 384, dyn, st>>>(args_dev, n, nWG, spread, prio); break;
        case 8: sweep_reg_kernel<RG_NUL, 8><<<dim3(nblocks), 512, dyn, st>>>(args_dev, n, nWG, spread, prio); break;
        case 10: sweep_reg_kernel<RG_NUL, 10><<<dim3(nblocks), 640, dyn, st>>>(args_dev, n, nWG, spread, prio); break;
        default: sweep_reg_kernel<RG_NUL, 12><<<dim3(nblocks), 768, dyn, st>>>(args_dev, n, nWG, spread, prio); break;
    }
    KERNEL_CHECK();
}

// max |device - host| of the wave reduction on pseudo-random values (debug entry emagls_self_test)
double reg_reduce_selftest() {
    double h_in[64 * 16], h_out[16], want[16] = {0};
    unsigned long long x = 88172645463325252ull;
    for (int i = 0; i < 64 * 16; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h_in[i] = (double)(x % 2000001ull) / 1000000.0 - 1.0; }
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 16; ++j) want[j] += h_in[l * 16 + j];
    double *d_in = nullptr, *d_out = nullptr;
    HIP_CHECK(hipMalloc(&d_in, sizeof h_in));
    HIP_CHECK(hipMalloc(&d_out, sizeof h_out));
    HIP_CHECK(hipMemcpy(d_in, h_in, sizeof h_in, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemset(d_out, 0, sizeof h_out));
    reg_reduce_selftest_kernel<<<1, 64>>>(d_in, d_out);
    KERNEL_CHECK();
    HIP_CHECK(hipMemcpy(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost));
    HIP_CHECK(hipFree(d_in));
    HIP_CHECK(hipFree(d_out));
    double err = 0.0;
    for (int i = 0; i < 16; ++i) err = std::max(err, std::fabs(h_out[i] - want[i]));
    return err;
}

void store_sweep_args(const HalfSweepArgs* host, int n, HalfSweepArgs* dev, hipStream_t st) {
    for (int first = 0; first < n; first += SWEEP_MULTI_MAX) {
        HalfSweepMulti m{};
        m.n = std::min(SWEEP_MULTI_MAX, n - first);
        for (int j = 0; j < m.n; ++j) m.a[j] = host[first + j];
        store_args_kernel<<<1, 64, 0, st>>>(m, dev + first);
        KERNEL_CHECK();
    }
}

}  // namespace emagls
