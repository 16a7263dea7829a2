// Direction-space operands of the magnitude-least-squares sweep, built once per design for all
// swept bins (bins are independent here; only the sweep itself is sequential).
//
//   G_k   = pwGrid_k.' = Q B_k = sum_n b_n(k) (Q T_n)            [D x C]   -- exact, 20 cMAC per entry
//   Yri_k = Y_reg_inv_k = conj(U_k) diag(s_reg) V^T
//         = conj(G_k) conj(M_k),  M_k = V diag(s_reg/s) V^H       [D x C]   -- U_k = G_k V S^-1 (applied inside the sweep)
//
// The second identity forms U_k from G_k V S^-1, which is only as orthonormal as eps*cond(G_k).  That is
// harmless for the swept bins (k >= k_cut, f >= 1 kHz: cond ~ 1e2..1e3, error ~ eps*100*cond), and the
// least-squares bins below k_cut -- where cond reaches 1e13 -- never take this route (factor.hip).
// Bins whose condition number exceeds COND_LIMIT get their Yri_k from the orthonormal S-space factor
// (Yri_k = conj(Q) Z_k) by yri_accurate_kernel.
//
// Why direction space: a launch starts with cold L2 (kernel boundaries invalidate it), so the per-launch
// cost of the sweep is set by the bytes EVERY workgroup must fetch; S-space operands (B_k, Z_k: 2 x 160 KB)
// are needed whole by every workgroup, direction-space slabs are disjoint (2 x 1.08 MB / nWG).
#include "kernels.hpp"

namespace emagls {

constexpr double COND_LIMIT = 1.0e4;

// Order terms of pwGrid.':  G_k = conj(Y) diag(b_n(k)) E^T = sum_n b_n(k) QT_n  with
//   QT[n][c][d] = sum_{s in [n^2,(n+1)^2)} Yc[d][s] E[c][s]
// (equal to Q T_n, but needs neither Q nor R: the order blocks of conj(Y) and of the array matrix E suffice).
// One workgroup = QT_TD directions (16, or fewer when S is large: the rows must fit the LDS), all orders; the rows of Yc
// sit in LDS, thread = (channel c, direction dl) with dl fastest: the lanes of a channel share every E load and store a
// contiguous run.
template <typename T, int QT_TD>
__global__ void __launch_bounds__(512) qt_kernel(const T* __restrict__ Yc, int64_t ldY, const T* __restrict__ E, int ldE,
                                                 int D, int S, int C, int nOrders, T* __restrict__ QT, int64_t ldD, size_t bstride) {
    Yc = boff(Yc, bstride); E = boff(E, bstride); QT = boff(QT, bstride);
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    T* ys = reinterpret_cast<T*>(dyn);  // [QT_TD][S+1]
    const int ldq = S + 1;
    const int d0 = blockIdx.x * QT_TD;
    for (int idx = threadIdx.x; idx < QT_TD * S; idx += blockDim.x) {
        const int dl = idx / S, s = idx % S;
        ys[(size_t)dl * ldq + s] = (d0 + dl < D) ? Yc[(int64_t)(d0 + dl) * ldY + s] : zero_of<T>();
    }
    __syncthreads();
    const int dl = threadIdx.x % QT_TD;
    const T* y = ys + (size_t)dl * ldq;
    for (int c = threadIdx.x / QT_TD; c < C; c += 512 / QT_TD) {
        const T* e = E + (int64_t)c * ldE;
        // (tried: eight masked loads of E in flight per pass instead of this two-term loop: 273 -> 313 us per 8-design launch)
        // (tried at the end of round 3: the order terms from the column-major SH matrix, lanes = 64 directions, no LDS: 154 us with
        // four channel groups per direction block -- the matrix is then re-read four times from beyond the L2 -- and 452 us with
        // all channels in one wave, against 155 us here: every multiply-add still needs its own load of an E entry)
        for (int n = 0; n < nOrders; ++n) {
            const int sb = n * n, se = min(S, (n + 1) * (n + 1));
            T a0 = zero_of<T>(), a1 = zero_of<T>();
            int s = sb;
            for (; s + 1 < se; s += 2) { cfma(a0, y[s], e[s]); cfma(a1, y[s + 1], e[s + 1]); }
            if (s < se) cfma(a0, y[s], e[s]);
            if (d0 + dl < D) QT[((int64_t)n * C + c) * ldD + d0 + dl] = a0 + a1;
        }
    }
}

// G kernel: one workgroup = 64 directions x a chunk of swept bins; a wave owns one channel pair (c, c + 4) at a time and
// its 64 lanes are 64 consecutive directions, so every store instruction writes one contiguous 1 KB run (short
// scattered runs cost HBM page locality).  The thread's order terms QT[n][c][d] stay in registers for the whole chunk,
// the chunk's b_n(k) table sits in LDS and every LDS broadcast of a b_n feeds two complex FMAs.
//   G [kb - k0][c][d]
constexpr int DSP_TD = 64;
constexpr int DSP_NMAX = 48;  // orders held in registers (simulation order <= 47: array radius up to 10.9 cm at 48 kHz)

template <typename T, int NMAX>
__global__ void __launch_bounds__(256) dspace_g_kernel(const T* __restrict__ QT, int64_t ldD, const cplx* __restrict__ bn,
                                                       int nOrders, int D, int C, int P, int k0, int bins_per_chunk,
                                                       cplx* __restrict__ G, int k_end, size_t bstride) {
    QT = boff(QT, bstride); bn = boff(bn, bstride); G = boff(G, bstride);
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    cplx* bs = reinterpret_cast<cplx*>(dyn);  // [bins_per_chunk][NMAX], rows zero padded (branch-free inner loop, see below)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int d = blockIdx.x * DSP_TD + lane;
    const int kb_begin = k0 + blockIdx.y * bins_per_chunk;
    const int kb_end = min(k_end, kb_begin + bins_per_chunk);   // (k_end <= P: the bins that need a materialised operand)
    for (int idx = threadIdx.x; idx < (kb_end - kb_begin) * NMAX; idx += 256) {
        const int kb = kb_begin + idx / NMAX, n = idx % NMAX;
        cplx b = mk(0, 0);
        if (n < nOrders) b = bn[(int64_t)kb * nOrders + n];
        if (kb == P - 1) b.y = 0.0;  // Nyquist: real(Bn)
        bs[idx] = b;
    }
    __syncthreads();
    if (d >= D) return;
    for (int cbase = 0; cbase < C; cbase += 8) {
        const int ca = cbase + wave, cb2 = ca + 4;
        if (ca >= C) break;   // (wave-uniform)
        const bool two = cb2 < C;
        T qa[NMAX], qb[NMAX];
#pragma unroll
        for (int n = 0; n < NMAX; ++n) {
            qa[n] = (n < nOrders) ? QT[((int64_t)n * C + ca) * ldD + d] : zero_of<T>();
            qb[n] = (two && n < nOrders) ? QT[((int64_t)n * C + cb2) * ldD + d] : zero_of<T>();
        }
        cplx* g = G + ((int64_t)(kb_begin - k0) * C + ca) * ldD + d;
        const int64_t gstep = (int64_t)C * ldD;
        for (int kb = kb_begin; kb < kb_end; ++kb, g += gstep) {
            const cplx* b = bs + (size_t)(kb - kb_begin) * NMAX;
            cplx bb[NMAX];   // all reads of the row before the first use (a test per order serialises read -> wait -> use)
#pragma unroll
            for (int n = 0; n < NMAX; ++n) bb[n] = b[n];
            cplx g0 = mk(0, 0), g1 = mk(0, 0), h0 = mk(0, 0), h1 = mk(0, 0);
#pragma unroll
            for (int n = 0; n + 1 < NMAX; n += 2) {
                cfma(g0, bb[n], qa[n]); cfma(h0, bb[n], qb[n]);
                cfma(g1, bb[n + 1], qa[n + 1]); cfma(h1, bb[n + 1], qb[n + 1]);
            }
            if (NMAX & 1) { cfma(g0, bb[NMAX - 1], qa[NMAX - 1]); cfma(h0, bb[NMAX - 1], qb[NMAX - 1]); }
            stream_store(g, g0 + g1);
            if (two) stream_store(g + 4 * ldD, h0 + h1);
        }
    }
}

// G kernel for the complex SH basis, in real arithmetic on the order terms.  With Y_c = Y_r T (T unitary, block diagonal
// per order) the array matrix is E_c = T_N^H E_r T and the order terms satisfy QT_c,n = QTr_n conj(T_N) with REAL
// QTr_n = Y_r,n E_r,n^T, so G_c = (sum_n b_n(k) QTr_n) conj(T_N): a complex-times-real sum (2 FMAs per term instead
// of 4) followed by the fixed two-term channel transform
//     G(n,+m) = (-1)^m/sqrt2 (ga - i gb),   G(n,-m) = 1/sqrt2 (ga + i gb),   ga/gb = the real-basis rows (n,+m)/(n,-m),
// QTr(n,+m) = Re((-1)^m QT(n,+m) + QT(n,-m))/sqrt2,  QTr(n,-m) = -Im((-1)^m QT(n,+m) - QT(n,-m))/sqrt2  (m = 0: unchanged).
// eMagLS2 (channels = microphones, E = Y_mic): QT_c is real as it stands (sh_order < 0: all channels independent).
// A wave owns one unit (an SH pair, or two independent channels) at a time; lanes = 64 consecutive directions.
template <int NMAX, bool NT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) dspace_g_real_kernel(const cplx* __restrict__ QT, int64_t ldD, const cplx* __restrict__ bn,
                                                            int nOrders, int D, int C, int P, int k0, int bins_per_chunk,
                                                            cplx* __restrict__ G, int sh_order, size_t bstride) {
    QT = boff(QT, bstride); bn = boff(bn, bstride); G = boff(G, bstride);
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    cplx* bs = reinterpret_cast<cplx*>(dyn);  // [bins_per_chunk][nOrders]
    __shared__ int u_a[32], u_b[32], u_t[32], u_n;   // unit: channels a, b (b = -1: none), type 0 = SH pair (+m, -m, sign in bit 1)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int d = blockIdx.x * DSP_TD + lane;
    const int kb_begin = k0 + blockIdx.y * bins_per_chunk;
    const int kb_end = min(P, kb_begin + bins_per_chunk);
    // rows padded to NMAX orders with zeros: the inner loop below is branch-free (a test per order made every LDS read
    // wait for its use: 1590 cycles per bin and unit instead of the 320 of its 80 FMAs)
    for (int idx = threadIdx.x; idx < (kb_end - kb_begin) * NMAX; idx += 256) {
        const int kb = kb_begin + idx / NMAX, n = idx % NMAX;
        cplx b = mk(0, 0);
        if (n < nOrders) b = bn[(int64_t)kb * nOrders + n];
        if (kb == P - 1) b.y = 0.0;  // Nyquist: real(Bn)
        bs[idx] = b;
    }
    if (threadIdx.x == 0) {
        int nu = 0;
        if (sh_order >= 0) {
            for (int n = 0; n <= sh_order; ++n)
                for (int mm = 1; mm <= n; ++mm) { u_a[nu] = n * n + n + mm; u_b[nu] = n * n + n - mm; u_t[nu] = (mm & 1) ? 2 : 0; ++nu; }
            for (int n = 0; n <= sh_order; n += 2) { u_a[nu] = n * n + n; u_b[nu] = (n + 1 <= sh_order) ? (n + 1) * (n + 1) + n + 1 : -1; u_t[nu] = 1; ++nu; }
        } else {
            for (int ca = 0; ca < C; ca += 2) { u_a[nu] = ca; u_b[nu] = ca + 1 < C ? ca + 1 : -1; u_t[nu] = 1; ++nu; }
        }
        u_n = nu;
    }
    __syncthreads();
    if (d >= D) return;
    const double r2 = 0.70710678118654752440;
    for (int u = wave; u < u_n; u += 4) {
        const int ca = u_a[u], cb2 = u_b[u], ty = u_t[u];
        const bool two = cb2 >= 0;
        double qa[NMAX], qb[NMAX];
#pragma unroll
        for (int n = 0; n < NMAX; ++n) {
            cplx xa = mk(0, 0), xb = mk(0, 0);
            if (n < nOrders) {
                xa = QT[((int64_t)n * C + ca) * ldD + d];
                if (two) xb = QT[((int64_t)n * C + cb2) * ldD + d];
            }
            if (ty == 1) { qa[n] = xa.x; qb[n] = xb.x; }
            else {
                const double sg = (ty & 2) ? -1.0 : 1.0;
                qa[n] = (sg * xa.x + xb.x) * r2;
                qb[n] = -(sg * xa.y - xb.y) * r2;
            }
        }
        cplx* ga_p = G + ((int64_t)(kb_begin - k0) * C + ca) * ldD + d;
        const int64_t boffs = ((int64_t)cb2 - ca) * ldD, gstep = (int64_t)C * ldD;
        for (int kb = kb_begin; kb < kb_end; ++kb, ga_p += gstep) {
            const cplx* b = bs + (size_t)(kb - kb_begin) * NMAX;
            cplx bb[NMAX];   // every read of the bin's b_n row is issued before the first use (LDS returns in order)
#pragma unroll
            for (int n = 0; n < NMAX; ++n) bb[n] = b[n];
            double gax = 0.0, gay = 0.0, gbx = 0.0, gby = 0.0, hax = 0.0, hay = 0.0, hbx = 0.0, hby = 0.0;
#pragma unroll
            for (int n = 0; n + 1 < NMAX; n += 2) {   // (orders beyond nOrders: b = 0 and q = 0)
                gax = fma(bb[n].x, qa[n], gax); gay = fma(bb[n].y, qa[n], gay); gbx = fma(bb[n].x, qb[n], gbx); gby = fma(bb[n].y, qb[n], gby);
                hax = fma(bb[n + 1].x, qa[n + 1], hax); hay = fma(bb[n + 1].y, qa[n + 1], hay); hbx = fma(bb[n + 1].x, qb[n + 1], hbx); hby = fma(bb[n + 1].y, qb[n + 1], hby);
            }
            if (NMAX & 1) {
                gax = fma(bb[NMAX - 1].x, qa[NMAX - 1], gax); gay = fma(bb[NMAX - 1].y, qa[NMAX - 1], gay);
                gbx = fma(bb[NMAX - 1].x, qb[NMAX - 1], gbx); gby = fma(bb[NMAX - 1].y, qb[NMAX - 1], gby);
            }
            const double ax = gax + hax, ay = gay + hay, bx = gbx + hbx, by = gby + hby;
            cplx o0, o1;
            if (ty == 1) { o0 = mk(ax, ay); o1 = mk(bx, by); }
            else {
                const double sg = (ty & 2) ? -r2 : r2;
                o0 = mk(sg * (ax + by), sg * (ay - bx));        // (-1)^m/sqrt2 (ga - i gb)
                o1 = mk(r2 * (ax - by), r2 * (ay + bx));        // 1/sqrt2 (ga + i gb)
            }
            if (NT) {
                double* q0 = reinterpret_cast<double*>(ga_p);
                __builtin_nontemporal_store(o0.x, q0); __builtin_nontemporal_store(o0.y, q0 + 1);
                if (two) { double* q1 = reinterpret_cast<double*>(ga_p + boffs); __builtin_nontemporal_store(o1.x, q1); __builtin_nontemporal_store(o1.y, q1 + 1); }
            } else {
                stream_store(ga_p, o0);
                if (two) stream_store(ga_p + boffs, o1);
            }
        }
    }
}

// cond_ok[kb] = 1 when smax <= COND_LIMIT * smin for bin kb (the cheap identity is accurate), else 0.  Only bins of the
// orthonormal route (kb < hh_end) can be cleared: they alone have the accurate S-space inverse Z_k to fall back on.  A
// Gram-route bin keeps 1 whatever its singular values say (the conditioning check of that route raises its own flag and the
// host moves the route's start; the operands of the fallback do not exist for it).
__global__ void cond_flag_kernel(const double* __restrict__ sv, int C, int P, int hh_end, double* __restrict__ cond_ok, size_t bstride) {
    sv = boff(sv, bstride); cond_ok = boff(cond_ok, bstride);
    const int kb = blockIdx.x * blockDim.x + threadIdx.x;
    if (kb >= P) return;
    double smax = 0.0, smin = INFINITY;
    for (int i = 0; i < C; ++i) { const double s = sv[(int64_t)kb * C + i]; smax = fmax(smax, s); smin = fmin(smin, s); }
    cond_ok[kb] = (kb >= hh_end || smax <= COND_LIMIT * smin) ? 1.0 : 0.0;
}

// ill-conditioned swept bins: Yri[c][d] = sum_s conj(Q[d][s]) Z_k[c][s]  (orthonormal S-space factor)
template <typename T>
__global__ void __launch_bounds__(256) yri_accurate_kernel(const T* __restrict__ Q, int64_t ldQ, const cplx* __restrict__ Z,
                                                           int ldS, const double* __restrict__ cond_ok, int D, int S, int C, int k0,
                                                           cplx* __restrict__ Yri, int64_t ldD, size_t bstride) {
    Q = boff(Q, bstride); Z = boff(Z, bstride); cond_ok = boff(cond_ok, bstride); Yri = boff(Yri, bstride);
    const int kb = k0 + blockIdx.y;
    if (cond_ok[kb] != 0.0) return;
    const int c = threadIdx.x >> 3, dl = threadIdx.x & 7;
    if (c >= C) return;
    for (int d = blockIdx.x * 8 + dl; d < D; d += gridDim.x * 8) {
        const T* q = Q + (int64_t)d * ldQ;
        const cplx* z = Z + ((int64_t)kb * C + c) * ldS;
        cplx a0 = mk(0, 0), a1 = mk(0, 0);
        int s = 0;
        for (; s + 1 < S; s += 2) { cfma(a0, conj(q[s]), z[s]); cfma(a1, conj(q[s + 1]), z[s + 1]); }
        if (s < S) cfma(a0, conj(q[s]), z[s]);
        Yri[((int64_t)(kb - k0) * C + c) * ldD + d] = a0 + a1;
    }
}

// The same product for the real basis from the COLUMN-major SH matrix (Ycm[s][d] = conj(Q[d][s]) there): lanes = 64 consecutive
// directions (one 512-byte load per basis function), a wave accumulates 16 channels of its 64 directions in registers, the
// flagged bin's rows Z_k[c][s] pass through LDS in chunks of 32 basis functions (broadcast reads).  The kernel above walks
// eight ROWS of the row-major matrix per wave (eight cache lines per load instruction) and re-reads every row once per
// channel: 5.2 ms for the 8 designs of a 2 cm array of config 4 (64 flagged bins each), a third of their batch.
constexpr int YA_SCH = 32;   // basis functions per LDS chunk
__global__ void __launch_bounds__(256) yri_accurate_cm_kernel(const double* __restrict__ Ycm, int64_t ldY, const cplx* __restrict__ Z, int ldS,
                                                              const double* __restrict__ cond_ok, int D, int S, int C, int k0,
                                                              cplx* __restrict__ Yri, int64_t ldD, size_t bstride) {
    Ycm = boff(Ycm, bstride); Z = boff(Z, bstride); cond_ok = boff(cond_ok, bstride); Yri = boff(Yri, bstride);
    const int kb = k0 + blockIdx.y;
    if (cond_ok[kb] != 0.0) return;   // (uniform for the workgroup)
    __shared__ __attribute__((aligned(16))) cplx zs[32][YA_SCH + 1];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int d = blockIdx.x * 128 + (wave & 1) * 64 + lane;     // two direction tiles x two channel halves per workgroup
    const int ch0 = (wave >> 1) * 16;
    const int dc = d < D ? d : D - 1;
    cplx acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = mk(0, 0);
    for (int s0 = 0; s0 < S; s0 += YA_SCH) {
        __syncthreads();   // (the previous chunk has been read)
        for (int idx = tid; idx < 32 * YA_SCH; idx += 256) {
            const int c = idx / YA_SCH, s = idx % YA_SCH;
            zs[c][s] = (c < C && s0 + s < S) ? Z[((int64_t)kb * C + c) * ldS + s0 + s] : mk(0, 0);
        }
        __syncthreads();
        const int ns = min(YA_SCH, S - s0);
        double y[YA_SCH];
#pragma unroll
        for (int s = 0; s < YA_SCH; ++s) y[s] = s < ns ? Ycm[(int64_t)(s0 + s) * ldY + dc] : 0.0;   // (all loads of the chunk before the first use)
#pragma unroll
        for (int s = 0; s < YA_SCH; ++s) {
#pragma unroll
            for (int i = 0; i < 16; ++i) cfma(acc[i], y[s], zs[ch0 + i][s]);
        }
    }
    if (d < D) {
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (ch0 + i < C) Yri[((int64_t)(kb - k0) * C + ch0 + i) * ldD + d] = acc[i];
    }
}

template <typename T, int TD>
static void qt_launch(const void* Yc, int64_t ldY, const void* E, int ldE, int D, int S, int C, int nOrders, void* QT, int64_t ldD,
                      hipStream_t st) {
    const size_t dyn = sizeof(T) * (size_t)TD * (S + 1);
    static PerDeviceOnce attr_once;   // (function attributes are per device)
    if (attr_once.first()) {
        HIP_CHECK(hipFuncSetAttribute((const void*)qt_kernel<T, TD>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    }
    qt_kernel<T, TD><<<bgrid((unsigned)ceil_div(D, TD)), 512, dyn, st>>>((const T*)Yc, ldY, (const T*)E, ldE, D, S, C, nOrders, (T*)QT, ldD, batch_ctx().stride);
    KERNEL_CHECK();
}
template <typename T>
static void qt_impl(const void* Yc, int64_t ldY, const void* E, int ldE, int D, int S, int C, int nOrders, void* QT, int64_t ldD,
                    hipStream_t st) {
    const size_t row = sizeof(T) * (size_t)(S + 1);
    if (16 * row <= 150 * 1024) qt_launch<T, 16>(Yc, ldY, E, ldE, D, S, C, nOrders, QT, ldD, st);
    else if (8 * row <= 150 * 1024) qt_launch<T, 8>(Yc, ldY, E, ldE, D, S, C, nOrders, QT, ldD, st);
    else if (4 * row <= 150 * 1024) qt_launch<T, 4>(Yc, ldY, E, ldE, D, S, C, nOrders, QT, ldD, st);
    else throw Error(2, "qt: simulation order too large for the LDS-resident tile");
}
void launch_qt(const void* Yc, int64_t ldY, const void* E, int ldE, int D, int S, int C, int nOrders, bool is_cplx, void* QT,
               int64_t ldD, hipStream_t st) {
    if (is_cplx) qt_impl<cplx>(Yc, ldY, E, ldE, D, S, C, nOrders, QT, ldD, st);
    else qt_impl<double>(Yc, ldY, E, ldE, D, S, C, nOrders, QT, ldD, st);
}

template <typename T>
static void dspace_g_impl(const void* QT, int64_t ldD, const void* bn, int nOrders, int D, int C, int P, int k0, void* G,
                          hipStream_t st, int k_end) {
    const int nbins = k_end - k0;
    if (nbins <= 0) return;
    if (nOrders > DSP_NMAX) throw Error(2, "dspace: simulation order above 47 is not supported in this build");
    int chunks = nbins >= 64 ? 8 : 1;
    const int nmax = nOrders <= 20 ? 20 : nOrders <= 32 ? 32 : DSP_NMAX;
    while (sizeof(cplx) * (size_t)ceil_div(nbins, chunks) * nmax > 56 * 1024) ++chunks;  // b_n table of a chunk in LDS
    const int bpc = (nbins + chunks - 1) / chunks;
    // (tried: capping the kernel at 1-3 workgroups per CU through a larger LDS request, so that other batches' kernels find
    // room next to it: 2099-2118 sets/s against 2142 uncapped in long runs -- it is not this kernel's occupancy that limits them)
    const size_t dyn = sizeof(cplx) * (size_t)bpc * nmax;
    const dim3 grid((unsigned)ceil_div(D, DSP_TD), chunks);
    if (nOrders <= 20)
        dspace_g_kernel<T, 20><<<bgrid(grid), 256, dyn, st>>>((const T*)QT, ldD, (const cplx*)bn, nOrders, D, C, P, k0, bpc, (cplx*)G, k_end, batch_ctx().stride);
    else if (nOrders <= 32)
        dspace_g_kernel<T, 32><<<bgrid(grid), 256, dyn, st>>>((const T*)QT, ldD, (const cplx*)bn, nOrders, D, C, P, k0, bpc, (cplx*)G, k_end, batch_ctx().stride);
    else
        dspace_g_kernel<T, DSP_NMAX><<<bgrid(grid), 256, dyn, st>>>((const T*)QT, ldD, (const cplx*)bn, nOrders, D, C, P, k0, bpc, (cplx*)G, k_end, batch_ctx().stride);
    KERNEL_CHECK();
}
void launch_dspace_g(const void* QT, int64_t ldD, bool is_cplx, const void* bn, int nOrders, int D, int C, int P, int k0, void* G,
                     hipStream_t st, int real_mode, int sh_order, int k_end) {
    if (k_end < 0 || k_end > P) k_end = P;
    if (k_end < P && is_cplx) throw Error(2, "dspace: a bin range is only available on the real order terms");
    if (is_cplx && real_mode && nOrders <= 32 && C <= 64 && P - k0 > 0) {
        int chunks = 8;    // 8-design launch: 2 -> 929, 4 -> 896, 8 -> 894, 12 -> 1032, 16 -> 1115, 24 -> 1385 us (every chunk re-reads QT)
        const bool nt = true;  // streaming stores: G is written once here and read once by the sweep
        const int nbins = P - k0;
        const int nmax = nOrders <= 12 ? 12 : nOrders <= 20 ? 20 : 32;   // orders held in registers (table rows padded to it)
        while (sizeof(cplx) * (size_t)ceil_div(nbins, chunks) * nmax > 56 * 1024) ++chunks;
        const int bpc = (nbins + chunks - 1) / chunks;
        const size_t dyn = sizeof(cplx) * (size_t)bpc * nmax;
        const dim3 grid((unsigned)ceil_div(D, DSP_TD), chunks);
#define EMAGLS_DSPR(NM, NTS) dspace_g_real_kernel<NM, NTS><<<bgrid(grid), 256, dyn, st>>>((const cplx*)QT, ldD, (const cplx*)bn, nOrders, D, C, P, k0, bpc, (cplx*)G, sh_order, batch_ctx().stride)
        if (nmax == 12) { if (nt) EMAGLS_DSPR(12, true); else EMAGLS_DSPR(12, false); }
        else if (nmax == 20) { if (nt) EMAGLS_DSPR(20, true); else EMAGLS_DSPR(20, false); }
        else EMAGLS_DSPR(32, false);
#undef EMAGLS_DSPR
        KERNEL_CHECK();
        return;
    }
    if (is_cplx) dspace_g_impl<cplx>(QT, ldD, bn, nOrders, D, C, P, k0, G, st, k_end);
    else dspace_g_impl<double>(QT, ldD, bn, nOrders, D, C, P, k0, G, st, k_end);
}
void launch_cond_flags(const double* sv, int C, int P, int hh_end, double* cond_ok, hipStream_t st) {
    cond_flag_kernel<<<bgrid((P + 255) / 256), 256, 0, st>>>(sv, C, P, hh_end, cond_ok, batch_ctx().stride);
    KERNEL_CHECK();
}
void launch_yri_accurate(const void* Q, int64_t ldQ, bool is_cplx, const void* Z, int ldS, const double* cond_ok, int D, int S, int C,
                         int P, int k0, void* Yri, int64_t ldD, hipStream_t st, const void* Ycm, int64_t ldYcm) {
    if (P - k0 <= 0) return;
    if (!is_cplx && Ycm && C <= 32) {   // real basis: the coalesced form on the column-major matrix
        yri_accurate_cm_kernel<<<bgrid(dim3((unsigned)ceil_div(D, 128), P - k0)), 256, 0, st>>>((const double*)Ycm, ldYcm, (const cplx*)Z, ldS, cond_ok, D, S, C, k0,
                                                                                               (cplx*)Yri, ldD, batch_ctx().stride);
        KERNEL_CHECK();
        return;
    }
    dim3 grid(32, P - k0);  // flagged bins are rare: workgroups of well-conditioned bins exit at once
    if (is_cplx) yri_accurate_kernel<cplx><<<bgrid(grid), 256, 0, st>>>((const cplx*)Q, ldQ, (const cplx*)Z, ldS, cond_ok, D, S, C, k0, (cplx*)Yri, ldD, batch_ctx().stride);
    else yri_accurate_kernel<double><<<bgrid(grid), 256, 0, st>>>((const double*)Q, ldQ, (const cplx*)Z, ldS, cond_ok, D, S, C, k0, (cplx*)Yri, ldD, batch_ctx().stride);
    KERNEL_CHECK();
}

}  // namespace emagls
