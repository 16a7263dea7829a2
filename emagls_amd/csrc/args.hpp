// Kernel argument blocks (plain structs passed by value at launch).
#pragma once
#include "common.hpp"

namespace emagls {

struct FactorArgs {
    int S;            // rows of B_k
    int C;            // columns (output channels)
    int ldS;          // leading dimension of [c][s] arrays
    int kb0;          // first bin index handled (blockIdx.x = kb - kb0)
    int P;            // number of positive-frequency bins (Nyquist bin = P-1 uses real(b_n))
    // assemble mode
    const void* Tn;   // [n][c][ldS]  (real or complex), nullptr in dense mode
    const cplx* bn;   // [P][bn_stride]
    int nOrders;      // orders assembled (the first nOrders of each row of bn)
    int bn_stride;    // 0: = nOrders
    // dense mode
    const cplx* Xd;   // [kb][c][ldS] (per-bin) ; stride 0 allowed through xd_stride
    int64_t xd_stride;
    // regularisation
    int reg_mode;     // 0: 1/max(s, reg_c*smax)   1: pinv tolerance tol_dim*eps(smax)
    double reg_c;
    double tol_dim;
    // outputs
    cplx* Z;          // [kb][c][ldS]
    cplx* Vws;        // [blockIdx.x][c][ldS] Householder vectors workspace
    double* sv;       // [kb][C] singular values (unsorted) or nullptr
    // least-squares bins: W[e][kb][c] = sum_s Hq[e][kb][s] Z[s][c] for kb < ls_end
    const cplx* Hq;   // [e][kb][ldHq] or nullptr
    int64_t ldHq;
    int64_t hq_estride;
    int ls_end;
    const double* cond_ok;  // optional [P]: the back-transform (Z_k) is only needed for kb < ls_end and where cond_ok[kb] == 0
    // Gram route (gramroute.hip): route[kb] = 2 where the direct inverse was taken, 1 where R2w holds A = B^H B for the Jacobi kernel
    int* route;       // [P] (zeroed per execute)
    int* status;      // plan status flags; [2] is set when a Gram-route bin turns out ill-conditioned, [3] = the highest such bin
                      // (the host then moves the start of the route behind it and re-runs)
    double cond_limit; // Gram form: cond(B_k) above which status[2..3] are raised (10x the host's estimate limit)
    int jrun;         // Jacobi: consecutive bins per workgroup (warm start from the neighbour's rotations); 0/1 = independent
    int nbins;        // set by the launcher
    int jsplit;       // set by the launcher: the first jsplit bins (Householder route: full Jacobi, the long ones) get one workgroup each
    int hq_conj;      // Hq holds conj(H conj(Q)) (the row-solve form used when Q is not materialised)
    cplx* W;          // [e][P][C]
    int* sweeps_out;  // optional [kb]
    // inter-kernel workspaces, indexed by blockIdx.x (= kb - kb0)
    double* tauw;     // [bin][C]       Householder scalars
    cplx* R2w;        // [bin][C][C]    triangular factor (upper)
    cplx* Nw;         // [bin][C][C]    U2 diag(s_reg) V^H
    cplx* Mw;         // [bin][C][C]    V diag(s_reg/s) V^H (optional, for the direction-space sweep operands)
};

struct DenseSweepArgs {
    int D, C, ldD, P;
    const void* X;        // [kb][c][ldD] (TX real or complex)
    int64_t x_stride;
    const void* Zd;       // [kb][c][ldD]
    int64_t z_stride;
    const double* Habs;   // [e][kb-kabs0][ldH]
    int64_t ldH;
    int kabs0;
    cplx* Wpart;          // [2][nWG][2][C]
    cplx* W;
    int nWG, dpw;
    int kfirst;
};

// sweep on ONE direction-space operand per bin: Y_reg_inv_k = conj(G_k) conj(M_k) is applied as
//   v = t conj(G_k) (per workgroup partial)  ->  W(k,:) = (sum of partials) conj(M_k)  at the start of the next launch
struct HalfSweepArgs {
    int D, C, ldD, P;
    const cplx* G;          // [kb][c][ldD], base shifted so that it is indexed by kb
    int64_t g_stride;
    const cplx* Yri;        // same indexing; only ill-conditioned bins (cond_ok == 0) are filled and used
    const cplx* Mw;         // [kb][C][C], base shifted so that it is indexed by kb
    const double* cond_ok;  // [P]
    const double* Habs;     // [e][kb-kabs0][ldH]
    int64_t ldH;
    int kabs0;
    cplx* Wpart;            // [2][pair][nWG]
    cplx* W;                // [e][P][C]
    int nWG;
    int kfirst;
    unsigned long long* ll;  // persistent sweep: [2][nWG_persist][2C][4] granules {payload32, tag32}
    int* abort_flag;         // persistent sweep: set when a workgroup gave up waiting for its peers
    const int* skip_flag;    // persistent sweep, optional: non-zero at launch = operands unusable (MagLS basis too ill-conditioned for the
                             // inverse form, status word 4): the design's workgroups leave at once, the host re-runs on the launch-per-bin sweep
    long long* timing;       // optional [P][16] wall-clock stamps (EMAGLS_SWEEP_TIMING), else null
    int fetch_mode;          // persistent sweep: where the next bin's operands are requested (sweep_persist.hip; EMAGLS_SWEEP_FETCH, default 0)
    long long wait_ticks;    // persistent sweep: ticks of the 100 MHz wall clock after which a wait for peers gives up (EMAGLS_SWEEP_WAIT_MS, default 20 ms)
    int force_global;        // persistent sweep: keep the write-through (sc1) stores even on one XCD (EMAGLS_PERSIST_GLOBAL=1)
    // operand synthesis (sweep_synth.hip): the chain runs on the C microphones, the slab of bin k is evaluated inside the launch as
    // g_k[d][j] = sum_n bsc[k][n] pi_n(cos(angle between HRIR direction d and microphone j)); Mw then holds Pm^T M_k Pm (C x C)
    const double* dir_azi;   // [D] HRIR grid
    const double* dir_zen;
    const double* mic_azi;   // [C] microphone grid (mic_zen null: equatorial array)
    const double* mic_zen;
    const int* smap;         // [34] row -> microphone in the chain's order (antipodal pairs first: rows 2u, 2u + 1), [32] pairs, [33] singles
    const cplx* bsc;         // [P][nord_pad] Chebyshev coefficients of the bin's Legendre series, zero beyond the design's own orders
    int nord_pad;            // even
    int synth_split;         // per cent of the producers' unit groups evaluated before barrier B1 (the rest between B1 and B2); 0 = all
    int synth_prio;          // issue priority of the producer waves (0..2; the chain's waves run at 3)
    const cplx* Winit;       // [2][32] W(kfirst-1,:) Pm: the chain's start value
    cplx* U;                 // [2][P][32] totals u(k) of every swept bin (the filters' rows are formed from them after the launch)
};
constexpr int SWEEP_MULTI_MAX = 16;   // designs per sweep launch: one per XCD up to 8, two per XCD (two workgroups per CU) up to 16
struct HalfSweepMulti {
    int n;
    HalfSweepArgs a[SWEEP_MULTI_MAX];
};
// designs per launch of the register-resident sweep (sweep_reg.hip): its argument blocks lie in device memory
constexpr int REG_SWEEP_MAX = 32;


}  // namespace emagls
