// Operand synthesis shared by the resident sweeps (sweep_synth.hip, sweep_reg.hip): the Chebyshev evaluation of one group of units
// with its coefficient reads issued a pass ahead.
#pragma once
#include "kernels.hpp"

namespace emagls {
namespace {

constexpr int SY_NORD = 96;     // orders a ring slot holds (simulation order <= 95)

// LDS reads whose latency the compiler must not "optimise": it sinks an ordinary read of the NEXT pass's coefficients to the top of
// that pass (no side effects, used only there), where every pass then waits an LDS round trip.  Issued through inline assembly
// at the top of the current pass and awaited at its end (lds_wait_all), the round trip hides behind the pass's 32 FP64 operations.
typedef double d2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
// (`anchor`: a value the pass's arithmetic starts from, passed through untouched -- it pins the request above that arithmetic, which
// the scheduler is otherwise free to hoist over the request)
__device__ __forceinline__ void lds_read16_async(d2_t& out, unsigned addr, double& anchor) { asm volatile("ds_read_b128 %0, %2" : "=v"(out), "+v"(anchor) : "v"(addr)); }
__device__ __forceinline__ void lds_read8_async(double& out, unsigned addr, double& anchor) { asm volatile("ds_read_b64 %0, %2" : "=v"(out), "+v"(anchor) : "v"(addr)); }
// the wait "defines" the requested values (no use can move above it) and follows `anchor`, the last value the pass computes
__device__ __forceinline__ void lds_wait2(d2_t& a0, d2_t& a1, double& anchor) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(anchor));
}

// one group of GS units (a direction and ONE microphone, or a direction and TWO antipodal microphones j, j' with
// x_dj' = -x_dj): E = sum_{m even} c[m] T_m(x), O = sum_{m odd} c[m] T_m(x) with the Chebyshev recurrence
// T_{m+1} = 2x T_m - T_{m-1} -- ONE fused operation per term and unit for the basis, two for the complex sums (the Legendre
// recurrence costs a multiplication more; synth_coeff_kernel converts the series).  The unit's operands are g(x) = E + O and
// g(-x) = E - O: an antipodal pair of microphones (15 of the em32's 16 pairs are exact) costs what one microphone does.
// Two terms per pass, the two polynomial registers of a unit swap roles (no moves); the coefficients of the next pass are
// requested before this pass's arithmetic into the other of two register sets.
template <int GS>
__device__ __forceinline__ void synth_group(const double (&x2)[GS], cplx (&accE)[GS], cplx (&accO)[GS], const cplx* bs, int nord_pad) {
    double pa[GS], pb[GS];   // T_m (m even), T_m (m odd)
#pragma unroll
    for (int i = 0; i < GS; ++i) { pa[i] = 1.0; pb[i] = 0.5 * x2[i]; accE[i] = mk(0.0, 0.0); accO[i] = mk(0.0, 0.0); }
    const unsigned bs0 = lds_addr(bs);
    d2_t b0, b1, c0, c1;
    auto request = [&](int n, d2_t& q0, d2_t& q1) __attribute__((always_inline)) {
        const int nn = n < nord_pad ? n : nord_pad - 2;   // (the last pass re-reads its own coefficients)
        lds_read16_async(q0, bs0 + 16 * nn, pa[0]); lds_read16_async(q1, bs0 + 16 * nn + 16, pa[0]);
    };
    auto pass = [&](const d2_t& q0, const d2_t& q1) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < GS; ++i) {
            accE[i].x = fma(q0.x, pa[i], accE[i].x); accE[i].y = fma(q0.y, pa[i], accE[i].y);
            accO[i].x = fma(q1.x, pb[i], accO[i].x); accO[i].y = fma(q1.y, pb[i], accO[i].y);
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

}  // namespace
}  // namespace emagls
