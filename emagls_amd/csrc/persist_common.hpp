// Helpers shared by the resident sweep kernels (sweep_persist.hip, sweep_synth.hip): tagged 8-byte granules that cross
// workgroups without fences, the bounded wait on them, the unit-phase step of the recurrence.
#pragma once
#include "kernels.hpp"

namespace emagls {
namespace {

// DPW directions per workgroup; 4 DPW compute threads (p phase: one direction and a quarter of the channels per
// thread) + one communication wave.  One workgroup per CU (register budget), and a design's workgroups share an
// XCD (32 CUs), so nWG = ceil(D / DPW) must not exceed 32: DPW = 64 up to 2048 directions, 96 up to 3072.
constexpr int PS_CMAX = 32;
constexpr unsigned PS_SPIN_LIMIT = 1u << 21;
// a wait for peers gives up after this many ticks of the 100 MHz wall clock (20 ms: a kernel of another batch that holds a CU
// for a millisecond or two is waited out; a peer that cannot become resident is not -- residency is decided before the launch,
// persist_sweep_fits, so this only fires when something else took the CUs in between)
constexpr long long PS_WAIT_TICKS = 2000000;
typedef unsigned long long u64;

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
// HW_REG_XCC_ID (register 20, bits 3:0): the XCD this wave runs on
__device__ __forceinline__ unsigned read_xcc_id() {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);
#else
    return 0;
#endif
}
// element-wise copy: a whole-struct assignment from global memory becomes a memcpy that pins the destination array
// in scratch memory
__device__ __forceinline__ cplx ldc(const cplx* p) { return mk(p->x, p->y); }
__device__ __forceinline__ u64 ll_load(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// local: every reader shares the writer's XCD L2 -> plain store (stays in L2); otherwise sc1 write-through
__device__ __forceinline__ void ll_put(u64* dst, u64 word, bool local) {
    if (local) __hip_atomic_store(dst, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else __hip_atomic_store(dst, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void ll_store(u64* lo, u64* hi, double v, unsigned tag, bool local) {
    const u64 bits = (u64)__double_as_longlong(v), t = (u64)tag << 32;
    ll_put(lo, t | (bits & 0xffffffffull), local);
    ll_put(hi, t | (bits >> 32), local);
}
__device__ __forceinline__ double ll_value(u64 lo, u64 hi) {
    return __longlong_as_double((long long)((lo & 0xffffffffull) | (hi << 32)));
}
__device__ __forceinline__ bool ll_ok(u64 w, unsigned tag) { return (unsigned)(w >> 32) == tag; }

// wave-uniform wait: returns false when the wait was abandoned
template <typename Load> __device__ __forceinline__ bool ll_wait(Load&& load_and_check, int* abort_flag, long long wait_ticks, unsigned* nspins = nullptr, long long* t_first = nullptr) {
    unsigned spins = 0;
    long long t_begin = 0;
    for (;;) {
        const bool ok = load_and_check();
        if (t_first && spins == 0) *t_first = (long long)wall_clock64();
        if (__builtin_amdgcn_ballot_w64(!ok) == 0) { if (nspins) *nspins += spins; return true; }
        __builtin_amdgcn_s_sleep(1);
        ++spins;
        if ((spins & 255u) == 0 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
        bool timed_out = false;
        if ((spins & 255u) == 0) {   // (the wall clock only every 256 polls: ~0.1 ms)
            const long long now = (long long)wall_clock64();
            if (t_begin == 0) t_begin = now; else timed_out = now - t_begin > wait_ticks;
        }
        if (spins >= PS_SPIN_LIMIT || timed_out) {
            __hip_atomic_store(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
}

}  // namespace
}  // namespace emagls
