// One wavefront = one 1024-point FP64 transform, 16 values per lane, three register passes (16 x 8 x 8) with two transpositions
// through a wave-private LDS buffer.  Nothing in it waits for another wave: LDS operations of one wave complete in program order,
// so the passes need no workgroup barrier, and the waves of a workgroup drift apart and hide each other's LDS and memory
// latencies (the radix-2^2 LDS form of lds_fft.hpp and the half-wave form of decode.hip stop all waves at every stage).
//
//   n = 64 n1 + l (lane l holds n1 = 0..15),  k = k1 + 16 (ka + 8 kb):
//   pass 1  lane l:               A[k1][l] = sum_n1 z[64 n1 + l] W_16^(n1 k1), times W_1024^(l k1)          -> row k1, column l
//   pass 2  lane (k1, s = l & 3): l = 8 a + b, b in {s, s + 4}:  B[k1][b][ka] = sum_a A'[k1][8 a + b] W_8^(a ka), times W_64^(b ka)
//   pass 3  lane (k1, s):         ka in {s, s + 4}:  Z[k1 + 16 (ka + 8 kb)] = sum_b B'[k1][b][ka] W_8^(b kb)
// Output: v[8 e + j] = Z[wf_bin(l, 8 e + j)],  wf_bin = (l >> 2) + 16 ((l & 3) + 4 e) + 128 bitrev3(j).
//
// LDS layouts in 16-byte slots (lane groups of ds_write_b128 = 8 contiguous lanes on 8 slots, of ds_read_b128 = 16 lanes on 16
// slots; every access below is conflict-free, checked by tools/experiments/lds_layout_check.py):
//   transposition 1   slot = k1 * 68 + 8 a + ((b + a) & 7),  l = 8 a + b
//   transposition 2   slot = k1 * 68 + 8 ka + ((b + ka) & 7)      (in place: a lane overwrites the diagonal it read)
//   natural order     slot = wf_slot(k) = (k & ~7) | ((k + 2 ((k >> 4) & 3)) & 7)      (wave_fft_store_natural)
#pragma once
#include "reg_fft.hpp"

namespace emagls {

constexpr int WF_N = 1024;
constexpr int WF_ROW = 68;
constexpr int WF_BUF = 16 * WF_ROW;      // slots of one wave's buffer (17 408 bytes)
constexpr int WF_TW1 = 1024;             // tw1[i * 64 + l] = W_1024^(l * bitrev4(i))
constexpr int WF_TW2 = 64;               // tw2[(8 d + j) * 4 + s] = W_64^((s + 4 d) * bitrev3(j))
constexpr int WF_TABLES = WF_TW1 + WF_TW2;

__host__ __device__ constexpr int wf_bin(int l, int i) { return (l >> 2) + 16 * ((l & 3) + 4 * (i >> 3)) + 128 * rf_bitrev<3>(i & 7); }
__host__ __device__ constexpr int wf_slot(int k) { return (k & ~7) | ((k + 2 * ((k >> 4) & 3)) & 7); }

// the two twiddle tables from the full-circle table circle[m] = exp(-2 pi i m / 1024) (launch_twiddles at nfft = 1024)
__device__ __forceinline__ void wave_fft_tables(cplx* tables, const cplx* __restrict__ circle, int tid, int nthreads) {
    for (int idx = tid; idx < WF_TW1; idx += nthreads) tables[idx] = circle[(idx & 63) * rf_bitrev<4>(idx >> 6)];
    for (int idx = tid; idx < WF_TW2; idx += nthreads) {
        const int d = idx >> 5, j = (idx >> 2) & 7, s = idx & 3;
        tables[WF_TW1 + idx] = circle[16 * ((s + 4 * d) * rf_bitrev<3>(j))];
    }
}

// LDS operations of one wave are executed in order; this only keeps the compiler from moving them across the phase boundary
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// length-16 transform whose inputs 8..15 are zero (a signal zero-padded to twice its length): the first stage is a copy
__device__ __forceinline__ void reg_fft16_upper_zero(cplx (&v)[16]) {
    constexpr double C16[8] = {1.0, 0.9238795325112867, 0.7071067811865476, 0.38268343236508984, 0.0, -0.3826834323650897,
                               -0.7071067811865475, -0.9238795325112867};
    constexpr double S16[8] = {0.0, -0.3826834323650898, -0.7071067811865475, -0.9238795325112867, -1.0, -0.9238795325112867,
                               -0.7071067811865476, -0.3826834323650899};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const cplx a = v[i];
        if (i == 0) v[8] = a;
        else if (i == 4) v[12] = mk(a.y, -a.x);
        else v[8 + i] = mk(a.x * C16[i] - a.y * S16[i], a.x * S16[i] + a.y * C16[i]);
    }
    cplx lo[8], hi[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { lo[i] = v[i]; hi[i] = v[8 + i]; }
    reg_fft<8>(lo);
    reg_fft<8>(hi);
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = lo[i]; v[8 + i] = hi[i]; }
}

// The three passes, separately for callers that put other work between them (decode.hip requests the next channel pair after
// pass 1, when the 16 input registers are free, and consumes pass 3 one half at a time).
//   wf_pass1        v[n1] = z[64 n1 + l] in; rows A'[k1][.] in the buffer afterwards
//   wf_pass2_half   b = s + 4 H: the diagonal {8 a + ((b + a) & 7)} of row k1 is replaced IN PLACE by B'[k1][b][ka] (a -> ka)
//   wf_pass3_half   ka = s + 4 E: out[j] = Z[k1 + 16 ka + 128 bitrev3(j)]
template <bool UPPER_ZERO>
__device__ __forceinline__ void wf_pass1(cplx (&v)[16], cplx* tb, const cplx* tables, int l) {
    if (UPPER_ZERO) reg_fft16_upper_zero(v); else reg_fft<16>(v);
    const int lsw = (l & ~7) | ((l + (l >> 3)) & 7);
    // (four twiddles at a time: sixteen table reads in flight cost 64 registers the callers' accumulators need)
#pragma unroll
    for (int i0 = 0; i0 < 16; i0 += 4) {
        cplx t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = tables[(i0 + i) * 64 + l];
#pragma unroll
        for (int i = 0; i < 4; ++i) tb[rf_bitrev<4>(i0 + i) * WF_ROW + lsw] = v[i0 + i] * t[i];
        __builtin_amdgcn_sched_barrier(0);
    }
    wave_lds_fence();
}
template <int H>
__device__ __forceinline__ void wf_pass2_half(cplx* tb, const cplx* tables, int l) {
    const int b = (l & 3) + 4 * H;
    cplx* row = tb + (l >> 2) * WF_ROW;
    const cplx* tw2 = tables + WF_TW1 + 32 * H + (l & 3);
    cplx u[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) u[a] = row[8 * a + ((b + a) & 7)];
    reg_fft<8>(u);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ka = rf_bitrev<3>(j);
        row[8 * ka + ((b + ka) & 7)] = (j == 0) ? u[j] : u[j] * tw2[4 * j];
    }
}
template <int E>
__device__ __forceinline__ void wf_pass3_half(cplx (&out)[8], const cplx* tb, int l) {
    const int ka = (l & 3) + 4 * E;
    const cplx* rowk = tb + (l >> 2) * WF_ROW + 8 * ka;
#pragma unroll
    for (int b = 0; b < 8; ++b) out[b] = rowk[(b + ka) & 7];
    reg_fft<8>(out);
}

// v[n1] = z[64 n1 + l] in, v[i] = Z[wf_bin(l, i)] out; tb = this wave's WF_BUF slots, tables = wave_fft_tables
template <bool UPPER_ZERO>
__device__ __forceinline__ void wave_fft1024(cplx (&v)[16], cplx* tb, const cplx* tables, int l) {
    wf_pass1<UPPER_ZERO>(v, tb, tables, l);
    wf_pass2_half<0>(tb, tables, l);
    wf_pass2_half<1>(tb, tables, l);
    wave_lds_fence();
    cplx u0[8], u1[8];
    wf_pass3_half<0>(u0, tb, l);
    wf_pass3_half<1>(u1, tb, l);
    wave_lds_fence();
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[j] = u0[j]; v[8 + j] = u1[j]; }
}

// the spectrum into the wave's buffer in natural order (slot wf_slot(k)); the caller fences before reading other lanes' bins
__device__ __forceinline__ void wave_fft_store_natural(const cplx (&v)[16], cplx* tb, int l) {
    const int base = (l >> 2) + 16 * (l & 3);
#pragma unroll
    for (int i = 0; i < 16; ++i) tb[wf_slot(base + 64 * (i >> 3) + 128 * rf_bitrev<3>(i & 7))] = v[i];
}

}  // namespace emagls
