// LDS-resident FP64 FFT passes shared by the HRIR prologue / filter epilogue (fft.hip) and the fused overlap-save decode
// (decode.hip).
#pragma once
#include "kernels.hpp"

namespace emagls {

__device__ __forceinline__ unsigned bitrev(unsigned v, int bits) { return __brev(v) >> (32 - bits); }

// in-place radix-2 DIT on `nt` transforms laid out buf[t*nfft + i] (input already bit-reversed).
// tws[j] = exp(-2 pi i j/nfft) for j < nfft/2 in LDS.  INVERSE uses conj twiddles (no scaling).
// Two consecutive stages are fused into one pass over groups of four points (radix 2^2: the same multiplications and additions
// in the same order as two radix-2 passes -- bitwise identical -- with half the LDS round trips and barriers); a remaining
// single stage runs as a plain radix-2 pass.
// PAD: element a of the buffer lives at a + (a >> 4).  Unpadded, the passes of the first stages read four-element groups at a
// stride of 64 bytes (16-way bank conflicts, 4x the cycles of a conflict-free 16-byte access) and a bit-reversed load puts a
// whole wave on one bank; one spare element per 16 spreads both over all banks.
template <bool PAD> __device__ __forceinline__ int lds_fft_ix(int a) { return PAD ? a + (a >> 4) : a; }

template <bool INVERSE, bool PAD = false>
__device__ __forceinline__ void lds_fft_stages(cplx* buf, const cplx* tws, int nfft, int log2n, int nt, int s_begin = 0) {
    const int half_n = nfft >> 1, lh = log2n - 1;
    const int total = nt * half_n;
    int s = s_begin;
    for (; s + 1 < log2n; s += 2) {
        const int h = 1 << s;
        const int tstep = nfft >> (s + 1), tstep2 = tstep >> 1;
        const int quarter_n = nfft >> 2, lq = log2n - 2;
        const int total4 = nt * quarter_n;
        // two groups per pass: their LDS reads are issued before the first use
        for (int g0 = threadIdx.x; g0 < total4; g0 += 2 * blockDim.x) {
            cplx a[2], b[2], c[2], d[2], w1[2], w2[2], w3[2];
            int i0[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int g = min(g0 + u * (int)blockDim.x, total4 - 1);   // (clamped: the surplus lanes recompute a group, unstored)
                const int t = g >> lq, gg = g & (quarter_n - 1);
                const int pos = gg & (h - 1);
                i0[u] = (t << log2n) + ((gg >> s) << (s + 2)) + pos;
                w1[u] = tws[pos * tstep];
                w2[u] = tws[pos * tstep2];
                w3[u] = tws[(pos + h) * tstep2];
                a[u] = buf[lds_fft_ix<PAD>(i0[u])];
                b[u] = buf[lds_fft_ix<PAD>(i0[u] + h)];
                c[u] = buf[lds_fft_ix<PAD>(i0[u] + 2 * h)];
                d[u] = buf[lds_fft_ix<PAD>(i0[u] + 3 * h)];
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (g0 + u * (int)blockDim.x < total4) {
                    if (INVERSE) { w1[u].y = -w1[u].y; w2[u].y = -w2[u].y; w3[u].y = -w3[u].y; }
                    // stage s: (a, b) and (c, d) with the twiddle of position pos
                    const cplx bw = b[u] * w1[u], dw = d[u] * w1[u];
                    const cplx p0 = a[u] + bw, p1 = a[u] - bw, p2 = c[u] + dw, p3 = c[u] - dw;
                    // stage s+1: (p0, p2) at position pos, (p1, p3) at position pos + h
                    const cplx q2 = p2 * w2[u], q3 = p3 * w3[u];
                    buf[lds_fft_ix<PAD>(i0[u])] = p0 + q2;
                    buf[lds_fft_ix<PAD>(i0[u] + 2 * h)] = p0 - q2;
                    buf[lds_fft_ix<PAD>(i0[u] + h)] = p1 + q3;
                    buf[lds_fft_ix<PAD>(i0[u] + 3 * h)] = p1 - q3;
                }
            }
        }
        __syncthreads();
    }
    for (; s < log2n; ++s) {
        const int half = 1 << s;
        const int tstep = nfft >> (s + 1);
        // four butterflies per pass: their seven LDS reads are issued before the first use
        for (int b0 = threadIdx.x; b0 < total; b0 += 4 * blockDim.x) {
            cplx a[4], c[4], w[4];
            int i0[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int b = min(b0 + u * (int)blockDim.x, total - 1);   // (clamped: the surplus lanes recompute a butterfly, unstored)
                const int t = b >> lh, bb = b & (half_n - 1);
                const int pos = bb & (half - 1);
                i0[u] = (t << log2n) + ((bb >> s) << (s + 1)) + pos;
                w[u] = tws[pos * tstep];
                a[u] = buf[lds_fft_ix<PAD>(i0[u])];
                c[u] = buf[lds_fft_ix<PAD>(i0[u] + half)];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (b0 + u * (int)blockDim.x < total) {
                    if (INVERSE) w[u].y = -w[u].y;
                    const cplx cw = c[u] * w[u];
                    buf[lds_fft_ix<PAD>(i0[u])] = a[u] + cw;
                    buf[lds_fft_ix<PAD>(i0[u] + half)] = a[u] - cw;
                }
            }
        }
        __syncthreads();
    }
}

}  // namespace emagls
