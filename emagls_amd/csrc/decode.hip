// binauralDecode core loop (dependencies/binauralDecode.m:33-42): out(:,ear) = sum_c fftfilt(w_ear(:,c), in(:,c)).
// Overlap-save with hipFFT: the C channel spectra of a block are multiplied with the filter spectra and
// accumulated in the frequency domain, so each ear needs ONE inverse transform per block instead of C.
#include <hipfft/hipfft.h>

#include <mutex>
#include <type_traits>

#include "kernels.hpp"
#include "lds_fft.hpp"
#include "reg_fft.hpp"
#include "wave_fft.hpp"

namespace emagls {

static void fft_check(hipfftResult r, const char* what) {
    if (r != HIPFFT_SUCCESS) {
        char buf[256];
        snprintf(buf, sizeof buf, "hipFFT error %d in %s", (int)r, what);
        throw Error(3, buf);
    }
}

// seg[c][b][i] = in[c*n + b*B - (len-1) + i]   (zero outside [0, n))
__global__ void ols_pack_kernel(const double* __restrict__ in, int64_t n, int C, int64_t nblocks, int Nf, int64_t B,
                                int64_t len, double* __restrict__ seg) {
    const int64_t total = (int64_t)C * nblocks * Nf;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = idx % Nf, cb = idx / Nf, b = cb % nblocks, c = cb / nblocks;
        const int64_t src = b * B - (len - 1) + i;
        seg[idx] = (src >= 0 && src < n) ? in[c * n + src] : 0.0;
    }
}
// wpad[e][c][i] = w_e[c*len + i] for i < len else 0
__global__ void ols_padfilt_kernel(const double* __restrict__ wL, const double* __restrict__ wR, int C, int64_t len, int Nf,
                                   double* __restrict__ wpad) {
    const int64_t total = (int64_t)2 * C * Nf;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = idx % Nf, ec = idx / Nf, c = ec % C, e = ec / C;
        const double* w = e ? wR : wL;
        wpad[idx] = (i < len) ? w[c * len + i] : 0.0;
    }
}
// Yf[e][b][k] = sum_c Xf[c][b][k] Wf[e][c][k]
__global__ void ols_mac_kernel(const cplx* __restrict__ Xf, const cplx* __restrict__ Wf, int C, int64_t nblocks, int Pf,
                               cplx* __restrict__ Yf) {
    const int64_t total = (int64_t)2 * nblocks * Pf;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = idx % Pf, eb = idx / Pf, b = eb % nblocks, e = eb / nblocks;
        cplx acc = mk(0, 0);
        for (int c = 0; c < C; ++c) cfma(acc, Xf[((int64_t)c * nblocks + b) * Pf + k], Wf[((int64_t)e * C + c) * Pf + k]);
        Yf[idx] = acc;
    }
}
// out[e*n + b*B + i] = y[e][b][len-1+i] / Nf
__global__ void ols_unpack_kernel(const double* __restrict__ y, int64_t n, int64_t nblocks, int Nf, int64_t B, int64_t len,
                                  double* __restrict__ out) {
    const int64_t total = (int64_t)2 * n;
    const double scale = 1.0 / (double)Nf;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = idx % n, e = idx / n;
        const int64_t b = t / B, i = t % B;
        out[idx] = y[((int64_t)e * nblocks + b) * Nf + (len - 1) + i] * scale;
    }
}

// ---------------------------------------------------------------------------------------------
// Fused overlap-save block (filters up to 2048 taps): one workgroup = one output block.  The hipFFT passes above write the
// segments, their spectra and the products to HBM and read them back (~6x the algorithmic bytes); here a block's segments go
// from the signal straight into LDS -- two real channels packed into one complex transform, `nt` transforms at a time --
// the spectra are unpacked, multiplied with the filter spectra and accumulated in registers (thread = frequency bin), and
// both ears leave through ONE packed inverse transform (y_L + i y_R).  HBM sees the signal (each sample in two
// segments: the second read is an L2 hit), the filter spectra (L2 resident) and the output.
//   sig [C][n], Wf [2][C][Pf] (spectra of the zero-padded filters, hipFFT D2Z), out [2][n]
// ---------------------------------------------------------------------------------------------
constexpr int OLSF_NT = 512;   // threads
// KU: frequency bins per thread (Pf <= KU * 512); NTP: transforms (channel pairs) per round -- (2, 4) up to Nf = 1024,
// (3, 2) for 2048, (5, 1) for 4096: NTP * Nf <= 4096 elements, i.e. 8 loads per thread and round
template <int KU, int NTP>
__global__ void __launch_bounds__(OLSF_NT) ols_fused_kernel(const double* __restrict__ sig, int64_t n, int C, const cplx* __restrict__ Wf,
                                                            int64_t len, int Nf, int log2n, int64_t B, double* __restrict__ out) {
    constexpr int NLD = 8;         // loads per thread and round
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    cplx* buf = reinterpret_cast<cplx*>(dyn);          // [NTP][Nf], padded (lds_fft_ix)
    cplx* tws = buf + (size_t)NTP * (Nf + Nf / 16);     // [Nf / 2]
    const int tid = threadIdx.x;
    const int64_t blk = blockIdx.x;
    const int Pf = Nf / 2 + 1, mask = Nf - 1;
    for (int j = tid; j < Nf / 2; j += OLSF_NT) {
        double sn, cs;
        sincospi(-2.0 * (double)j / (double)Nf, &sn, &cs);   // (exact at the multiples of 1/4: Nf is a power of two)
        tws[j] = mk(cs, sn);
    }
    cplx accL[KU], accR[KU];
#pragma unroll
    for (int u = 0; u < KU; ++u) { accL[u] = mk(0, 0); accR[u] = mk(0, 0); }
    const int npairs = (C + 1) / 2;
    const int64_t s0 = blk * B - (len - 1);
    // a round's samples travel global -> registers -> LDS; the loads of round r + 1 are issued before the transforms of round r
    // (all of a thread's loads back to back: one latency per round instead of one per element)
    double xa[NLD], xb[NLD];
    auto fetch = [&](int p0) __attribute__((always_inline)) {
        const int np = min(NTP, npairs - p0);
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int idx = tid + OLSF_NT * j;
            const int t = idx >> log2n, i = idx & mask;
            const int ca = 2 * (p0 + t), cb = ca + 1;
            const int64_t src = s0 + i;
            const bool in = idx < np * Nf && src >= 0 && src < n;
            xa[j] = in ? sig[(int64_t)ca * n + src] : 0.0;
            xb[j] = (in && cb < C) ? sig[(int64_t)cb * n + src] : 0.0;
        }
    };
    fetch(0);
    for (int p0 = 0; p0 < npairs; p0 += NTP) {
        const int np = min(NTP, npairs - p0);
        __syncthreads();   // the previous round's spectra have been read (first round: the twiddles are written)
#pragma unroll
        for (int j = 0; j < NLD; ++j) {
            const int idx = tid + OLSF_NT * j;
            if (idx < np * Nf) {
                const int t = idx >> log2n, i = idx & mask;
                buf[lds_fft_ix<true>((t << log2n) + (int)bitrev((unsigned)i, log2n))] = mk(xa[j], xb[j]);
            }
        }
        if (p0 + NTP < npairs) fetch(p0 + NTP);
        __syncthreads();
        lds_fft_stages<false, true>(buf, tws, Nf, log2n, np);
        // Z = X_a + i X_b with real x_a, x_b:  X_a[k] = (Z[k] + conj(Z[N-k])) / 2,  X_b[k] = (Z[k] - conj(Z[N-k])) / (2i)
        // (the filter spectra of the round are requested together, before the first use)
        cplx wla[KU][NTP], wra[KU][NTP], wlb[KU][NTP], wrb[KU][NTP];
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int k = min(tid + OLSF_NT * u, Pf - 1);
#pragma unroll
            for (int t = 0; t < NTP; ++t) {
                const int ca = min(2 * (p0 + t), C - 1), cb = min(ca + 1, C - 1);
                wla[u][t] = Wf[(int64_t)ca * Pf + k];
                wra[u][t] = Wf[((int64_t)C + ca) * Pf + k];
                wlb[u][t] = Wf[(int64_t)cb * Pf + k];
                wrb[u][t] = Wf[((int64_t)C + cb) * Pf + k];
            }
        }
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int k = tid + OLSF_NT * u;
            if (k < Pf) {
#pragma unroll
                for (int t = 0; t < NTP; ++t) {
                    if (t < np) {
                        const cplx z = buf[lds_fft_ix<true>((t << log2n) + k)], zr = conj(buf[lds_fft_ix<true>((t << log2n) + ((Nf - k) & mask))]);
                        const cplx pa = mk(0.5 * (z.x + zr.x), 0.5 * (z.y + zr.y));
                        const cplx pb = mk(0.5 * (z.y - zr.y), -0.5 * (z.x - zr.x));
                        cfma(accL[u], pa, wla[u][t]);
                        cfma(accR[u], pa, wra[u][t]);
                        if (2 * (p0 + t) + 1 < C) {
                            cfma(accL[u], pb, wlb[u][t]);
                            cfma(accR[u], pb, wrb[u][t]);
                        }
                    }
                }
            }
        }
    }
    __syncthreads();
    // packed inverse: Yc = Y_L + i Y_R on all Nf bins (Y_e[N-k] = conj(Y_e[k]): the outputs are real)
#pragma unroll
    for (int u = 0; u < KU; ++u) {
        const int k = tid + OLSF_NT * u;
        if (k < Pf) {
            const cplx l = accL[u], r = accR[u];
            buf[lds_fft_ix<true>((int)bitrev((unsigned)k, log2n))] = mk(l.x - r.y, l.y + r.x);
            if (k > 0 && k < Nf / 2) buf[lds_fft_ix<true>((int)bitrev((unsigned)(Nf - k), log2n))] = mk(l.x + r.y, r.x - l.y);
        }
    }
    __syncthreads();
    lds_fft_stages<true, true>(buf, tws, Nf, log2n, 1);
    const double scale = 1.0 / (double)Nf;
    for (int64_t i = tid; i < B; i += OLSF_NT) {
        const int64_t t = blk * B + i;
        if (t < n) {
            const cplx y = buf[lds_fft_ix<true>((int)((len - 1) + i))];
            stream_store(out + t, y.x * scale);
            stream_store(out + n + t, y.y * scale);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same block on register-resident transforms (reg_fft.hpp) for Nf = N1 * N2 = 256 (16 x 16), 512 (16 x 32), 1024 (32 x 32):
// 32 threads per transform, 8 transforms (channel pairs) per round.
//   stage A  lane n2 (< N2): a[n1] = z[N2 n1 + n2], length-N1 transform in registers, times W_N^(n2 k1), into LDS row k1
//   stage B  lane k1 (< N1): the row [k1][n2], length-N2 transform in registers: Z[k1 + N1 k2], back into LDS in natural order
//   then thread = frequency bin as above (unpack the two real channels, multiply, accumulate), and the packed inverse
//   transform of Y_L + i Y_R as conj(FFT(conj(.))) by the first 32 threads.
// ---------------------------------------------------------------------------------------------
// OLSR_NT threads, OLSR_TP = OLSR_NT / 32 transforms per round: (256, 8) -- one block per CU (143 KB of LDS at Nf = 1024), the
// shortest path through a block, for signals of a few hundred blocks; (128, 4) -- 76 KB, two blocks per CU hide each other's
// latencies: 10 % more throughput on long signals (1.24 against 1.37 ms for 100 s x 25 channels), 30 % slower on 2.5 s
template <int N1, int N2, int OLSR_NT>
__global__ void __launch_bounds__(OLSR_NT) ols_fused_rr_kernel(const double* __restrict__ sig, int64_t n, int C, const cplx* __restrict__ Wf,
                                                               int64_t len, int64_t B, double* __restrict__ out) {
    constexpr int OLSR_TP = OLSR_NT / 32;
    constexpr int Nf = N1 * N2, Pf = Nf / 2 + 1, L1 = rf_log2<N1>(), L2 = rf_log2<N2>();
    constexpr int ROW = N2 + 1;                   // padded row of the transposition buffer (lane k1 reads row k1: stride != 0 mod 64 dwords)
    constexpr int TSZ = (N1 * ROW > Nf ? N1 * ROW : Nf);   // elements per transform buffer
    constexpr int KU = (Pf + OLSR_NT - 1) / OLSR_NT;
    static_assert(N1 <= N2 && N2 <= 32, "factor sizes");
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    cplx* buf = reinterpret_cast<cplx*>(dyn);     // [OLSR_TP][TSZ]
    cplx* tws = buf + (size_t)OLSR_TP * TSZ;       // [Nf / 2]  exp(-2 pi i j / Nf)
    const int tid = threadIdx.x, tr = tid >> 5, l = tid & 31;
    const int64_t blk = blockIdx.x;
    for (int j = tid; j < Nf / 2; j += OLSR_NT) {
        double sn, cs;
        sincospi(-2.0 * (double)j / (double)Nf, &sn, &cs);
        tws[j] = mk(cs, sn);
    }
    auto twiddle = [&](int m) __attribute__((always_inline)) {   // W_Nf^m, 0 <= m < Nf
        const cplx w = tws[m & (Nf / 2 - 1)];
        return (m & (Nf / 2)) ? mk(-w.x, -w.y) : w;
    };
    // one forward transform of the 32-thread group's buffer `tb`; input in registers a[n1] = z[N2 n1 + lane] (lanes < N2)
    auto transform = [&](cplx (&a)[N1], cplx* tb) __attribute__((always_inline)) {
        if (l < N2) {
            reg_fft<N1>(a);
#pragma unroll
            for (int i = 0; i < N1; ++i) {
                const int k1 = rf_bitrev<L1>(i);
                tb[k1 * ROW + l] = a[i] * twiddle(l * k1);
            }
        }
        __syncthreads();
        cplx b[N2];
        if (l < N1) {
#pragma unroll
            for (int n2 = 0; n2 < N2; ++n2) b[n2] = tb[l * ROW + n2];
        }
        __syncthreads();   // (every row has been read: the natural-order result may overwrite the buffer)
        if (l < N1) {
            reg_fft<N2>(b);
#pragma unroll
            for (int i = 0; i < N2; ++i) tb[l + N1 * rf_bitrev<L2>(i)] = b[i];
        }
        __syncthreads();
    };
    cplx accL[KU], accR[KU];
#pragma unroll
    for (int u = 0; u < KU; ++u) { accL[u] = mk(0, 0); accR[u] = mk(0, 0); }
    const int npairs = (C + 1) / 2;
    const int64_t s0 = blk * B - (len - 1);
    cplx* tb = buf + (size_t)tr * TSZ;
    __syncthreads();   // the twiddles are written
    // a round's samples: global -> registers, requested one round ahead (before the previous round's multiply-accumulate)
    cplx a[N1];
    auto fetch = [&](int p0) __attribute__((always_inline)) {
        const int np = min(OLSR_TP, npairs - p0);
        const int ca = min(2 * (p0 + tr), C - 1), cb = 2 * (p0 + tr) + 1;
#pragma unroll
        for (int n1 = 0; n1 < N1; ++n1) {
            const int64_t src = s0 + N2 * n1 + l;
            const bool in = tr < np && l < N2 && src >= 0 && src < n;
            a[n1] = mk(in ? sig[(int64_t)ca * n + src] : 0.0, (in && cb < C) ? sig[(int64_t)cb * n + src] : 0.0);
        }
    };
    fetch(0);
    for (int p0 = 0; p0 < npairs; p0 += OLSR_TP) {
        const int np = min(OLSR_TP, npairs - p0);
        transform(a, tb);
        if (p0 + OLSR_TP < npairs) fetch(p0 + OLSR_TP);
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int k = tid + OLSR_NT * u;
            if (k < Pf) {
                // all 32 filter values of the round are requested before the first use (a loop that loads and uses one transform
                // at a time pays an L2 round trip per transform: 48 us per block, four times the transforms themselves)
                cplx wl[2 * OLSR_TP], wr[2 * OLSR_TP];
#pragma unroll
                for (int t = 0; t < OLSR_TP; ++t) {
                    const int c0 = min(2 * (p0 + t), C - 1), c1 = min(c0 + 1, C - 1);
                    wl[2 * t] = Wf[(int64_t)c0 * Pf + k];
                    wr[2 * t] = Wf[((int64_t)C + c0) * Pf + k];
                    wl[2 * t + 1] = Wf[(int64_t)c1 * Pf + k];
                    wr[2 * t + 1] = Wf[((int64_t)C + c1) * Pf + k];
                }
#pragma unroll
                for (int t = 0; t < OLSR_TP; ++t) {
                    if (t < np) {
                        const cplx* x = buf + (size_t)t * TSZ;
                        const cplx z = x[k], zr = conj(x[(Nf - k) & (Nf - 1)]);
                        const cplx pa = mk(0.5 * (z.x + zr.x), 0.5 * (z.y + zr.y));
                        const cplx pb = mk(0.5 * (z.y - zr.y), -0.5 * (z.x - zr.x));
                        cfma(accL[u], pa, wl[2 * t]);
                        cfma(accR[u], pa, wr[2 * t]);
                        if (2 * (p0 + t) + 1 < C) {
                            cfma(accL[u], pb, wl[2 * t + 1]);
                            cfma(accR[u], pb, wr[2 * t + 1]);
                        }
                    }
                }
            }
        }
        __syncthreads();   // the spectra have been read: the next round may overwrite the buffers
    }
    // packed inverse: y_L + i y_R = IFFT(Y_L + i Y_R) = conj(FFT(conj(Y_L + i Y_R))) / Nf; transform 0's buffer, natural order in
#pragma unroll
    for (int u = 0; u < KU; ++u) {
        const int k = tid + OLSR_NT * u;
        if (k < Pf) {
            const cplx lft = accL[u], r = accR[u];
            buf[k] = mk(lft.x - r.y, -(lft.y + r.x));                                  // conj(Y_L + i Y_R)
            if (k > 0 && k < Nf / 2) buf[Nf - k] = mk(lft.x + r.y, -(r.x - lft.y));     // conj of the mirrored bin
        }
    }
    __syncthreads();
    {
        cplx a[N1];
#pragma unroll
        for (int n1 = 0; n1 < N1; ++n1) a[n1] = (tr == 0 && l < N2) ? buf[N2 * n1 + l] : mk(0, 0);
        __syncthreads();   // (read before stage A of transform 0 rewrites the buffer)
        transform(a, tb);
    }
    const double scale = 1.0 / (double)Nf;
    for (int64_t i = tid; i < B; i += OLSR_NT) {
        const int64_t t = blk * B + i;
        if (t < n) {
            const cplx y = buf[(len - 1) + i];
            stream_store(out + t, y.x * scale);
            stream_store(out + n + t, -y.y * scale);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The block at Nf = 1024 (257..512 taps) on wave-private transforms (wave_fft.hpp).  One wave carries a 1024-point transform in
// 16 registers per lane and keeps its part of the output spectrum in registers as well, so nothing but the final sum leaves it:
// with z_p = FFT(x_2p + i x_2p+1) and zc_p[k] = conj(z_p[N - k]),
//     Y_L[k] + i Y_R[k] = sum_p z_p[k] A_p[k] + zc_p[k] B_p[k],   A_p = U_L + i U_R,  B_p = V_L + i V_R,
//     U_e = (W_e,2p - i W_e,2p+1) / 2,  V_e = (W_e,2p + i W_e,2p+1) / 2,
// and the mirrored term is conj(Q[N - k]) with Q[m] = sum_p z_p[m] conj(B_p[N - m]): both R = sum z A and Q accumulate at the bins a
// lane already holds (no unpacking of the channel pair, no pass through LDS per pair); one mirrored exchange per block, then the
// packed inverse transform of both ears as conj(FFT(conj(.))).
//   tab [npairs][16][2][64]  A_p and conj(B_p[N - .]) at bin wf_bin(lane, i)   (ols_wave_tables_kernel)
// G waves share a block (pairs g, g + G, ...; their partial sums meet in LDS): G = 1 for long signals -- every wave its own
// block, no workgroup barrier at all -- G = 2, 4, 8 as the blocks get fewer, so that every CU still has a workgroup.
// ---------------------------------------------------------------------------------------------
constexpr int OLSW_WAVES = 8;
// pair_mode 0: pair p = channels (2 p, 2 p + 1); 1: (p, p + C / 2) -- the real and the imaginary plane of complex channel p
__global__ void ols_wave_tables_kernel(const cplx* __restrict__ Wf, int C, int pair_mode, cplx* __restrict__ tab) {
    constexpr int N = WF_N, Pf = N / 2 + 1;
    const int npairs = (C + 1) / 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= npairs * 16 * 2 * 64) return;
    const int l = idx & 63, t = (idx >> 6) & 1, i = (idx >> 7) & 15, p = idx >> 11;
    const int m = wf_bin(l, i);
    const int k = t ? ((N - m) & (N - 1)) : m;
    auto W = [&](int e, int c) {
        if (c >= C) return mk(0.0, 0.0);
        const cplx* w = Wf + ((int64_t)e * C + c) * Pf;
        return k <= N / 2 ? w[k] : conj(w[N - k]);
    };
    const int ca = pair_mode ? p : 2 * p, cb = pair_mode ? p + C / 2 : 2 * p + 1;
    const cplx aL = W(0, ca), bL = W(0, cb), aR = W(1, ca), bR = W(1, cb);
    cplx r;
    if (t == 0) {   // A = U_L + i U_R,  U = (a - i b) / 2
        const cplx uL = mk(0.5 * (aL.x + bL.y), 0.5 * (aL.y - bL.x)), uR = mk(0.5 * (aR.x + bR.y), 0.5 * (aR.y - bR.x));
        r = mk(uL.x - uR.y, uL.y + uR.x);
    } else {        // conj(B),  B = V_L + i V_R,  V = (a + i b) / 2
        const cplx vL = mk(0.5 * (aL.x - bL.y), 0.5 * (aL.y + bL.x)), vR = mk(0.5 * (aR.x - bR.y), 0.5 * (aR.y + bR.x));
        r = mk(vL.x - vR.y, -(vL.y + vR.x));
    }
    tab[idx] = r;
}

// The same tables straight from the filter taps (no zero-padding pass, no hipFFT, no spectra in HBM): workgroup = channel pair, wave =
// ear, one packed transform Z_e = FFT(w_e,a + i w_e,b) of the zero-padded taps each.  With it U_e[k] = conj(Z_e[N - k]) / 2 and
// V_e[k] = Z_e[k] / 2, so both table entries of bin m come from c_e = conj(Z_e[N - m]):  A_p[m] = (c_L + i c_R) / 2,
// conj(B_p[N - m]) = (c_L - i c_R) / 2.
//   wL, wR [C][len] taps (len <= 512)
__global__ void __launch_bounds__(128) ols_wave_filter_kernel(const double* __restrict__ wL, const double* __restrict__ wR, int C, int64_t len,
                                                              int pair_mode, const cplx* __restrict__ circle, cplx* __restrict__ tab) {
    constexpr int N = WF_N;
    __shared__ __attribute__((aligned(16))) cplx lds[WF_TABLES + 2 * WF_BUF];
    cplx* tables = lds;
    const int tid = threadIdx.x, e = tid >> 6, l = tid & 63, p = blockIdx.x;
    cplx* tb = lds + WF_TABLES + (size_t)e * WF_BUF;
    wave_fft_tables(tables, circle, tid, 128);
    __syncthreads();
    const int ca = pair_mode ? p : 2 * p, cb = pair_mode ? p + C / 2 : 2 * p + 1;
    const double* w = e ? wR : wL;
    cplx v[16];
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) {
        const int64_t t = 64 * n1 + l;
        v[n1] = (n1 < 8 && t < len) ? mk(w[(int64_t)ca * len + t], cb < C ? w[(int64_t)cb * len + t] : 0.0) : mk(0.0, 0.0);
    }
    wave_fft1024<true>(v, tb, tables, l);
    wave_fft_store_natural(v, tb, l);
    __syncthreads();
    const cplx* zL = lds + WF_TABLES;
    const cplx* zR = zL + WF_BUF;
    cplx* out = tab + (size_t)p * (16 * 2 * 64);
    for (int idx = tid; idx < 16 * 64; idx += 128) {
        const int i = idx >> 6, ll = idx & 63;
        const int m = wf_bin(ll, i), km = wf_slot((N - m) & (N - 1));
        const cplx cl = conj(zL[km]), cr = conj(zR[km]);
        out[(2 * i) * 64 + ll] = mk(0.5 * (cl.x - cr.y), 0.5 * (cl.y + cr.x));       // (c_L + i c_R) / 2
        out[(2 * i + 1) * 64 + ll] = mk(0.5 * (cl.x + cr.y), 0.5 * (cl.y - cr.x));   // (c_L - i c_R) / 2
    }
}

template <int G>
__global__ void __launch_bounds__(64 * OLSW_WAVES) ols_wave_kernel(const double* __restrict__ sig, const cplx* __restrict__ sigc, int64_t n, int C, const cplx* __restrict__ tab,
                                                                   const cplx* __restrict__ circle, int64_t len, int64_t B, int64_t nblocks,
                                                                   double* __restrict__ out) {
    constexpr int N = WF_N;
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    cplx* tables = reinterpret_cast<cplx*>(dyn);
    cplx* bufs = tables + WF_TABLES;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const int g = w % G;                                   // this wave's place among the waves of its block
    const int64_t blk = (int64_t)blockIdx.x * (OLSW_WAVES / G) + w / G;
    wave_fft_tables(tables, circle, tid, 64 * OLSW_WAVES);
    __syncthreads();
    cplx* tb = bufs + (size_t)w * WF_BUF;
    const bool live = blk < nblocks;
    const int npairs = (C + 1) / 2;
    const int64_t s0 = blk * B - (len - 1);
    const bool interior = s0 >= 0 && s0 + N <= n;
    cplx R[16], Q[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { R[i] = mk(0, 0); Q[i] = mk(0, 0); }
    // the samples of channel pair p: x_2p + i x_2p+1 at n = 64 n1 + l
    auto fetch = [&](cplx (&v)[16], int p) __attribute__((always_inline)) {
        if (sigc) {   // complex channel p as it lies in memory: its planes are the pair
            const cplx* xc = sigc + (int64_t)p * n;
            if (interior) {
#pragma unroll
                for (int n1 = 0; n1 < 16; ++n1) v[n1] = xc[s0 + 64 * n1 + l];
            } else {
#pragma unroll
                for (int n1 = 0; n1 < 16; ++n1) {
                    const int64_t src = s0 + 64 * n1 + l;
                    const int64_t at = min(max(src, (int64_t)0), n - 1);
                    const cplx x = xc[at];
                    v[n1] = (src == at) ? x : mk(0.0, 0.0);
                }
            }
            return;
        }
        const double* xa = sig + (int64_t)(2 * p) * n;
        const double* xb = sig + (int64_t)min(2 * p + 1, C - 1) * n;
        if (interior) {
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) v[n1] = mk(xa[s0 + 64 * n1 + l], xb[s0 + 64 * n1 + l]);
        } else {   // the first and the last blocks: zero outside the signal (loads from a clamped address, then a select)
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                const int64_t src = s0 + 64 * n1 + l;
                const int64_t at = min(max(src, (int64_t)0), n - 1);
                const double a = xa[at], b = xb[at];
                v[n1] = (src == at) ? mk(a, b) : mk(0.0, 0.0);
            }
        }
    };
    // (requesting the next pair's samples while this one is transformed would hide the 2 us the loads take -- a third of the
    // kernel's time -- but a wave has no registers for them: 128 hold the sums, 64 the transform, and the compiler spills 500+ bytes
    // per lane with 64 more in flight; pass 3 is consumed one half at a time for the same reason)
    if (live) {
        for (int p = g; p < npairs; p += G) {
            cplx v[16];
            fetch(v, p);
            if (!sigc && 2 * p + 1 >= C) {
#pragma unroll
                for (int n1 = 0; n1 < 16; ++n1) v[n1].y = 0.0;
            }
            wf_pass1<false>(v, tb, tables, l);
            wf_pass2_half<0>(tb, tables, l);
            wf_pass2_half<1>(tb, tables, l);
            wave_lds_fence();
            const cplx* f = tab + (size_t)p * (16 * 2 * 64) + l;
            auto half = [&](auto e_tag) __attribute__((always_inline)) {
                constexpr int E = decltype(e_tag)::value;
                cplx z[8];
                wf_pass3_half<E>(z, tb, l);
#pragma unroll
                for (int j0 = 0; j0 < 8; j0 += 4) {   // (eight filter values in flight)
                    cplx fa[4], fb[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { fa[j] = f[(2 * (8 * E + j0 + j)) * 64]; fb[j] = f[(2 * (8 * E + j0 + j) + 1) * 64]; }
#pragma unroll
                    for (int j = 0; j < 4; ++j) { cfma(R[8 * E + j0 + j], z[j0 + j], fa[j]); cfma(Q[8 * E + j0 + j], z[j0 + j], fb[j]); }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            half(std::integral_constant<int, 0>{});
            half(std::integral_constant<int, 1>{});
            wave_lds_fence();
        }
    }
    if (G > 1) {   // the partial sums of the block's waves: everyone parks R, the first wave adds; then Q the same way
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            cplx (&acc)[16] = pass ? Q : R;
            __syncthreads();
            if (g != 0) {
#pragma unroll
                for (int i = 0; i < 16; ++i) tb[i * 64 + l] = acc[i];
            }
            __syncthreads();
            if (g == 0) {
                for (int o = 1; o < G; ++o) {
                    const cplx* ob = tb + (size_t)o * WF_BUF;
#pragma unroll
                    for (int i = 0; i < 16; ++i) { const cplx x = ob[i * 64 + l]; acc[i].x += x.x; acc[i].y += x.y; }
                }
            }
        }
        __syncthreads();
    }
    if (!live || g != 0) return;
    // conj(P[k]) = conj(R[k]) + Q[N - k] at k = 64 n1 + l: Q through the buffer in natural order, read mirrored; then R
    cplx v[16];
    wave_fft_store_natural(Q, tb, l);
    wave_lds_fence();
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) v[n1] = tb[wf_slot((N - (64 * n1 + l)) & (N - 1))];
    wave_lds_fence();
    wave_fft_store_natural(R, tb, l);
    wave_lds_fence();
#pragma unroll
    for (int n1 = 0; n1 < 16; ++n1) { const cplx r = tb[wf_slot(64 * n1 + l)]; v[n1] = mk(v[n1].x + r.x, v[n1].y - r.y); }
    wave_lds_fence();
    wave_fft1024<false>(v, tb, tables, l);     // y_L + i y_R = conj(FFT(conj(P))) / N
    const double scale = 1.0 / (double)N;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int64_t j = wf_bin(l, i) - (len - 1);
        const int64_t t = blk * B + j;
        if (j >= 0 && t < n) {
            stream_store(out + t, v[i].x * scale);
            stream_store(out + n + t, -v[i].y * scale);
        }
    }
}

// [re(x_0..x_C-1), im(x_0..x_C-1)] planes (2C real channels of n samples) from interleaved complex columns; `swap` exchanges
// the two halves and `neg_im` negates the imaginary planes
__global__ void split_complex_kernel(const cplx* __restrict__ x, int64_t n, int C, int swap, int neg_im, double* __restrict__ out) {
    const int64_t total = (int64_t)C * n;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const cplx v = x[idx];
        const double im = neg_im ? -v.y : v.y;
        out[(swap ? total : 0) + idx] = v.x;
        out[(swap ? 0 : total) + idx] = im;
    }
}
__global__ void widen_real_kernel(const double* __restrict__ x, int64_t total, int second_half, double* __restrict__ out) {
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        out[(second_half ? total : 0) + idx] = x[idx];
        out[(second_half ? 0 : total) + idx] = 0.0;
    }
}
__global__ void __launch_bounds__(1024) abs_sum_kernel(const double* __restrict__ x, int64_t n, int64_t skip, double* __restrict__ out) {
    __shared__ double sh[1024];
    const double* col = x + (int64_t)blockIdx.x * n;
    double acc = 0.0;
    for (int64_t i = skip + threadIdx.x; i < n; i += blockDim.x) acc += fabs(col[i]);
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s2 = 512; s2 > 0; s2 >>= 1) {
        if ((int)threadIdx.x < s2) sh[threadIdx.x] += sh[threadIdx.x + s2];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = sh[0];
}

// hipFFT plans and work buffers of the last decode shape, kept across calls (creating three plans and six buffers costs
// more than rendering a short signal); released by emagls_cache_clear().  One render at a time per process.
namespace {
struct DecodeWork {
    int C = 0, Nf = 0, device = -1;
    int64_t nblocks = 0;
    double *seg = nullptr, *wpad = nullptr, *y = nullptr;
    cplx *Xf = nullptr, *Wf = nullptr, *Yf = nullptr;
    cplx *wtab = nullptr, *circle = nullptr;   // ols_wave_kernel: pair filters at the lanes' bins, exp(-2 pi i m / 1024)
    hipfftHandle pf = 0, pw = 0, pi = 0;
    void release() {
        if (pf) hipfftDestroy(pf);
        if (pw) hipfftDestroy(pw);
        if (pi) hipfftDestroy(pi);
        pf = pw = pi = 0;
        hipFree(seg); hipFree(wpad); hipFree(y); hipFree(Xf); hipFree(Wf); hipFree(Yf); hipFree(wtab); hipFree(circle);
        seg = wpad = y = nullptr; Xf = Wf = Yf = nullptr; wtab = circle = nullptr;
        C = Nf = 0; nblocks = 0; device = -1;
    }
    void ensure(int C_, int64_t nblocks_, int Nf_) {
        int dev = 0;
        HIP_CHECK(hipGetDevice(&dev));
        if (C_ == C && nblocks_ == nblocks && Nf_ == Nf && dev == device) return;
        release();
        const int Pf = Nf_ / 2 + 1;
        try {
            int nn[1] = {Nf_};
            HIP_CHECK(hipMalloc(&wpad, sizeof(double) * 2 * C_ * Nf_));
            HIP_CHECK(hipMalloc(&Wf, sizeof(cplx) * 2 * C_ * Pf));
            fft_check(hipfftPlanMany(&pw, 1, nn, nullptr, 1, Nf_, nullptr, 1, Pf, HIPFFT_D2Z, 2 * C_), "plan D2Z filters");
            if (Nf_ == WF_N && nblocks_ == 0) {
                HIP_CHECK(hipMalloc(&wtab, sizeof(cplx) * (size_t)((C_ + 1) / 2) * 16 * 2 * 64));
                HIP_CHECK(hipMalloc(&circle, sizeof(cplx) * WF_N));
                launch_twiddles(WF_N, circle, nullptr);
                HIP_CHECK(hipStreamSynchronize(nullptr));
            }
            if (nblocks_ > 0) {   // (0: the fused kernel keeps the signal side in LDS; only the filter spectra pass through hipFFT)
                HIP_CHECK(hipMalloc(&seg, sizeof(double) * C_ * nblocks_ * Nf_));
                HIP_CHECK(hipMalloc(&y, sizeof(double) * 2 * nblocks_ * Nf_));
                HIP_CHECK(hipMalloc(&Xf, sizeof(cplx) * C_ * nblocks_ * Pf));
                HIP_CHECK(hipMalloc(&Yf, sizeof(cplx) * 2 * nblocks_ * Pf));
                fft_check(hipfftPlanMany(&pf, 1, nn, nullptr, 1, Nf_, nullptr, 1, Pf, HIPFFT_D2Z, (int)(C_ * nblocks_)), "plan D2Z signal");
                fft_check(hipfftPlanMany(&pi, 1, nn, nullptr, 1, Pf, nullptr, 1, Nf_, HIPFFT_Z2D, (int)(2 * nblocks_)), "plan Z2D");
            }
        } catch (...) { release(); throw; }
        C = C_; nblocks = nblocks_; Nf = Nf_; device = dev;
    }
};
std::mutex g_decode_mu;
DecodeWork g_decode;
}  // namespace

void decode_cache_clear() {
    std::lock_guard<std::mutex> lk(g_decode_mu);
    g_decode.release();
}

// EMAGLS_DECODE_FUSED=0: always the hipFFT passes
static bool decode_fused_enabled() { const char* e = getenv("EMAGLS_DECODE_FUSED"); return !(e && e[0] == '0'); }

// true when binaural_decode_real takes the wave-private transforms for filters of `len` taps (ols_wave_kernel): the complex render
// then hands over its interleaved signal as it is (sigc) instead of splitting it into planes first
bool decode_wave_form(int64_t len) {
    const char* e_rr = getenv("EMAGLS_DECODE_REGFFT");
    const char* e_wave = getenv("EMAGLS_DECODE_WAVE");
    return len <= 2048 && decode_fused_enabled() && 2 * len > WF_N / 2 && 2 * len <= WF_N && !(e_rr && e_rr[0] == '0') && !(e_wave && e_wave[0] == '0');
}

void binaural_decode_real(const double* sig, int64_t n, int C, const double* wL, const double* wR, int64_t len,
                          double* out, hipStream_t st, const cplx* sigc) {
    if (n <= 0) return;
    if (sigc && !decode_wave_form(len)) throw Error(3, "binaural_decode_real: an interleaved complex signal needs the wave form");
    if (len <= 2048 && decode_fused_enabled()) {
        // block length ~ filter length: twice the transforms of the 4x blocks below, but every one of them stays in LDS
        int Nf = 256, log2n = 8;
        while (Nf < 2 * len) { Nf <<= 1; ++log2n; }
        const int64_t B = Nf - (len - 1);
        const int64_t nblocks = ceil_div(n, B);
        std::lock_guard<std::mutex> lk(g_decode_mu);
        DecodeWork& w = g_decode;
        w.ensure(C, 0, Nf);
        const char* e_rr = getenv("EMAGLS_DECODE_REGFFT");   // (read at every call: a test switches forms inside one process)
        const bool use_rr = !(e_rr && e_rr[0] == '0');
        const char* e_wave = getenv("EMAGLS_DECODE_WAVE");
        const bool wave_form = use_rr && Nf == WF_N && !(e_wave && e_wave[0] == '0');
        const char* e_ft = getenv("EMAGLS_DECODE_FILTER_FFT");   // =hipfft: the filter side of the wave form through hipFFT + ols_wave_tables_kernel
        const bool own_filter_side = wave_form && !(e_ft && e_ft[0] == 'h');
        if (!own_filter_side) {
            fft_check(hipfftSetStream(w.pw, st), "set stream");
            ols_padfilt_kernel<<<256, 256, 0, st>>>(wL, wR, C, len, Nf, w.wpad);
            KERNEL_CHECK();
            fft_check(hipfftExecD2Z(w.pw, w.wpad, (hipfftDoubleComplex*)w.Wf), "exec D2Z filters");
        }
        if (wave_form) {   // wave-private transforms
            static thread_local int ncu_w = 0;   // (per calling thread: one device query instead of one per call)
            if (!ncu_w) {
                int dev_w = 0;
                HIP_CHECK(hipGetDevice(&dev_w));
                HIP_CHECK(hipDeviceGetAttribute(&ncu_w, hipDeviceAttributeMultiprocessorCount, dev_w));
            }
            const int npairs = (C + 1) / 2;
            if (own_filter_side) ols_wave_filter_kernel<<<npairs, 128, 0, st>>>(wL, wR, C, len, sigc ? 1 : 0, w.circle, w.wtab);
            else ols_wave_tables_kernel<<<(unsigned)ceil_div((int64_t)npairs * 2048, 256), 256, 0, st>>>(w.Wf, C, sigc ? 1 : 0, w.wtab);
            KERNEL_CHECK();
            const size_t dyn_w = sizeof(cplx) * (WF_TABLES + (size_t)OLSW_WAVES * WF_BUF);
            static PerDeviceOnce wave_once;
            if (wave_once.first()) {
                HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ols_wave_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ols_wave_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ols_wave_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ols_wave_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            }
            // waves per block: as few as still give every CU a workgroup (8 / G blocks per workgroup)
            const int64_t cu = ncu_w;
#define EMAGLS_OLSW_GO(G_) ols_wave_kernel<G_><<<(unsigned)ceil_div(nblocks, OLSW_WAVES / G_), 64 * OLSW_WAVES, dyn_w, st>>>(sig, sigc, n, C, w.wtab, w.circle, len, B, nblocks, out)
            if (nblocks >= 8 * cu) EMAGLS_OLSW_GO(1);
            else if (nblocks >= 4 * cu) EMAGLS_OLSW_GO(2);
            else if (nblocks >= 2 * cu) EMAGLS_OLSW_GO(4);
            else EMAGLS_OLSW_GO(8);
#undef EMAGLS_OLSW_GO
            KERNEL_CHECK();
            HIP_CHECK(hipStreamSynchronize(st));
            return;
        }
        if (use_rr && Nf <= 1024) {   // register-resident transforms (two-factor form)
            const int n1 = Nf == 1024 ? 32 : 16, n2 = Nf == 256 ? 16 : 32;
            int dev_rr = 0, ncu = 256;
            HIP_CHECK(hipGetDevice(&dev_rr));
            HIP_CHECK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev_rr));
            const bool wide_blocks = nblocks < 2 * (int64_t)ncu;   // few blocks: the shortest path through each
            const int nt_rr = wide_blocks ? 256 : 128;
            const size_t tsz = (size_t)std::max(n1 * (n2 + 1), Nf);
            const size_t dyn_rr = sizeof(cplx) * ((size_t)(nt_rr / 32) * tsz + Nf / 2);
            static PerDeviceOnce rr_once;
            if (rr_once.first()) {
#define EMAGLS_RR_ATTR(A, B_, T) HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ols_fused_rr_kernel<A, B_, T>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024))
                EMAGLS_RR_ATTR(16, 16, 128); EMAGLS_RR_ATTR(16, 32, 128); EMAGLS_RR_ATTR(32, 32, 128);
                EMAGLS_RR_ATTR(16, 16, 256); EMAGLS_RR_ATTR(16, 32, 256); EMAGLS_RR_ATTR(32, 32, 256);
#undef EMAGLS_RR_ATTR
            }
            const dim3 grid_rr((unsigned)nblocks);
#define EMAGLS_RR_GO(A, B_) do { if (wide_blocks) ols_fused_rr_kernel<A, B_, 256><<<grid_rr, 256, dyn_rr, st>>>(sig, n, C, w.Wf, len, B, out); \
                                 else ols_fused_rr_kernel<A, B_, 128><<<grid_rr, 128, dyn_rr, st>>>(sig, n, C, w.Wf, len, B, out); } while (0)
            if (Nf == 1024) EMAGLS_RR_GO(32, 32); else if (Nf == 512) EMAGLS_RR_GO(16, 32); else EMAGLS_RR_GO(16, 16);
#undef EMAGLS_RR_GO
            KERNEL_CHECK();
            HIP_CHECK(hipStreamSynchronize(st));
            return;
        }
        const int ntp = Nf <= 1024 ? 4 : (Nf == 2048 ? 2 : 1);
        const size_t dyn = sizeof(cplx) * ((size_t)ntp * (Nf + Nf / 16) + Nf / 2);
        static PerDeviceOnce attr_once;
        if (attr_once.first()) {
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ols_fused_kernel<2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ols_fused_kernel<3, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
            HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(ols_fused_kernel<5, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        }
        const dim3 grid((unsigned)nblocks);
        if (ntp == 4) ols_fused_kernel<2, 4><<<grid, OLSF_NT, dyn, st>>>(sig, n, C, w.Wf, len, Nf, log2n, B, out);
        else if (ntp == 2) ols_fused_kernel<3, 2><<<grid, OLSF_NT, dyn, st>>>(sig, n, C, w.Wf, len, Nf, log2n, B, out);
        else ols_fused_kernel<5, 1><<<grid, OLSF_NT, dyn, st>>>(sig, n, C, w.Wf, len, Nf, log2n, B, out);
        KERNEL_CHECK();
        HIP_CHECK(hipStreamSynchronize(st));
        return;
    }
    int Nf = 1024;
    while (Nf < 4 * len) Nf <<= 1;
    const int64_t B = Nf - (len - 1);
    const int64_t nblocks = ceil_div(n, B);
    const int Pf = Nf / 2 + 1;
    std::lock_guard<std::mutex> lk(g_decode_mu);
    DecodeWork& w = g_decode;
    w.ensure(C, nblocks, Nf);
    fft_check(hipfftSetStream(w.pf, st), "set stream");
    fft_check(hipfftSetStream(w.pw, st), "set stream");
    fft_check(hipfftSetStream(w.pi, st), "set stream");
    ols_pack_kernel<<<2048, 256, 0, st>>>(sig, n, C, nblocks, Nf, B, len, w.seg);
    KERNEL_CHECK();
    ols_padfilt_kernel<<<256, 256, 0, st>>>(wL, wR, C, len, Nf, w.wpad);
    KERNEL_CHECK();
    fft_check(hipfftExecD2Z(w.pf, w.seg, (hipfftDoubleComplex*)w.Xf), "exec D2Z signal");
    fft_check(hipfftExecD2Z(w.pw, w.wpad, (hipfftDoubleComplex*)w.Wf), "exec D2Z filters");
    ols_mac_kernel<<<2048, 256, 0, st>>>(w.Xf, w.Wf, C, nblocks, Pf, w.Yf);
    KERNEL_CHECK();
    fft_check(hipfftExecZ2D(w.pi, (hipfftDoubleComplex*)w.Yf, w.y), "exec Z2D");
    ols_unpack_kernel<<<2048, 256, 0, st>>>(w.y, n, nblocks, Nf, B, len, out);
    KERNEL_CHECK();
    HIP_CHECK(hipStreamSynchronize(st));
}

// Complex-SH rendering (dependencies/binauralDecode.m:39-42,59-64): the reference accumulates complex fftfilt products and
// keeps real(.) of the sum.  real(w * x) = re(w) * re(x) - im(w) * im(x): the same overlap-save path on 2C real channels
// [re x; im x] with the filters [re w; -im w].  The discarded imaginary part, whose absolute sum the reference prints in a
// warning, is re(w) * im(x) + im(w) * re(x): a second pass on [im x; re x] with [re w; im w], only when asked for.
// sig2 / w2L / w2R: work buffers of 2 C n and 2 C len doubles; sig, wL, wR device pointers (interleaved complex when flagged)
void binaural_decode_complex(const void* sig, bool sig_cplx, int64_t n, int C, const void* wL, const void* wR, bool w_cplx, int64_t len,
                             double* sig2, double* w2L, double* w2R, double* out, double* imag_abs, double* d_tmp, hipStream_t st, int64_t imag_skip) {
    if (n <= 0) return;
    auto planes = [&](const void* x, bool is_cplx, int64_t rows, int swap, int neg_im, double* dst) {
        const int64_t total = (int64_t)C * rows;
        const unsigned grid = (unsigned)std::min<int64_t>(2048, ceil_div(total, 256));
        if (is_cplx) split_complex_kernel<<<grid, 256, 0, st>>>((const cplx*)x, rows, C, swap, neg_im, dst);
        else widen_real_kernel<<<grid, 256, 0, st>>>((const double*)x, total, swap, dst);
        KERNEL_CHECK();
    };
    // (the wave form reads a complex signal as it lies in memory: sample = the pair (re, im) of one transform)
    const cplx* sigc = (sig_cplx && decode_wave_form(len)) ? (const cplx*)sig : nullptr;
    if (!sigc) planes(sig, sig_cplx, n, 0, 0, sig2);
    planes(wL, w_cplx, len, 0, 1, w2L);
    planes(wR, w_cplx, len, 0, 1, w2R);
    binaural_decode_real(sig2, n, 2 * C, w2L, w2R, len, out, st, sigc);
    if (imag_abs) {
        if (sigc) {                                    // [re x; im x] stays, the filter planes change places: [im w; re w]
            planes(wL, w_cplx, len, 1, 0, w2L);
            planes(wR, w_cplx, len, 1, 0, w2R);
        } else {
            planes(sig, sig_cplx, n, 1, 0, sig2);      // [im x; re x]
            planes(wL, w_cplx, len, 0, 0, w2L);        // [re w; im w]
            planes(wR, w_cplx, len, 0, 0, w2R);
        }
        binaural_decode_real(sig2, n, 2 * C, w2L, w2R, len, d_tmp, st, sigc);
        abs_sum_kernel<<<2, 1024, 0, st>>>(d_tmp, n, imag_skip, d_tmp + 2 * n);   // (binauralDecode.m:53-62: summed after the delay cut)
        KERNEL_CHECK();
        HIP_CHECK(hipMemcpyAsync(imag_abs, d_tmp + 2 * n, 2 * sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_CHECK(hipStreamSynchronize(st));
    }
}

// ---------------------------------------------------------------------------------------------
// applyRadialFilter's convolution (dependencies/applyRadialFilter.m:24-30): out(:,c) = fftfilt(ir(:,order(c)), in(:,c)) with
// the first `skip` samples dropped (the filter delay nfft/2) -- the same overlap-save blocks as above, but every channel keeps
// its own output and the filter of channel c is the one of its SH order floor(sqrt(c)).
// ---------------------------------------------------------------------------------------------
// wpad[o][i] = ir[o*len + i] for i < len else 0
__global__ void olsc_padfilt_kernel(const double* __restrict__ ir, int nOrd, int64_t len, int Nf, double* __restrict__ wpad) {
    const int64_t total = (int64_t)nOrd * Nf;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = idx % Nf, o = idx / Nf;
        wpad[idx] = (i < len) ? ir[o * len + i] : 0.0;
    }
}
// Xf[c][b][k] *= Wf[order(c)][k]
__global__ void olsc_mul_kernel(cplx* __restrict__ Xf, const cplx* __restrict__ Wf, int C, int64_t nblocks, int Pf) {
    const int64_t total = (int64_t)C * nblocks * Pf;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = idx % Pf, c = idx / ((int64_t)nblocks * Pf);
        int o = (int)sqrt((double)c);
        while ((o + 1) * (o + 1) <= c) ++o;
        while (o * o > c) --o;
        Xf[idx] = Xf[idx] * Wf[(int64_t)o * Pf + k];
    }
}
// out[c*(n-skip) + t - skip] = y[c][b][len-1+i] / Nf,  t = b*B + i in [skip, n)
__global__ void olsc_unpack_kernel(const double* __restrict__ y, int64_t n, int C, int64_t nblocks, int Nf, int64_t B, int64_t len,
                                   int64_t skip, double* __restrict__ out) {
    const int64_t nout = n - skip, total = (int64_t)C * nout;
    const double scale = 1.0 / (double)Nf;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t t = idx % nout + skip, c = idx / nout;
        const int64_t b = t / B, i = t % B;
        out[idx] = y[((int64_t)c * nblocks + b) * Nf + (len - 1) + i] * scale;
    }
}

// sig holds n_in samples per channel; the signal is taken as zero-padded to n >= n_in samples (applyRadialFilter.m:20-22)
void filter_channels_by_order(const double* sig, int64_t n_in, int64_t n, int C, const double* ir, int nOrd, int64_t len, int64_t skip,
                              double* out, hipStream_t st) {
    if (n <= skip) return;
    int Nf = 1024;
    while (Nf < 4 * len) Nf <<= 1;
    const int64_t B = Nf - (len - 1);
    const int64_t nblocks = ceil_div(n, B);
    const int Pf = Nf / 2 + 1;
    double *seg = nullptr, *wpad = nullptr;
    cplx *Xf = nullptr, *Wf = nullptr;
    hipfftHandle pf = 0, pw = 0, pi = 0;
    auto cleanup = [&] {
        if (pf) hipfftDestroy(pf);
        if (pw) hipfftDestroy(pw);
        if (pi) hipfftDestroy(pi);
        hipFree(seg); hipFree(wpad); hipFree(Xf); hipFree(Wf);
    };
    try {
        HIP_CHECK(hipMalloc(&seg, sizeof(double) * C * nblocks * Nf));
        HIP_CHECK(hipMalloc(&wpad, sizeof(double) * nOrd * Nf));
        HIP_CHECK(hipMalloc(&Xf, sizeof(cplx) * C * nblocks * Pf));
        HIP_CHECK(hipMalloc(&Wf, sizeof(cplx) * nOrd * Pf));
        int nn[1] = {Nf};
        fft_check(hipfftPlanMany(&pf, 1, nn, nullptr, 1, Nf, nullptr, 1, Pf, HIPFFT_D2Z, (int)(C * nblocks)), "plan D2Z signal");
        fft_check(hipfftPlanMany(&pw, 1, nn, nullptr, 1, Nf, nullptr, 1, Pf, HIPFFT_D2Z, nOrd), "plan D2Z filters");
        fft_check(hipfftPlanMany(&pi, 1, nn, nullptr, 1, Pf, nullptr, 1, Nf, HIPFFT_Z2D, (int)(C * nblocks)), "plan Z2D");
        fft_check(hipfftSetStream(pf, st), "set stream");
        fft_check(hipfftSetStream(pw, st), "set stream");
        fft_check(hipfftSetStream(pi, st), "set stream");
        // (ols_pack_kernel reads in[c*n + src] for src < n: the channel stride is the true length, the padding comes from the bound)
        ols_pack_kernel<<<2048, 256, 0, st>>>(sig, n_in, C, nblocks, Nf, B, len, seg);
        KERNEL_CHECK();
        olsc_padfilt_kernel<<<64, 256, 0, st>>>(ir, nOrd, len, Nf, wpad);
        KERNEL_CHECK();
        fft_check(hipfftExecD2Z(pf, seg, (hipfftDoubleComplex*)Xf), "exec D2Z signal");
        fft_check(hipfftExecD2Z(pw, wpad, (hipfftDoubleComplex*)Wf), "exec D2Z filters");
        olsc_mul_kernel<<<2048, 256, 0, st>>>(Xf, Wf, C, nblocks, Pf);
        KERNEL_CHECK();
        fft_check(hipfftExecZ2D(pi, (hipfftDoubleComplex*)Xf, seg), "exec Z2D");
        olsc_unpack_kernel<<<2048, 256, 0, st>>>(seg, n, C, nblocks, Nf, B, len, skip, out);
        KERNEL_CHECK();
        HIP_CHECK(hipStreamSynchronize(st));
    } catch (...) { cleanup(); throw; }
    cleanup();
}

}  // namespace emagls
