// HRIR prologue and filter epilogue: LDS-resident FP64 FFTs fused with the reference's delay /
// mirroring / windowing arithmetic.
//
//   prologue  lib/getEMagLsFilters.m:72-81  (zero-pad, grpdelay -> median, applySubsampleDelay, fft)
//             lib/getEMagLsFiltersFromAtf.m:43-53 (integer circshift variant)
//   epilogue  lib/getEMagLsFilters.m:110-142 (DC rule, Hermitian mirror or getShFreqDomainConjugate,
//             ifft, applySubsampleDelay / circshift, truncate, getFadeWindow)
//
// The reference's ifft(fft(h) .* E) followed by fft(.) collapses to fft(h) .* E, and
// applySubsampleDelay(ifft(W)) to ifft(W .* E): one transform instead of three.
#include "kernels.hpp"
#include "lds_fft.hpp"
#include "wave_fft.hpp"

namespace emagls {

// tw[j] = exp(-2 pi i j / nfft), j = 0..nfft-1
__global__ void twiddle_kernel(int nfft, cplx* __restrict__ tw, size_t bstride) {
    tw = boff(tw, bstride);
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nfft) return;
    double s, c;
    // exact octant symmetries are not needed at 1e-16; sincos of the reduced angle is enough
    sincos(-2.0 * kPi * (double)j / (double)nfft, &s, &c);
    if (j == 0) { c = 1.0; s = 0.0; }
    if (2 * j == nfft) { c = -1.0; s = 0.0; }
    if (4 * j == nfft) { c = 0.0; s = -1.0; }
    if (4 * j == 3 * nfft) { c = 0.0; s = 1.0; }
    tw[j] = mk(c, s);
}

// ---------------------------------------------------------------------------------------------
// sum over directions (input of grpdelay(sum(h,2),...), lib/getEMagLsFilters.m:74)
// partial[chunk][e][n] = sum_{d in chunk} h_e[d*L + n]
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) hrir_dirsum_kernel(const double* __restrict__ hL, const double* __restrict__ hR,
                                                          int64_t L, int64_t D, int chunk, double* __restrict__ partial, size_t bstride) {
    hL = boff(hL, bstride); hR = boff(hR, bstride); partial = boff(partial, bstride);
    const int e = blockIdx.y;
    const double* h = e ? hR : hL;
    const int64_t d0 = (int64_t)blockIdx.x * chunk;
    const int64_t d1 = min(D, d0 + chunk);
    for (int64_t n = threadIdx.x; n < L; n += blockDim.x) {
        double acc = 0.0;
        for (int64_t d = d0; d < d1; ++d) acc += h[d * L + n];
        partial[((int64_t)blockIdx.x * 2 + e) * L + n] = acc;
    }
}

// one workgroup per ear: b = sum of partials; gd(k) = Re{ sum n b[n] z^-n / sum b[n] z^-n };
// grpd[e] = median_k gd(k)   (MATLAB grpdelay FIR branch: |den| < 10 eps -> 0)
// and behind the two delays the phases applySubsampleDelay multiplies the spectra with (applySubsampleDelay.m:8-13),
// phs[e][kb] = exp(-1j*2*pi*omega(kb)*(-grpd[e])), omega = linspace(0, 0.5, nfft/2+1), Nyquist bin forced real: one table per design
// instead of one per workgroup of hrir_fft_wave_kernel.  grpd: 2 + 4 P doubles.
__global__ void __launch_bounds__(1024) grpdelay_median_kernel(const double* __restrict__ partial, int nchunks, int64_t L,
                                                               int nfft, const cplx* __restrict__ tw,
                                                               double* __restrict__ grpd, size_t bstride) {
    partial = boff(partial, bstride); tw = boff(tw, bstride); grpd = boff(grpd, bstride);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* b = reinterpret_cast<double*>(smem);   // L
    double* v = b + ((L + 1) & ~(int64_t)1);       // npow2
    const int e = blockIdx.x;
    const int P = nfft / 2 + 1;
    int npow2 = 1;
    while (npow2 < P) npow2 <<= 1;
    // b = sum of the chunks' partial sums: nq threads per tap, four loads in flight each (one thread per tap with one accumulator
    // walks nchunks dependent L2 round trips)
    {
        double* part = reinterpret_cast<double*>(reinterpret_cast<cplx*>(v + npow2) + nfft);   // [nq][L] behind the twiddle circle
        const int nq = (int)max((int64_t)1, min((int64_t)8, (int64_t)blockDim.x / max(L, (int64_t)1)));
        for (int64_t idx = threadIdx.x; idx < (int64_t)nq * L; idx += blockDim.x) {
            const int64_t n = idx % L;
            const int q = (int)(idx / L);
            double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
            int c = q;
            for (; c + 3 * nq < nchunks; c += 4 * nq) {
                a0 += partial[((int64_t)c * 2 + e) * L + n];
                a1 += partial[((int64_t)(c + nq) * 2 + e) * L + n];
                a2 += partial[((int64_t)(c + 2 * nq) * 2 + e) * L + n];
                a3 += partial[((int64_t)(c + 3 * nq) * 2 + e) * L + n];
            }
            for (; c < nchunks; c += nq) a0 += partial[((int64_t)c * 2 + e) * L + n];
            part[idx] = (a0 + a1) + (a2 + a3);
        }
        __syncthreads();
        for (int64_t n = threadIdx.x; n < L; n += blockDim.x) {
            double acc = 0.0;
            for (int q = 0; q < nq; ++q) acc += part[(int64_t)q * L + n];
            b[n] = acc;
        }
    }
    __syncthreads();
    // the twiddle circle in LDS, walked by an index recurrence ((k n) mod nfft: one modulo and one dependent L2 round trip per
    // tap cost 25 of the kernel's 75 us at nfft = 2048)
    cplx* tws = reinterpret_cast<cplx*>(v + npow2);   // nfft
    for (int j = threadIdx.x; j < nfft; j += blockDim.x) tws[j] = tw[j];
    __syncthreads();
    auto finish = [](cplx num, cplx den) {
        if (cabs(den) < 10.0 * 2.220446049250313e-16) { num = mk(0, 0); den = mk(1, 0); }
        return cdiv(num, den).x;
    };
    const int kmain = min(P, (int)blockDim.x);          // one thread per bin; the bins beyond the workgroup's size: all threads per bin
    for (int k = threadIdx.x; k < npow2; k += blockDim.x) v[k] = INFINITY;
    if ((int)threadIdx.x < kmain) {
        const int k = threadIdx.x;
        cplx num = mk(0, 0), den = mk(0, 0);
        int j = 0;
        for (int64_t n = 0; n < L; ++n) {
            const cplx w = tws[j];
            j += k; if (j >= nfft) j -= nfft;
            cfma(den, b[n], w);
            cfma(num, (double)n * b[n], w);
        }
        v[k] = finish(num, den);
    }
    for (int k = kmain; k < P; ++k) {   // (nfft = 2048: the Nyquist bin; a second pass of one thread per bin would leave 1023 of 1024 idle)
        __shared__ double red[16];
        cplx num = mk(0, 0), den = mk(0, 0);
        for (int64_t n = threadIdx.x; n < L; n += blockDim.x) {
            const cplx w = tws[(int)(((int64_t)k * n) % nfft)];
            cfma(den, b[n], w);
            cfma(num, (double)n * b[n], w);
        }
        double parts[4] = {num.x, num.y, den.x, den.y};
        for (int q = 0; q < 4; ++q) {
            double x = wave_sum(parts[q]);
            __syncthreads();
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
            __syncthreads();
            double t = 0.0;
            for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
            parts[q] = t;
        }
        if (threadIdx.x == 0) v[k] = finish(mk(parts[0], parts[1]), mk(parts[2], parts[3]));
    }
    __syncthreads();
    // bitonic sort ascending
    for (int size = 2; size <= npow2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int i = threadIdx.x; i < npow2 / 2; i += blockDim.x) {
                const int lo = ((i / stride) * 2 * stride) + (i % stride);
                const int hi = lo + stride;
                const bool up = ((lo & size) == 0);
                const double a = v[lo], c = v[hi];
                if ((a > c) == up) { v[lo] = c; v[hi] = a; }
            }
            __syncthreads();
        }
    }
    const double med = (P & 1) ? v[(P - 1) / 2] : 0.5 * (v[P / 2 - 1] + v[P / 2]);
    if (threadIdx.x == 0) grpd[e] = med;
    cplx* phs = reinterpret_cast<cplx*>(grpd + 2) + (size_t)e * P;
    for (int kb = threadIdx.x; kb < P; kb += blockDim.x) {
        double sn, cs;
        sincos((6.283185307179586 * ((double)kb / (double)nfft)) * med, &sn, &cs);
        if (kb == P - 1) sn = 0.0;
        phs[kb] = mk(cs, sn);
    }
}

// ---------------------------------------------------------------------------------------------
// HRIR spectra.  8 directions per workgroup (the bin-major outputs are written in runs of 8 directions = 64 bytes of |H|),
// transformed TS directions at a time, both ears packed into one complex transform, so that the LDS footprint stays at
// TS * nfft * 16 B + one 16 KB region (twiddles during the transform, the delay phases after it): 82 KB at nfft = 1024, TS = 4.
// That leaves room for a workgroup of another batch's persistent sweep (77 KB, sweep_persist.hip) on the same CU -- a kernel
// that needs a whole CU only starts once a CU is completely empty, i.e. between two sweeps, and stalls its batch until then.
// The |H| values of the first sub-tiles wait in registers for the last one.
//   kcut0 = 0-based index of the first magnitude-least-squares bin
//   Hc  [e][kb][d]  complex, kb <  n_c  (bins that need the complex HRTF)
//   Habs[e][kb - kabs0][d] real, kb >= kabs0
// mode 0: fractional delay by -grpd[e] (applySubsampleDelay);  mode 1: circshift by -round(grpd[e])
// ---------------------------------------------------------------------------------------------
constexpr int HF_TD = 8;      // directions per workgroup
constexpr int HF_MAXS = 3;    // bins per thread: ceil((nfft/2+1) / 512) for nfft <= 2048
template <int TS>
__global__ void __launch_bounds__(512) hrir_fft_kernel(const double* __restrict__ hL, const double* __restrict__ hR,
                                                       int64_t L, int64_t D, const int64_t* __restrict__ didx, int nfft,
                                                       int log2n,
                                                       const cplx* __restrict__ tw, const double* __restrict__ grpd,
                                                       int mode, int n_c, int kabs0, cplx* __restrict__ Hc,
                                                       double* __restrict__ Habs, int64_t ldD, double* __restrict__ HcT, int ldT,
                                                       size_t bstride) {
    hL = boff(hL, bstride); hR = boff(hR, bstride); didx = boff(didx, bstride); tw = boff(tw, bstride); grpd = boff(grpd, bstride); Hc = boff(Hc, bstride); Habs = boff(Habs, bstride);
    HcT = boff(HcT, bstride);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int P = nfft / 2 + 1;
    cplx* tws = reinterpret_cast<cplx*>(smem);  // nfft/2 twiddles while the stages run ...
    cplx* phs = tws;                            // ... [2][P] delay phase per ear and bin (mode 0) afterwards
    cplx* buf = tws + 2 * P;                    // TS * nfft
    const int64_t d0 = (int64_t)blockIdx.x * HF_TD;
    const int nt = (int)min((int64_t)HF_TD, D - d0);
    const double gL = grpd[0], gR = grpd[1];
    int sL = 0, sR = 0;
    if (mode == 1) { sL = (int)round(gL); sR = (int)round(gR); }
    // Zero padding (mode 0, no circular shift): with L * 2^z <= nfft only every 2^z-th entry of the bit-reversed input is
    // non-zero, and the first z radix-2 stages turn each group of 2^z entries into copies of its first one.  The buffer is
    // filled with that state directly and the transform starts at stage z (3 of 10 stages at 128 taps, nfft 1024).
    int zskip = 0;
    if (mode == 0) while (zskip < log2n && (L << (zskip + 1)) <= (int64_t)nfft) ++zskip;
    double habs[HF_MAXS][2][HF_TD];
#pragma unroll
    for (int s = 0; s < HF_MAXS; ++s)
#pragma unroll
        for (int t = 0; t < HF_TD; ++t) habs[s][0][t] = habs[s][1][t] = 0.0;

#pragma unroll
    for (int sub = 0; sub < HF_TD / TS; ++sub) {
        const int t0 = sub * TS;
        const int nts = min(max(nt - t0, 0), TS);
        if (sub > 0) __syncthreads();   // (the previous sub-tile's readers of phs / buf are done)
        for (int j = threadIdx.x; j < nfft / 2; j += blockDim.x) tws[j] = tw[j];
        for (int idx = threadIdx.x; idx < nts * nfft; idx += blockDim.x) {
            const int t = idx >> log2n, j = idx & (nfft - 1);
            const int n = (int)bitrev((unsigned)(j & ~((1 << zskip) - 1)), log2n);
            // circshift(h, -s): out[n] = h[(n + s) mod nfft]; zero beyond the L recorded taps
            int nl = (n + sL) % nfft; if (nl < 0) nl += nfft;
            int nr = (n + sR) % nfft; if (nr < 0) nr += nfft;
            const int64_t dsrc = didx ? didx[d0 + t0 + t] : d0 + t0 + t;  // optional gather of matched directions
            const double a = (nl < L) ? hL[dsrc * L + nl] : 0.0;
            const double c = (nr < L) ? hR[dsrc * L + nr] : 0.0;
            buf[(size_t)t * nfft + j] = mk(a, c);
        }
        __syncthreads();
        lds_fft_stages<false>(buf, tws, nfft, log2n, nts, zskip);   // (ends with a barrier: the twiddles are free)
        if (mode == 0) {
            // exp(-1j*2*pi*omega*(-grpD)), omega = linspace(0, 0.5, nfft/2+1): once per sub-tile, not once per direction
            for (int j = threadIdx.x; j < 2 * P; j += blockDim.x) {
                const int e = j / P, kb = j - e * P;
                double sn, cs;
                sincos((6.283185307179586 * ((double)kb / (double)nfft)) * (e ? gR : gL), &sn, &cs);
                if (kb == P - 1) sn = 0.0;  // Nyquist bin forced real (applySubsampleDelay.m:12)
                phs[j] = mk(cs, sn);
            }
            __syncthreads();
        }
        // unpack ears, apply delay phase; complex rows are written now, |H| is kept for the 64-byte runs at the end
#pragma unroll
        for (int s = 0; s < HF_MAXS; ++s) {
            const int kb = threadIdx.x + s * 512;
            if (kb < P) {
#pragma unroll
                for (int t = 0; t < TS; ++t) {
                    if (t < nts) {
                        const cplx* x = buf + (size_t)t * nfft;
                        const cplx z = x[kb], zc = conj(x[(nfft - kb) & (nfft - 1)]);
                        cplx HLv = mk(0.5 * (z.x + zc.x), 0.5 * (z.y + zc.y));
                        cplx HRv = mk(0.5 * (z.y - zc.y), -0.5 * (z.x - zc.x));  // (z - zc) / (2i)
                        if (mode == 0) {
                            HLv = HLv * phs[kb];
                            HRv = HRv * phs[P + kb];
                        }
                        const int64_t d = d0 + t0 + t;
                        if (kb < n_c) {
                            Hc[((int64_t)0 * n_c + kb) * ldD + d] = HLv;
                            Hc[((int64_t)1 * n_c + kb) * ldD + d] = HRv;
                        }
                        // |H| = n rsqrt(n): spectra are O(1), |H|^2 can neither overflow nor underflow
                        const double nl2 = norm2(HLv), nr2 = norm2(HRv);
                        habs[s][0][t0 + t] = nl2 > 0.0 ? nl2 * fast_rsqrt(nl2) : 0.0;
                        habs[s][1][t0 + t] = nr2 > 0.0 ? nr2 * fast_rsqrt(nr2) : 0.0;
                    }
                }
            }
        }
        // direction-major copy of the complex rows, HcT[d][2 (e n_c + kb) + re/im]: the K-major operand of the MFMA product
        // H conj(Yc) of the least-squares rows (launch_hy_conj_mfma)
        if (HcT) {
            for (int idx = threadIdx.x; idx < TS * n_c; idx += blockDim.x) {
                const int t = idx / n_c, kb = idx - t * n_c;
                if (t >= nts) continue;
                const cplx* x = buf + (size_t)t * nfft;
                const cplx z = x[kb], zc = conj(x[(nfft - kb) & (nfft - 1)]);
                cplx HLv = mk(0.5 * (z.x + zc.x), 0.5 * (z.y + zc.y));
                cplx HRv = mk(0.5 * (z.y - zc.y), -0.5 * (z.x - zc.x));
                if (mode == 0) { HLv = HLv * phs[kb]; HRv = HRv * phs[P + kb]; }
                double* row = HcT + (int64_t)(d0 + t0 + t) * ldT;
                row[2 * kb] = HLv.x; row[2 * kb + 1] = HLv.y;
                row[2 * (n_c + kb)] = HRv.x; row[2 * (n_c + kb) + 1] = HRv.y;
            }
        }
    }
    // |H| rows: 8 consecutive directions per (bin, ear) and thread
    const int64_t na = P - kabs0;
#pragma unroll
    for (int s = 0; s < HF_MAXS; ++s) {
        const int kb = threadIdx.x + s * 512;
        if (kb < P && kb >= kabs0) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                double* row = Habs + ((int64_t)e * na + (kb - kabs0)) * ldD + d0;   // (ldD and d0 are multiples of 8: 64-byte aligned)
                if (nt == HF_TD) {
#pragma unroll
                    for (int t = 0; t < HF_TD; t += 2) *reinterpret_cast<double2*>(row + t) = make_double2(habs[s][e][t], habs[s][e][t + 1]);
                } else {
#pragma unroll
                    for (int t = 0; t < HF_TD; ++t) if (t < nt) row[t] = habs[s][e][t];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same spectra at nfft = 1024 on wave-private transforms (wave_fft.hpp): one wave = one direction (both ears packed), eight
// directions per workgroup, no barrier until the |H| runs of 8 directions are put together.  The delay phases come from the
// table grpdelay_median_kernel leaves behind grpd (mode 0).
// ---------------------------------------------------------------------------------------------
constexpr int HW_TD = 8;
constexpr int HW_STRIDE = WF_BUF + 1;   // slots between the waves' buffers (+1: the |H| gather reads 8 buffers at one slot index)
__global__ void __launch_bounds__(64 * HW_TD) hrir_fft_wave_kernel(const double* __restrict__ hL, const double* __restrict__ hR,
                                                                   int64_t L, int64_t D, const int64_t* __restrict__ didx,
                                                                   const cplx* __restrict__ tw, const double* __restrict__ grpd,
                                                                   int mode, int n_c, int kabs0, cplx* __restrict__ Hc,
                                                                   double* __restrict__ Habs, int64_t ldD, double* __restrict__ HcT, int ldT,
                                                                   size_t bstride) {
    hL = boff(hL, bstride); hR = boff(hR, bstride); didx = boff(didx, bstride); tw = boff(tw, bstride); grpd = boff(grpd, bstride); Hc = boff(Hc, bstride); Habs = boff(Habs, bstride);
    HcT = boff(HcT, bstride);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int P = WF_N / 2 + 1;
    cplx* tables = reinterpret_cast<cplx*>(smem);
    cplx* bufs = tables + WF_TABLES;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    wave_fft_tables(tables, tw, tid, 64 * HW_TD);
    __syncthreads();
    const int64_t d0 = (int64_t)blockIdx.x * HW_TD;
    const int nt = (int)min((int64_t)HW_TD, D - d0);
    cplx* tb = bufs + (size_t)w * HW_STRIDE;
    if (w < nt) {
        const int64_t d = d0 + w;
        const int64_t dsrc = didx ? didx[d] : d;
        const double* pl = hL + dsrc * L;
        const double* pr = hR + dsrc * L;
        cplx v[16];
        const bool upper_zero = mode == 0 && L <= WF_N / 2;
        if (mode == 0) {
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                const int n = 64 * n1 + l;
                v[n1] = (n < L && (n1 < 8 || !upper_zero)) ? mk(pl[n], pr[n]) : mk(0.0, 0.0);
            }
        } else {   // circshift(h, -s): out[n] = h[(n + s) mod nfft], zero beyond the L recorded taps
            const int sL = (int)round(grpd[0]), sR = (int)round(grpd[1]);
#pragma unroll
            for (int n1 = 0; n1 < 16; ++n1) {
                const int n = 64 * n1 + l;
                int nl = (n + sL) % WF_N; if (nl < 0) nl += WF_N;
                int nr = (n + sR) % WF_N; if (nr < 0) nr += WF_N;
                v[n1] = mk(nl < L ? pl[nl] : 0.0, nr < L ? pr[nr] : 0.0);
            }
        }
        if (upper_zero) wave_fft1024<true>(v, tb, tables, l); else wave_fft1024<false>(v, tb, tables, l);
        wave_fft_store_natural(v, tb, l);
        wave_lds_fence();
        const cplx* phs = reinterpret_cast<const cplx*>(grpd + 2);
        // lane = bin: unpack the ears, apply the delay phase, complex rows out, |H_L|, |H_R| into the slot of the bin
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            const int kb = l + 64 * s;
            if (kb < P) {
                const cplx z = tb[wf_slot(kb)], zc = conj(tb[wf_slot((WF_N - kb) & (WF_N - 1))]);
                cplx HLv = mk(0.5 * (z.x + zc.x), 0.5 * (z.y + zc.y));
                cplx HRv = mk(0.5 * (z.y - zc.y), -0.5 * (z.x - zc.x));  // (z - zc) / (2i)
                if (mode == 0) {
                    HLv = HLv * phs[kb];
                    HRv = HRv * phs[P + kb];
                }
                if (kb < n_c) {
                    Hc[((int64_t)0 * n_c + kb) * ldD + d] = HLv;
                    Hc[((int64_t)1 * n_c + kb) * ldD + d] = HRv;
                    if (HcT) {   // direction-major copy, HcT[d][2 (e n_c + kb) + re/im] (launch_hy_conj_mfma)
                        double* rowT = HcT + d * ldT;
                        *reinterpret_cast<cplx*>(rowT + 2 * kb) = HLv;
                        *reinterpret_cast<cplx*>(rowT + 2 * (n_c + kb)) = HRv;
                    }
                }
                const double nl2 = norm2(HLv), nr2 = norm2(HRv);
                tb[wf_slot(kb)] = mk(nl2 > 0.0 ? nl2 * fast_rsqrt(nl2) : 0.0, nr2 > 0.0 ? nr2 * fast_rsqrt(nr2) : 0.0);
            }
        }
    }
    __syncthreads();
    // |H| rows: lane = (row, direction), so that a wave's store covers 8 rows x 64 contiguous bytes (one thread per row with its
    // eight directions costs 64 separate 16-byte pieces per store instruction: a third of the kernel's time)
    const int64_t na = P - kabs0;
    const int t = tid & (HW_TD - 1);
    if (t < nt) {
        const double* src = reinterpret_cast<const double*>(bufs + (size_t)t * HW_STRIDE);
        for (int idx = tid >> 3; idx < 2 * (P - kabs0); idx += 64) {
            const int e = idx >= P - kabs0, kr = idx - e * (P - kabs0);
            Habs[((int64_t)e * na + kr) * ldD + d0 + t] = src[2 * wf_slot(kr + kabs0) + e];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// generic batched forward FFT of real columns (ATF path: atfs = fft(atfIrs, nfft)),
// two real columns packed per complex transform.  Column j of `x` is x[j*L .. j*L+L), columns are
// selected through `colidx` (gather of matched directions).  Output out[kb][j] for kb < P
// (bin-major, columns contiguous, leading dimension ldo).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512) real_fft_gather_kernel(const double* __restrict__ x, int64_t L, int64_t ncols,
                                                              const int64_t* __restrict__ colidx, int nfft, int log2n,
                                                              int TP, const cplx* __restrict__ tw,
                                                              cplx* __restrict__ out, int64_t ldo, int64_t inner,
                                                              int64_t ld_inner) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx* tws = reinterpret_cast<cplx*>(smem);
    cplx* buf = tws + nfft / 2;
    const int P = nfft / 2 + 1;
    const int64_t p0 = (int64_t)blockIdx.x * TP;  // pair index
    const int64_t npairs = (ncols + 1) / 2;
    const int nt = (int)min((int64_t)TP, npairs - p0);
    for (int j = threadIdx.x; j < nfft / 2; j += blockDim.x) tws[j] = tw[j];
    for (int idx = threadIdx.x; idx < nt * nfft; idx += blockDim.x) {
        const int t = idx / nfft, n = idx - t * nfft;
        const int64_t j0 = 2 * (p0 + t), j1 = j0 + 1;
        double a = 0.0, c = 0.0;
        if (n < L) {
            a = x[(colidx ? colidx[j0] : j0) * L + n];
            if (j1 < ncols) c = x[(colidx ? colidx[j1] : j1) * L + n];
        }
        buf[(size_t)t * nfft + bitrev((unsigned)n, log2n)] = mk(a, c);
    }
    __syncthreads();
    lds_fft_stages<false>(buf, tws, nfft, log2n, nt);
    for (int idx = threadIdx.x; idx < P * 2 * TP; idx += blockDim.x) {
        const int kb = idx / (2 * TP), r = idx - kb * 2 * TP;
        const int t = r >> 1, which = r & 1;
        if (t >= nt) continue;
        const int64_t j = 2 * (p0 + t) + which;
        if (j >= ncols) continue;
        const cplx* xx = buf + (size_t)t * nfft;
        const cplx z = xx[kb], zc = conj(xx[(nfft - kb) & (nfft - 1)]);
        // column j = (j / inner, j % inner) lands at (j / inner) * ld_inner + j % inner
        out[(int64_t)kb * ldo + (j / inner) * ld_inner + (j % inner)] = which ? mk(0.5 * (z.y - zc.y), -0.5 * (z.x - zc.x))
                                           : mk(0.5 * (z.x + zc.x), 0.5 * (z.y + zc.y));
    }
}

// ---------------------------------------------------------------------------------------------
// epilogue: one workgroup per (channel, ear).
//   W [e][kb][c]  positive-frequency filter spectra, kb = 0..P-1, ldW = channel stride count C
//   conj_mode 0: Hermitian mirror (real SH / raw microphones)
//   conj_mode 1: getShFreqDomainConjugate  W(nfft-k,(n,m)) = (-1)^m conj(W(k,(n,-m)))
//   conj_mode 2: getChFreqDomainConjugate  W(nfft-k, m) = conj(W(k, -m)), channels [C_0, C_-1, C_1, ..., C_-N, C_N]
//   dc_rule   1: W(0) := real(W(1))   (lib/getEMagLsFilters.m:110-111)
//   shift_mode 0: applySubsampleDelay by nfft/2 (left) / nfft/2 + grpDR - grpDL (right)
//   shift_mode 1: circshift by round(nfft/2)     (lib/getEMagLsFiltersFromAtf.m:136-138)
// out: real [e][c*len + t] (out_cplx = 0) or complex (out_cplx = 1)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) filter_epilogue_kernel(const cplx* __restrict__ W, int C, int nfft, int log2n,
                                                              int len, const cplx* __restrict__ tw,
                                                              const double* __restrict__ grpd, int conj_mode,
                                                              int dc_rule, int shift_mode, int out_cplx, double fade_rel,
                                                              void* __restrict__ outL, void* __restrict__ outR, size_t bstride) {
    W = boff(W, bstride); tw = boff(tw, bstride); grpd = boff(grpd, bstride); outL = boff(outL, bstride); outR = boff(outR, bstride);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx* tws = reinterpret_cast<cplx*>(smem);
    cplx* buf = tws + nfft / 2;
    const int c = blockIdx.x, e = blockIdx.y;
    const int P = nfft / 2 + 1;
    const cplx* We = W + (size_t)e * P * C;
    for (int j = threadIdx.x; j < nfft / 2; j += blockDim.x) tws[j] = tw[j];
    // partner channel and sign for the complex-SH conjugate rule
    int cpart = c;
    double sgn = 1.0;
    if (conj_mode == 1) {
        const int n = (int)floor(sqrt((double)c));
        const int m = c - n * n - n;
        cpart = n * n + n - m;
        sgn = (m & 1) ? -1.0 : 1.0;
    } else if (conj_mode == 2) {
        cpart = (c == 0) ? 0 : ((c & 1) ? c + 1 : c - 1);   // (2m-1, 2m) are the pair (-m, +m)
    }
    const int n_shift = nfft / 2;
    const double delay = (e == 0) ? (double)n_shift : ((double)n_shift + grpd[1]) - grpd[0];
    for (int k = threadIdx.x; k < nfft; k += blockDim.x) {
        cplx v;
        int kb;
        if (k < P) {
            kb = k;
            v = We[(size_t)kb * C + c];
            if (kb == 0 && dc_rule) v = mk(We[(size_t)1 * C + c].x, 0.0);
        } else {
            kb = nfft - k;
            v = conj(We[(size_t)kb * C + cpart]);
            v.x *= sgn; v.y *= sgn;
        }
        if (shift_mode == 0) {
            const double omega = (double)kb / (double)nfft;
            double s, cs;
            sincos((6.283185307179586 * omega) * delay, &s, &cs);  // exp(-1j*2*pi*omega*delay) = cs - i s
            cplx E = mk(cs, -s);
            if (kb == P - 1) E.y = 0.0;
            if (k >= P) E.y = -E.y;  // mirrored half is conj (applySubsampleDelay.m:13)
            v = v * E;
        }
        buf[bitrev((unsigned)k, log2n)] = v;
    }
    __syncthreads();
    lds_fft_stages<true>(buf, tws, nfft, log2n, 1);
    const double inv_n = 1.0 / (double)nfft;
    const int nf = (int)round(fade_rel * (double)len);  // getFadeWindow.m:11-12 (relFadeLen: 0.15 unless the caller says otherwise)
    const int t0 = n_shift - len / 2;
    for (int tt = threadIdx.x; tt < len; tt += blockDim.x) {
        int t = t0 + tt;
        if (shift_mode == 1) { t = (t - n_shift) % nfft; if (t < 0) t += nfft; }
        cplx v = buf[t];
        double win = 1.0;
        if (tt < nf || tt >= len - nf) {
            // hann(2 nf): first half evaluated, second half mirrored (MATLAB builds symmetric windows that way)
            int i = (tt < nf) ? tt : (2 * nf - 1 - (nf + (tt - (len - nf))));
            win = 0.5 - 0.5 * cos(2.0 * kPi * (double)i / (double)(2 * nf - 1));
        }
        v.x *= inv_n * win;
        v.y *= inv_n * win;
        if (out_cplx) {
            cplx* o = reinterpret_cast<cplx*>(e ? outR : outL);
            o[(size_t)c * len + tt] = v;
        } else {
            double* o = reinterpret_cast<double*>(e ? outR : outL);
            o[(size_t)c * len + tt] = v.x;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The same three transforms for an FFT length that is not a power of two (lib/getEMagLsFilters.m:44: nfft = min(2048, 2*len)
// for any even len), as direct DFTs: the HRIRs have ~128 taps and only len output samples of the inverse transform are kept,
// so O(N L) costs a few GFLOP at most.  Same outputs, same quirks (Nyquist phase forced real, DC rule, mirror rules).
// ---------------------------------------------------------------------------------------------
constexpr int HD_TD = 8;
__global__ void __launch_bounds__(256) hrir_dft_kernel(const double* __restrict__ hL, const double* __restrict__ hR, int64_t L, int64_t D,
                                                       const int64_t* __restrict__ didx, int nfft, const cplx* __restrict__ tw,
                                                       const double* __restrict__ grpd, int mode, int n_c, int kabs0, cplx* __restrict__ Hc,
                                                       double* __restrict__ Habs, int64_t ldD, double* __restrict__ HcT, int ldT, int td, size_t bstride) {
    hL = boff(hL, bstride); hR = boff(hR, bstride); didx = boff(didx, bstride); tw = boff(tw, bstride); grpd = boff(grpd, bstride);
    Hc = boff(Hc, bstride); Habs = boff(Habs, bstride); HcT = boff(HcT, bstride);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* hs = reinterpret_cast<double*>(smem);   // [td][2][L]
    const int P = nfft / 2 + 1;
    const int64_t d0 = (int64_t)blockIdx.x * td;
    const int nt = (int)min((int64_t)td, D - d0);
    for (int64_t idx = threadIdx.x; idx < (int64_t)nt * 2 * L; idx += blockDim.x) {
        const int t = (int)(idx / (2 * L)), e = (int)((idx / L) & 1);
        const int64_t n = idx % L;
        const int64_t dsrc = didx ? didx[d0 + t] : d0 + t;
        hs[idx] = (e ? hR : hL)[dsrc * L + n];
    }
    __syncthreads();
    const double g[2] = {grpd[0], grpd[1]};
    const int64_t na = P - kabs0;
    for (int kb = threadIdx.x; kb < P; kb += blockDim.x) {
        cplx ph[2];
        for (int e = 0; e < 2; ++e) {
            if (mode == 0) {   // exp(-1j*2*pi*omega*(-grpD)), Nyquist bin forced real (applySubsampleDelay.m:12)
                double sn, cs;
                sincos((6.283185307179586 * ((double)kb / (double)nfft)) * g[e], &sn, &cs);
                if (kb == P - 1) sn = 0.0;
                ph[e] = mk(cs, sn);
            } else {           // circshift by -round(grpD): exp(+2 pi i kb s / nfft), an exact table entry
                const int64_t sft = (int64_t)round(g[e]);
                int64_t j = (-(int64_t)kb * sft) % nfft;
                if (j < 0) j += nfft;
                ph[e] = tw[j];
            }
        }
        for (int t = 0; t < nt; ++t) {
            cplx H[2] = {mk(0.0, 0.0), mk(0.0, 0.0)};
            const double* h0 = hs + (int64_t)t * 2 * L;
            int j = 0;
            for (int64_t n = 0; n < L; ++n) {
                const cplx w = tw[j];
                cfma(H[0], h0[n], w); cfma(H[1], h0[L + n], w);
                j += kb; if (j >= nfft) j -= nfft;
            }
            const int64_t d = d0 + t;
            for (int e = 0; e < 2; ++e) {
                const cplx v = H[e] * ph[e];
                if (kb < n_c) {
                    Hc[((int64_t)e * n_c + kb) * ldD + d] = v;
                    if (HcT) { double* row = HcT + d * ldT; row[2 * (e * n_c + kb)] = v.x; row[2 * (e * n_c + kb) + 1] = v.y; }
                }
                if (kb >= kabs0) Habs[((int64_t)e * na + (kb - kabs0)) * ldD + d] = sqrt(norm2(v));
            }
        }
    }
}

__global__ void __launch_bounds__(256) real_dft_gather_kernel(const double* __restrict__ x, int64_t L, int64_t ncols, const int64_t* __restrict__ colidx,
                                                              int nfft, const cplx* __restrict__ tw, cplx* __restrict__ out, int64_t ldo, int64_t inner,
                                                              int64_t ld_inner) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* xs = reinterpret_cast<double*>(smem);   // [L]
    const int64_t j = blockIdx.x;
    const int P = nfft / 2 + 1;
    const double* col = x + (colidx ? colidx[j] : j) * L;
    for (int64_t n = threadIdx.x; n < L; n += blockDim.x) xs[n] = col[n];
    __syncthreads();
    for (int kb = threadIdx.x; kb < P; kb += blockDim.x) {
        cplx acc = mk(0.0, 0.0);
        int jj = 0;
        for (int64_t n = 0; n < L && n < nfft; ++n) { cfma(acc, xs[n], tw[jj]); jj += kb; if (jj >= nfft) jj -= nfft; }
        out[(int64_t)kb * ldo + (j / inner) * ld_inner + (j % inner)] = acc;
    }
}

__global__ void __launch_bounds__(256) filter_epilogue_dft_kernel(const cplx* __restrict__ W, int C, int nfft, int len, const cplx* __restrict__ tw,
                                                                  const double* __restrict__ grpd, int conj_mode, int dc_rule, int shift_mode, int out_cplx,
                                                                  double fade_rel, void* __restrict__ outL, void* __restrict__ outR, size_t bstride) {
    W = boff(W, bstride); tw = boff(tw, bstride); grpd = boff(grpd, bstride); outL = boff(outL, bstride); outR = boff(outR, bstride);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx* buf = reinterpret_cast<cplx*>(smem);   // the full spectrum, nfft entries
    const int c = blockIdx.x, e = blockIdx.y;
    const int P = nfft / 2 + 1;
    const cplx* We = W + (size_t)e * P * C;
    int cpart = c;
    double sgn = 1.0;
    if (conj_mode == 1) {
        const int n = (int)floor(sqrt((double)c));
        const int m = c - n * n - n;
        cpart = n * n + n - m;
        sgn = (m & 1) ? -1.0 : 1.0;
    } else if (conj_mode == 2) {
        cpart = (c == 0) ? 0 : ((c & 1) ? c + 1 : c - 1);
    }
    const int n_shift = nfft / 2;
    const double delay = (e == 0) ? (double)n_shift : ((double)n_shift + grpd[1]) - grpd[0];
    for (int k = threadIdx.x; k < nfft; k += blockDim.x) {
        cplx v;
        int kb;
        if (k < P) {
            kb = k;
            v = We[(size_t)kb * C + c];
            if (kb == 0 && dc_rule) v = mk(We[(size_t)1 * C + c].x, 0.0);
        } else {
            kb = nfft - k;
            v = conj(We[(size_t)kb * C + cpart]);
            v.x *= sgn; v.y *= sgn;
        }
        if (shift_mode == 0) {
            const double omega = (double)kb / (double)nfft;
            double sn, cs;
            sincos((6.283185307179586 * omega) * delay, &sn, &cs);
            cplx E = mk(cs, -sn);
            if (kb == P - 1) E.y = 0.0;
            if (k >= P) E.y = -E.y;
            v = v * E;
        }
        buf[k] = v;
    }
    __syncthreads();
    const double inv_n = 1.0 / (double)nfft;
    const int nf = (int)round(fade_rel * (double)len);
    const int t0 = n_shift - len / 2;
    for (int tt = threadIdx.x; tt < len; tt += blockDim.x) {
        int t = t0 + tt;
        if (shift_mode == 1) { t = (t - n_shift) % nfft; if (t < 0) t += nfft; }
        cplx v = mk(0.0, 0.0);
        int j = 0;
        for (int k = 0; k < nfft; ++k) { cfma(v, buf[k], conj(tw[j])); j += t; if (j >= nfft) j -= nfft; }   // ifft: exp(+2 pi i k t / nfft)
        double win = 1.0;
        if (tt < nf || tt >= len - nf) {
            int i = (tt < nf) ? tt : (2 * nf - 1 - (nf + (tt - (len - nf))));
            win = 0.5 - 0.5 * cos(2.0 * kPi * (double)i / (double)(2 * nf - 1));
        }
        v.x *= inv_n * win;
        v.y *= inv_n * win;
        if (out_cplx) reinterpret_cast<cplx*>(e ? outR : outL)[(size_t)c * len + tt] = v;
        else reinterpret_cast<double*>(e ? outR : outL)[(size_t)c * len + tt] = v.x;
    }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
static int ilog2(int n) { int l = 0; while ((1 << l) < n) ++l; return l; }
static bool is_pow2(int n) { return n > 0 && (n & (n - 1)) == 0; }

void launch_twiddles(int nfft, void* tw, hipStream_t st) {
    twiddle_kernel<<<bgrid((nfft + 255) / 256), 256, 0, st>>>(nfft, (cplx*)tw, batch_ctx().stride);
    KERNEL_CHECK();
}

int hrir_dirsum_chunks(int64_t D) { return (int)ceil_div(D, 64); }

void launch_hrir_grpdelay(const double* hL, const double* hR, int64_t L, int64_t D, int nfft, const void* tw,
                          double* partial, double* grpd, hipStream_t st) {
    const int chunk = 64;
    const int nchunks = hrir_dirsum_chunks(D);
    hrir_dirsum_kernel<<<bgrid(dim3(nchunks, 2)), 256, 0, st>>>(hL, hR, L, D, chunk, partial, batch_ctx().stride);
    KERNEL_CHECK();
    const int P = nfft / 2 + 1;
    int npow2 = 1;
    while (npow2 < P) npow2 <<= 1;
    size_t sm = (((size_t)L + 1) & ~(size_t)1) * 8 + (size_t)npow2 * 8 + (size_t)nfft * 16 + (size_t)std::max<int64_t>(1024, L) * 8;
    static PerDeviceOnce median_once;
    if (median_once.first()) HIP_CHECK(hipFuncSetAttribute((const void*)grpdelay_median_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    grpdelay_median_kernel<<<bgrid(2), 1024, sm, st>>>(partial, nchunks, L, nfft, (const cplx*)tw, grpd, batch_ctx().stride);
    KERNEL_CHECK();
}

// EMAGLS_HRIR_FFT_WAVE=0: the LDS radix-2^2 form at every length (read per call: a test compares the two in one process)
static bool hrir_fft_wave_enabled() { const char* e = getenv("EMAGLS_HRIR_FFT_WAVE"); return !(e && e[0] == '0'); }

void launch_hrir_fft(const double* hL, const double* hR, int64_t L, int64_t D, const int64_t* didx, int nfft,
                     const void* tw, const double* grpd, int mode, int n_c, int kabs0, void* Hc, double* Habs,
                     int64_t ldD, hipStream_t st, double* HcT, int ldT) {
    if (!is_pow2(nfft)) {   // direct DFT (hrir_dft_kernel)
        int td = HD_TD;
        while (td > 1 && (size_t)td * 2 * L * 8 > 48 * 1024) td >>= 1;
        if ((size_t)td * 2 * L * 8 > 64 * 1024) throw Error(2, "HRIR DFT: more than 4096 taps is not supported");
        hrir_dft_kernel<<<bgrid((unsigned)ceil_div(D, td)), 256, (size_t)td * 2 * L * 8, st>>>(hL, hR, L, D, didx, nfft, (const cplx*)tw, grpd, mode, n_c, kabs0,
                                                                                                (cplx*)Hc, Habs, ldD, HcT, ldT, td, batch_ctx().stride);
        KERNEL_CHECK();
        return;
    }
    const int log2n = ilog2(nfft);
    const int P = nfft / 2 + 1;
    if (P > HF_MAXS * 512) throw Error(2, "HRIR FFT: nfft above 2048 is not supported");
    if (nfft == WF_N && L <= WF_N && hrir_fft_wave_enabled()) {   // wave-private transforms
        const size_t smw = sizeof(cplx) * (WF_TABLES + (size_t)HW_TD * HW_STRIDE);
        static PerDeviceOnce wave_once;
        if (wave_once.first())
            HIP_CHECK(hipFuncSetAttribute((const void*)hrir_fft_wave_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hrir_fft_wave_kernel<<<bgrid((unsigned)ceil_div(D, HW_TD)), 64 * HW_TD, smw, st>>>(hL, hR, L, D, didx, (const cplx*)tw, grpd, mode, n_c, kabs0,
                                                                                      (cplx*)Hc, Habs, ldD, HcT, ldT, batch_ctx().stride);
        KERNEL_CHECK();
        return;
    }
    // directions per sub-tile: the largest power of two whose buffers stay below the LDS a CU has left next to a resident
    // sweep workgroup (160 KB - 77 KB)
    // (tried: 2 or 1 directions per sub-tile, 49 / 33 KB, so that three or four workgroups share a CU and the kernel fits next to a
    // twin sweep workgroup: 2062 / 2047 against 2174 sets/s in long runs, 1393 / 1633 against 1697 at 20 steps; all 8 directions in
    // one sub-tile, 148 KB: 2149-2152 against 2176-2186)
    // (nfft = 2048 -- configs 4 and 5 -- gets two directions per sub-tile, 98 KB: nothing fits next to the round-4 sweep's twin
    // workgroups anyway, and four sub-tiles instead of eight are 2-3 % of a config 4 / 5 batch)
    size_t budget = (nfft >= 2048 ? 110 : 83) * 1024;
    if (const char* e = getenv("EMAGLS_HRIR_FFT_LDS_KB")) budget = (size_t)std::max(40, std::min(158, atoi(e))) * 1024;
    const size_t shared = (size_t)2 * P * 16;
    int TS = HF_TD;
    while (TS > 1 && (size_t)TS * nfft * 16 + shared > budget) TS >>= 1;
    const size_t sm = (size_t)TS * nfft * 16 + shared;
    static PerDeviceOnce attr_once;   // (function attributes are per device)
    if (attr_once.first()) {
        HIP_CHECK(hipFuncSetAttribute((const void*)hrir_fft_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_CHECK(hipFuncSetAttribute((const void*)hrir_fft_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_CHECK(hipFuncSetAttribute((const void*)hrir_fft_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_CHECK(hipFuncSetAttribute((const void*)hrir_fft_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        HIP_CHECK(hipFuncSetAttribute((const void*)real_fft_gather_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    const dim3 grid = bgrid((unsigned)ceil_div(D, HF_TD));
#define EMAGLS_HF(TSV)                                                                                                               \
    hrir_fft_kernel<TSV><<<grid, 512, sm, st>>>(hL, hR, L, D, didx, nfft, log2n, (const cplx*)tw, grpd, mode, n_c, kabs0, (cplx*)Hc, \
                                                Habs, ldD, HcT, ldT, batch_ctx().stride)
    switch (TS) {
        case 8: EMAGLS_HF(8); break;
        case 4: EMAGLS_HF(4); break;
        case 2: EMAGLS_HF(2); break;
        default: EMAGLS_HF(1); break;
    }
#undef EMAGLS_HF
    KERNEL_CHECK();
}

void launch_real_fft_gather(const double* x, int64_t L, int64_t ncols, const int64_t* colidx, int nfft, const void* tw,
                            void* out, int64_t ldo, int64_t inner, int64_t ld_inner, hipStream_t st) {
    if (!is_pow2(nfft)) {   // direct DFT, one workgroup per column
        if ((size_t)L * 8 > 64 * 1024) throw Error(2, "ATF DFT: more than 8192 taps is not supported");
        real_dft_gather_kernel<<<(unsigned)ncols, 256, (size_t)L * 8, st>>>(x, L, ncols, colidx, nfft, (const cplx*)tw, (cplx*)out, ldo, inner, ld_inner);
        KERNEL_CHECK();
        return;
    }
    const int log2n = ilog2(nfft);
    int TP = 8;
    while (TP > 1 && (size_t)TP * nfft * 16 + (size_t)nfft * 8 > 150 * 1024) TP >>= 1;
    const size_t sm = (size_t)TP * nfft * 16 + (size_t)nfft * 8;
    static PerDeviceOnce attr_once;   // (function attributes are per device)
    if (attr_once.first()) {
        HIP_CHECK(hipFuncSetAttribute((const void*)real_fft_gather_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    const int64_t npairs = (ncols + 1) / 2;
    real_fft_gather_kernel<<<(unsigned)ceil_div(npairs, TP), 512, sm, st>>>(x, L, ncols, colidx, nfft, log2n, TP,
                                                                          (const cplx*)tw, (cplx*)out, ldo, inner, ld_inner);
    KERNEL_CHECK();
}

void launch_filter_epilogue(const void* W, int C, int nfft, int len, const void* tw, const double* grpd, int conj_mode,
                            int dc_rule, int shift_mode, int out_cplx, void* outL, void* outR, hipStream_t st, int n_ears,
                            double fade_rel) {
    if (!is_pow2(nfft)) {   // direct inverse DFT of the len samples that are kept
        const size_t smd = (size_t)nfft * 16;
        if (smd > 64 * 1024) throw Error(2, "filter epilogue: a non-power-of-two FFT length above 4096 is not supported");
        filter_epilogue_dft_kernel<<<bgrid(dim3(C, n_ears)), 256, smd, st>>>((const cplx*)W, C, nfft, len, (const cplx*)tw, grpd, conj_mode, dc_rule, shift_mode,
                                                                           out_cplx, fade_rel, outL, outR, batch_ctx().stride);
        KERNEL_CHECK();
        return;
    }
    const int log2n = ilog2(nfft);
    const size_t sm = (size_t)nfft * 16 + (size_t)nfft * 8;
    if (sm > 48 * 1024) {   // (only the radial-filter IRs get here: the designs stop at nfft = 2048)
        static PerDeviceOnce attr_once;   // (function attributes are per device)
        if (attr_once.first()) {
            HIP_CHECK(hipFuncSetAttribute((const void*)filter_epilogue_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
    }
    filter_epilogue_kernel<<<bgrid(dim3(C, n_ears)), 256, sm, st>>>((const cplx*)W, C, nfft, log2n, len, (const cplx*)tw, grpd,
                                                        conj_mode, dc_rule, shift_mode, out_cplx, fade_rel, outL, outR, batch_ctx().stride);
    KERNEL_CHECK();
}

}  // namespace emagls
