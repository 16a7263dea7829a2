// Render-side neighbours of the filter designs: everything the reference's harness runs between the
// microphone recording and the binaural decoder, plus the two optional equalisation filters.
//
//   getRadialFilter             dependencies/getRadialFilter.m:25-71     (plane-wave model: tikhonov / softlimit / full)
//   applyRadialFilter           dependencies/applyRadialFilter.m:9-31    (the impulse responses; the convolution is in decode.hip)
//   SH encoding                 verifyEMagLs.m:235-236                   sig * pinv(Y_mic.')  ==  sig * pinv(Y_mic).'
//   getMagLsSphericalHeadFilter lib/getMagLsSphericalHeadFilter.m:28-49
//   getMagLsArrayDiffuseFilter  lib/getMagLsArrayDiffuseFilter.m:40-74
//
// All of it is elementwise or a thin product on top of b_n(kr) (modal.hip): HBM-bound, a few kB per call.
#include "kernels.hpp"

namespace emagls {

// rad[k][n] from b_n(kr_k), written twice: [k][n] (the layout the IR epilogue reads) and [n][P] column-major for the caller.
//   type 0 tikhonov   conj(b) / (conj(b) b + regul)
//   type 1 softlimit  2g/pi |b|/b atan(pi / (2 g |b|))
//   type 2 full       1 / b
//   type 3 none       1
// The last bin is replaced by its magnitude when nfft is even (:68-70).  b = 0 (orders > 0 at DC) gives NaN for softlimit
// and Inf + NaN i for full, as IEEE arithmetic does in the reference; the IR path zeroes them (applyRadialFilter.m:14).
__global__ void __launch_bounds__(256) radial_filter_kernel(const cplx* __restrict__ bn, int nOrd, int P, int type, double regul,
                                                            double g, int nyq_abs, int zero_nan, cplx* __restrict__ out_kn,
                                                            cplx* __restrict__ out_cm) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= P * nOrd) return;
    const int k = idx / nOrd, n = idx % nOrd;
    const cplx b = bn[idx];
    cplx r;
    if (type == 0) {
        const double den = norm2(b) + regul;
        r = mk(b.x / den, -b.y / den);
    } else if (type == 1) {
        const double a = cabs(b);
        const double s = 2.0 * g / kPi * atan(kPi / (2.0 * g * a));   // atan(inf) = pi/2 at a = 0 ...
        const cplx u = cdiv(mk(a, 0.0), b);                           // ... and 0/0 = NaN here
        r = mk(s * u.x, s * u.y);
        if (a == 0.0) r = mk(nan(""), nan(""));
    } else if (type == 2) {
        r = cdiv(mk(1.0, 0.0), b);
        if (b.x == 0.0 && b.y == 0.0) r = mk(INFINITY, nan(""));
    } else {
        r = mk(1.0, 0.0);
    }
    if (nyq_abs && k == P - 1) r = mk(cabs(r), 0.0);
    if (out_cm) out_cm[(size_t)n * P + k] = r;
    if (out_kn) {
        if (zero_nan && (isnan(r.x) || isnan(r.y))) r = mk(0.0, 0.0);
        out_kn[idx] = r;
    }
}
void launch_radial_filter(const void* bn, int nOrd, int P, int type, double regul, double g, bool nyq_abs, bool zero_nan,
                          void* out_kn, void* out_cm, hipStream_t st) {
    radial_filter_kernel<<<ceil_div(P * nOrd, 256), 256, 0, st>>>((const cplx*)bn, nOrd, P, type, regul, g, nyq_abs ? 1 : 0,
                                                                  zero_nan ? 1 : 0, (cplx*)out_kn, (cplx*)out_cm);
    KERNEL_CHECK();
}

// Diffuse-field responses of the order-expanded modal coefficients: rms(abs(sh_repToOrder(b)), 2) * sqrt(#SH) / (4 pi)
//   = sqrt( sum_n (2n+1) |b_n|^2 ) / (4 pi)      (getMagLsSphericalHeadFilter.m:37-42)
// df_hi over all nOrd orders, df_lo over the first n_lo.  One thread per bin.
__global__ void __launch_bounds__(256) diffuse_field_kernel(const cplx* __restrict__ bn, int nOrd, int n_lo, int P,
                                                            double* __restrict__ df_hi, double* __restrict__ df_lo) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P) return;
    double hi = 0.0, lo = 0.0;
    for (int n = 0; n < nOrd; ++n) {
        hi = fma((double)(2 * n + 1), norm2(bn[(size_t)k * nOrd + n]), hi);
        if (n == n_lo - 1) lo = hi;
    }
    df_hi[k] = sqrt(hi) / (4.0 * kPi);
    if (df_lo) df_lo[k] = sqrt(lo) / (4.0 * kPi);
}
void launch_diffuse_field(const void* bn, int nOrd, int n_lo, int P, double* df_hi, double* df_lo, hipStream_t st) {
    diffuse_field_kernel<<<ceil_div(P, 256), 256, 0, st>>>((const cplx*)bn, nOrd, n_lo, P, df_hi, df_lo);
    KERNEL_CHECK();
}

// Diffuse-field response of the array as it is encoded at the low order (getMagLsArrayDiffuseFilter.m:47-56):
//   bn_Lo(k,:) = ( bn_Hi(k,:) Y_Hi' ) Y_Lo,   df_lo[k] = sqrt( sum_c |bn_Lo(k,c)|^2 ) / (4 pi)
// One workgroup per bin: t[m] = sum_s b_n(s) conj(Y[m][s]) into LDS, then one wave per output channel.
// Y is [S][ldY] (harmonic-major, the layout sh_basis writes); the low-order matrix is its first nOut rows.
template <typename T>
__global__ void __launch_bounds__(256) array_diffuse_kernel(const cplx* __restrict__ bn, int nOrd, const T* __restrict__ Y, int ldY,
                                                            int S, int M, int nOut, double* __restrict__ df_lo) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx* t = reinterpret_cast<cplx*>(smem);
    __shared__ double acc[4];
    const int k = blockIdx.x;
    const cplx* b = bn + (size_t)k * nOrd;
    for (int m = threadIdx.x; m < M; m += blockDim.x) {
        cplx v = mk(0.0, 0.0);
        int s = 0;
        for (int n = 0; n < nOrd; ++n) {
            cplx u = mk(0.0, 0.0);                       // sum over the 2n+1 harmonics of order n, then one multiply by b_n
            for (int j = 0; j < 2 * n + 1; ++j, ++s) {
                const T y = Y[(size_t)s * ldY + m];
                u += mk(1.0, 0.0) * conj(y);
            }
            cfma(v, b[n], u);
        }
        t[m] = v;
    }
    if (threadIdx.x < 4) acc[threadIdx.x] = 0.0;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double a = 0.0;
    for (int c = wave; c < nOut; c += 4) {
        cplx v = mk(0.0, 0.0);
        for (int m = lane; m < M; m += 64) cfma(v, t[m], Y[(size_t)c * ldY + m]);
        for (int off = 32; off > 0; off >>= 1) {
            v.x += __shfl_down(v.x, off);
            v.y += __shfl_down(v.y, off);
        }
        a += norm2(v);
    }
    if (lane == 0) acc[wave] = a;
    __syncthreads();
    if (threadIdx.x == 0) df_lo[k] = sqrt(acc[0] + acc[1] + acc[2] + acc[3]) / (4.0 * kPi);
}
void launch_array_diffuse(const void* bn, int nOrd, const void* Y, bool y_cplx, int ldY, int S, int M, int nOut, int P,
                          double* df_lo, hipStream_t st) {
    const size_t sm = (size_t)M * sizeof(cplx);
    if (sm > 64 * 1024) throw Error(2, "array diffuse-field filter: more than 4096 microphones is not supported in this build");
    if (y_cplx)
        array_diffuse_kernel<cplx><<<P, 256, sm, st>>>((const cplx*)bn, nOrd, (const cplx*)Y, ldY, S, M, nOut, df_lo);
    else
        array_diffuse_kernel<double><<<P, 256, sm, st>>>((const cplx*)bn, nOrd, (const double*)Y, ldY, S, M, nOut, df_lo);
    KERNEL_CHECK();
}

// Equalisation spectra as the IR epilogue reads them (one channel, [k] complex with zero imaginary part):
//   mode 0 (spherical head):  W = 1 / (df_hi / df_lo)                                  (SphericalHeadFilter.m:45-48)
//   mode 1 (array diffuse):   W = W_Shf * df_hi / (df_arr / df_arr[0])                 (ArrayDiffuseFilter.m:59-66)
// and, optionally, the real spectrum mirrored to nfft bins (the second output of getMagLsSphericalHeadFilter).
__global__ void __launch_bounds__(256) eq_spectrum_kernel(const double* __restrict__ df_hi, const double* __restrict__ df_lo,
                                                          const double* __restrict__ df_arr, int P, int mode,
                                                          cplx* __restrict__ W, double* __restrict__ W_full) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= P) return;
    double w = 1.0 / (df_hi[k] / df_lo[k]);
    if (mode == 1) w = w * (df_hi[k] / (df_arr[k] / df_arr[0]));
    W[k] = mk(w, 0.0);
    if (W_full) {
        W_full[k] = w;
        const int nfft = 2 * (P - 1);
        if (k > 0 && k < P - 1) W_full[nfft - k] = w;
    }
}
void launch_eq_spectrum(const double* df_hi, const double* df_lo, const double* df_arr, int P, int mode, void* W, double* W_full,
                        hipStream_t st) {
    eq_spectrum_kernel<<<ceil_div(P, 256), 256, 0, st>>>(df_hi, df_lo, df_arr, P, mode, (cplx*)W, W_full);
    KERNEL_CHECK();
}

// SH encoding: out[c][t] = sum_m sig[m][t] Z[c][m]  with Z = pinv(Y_mic) as [c][ldZ] complex (the factorisation's output).
// One thread per sample and chunk of 8 channels; Z is read wave-uniformly.  out is real (real basis) or complex.
constexpr int ENC_CH = 8;
template <bool OUT_CPLX>
__global__ void __launch_bounds__(256) sh_encode_kernel(const double* __restrict__ sig, int64_t n, int M, const cplx* __restrict__ Z,
                                                        int ldZ, int nOut, void* __restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int c0 = blockIdx.y * ENC_CH;
    if (t >= n) return;
    cplx acc[ENC_CH];
#pragma unroll
    for (int j = 0; j < ENC_CH; ++j) acc[j] = mk(0.0, 0.0);
    for (int m = 0; m < M; ++m) {
        const double x = sig[(int64_t)m * n + t];
#pragma unroll
        for (int j = 0; j < ENC_CH; ++j) {
            if (c0 + j < nOut) {
                const cplx z = Z[(size_t)(c0 + j) * ldZ + m];
                acc[j].x = fma(x, z.x, acc[j].x);
                if (OUT_CPLX) acc[j].y = fma(x, z.y, acc[j].y);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < ENC_CH; ++j) {
        if (c0 + j >= nOut) break;
        if (OUT_CPLX) reinterpret_cast<cplx*>(out)[(int64_t)(c0 + j) * n + t] = acc[j];
        else reinterpret_cast<double*>(out)[(int64_t)(c0 + j) * n + t] = acc[j].x;
    }
}
void launch_sh_encode(const double* sig, int64_t n, int M, const void* Z, int ldZ, int nOut, bool out_cplx, void* out, hipStream_t st) {
    if (n <= 0) return;
    dim3 grid((unsigned)ceil_div(n, 256), (unsigned)ceil_div(nOut, ENC_CH));
    if (out_cplx) sh_encode_kernel<true><<<grid, 256, 0, st>>>(sig, n, M, (const cplx*)Z, ldZ, nOut, out);
    else sh_encode_kernel<false><<<grid, 256, 0, st>>>(sig, n, M, (const cplx*)Z, ldZ, nOut, out);
    KERNEL_CHECK();
}

// ---------------------------------------------------------------------------------------------
// Diffuseness (covariance) constraint, SURVEY 8(f) rank 1 (absent from the reference snapshot; specification and what the
// *_wDC fixtures pin of it: DESIGN.md section 7 and the CPU restatement under oracle/).  One workgroup per solved bin:
//   Hhat_e(d) = W_e(k,:) pwGrid_k(:,d)          rendered HRTFs over the HRIR grid
//   Rhat = E_d[conj(Hhat_i) Hhat_j],  R = E_d[conj(H_i) H_j]   (2x2, the time-aligned HRTFs H)
//   M = the Hermitian positive definite solution of M Rhat M = R,   W(k,:,[l r]) <- W(k,:,[l r]) M
// G is pwGrid_k.' as [c][ldD] per bin (g_stride elements between bins; 0: one matrix for every bin, MagLS' Y_conj),
// W [e][P][C], H [e][P][ldD].  sqrt of a 2x2 Hermitian PSD matrix in closed form: (A + sqrt(det) I) / sqrt(tr + 2 sqrt(det)).
// ---------------------------------------------------------------------------------------------
struct Herm2 { double a, d; cplx b; };   // [[a, b], [conj(b), d]]
__device__ __forceinline__ Herm2 sqrt_herm2(Herm2 A) {
    const double det = fmax(A.a * A.d - norm2(A.b), 0.0);
    const double s = sqrt(det), t = sqrt(fmax(A.a + A.d + 2.0 * s, 1e-300));
    return Herm2{(A.a + s) / t, (A.d + s) / t, mk(A.b.x / t, A.b.y / t)};
}
__device__ __forceinline__ Herm2 congruence2(Herm2 S, Herm2 R) {   // S R S for Hermitian S, R
    // X = S R
    const cplx x00 = mk(S.a * R.a, 0.0) + S.b * conj(R.b), x01 = S.a * R.b + S.b * R.d;
    const cplx x10 = conj(S.b) * R.a + S.d * conj(R.b), x11 = conj(S.b) * R.b + mk(S.d * R.d, 0.0);
    // Y = X S
    const cplx y00 = x00 * S.a + x01 * conj(S.b), y01 = x00 * S.b + x01 * S.d;
    const cplx y11 = x10 * S.b + x11 * S.d;
    (void)x10;
    return Herm2{y00.x, y11.x, y01};
}
template <typename T>
__global__ void __launch_bounds__(256) diffuse_constraint_kernel(cplx* __restrict__ W, const T* __restrict__ G, int64_t g_stride, int g0,
                                                                 const cplx* __restrict__ H, int D, int C, int64_t ldD, int P,
                                                                 size_t bstride) {
    W = boff(W, bstride); G = boff(G, bstride); H = boff(H, bstride);
    __shared__ cplx w_s[2][64];
    __shared__ double red[4][8];
    __shared__ cplx m_s[4];
    const int kb = blockIdx.x + 1;
    cplx* Wl = W + (int64_t)kb * C;
    cplx* Wr = W + ((int64_t)P + kb) * C;
    if (threadIdx.x < C) { w_s[0][threadIdx.x] = Wl[threadIdx.x]; w_s[1][threadIdx.x] = Wr[threadIdx.x]; }
    __syncthreads();
    const T* Gk = G + (int64_t)(kb - g0) * g_stride;
    const cplx* Hl = H + (int64_t)kb * ldD;
    const cplx* Hr = H + ((int64_t)P + kb) * ldD;
    double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // Rhat: a, d, Re b, Im b;  R: a, d, Re b, Im b
    for (int d = threadIdx.x; d < D; d += 256) {
        cplx hl = mk(0, 0), hr = mk(0, 0);
        for (int c = 0; c < C; ++c) {
            const T g = Gk[(int64_t)c * ldD + d];
            cfma(hl, w_s[0][c], g);
            cfma(hr, w_s[1][c], g);
        }
        acc[0] += norm2(hl); acc[1] += norm2(hr);
        acc[2] += hl.x * hr.x + hl.y * hr.y; acc[3] += hl.x * hr.y - hl.y * hr.x;     // conj(hl) hr
        const cplx tl = Hl[d], tr = Hr[d];
        acc[4] += norm2(tl); acc[5] += norm2(tr);
        acc[6] += tl.x * tr.x + tl.y * tr.y; acc[7] += tl.x * tr.y - tl.y * tr.x;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        double v = acc[i];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double r[8];
        for (int i = 0; i < 8; ++i) r[i] = (red[0][i] + red[1][i] + red[2][i] + red[3][i]) / (double)D;
        const Herm2 Rh{r[0], r[1], mk(r[2], r[3])}, R{r[4], r[5], mk(r[6], r[7])};
        const Herm2 S = sqrt_herm2(Rh);
        const double dS = S.a * S.d - norm2(S.b);
        cplx M[4] = {mk(1, 0), mk(0, 0), mk(0, 0), mk(1, 0)};
        if (dS > 0.0 && isfinite(dS)) {
            const Herm2 Si{S.d / dS, S.a / dS, mk(-S.b.x / dS, -S.b.y / dS)};   // inverse of a Hermitian 2x2
            const Herm2 Q = sqrt_herm2(congruence2(S, R));
            const Herm2 Mh = congruence2(Si, Q);
            M[0] = mk(Mh.a, 0.0); M[1] = Mh.b; M[2] = conj(Mh.b); M[3] = mk(Mh.d, 0.0);
        }
        m_s[0] = M[0]; m_s[1] = M[1]; m_s[2] = M[2]; m_s[3] = M[3];
    }
    __syncthreads();
    if (threadIdx.x < C) {
        const cplx wl = w_s[0][threadIdx.x], wr = w_s[1][threadIdx.x];
        Wl[threadIdx.x] = wl * m_s[0] + wr * m_s[2];
        Wr[threadIdx.x] = wl * m_s[1] + wr * m_s[3];
    }
}
void launch_diffuse_constraint(void* W, const void* G, bool g_cplx, int64_t g_stride, int g0, const void* H, int D, int C, int64_t ldD,
                               int P, hipStream_t st) {
    if (P < 2) return;
    if (C > 64) throw Error(2, "diffuseness constraint: more than 64 channels is not supported");
    if (g_cplx)
        diffuse_constraint_kernel<cplx><<<bgrid(P - 1), 256, 0, st>>>((cplx*)W, (const cplx*)G, g_stride, g0, (const cplx*)H, D, C, ldD, P,
                                                                       batch_ctx().stride);
    else
        diffuse_constraint_kernel<double><<<bgrid(P - 1), 256, 0, st>>>((cplx*)W, (const double*)G, g_stride, g0, (const cplx*)H, D, C, ldD, P,
                                                                         batch_ctx().stride);
    KERNEL_CHECK();
}

// getSMAIRMatrix materialised (dependencies/getSMAIRMatrix.m:110-140) for callers that want the array model itself; the filter
// designs never form it.  out[(k S + s) rows + c] = rad_n(c)(k) E[c][s] b_n(s)(k), the last bin with
// real(b_n) (:115-117) -- MATLAB's [rows x S x P] column-major array.  rad (optional, [P][nOut orders]) are the radial filters
// applied to the SH-domain model (:129-138): r at every bin and, literally like the reference, r real(r) at the Nyquist bin.
template <typename T>
__global__ void __launch_bounds__(256) smair_kernel(const T* __restrict__ E, int ldS, const cplx* __restrict__ bn, int nOrd,
                                                    const cplx* __restrict__ rad, int nRad, int rows, int S, int P, cplx* __restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)rows * S * P) return;
    const int c = (int)(idx % rows);
    const int64_t ks = idx / rows;
    const int s = (int)(ks % S), k = (int)(ks / S);
    int n = 0;
    while ((n + 1) * (n + 1) <= s) ++n;
    cplx b = bn[(size_t)k * nOrd + n];
    if (k == P - 1) b.y = 0.0;
    cplx v = mk(1.0, 0.0) * E[(size_t)c * ldS + s];
    v = v * b;
    if (rad) {
        int nc = 0;
        while ((nc + 1) * (nc + 1) <= c) ++nc;
        const cplx r = rad[(size_t)k * nRad + nc];
        v = r * v;
        // the reference applies the filter a SECOND time at the Nyquist bin: smairMat(:,:,k) = BnTi * smairMat(:,:,k) and then, for
        // k == numPosFreqs, smairMat(:,:,k) = real(BnTi) * smairMat(:,:,k) on the already filtered slice (getSMAIRMatrix.m:134-137)
        if (k == P - 1) v = r.x * v;
    }
    out[idx] = v;
}
void launch_smair(const void* E, bool e_cplx, int ldS, const void* bn, int nOrd, const void* rad, int nRad, int rows, int S, int P, void* out,
                  hipStream_t st) {
    const unsigned g = (unsigned)ceil_div((int64_t)rows * S * P, 256);
    if (e_cplx) smair_kernel<cplx><<<g, 256, 0, st>>>((const cplx*)E, ldS, (const cplx*)bn, nOrd, (const cplx*)rad, nRad, rows, S, P, (cplx*)out);
    else smair_kernel<double><<<g, 256, 0, st>>>((const double*)E, ldS, (const cplx*)bn, nOrd, (const cplx*)rad, nRad, rows, S, P, (cplx*)out);
    KERNEL_CHECK();
}

}  // namespace emagls
