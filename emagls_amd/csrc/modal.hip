// Rigid-sphere modal coefficients b_n(kr) (replaces polarch sphModalCoeffs; call site
// dependencies/getSMAIRMatrix.m:107).  One thread per kr value, orders 0..N by upward recurrence.
//
//   b_n = 4 pi i^n ( j_n - j_n'/h_n^(2)' h_n^(2) )(kr) = 4 pi i^(n-1) / ( (kr)^2 h_n^(2)'(kr) )      (Wronskian)
//
// Only h_n^(2)' = j_n' - i y_n' is needed.  y_n is stable under upward recurrence; j_n is not for
// n > kr, but its error is O(eps |y_n|), i.e. O(eps) relative to |h_n'|, so b_n keeps full relative
// accuracy as a complex number.  kr == 0 -> [4 pi, 0, ...]; overflow of y_n -> 0 (the reference
// library zeroes the NaNs it gets there).
#include "kernels.hpp"

namespace emagls {

__global__ void __launch_bounds__(256) modal_bn_kernel(int N, int64_t nfreq, const double* __restrict__ kr,
                                                       double kr_scale, double out_scale, cplx* __restrict__ bn,
                                                       int64_t stride_k, int64_t stride_n, const int* __restrict__ n_valid,
                                                       size_t bstride) {
    kr = boff(kr, bstride); bn = boff(bn, bstride); n_valid = boff(n_valid, bstride);
    // orders above n_valid are written as zeros: a design simulated at a padded order (a lane batch across neighbouring
    // simulation-order classes) sums the same terms as at its own order (dependencies/getSMAIRMatrix.m:95,107)
    const int nv = n_valid ? *n_valid : N;
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nfreq) return;
    // kr == nullptr: kr_k = k * kr_scale (the FFT-bin grid 2 pi f_k r / c)
    const double x = kr ? kr[k] * kr_scale : (double)k * kr_scale;
    if (x == 0.0) {
        for (int n = 0; n <= N; ++n) bn[k * stride_k + n * stride_n] = mk(n == 0 ? out_scale * 4.0 * kPi : 0.0, 0.0);
        return;
    }
    double sx, cx;
    sincos(x, &sx, &cx);
    const double ix = 1.0 / x;
    double jm = sx * ix, ym = -cx * ix;                       // order 0
    double j = (sx * ix - cx) * ix, y = (-cx * ix - sx) * ix;  // order 1
    const double c0 = out_scale * 4.0 * kPi / (x * x);
    bool dead = false;
    // n = 0: f_0' = -f_1
    {
        cplx dh = mk(-j, y);  // j0' - i y0' = -j1 + i y1
        bn[k * stride_k] = cdiv(mk(0.0, -c0), dh);  // i^0 * (-i) = -i
    }
    for (int n = 1; n <= N; ++n) {
        // here j,y are order n; jm,ym order n-1
        cplx val = mk(0.0, 0.0);
        if (n > nv) dead = true;
        if (!dead) {
            const double dj = jm - (n + 1) * ix * j;
            const double dy = ym - (n + 1) * ix * y;
            if (!isfinite(dy) || !isfinite(y)) {
                dead = true;
            } else {
                cplx dh = mk(dj, -dy);
                cplx num;
                switch (n & 3) {  // i^n * (-i)
                    case 0: num = mk(0.0, -c0); break;
                    case 1: num = mk(c0, 0.0); break;
                    case 2: num = mk(0.0, c0); break;
                    default: num = mk(-c0, 0.0); break;
                }
                val = cdiv(num, dh);
                if (!isfinite(val.x) || !isfinite(val.y)) val = mk(0.0, 0.0);
            }
        }
        bn[k * stride_k + n * stride_n] = val;
        const double jn = (2 * n + 1) * ix * j - jm;
        const double yn = (2 * n + 1) * ix * y - ym;
        jm = j; ym = y; j = jn; y = yn;
    }
}

void launch_modal_bn(int N, int64_t nfreq, const double* kr, double kr_scale, double out_scale, void* bn,
                     int64_t stride_k, int64_t stride_n, hipStream_t st, const int* n_valid) {
    if (nfreq <= 0) return;
    modal_bn_kernel<<<bgrid((unsigned)ceil_div(nfreq, 256)), 256, 0, st>>>(N, nfreq, kr, kr_scale, out_scale, (cplx*)bn,
                                                                    stride_k, stride_n, n_valid, batch_ctx().stride);
    KERNEL_CHECK();
}

}  // namespace emagls
