// Register-resident FP64 FFTs of length 16 / 32 (one thread = one transform, fully unrolled radix-2 decimation in frequency) for
// the two-factor overlap-save kernel of decode.hip: a length-N1*N2 transform is N2 thread-local transforms of length N1, one
// twiddle multiplication, ONE transposition through LDS and N1 thread-local transforms of length N2 -- two LDS exchanges per
// transform instead of the five read-modify-write passes of the radix-2^2 LDS form (lds_fft.hpp), which is instruction bound.
#pragma once
#include "kernels.hpp"

namespace emagls {

// v[i] <- X[bitrev_R(i)],  X[k] = sum_n v[n] exp(-2 pi i n k / R)     (R = 2, 4, 8, 16, 32)
template <int R> __device__ __forceinline__ void reg_fft(cplx (&v)[R]) {
    // W_32^k = exp(-2 pi i k / 32), k < 16
    constexpr double C32[16] = {1.0, 0.9807852804032304, 0.9238795325112867, 0.8314696123025452, 0.7071067811865476, 0.5555702330196023,
                                0.38268343236508984, 0.19509032201612833, 0.0, -0.1950903220161282, -0.3826834323650897, -0.555570233019602,
                                -0.7071067811865475, -0.8314696123025453, -0.9238795325112867, -0.9807852804032304};
    constexpr double S32[16] = {0.0, -0.19509032201612825, -0.3826834323650898, -0.5555702330196022, -0.7071067811865475, -0.8314696123025452,
                                -0.9238795325112867, -0.9807852804032304, -1.0, -0.9807852804032304, -0.9238795325112867, -0.8314696123025455,
                                -0.7071067811865476, -0.5555702330196022, -0.3826834323650899, -0.1950903220161286};
    static_assert(R == 2 || R == 4 || R == 8 || R == 16 || R == 32, "reg_fft length");
#pragma unroll
    for (int span = R / 2; span >= 1; span >>= 1) {
#pragma unroll
        for (int i = 0; i < R; ++i) {
            if ((i & span) == 0) {                       // (a compile-time condition once the loops are unrolled)
                const int j = i + span;
                const int k = (i & (span - 1)) * (16 / span);   // twiddle W_{2 span}^{i mod span} = W_32^k
                const cplx a = v[i], b = v[j];
                v[i] = mk(a.x + b.x, a.y + b.y);
                const double tx = a.x - b.x, ty = a.y - b.y;
                if (k == 0) v[j] = mk(tx, ty);
                else if (k == 8) v[j] = mk(ty, -tx);     // times -i
                else v[j] = mk(tx * C32[k] - ty * S32[k], tx * S32[k] + ty * C32[k]);
            }
        }
    }
}
template <int BITS> __host__ __device__ constexpr int rf_bitrev(int i) {
    int r = 0;
    for (int b = 0; b < BITS; ++b) r |= ((i >> b) & 1) << (BITS - 1 - b);
    return r;
}
template <int R> __host__ __device__ constexpr int rf_log2() { return R == 2 ? 1 : R == 4 ? 2 : R == 8 ? 3 : R == 16 ? 4 : 5; }

}  // namespace emagls
