// Grid matching for the ATF-based design (lib/getEMagLsFiltersFromAtf.m:56-95): for every point of
// the smaller grid find the nearest point (Euclidean distance of unit vectors, first index on ties
// like MATLAB's min) of the larger grid, and the angular deviation in degrees whose mean the
// reference prints (:96).
#include "kernels.hpp"

namespace emagls {

__device__ __forceinline__ void sph2cart_unit(double azi, double zen, double* v) {
    const double ele = kPi / 2 - zen;  // sph2cart(azi, pi/2 - zen, 1)
    const double ce = cos(ele);
    v[0] = ce * cos(azi);
    v[1] = ce * sin(azi);
    v[2] = sin(ele);
}

__global__ void cart_kernel(const double* __restrict__ azi, const double* __restrict__ zen, int64_t n, double* __restrict__ cart) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v[3];
    sph2cart_unit(azi[i], zen[i], v);
    cart[3 * i] = v[0]; cart[3 * i + 1] = v[1]; cart[3 * i + 2] = v[2];
}

__global__ void __launch_bounds__(256) grid_match_kernel(const double* __restrict__ aziA, const double* __restrict__ zenA,
                                                         const double* __restrict__ cartB, int64_t nB,
                                                         int64_t* __restrict__ idx, double* __restrict__ dev) {
    __shared__ double bd[256];
    __shared__ int64_t bi[256];
    const int64_t a = blockIdx.x;
    double va[3];
    sph2cart_unit(aziA[a], zenA[a], va);
    double best = INFINITY;
    int64_t besti = 0;
    for (int64_t b = threadIdx.x; b < nB; b += blockDim.x) {
        const double dx = cartB[3 * b] - va[0], dy = cartB[3 * b + 1] - va[1], dz = cartB[3 * b + 2] - va[2];
        const double dist = sqrt(dx * dx + dy * dy + dz * dz);
        if (dist < best) { best = dist; besti = b; }
    }
    bd[threadIdx.x] = best;
    bi[threadIdx.x] = besti;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const double o = bd[threadIdx.x + s];
            const int64_t oi = bi[threadIdx.x + s];
            if (o < bd[threadIdx.x] || (o == bd[threadIdx.x] && oi < bi[threadIdx.x])) { bd[threadIdx.x] = o; bi[threadIdx.x] = oi; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int64_t b = bi[0];
        idx[a] = b;
        double dot = va[0] * cartB[3 * b] + va[1] * cartB[3 * b + 1] + va[2] * cartB[3 * b + 2];
        dot = fmin(1.0, fmax(-1.0, dot));
        dev[a] = acos(dot) * 180.0 / kPi;
    }
}

__global__ void __launch_bounds__(1024) mean_kernel(const double* __restrict__ x, int64_t n, double* __restrict__ out) {
    __shared__ double sh[1024];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += blockDim.x) acc += x[i];
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = sh[0] / (double)n;
}

// atfIrs is [taps x M x dirs] column-major: column (m, dir) starts at (dir*M + m)*taps.
// colidx[m*nA + a] = idx[a]*M + m  (idx == nullptr: identity)
__global__ void atf_colidx_kernel(const int64_t* __restrict__ idx, int64_t nA, int M, int64_t* __restrict__ colidx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nA * M) return;
    const int64_t m = i / nA, a = i % nA;
    colidx[i] = (idx ? idx[a] : a) * M + m;
}

void launch_grid_match(const double* aziA, const double* zenA, int64_t nA, const double* aziB, const double* zenB,
                       int64_t nB, double* cartB, int64_t* idx, double* dev_deg, double* mean_dev, hipStream_t st) {
    cart_kernel<<<(unsigned)ceil_div(nB, 256), 256, 0, st>>>(aziB, zenB, nB, cartB);
    KERNEL_CHECK();
    grid_match_kernel<<<(unsigned)nA, 256, 0, st>>>(aziA, zenA, cartB, nB, idx, dev_deg);
    KERNEL_CHECK();
    mean_kernel<<<1, 1024, 0, st>>>(dev_deg, nA, mean_dev);
    KERNEL_CHECK();
}

void launch_atf_colidx(const int64_t* idx, int64_t nA, int M, int64_t* colidx, hipStream_t st) {
    atf_colidx_kernel<<<(unsigned)ceil_div(nA * M, 256), 256, 0, st>>>(idx, nA, M, colidx);
    KERNEL_CHECK();
}

}  // namespace emagls
