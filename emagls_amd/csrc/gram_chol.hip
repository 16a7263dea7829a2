// Orthonormal compression of the HRIR-grid SH matrix:  conj(Y) = Q R  (Cholesky-QR)
//
//   gram      G = Yc^H Yc  (S x S, upper block triangle)   FP64 MFMA v_mfma_f64_16x16x4_f64, split-K
//   cholesky  G = R^H R    blocked right-looking, 2 launches per 32-column panel
//   qform     Q = Yc R^-1  row-parallel forward substitution (rows are independent)
//   tn        T_n = R(:,blk_n) E(:,blk_n)^T  so that  R diag(b_n) E^T = sum_n b_n T_n
//
// This replaces the reference's per-bin  pwGrid = smairMat(:,:,k) * Y_Hi_conj  (lib/getEMagLsFilters.m:87)
// + svd(pwGrid.')  (:88) on a D x C matrix by an S x C problem:  pwGrid.' = conj(Y) A_k^T = Q (R A_k^T).
// The HRIR grid's SH matrix is well conditioned (cond 1.5 at N=19 on the 2702-point grid), so
// Cholesky-QR loses nothing; a non-positive pivot raises a device flag and the host reports it.
#include <cstdlib>
#include <vector>

#include "kernels.hpp"

namespace emagls {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------------------------
// Gram.  Workgroup = 4 waves = 64x64 tile of G, each wave a 32x32 sub-tile (2x2 MFMA tiles).
// grid.x enumerates upper block-triangle tiles (ti <= tj), grid.y = K split.
// Yc is [Dpad][ld] with rows >= D zero.  Lane l feeds A[i=l&15][k=l>>4], B[k=l>>4][j=l&15];
// f64 C/D layout: col = l&15, row = (l>>4) + 4*reg.
// ---------------------------------------------------------------------------------------------
template <typename T> struct GramAcc;
template <> struct GramAcc<double> {
    double4_t r[2][2];
    __device__ void init() { for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) r[a][b] = double4_t{0, 0, 0, 0}; }
};
template <> struct GramAcc<cplx> {
    double4_t r[2][2], i[2][2];
    __device__ void init() {
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) { r[a][b] = double4_t{0, 0, 0, 0}; i[a][b] = double4_t{0, 0, 0, 0}; }
    }
};

__device__ __forceinline__ void gram_step(GramAcc<double>& acc, const double (&a)[2], const double (&b)[2]) {
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) acc.r[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x], b[y], acc.r[x][y], 0, 0, 0);
}
__device__ __forceinline__ void gram_step(GramAcc<cplx>& acc, const cplx (&a)[2], const cplx (&b)[2]) {
    // G = conj(A)^T B :  Gr += ar br + ai bi ;  Gi += ar bi - ai br
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            acc.r[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x].x, b[y].x, acc.r[x][y], 0, 0, 0);
            acc.r[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x].y, b[y].y, acc.r[x][y], 0, 0, 0);
            acc.i[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x].x, b[y].y, acc.i[x][y], 0, 0, 0);
            acc.i[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[x].y, b[y].x, acc.i[x][y], 0, 0, 0);
        }
}
__device__ __forceinline__ double gram_get(const GramAcc<double>& acc, int x, int y, int r, double*) { return acc.r[x][y][r]; }
__device__ __forceinline__ cplx gram_get(const GramAcc<cplx>& acc, int x, int y, int r, cplx*) {
    return mk(acc.r[x][y][r], acc.i[x][y][r]);
}

template <typename T>
__global__ void __launch_bounds__(256) gram_mfma_kernel(const T* __restrict__ Yc, int64_t ld, int S, int kc, int nbt,
                                                        T* __restrict__ Gp, size_t bstride) {
    Yc = boff(Yc, bstride); Gp = boff(Gp, bstride);
    // decode upper-triangle tile index
    int t = blockIdx.x, ti = 0;
    while (t >= nbt - ti) { t -= nbt - ti; ++ti; }
    const int tj = ti + t;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int i0 = ti * 64 + (wave >> 1) * 32, j0 = tj * 64 + (wave & 1) * 32;
    const int ii = lane & 15, kk = lane >> 4;
    const int64_t dbeg = (int64_t)blockIdx.y * kc;
    GramAcc<T> acc;
    acc.init();
    const bool va0 = i0 + ii < S, va1 = i0 + 16 + ii < S, vb0 = j0 + ii < S, vb1 = j0 + 16 + ii < S;
    for (int64_t d = dbeg; d < dbeg + kc; d += 4) {
        const T* row = Yc + (d + kk) * ld;
        T a[2], b[2];
        a[0] = va0 ? row[i0 + ii] : zero_of<T>();
        a[1] = va1 ? row[i0 + 16 + ii] : zero_of<T>();
        b[0] = vb0 ? row[j0 + ii] : zero_of<T>();
        b[1] = vb1 ? row[j0 + 16 + ii] : zero_of<T>();
        gram_step(acc, a, b);
    }
    T* out = Gp + (int64_t)blockIdx.y * S * S;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = i0 + 16 * x + kk + 4 * r, gj = j0 + 16 * y + ii;
                if (gi < S && gj < S) out[(int64_t)gi * S + gj] = gram_get(acc, x, y, r, (T*)nullptr);
            }
}

// The same Gram matrix without the K split, for launches that fill the chip with tiles alone (lane batches: 28 tiles x 8 designs
// at config 3): one workgroup walks ALL rows of its 64 x 64 tile, 16 rows at a time through a double-buffered LDS stage (the
// next 16 rows are in flight in registers while the MFMAs run on the current ones), and writes the finished tile straight into
// Gy and into the leading block of R -- no K-split partials (20 MB per design written and read back) and no reduce kernel.
// The MFMA fragments come from LDS (16 consecutive doubles per k row: conflict free), not from global memory.
constexpr int GL_KC = 16;          // rows per stage
constexpr int GL_SB = 8;           // tiles per side of a super-block (the order in which a lane's tiles are taken)
constexpr int GL_LD = 64 + 16;     // row stride of a stage in doubles (32 banks mod 64: the four k rows of a fragment read tile the banks)
__global__ void __launch_bounds__(256) gram_lds_kernel(const double* __restrict__ Yc, int64_t ld, int S, int64_t rows, int nbt,
                                                       double* __restrict__ G, double* __restrict__ R, int Sh, size_t bstride, int xcd_runs) {
    // workgroup -> (lane, tile), XCD aware (xcd_run_index, common.hpp): the plain mapping dealt the 28 tiles of a design over all eight L2s and
    // every tile fetched its two panels itself -- 68 MB per design for a 9.9 MB operand, 15 MB this way (profiles/r06_xcd_runs.md)
    unsigned zl = blockIdx.z, tl = blockIdx.x;
    if (xcd_runs) xcd_run_index(tl, zl);
    int t = (int)tl;
    Yc = boffz(Yc, bstride, zl); G = boffz(G, bstride, zl); R = boffz(R, bstride, zl);
    __shared__ __attribute__((aligned(16))) double As[2][GL_KC][GL_LD];
    __shared__ __attribute__((aligned(16))) double Bs[2][GL_KC][GL_LD];
    // tile order inside a lane: super-blocks of GL_SB x GL_SB tiles over the upper block triangle, row-major inside a super-block.  The
    // ~100 tiles an XCD runs at a time then share 2 GL_SB panels instead of the 1 + 32 of a tile row (S = 2025: 528 tiles per design, 4.9 GB
    // per 8-lane launch in row order); up to GL_SB tile rows (config 3: 7) it IS the row order.
    int ti = 0, tj = 0;
    {
        const int nsb = (nbt + GL_SB - 1) / GL_SB;
        bool found = false;
        for (int I = 0; I < nsb && !found; ++I) {
            const int ri = min(GL_SB, nbt - I * GL_SB);
            for (int J = I; J < nsb && !found; ++J) {
                const int rj = min(GL_SB, nbt - J * GL_SB);
                const int cnt = I == J ? ri * (ri + 1) / 2 : ri * rj;
                if (t >= cnt) { t -= cnt; continue; }
                if (I == J) {
                    int a = 0;
                    while (t >= ri - a) { t -= ri - a; ++a; }
                    ti = I * GL_SB + a; tj = ti + t;
                } else {
                    ti = I * GL_SB + t / rj; tj = J * GL_SB + t % rj;
                }
                found = true;
            }
        }
    }
    const bool diag = ti == tj;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;      // the wave's 32 x 32 sub-tile
    const int ii = lane & 15, kk = lane >> 4;
    // loader role: row lr of the stage, columns 4 lc .. 4 lc + 3 of the panel (16 threads cover one 512-byte row segment)
    const int lr = tid >> 4, lc = (tid & 15) * 4;
    const double* pa = Yc + (int64_t)lr * ld + ti * 64 + lc;
    const double* pb = Yc + (int64_t)lr * ld + tj * 64 + lc;
    // (columns beyond S are the zero padding of Yc's rows, ld >= 64 nbt; rows beyond D are zero up to `rows`)
    double4_t ra, rb;
    auto fetch = [&](int64_t d0) __attribute__((always_inline)) {
        const bool ok = d0 + lr < rows;
        ra = ok ? *reinterpret_cast<const double4_t*>(pa + d0 * ld) : double4_t{0, 0, 0, 0};
        if (!diag) rb = ok ? *reinterpret_cast<const double4_t*>(pb + d0 * ld) : double4_t{0, 0, 0, 0};
    };
    auto stage = [&](int buf) __attribute__((always_inline)) {
        As[buf][lr][lc] = ra[0]; As[buf][lr][lc + 1] = ra[1]; As[buf][lr][lc + 2] = ra[2]; As[buf][lr][lc + 3] = ra[3];
        if (!diag) { Bs[buf][lr][lc] = rb[0]; Bs[buf][lr][lc + 1] = rb[1]; Bs[buf][lr][lc + 2] = rb[2]; Bs[buf][lr][lc + 3] = rb[3]; }
    };
    GramAcc<double> acc;
    acc.init();
    fetch(0);
    stage(0);
    __syncthreads();
    const int64_t nst = (rows + GL_KC - 1) / GL_KC;
    for (int64_t c = 0; c < nst; ++c) {
        const int buf = (int)(c & 1);
        if (c + 1 < nst) fetch((c + 1) * GL_KC);
        const double (*A)[GL_LD] = As[buf];
        const double (*B)[GL_LD] = diag ? As[buf] : Bs[buf];
#pragma unroll
        for (int k4 = 0; k4 < GL_KC; k4 += 4) {
            double a[2], b[2];
            a[0] = A[k4 + kk][wi + ii]; a[1] = A[k4 + kk][wi + 16 + ii];
            b[0] = B[k4 + kk][wj + ii]; b[1] = B[k4 + kk][wj + 16 + ii];
            gram_step(acc, a, b);
        }
        if (c + 1 < nst) stage(buf ^ 1);
        __syncthreads();
    }
    const int i0 = ti * 64 + wi, j0 = tj * 64 + wj;
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = i0 + 16 * x + kk + 4 * r, gj = j0 + 16 * y + ii;
                if (gi < S && gj < S) {
                    const double v = acc.r[x][y][r];
                    if (G) { G[(int64_t)gi * S + gj] = v; if (!diag) G[(int64_t)gj * S + gi] = 0.0; }   // (block lower triangle: zeros, like the reduce kernel)
                    if (R && gi < Sh && gj < Sh) { R[(int64_t)gi * Sh + gj] = v; if (!diag) R[(int64_t)gj * Sh + gi] = 0.0; }
                }
            }
}

// Round 5: the same tile on v_mfma_f64_4x4x4_4b.  On gfx950 the 16 x 16 x 4 FP64 shape sustains 49 TFLOP/s whatever is done around it
// (62 % of the pipe's nominal rate), the four-block 4 x 4 x 4 shape 75 (tools/experiments/mfma_peak.hip); operand layout found by
// experiment (tools/experiments/mfma_4x4_layout.hip): lane l = x + 4 b + 16 y supplies A_b[i = x][k = y] and B_b[k = y][j = x] and
// receives D_b[i = y][j = x].  A wave's 32 x 32 sub-tile is 8 row tiles (ra) x 2 column groups (rb) of four blocks: element
// (4 ra + i, 16 rb + 4 b + j) accumulates in register [ra][rb] of lane j + 4 b + 16 i -- sixteen independent instructions per four
// rows of Y.  The stages hold the panels permuted so that a lane's eight A values and two B values of a row are contiguous
// (four + one ds_read_b128 per sixteen instructions; lanes that differ in b only read the same A address: a broadcast).
//   A stage  [k][half][i][ra]      column 32 half + 4 ra + i of the row panel
//   B stage  [k][half][4 b + j][rb]  column 32 half + 16 rb + 4 b + j of the column panel
constexpr int G4_LD = 64 + 2;   // (row stride 132 banks = 4 mod 64: the four k rows x four 64-byte A segments of a read tile the banks)
__global__ void __launch_bounds__(256) gram_lds4_kernel(const double* __restrict__ Yc, int64_t ld, int S, int64_t rows, int nbt,
                                                        double* __restrict__ G, double* __restrict__ R, int Sh, size_t bstride) {
    Yc = boff(Yc, bstride); G = boff(G, bstride); R = boff(R, bstride);
    __shared__ __attribute__((aligned(16))) double As[2][GL_KC][G4_LD];
    __shared__ __attribute__((aligned(16))) double Bs[2][GL_KC][G4_LD];
    int t = blockIdx.x, ti = 0;
    while (t >= nbt - ti) { t -= nbt - ti; ++ti; }
    const int tj = ti + t;
    const bool diag = ti == tj;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int hi = wave >> 1, hj = wave & 1;                    // the wave's 32 x 32 sub-tile: row half, column half
    const int lx = lane & 3, lb = (lane >> 2) & 3, ly = lane >> 4;
    // loader role: row lr of the stage, columns lc .. lc + 3 of the panel (16 threads cover one 512-byte row segment)
    const int lr = tid >> 4, lc = (tid & 15) * 4;
    const double* pa = Yc + (int64_t)lr * ld + ti * 64 + lc;
    const double* pb = Yc + (int64_t)lr * ld + tj * 64 + lc;
    // where the loader's four columns go: A layout (i = q, ra = (lc & 31) / 4), B layout (j = q, b = (lc / 4) & 3, rb = (lc & 31) / 16)
    const int sa = (lc >> 5) * 32 + ((lc & 31) >> 2);                               // + 8 q
    const int sb = (lc >> 5) * 32 + (((lc >> 2) & 3) * 4) * 2 + ((lc & 31) >> 4);    // + 2 q
    double4_t ra4, rb4;
    auto fetch = [&](int64_t d0) __attribute__((always_inline)) {
        const bool ok = d0 + lr < rows;
        ra4 = ok ? *reinterpret_cast<const double4_t*>(pa + d0 * ld) : double4_t{0, 0, 0, 0};
        rb4 = diag ? ra4 : (ok ? *reinterpret_cast<const double4_t*>(pb + d0 * ld) : double4_t{0, 0, 0, 0});
    };
    auto stage = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { As[buf][lr][sa + 8 * q] = ra4[q]; Bs[buf][lr][sb + 2 * q] = rb4[q]; }
    };
    double acc[8][2];
#pragma unroll
    for (int x = 0; x < 8; ++x) { acc[x][0] = 0.0; acc[x][1] = 0.0; }
    fetch(0);
    stage(0);
    __syncthreads();
    const int64_t nst = (rows + GL_KC - 1) / GL_KC;
    const int oa = hi * 32 + lx * 8, ob = hj * 32 + (lb * 4 + lx) * 2;
    for (int64_t c = 0; c < nst; ++c) {
        const int buf = (int)(c & 1);
        if (c + 1 < nst) fetch((c + 1) * GL_KC);
#pragma unroll
        for (int k4 = 0; k4 < GL_KC; k4 += 4) {
            const double* arow = &As[buf][k4 + ly][oa];
            const double* brow = &Bs[buf][k4 + ly][ob];
            double a[8], b[2];
#pragma unroll
            for (int x = 0; x < 8; x += 2) { const double2_t v = *reinterpret_cast<const double2_t*>(arow + x); a[x] = v[0]; a[x + 1] = v[1]; }
            { const double2_t v = *reinterpret_cast<const double2_t*>(brow); b[0] = v[0]; b[1] = v[1]; }
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                acc[x][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[x], b[0], acc[x][0], 0, 0, 0);
                acc[x][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[x], b[1], acc[x][1], 0, 0, 0);
            }
        }
        if (c + 1 < nst) stage(buf ^ 1);
        __syncthreads();
    }
    const int i0 = ti * 64 + hi * 32, j0 = tj * 64 + hj * 32;
#pragma unroll
    for (int x = 0; x < 8; ++x)
#pragma unroll
        for (int y = 0; y < 2; ++y) {
            const int gi = i0 + 4 * x + ly, gj = j0 + 16 * y + 4 * lb + lx;
            if (gi < S && gj < S) {
                const double v = acc[x][y];
                if (G) { G[(int64_t)gi * S + gj] = v; if (!diag) G[(int64_t)gj * S + gi] = 0.0; }   // (block lower triangle: zeros, like the reduce kernel)
                if (R && gi < Sh && gj < Sh) { R[(int64_t)gi * Sh + gj] = v; if (!diag) R[(int64_t)gj * Sh + gi] = 0.0; }
            }
        }
}

// both tile kernels on a pseudo-random matrix against a plain host sum (debug entry emagls_self_test(1)): largest error relative to
// the largest element of the Gram matrix; `four` selects the kernel
double gram_tile_selftest(bool four) {
    const int S = 100, D = 333, ld = 128, nbt = 2, ntiles = 3;
    const int64_t rows = 336;   // (a multiple of four, like gram_dpad)
    std::vector<double> Y((size_t)rows * ld, 0.0), G((size_t)S * S, 0.0), want((size_t)S * S, 0.0);
    unsigned long long x = 88172645463325252ull;
    for (int d = 0; d < D; ++d)
        for (int c = 0; c < S; ++c) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; Y[(size_t)d * ld + c] = (double)(x % 2000001ull) / 1000000.0 - 1.0; }
    double gmax = 0.0;
    for (int i = 0; i < S; ++i)
        for (int j = i; j < S; ++j) {
            double acc = 0.0;
            for (int d = 0; d < D; ++d) acc += Y[(size_t)d * ld + i] * Y[(size_t)d * ld + j];
            want[(size_t)i * S + j] = acc;
            gmax = std::max(gmax, std::fabs(acc));
        }
    double *dY = nullptr, *dG = nullptr;
    HIP_CHECK(hipMalloc(&dY, sizeof(double) * Y.size()));
    HIP_CHECK(hipMalloc(&dG, sizeof(double) * G.size()));
    HIP_CHECK(hipMemcpy(dY, Y.data(), sizeof(double) * Y.size(), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemset(dG, 0, sizeof(double) * G.size()));
    if (four) gram_lds4_kernel<<<dim3(ntiles), 256>>>(dY, ld, S, rows, nbt, dG, nullptr, 0, 0);
    else gram_lds_kernel<<<dim3(ntiles), 256>>>(dY, ld, S, rows, nbt, dG, nullptr, 0, 0, 1);
    KERNEL_CHECK();
    HIP_CHECK(hipMemcpy(G.data(), dG, sizeof(double) * G.size(), hipMemcpyDeviceToHost));
    HIP_CHECK(hipFree(dY));
    HIP_CHECK(hipFree(dG));
    double err = 0.0;
    for (int i = 0; i < S; ++i)
        for (int j = 0; j < S; ++j) {
            // (upper block triangle: the tiles (0,0), (0,1), (1,1); inside a diagonal tile both triangles are written)
            const bool upper_block = (i >> 6) <= (j >> 6);
            const double w = upper_block ? want[(size_t)std::min(i, j) * S + std::max(i, j)] : 0.0;
            err = std::max(err, std::fabs(G[(size_t)i * S + j] - w));
        }
    return err / gmax;
}

// sums the K-split partials into the Gram matrix Gy (S x S, upper block triangle) and copies its leading Sh x Sh block to R
// (compact, leading dimension Sh), which the Cholesky factorisation then overwrites: Gy itself stays intact for the Gram route
template <typename T>
__global__ void __launch_bounds__(256) gram_reduce_kernel(const T* __restrict__ Gp, int S, int ksplit, T* __restrict__ G, T* __restrict__ R, int Sh,
                                                          size_t bstride) {
    Gp = boff(Gp, bstride); G = boff(G, bstride); R = boff(R, bstride);
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)S * S) return;
    const int i = (int)(idx / S), j = (int)(idx % S);
    T acc = zero_of<T>();
    if ((i >> 6) <= (j >> 6))
        for (int y = 0; y < ksplit; ++y) acc = acc + Gp[(int64_t)y * S * S + idx];
    if (G) G[idx] = acc;
    if (R && i < Sh && j < Sh) R[(int64_t)i * Sh + j] = acc;
}

// ---------------------------------------------------------------------------------------------
// blocked Cholesky  G = R^H R  (upper, in place; only the upper triangle is read or written)
// ---------------------------------------------------------------------------------------------
constexpr int NB = 32;

template <typename T> __device__ __forceinline__ double real_of(T v);
template <> __device__ __forceinline__ double real_of<double>(double v) { return v; }
template <> __device__ __forceinline__ double real_of<cplx>(cplx v) { return v.x; }
template <typename T> __device__ __forceinline__ T to_T(double v);
template <> __device__ __forceinline__ double to_T<double>(double v) { return v; }
template <> __device__ __forceinline__ cplx to_T<cplx>(double v) { return mk(v, 0.0); }
template <typename T> __device__ __forceinline__ T scale_real(T v, double s);
template <> __device__ __forceinline__ double scale_real<double>(double v, double s) { return v * s; }
template <> __device__ __forceinline__ cplx scale_real<cplx>(cplx v, double s) { return mk(v.x * s, v.y * s); }

// panel kernel: every workgroup factors the diagonal block redundantly (wave 0, wave-synchronous LDS: no
// workgroup barrier inside the 32-step elimination), inverts L = R_JJ^H once, then forms its own 32-column
// slice of the block row as a small product:  R(J, c) = L^-1 G(J, c).
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// diagonal block: one wave factors G(J,J) in place (wave-synchronous LDS, the trailing update of each of the
// 32 steps spread over the 64 lanes).  A separate launch, so the row-panel workgroups below read a factor
// that can no longer change (they used to re-factor a block that workgroup 0 was overwriting: a race).
template <typename T>
__global__ void __launch_bounds__(256) chol_diag_kernel(T* __restrict__ G, int S, int j0, int* __restrict__ flag, size_t bstride) {
    G = boff(G, bstride); flag = boff(flag, bstride);
    // 256 threads, two barriers per elimination step: every trailing element has its own thread, so a step costs two
    // LDS round trips (~0.3 us) instead of the ~1 us of a single wave walking the 32 x 32 block
    __shared__ T Rd[NB][NB + 1];
    __shared__ int bad_s;
    const int nb = min(NB, S - j0);
    const int tid = threadIdx.x;
    if (tid == 0) bad_s = 0;
    for (int idx = tid; idx < NB * NB; idx += 256) {
        const int r = idx / NB, c = idx % NB;
        Rd[r][c] = (r < nb && c < nb && r <= c) ? G[(int64_t)(j0 + r) * S + j0 + c] : zero_of<T>();
    }
    __syncthreads();
    const int r4 = tid >> 5, c = tid & 31;   // thread = (row r4 + 8 i, column c)
    for (int j = 0; j < nb; ++j) {
        const double piv = real_of(Rd[j][j]);
        if (!(piv > 0.0) && tid == 0) bad_s = 1;
        const double dinv = fast_rsqrt(piv > 0.0 ? piv : 1.0);
        __syncthreads();
        if (tid >= j && tid < nb) Rd[j][tid] = scale_real(Rd[j][tid], dinv);  // diag becomes sqrt(piv)
        __syncthreads();
        const T rjc = Rd[j][c];
#pragma unroll
        for (int i = 0; i < NB / 8; ++i) {
            const int r = r4 + 8 * i;
            if (r > j && r <= c && c < nb) {
                T acc = zero_of<T>();
                cfma_conj(acc, Rd[j][r], rjc);
                Rd[r][c] = Rd[r][c] - acc;
            }
        }
        // (the next step reads Rd[j+1][j+1] and row j+1: written by this step's update -> barrier at its top)
        __syncthreads();
    }
    if (bad_s && tid == 0) atomicExch(flag, 1 + j0);
    for (int idx = tid; idx < NB * NB; idx += 256) {
        const int r = idx / NB, cc = idx % NB;
        if (r < nb && cc < nb && r <= cc) G[(int64_t)(j0 + r) * S + j0 + cc] = Rd[r][cc];
    }
}

// row panel: R(J, c) = L^-1 G(J, c), L = R_JJ^H (already factored): right-looking forward substitution; each wave
// owns 8 of the workgroup's 32 columns (lane = (column cl, row lane rl)), no workgroup barriers inside
template <typename T>
__global__ void __launch_bounds__(256) chol_panel_kernel(T* __restrict__ G, int S, int j0, size_t bstride) {
    G = boff(G, bstride);
    __shared__ T Rd[NB][NB + 1];
    __shared__ T Gs[NB][NB + 1];
    const int nb = min(NB, S - j0);
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c0 = j0 + (blockIdx.x + 1) * NB;
    for (int idx = tid; idx < NB * NB; idx += blockDim.x) {
        const int r = idx / NB, c = idx % NB;
        Rd[r][c] = (r < nb && c < nb && r <= c) ? G[(int64_t)(j0 + r) * S + j0 + c] : zero_of<T>();
        Gs[r][c] = (r < nb && c0 + c < S) ? G[(int64_t)(j0 + r) * S + c0 + c] : zero_of<T>();
    }
    __syncthreads();
    {
        const int cl = wave * 8 + (lane & 7), rl = lane >> 3;
        for (int i = 0; i < nb; ++i) {
            const double dinv = 1.0 / real_of(Rd[i][i]);
            wave_lds_fence();
            if (rl == 0) Gs[i][cl] = scale_real(Gs[i][cl], dinv);
            wave_lds_fence();
            const T xi = Gs[i][cl];
            T rir[NB / 8], cur[NB / 8];
#pragma unroll
            for (int k = 0; k < NB / 8; ++k) {
                const int r = rl + 8 * k;
                rir[k] = Rd[i][r];
                cur[k] = Gs[r][cl];
            }
#pragma unroll
            for (int k = 0; k < NB / 8; ++k) {
                const int r = rl + 8 * k;
                if (r > i && r < nb) {
                    T acc = zero_of<T>();
                    cfma_conj(acc, rir[k], xi);
                    Gs[r][cl] = cur[k] - acc;
                }
            }
        }
    }
    __syncthreads();
    for (int idx = tid; idx < NB * NB; idx += blockDim.x) {
        const int i = idx / NB, c = idx % NB;
        if (i < nb && c0 + c < S) G[(int64_t)(j0 + i) * S + c0 + c] = Gs[i][c];
    }
}

// trailing update: G(r, c) -= sum_{i in J} conj(R(i, r)) R(i, c) for r <= c beyond the panel
template <typename T>
__global__ void __launch_bounds__(256) chol_update_kernel(T* __restrict__ G, int S, int j0, int nbt, size_t bstride) {
    G = boff(G, bstride);
    __shared__ T Pr[NB][NB + 1], Pc[NB][NB + 1];
    int t = blockIdx.x, bi = 0;
    while (t >= nbt - bi) { t -= nbt - bi; ++bi; }
    const int bj = bi + t;
    const int j1 = j0 + NB;
    const int r0 = j1 + bi * NB, c0 = j1 + bj * NB;
    const int tid = threadIdx.x;
    for (int idx = tid; idx < NB * NB; idx += blockDim.x) {
        const int i = idx / NB, x = idx % NB;
        Pr[i][x] = (r0 + x < S) ? G[(int64_t)(j0 + i) * S + r0 + x] : zero_of<T>();
        Pc[i][x] = (c0 + x < S) ? G[(int64_t)(j0 + i) * S + c0 + x] : zero_of<T>();
    }
    __syncthreads();
    for (int idx = tid; idx < NB * NB; idx += blockDim.x) {
        const int r = idx / NB, c = idx % NB;
        if (r0 + r < S && c0 + c < S && r0 + r <= c0 + c) {
            T acc = zero_of<T>();
#pragma unroll
            for (int i = 0; i < NB; ++i) cfma_conj(acc, Pr[i][r], Pc[i][c]);
            T* g = &G[(int64_t)(r0 + r) * S + c0 + c];
            *g = *g - acc;
        }
    }
}

// The whole factorisation in ONE launch (opt-in, see chol_fused_enabled: measured slower than the 22 launches at S = 256): a
// workgroup of 256 threads per design walks the block columns (diagonal block, row panel, trailing update) with the panel's rows
// R(J, :) in LDS; the matrix itself stays in global memory (L2 resident: 512 KB at S = 256).
//   panel: thread = one trailing column, forward substitution in registers against the factored diagonal block (LDS broadcast)
//   update: 4 x 4 register tiles of the upper triangle of the trailing block, panel rows from LDS
template <typename T>
__global__ void __launch_bounds__(256) chol_fused_kernel(T* __restrict__ G, int S, int* __restrict__ flag, int ldp, size_t bstride) {
    G = boff(G, bstride); flag = boff(flag, bstride);
    __shared__ T Rd[NB][NB + 1];
    __shared__ int bad_s;
    extern __shared__ __attribute__((aligned(16))) char dynp[];
    T* Ps = reinterpret_cast<T*>(dynp);   // [NB][ldp]: the panel's rows over the trailing columns
    const int tid = threadIdx.x;
    if (tid == 0) bad_s = 0;
    for (int j0 = 0; j0 < S; j0 += NB) {
        const int nb = min(NB, S - j0);
        const int c0 = j0 + NB, W = max(S - c0, 0);   // trailing columns
        __syncthreads();   // (the previous step's updates of G are visible; Rd / Ps are free)
        // ---- diagonal block: 256 threads, two barriers per elimination step (chol_diag_kernel)
        for (int idx = tid; idx < NB * NB; idx += 256) {
            const int r = idx / NB, c = idx % NB;
            Rd[r][c] = (r < nb && c < nb && r <= c) ? G[(int64_t)(j0 + r) * S + j0 + c] : zero_of<T>();
        }
        // the panel's rows load while the diagonal block is factored (they are only read after it)
        for (int idx = tid; idx < NB * W; idx += 256) {
            const int i = idx / W, c = idx % W;
            Ps[(size_t)i * ldp + c] = i < nb ? G[(int64_t)(j0 + i) * S + c0 + c] : zero_of<T>();
        }
        __syncthreads();
        const int r4 = tid >> 5, cc = tid & 31;
        for (int j = 0; j < nb; ++j) {
            const double piv = real_of(Rd[j][j]);
            if (!(piv > 0.0) && tid == 0) bad_s = 1 + j0;
            const double dinv = fast_rsqrt(piv > 0.0 ? piv : 1.0);
            __syncthreads();
            if (tid >= j && tid < nb) Rd[j][tid] = scale_real(Rd[j][tid], dinv);
            __syncthreads();
            const T rjc = Rd[j][cc];
#pragma unroll
            for (int i = 0; i < NB / 8; ++i) {
                const int r = r4 + 8 * i;
                if (r > j && r <= cc && cc < nb) {
                    T acc = zero_of<T>();
                    cfma_conj(acc, Rd[j][r], rjc);
                    Rd[r][cc] = Rd[r][cc] - acc;
                }
            }
            __syncthreads();
        }
        for (int idx = tid; idx < NB * NB; idx += 256) {
            const int r = idx / NB, c = idx % NB;
            if (r < nb && c < nb && r <= c) G[(int64_t)(j0 + r) * S + j0 + c] = Rd[r][c];
        }
        if (W == 0) break;
        // ---- row panel: R(J, c) = L^-1 G(J, c), L = R_JJ^H; one column per thread, the column in registers
        for (int c = tid; c < W; c += 256) {
            T x[NB];
#pragma unroll
            for (int i = 0; i < NB; ++i) x[i] = Ps[(size_t)i * ldp + c];
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                if (i < nb) {
                    T acc = x[i];
#pragma unroll
                    for (int k = 0; k < NB; ++k)
                        if (k < i) { T t = zero_of<T>(); cfma_conj(t, Rd[k][i], x[k]); acc = acc - t; }
                    x[i] = scale_real(acc, 1.0 / real_of(Rd[i][i]));
                }
            }
#pragma unroll
            for (int i = 0; i < NB; ++i) {
                if (i < nb) {
                    Ps[(size_t)i * ldp + c] = x[i];
                    G[(int64_t)(j0 + i) * S + c0 + c] = x[i];
                }
            }
        }
        __syncthreads();
        // ---- trailing update: G(r, c) -= sum_i conj(R(i, r)) R(i, c), r <= c, in 4 x 4 tiles (tile rows tr <= tile columns tc)
        const int nt = (W + 3) / 4;
        const int ntiles = nt * (nt + 1) / 2;
        for (int t = tid; t < ntiles; t += 256) {
            // tile index -> (tr, tc), tr <= tc: row tr holds nt - tr tiles
            int tr = 0, rem = t;
            {   // closed form of the triangular index, then a correction step
                const double nn = (double)nt + 0.5;
                tr = (int)(nn - sqrt(nn * nn - 2.0 * (double)t));
                if (tr < 0) tr = 0;
                while (tr > 0 && tr * nt - tr * (tr - 1) / 2 > t) --tr;
                while ((tr + 1) * nt - (tr + 1) * tr / 2 <= t) ++tr;
                rem = t - (tr * nt - tr * (tr - 1) / 2);
            }
            const int tc = tr + rem;
            const int r0 = 4 * tr, q0 = 4 * tc;
            T acc[4][4];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = zero_of<T>();
            for (int i = 0; i < nb; ++i) {
                T pr[4], pc[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    pr[a] = Ps[(size_t)i * ldp + r0 + a];   // (ldp >= W + 4, zero beyond W)
                    pc[a] = Ps[(size_t)i * ldp + q0 + a];
                }
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) cfma_conj(acc[a][b], pr[a], pc[b]);
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int r = r0 + a, c = q0 + b;
                    if (r < W && c < W && r <= c) {
                        T* g = &G[(int64_t)(c0 + r) * S + c0 + c];
                        *g = *g - acc[a][b];
                    }
                }
        }
    }
    __syncthreads();
    if (bad_s && tid == 0) atomicExch(flag, bad_s);
}

// ---------------------------------------------------------------------------------------------
// Q = Yc R^-1, blocked by 32 columns.  A workgroup owns 8 rows (directions) and keeps their finished Q
// entries in LDS; thread = (row r = tid/32, column-in-block c = tid%32).  Per block J:
//   acc(r,c) = Yc(r, j0+c) - sum_{i < j0} Q(r,i) R(i, j0+c)         (Q from LDS, R streamed from L2)
//   x = acc R_JJ^-1 by forward substitution inside the 32-lane half wave (shuffles, no barrier)
// Rows are independent, so there is no inter-workgroup dependency and one barrier per block.
// ---------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ T shfl_T(T v, int src);
template <> __device__ __forceinline__ double shfl_T<double>(double v, int src) { return __shfl(v, src, 64); }
template <> __device__ __forceinline__ cplx shfl_T<cplx>(cplx v, int src) { return {__shfl(v.x, src, 64), __shfl(v.y, src, 64)}; }

// inverses of the 32 x 32 diagonal blocks of R (upper triangular), one wave per block:
// lane c builds column c of X = R_JJ^-1 by back substitution.  Rinv[J][k][c]
template <typename T>
__global__ void __launch_bounds__(64) rinv_diag_kernel(const T* __restrict__ R, int S, T* __restrict__ Rinv, size_t bstride) {
    R = boff(R, bstride); Rinv = boff(Rinv, bstride);
    __shared__ T Rd[NB][NB + 1];
    const int j0 = blockIdx.x * NB, nb = min(NB, S - j0), lane = threadIdx.x;
    for (int idx = lane; idx < NB * NB; idx += 64) {
        const int r = idx / NB, c = idx % NB;
        Rd[r][c] = (r < nb && c < nb && r <= c) ? R[(int64_t)(j0 + r) * S + j0 + c] : zero_of<T>();
    }
    __syncthreads();
    T* out = Rinv + (int64_t)blockIdx.x * NB * NB;
    if (lane < NB) {
        const int c = lane;
        T x[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) x[i] = zero_of<T>();
        if (c < nb) {
#pragma unroll
            for (int i = NB - 1; i >= 0; --i) {
                if (i <= c) {
                    T acc = (i == c) ? to_T<T>(1.0) : zero_of<T>();
#pragma unroll
                    for (int l = NB - 1; l > i; --l)
                        if (l <= c) { T p = zero_of<T>(); cfma(p, Rd[i][l], x[l]); acc = acc - p; }
                    x[i] = scale_real(acc, 1.0 / real_of(Rd[i][i]));
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) out[i * NB + c] = x[i];
    }
}

// Q = Yc R^-1, blocked by 32 columns.  A workgroup owns 8 rows (directions) and keeps their finished Q entries in
// LDS; thread = (row r = tid/32, column-in-block c = tid%32).  Per block J:
//   acc(r,c) = Yc(r, j0+c) - sum_{i < j0} Q(r,i) R(i, j0+c)     R streamed through LDS in 128-row chunks, the next
//                                                                 chunk's loads in flight while the current one is used
//   Q(r, j0+c) = sum_k acc(r,k) Rinv_J(k,c)                      no sequential solve
// TR rows per workgroup (32 TR threads).  TR = 8 for the D rows of Q; TR = 4 with CH = 32 keeps the LDS at 62 KB for the few
// least-squares rows of an array design, so that the kernel fits on a CU next to a resident sweep workgroup.
template <typename T, int CH, int TR>
__global__ void __launch_bounds__(32 * TR) qform_kernel(const T* __restrict__ Yc, const T* __restrict__ R,
                                                    const T* __restrict__ Rinv, int S, int64_t D, int64_t ld,
                                                    T* __restrict__ Q, size_t bstride) {
    Yc = boff(Yc, bstride); R = boff(R, bstride); Rinv = boff(Rinv, bstride); Q = boff(Q, bstride);
    constexpr int NV = CH / TR;  // chunk elements per thread
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    T* qs = reinterpret_cast<T*>(dyn);        // [TR][ldq]
    const int ldq = S + 1;
    T* rt = qs + (size_t)TR * ldq;            // [CH][33] chunk of R(:, block J)
    T* ri = rt + (size_t)CH * 33;             // [32][33] Rinv_J
    T* as = ri + (size_t)32 * 33;             // [TR][33] acc rows
    const int tid = threadIdx.x, r = tid >> 5, c = tid & 31;
    const int64_t d = (int64_t)blockIdx.x * TR + r;
    const bool rowok = d < D;
    for (int j0 = 0; j0 < S; j0 += 32) {
        const int col = j0 + c;
        const bool colok = col < S;
        T acc0 = (rowok && colok) ? Yc[d * ld + col] : zero_of<T>();
        T acc1 = zero_of<T>();
        const int nch = (j0 + CH - 1) / CH;  // chunks covering rows [0, j0)
        T v[NV];
        auto load_chunk = [&](int i0) {
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int i = i0 + r + TR * k;
                v[k] = (i < j0 && colok) ? R[(int64_t)i * S + col] : zero_of<T>();
            }
        };
        // Rinv_J for this block (independent of everything else: goes out first)
        T rv[32 / TR];
#pragma unroll
        for (int k = 0; k < 32 / TR; ++k) rv[k] = Rinv[((int64_t)(j0 / 32) * NB + r + TR * k) * NB + c];
        if (nch > 0) load_chunk(0);
#pragma unroll
        for (int k = 0; k < 32 / TR; ++k) ri[(r + TR * k) * 33 + c] = rv[k];
        for (int t = 0; t < nch; ++t) {
            __syncthreads();  // previous chunk fully consumed
#pragma unroll
            for (int k = 0; k < NV; ++k) rt[(r + TR * k) * 33 + c] = v[k];
            if (t + 1 < nch) load_chunk(CH * (t + 1));
            __syncthreads();
            const T* qrow = qs + (size_t)r * ldq + CH * t;
            const int lim = min(CH, j0 - CH * t);
            int ii = 0;
            for (; ii + 1 < lim; ii += 2) {
                T p0 = zero_of<T>(), p1 = zero_of<T>();
                cfma(p0, qrow[ii], rt[ii * 33 + c]);
                cfma(p1, qrow[ii + 1], rt[(ii + 1) * 33 + c]);
                acc0 = acc0 - p0;
                acc1 = acc1 - p1;
            }
            if (ii < lim) { T p0 = zero_of<T>(); cfma(p0, qrow[ii], rt[ii * 33 + c]); acc0 = acc0 - p0; }
        }
        as[r * 33 + c] = acc0 + acc1;
        __syncthreads();
        T x0 = zero_of<T>(), x1 = zero_of<T>();
#pragma unroll
        for (int k = 0; k < 32; k += 2) {
            cfma(x0, as[r * 33 + k], ri[k * 33 + c]);       // Rinv is upper triangular: entries below the diagonal are 0
            cfma(x1, as[r * 33 + k + 1], ri[(k + 1) * 33 + c]);
        }
        const T x = x0 + x1;
        if (colok) {
            qs[(size_t)r * ldq + col] = x;
            if (rowok) Q[d * ld + col] = x;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// T[n][c][s] = sum_{j in [n^2,(n+1)^2)} R[s][j] E[c][j]   (zero for s >= (n+1)^2: R is upper triangular)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) tn_kernel(const T* __restrict__ R, const T* __restrict__ E, int S, int C, int ldE,
                                                 T* __restrict__ Tn, int64_t ldS, size_t bstride) {
    R = boff(R, bstride); E = boff(E, bstride); Tn = boff(Tn, bstride);
    const int n = blockIdx.x, c = blockIdx.y;
    const int jb = n * n, je = (n + 1) * (n + 1);
    T* out = Tn + ((int64_t)n * C + c) * ldS;
    for (int s = threadIdx.x; s < ldS; s += blockDim.x) {
        T acc = zero_of<T>();
        if (s < je && s < S)
            for (int j = max(jb, s); j < je; ++j) cfma(acc, R[(int64_t)s * S + j], E[(int64_t)c * ldE + j]);
        out[s] = acc;
    }
}

// small dense product  Cm[i][j] = sum_l A[i][l] Bm[l][j]   (E = pinv(Y_Lo) Y_mic etc.)
template <typename TA, typename TB, typename TC>
__global__ void __launch_bounds__(256) small_gemm_kernel(const TA* __restrict__ A, int lda, const TB* __restrict__ Bm, int ldb,
                                                         TC* __restrict__ Cm, int ldc, int M, int N, int K, size_t bstride) {
    A = boff(A, bstride); Bm = boff(Bm, bstride); Cm = boff(Cm, bstride);
    const int i = blockIdx.y;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < N; j += gridDim.x * blockDim.x) {
        cplx acc = mk(0, 0);
        for (int l = 0; l < K; ++l) { cplx p = to_cplx(A[(int64_t)i * lda + l]) * to_cplx(Bm[(int64_t)l * ldb + j]); acc += p; }
        if constexpr (sizeof(TC) == sizeof(double)) Cm[(int64_t)i * ldc + j] = acc.x; else Cm[(int64_t)i * ldc + j] = acc;
    }
    (void)M;
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
// K splits of the Gram product: enough workgroups to fill the chip (about a thousand), never slices shorter than 32 directions.
// Few tiles (S = 400: 28) take the full 32 splits; many tiles (S = 2025: 528) need only two, which also keeps the partials small.
int gram_ksplit(int64_t D, int S) {
    const int nbt = (S + 63) / 64;
    const int ntiles = nbt * (nbt + 1) / 2;
    int ks = 32;
    while (ks > 1 && (D / ks < 32 || (int64_t)ntiles * ks > 1024 + ntiles)) ks >>= 1;
    return ks;
}
int64_t gram_dpad(int64_t D, int S) {
    const int ks = gram_ksplit(D, S);
    const int64_t kc = ceil_div(ceil_div(D, ks), 4) * 4;
    return kc * ks;
}

template <typename T>
static void gram_impl(const void* Yc, int64_t D, int S, int64_t ld, void* Gp, void* G, void* R, int Sh, hipStream_t st) {
    const int ks = gram_ksplit(D, S);   // (fewer splits in lane mode were measured slower: 595 vs 454 + 104 us per 8 designs)
    const int kc = (int)(gram_dpad(D, S) / ks);
    const int nbt = (S + 63) / 64;
    const int ntiles = nbt * (nbt + 1) / 2;
    if constexpr (std::is_same<T, double>::value) {
        // enough tiles to fill the chip without a K split (lane batches of array designs): the LDS-staged kernel, no partials
        static const bool lds_ok = [] { const char* e = getenv("EMAGLS_GRAM_LDS"); return !(e && e[0] == '0'); }();
        if (lds_ok && (int64_t)ntiles * batch_ctx().n >= 128 && ld >= 64 * nbt) {
            // EMAGLS_GRAM_MFMA4=1: the tile on the 4 x 4 x 4 shape.  Measured (16 designs per launch, 448 workgroups, nothing else on the
            // GPU): 304 us against 236 us for the 16 x 16 x 4 kernel -- that one runs AT its shape's limit (42 of 49 TFLOP/s), this one
            // at 47 % of the pipe: with two waves per SIMD (the tile count allows no more) the LDS reads and the stage barrier are not
            // hidden behind sixteen-cycle instructions.  Off by default.
            const char* e4 = getenv("EMAGLS_GRAM_MFMA4");   // (read at every launch: a test switches forms inside one process)
            const bool four = e4 && e4[0] == '1';
            if (four) gram_lds4_kernel<<<bgrid(ntiles), 256, 0, st>>>((const double*)Yc, ld, S, gram_dpad(D, S), nbt, (double*)G, (double*)R, Sh, batch_ctx().stride);
            else {
                gram_lds_kernel<<<bgrid(ntiles), 256, 0, st>>>((const double*)Yc, ld, S, gram_dpad(D, S), nbt, (double*)G, (double*)R, Sh, batch_ctx().stride, xcd_runs_enabled());
            }
            KERNEL_CHECK();
            return;
        }
    }
    gram_mfma_kernel<T><<<bgrid(dim3(ntiles, ks)), 256, 0, st>>>((const T*)Yc, ld, S, kc, nbt, (T*)Gp, batch_ctx().stride);
    KERNEL_CHECK();
    gram_reduce_kernel<T><<<bgrid((unsigned)ceil_div((int64_t)S * S, 256)), 256, 0, st>>>((const T*)Gp, S, ks, (T*)G, (T*)R, Sh, batch_ctx().stride);
    KERNEL_CHECK();
}
// Gy = Yc^H Yc (S x S, may be null) and R = its leading Sh x Sh block (compact; may be null), ready for launch_cholesky(R, Sh)
void launch_gram(const void* Yc, int64_t D, int S, int64_t ld, bool is_cplx, void* Gp, void* G, void* R, int Sh, hipStream_t st) {
    if (is_cplx) gram_impl<cplx>(Yc, D, S, ld, Gp, G, R, Sh, st); else gram_impl<double>(Yc, D, S, ld, Gp, G, R, Sh, st);
}

// EMAGLS_CHOL_FUSED=1 selects the single-launch form below.  Measured (round 4, config 3, S_h = 256, 8 designs per launch): 448 us
// against 290 us for the 22 launches of the three-kernel form alone, 1.1 ms against ~0.4 ms next to other batches (one resident
// workgroup per design waits for a CU with 57 KB of LDS to spare), no difference in filter sets/s -- the diagonal block's 32
// dependent elimination steps dominate either way and the single workgroup adds the panel and the update to the same chain.
// Faster only for small factors (S_h = 121: 152 us).  Off by default.
static bool chol_fused_enabled() { static const bool on = [] { const char* e = getenv("EMAGLS_CHOL_FUSED"); return e && e[0] == '1'; }(); return on; }
template <typename T> static void chol_impl(void* G, int S, int* flag, hipStream_t st) {
    {   // one resident workgroup per design while the panel's rows fit the LDS (S <= 544 real, 288 complex)
        const int W = std::max(S - NB, 0), ldp = (W + 4 + 3) / 4 * 4 + 4;
        const size_t dyn = sizeof(T) * (size_t)NB * ldp;
        if (chol_fused_enabled() && S > NB && dyn <= 130 * 1024) {
            static PerDeviceOnce attr_once;
            if (attr_once.first()) {
                HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(chol_fused_kernel<double>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
                HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(chol_fused_kernel<cplx>), hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024));
            }
            chol_fused_kernel<T><<<bgrid(1), 256, dyn, st>>>((T*)G, S, flag, ldp, batch_ctx().stride);
            KERNEL_CHECK();
            return;
        }
    }
    for (int j0 = 0; j0 < S; j0 += NB) {
        const int rem = S - j0;
        const int ncb = (rem + NB - 1) / NB;  // column blocks incl. the diagonal one
        chol_diag_kernel<T><<<bgrid(1), 256, 0, st>>>((T*)G, S, j0, flag, batch_ctx().stride);
        KERNEL_CHECK();
        const int nbt = ncb - 1;
        if (nbt > 0) {
            chol_panel_kernel<T><<<bgrid(nbt), 256, 0, st>>>((T*)G, S, j0, batch_ctx().stride);
            KERNEL_CHECK();
            chol_update_kernel<T><<<bgrid(nbt * (nbt + 1) / 2), 256, 0, st>>>((T*)G, S, j0, nbt, batch_ctx().stride);
            KERNEL_CHECK();
        }
    }
}
void launch_cholesky(void* G, int S, bool is_cplx, int* flag, hipStream_t st) {
    if (is_cplx) chol_impl<cplx>(G, S, flag, st); else chol_impl<double>(G, S, flag, st);
}

// Ill-conditioned swept bins when Q is not materialised: Y_reg_inv = conj(Q) Z_k = conj(Yc) (Z_k R^-H), so the rows
// Z[kb][c][:] of the flagged bins are replaced by zs with zs R^H = z (R upper triangular: columns from the back).
//   block J (from the last):  acc(c, jj) = z(c, j0+jj) - sum_{i >= j0+32} zs(c, i) conj(R(j0+jj, i))
//                             zs(c, j0+k) = sum_jj acc(c, jj) conj(Rinv_J(k, jj))
// One workgroup per flagged bin (rare: tiny arrays); thread = (c, 8 column slices).
__global__ void __launch_bounds__(256) zsolve_flagged_kernel(cplx* __restrict__ Z, int ldS, const cplx* __restrict__ R,
                                                             const cplx* __restrict__ Rinv, const double* __restrict__ cond_ok,
                                                             int S, int C, int k0, size_t bstride) {
    Z = boff(Z, bstride); R = boff(R, bstride); Rinv = boff(Rinv, bstride); cond_ok = boff(cond_ok, bstride);
    const int kb = k0 + blockIdx.x;
    if (cond_ok[kb] != 0.0) return;
    __shared__ cplx as[32][33];
    __shared__ cplx ri[32][33];
    const int c = threadIdx.x >> 3, part = threadIdx.x & 7;
    cplx* z = Z + ((int64_t)kb * C + (c < C ? c : 0)) * ldS;
    const int nblk = (S + NB - 1) / NB;
    for (int J = nblk - 1; J >= 0; --J) {
        const int j0 = J * NB;
        for (int idx = threadIdx.x; idx < NB * NB; idx += 256) ri[idx / NB][idx % NB] = Rinv[(int64_t)J * NB * NB + idx];
        for (int m = 0; m < 4; ++m) {
            const int jj = part + 8 * m, j = j0 + jj;
            cplx acc = mk(0, 0);
            if (c < C && j < S) {
                acc = z[j];
                const cplx* rrow = R + (int64_t)j * S;
                for (int i = j0 + NB; i < S; ++i) { cplx p = mk(0, 0); cfma(p, z[i], conj(rrow[i])); acc = acc - p; }
            }
            as[c][jj] = acc;
        }
        __syncthreads();
        for (int m = 0; m < 4; ++m) {
            const int k = part + 8 * m;
            if (c < C && j0 + k < S) {
                cplx x = mk(0, 0);
                for (int jj = 0; jj < NB; ++jj) cfma(x, as[c][jj], conj(ri[k][jj]));
                z[j0 + k] = x;
            }
        }
        __threadfence_block();
        __syncthreads();
    }
}

void launch_zsolve_flagged(void* Z, int ldS, const void* R, const void* Rinv, const double* cond_ok, int S, int C, int P, int k0,
                           hipStream_t st) {
    if (P - k0 <= 0) return;
    if (C > 32) throw Error(2, "zsolve: more than 32 channels");
    zsolve_flagged_kernel<<<bgrid(P - k0), 256, 0, st>>>((cplx*)Z, ldS, (const cplx*)R, (const cplx*)Rinv, cond_ok, S, C, k0,
                                                        batch_ctx().stride);
    KERNEL_CHECK();
}

template <typename T> static void qform_impl(const void* Yc, const void* R, void* Rinv, int S, int64_t D, int64_t ld, void* Q,
                                             hipStream_t st) {
    auto lds = [&](int ch, int tr) { return sizeof(T) * ((size_t)tr * (S + 1) + (size_t)ch * 33 + 32 * 33 + tr * 33); };
    static PerDeviceOnce attr_once;   // (function attributes are per device)
    if (attr_once.first()) {
        HIP_CHECK(hipFuncSetAttribute((const void*)qform_kernel<T, 128, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        HIP_CHECK(hipFuncSetAttribute((const void*)qform_kernel<T, 32, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
        HIP_CHECK(hipFuncSetAttribute((const void*)qform_kernel<T, 32, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    }
    rinv_diag_kernel<T><<<bgrid((unsigned)ceil_div(S, NB)), 64, 0, st>>>((const T*)R, S, (T*)Rinv, batch_ctx().stride);
    KERNEL_CHECK();
    if (D <= 1024 && lds(32, 4) <= 80 * 1024)   // a few rows (the least-squares rows of an array design): small footprint, more workgroups
        qform_kernel<T, 32, 4><<<bgrid((unsigned)ceil_div(D, 4)), 128, lds(32, 4), st>>>((const T*)Yc, (const T*)R, (const T*)Rinv, S, D, ld, (T*)Q, batch_ctx().stride);
    else if (lds(128, 8) <= 150 * 1024)
        qform_kernel<T, 128, 8><<<bgrid((unsigned)ceil_div(D, 8)), 256, lds(128, 8), st>>>((const T*)Yc, (const T*)R, (const T*)Rinv, S, D, ld, (T*)Q, batch_ctx().stride);
    else if (lds(32, 8) <= 150 * 1024)
        qform_kernel<T, 32, 8><<<bgrid((unsigned)ceil_div(D, 8)), 256, lds(32, 8), st>>>((const T*)Yc, (const T*)R, (const T*)Rinv, S, D, ld, (T*)Q, batch_ctx().stride);
    else
        throw Error(2, "qform: too many SH channels for the LDS-resident rows");
    KERNEL_CHECK();
}
void launch_qform(const void* Yc, const void* R, void* Rinv, int S, int64_t D, int64_t ld, bool is_cplx, void* Q, hipStream_t st) {
    if (is_cplx) qform_impl<cplx>(Yc, R, Rinv, S, D, ld, Q, st); else qform_impl<double>(Yc, R, Rinv, S, D, ld, Q, st);
}

void launch_tn(const void* R, const void* E, int S, int C, int ldE, int nOrders, bool is_cplx, void* Tn, int64_t ldS,
               hipStream_t st) {
    if (is_cplx) tn_kernel<cplx><<<bgrid(dim3(nOrders, C)), 256, 0, st>>>((const cplx*)R, (const cplx*)E, S, C, ldE, (cplx*)Tn, ldS, batch_ctx().stride);
    else tn_kernel<double><<<bgrid(dim3(nOrders, C)), 256, 0, st>>>((const double*)R, (const double*)E, S, C, ldE, (double*)Tn, ldS, batch_ctx().stride);
    KERNEL_CHECK();
}

void launch_small_gemm(const void* A, int lda, bool a_cplx, const void* B, int ldb, bool b_cplx, void* Cm, int ldc,
                       bool c_cplx, int M, int N, int K, hipStream_t st) {
    dim3 grid((unsigned)ceil_div(N, 256), M);
    if (a_cplx && b_cplx && c_cplx)
        small_gemm_kernel<cplx, cplx, cplx><<<bgrid(grid), 256, 0, st>>>((const cplx*)A, lda, (const cplx*)B, ldb, (cplx*)Cm, ldc, M, N, K, batch_ctx().stride);
    else if (a_cplx && !b_cplx && c_cplx)
        small_gemm_kernel<cplx, double, cplx><<<bgrid(grid), 256, 0, st>>>((const cplx*)A, lda, (const double*)B, ldb, (cplx*)Cm, ldc, M, N, K, batch_ctx().stride);
    else if (a_cplx && !b_cplx && !c_cplx)
        small_gemm_kernel<cplx, double, double><<<bgrid(grid), 256, 0, st>>>((const cplx*)A, lda, (const double*)B, ldb, (double*)Cm, ldc, M, N, K, batch_ctx().stride);
    else if (!a_cplx && !b_cplx && !c_cplx)
        small_gemm_kernel<double, double, double><<<bgrid(grid), 256, 0, st>>>((const double*)A, lda, (const double*)B, ldb, (double*)Cm, ldc, M, N, K, batch_ctx().stride);
    else
        throw Error(2, "small_gemm: unsupported type combination");
    KERNEL_CHECK();
}

// MagLS on the persistent sweep: M = (R^H R)^-1 = R^-1 R^-H from the Cholesky factor of the basis' Gram matrix (C = S <= 32),
// written to every bin's slot (the sweep kernel reads M of bin kb at slot kb - 1), and cond_ok = 1 for every bin.  A factor
// whose diagonal spans more than 1e6 raises status[4] ("take the launch-per-bin sweep for this call"; a word of its own, not the
// sweep's residency time-out status[1]): the reference's pinv would drop singular values there, which the inverse cannot
// follow.  One workgroup.
template <typename T>
__global__ void __launch_bounds__(256) magls_m_kernel(const T* __restrict__ R, int C, int P, cplx* __restrict__ Mw, double* __restrict__ cond_ok,
                                                      int* __restrict__ status, size_t bstride) {
    R = boff(R, bstride); Mw = boff(Mw, bstride); cond_ok = boff(cond_ok, bstride); status = boff(status, bstride);
    __shared__ cplx Ri[32][33];   // R^-1 (upper triangular)
    __shared__ cplx Ms[32][33];
    const int tid = threadIdx.x;
    for (int idx = tid; idx < 32 * 33; idx += 256) Ri[idx / 33][idx % 33] = mk(0.0, 0.0);
    __syncthreads();
    if (tid < C) {   // column j of the inverse by back substitution:  sum_k R[i][k] X[k][j] = delta_ij
        const int j = tid;
        for (int i = j; i >= 0; --i) {
            cplx acc = mk(i == j ? 1.0 : 0.0, 0.0);
            for (int k = i + 1; k <= j; ++k) { cplx t = mk(0.0, 0.0); cfma(t, mk(1.0, 0.0) * R[(size_t)i * C + k], Ri[k][j]); acc = acc - t; }
            Ri[i][j] = cdiv(acc, mk(1.0, 0.0) * R[(size_t)i * C + i]);
        }
    }
    if (tid == 0) {
        double dmin = INFINITY, dmax = 0.0;
        for (int i = 0; i < C; ++i) { const double v = norm2(mk(1.0, 0.0) * R[(size_t)i * C + i]); dmin = fmin(dmin, v); dmax = fmax(dmax, v); }
        if (!(dmin > 1e-12 * dmax)) atomicExch(status + 4, 1);
    }
    __syncthreads();
    for (int idx = tid; idx < C * C; idx += 256) {
        const int i = idx / C, j = idx % C;
        cplx acc = mk(0.0, 0.0);
        for (int k = (i > j ? i : j); k < C; ++k) cfma(acc, Ri[i][k], conj(Ri[j][k]));
        Ms[i][j] = acc;
    }
    __syncthreads();
    for (int64_t idx = tid; idx < (int64_t)P * C * C; idx += 256) {
        const int f = (int)(idx % (C * C));
        Mw[idx] = Ms[f / C][f % C];
    }
    for (int kb = tid; kb < P; kb += 256) cond_ok[kb] = 1.0;
}
void launch_magls_m(const void* R, int C, bool is_cplx, int P, void* Mw, double* cond_ok, int* status, hipStream_t st) {
    if (C > 32) throw Error(2, "MagLS: more than 32 channels is not supported");
    if (is_cplx) magls_m_kernel<cplx><<<bgrid(1), 256, 0, st>>>((const cplx*)R, C, P, (cplx*)Mw, cond_ok, status, batch_ctx().stride);
    else magls_m_kernel<double><<<bgrid(1), 256, 0, st>>>((const double*)R, C, P, (cplx*)Mw, cond_ok, status, batch_ctx().stride);
    KERNEL_CHECK();
}

}  // namespace emagls
