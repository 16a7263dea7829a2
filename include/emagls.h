/* emagls.h -- C ABI of the MI355X-native eMagLS filter-design / binaural-render library.
 *
 * Drop-in boundary for the reference's MATLAB entry points (paths relative to thomasdeppisch/eMagLS):
 *
 *   emagls_get_ls_filters              <-  lib/getLsFilters.m:1-2
 *   emagls_get_magls_filters           <-  lib/getMagLsFilters.m:1-2
 *   emagls_get_emagls_filters          <-  lib/getEMagLsFilters.m:1-2
 *   emagls_get_emagls2_filters         <-  lib/getEMagLs2Filters.m:1-2
 *   emagls_get_emagls_filters_from_atf <-  lib/getEMagLsFiltersFromAtf.m:1
 *   emagls_get_emagls_filters_ema_in_ch <- lib/getEMagLsFiltersEMAinCH.m:1-2  (default chFunction @getCH, dependencies/getCH.m)
 *   emagls_binaural_decode[_complex]   <-  dependencies/binauralDecode.m:1-2 (core loop :33-42,53-64)
 *   emagls_sh_basis                    <-  getSH (polarch/Spherical-Harmonic-Transform, call site lib/getLsFilters.m:30)
 *   emagls_modal_bn                    <-  sphModalCoeffs (polarch/Array-Response-Simulator, call site dependencies/getSMAIRMatrix.m:107)
 *
 * Conventions (identical to the MATLAB side, so a MEX gateway passes mxGetDoubles() pointers through):
 *   - all arrays are column-major FP64; complex arrays are interleaved (re,im) pairs
 *     (MATLAB R2018a+ interleaved complex API, numpy complex128);
 *   - HRIRs are [numSamples x numDirections]; filters come back [len x numChannels];
 *   - angles in radians, zenith (0..pi), not elevation;
 *   - basis: EMAGLS_BASIS_REAL -> real outputs (double), EMAGLS_BASIS_COMPLEX -> complex outputs;
 *   - pointers may be host or device pointers (copies use hipMemcpyDefault); outputs are caller-allocated;
 *   - every function returns EMAGLS_OK or an error code; emagls_last_error() gives the message
 *     (the MEX shim forwards it to mexErrMsgIdAndTxt);
 *   - the default shFunction (@getSH) is built in; a custom MATLAB shFunction handle cannot cross a C ABI: the wrapper
 *     evaluates it and calls the *_with_basis entry points with the resulting matrices.
 *
 * Threading: one host thread per plan; one process per GPU for multi-GPU batches.
 */
#ifndef EMAGLS_H
#define EMAGLS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EMAGLS_OK 0
#define EMAGLS_ERR_ARG 1         /* invalid argument; mirrors the reference's assert()s, e.g. "len too short" */
#define EMAGLS_ERR_UNSUPPORTED 2 /* shape outside what this build supports */
#define EMAGLS_ERR_HIP 3         /* HIP / hipFFT runtime failure */
#define EMAGLS_ERR_NUMERIC 4     /* e.g. SH Gram matrix of the HRIR grid not positive definite */

#define EMAGLS_BASIS_REAL 0
#define EMAGLS_BASIS_COMPLEX 1

#define EMAGLS_KIND_LS 0
#define EMAGLS_KIND_MAGLS 1
#define EMAGLS_KIND_EMAGLS 2
#define EMAGLS_KIND_EMAGLS2 3
#define EMAGLS_KIND_FROM_ATF 4
#define EMAGLS_KIND_EMA_CH 5   /* equatorial array, output in circular harmonics (2*order+1 channels) */
#define EMAGLS_KIND_MAGLS_2D 6 /* MagLS on a horizontal HRIR set in circular harmonics (2*order+1 channels) */
#define EMAGLS_KIND_EMA_SH 7   /* equatorial array, output in spherical harmonics ((order+1)^2 channels) */

const char* emagls_last_error(void);
int emagls_version(void);
int emagls_device_count(int* count);
int emagls_set_device(int device);

/* The one-shot entry points below keep the plans of their most recent shapes alive (device buffers, captured hipGraphs; at most
 * EMAGLS_PLAN_CACHE plans, default 4, 0 disables), and emagls_binaural_decode its hipFFT plans and work buffers.  This call
 * releases all of it (a MEX gateway registers it with mexAtExit), together with the resident chunks of the job lists and the pool of
 * device-memory blocks that plans and batches hand back. */
int emagls_cache_clear(void);
/* The same without the block pool: every resident plan, batch and job chunk goes, their device memory stays with the library for the
 * next designs (a long-running process that moves on to another study: fresh device memory is what costs -- hipMalloc of a few GB took
 * 0.3 ms on some boxes and seconds on others, profiles/r06_cold_path.md). */
int emagls_cache_release_designs(void);

/* Measured FP64 peak of the current device in TFLOP/s (best of a few launches that keep every CU busy): which = 0 the matrix
 * pipe on v_mfma_f64_16x16x4_f64 (the shape the pipeline's GEMM kernels issue), which = 1 the vector pipe (v_fma_f64), which = 2
 * the matrix pipe on v_mfma_f64_4x4x4_4b_f64 -- on gfx950 the shape that reaches the pipe's nominal rate (75 of 78.6 TFLOP/s; the
 * 16 x 16 x 4 shape sustains 49).  bench.py prices its executed flops against the vector figure. */
int emagls_fp64_peak_tflops(int which, double* tflops);
/* The same with the launch length chosen: burst != 0 times launches of <= 1 ms (before the chip settles at its sustained power
 * state), burst == 0 the ~10 ms launches of the call above; shader_mhz (optional) receives the shader clock the timed loop ran
 * at (in-kernel cycle counter over the 100 MHz wall counter). */
int emagls_fp64_peak_tflops_ex(int which, int burst, double* tflops, double* shader_mhz);
/* Device-side self tests of building blocks that have no entry point of their own.  which = 0: the wave reduction of the
 * register-resident sweep (permlane swaps and DPP steps against a plain sum; max_err: largest absolute difference); which = 1 / 2:
 * the LDS-staged Gram tile on v_mfma_f64_16x16x4 / on v_mfma_f64_4x4x4_4b against a host sum (max_err relative to the largest element). */
int emagls_self_test(int which, double* max_err);

/* ---- kernel-level entry points ------------------------------------------------------------- */

/* Y [ndirs x (order+1)^2], column-major; real (8 B) or interleaved complex (16 B) per entry. */
int emagls_sh_basis(int order, int64_t ndirs, const double* azi, const double* zen, int basis, void* Y);

/* Same kernel on buffers that are already in HBM, enqueued on `stream` (a hipStream_t, NULL = default
 * stream) without any staging copy or synchronisation: used for the bandwidth measurement of the
 * SH-basis assembly and by callers that keep their data on the device. */
int emagls_sh_basis_device(int order, int64_t ndirs, const double* d_azi, const double* d_zen, int basis, void* d_Y,
                           void* stream);

/* bn [nfreq x (order+1)] interleaved complex, column-major: rigid-sphere modal coefficients b_n(kr). */
int emagls_modal_bn(int order, int64_t nfreq, const double* kr, void* bn);

/* ---- one-shot filter design (signatures follow the MATLAB functions) ------------------------ */

int emagls_get_ls_filters(const double* hL, const double* hR, int64_t nsamp, int64_t ndirs,
                          const double* hrir_azi, const double* hrir_zen, int order, int basis,
                          void* wL, void* wR /* [nsamp x (order+1)^2] */);

int emagls_get_magls_filters(const double* hL, const double* hR, int64_t nsamp, int64_t ndirs,
                             const double* hrir_azi, const double* hrir_zen, int order, double fs, int64_t len,
                             int basis, void* wL, void* wR /* [len x (order+1)^2] */);

int emagls_get_emagls_filters(const double* hL, const double* hR, int64_t nsamp, int64_t ndirs,
                              const double* hrir_azi, const double* hrir_zen, double mic_radius,
                              const double* mic_azi, const double* mic_zen, int64_t nmics, int order, double fs,
                              int64_t len, int basis, void* wL, void* wR /* [len x (order+1)^2] */);

int emagls_get_emagls2_filters(const double* hL, const double* hR, int64_t nsamp, int64_t ndirs,
                               const double* hrir_azi, const double* hrir_zen, double mic_radius,
                               const double* mic_azi, const double* mic_zen, int64_t nmics, int order, double fs,
                               int64_t len, int basis, void* wL, void* wR /* [len x nmics] */);

/* Equatorial microphone array (all microphones at zenith pi/2), filters in circular harmonics ordered
 * [C_0, C_-1, C_1, ..., C_-N, C_N]: wL, wR [len x (2*order+1)], real or complex like the basis. */
int emagls_get_emagls_filters_ema_in_ch(const double* hL, const double* hR, int64_t nsamp, int64_t ndirs,
                                        const double* hrir_azi, const double* hrir_zen, double mic_radius,
                                        const double* mic_azi, int64_t nmics, int order, double fs, int64_t len, int basis,
                                        void* wL, void* wR);

/* The same designs with a custom shFunction (lib/getEMagLsFilters.m:32,68; dependencies/getSMAIRMatrix.m:101): a function handle
 * cannot cross a C ABI, so the MATLAB-side wrapper evaluates it and passes the matrices -- Y_hrir = shFunction(n, [azi zen],
 * shDefinition) [ndirs x (n+1)^2] and Y_mic = shFunction(n, micGrid, shDefinition) [nmics x (n+1)^2], column-major, real or
 * interleaved complex like `basis`, with n = emagls_simulation_order(kind, order, fs, mic_radius) (n = order for LS / MagLS). */
int emagls_simulation_order(int kind, int order, double fs, double mic_radius);
int emagls_get_ls_filters_with_basis(const double* hL, const double* hR, int64_t nsamp, int64_t ndirs, const void* Y_hrir, int order,
                                     int basis, void* wL, void* wR);
int emagls_get_magls_filters_with_basis(const double* hL, const double* hR, int64_t nsamp, int64_t ndirs, const void* Y_hrir, int order,
                                        double fs, int64_t len, int basis, void* wL, void* wR);
int emagls_get_emagls_filters_with_basis(const double* hL, const double* hR, int64_t nsamp, int64_t ndirs, const void* Y_hrir,
                                         double mic_radius, const void* Y_mic, int64_t nmics, int order, double fs, int64_t len,
                                         int basis, void* wL, void* wR);
int emagls_get_emagls2_filters_with_basis(const double* hL, const double* hR, int64_t nsamp, int64_t ndirs, const void* Y_hrir,
                                          double mic_radius, const void* Y_mic, int64_t nmics, int order, double fs, int64_t len,
                                          int basis, void* wL, void* wR);

/* lib/getEMagLsFiltersEMAinSH.m:1-2 -- [wMlsL, wMlsR] = getEMagLsFiltersEMAinSH(hL, hR, hrirGridAziRad, hrirGridZenRad, micRadius,
 * micGridAziRad, order, fs, len, shDefinition): equatorial array (zenith pi/2 for every microphone), filters [len x (order+1)^2] */
int emagls_get_emagls_filters_ema_in_sh(const double* hL, const double* hR, int64_t nsamp, int64_t ndirs, const double* azi,
                                        const double* zen, double mic_radius, const double* mic_azi, int64_t nmics, int order,
                                        double fs, int64_t len, int basis, void* wL, void* wR);

/* atf_irs [atf_taps x nmics x natf]; outputs real [filter_len x nmics];
 * mean_grid_dev_deg (optional) receives the value the reference prints (getEMagLsFiltersFromAtf.m:96). */
int emagls_get_emagls_filters_from_atf(const double* hL, const double* hR, int64_t nsamp, int64_t ndirs,
                                       const double* hrir_azi, const double* hrir_zen, const double* atf_irs,
                                       int64_t atf_taps, int64_t nmics, int64_t natf, const double* atf_azi,
                                       const double* atf_zen, double fs, int64_t filter_len, double f_trans,
                                       double* wL, double* wR, double* mean_grid_dev_deg);

/* out [nsamp_out x 2] real, nsamp_out = nsamp (compensate_delay == 0) or nsamp - len/2 + 1 (!= 0).
 * in [nsamp x nch], wL/wR [len x nch], all real. */
int emagls_binaural_decode(const double* in, int64_t nsamp, int64_t nch, const double* wL, const double* wR,
                           int64_t len, int compensate_delay, double* out);

/* Complex-SH rendering (dependencies/binauralDecode.m:39-42,59-64: complex products accumulated, real part kept).
 * in [nsamp x nch] and wL / wR [len x nch] are interleaved complex where the flag says so, real otherwise; out as above, real.
 * imag_abs_sum (optional, [2]) receives sum(abs(imag(.))) of the discarded imaginary part per ear, the two numbers the
 * reference prints in its warning (:61-62) -- over the samples that are returned, i.e. after the compensate_delay cut (:53-57). */
int emagls_binaural_decode_complex(const void* in, int in_is_complex, int64_t nsamp, int64_t nch, const void* wL, const void* wR,
                                   int filters_are_complex, int64_t len, int compensate_delay, double* out, double* imag_abs_sum);

/* The render loop on buffers that are already in HBM (no staging copies, no allocation after the first call of a shape): d_in
 * [nsamp x nch], d_wL / d_wR [len x nch] real or interleaved complex as flagged, d_out [nsamp x 2] real (no delay cut: the caller
 * offsets its read).  Enqueued on `stream` (hipStream_t, NULL = default) and synchronised before returning (the hipFFT work
 * buffers are shared per process).  What bench.py times for north_star item (iii). */
int emagls_binaural_decode_device(const void* d_in, int in_is_complex, int64_t nsamp, int64_t nch, const void* d_wL, const void* d_wR,
                                  int filters_are_complex, int64_t len, double* d_out, double* imag_abs_sum, void* stream);

/* The three designs with a covariance constraint in the place of the `applyDiffusenessConst` argument the reference's
 * functions used to take after `len` (verifyEMagLs.m:106-114 still shows the call form).  OWN SPECIFICATION, not the reference's
 * implementation: that code is not in the snapshot (CHANGELOG.md:10-12).  Per solved bin the two ears' filters are mixed by the
 * Hermitian positive definite 2x2 matrix M with M Rhat M = R (Rhat: ear covariance of the rendered HRTFs over the HRIR grid,
 * R: that of the time-aligned HRTFs) -- the closed form of Zaunschirm/Schoerkhuber/Hoeldrich 2018's covariance constraint.
 * The reference's surviving *_wDC fixtures agree with it for eMagLS / eMagLS2 (their mixing is predicted to 4e-3) and do NOT
 * for MagLS (interaural cross term off by 6e-2; candidates tried and rejected: DESIGN.md section 7).  Same arguments as the
 * functions without the suffix plus the flag. */
int emagls_get_magls_filters_dc(const double* hL, const double* hR, int64_t nsamp, int64_t ndirs, const double* azi, const double* zen,
                                int order, double fs, int64_t len, int apply_diffuseness_const, int basis, void* wL, void* wR);
int emagls_get_emagls_filters_dc(const double* hL, const double* hR, int64_t nsamp, int64_t ndirs, const double* azi, const double* zen,
                                 double mic_radius, const double* mic_azi, const double* mic_zen, int64_t nmics, int order, double fs,
                                 int64_t len, int apply_diffuseness_const, int basis, void* wL, void* wR);
int emagls_get_emagls2_filters_dc(const double* hL, const double* hR, int64_t nsamp, int64_t ndirs, const double* azi, const double* zen,
                                  double mic_radius, const double* mic_azi, const double* mic_zen, int64_t nmics, int order, double fs,
                                  int64_t len, int apply_diffuseness_const, int basis, void* wL, void* wR);

/* ---- render side: what the reference's harness runs between the recording and the decoder ---- */

/* lib/getMagLsFilters2D.m:1 -- [wMlsL, wMlsR] = getMagLsFilters2D(hLHor, hRHor, horHrirGridAziRad, order, fs, len, chDefinition)
 * hL/hR [nsamp x ndirs]; wL/wR [len x 2*order+1], channels [C_0, C_-1, C_1, ..., C_-N, C_N] (dependencies/getCH.m:17-28),
 * real, or interleaved complex for basis == EMAGLS_BASIS_COMPLEX. */
int emagls_get_magls_filters_2d(const double* hL, const double* hR, int64_t nsamp, int64_t ndirs, const double* hrir_azi,
                                int order, double fs, int64_t len, int basis, void* wL, void* wR);

#define EMAGLS_RADIAL_TIKHONOV 0
#define EMAGLS_RADIAL_SOFTLIMIT 1
#define EMAGLS_RADIAL_FULL 2
#define EMAGLS_RADIAL_NONE 3
/* dependencies/getRadialFilter.m:1 -- radFilts = getRadialFilter(params), plane-wave model, rigid sphere.
 * nfft = oversampling * ir_len; rad: interleaved complex [nfft/2+1 x order+1] column-major.  regul_const is read by
 * tikhonov, noise_gain_db by softlimit.  Entries the reference computes as 0/0 or 1/0 (orders > 0 at DC, softlimit / full)
 * come back as NaN / Inf+NaN i like there. */
int emagls_get_radial_filter(int order, double fs, double sma_radius, int64_t ir_len, int oversampling, int filter_type,
                             double regul_const, double noise_gain_db, void* rad);
/* dependencies/applyRadialFilter.m:1 -- outSig = applyRadialFilter(inSig, params) with params.nfft = oversampling * ir_len
 * (verifyEMagLs.m:250).  sig [nsamp x (order+1)^2] real; out [emagls_apply_radial_filter_rows(...) x (order+1)^2]:
 * the signal (zero-padded to nfft if shorter) filtered per SH order, the filter delay nfft/2 removed. */
int64_t emagls_apply_radial_filter_rows(int64_t nsamp, int64_t ir_len, int oversampling);
int emagls_apply_radial_filter(const double* sig, int64_t nsamp, int order, double fs, double sma_radius, int64_t ir_len,
                               int oversampling, int filter_type, double regul_const, double noise_gain_db, double* out);
/* verifyEMagLs.m:235-236 -- E = getSH(order, micGrid, shDefinition).'; shRecording = smaRecording * pinv(E)
 * sig [nsamp x nmics] real; out [nsamp x (order+1)^2], real or interleaved complex like the basis. */
int emagls_sh_encode(const double* sig, int64_t nsamp, int64_t nmics, const double* mic_azi, const double* mic_zen, int order,
                     int basis, void* out);
/* dependencies/getCH.m:1 -- Y = getCH(N, aziRad, basisType): [ndirs x 2N+1] column-major, channels [C_0, C_-1, C_1, ..., C_-N, C_N],
 * real or interleaved complex like the basis */
int emagls_ch_basis(int order, int64_t ndirs, const double* azi, int basis, void* Y);
/* dependencies/getSMAIRMatrix.m:1 -- smairMat = getSMAIRMatrix(params), plane-wave model of a rigid sphere, built-in getSH.
 * The filter designs never form this array (they work on its factors); this entry materialises it for callers that use the
 * array model itself.  nfft = oversampling * ir_len (even); smair: interleaved complex [rows x S x nfft/2+1] column-major,
 * rows = (order+1)^2, or nmics with return_raw_mic_sigs; S = (sim_order+1)^2 with sim_order = max(order, ceil(fs*pi*r/343))
 * returned through sim_order (optional).  radial_filter_type other than EMAGLS_RADIAL_NONE applies getRadialFilter to the
 * SH-domain model (:129-138). */
int emagls_get_smair_matrix(int order, double fs, int64_t ir_len, int oversampling, double sma_radius, const double* mic_azi,
                            const double* mic_zen, int64_t nmics, int basis, int return_raw_mic_sigs, int radial_filter_type,
                            double regul_const, double noise_gain_db, void* smair, int* sim_order);
/* lib/getMagLsSphericalHeadFilter.m:1 -- [wShf, W_Shf] = getMagLsSphericalHeadFilter(micRadius, order, fs, len)
 * w_shf [len]; W_shf (optional) [emagls_eq_filter_nfft(len)] real, the mirrored zero-phase spectrum. */
int64_t emagls_eq_filter_nfft(int64_t len);
int emagls_get_magls_spherical_head_filter(double mic_radius, int order, double fs, int64_t len, double* w_shf, double* W_shf);
/* lib/getMagLsArrayDiffuseFilter.m:1 -- wAdf = getMagLsArrayDiffuseFilter(micRadius, micGridAziRad, micGridZenRad, order, fs,
 * len, shDefinition, shFunction).  Y_hi (optional) replaces the built-in getSH: the caller's shFunction evaluated at the
 * simulation order ceil(fs*pi*micRadius/343), [nmics x (simOrder+1)^2] column-major (the grid may then be NULL).  w_adf [len]. */
int emagls_get_magls_array_diffuse_filter(double mic_radius, const double* mic_azi, const double* mic_zen, int64_t nmics, int order,
                                          double fs, int64_t len, int basis, const void* Y_hi, double* w_adf);

/* ---- plan API: inputs resident in HBM, repeated execution (benchmarks, batches) ------------- */

typedef struct emagls_plan emagls_plan;

typedef struct emagls_design_desc {
    int kind;            /* EMAGLS_KIND_* */
    int basis;           /* EMAGLS_BASIS_* */
    int order;           /* SH output order N */
    double fs;           /* Hz */
    int64_t len;         /* filter length (ignored for LS) */
    int64_t nsamp;       /* HRIR taps */
    int64_t ndirs;       /* HRIR directions */
    double mic_radius;   /* EMAGLS / EMAGLS2 */
    int64_t nmics;       /* EMAGLS / EMAGLS2 / FROM_ATF */
    double f_trans;      /* FROM_ATF: transition frequency in Hz */
    int64_t atf_taps;    /* FROM_ATF */
    int64_t natf;        /* FROM_ATF: ATF directions */
    int custom_basis;    /* != 0: the SH matrices are supplied by emagls_plan_set_basis (a custom shFunction, lib/getEMagLsFilters.m:32,68) */
    int diffuseness;     /* != 0: apply the covariance constraint (own specification, see emagls_get_magls_filters_dc) in the
                          * place of the applyDiffusenessConst option the reference removed (CHANGELOG.md:10-12,
                          * verifyEMagLs.m:137-145); MAGLS, MAGLS_2D, EMAGLS, EMAGLS2, EMA_CH */
    int sim_order_pad;   /* EMAGLS / EMAGLS2 / EMA_CH, 0 = off: lay the design out for max(its own simulation order, this) SH orders
                          * with b_n = 0 above its own order (dependencies/getSMAIRMatrix.m:95,107: the same sum, the same filters).
                          * Array radii of neighbouring simulation-order classes then have ONE shape and share a lane batch
                          * (emagls_batch_create; BASELINE config 4: 256 radii = 32 batches of 8) */
} emagls_design_desc;

typedef struct emagls_plan_info {
    int nfft, num_pos_freqs, k_cut /* 1-based like the reference */, sim_order, num_sh_sim, num_channels;
    int out_is_complex;
    int64_t out_rows, out_cols;
    double grp_delay_l, grp_delay_r; /* valid after an execute + synchronize */
    double mean_grid_dev_deg;        /* FROM_ATF */
    int num_sweep_launches;
    int64_t device_bytes;
    /* routes of the per-bin factorisation of the array designs (0 elsewhere), 0-based bins: [1, hh_end) orthonormal route
     * (Householder QR + Jacobi SVD) on the lowest hh_orders orders, [gram_from, P) Gram route (gram_from == 0: none);
     * g_first: first bin whose direction-space operand G_k is formed */
    int gram_from, hh_end, hh_orders, g_first;
    int sim_order_own;               /* the design's own simulation order (sim_order is the padded one with sim_order_pad) */
    /* form of the phase sweep the next execute takes: 0 one launch per bin, 1 one resident launch on operands G_k materialised
     * in HBM, 2 one resident launch that evaluates its operands itself from the angles between HRIR directions and microphones
     * (array designs on the built-in SH basis; EMAGLS_SWEEP_SYNTH=0 selects form 1), 3 the same with the operand of a bin held
     * in registers by the waves that run the recurrence (up to 18 units; EMAGLS_SWEEP_REG=0 selects form 2; several such sweeps
     * share the device).  sweep_units: forms 2 and 3, the polynomial evaluations per direction and bin -- antipodal microphone
     * pairs count once (15 pairs + 2 single capsules on the em32). */
    int sweep_form, sweep_units;
} emagls_plan_info;

int emagls_plan_create(const emagls_design_desc* desc, emagls_plan** plan);
int emagls_plan_destroy(emagls_plan* plan);
int emagls_plan_set_hrir_grid(emagls_plan* plan, const double* azi, const double* zen);
/* zen is ignored (may be NULL) for EMAGLS_KIND_EMA_CH: an equatorial array, every microphone at pi/2 */
int emagls_plan_set_mic_grid(emagls_plan* plan, const double* azi, const double* zen);
int emagls_plan_set_hrirs(emagls_plan* plan, const double* hL, const double* hR);
/* custom_basis plans: Y_hrir [ndirs x S] and (array designs) Y_mic [nmics x S], column-major, real or interleaved complex like
 * the plan's basis, S = (emagls_simulation_order(...) + 1)^2; replaces the two grid setters */
int emagls_plan_set_basis(emagls_plan* plan, const void* Y_hrir, const void* Y_mic);
int emagls_plan_set_atfs(emagls_plan* plan, const double* atf_irs, const double* atf_azi, const double* atf_zen);
/* enqueue the whole design (SH basis ... windowed filters) on the plan's stream; returns immediately */
int emagls_plan_execute(emagls_plan* plan);
int emagls_plan_synchronize(emagls_plan* plan);
/* synchronise, check device-side status flags, copy the filters out */
int emagls_plan_get_filters(emagls_plan* plan, void* wL, void* wR);
int emagls_plan_get_info(emagls_plan* plan, emagls_plan_info* info);
/* the sweep form (emagls_plan_info.sweep_form) a batch of `designs` plans shaped like `plan` takes: what bench.py names its
 * dominant kernel by (the form is decided per launch, by the number of designs in it) */
int emagls_plan_sweep_form_in_batch(emagls_plan* plan, int designs, int* form);
/* 1..4: number of HIP streams one design may use (independent branches fork onto side streams; default 3,
 * best for the latency of ONE design; use 1 when several plans are in flight). Drops the captured graph. */
int emagls_plan_set_streams(emagls_plan* plan, int nstreams);
/* profiling: level 0 none, 1 = HIP events between stages, 2 = additionally around every sweep launch */
int emagls_plan_set_profiling(emagls_plan* plan, int level);
int emagls_plan_num_stages(emagls_plan* plan);
const char* emagls_plan_stage_name(emagls_plan* plan, int stage);
/* after execute + synchronize with profiling >= 1: per-stage milliseconds of the last execute */
int emagls_plan_stage_times(emagls_plan* plan, double* ms, int n);
/* profiling level 2: sum and count of per-launch sweep kernel durations (ms) of the last execute */
int emagls_plan_sweep_kernel_time(emagls_plan* plan, double* total_ms, int* launches);
/* copy an internal device buffer to the host (tests): returns its size in *nbytes when dst == NULL */
int emagls_plan_debug_buffer(emagls_plan* plan, const char* name, void* dst, size_t* nbytes);
/* the plan's hipStream_t, for callers that interleave their own work */
void* emagls_plan_stream(emagls_plan* plan);

/* ---- batches: several eMagLS / eMagLS2 designs of identical shape (different arrays / HRIR sets) ---------
 * Plans of identical shape (same simulation order: same array radius class) are executed in lane mode: their buffers
 * are moved into one arena at a constant stride and every launch of the pipeline covers all designs (grid.z = design);
 * the sequential sweep is one resident launch in which each design's workgroups occupy one XCD.  Other batches run the
 * per-design stages on the plans' own streams and share only the sweep.  At most 8 plans (one XCD per design in the sweep);
 * with EMAGLS_BATCH_MAX=16 in the environment up to 16 (two designs per XCD, two sweep workgroups per CU: for an otherwise
 * idle device only, see emagls_batch_create in capi.hip); the plans stay owned by the
 * caller and must outlive the batch.  Results: emagls_batch_get_filters, or emagls_plan_get_filters on each plan.
 * LS / MagLS / MagLS-2D plans of one kind, order and basis (up to 32 channels) form batches as well (lib/getLsFilters.m:30,
 * lib/getMagLsFilters.m:30 in a loop over HRIR sets): their stages run on the batch's stream and ONE resident sweep launch serves all designs; with
 * emagls_batch_set_geometry_sharing, sets on one grid compute the SH side once. */
typedef struct emagls_batch emagls_batch;
int emagls_batch_create(emagls_plan** plans, int nplans, emagls_batch** batch);
/* Largest batch emagls_batch_create accepts from now on: 8 by default (EMAGLS_BATCH_MAX in the environment sets the initial
 * value), up to 16 -- two designs per XCD in the resident sweep, two sweep workgroups per CU (154 of 160 KB of LDS): fastest per
 * design when the batch has the device to itself, but kernels of other batches then only find room on the CUs the sweep does not
 * use, so keep 8 when several batches are in flight.  *previous (optional) receives the old value. */
int emagls_set_batch_max(int max_designs, int* previous);
/* Batches of HRIR sets on ONE geometry (the loop over subjects around lib/getEMagLsFilters.m:32 / getEMagLs2Filters.m:32 /
 * getEMagLsFiltersEMAinCH.m:32 with the same grids and array): with sharing enabled, a batch whose plans agree in every
 * geometry input (compared on the device whenever a grid is replaced) runs the SH matrices, the array model, pwGrid_k and its
 * regularised inverses ONCE (plan 0) and per plan only what its HRIRs enter: spectra, least-squares rows, the sweep (on plan
 * 0's operands) and the epilogue.  Off by default: a batch then treats its designs as independent.  Plans that do not agree
 * (or kinds without the option: FromAtf / EMAinSH / more than 32 channels / designs with the covariance constraint) run
 * as before;
 * emagls_batch_shares_geometry reports what the last execute did. */
int emagls_batch_set_geometry_sharing(emagls_batch* batch, int enable);
int emagls_batch_shares_geometry(emagls_batch* batch, int* shared);
/* ---- job lists ---------------------------------------------------------------------------------------------------
 * Independent designs are the unit of parallelism of the reference's users: the loop over array radii, HRIR sets or subjects
 * around one of its functions (testEMagLs.m:75-95, testEMagLsFromAtfs.m:72-73).  One job = one design: its descriptor, its inputs
 * and room for its filters.  hL / hR / atf and wL / wR may be host or device buffers of the current device (device buffers are
 * read and written stream-ordered: nothing crosses PCIe); the grids are host arrays. */
typedef struct emagls_job {
    emagls_design_desc desc;
    const double* hL;           /* [nsamp x ndirs] */
    const double* hR;
    const double* hrir_azi;     /* [ndirs] */
    const double* hrir_zen;     /* [ndirs]; NULL: horizontal grid (MAGLS_2D) */
    const double* mic_azi;      /* [nmics], array designs */
    const double* mic_zen;      /* [nmics]; NULL: equatorial array (EMA_CH / EMA_SH) */
    const double* atf;          /* FROM_ATF: [atf_taps x nmics x natf] */
    const double* atf_azi;      /* FROM_ATF: [natf] */
    const double* atf_zen;
    void* wL;                   /* [len x channels] real, or interleaved complex for a complex basis (emagls_plan_info.out_is_complex) */
    void* wR;
} emagls_job;
/* Runs the whole list and returns when every job's filters are in place.  Consecutive jobs of one shape form chunks of up to
 * batch_size designs (<= 0: 32; more than 16 only for array designs that take the register-resident sweep) that run as lane
 * batches -- one launch of every kernel for the chunk, one resident sweep launch --; up to in_flight chunks (<= 0: 4) are between
 * upload and collection at any time, each driven by a thread of the library, so that uploads, launches and the collection of
 * results overlap with the GPU's work on the other chunks.  Plans and batches of chunks whose descriptors repeat stay resident
 * between calls (emagls_cache_clear releases them).  Same filters as the single calls.
 * flags: EMAGLS_JOBS_SHARE_GEOMETRY -- the designs of a chunk that agree in everything but their HRIRs (checked on the device) compute
 * the geometry stages once (emagls_batch_set_geometry_sharing; the filters are bit-identical to the independent designs').  eMagLS /
 * eMagLS2 designs with 33..64 channels run plan by plan: with the flag they are cut one per chunk and a plan keeps G_k, the per-bin
 * factors and Y_reg_inv_k from its last clean run while its own grids stay the same (a 64-capsule array: 31 -> 8 ms per set). */
#define EMAGLS_JOBS_SHARE_GEOMETRY 1
int emagls_jobs_run(const emagls_job* jobs, int64_t njobs, int batch_size, int in_flight, int flags);
/* ---- job lists over several GPUs ------------------------------------------------------------------------------------------
 * Independent jobs shard without any collective on the data path.  emagls_jobs_shard is the split every runner of this
 * library uses (emagls_amd/batch.py restated in C): array-radius studies -- jobs that differ only in mic_radius, up to 32 microphones
 * -- are sorted by simulation order (dependencies/getSMAIRMatrix.m:95), cut into lane batches of equal COST (on average max_batch
 * designs, <= 32; every design of a batch laid out for the batch's highest order: sim_order_pad[j], to be written into
 * desc.sim_order_pad) and whole batches go to ranks by longest processing time; any other job is a unit of its own.  rank_of_job[j]:
 * the rank of job j; order_in_rank[j] (optional): its position in that rank's share (the jobs of a batch adjacent, cheap batches
 * first); sim_order_pad (optional).  Needs no GPU. */
int emagls_jobs_shard(const emagls_job* jobs, int64_t njobs, int world, int max_batch, int* rank_of_job, int* order_in_rank, int* sim_order_pad);
/* The GPUs of ONE process: the list is split with emagls_jobs_shard over `ndevices` devices (their HIP ordinals in `devices`; the
 * same ordinal may appear twice) and every device's share runs through emagls_jobs_run from a host thread of its own.  Inputs and
 * outputs are host arrays (or memory every listed device can reach); there is no gather: each device writes its jobs' filters where
 * the jobs point.  What a MEX caller -- one MATLAB process -- uses for BASELINE config 4's 256 radii or config 5's subjects
 * (emagls_mex('jobs', jobs, batchSize, inFlight, shareGeometry, devices)); one process per GPU with an RCCL gather of device buffers
 * is emagls_amd/batch.py. */
int emagls_jobs_run_devices(const emagls_job* jobs, int64_t njobs, const int* devices, int ndevices, int batch_size, int in_flight, int flags);
/* Shape of a design's filters from its descriptor alone (no plan, no device memory: what a caller needs to allocate wL / wR of a
 * job): rows x cols, real or interleaved complex -- len x channels like the reference's outputs (lib/getEMagLsFilters.m:139-142);
 * LS keeps the HRIR length (lib/getLsFilters.m:33); channels = (N+1)^2 in the SH domain, 2N+1 circular harmonics (MAGLS_2D,
 * EMA_CH), the microphones for EMAGLS2 / FROM_ATF; complex for a complex basis except FROM_ATF.  The same numbers as
 * emagls_plan_info.out_rows / out_cols / out_is_complex of a plan of that descriptor. */
int emagls_design_out_shape(const emagls_design_desc* desc, int64_t* rows, int64_t* cols, int* is_complex);
/* Measurement hooks of the job lists (bench.py's roofline figure): level > 0 makes the chunks' batches bracket their sweep launch
 * with HIP events on the stream it is launched on; emagls_jobs_sweep_times then reports, for every resident chunk that ran since,
 * the duration of its LAST sweep launch (ms) and the designs it covered (count: chunks available, capacity: room in the arrays). */
int emagls_jobs_set_profiling(int level);
int emagls_jobs_sweep_times(double* ms, int* designs, int capacity, int* count);

/* HRIR sets on ONE grid (and, for the array kinds, ONE array) in one call -- the loop
 *     for i = 1:nsets, [wL(:,:,i), wR(:,:,i)] = getEMagLsFilters(hL(:,:,i), hR(:,:,i), grid..., array..., order, fs, len, shDefinition); end
 * around lib/getLsFilters.m:30 / getMagLsFilters.m:30 / getMagLsFilters2D.m:1 (hrir_zen NULL) / getEMagLsFilters.m:32 /
 * getEMagLs2Filters.m:32 / getEMagLsFiltersEMAinCH.m:32 / getEMagLsFiltersEMAinSH.m:32 (mic_zen NULL; EMAinSH plan by plan), kind =
 * EMAGLS_KIND_LS / _MAGLS / _MAGLS_2D / _EMAGLS / _EMAGLS2 / _EMA_CH / _EMA_SH.  hL, hR [nsamp x ndirs x nsets] (MATLAB 3-D arrays), wL, wR [len x channels x nsets] (LS: nsamp rows; `fs`
 * and `len` are ignored for LS).  Internally: plans and geometry-sharing batches of up to 16 sets (kept for the next call of the
 * same shape; emagls_cache_clear releases them), one resident sweep launch per batch; the same filters as nsets single calls.
 * eMagLS / eMagLS2 with 33..64 channels: the sets pass through two plans that keep their geometry stages between sets. */
int emagls_design_hrir_sets(int kind, const double* hL, const double* hR, int64_t nsamp, int64_t ndirs, int64_t nsets,
                            const double* hrir_azi, const double* hrir_zen, double mic_radius, const double* mic_azi, const double* mic_zen,
                            int64_t nmics, int order, double fs, int64_t len, int basis, void* wL, void* wR);
/* The HRTF subjects of ONE ATF set in one call (BASELINE config 5: the loop over subjects around
 * lib/getEMagLsFiltersFromAtf.m:1): hL, hR [nsamp x ndirs x nsets], the other arguments as emagls_get_emagls_filters_from_atf;
 * wL, wR [filter_len x nmics x nsets].  The ATF set is uploaded once and its side (spectra, matching, per-bin factors) computed
 * once per batch of up to 16 subjects; one resident sweep launch per batch.  The same filters as nsets single calls. */
int emagls_from_atf_hrir_sets(const double* hL, const double* hR, int64_t nsamp, int64_t ndirs, int64_t nsets, const double* hrir_azi,
                              const double* hrir_zen, const double* atf_irs, int64_t atf_taps, int64_t nmics, int64_t natf, const double* atf_azi,
                              const double* atf_zen, double fs, int64_t filter_len, double f_trans, double* wL, double* wR, double* mean_dev);
/* A batch may also hold EMAGLS_KIND_FROM_ATF plans of one shape -- the HRTF subjects of one ATF set (BASELINE config 5: 8 subjects).
 * lib/getEMagLsFiltersFromAtf.m:54-95,100-104: the spectra of the matched ATFs and their per-bin factors do not depend on the
 * HRIRs.  When all plans hold the same grids and the same ATF set (compared on the device whenever one of them was replaced) the
 * batch computes that side ONCE, on plan 0, and every subject's prologue, least-squares rows and sweep lane read it; the sweep is
 * one resident launch for all subjects.  Plans with different ATF sets, or ATFs whose low bins need the dense route, run
 * unshared (still one sweep launch).  *shared reports which after an execute. */
int emagls_batch_shares_atf_side(emagls_batch* batch, int* shared);
int emagls_batch_execute(emagls_batch* batch);
int emagls_batch_synchronize(emagls_batch* batch);
/* synchronise once, check every plan's device-side status flags, copy all filters out: wL[j], wR[j] receive the filters of
 * plan j (host or device pointers).  Equivalent to emagls_plan_get_filters on every plan, without the per-plan round trips. */
int emagls_batch_get_filters(emagls_batch* batch, void* const* wL, void* const* wR);
/* profiling: with level >= 1 HIP events bracket the sweep launch of every execute (on the batch stream);
 * emagls_batch_sweep_time returns the duration in ms of the last execute's sweep (synchronises the batch). */
int emagls_batch_set_profiling(emagls_batch* batch, int level);
/* *lanes = 1 when the batch runs in lane mode (one launch of every kernel for all its designs), 0 in stream mode.  Designs of
 * one shape class whose routes differ by a bin or an order (array radii of one simulation order) are given common routes. */
int emagls_batch_lane_mode(emagls_batch* batch, int* lanes);
/* Run the batch on the caller's hipStream_t instead of its own (the caller keeps ownership and must not use the stream while a
 * batch call is in progress).  Why one would: the HIP runtime multiplexes all streams of a process onto 4 hardware queues, and
 * two batches whose streams land on the same queue execute strictly one after the other; a caller that creates its streams
 * first and hands one to each batch in flight decides the mapping itself (bench.py does). */
int emagls_batch_set_stream(emagls_batch* batch, void* hip_stream);
/* A lane batch of more than 8 designs runs the stages before its sweep as two lane groups on two streams (then one sweep launch
 * for all designs).  The second group's stream comes from the library unless the caller hands one over here -- for the same
 * reason as emagls_batch_set_stream: which hardware queue a stream lands on is decided by the order the streams are created in. */
int emagls_batch_set_side_stream(emagls_batch* batch, void* hip_stream);
int emagls_batch_sweep_time(emagls_batch* batch, double* ms);
/* Lane mode: 1..4 HIP streams for the stages before the sweep (default 1).  With more than one the independent branches of the
 * design fork onto side streams and the captured graph carries the forks: [HRIR-grid SH matrix, Gram matrix, Cholesky factor,
 * Householder-route factors] | [array model, order terms, G_k of every bin] | [HRIR prologue, least-squares right-hand sides]
 * | [Gram-route factors].  Shortens the path to the batch's sweep from the sum of the kernels to its longest branch (what a
 * short run, a pipeline filling from empty, is bound by); with many batches in flight it only adds queue contention.  A batch with
 * forked stages is taken to have the device to itself: what its sweep does not need -- the Cholesky factor and the orthonormal
 * route of the ill-conditioned low bins, which only feed the filters' rows (lib/getEMagLsFilters.m:94) -- then runs NEXT to the sweep
 * on a stream of its own (designs on the synthesising sweep; EMAGLS_DEFER_HH=0 keeps everything before the sweep). */
int emagls_batch_set_streams(emagls_batch* batch, int nstreams);
/* Lane mode, one stream: the order in which a batch of up to 8 designs issues the stages before its sweep.  0 (default): the
 * order of a single design.  1: the kernels that fill the chip first (HRIR transform, SH Gram matrix, G_k of every bin), the
 * latency-bound chains (Cholesky, per-bin QR / Jacobi) after them.  2: the chains first.  Batches that run side by side in the
 * SAME order meet at the same kernels and add up their times; complementary orders hide one batch's chains behind the other's
 * bandwidth-bound kernels.  The two lane groups of a batch of more than 8 designs take orders 1 and 2 by themselves
 * (EMAGLS_STAGGER=0 switches that off).  Same filters in every order (no arithmetic changes, only the issue order). */
int emagls_batch_set_stage_order(emagls_batch* batch, int order);
int emagls_batch_destroy(emagls_batch* batch);

#ifdef __cplusplus
}
#endif
#endif /* EMAGLS_H */
