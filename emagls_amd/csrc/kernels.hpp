// Launcher declarations shared by the kernel translation units and the C ABI (capi.hip).
#pragma once
#include <algorithm>
#include <functional>

#include "common.hpp"

namespace emagls {

// ---- sh_basis.hip
void launch_zero(void* p, size_t bytes, hipStream_t st);
void launch_compare_words(const void* a, const void* b, size_t bytes, int* differ, hipStream_t st);
void launch_broadcast_lanes(const void* src, size_t bytes, size_t stride, int nlanes, hipStream_t st);
struct LanePtrs { void* p[64]; };   // [2 j + ear] for up to 32 designs
void launch_scatter_lanes(const void* srcL, const void* srcR, size_t stride, size_t bytes, int n, const LanePtrs& dst, hipStream_t st);
void launch_gather_buffers(const LanePtrs& src, const LanePtrs& dst, int nbuf, size_t bytes, hipStream_t st);   // nbuf <= 64 device buffers of `bytes` each
struct BufferMoves { const void* src[96]; void* dst[96]; size_t bytes[96]; int n; };   // device-to-device copies of one launch
void launch_move_buffers(const BufferMoves& m, hipStream_t st);
void launch_ch_basis(int N, int M, const double* azi, bool cplx_basis, void* out, int ld, hipStream_t st, bool out_real = false);
void launch_sh_coeff(int N, double* tab, hipStream_t st);
inline size_t esz(bool c) { return c ? sizeof(cplx) : sizeof(double); }   // element size of a real / complex basis
inline size_t sh_coeff_count(int N) { return (size_t)2 * (N + 1) * (N + 1) + 2 * (N + 1); }
void launch_sh_basis(int N, int64_t D, const double* azi, const double* zen, const double* tab, bool cplx_basis,
                     void* Y, int64_t ld, hipStream_t st);
void launch_transpose_conj(const void* Y, int64_t D, int64_t S, int64_t ldY, void* Yt, int64_t Dpad, int64_t ldYt,
                           bool is_cplx, bool conj_it, hipStream_t st);

// ---- modal.hip
void launch_modal_bn(int N, int64_t nfreq, const double* kr, double kr_scale, double out_scale, void* bn,
                     int64_t stride_k, int64_t stride_n, hipStream_t st, const int* n_valid = nullptr);

// ---- fft.hip
void launch_twiddles(int nfft, void* tw, hipStream_t st);
int hrir_dirsum_chunks(int64_t D);
void launch_hrir_grpdelay(const double* hL, const double* hR, int64_t L, int64_t D, int nfft, const void* tw,
                          double* partial, double* grpd, hipStream_t st);
// HcT (optional): direction-major real copy of the complex rows [D][ldT], HcT[d][2 (e n_c + kb) + re/im]
void launch_hrir_fft(const double* hL, const double* hR, int64_t L, int64_t D, const int64_t* didx, int nfft,
                     const void* tw, const double* grpd, int mode, int n_c, int kabs0, void* Hc, double* Habs,
                     int64_t ldD, hipStream_t st, double* HcT = nullptr, int ldT = 0);
void launch_real_fft_gather(const double* x, int64_t L, int64_t ncols, const int64_t* colidx, int nfft, const void* tw,
                            void* out, int64_t ldo, int64_t inner, int64_t ld_inner, hipStream_t st);
void launch_filter_epilogue(const void* W, int C, int nfft, int len, const void* tw, const double* grpd, int conj_mode,
                            int dc_rule, int shift_mode, int out_cplx, void* outL, void* outR, hipStream_t st, int n_ears = 2,
                            double fade_rel = 0.15);

// ---- gram_chol.hip
int gram_ksplit(int64_t D, int S);
int64_t gram_dpad(int64_t D, int S);
void launch_gram(const void* Yc, int64_t D, int S, int64_t ld, bool is_cplx, void* Gp, void* G, void* R, int Sh, hipStream_t st);
void launch_cholesky(void* G, int S, bool is_cplx, int* flag, hipStream_t st);
void launch_qform(const void* Yc, const void* R, void* Rinv, int S, int64_t D, int64_t ld, bool is_cplx, void* Q, hipStream_t st);
// zs R^H = z for the rows Z[kb][c][:] of the flagged bins (cond_ok[kb] == 0), in place (complex basis)
void launch_zsolve_flagged(void* Z, int ldS, const void* R, const void* Rinv, const double* cond_ok, int S, int C, int P, int k0,
                           hipStream_t st);
void launch_tn(const void* R, const void* E, int S, int C, int ldE, int nOrders, bool is_cplx, void* Tn, int64_t ldS,
               hipStream_t st);
void launch_small_gemm(const void* A, int lda, bool a_cplx, const void* B, int ldb, bool b_cplx, void* Cm, int ldc,
                       bool c_cplx, int M, int N, int K, hipStream_t st);

void launch_magls_m(const void* R, int C, bool is_cplx, int P, void* Mw, double* cond_ok, int* status, hipStream_t st);

// ---- factor.hip
struct FactorArgs;
void launch_factor(const FactorArgs& a, int nbins, bool tn_cplx, hipStream_t st, int phases = 3);
// Jacobi SVD only, on Gram matrices that gram_solve_kernel handed over (route[kb] == 1); bins with route[kb] == 2 are skipped
void launch_factor_jacobi_gram(const FactorArgs& a, int nbins, hipStream_t st);
void launch_factor_jacobi_pair(const FactorArgs& gram, int nb_gram, const FactorArgs& hh, int nb_hh, hipStream_t st);

// ---- gramroute.hip
void launch_gram_kmat(const void* Gy, const void* E, int S, int ldE, int C, int nOrd, bool is_cplx, void* F, int64_t ldF, double* Kmat, int ldK,
                      hipStream_t st);
void launch_gram_gemm(const void* bn, int nOrd, int P, int kb0, int nbins, double* Cf, int ldC, const double* Kmat, int ldK, int C, double* Apk,
                      int ldA, hipStream_t st);
void launch_gram_solve(const double* Apk, int ldA, int C, int kb0, int nbins, double reg_c, void* Mw, void* R2w, double* sv, int* route,
                       int* sweeps_out, hipStream_t st);
// out[r][s] = conj( sum_d H[r][d] conj(Yc[d][s]) ) for the 2 n_c complex rows r = e n_c + kb held direction-major in HcT
// (launch_hrir_fft); Yc [Dpad][ldY] real or complex, rows >= D zero.  Pw: workspace hy_mfma_workspace_doubles(...)
size_t hy_mfma_workspace_doubles(int n_c, int S, bool y_cplx);
int hy_mfma_kpad(int D);   // rows of both operands that the product reads (zero beyond D)
void launch_hy_conj_mfma(const double* HcT, int ldT, int n_c, const void* Yc, int64_t ldY, bool y_cplx, int D, int S, double* Pw, void* out,
                         int ldS, hipStream_t st);
void launch_ls_gram(const void* Hc, int64_t ldH, int n_c, const void* G, int64_t g_stride, int64_t ldD, const void* Mw, int D, int C, int P,
                    int kb_lo, int kb_hi, void* W, hipStream_t st);

// ---- sweep.hip
struct DenseSweepArgs;
struct HalfSweepArgs;
struct HalfSweepMulti;
void launch_sweep_dense(const DenseSweepArgs& a, int kb, bool x_cplx, hipStream_t st);
int dense_sweep_nwg(int D, int C);
void launch_sweep_half(const HalfSweepMulti& m, int kb, hipStream_t st);
void launch_sweep_half_finalize(const HalfSweepMulti& m, int kb_last, hipStream_t st);
// ---- sweep_persist.hip
int persist_sweep_nwg(int D);
bool persist_sweep_supported(int D, int C);
size_t persist_sweep_ll_bytes(int D, int C);
void launch_sweep_persist(const HalfSweepMulti& m, hipStream_t st);
int persist_sweep_dpw(int D);
// residency, decided before a launch: the runtime's occupancy of the kernel variant x the CUs of an XCD (EMAGLS_CU_BUDGET overrides
// the device's CU count) against the workgroups `ndesigns` designs place there
int sweep_cu_budget();
bool persist_sweep_fits(int D, int C, int ndesigns);
bool synth_sweep_fits(int D, int nmics, int nOrd, int ndesigns);
// ---- sweep_synth.hip: the resident sweep with the slab of every bin evaluated inside the launch (Legendre addition theorem)
bool synth_sweep_supported(int D, int nmics, int nOrd);
int synth_nord_pad(int nOrd);
// bsc [P][nord_pad] from bn [P][nOrd]; Pm [32][32] from the complex copy of pinv(Y_lo) (Zlo null: identity, raw microphones)
// smap [34]: the chain's row order of the microphones (row -> microphone), then the numbers of antipodal pairs and of single microphones
void launch_synth_prepare(const void* bn, int nOrd, int P, void* bsc, const void* Zlo, int ldZ, int nOut, int M, const int* smap, double* Pm, hipStream_t st);
// Mt[kb] = Pm^T M_kb Pm for kb in [k0, P) (slot kb - 1 like Mw) and Winit = W(k0-1,:) Pm
void launch_synth_mt(const void* Mw, const double* Pm, int nOut, int M, int k0, int P, const void* W, void* Mt, void* Winit, hipStream_t st);
// W[e][kb][:] = (U[e][kb][:] Pm^T) conj(M_kb) for kb in [k0, P)
void launch_synth_winit(const void* W, const void* Pm, int nOut, int M, int k0, int P, void* Winit, hipStream_t st, bool shared_geometry);
// (shared_geometry: Pm and Mw are ONE design's for every lane of the launch)
void launch_synth_rows(const void* U, int nchunks, const void* Pm, const void* Mw, int nOut, int M, int k_lo, int k_hi, int P, void* W, hipStream_t st,
                       bool shared_geometry = false);
// least-squares bins [kb_lo, kb_hi) of the Gram route: Upart[chunk][e][kb][row] = partial sums of H(kb,:) conj(g_kb) over a chunk of directions
// (synth_ls_chunks(D) chunks; launch_synth_rows with that many chunks turns them into W)
int synth_ls_chunks(int D);
void launch_synth_ls(const void* Hc, int64_t ldH, int n_c, const void* bsc, int nord_pad, const double* dir_azi, const double* dir_zen, const double* mic_azi,
                     const double* mic_zen, const int* smap, int D, int M, int P, int kb_lo, int kb_hi, void* Upart, hipStream_t st, bool shared_geometry = false);
void launch_sweep_synth(const HalfSweepMulti& m, hipStream_t st);
// ---- sweep_reg.hip: the synthesising sweep with the operand in registers (a lane = a direction; no slab in LDS).  Units = antipodal
// microphone pairs + single microphones (smap[32] + smap[33]); argument blocks in device memory (store_sweep_args)
int reg_sweep_max_units();
bool reg_sweep_supported(int D, int nmics, int nunits, int nOrd);
size_t reg_sweep_ll_bytes(int D, int nmics);
int reg_sweep_capacity(int D);                  // designs one launch can keep resident on this device
int reg_sweep_slots_per_xcd();                  // the sweep gate's capacity: thirds of a CU per XCD
int reg_sweep_pick_waves(int D, int ndesigns);  // waves per workgroup of a launch of `ndesigns` designs (4, 8 or 12; 0: cannot be resident)
int reg_sweep_gate_cost(int D, int ndesigns);   // what such a launch takes of the gate's capacity
bool reg_sweep_fits(int D, int nmics, int nunits, int nOrd, int ndesigns);
void launch_sweep_reg(const HalfSweepArgs* args_dev, const HalfSweepArgs& a0, int n, hipStream_t st);
double reg_reduce_selftest();
double gram_tile_selftest(bool four);   // gram_chol.hip: the LDS-staged Gram tile kernels against a host sum
void store_sweep_args(const HalfSweepArgs* host, int n, HalfSweepArgs* dev, hipStream_t st);
void launch_sweep_finalize(const void* Wpart, void* W, int nWG, int C, int P, int kb_last, hipStream_t st);
void launch_hy_conj(const void* Hc, int64_t ldD, int nrows, const void* Yc, int64_t ldY, bool y_cplx, int D, int S, void* Pw, void* out,
                    int ldS, hipStream_t st);
size_t hy_workspace_elems(int nrows, int ldS);
void launch_ypinv(const void* Q, int64_t ldQ, bool q_cplx, const void* Zb, int ldS, int D, int S, int C, void* Ypinv,
                  int64_t ldD, hipStream_t st);
void launch_ls_apply(const void* Hc, int64_t ldH, int n_c, const void* Zf, bool z_cplx, int64_t ldD, int D, int C, int P,
                     int kb_lo, int kb_hi, void* W, hipStream_t st);
void launch_ls_filters(const double* hL, const double* hR, int64_t L, int D, const void* Yp, bool cplx_basis, int64_t ldD,
                       int C, void* wL, void* wR, hipStream_t st);
void launch_conj_copy(const void* in, void* out, int64_t n, bool is_cplx, hipStream_t st);
// rows of real-SH coefficients -> complex-SH coefficients in place: W_c = W_r T_N (Y_c = Y_r T_N, sh_basis.hip conventions)
void launch_sh_rows_to_complex(void* W, int C, int nrows, int order, hipStream_t st);
void launch_widen(const void* in, int64_t ldi, bool in_cplx, void* out, int64_t ldo, int rows, int cols, bool transpose,
                  bool upper_only, hipStream_t st);

// ---- wide.hip: LS / MagLS above 32 channels (SH orders 5..7)
void launch_gram_inverse(const void* R, int S, bool is_cplx, void* M, int* status, hipStream_t st, void* work = nullptr);   // work (S > 64): S x S complex + 2 doubles
void launch_ypinv_gram(const void* Ycm, int64_t ldD, bool is_cplx, const void* M, int S, int D, void* Ypinv, hipStream_t st);
void launch_sweep_wide(const DenseSweepArgs& a, int kb, bool x_cplx, hipStream_t st);
void launch_sweep_wide_finalize(const void* Wpart, void* W, int nWG, int C, int P, int kb_last, hipStream_t st);

// ---- wide_array.hip: eMagLS / eMagLS2 with 33..64 channels
void launch_wa_assemble(const void* Tn, const void* bn, int nOrd, int S, int C, int ldS, int P, int kb0, int nbins, void* B, hipStream_t st);
void launch_wa_factor(void* B, void* Vw, int S, int C, int ldS, int nbins, double reg_c, double* tauw, void* R2w, void* Nw, double* sv, int* sweeps,
                      void* Z, hipStream_t st);
void launch_wa_yri(const void* Q, int64_t ldQ, const void* Z, int S, int C, int ldS, int D, int64_t ldD, int nbins, void* Yri, hipStream_t st);
void launch_wa_ls(const void* Hc, int64_t ldH, int n_c, const void* Yri, int64_t ldD, int D, int C, int P, int kb_first, int kb_end, void* W, hipStream_t st);
void launch_wa_lo_gram(const void* Ycm, int M, int nOut, double* Ag, hipStream_t st);
void launch_wa_e(const void* Ycm, int M, int nOut, int S, const void* Minv, void* E, int ldE, hipStream_t st);

// ---- dspace.hip
void launch_qt(const void* Yc, int64_t ldY, const void* E, int ldE, int D, int S, int C, int nOrders, bool is_cplx, void* QT,
               int64_t ldD, hipStream_t st);
// real_mode: complex basis evaluated on the real order terms (sh_order >= 0: channels are SH coefficients up to that order;
// < 0: channels are independent, e.g. microphones)
// k_end (< 0: P): bins [k0, k_end) only -- the designs whose sweep evaluates its operands itself (sweep_synth.hip) need G_k for their
// least-squares bins alone
void launch_dspace_g(const void* QT, int64_t ldD, bool is_cplx, const void* bn, int nOrders, int D, int C, int P, int k0, void* G,
                     hipStream_t st, int real_mode = 0, int sh_order = -1, int k_end = -1);
void launch_cond_flags(const double* sv, int C, int P, int hh_end, double* cond_ok, hipStream_t st);
// (Ycm: the column-major SH matrix [S][ldYcm] of a real basis -- the coalesced form; null: the row-major walk)
void launch_yri_accurate(const void* Q, int64_t ldQ, bool is_cplx, const void* Z, int ldS, const double* cond_ok, int D, int S,
                         int C, int P, int k0, void* Yri, int64_t ldD, hipStream_t st, const void* Ycm = nullptr, int64_t ldYcm = 0);

// ---- atf.hip
void launch_grid_match(const double* aziA, const double* zenA, int64_t nA, const double* aziB, const double* zenB,
                       int64_t nB, double* cartB, int64_t* idx, double* dev_deg, double* mean_dev, hipStream_t st);
void launch_atf_colidx(const int64_t* idx, int64_t nA, int M, int64_t* colidx, hipStream_t st);

// ---- microbench.hip
double measure_fp64_peak(int which, int reps, bool burst = false, double* mhz = nullptr);

// ---- decode.hip
void binaural_decode_real(const double* sig, int64_t n, int C, const double* wL, const double* wR, int64_t len,
                          double* out /* [n x 2] column-major */, hipStream_t st,
                          const cplx* sigc = nullptr /* the C / 2 complex channels whose planes are `sig`'s channels, interleaved (wave form only) */);
bool decode_wave_form(int64_t len);
void binaural_decode_complex(const void* sig, bool sig_cplx, int64_t n, int C, const void* wL, const void* wR, bool w_cplx, int64_t len,
                             double* sig2, double* w2L, double* w2R, double* out, double* imag_abs, double* d_tmp, hipStream_t st,
                             int64_t imag_skip = 0 /* samples left out of the imaginary-part sums (the compensateDelay cut) */);
void decode_cache_clear();
void filter_channels_by_order(const double* sig, int64_t n_in, int64_t n, int C, const double* ir /* [nOrd][len] */, int nOrd, int64_t len,
                              int64_t skip, double* out /* [C][n-skip] */, hipStream_t st);

// ---- emash.hip (getEMagLsFiltersEMAinSH)
void launch_ema_sh_e0(const void* Ech, int ldS, const void* Ypts, int C, int S, bool cb, void* E0, hipStream_t st);
void launch_rot_points(const double* azi, const double* zen, int D, int npts, double* azr, double* znr, hipStream_t st);
void launch_rot_from_points(const void* A, int64_t ldA, const void* Z, int ldP, int C, int npts, const double* zen, int D, bool cb, void* Rot,
                            hipStream_t st);
void launch_qt_rotate(void* QT, int64_t ldD, int nOrd, int C, int N, int D, const void* Rot, bool cb, hipStream_t st);
void launch_gram_from_g(const void* G, int64_t g_stride, int64_t ldD, int D, int C, int kb0, int nbins, int g0, double* Apk, int ldK,
                        hipStream_t st);

// ---- render.hip
void launch_radial_filter(const void* bn, int nOrd, int P, int type, double regul, double g, bool nyq_abs, bool zero_nan,
                          void* out_kn, void* out_cm, hipStream_t st);
void launch_diffuse_field(const void* bn, int nOrd, int n_lo, int P, double* df_hi, double* df_lo, hipStream_t st);
void launch_array_diffuse(const void* bn, int nOrd, const void* Y, bool y_cplx, int ldY, int S, int M, int nOut, int P,
                          double* df_lo, hipStream_t st);
void launch_eq_spectrum(const double* df_hi, const double* df_lo, const double* df_arr, int P, int mode, void* W, double* W_full,
                        hipStream_t st);
void launch_diffuse_constraint(void* W, const void* G, bool g_cplx, int64_t g_stride, int g0, const void* H, int D, int C, int64_t ldD,
                               int P, hipStream_t st);
void launch_smair(const void* E, bool e_cplx, int ldS, const void* bn, int nOrd, const void* rad, int nRad, int rows, int S, int P, void* out,
                  hipStream_t st);
void launch_sh_encode(const double* sig, int64_t n, int M, const void* Z, int ldZ, int nOut, bool out_cplx, void* out, hipStream_t st);

// ---- capi.hip: process-wide stream pool (streams are recycled, never destroyed: see StreamPool)
hipStream_t pool_stream_take();
void pool_stream_give(hipStream_t st);
// ---- capi.hip: runs f, maps exceptions to the C status codes and records the message for emagls_last_error()
int guarded_call(const std::function<void()>& f);

}  // namespace emagls

// full definitions of the argument structs (kept in their kernel files' header section)
#include "args.hpp"
