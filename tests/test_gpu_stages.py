"""GPU parity, stage by stage: every intermediate of the HIP pipeline against NumPy on the same
inputs (debug buffers of a Plan), then the kernel-level C-ABI entry points against the oracle."""
import numpy as np
import pytest

from oracle import emagls_oracle as O

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


@pytest.mark.parametrize("basis", ["real", "complex"])
@pytest.mark.parametrize("N", [4, 19, 44])
def test_sh_basis_vs_oracle(grids, basis, N):
    import emagls_amd as E
    d = np.column_stack([grids["azi"], grids["zen"]])
    Y = E.getSH(N, d, basis)
    Yo = O.getSH(N, d, basis)
    assert Y.shape == Yo.shape and Y.dtype == Yo.dtype
    err = np.abs(Y - Yo)
    # at the poles sqrt(1-x^2) amplifies the last-ulp difference of cos(zen): ~1e-10 absolute there
    polar = np.abs(np.sin(grids["zen"])) < 1e-6
    assert err[~polar].max() < 2e-13, err[~polar].max()
    assert err.max() < 1e-9


def test_sh_basis_mic_grid_and_empty(grids):
    import emagls_amd as E
    d = np.column_stack([grids["mic_azi"], grids["mic_zen"]])
    assert rel(E.getSH(19, d, "complex"), O.getSH(19, d, "complex")) < 1e-13
    assert E.getSH(3, np.zeros((0, 2)), "real").shape == (0, 16)


def test_modal_bn_vs_oracle():
    import emagls_amd as E
    for r, N, P in ((0.042, 19, 513), (0.10, 44, 1025), (0.02, 9, 257)):
        kr = 2 * np.pi * np.linspace(0, 24000, P) / 343.0 * r
        b = E.sphModalCoeffs(N, kr)
        bo = O.sphModalCoeffs(N, kr)
        assert b.shape == bo.shape
        assert np.all(b[0] == bo[0])
        e = np.abs(b[1:] - bo[1:]) / np.abs(bo[1:])
        assert e.max() < 1e-12, (r, N, e.max())


@pytest.fixture(scope="module")
def emagls_plan(grids, hrirs):
    """config-3 shaped plan (em32, N=4, complex SH) on a thinned grid and 128-tap filters so the
    NumPy cross-checks stay fast; keeps all intermediates on the device."""
    import os
    from emagls_amd import Plan, _lib as L
    sub = slice(0, 2702, 3)
    hL, hR = hrirs[0][:, sub], hrirs[1][:, sub]
    azi, zen = grids["azi"][sub], grids["zen"][sub]
    # complex-basis designs are normally served by the real-arithmetic pipeline plus a channel transform; the stage checks
    # below look at the complex pipeline itself (still the path of LS / MagLS / EMAinCH in the complex basis)
    os.environ["EMAGLS_REAL_INTERNAL"] = "0"
    try:
        p = Plan(L.KIND_EMAGLS, "complex", 4, 48000.0, 128, hL.shape[0], hL.shape[1], grids["mic_radius"], 32)
    finally:
        del os.environ["EMAGLS_REAL_INTERNAL"]
    p.set_hrir_grid(azi, zen)
    p.set_mic_grid(grids["mic_azi"], grids["mic_zen"])
    p.set_hrirs(hL, hR)
    p.set_profiling(1)
    p.execute()
    p.synchronize()
    yield dict(p=p, hL=hL, hR=hR, azi=azi, zen=zen)
    p.close()


def test_stage_basis_gram_cholesky_q(emagls_plan, grids):
    p = emagls_plan["p"]
    i = p.info()
    S, D = i.num_sh_sim, emagls_plan["hL"].shape[1]
    assert (i.sim_order, S, i.num_channels, i.nfft, i.num_pos_freqs) == (19, 400, 25, 256, 129)
    ldD, ldS = -(-D // 64) * 64, -(-S // 64) * 64
    Yo = O.getSH(19, np.column_stack([emagls_plan["azi"], emagls_plan["zen"]]), "complex")
    Ycm = p.debug("Ycm", np.complex128, (S, ldD))[:, :D]
    assert np.abs(Ycm.T - Yo).max() < 1e-9
    Yc = p.debug("Yc", np.complex128).reshape(-1, ldS)
    assert np.abs(Yc[:D, :S] - np.conj(Yo)).max() < 1e-9
    assert np.all(Yc[D:, :S] == 0)
    # the Cholesky factor covers the orders the Householder-route bins need (hh_orders: the higher ones are below the
    # noise floor of those low bins): the leading Sh x Sh block of the Gram matrix
    Sh = i.hh_orders ** 2
    ldSh = -(-Sh // 64) * 64
    assert 25 <= Sh <= S
    R = np.triu(p.debug("R", np.complex128, (Sh, Sh)))
    G = Yc[:D, :Sh].conj().T @ Yc[:D, :Sh]
    assert rel(R.conj().T @ R, G) < 1e-13
    # Q = conj(Y) R^-1 is not materialised for complex-basis eMagLS designs (only H conj(Q) and, for ill-conditioned
    # bins, Z_k R^-H are formed): the Cholesky factor must make it orthonormal
    Q = np.linalg.solve(R.T, Yc[:D, :Sh].T).T
    assert np.abs(Q.conj().T @ Q - np.eye(Sh)).max() < 1e-13
    Rinv = p.debug("Rinv", np.complex128).reshape(-1, 32, 32)
    for J in range(Sh // 32):
        assert rel(Rinv[J] @ R[32 * J:32 * J + 32, 32 * J:32 * J + 32], np.eye(32)) < 1e-13
    # least-squares rows: Hq holds conj(H conj(Q)) for the bins below k_cut
    kcut0 = i.k_cut - 1
    Hc = p.debug("Hc", np.complex128).reshape(2, kcut0, ldD)[:, :, :D]
    Hq = p.debug("Hq", np.complex128).reshape(2, kcut0, ldSh)[:, :, :Sh]
    for e in range(2):
        assert rel(np.conj(Hq[e, 1:]), Hc[e, 1:] @ np.conj(Q)) < 1e-12


def test_stage_array_model(emagls_plan, grids):
    p = emagls_plan["p"]
    S, C, P = 400, 25, 129
    ldS = 448
    micd = np.column_stack([grids["mic_azi"], grids["mic_zen"]])
    Ym = O.getSH(19, micd, "complex")
    Eo = O.pinv(Ym[:, :25]) @ Ym
    E = p.debug("E", np.complex128, (C, ldS))[:, :S]
    assert rel(E, Eo) < 1e-12
    f = np.linspace(0, 24000, P)
    bo = -O.sphModalCoeffs(19, 2 * np.pi * f / 343.0 * grids["mic_radius"])
    b = p.debug("bn", np.complex128, (P, 20))
    assert (np.abs(b[1:] - bo[1:]) / np.abs(bo[1:])).max() < 1e-12
    nh = p.info().hh_orders
    Sh = nh * nh
    ldSh = -(-Sh // 64) * 64
    R = np.triu(p.debug("R", np.complex128, (Sh, Sh)))
    Tn = p.debug("Tn", np.complex128)[:nh * C * ldSh].reshape(nh, C, ldSh)[:, :, :Sh]
    for n in (0, 3, nh - 1):
        blk = slice(n * n, (n + 1) ** 2)
        To = (R[:, blk] @ Eo[:, blk].T).T  # [c][s]
        assert rel(Tn[n], To) < 1e-12


def test_stage_prologue(emagls_plan):
    p = emagls_plan["p"]
    hL, hR = emagls_plan["hL"], emagls_plan["hR"]
    D = hL.shape[1]
    ldD = -(-D // 64) * 64
    i = p.info()
    nfft, P, kcut0 = i.nfft, i.num_pos_freqs, i.k_cut - 1
    HL, HR, gL, gR = O._hrir_prologue(hL, hR, nfft, P)
    assert abs(i.grp_delay_l - gL) < 1e-9 and abs(i.grp_delay_r - gR) < 1e-9
    Hc = p.debug("Hc", np.complex128).reshape(2, kcut0, ldD)[:, :, :D]
    assert rel(Hc[0], HL[:kcut0]) < 1e-12 and rel(Hc[1], HR[:kcut0]) < 1e-12
    Ha = p.debug("Habs", np.float64).reshape(2, P - kcut0, ldD)[:, :, :D]
    assert rel(Ha[0], np.abs(HL[kcut0:P])) < 1e-12 and rel(Ha[1], np.abs(HR[kcut0:P])) < 1e-12


@pytest.mark.parametrize("nsamp,ndirs", [(512, 157), (300, 64), (37, 9)])
def test_stage_prologue_1024_wave_form(grids, hrirs, monkeypatch, nsamp, ndirs):
    """nfft = 1024 (512-tap filters, the headline configuration) takes the wave-private transforms of wave_fft.hpp
    (hrir_fft_wave_kernel); EMAGLS_HRIR_FFT_WAVE=0 keeps the LDS radix-2^2 form.  Both against the oracle's prologue
    (lib/getEMagLsFilters.m:72-81), direction counts that are not a multiple of the 8 directions per workgroup included."""
    from emagls_amd import Plan, _lib as L
    rng = np.random.default_rng(5)
    sel = np.sort(rng.choice(2702, ndirs, replace=False))
    hL, hR = hrirs[0][:nsamp, sel], hrirs[1][:nsamp, sel]
    if nsamp > hrirs[0].shape[0]:
        extra = rng.standard_normal((nsamp - hrirs[0].shape[0], ndirs)) * 1e-3
        hL, hR = np.vstack([hL, extra]), np.vstack([hR, -extra])
    hL, hR = np.ascontiguousarray(hL), np.ascontiguousarray(hR)
    azi, zen = grids["azi"][sel], grids["zen"][sel]
    got = {}
    for form in ("1", "0"):
        monkeypatch.setenv("EMAGLS_HRIR_FFT_WAVE", form)
        p = Plan(L.KIND_MAGLS, "real", 2, 48000.0, 512, hL.shape[0], ndirs)
        p.set_hrir_grid(azi, zen)
        p.set_hrirs(hL, hR)
        p.set_profiling(1)
        p.execute()
        p.synchronize()
        i = p.info()
        nfft, P, kcut0 = i.nfft, i.num_pos_freqs, i.k_cut - 1
        assert (nfft, P) == (1024, 513)
        ldD = -(-ndirs // 64) * 64
        Hc = p.debug("Hc", np.complex128).reshape(2, kcut0, ldD)[:, :, :ndirs].copy()
        Ha = p.debug("Habs", np.float64).reshape(2, P - kcut0, ldD)[:, :, :ndirs].copy()
        got[form] = (Hc, Ha, i.grp_delay_l, i.grp_delay_r)
        p.close()
    HL, HR, gL, gR = O._hrir_prologue(hL, hR, 1024, 513)
    for form, (Hc, Ha, dl, dr) in got.items():
        assert abs(dl - gL) < 1e-9 and abs(dr - gR) < 1e-9
        e = max(rel(Hc[0], HL[:kcut0]), rel(Hc[1], HR[:kcut0]), rel(Ha[0], np.abs(HL[kcut0:513])), rel(Ha[1], np.abs(HR[kcut0:513])))
        print(f"HRIR prologue {nsamp} taps x {ndirs} directions, nfft 1024, wave form {form}: rel vs oracle = {e:.3e}")
        assert e < 1e-10    # (the complex rows carry the delay phase: pi x the difference of the two medians, itself held to 1e-9 above)
    assert rel(got["1"][1], got["0"][1]) < 1e-13 and rel(got["1"][0], got["0"][0]) < 1e-13


def test_stage_factor_and_sweep(emagls_plan, grids):
    """Per-bin factors against LAPACK on the SAME B_k: singular values, Jacobi sweep counts, the
    S-space inverse Z_k of the least-squares bins, the direction-space operands G_k / Yri_k of the swept
    bins, and the final filters against the oracle."""
    p = emagls_plan["p"]
    S, C, ldS = 400, 25, 448
    D = emagls_plan["hL"].shape[1]
    ldD = -(-D // 64) * 64
    i = p.info()
    P, kcut0 = i.num_pos_freqs, i.k_cut - 1
    # routes: bins [1, hh_end) Householder QR + Jacobi in S space on the orders above the noise floor of those bins
    # (hh_orders: R, T_n, Z cover Sh = hh_orders^2 rows), bins [gram_from, P) through the Gram matrices
    assert 5 <= i.hh_orders <= 20 and 1 < i.gram_from == i.hh_end <= kcut0 + 1 and i.g_first == min(i.gram_from, kcut0)
    hh_end, g0, nh = i.hh_end, i.g_first, i.hh_orders
    Sh = nh * nh
    ldSh = -(-Sh // 64) * 64
    Tn = p.debug("Tn", np.complex128)[:nh * C * ldSh].reshape(nh, C, ldSh)[:, :, :Sh]
    bn = p.debug("bn", np.complex128, (P, 20))
    Yc = p.debug("Yc", np.complex128).reshape(-1, ldS)[:D, :S]
    E = p.debug("E", np.complex128, (C, ldS))[:, :S]
    Qh = np.linalg.solve(np.triu(p.debug("R", np.complex128, (Sh, Sh))).T, Yc[:, :Sh].T).T  # (not materialised on the GPU)
    Z = p.debug("Z", np.complex128)[:hh_end * C * ldSh].reshape(hh_end, C, ldSh)[:, :, :Sh]
    sv = p.debug("sv", np.float64).reshape(P, C)
    js = p.debug("jsweeps", np.int32)
    route = p.debug("route", np.int32)   # 0 Householder + Jacobi, 1 Gram + Jacobi, 2 Gram + Cholesky inverse (no SVD)
    assert js[1:P].max() <= 20 and np.all((js[1:P] >= 1) | (route[1:P] == 2)), (js[1:P].min(), js[1:P].max())
    assert (route[kcut0:P] == 2).sum() > 0.5 * (P - kcut0)
    nsw = P - g0  # bins with a direction-space operand; the buffers carry padding for the persistent sweep's whole-row-group loads
    G = p.debug("G", np.complex128)[:nsw * C * ldD].reshape(nsw, C, ldD)[:, :, :D]
    Mw = p.debug("Mw", np.complex128)[:P * C * C].reshape(P, C, C)  # bin kb is stored at slot kb-1

    rep = np.repeat(np.arange(20), 2 * np.arange(20) + 1)

    def Xk(kb):
        """pwGrid_k.' = conj(Y) diag(b_n) E^T, D x C, all orders"""
        b = bn[kb].copy()
        if kb == P - 1:
            b = b.real
        return Yc @ (b[rep][:, None] * E.T)

    def Bk(kb):
        """the S-space matrix of a Householder-route bin: sum_n b_n T_n over the orders kept, Sh x C"""
        return np.tensordot(bn[kb][:nh], Tn, axes=(0, 0)).T

    for kb in sorted({1, 2, hh_end - 1, hh_end, kcut0 - 1, kcut0, kcut0 + 1, P // 2, P - 1}):
        X = Xk(kb)
        U, s, Vh = np.linalg.svd(X, full_matrices=False)
        if kb < hh_end:   # the truncated S-space matrix has the singular values of the full pwGrid_k to rounding
            B = Bk(kb)
            assert np.abs(np.linalg.svd(B, compute_uv=False) - s).max() < 1e-13 * s[0]
        AtA = X.conj().T @ X
        if route[kb] == 2:
            # direct route (no singular value is clipped: M = (B^H B)^-1): sv holds certified bounds, not the values
            assert sv[kb].max() >= s[0] * (1 - 1e-12) and sv[kb].min() <= s[-1] * (1 + 1e-12)
            assert sv[kb].max() <= 100 * sv[kb].min()
            assert rel(Mw[kb - 1], np.linalg.inv(AtA)) < 1e-10
        else:
            # (well-conditioned swept bins take the Gram route: error eps cond^2 instead of eps)
            assert np.abs(np.sort(sv[kb])[::-1] - s).max() < (1e-13 if kb < hh_end else 1e-9) * s[0]
        sreg = 1 / np.maximum(s, 0.01 * s[0])
        Yri_o = np.conj(U) @ (sreg[:, None] * Vh.conj())       # Y_reg_inv_k, D x C  (lib/getEMagLsFilters.m:90)
        if kb < min(kcut0, hh_end):  # Z_k is only formed for the Householder-route least-squares bins (and for ill-conditioned swept bins)
            assert rel(np.conj(Qh) @ Z[kb].T, Yri_o) < 1e-8, (kb, rel(np.conj(Qh) @ Z[kb].T, Yri_o))
        if kb >= g0:
            assert rel(G[kb - g0].T, X) < 1e-12
            # Y_reg_inv_k = conj(G_k) conj(M_k): the sweep applies conj(M_k) after the cross-workgroup sum
            Yri = np.conj(G[kb - g0].T) @ np.conj(Mw[kb - 1])
            assert rel(Yri, Yri_o) < 1e-8, (kb, rel(Yri, Yri_o))
    wL, wR = p.get_filters()
    oL, oR = O.getEMagLsFilters(emagls_plan["hL"], emagls_plan["hR"], emagls_plan["azi"], emagls_plan["zen"],
                                grids["mic_radius"], grids["mic_azi"], grids["mic_zen"], 4, 48000.0, 128, "complex")
    assert rel(wL, oL) < 1e-6 and rel(wR, oR) < 1e-6, (rel(wL, oL), rel(wR, oR))
    print("stage times (ms):", p.stage_times())


def test_wave_reduction_of_the_register_resident_sweep():
    """sweep_reg.hip sums a wave's 32 directions per lane parity with v_permlane32_swap / v_permlane16_swap halving steps, a DPP
    row rotation and two all-reduce steps inside the eight-lane groups: against a plain sum on the host."""
    import ctypes
    from emagls_amd import _lib as L
    err = ctypes.c_double(-1.0)
    L.check(L.load().emagls_self_test(0, ctypes.byref(err)))
    print(f"wave reduction self test: max abs error = {err.value:.3e}")
    assert 0.0 <= err.value < 1e-13


def test_gram_tile_kernels_against_a_host_sum():
    """gram_chol.hip's LDS-staged Gram tiles -- on v_mfma_f64_16x16x4 (what lane batches run) and on the four-block
    v_mfma_f64_4x4x4_4b shape (EMAGLS_GRAM_MFMA4=1; operand layout found by experiment) -- on a 333 x 100 pseudo-random matrix
    (ragged: partial tiles, rows that are not a multiple of the stage) against a plain host sum."""
    import ctypes
    from emagls_amd import _lib as L
    for which, name in ((1, "16 x 16 x 4"), (2, "4 x 4 x 4, four blocks")):
        err = ctypes.c_double(-1.0)
        L.check(L.load().emagls_self_test(which, ctypes.byref(err)))
        print(f"Gram tile on the {name} shape: max error relative to the largest element = {err.value:.3e}")
        assert 0.0 <= err.value < 1e-13
