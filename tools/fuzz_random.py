"""Random shape campaign on the GPU box: `n` seeded cases over every design entry point -- odd direction counts, filter lengths
whose FFT length is not a power of two, SH orders up to 7, arrays of up to 64 microphones, sampling rates 16-96 kHz -- GPU
result against the oracle (dev tooling; the fixed cases of tests/shape_cases.py are what the test suite runs).

    python tools/fuzz_random.py [n] [seed]

A case the library refuses with EMAGLS_ERR_UNSUPPORTED / _ARG counts as 'refused' (its message is printed), not as a failure.
A case above 1e-6 is re-examined, and never counted as ok:
  * the oracle is run twice more with the two LAPACK SVD drivers (gesdd / gesvd); where those two disagree at the same level (a
    bin's smallest singular values at eps * s_max: the clipped subspace's singular vectors are rounding noise, so is the reference's
    own result) the case counts as 'ill_posed';
  * otherwise, for eMagLS2 / real-basis eMagLS designs whose lowest solved bins are least-squares bins, the rows W(k,:) of those bins
    are carried through 40-digit arithmetic on the oracle's own FP64 inputs (tools/exact_rows.py, a minute per bin) and both the
    GPU's rows (plan buffer "W") and the oracle's FP64 rows are compared with them: where the GPU's rows are exact to 1e-9 and the
    oracle's are not, the distance is the REFERENCE ARITHMETIC's own rounding error (LAPACK's SVD of the rounded pwGrid) and the case
    counts as 'reference_noise', with both distances printed (round 6; round 5 excused such cases by a factor-100 rule on a
    perturbation probe -- removed);
  * otherwise, for array designs, the singular values of the reference's FP64 pwGrid in its lowest bins are counted: where some lie below
    1e-14 of the largest (the noise floor of the FP64 product is 3e-17), the reference weights singular vectors that its own rounding errors
    decide with 100 / s_max -- such a case counts as 'ill_posed' as well, with the count printed (round 6, after seed 81's case 83: 127 x the
    drivers' disagreement, 12 of 25 singular values at the noise floor; profiles/r06_fuzz_random.md);
  * anything else is a MISMATCH."""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402


def draw(rng):
    kind = str(rng.choice(["emagls", "emagls2", "magls", "ls", "emainch", "atf", "emainsh", "magls2d", "decode", "geo"],
                          p=[0.17, 0.13, 0.13, 0.05, 0.1, 0.1, 0.07, 0.05, 0.1, 0.1]))
    if kind == "geo":        # HRIR sets on one geometry: (sets, directions, taps, len, fs, radius, mics, order, basis, design kind)
        sub = str(rng.choice(["emagls", "emagls2", "emainch"]))
        N = int(rng.integers(0, 5)) if sub != "emainch" else int(rng.integers(0, 9))
        M = int(rng.integers(max(2, (N + 1) ** 2), 33)) if sub == "emagls" else (int(rng.integers(2 * N + 1, 33)) if sub == "emainch" else int(rng.integers(4, 33)))
        taps = int(rng.choice([16, 33, 64, 100]))
        return (kind, int(rng.integers(2, 8)), int(rng.integers(300, 1500)), taps, int(2 * rng.integers(max(4, taps // 2), 130)),
                float(rng.choice([16000.0, 32000.0, 44100.0, 48000.0])), float(rng.uniform(0.01, 0.07)), M, N, str(rng.choice(["real", "complex"])), sub)
    if kind == "decode":     # binauralDecode: (samples, channels, taps, complex signal / filters, compensateDelay)
        return (kind, int(rng.integers(1, 30000)), int(rng.integers(1, 50)), int(rng.integers(1, 1600)) * (2 if rng.random() < 0.5 else 1),
                bool(rng.random() < 0.3), bool(rng.random() < 0.3), bool(rng.random() < 0.5))
    fs = float(rng.choice([16000.0, 32000.0, 44100.0, 48000.0, 96000.0]))
    D = int(rng.integers(60, 1600))
    taps = int(rng.choice([16, 33, 64, 100, 128, 200]))
    ln = int(2 * rng.integers(max(4, taps // 2), 200))          # even, >= taps (the reference asserts len >= HRIR length)
    if os.environ.get("EMAGLS_FUZZ_ROUND4"):   # round 4's kernels: nfft = 1024 (wave-private HRIR prologue) and grids above 3072 directions
        u = rng.random()
        if u < 0.35:
            ln, taps = 512, int(rng.choice([64, 200, 256, 400, 512]))
        if 0.25 < u < 0.45 and kind in ("emagls", "emagls2", "magls", "ls", "emainch", "atf"):
            D = int(rng.integers(3100, 7000))
    basis = str(rng.choice(["real", "complex"]))
    if kind in ("emagls", "emagls2"):
        N = int(rng.integers(0, 8))
        M = int(rng.integers(max(2, (N + 1) ** 2 // 2), 65)) if kind == "emagls" else int(rng.integers(2, 65))
        r = float(rng.uniform(0.005, 0.058))
        return (kind, D, taps, ln, fs, r, M, N, basis)
    wide6 = bool(os.environ.get("EMAGLS_FUZZ_ROUND6B"))   # (second half of round 6: LS / MagLS orders up to 15, MagLS-2D up to 110)
    if kind in ("magls", "ls"):
        return (kind, D, taps, ln, fs, 0, 0, int(rng.integers(0, 16 if wide6 else 8)), basis)
    if kind == "emainch":
        N = int(rng.integers(0, 13))
        return (kind, D, taps, ln, fs, float(rng.uniform(0.01, 0.08)), int(rng.integers(2 * N + 1, 33)), N, basis)
    if kind == "emainsh":
        N = int(rng.integers(1, 8)) if (os.environ.get("EMAGLS_FUZZ_ROUND6") or wide6) else int(rng.integers(1, 5))   # (orders 5..7 since round 6)
        return (kind, D, taps, ln, fs, float(rng.uniform(0.02, 0.08)), int(rng.integers(2 * N + 1, 33)), N, basis)
    if kind == "magls2d":
        return (kind, int(rng.integers(40, 720)), taps, ln, fs, 0, 0, int(rng.integers(0, 110 if wide6 else 20)), basis)
    return ("atf", D, taps, ln, 48000.0, int(rng.integers(30, 3000)), int(rng.integers(1, 17)), int(rng.choice([16, 50, 64, 128])),
            float(rng.choice([500.0, 1500.0, 2000.0, 3000.0])))


def run(case):
    import emagls_amd as E
    from emagls_amd import synth
    from oracle import emagls_oracle as O
    import shape_cases as SC
    kind = case[0]
    if kind == "geo":
        from emagls_amd import synth
        from emagls_amd.batch import emagls_hrir_sets
        _, nset, D, taps, ln, fs, r, M, N, basis, sub = case
        azi, zen = synth.fibonacci_grid(D)
        if sub == "emainch":
            ma, mz = np.linspace(0, 2 * np.pi, M, endpoint=False) + 0.2, None
        else:
            ma, mz = SC.mics(M, D + M)
        subjects = [synth.rigid_sphere_hrirs(azi, zen, fs=fs, taps=taps, centre_delay=taps / 4, seed=7 + j, head_radius=0.08 + 0.003 * j)
                    for j in range(nset)]
        res = emagls_hrir_sets(subjects, azi, zen, r, ma, mz, N, fs, ln, basis, kind=sub, max_batch=8)
        fn = {"emagls": E.getEMagLsFilters, "emagls2": E.getEMagLs2Filters}.get(sub)
        worst = 0.0
        for j in (0, nset - 1):     # the shared batch against the single designs (the oracle comparison is the other kinds' job)
            w = fn(subjects[j][0], subjects[j][1], azi, zen, r, ma, mz, N, fs, ln, basis) if fn else \
                E.getEMagLsFiltersEMAinCH(subjects[j][0], subjects[j][1], azi, zen, r, ma, N, fs, ln, basis)
            worst = max(worst, SC.rel(res[j][0], w[0]), SC.rel(res[j][1], w[1]))
        return worst
    if kind == "decode":
        import warnings
        _, nsamp, nch, length, sig_c, w_c, comp = case
        rng = np.random.default_rng(nsamp + 31 * nch + length)
        cx = lambda shape, on: rng.standard_normal(shape) + (1j * rng.standard_normal(shape) if on else 0.0)
        env = np.exp(-np.arange(length) / (0.3 * length + 1.0))[:, None]
        sig, wL, wR = cx((nsamp, nch), sig_c), cx((length, nch), w_c) * env, cx((length, nch), w_c) * env
        if comp and (nsamp <= length // 2 or length < 2):
            comp = False
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = E.binauralDecode(sig, 48000, wL, wR, 48000, comp)
            ref = O.binauralDecode(sig, wL, wR, comp)
        assert out.shape == ref.shape
        return SC.rel(out, ref)
    if kind not in ("emainsh", "magls2d"):
        return SC.run(case)
    _, D, taps, ln, fs, r, M, N, basis = case
    if kind == "magls2d":
        azi = np.sort(np.mod(np.linspace(0, 2 * np.pi, D, endpoint=False) + 0.003 * np.random.default_rng(D).standard_normal(D), 2 * np.pi))
        hL, hR = synth.rigid_sphere_hrirs(azi, np.full(D, np.pi / 2), fs=fs, taps=taps, centre_delay=taps / 4)
        w, o = E.getMagLsFilters2D(hL, hR, azi, N, fs, ln, basis), O.getMagLsFilters2D(hL, hR, azi, N, fs, ln, basis)
    else:
        azi, zen = synth.fibonacci_grid(D)
        hL, hR = synth.rigid_sphere_hrirs(azi, zen, fs=fs, taps=taps, centre_delay=taps / 4)
        ma = np.linspace(0, 2 * np.pi, M, endpoint=False) + 0.2
        w = E.getEMagLsFiltersEMAinSH(hL, hR, azi, zen, r, ma, N, fs, ln, basis)
        o = O.getEMagLsFiltersEMAinSH(hL, hR, azi, zen, r, ma, N, fs, ln, basis)
    assert w[0].shape == o[0].shape and w[0].dtype == o[0].dtype
    return max(SC.rel(w[0], o[0]), SC.rel(w[1], o[1]))


def oracle_filters(case, driver):
    """The oracle's filters for a case with the given LAPACK SVD driver."""
    import scipy.linalg as sl
    from emagls_amd import synth
    from oracle import emagls_oracle as O
    import shape_cases as SC
    orig = np.linalg.svd
    np.linalg.svd = lambda a, full_matrices=False: sl.svd(a, full_matrices=full_matrices, lapack_driver=driver)
    try:
        kind = case[0]
        if kind == "atf":
            _, D, taps, ln, fs, natf, M, ataps, ft = case
            azi, zen = synth.fibonacci_grid(D)
            hL, hR = synth.rigid_sphere_hrirs(azi, zen, fs=fs, taps=taps, centre_delay=taps / 4)
            atf, aazi, azen = synth.glasses_atfs(natf=natf, nmics=M, taps=ataps, fs=fs)
            o = O.getEMagLsFiltersFromAtf(hL, hR, np.column_stack([azi, zen]), atf, np.column_stack([aazi + 0.01, azen]), fs, ln, ft)
            return o[0], o[1]
        _, D, taps, ln, fs, r, M, N, basis = case
        if kind == "magls2d":
            azi = np.sort(np.mod(np.linspace(0, 2 * np.pi, D, endpoint=False) + 0.003 * np.random.default_rng(D).standard_normal(D), 2 * np.pi))
            hL, hR = synth.rigid_sphere_hrirs(azi, np.full(D, np.pi / 2), fs=fs, taps=taps, centre_delay=taps / 4)
            return O.getMagLsFilters2D(hL, hR, azi, N, fs, ln, basis)
        azi, zen = synth.fibonacci_grid(D)
        hL, hR = synth.rigid_sphere_hrirs(azi, zen, fs=fs, taps=taps, centre_delay=taps / 4)
        if kind == "ls":
            return O.getLsFilters(hL, hR, azi, zen, N, basis)
        if kind == "magls":
            return O.getMagLsFilters(hL, hR, azi, zen, N, fs, ln, basis)
        if kind in ("emainch", "emainsh"):
            ma = np.linspace(0, 2 * np.pi, M, endpoint=False) + 0.2
            fn = O.getEMagLsFiltersEMAinCH if kind == "emainch" else O.getEMagLsFiltersEMAinSH
            return fn(hL, hR, azi, zen, r, ma, N, fs, ln, basis)
        ma, mz = SC.mics(M, D + M)
        fn = O.getEMagLsFilters if kind == "emagls" else O.getEMagLs2Filters
        return fn(hL, hR, azi, zen, r, ma, mz, N, fs, ln, basis)
    finally:
        np.linalg.svd = orig


def exact_row_probe(case):
    """(GPU rows vs 40-digit rows, oracle FP64 rows vs 40-digit rows), worst over bins 2-3 (1-based) and both ears, for an eMagLS2 or
    real-basis eMagLS case whose bins 2-3 are least-squares bins; None when they are not."""
    from emagls_amd import Plan, synth, _lib as L
    from oracle import emagls_oracle as O
    import shape_cases as SC
    from tools.exact_rows import exact_ls_rows, oracle_ls_rows
    kind, D, taps, ln, fs, r, M, N, basis = case
    raw = kind == "emagls2"
    azi, zen = synth.fibonacci_grid(D)
    hL, hR = synth.rigid_sphere_hrirs(azi, zen, fs=fs, taps=taps, centre_delay=taps / 4)
    ma, mz = SC.mics(M, D + M)
    nfft, f, P, k_cut = O._design_consts(fs, ln, max(O.F_CUT_MIN_FREQ, 500 * N))
    if k_cut < 4:
        return None
    p = Plan(L.KIND_EMAGLS2 if raw else L.KIND_EMAGLS, basis, N, fs, ln, hL.shape[0], hL.shape[1], r, M)
    p.set_hrir_grid(azi, zen)
    p.set_mic_grid(ma, mz)
    p.set_hrirs(hL, hR)
    p.execute()
    p.get_filters()
    C = M if raw else (N + 1) ** 2
    W = p.debug("W", np.complex128).reshape(2, P, -1)[:, :, :C]
    p.close()
    HL, HR, gL, gR = O._hrir_prologue(hL, hR, nfft, P)
    smair, simOrder = O.getSMAIRMatrix(O.SMAIR_DEFAULT_ORDER if raw else N, fs, nfft, r, np.column_stack([ma, mz]), basis, returnRawMicSigs=raw)
    Yc = O.getSH(simOrder, np.column_stack([azi, zen]), basis).conj().T
    nrm = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
    g = o = 0.0
    for k in (2, 3):
        rows_x, _ = exact_ls_rows(smair[:, :, k - 1], Yc, (HL[k - 1], HR[k - 1]))
        rows_o, _ = oracle_ls_rows(smair[:, :, k - 1], Yc, (HL[k - 1], HR[k - 1]))
        for e in range(2):
            g, o = max(g, nrm(W[e, k - 1], rows_x[e])), max(o, nrm(rows_o[e], rows_x[e]))
    return g, o


def noise_rank_probe(case):
    """For an array design: how many singular values of the reference's FP64 pwGrid lie below 1e-14 of the largest (the FP64 noise floor is
    3e-17) in its lowest solved bins -- (worst count, channels, bin).  The reference weights the singular vectors of ALL of them with
    100 / s_max (1 % clipping, lib/getEMagLsFilters.m:96-99): vectors of singular values at the noise floor are decided by the rounding
    errors of the reference's own matrix product, and no other arithmetic reproduces them."""
    from emagls_amd import synth
    from oracle import emagls_oracle as O
    import shape_cases as SC
    kind, D, taps, ln, fs, r, M, N, basis = case
    if kind not in ("emagls", "emagls2", "emainch"):
        return None
    azi, zen = synth.fibonacci_grid(D)
    nfft, f, P, k_cut = O._design_consts(fs, ln, max(O.F_CUT_MIN_FREQ, 500 * N))
    if kind == "emainch":
        ma = np.linspace(0, 2 * np.pi, M, endpoint=False) + 0.2
        smair, simOrder = O.getSMAIRMatrix(N, fs, nfft, r, np.column_stack([ma, np.full(M, np.pi / 2)]), basis, returnRawMicSigs=True)
        Lp = O.pinv(O.getCH(N, ma, basis))
    else:
        ma, mz = SC.mics(M, D + M)
        raw = kind == "emagls2"
        smair, simOrder = O.getSMAIRMatrix(O.SMAIR_DEFAULT_ORDER if raw else N, fs, nfft, r, np.column_stack([ma, mz]), basis, returnRawMicSigs=raw)
        Lp = None
    Yc = O.getSH(simOrder, np.column_stack([azi, zen]), basis).conj().T
    worst = (0, 0, 0)
    for kb in range(1, min(P - 1, 8)):
        pw = smair[:, :, kb] @ Yc
        if Lp is not None:
            pw = Lp @ pw
        sv = np.linalg.svd(pw, compute_uv=False)
        n = int((sv < 1e-14 * sv.max()).sum())
        if n > worst[0]:
            worst = (n, sv.size, kb + 1)
    return worst


def main():
    from emagls_amd._lib import EmaglsError
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    tally = dict(ok=0, refused=0, ill_posed=0, reference_noise=0, mismatch=0, error=0)
    worst = 0.0
    for i in range(n):
        c = draw(rng)
        t = time.time()
        try:
            e = run(c)
            if e < 1e-6:
                tally["ok"] += 1
                worst = max(worst, e)
                print(f"case {i} {c} -> ok rel={e:.2e} ({time.time() - t:.1f} s)", flush=True)
            elif c[0] in ("decode", "geo"):
                tally["mismatch"] += 1
                print(f"case {i} {c} -> MISMATCH rel={e:.2e} ({time.time() - t:.1f} s)", flush=True)
            else:
                # is the REFERENCE's result defined to that accuracy?  the oracle against itself with the other LAPACK SVD driver
                import shape_cases as SC
                a, b = oracle_filters(c, "gesdd"), oracle_filters(c, "gesvd")
                self_dev = max(SC.rel(a[0], b[0]), SC.rel(a[1], b[1]))
                ill = self_dev > 1e-7 and e < 100.0 * self_dev
                verdict, extra = ("ill-posed", "") if ill else ("MISMATCH", "")
                if not ill and c[0] in ("emagls", "emagls2") and (c[0] == "emagls2" or c[8] == "real"):
                    probe = exact_row_probe(c)
                    if probe is not None:
                        g, o = probe
                        extra = f", rows of bins 2-3 against 40-digit arithmetic: GPU {g:.2e}, FP64 oracle {o:.2e}"
                        if g < 1e-9 and o > 10.0 * g:
                            verdict = "reference-noise"
                if verdict == "MISMATCH":
                    nr = noise_rank_probe(c)
                    if nr is not None and nr[0] > 0:
                        verdict = "ill-posed"
                        extra += f", {nr[0]} of {nr[1]} singular values of the reference's FP64 pwGrid below 1e-14 s_max in bin {nr[2]} (weighted by 100 / s_max: noise vectors)"
                tally[{"ill-posed": "ill_posed", "MISMATCH": "mismatch", "reference-noise": "reference_noise"}[verdict]] += 1
                print(f"case {i} {c} -> {verdict} rel={e:.2e}, oracle gesdd vs gesvd {self_dev:.2e}{extra} ({time.time() - t:.1f} s)", flush=True)
        except EmaglsError as ex:
            tally["refused"] += 1
            print(f"case {i} {c} -> refused: {str(ex)[:140]}", flush=True)
        except AssertionError as ex:     # the oracle mirrors the reference's asserts ('len too short' ...)
            tally["refused"] += 1
            print(f"case {i} {c} -> oracle assert: {str(ex)[:100]}", flush=True)
        except Exception:
            tally["error"] += 1
            print(f"case {i} {c} -> ERROR\n{traceback.format_exc()[-1200:]}", flush=True)
    print("summary:", tally, "worst accepted rel = %.2e" % worst)


if __name__ == "__main__":
    main()
