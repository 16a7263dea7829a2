"""Generator of tests/golden/case27_truth.npz (CPU only, ~15 min): the least-squares rows W(k,:) = H(k,:) Y_reg_inv_k of the three
lowest solved bins of the round-5 campaign's case 27 family -- eMagLS2, r = 8.5 mm, 96 kHz (k r = 0.04 at the first solved bin),
1016 directions, 184 taps, 32 / 36 / 42 / 48 microphones -- in 40-digit arithmetic on the oracle's own FP64 inputs
(tools/exact_rows.py), next to the oracle's FP64 rows.  tests/test_gpu_wide_arrays.py compares the GPU path with both.

    python tests/golden/make_case27_truth.py [nmics ...]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from emagls_amd import synth  # noqa: E402
from oracle import emagls_oracle as O  # noqa: E402
import shape_cases as SC  # noqa: E402
from tools.exact_rows import exact_ls_rows, oracle_ls_rows  # noqa: E402

D, taps, ln, fs, r, N, basis = 1016, 16, 184, 96000.0, 0.008520696789501618, 2, "complex"
BINS = (2, 3, 4)      # 1-based, the lowest solved bins
OUT = os.path.join(ROOT, "tests", "golden", "case27_truth.npz")


def inputs(M):
    azi, zen = synth.fibonacci_grid(D)
    hL, hR = synth.rigid_sphere_hrirs(azi, zen, fs=fs, taps=taps, centre_delay=taps / 4)
    ma, mz = SC.mics(M, D + M)
    return azi, zen, hL, hR, ma, mz


def main():
    counts = [int(a) for a in sys.argv[1:]] or [32, 36, 42, 48]
    store = dict(np.load(OUT)) if os.path.exists(OUT) else {}
    for M in counts:
        azi, zen, hL, hR, ma, mz = inputs(M)
        nfft, f, P, k_cut = O._design_consts(fs, ln, max(O.F_CUT_MIN_FREQ, 500 * N))
        HL, HR, gL, gR = O._hrir_prologue(hL, hR, nfft, P)
        smair, simOrder = O.getSMAIRMatrix(O.SMAIR_DEFAULT_ORDER, fs, nfft, r, np.column_stack([ma, mz]), basis, returnRawMicSigs=True)
        Yc = O.getSH(simOrder, np.column_stack([azi, zen]), basis).conj().T
        for k in BINS:
            t0 = time.time()
            pM = smair[:, :, k - 1]
            (wl_o, wr_o), s = oracle_ls_rows(pM, Yc, (HL[k - 1], HR[k - 1]))
            (wl_x, wr_x), s_x = exact_ls_rows(pM, Yc, (HL[k - 1], HR[k - 1]))
            el = np.linalg.norm(wl_o - wl_x) / np.linalg.norm(wl_x)
            er = np.linalg.norm(wr_o - wr_x) / np.linalg.norm(wr_x)
            print(f"M={M} bin {k}: kr={2 * np.pi * f[k - 1] / 343.0 * r:.4f} s_min/s_max {s_x.min() / s_x.max():.2e}; oracle FP64 row vs 40-digit row: "
                  f"L {el:.2e} R {er:.2e}  |W(k)|={np.linalg.norm(wl_x):.3e} ({time.time() - t0:.0f} s)", flush=True)
            for nm, v in (("wl", wl_x), ("wr", wr_x), ("s", s_x), ("wl_fp64", wl_o), ("wr_fp64", wr_o)):
                store[f"m{M}_k{k}_{nm}"] = v
        np.savez(OUT, **store)


if __name__ == "__main__":
    main()
