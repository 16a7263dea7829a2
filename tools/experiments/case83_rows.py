"""Campaign 4's one case beyond the factor-100 rule (seed 81, case 83: EMAinCH order 12, 27 microphones, r = 3.3 cm, 3832 directions):
  gpu   dump the library's rows W(k,:) of every bin (plan buffer "W") and the filters  -> gpurun_out/case83_gpu.npz
  cpu   the oracle's rows per bin against them; the bins furthest apart; their least-squares rows in 40-digit arithmetic
        (pwGrid_CH = pinv(CH) (pMics Y^H) from the oracle's own FP64 factors taken as exact: tools/exact_rows.py)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from emagls_amd import synth  # noqa: E402

D, taps, ln, fs, r, M, N, basis = 3832, 100, 376, 48000.0, 0.03342296456120314, 27, 12, "complex"
azi, zen = synth.fibonacci_grid(D)
hL, hR = synth.rigid_sphere_hrirs(azi, zen, fs=fs, taps=taps, centre_delay=taps / 4)
ma = np.linspace(0, 2 * np.pi, M, endpoint=False) + 0.2
C = 2 * N + 1
OUT = os.path.join(ROOT, "gpurun_out", "case83_gpu.npz")
nrm = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))

if sys.argv[1] == "gpu":
    from emagls_amd import Plan, _lib as L
    p = Plan(L.KIND_EMA_CH, basis, N, fs, ln, hL.shape[0], hL.shape[1], r, M)
    p.set_hrir_grid(azi, zen)
    p.set_mic_grid(ma, None)
    p.set_hrirs(hL, hR)
    p.execute()
    wl, wr = p.get_filters()
    P = p.info().num_pos_freqs
    W = p.debug("W", np.complex128).reshape(2, P, -1)[:, :, :C]
    np.savez(OUT, W=W, wl=wl, wr=wr)
    print("saved", OUT, W.shape)
else:
    from oracle import emagls_oracle as O
    from tools.exact_rows import _to_mp
    import shape_cases as SC
    G = np.load(OUT)
    nfft, f, P, k_cut = O._design_consts(fs, ln, max(O.F_CUT_MIN_FREQ, 500 * N))
    micGrid = np.column_stack([ma, np.full(M, np.pi / 2)])
    smair, simOrder = O.getSMAIRMatrix(N, fs, nfft, r, micGrid, basis, returnRawMicSigs=True)
    Yc = O.getSH(simOrder, np.column_stack([azi, zen]), basis).conj().T
    Lp = O.pinv(O.getCH(N, ma, basis))
    HL, HR, gL, gR = O._hrir_prologue(hL, hR, nfft, P)
    Wl, Wr = O._emagls_core(HL, HR, lambda k: Lp @ (smair[:, :, k - 1] @ Yc), P, k_cut, C)
    dev = np.array([max(nrm(G["W"][0, k], Wl[k]), nrm(G["W"][1, k], Wr[k])) for k in range(1, P)])
    worst = 1 + np.argsort(-dev)[:6]
    print("k_cut (1-based) %d of %d bins, simulation order %d; rows GPU vs oracle, worst bins (0-based): %s" % (k_cut, P, simOrder, ", ".join("%d: %.2e" % (k, dev[k - 1]) for k in worst)))
    o = O.getEMagLsFiltersEMAinCH(hL, hR, azi, zen, r, ma, N, fs, ln, basis)
    print("filters GPU vs oracle: %.2e" % max(SC.rel(G["wl"], o[0]), SC.rel(G["wr"], o[1])))
    import mpmath as mp
    ls = [int(k) for k in worst if k < k_cut - 1][:2]      # least-squares bins among them (0-based kb < k_cut - 1)
    for kb in ls:
        t0 = time.time()
        with mp.workdps(40):
            A = (_to_mp(Lp) * (_to_mp(smair[:, :, kb]) * _to_mp(Yc))).T       # D x C = pwGrid_CH.'
            Q, R = mp.qr(A, mode="skinny")
            U2, s, Vh = mp.svd_c(R)
            smax = max(s)
            sreg = [1 / max(x, mp.mpf("0.01") * smax) for x in s]
            Qc, U2c, Vhc = Q.apply(mp.conj), U2.apply(mp.conj), Vh.apply(mp.conj)
            rows = []
            for H in (HL[kb], HR[kb]):
                t = (_to_mp(np.asarray(H)[None, :]) * Qc) * U2c
                for j in range(C):
                    t[0, j] = t[0, j] * sreg[j]
                w = t * Vhc
                rows.append(np.array([complex(w[0, j]) for j in range(C)]))
            sv = np.array([float(x) for x in s])
        print("bin %d (0-based): s_min / s_max = %.2e; GPU vs 40-digit rows: L %.2e R %.2e | FP64 oracle vs 40-digit rows: L %.2e R %.2e  (%.0f s)" %
              (kb, sv.min() / sv.max(), nrm(G["W"][0, kb], rows[0]), nrm(G["W"][1, kb], rows[1]), nrm(Wl[kb], rows[0]), nrm(Wr[kb], rows[1]), time.time() - t0), flush=True)
