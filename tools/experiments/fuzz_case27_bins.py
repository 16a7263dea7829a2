"""Per-bin error profile of case 27 of the round-5 campaign (42-microphone eMagLS2, r = 8.5 mm, 96 kHz, 184 taps)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np
import emagls_amd as E
from emagls_amd import synth
from oracle import emagls_oracle as O
import shape_cases as SC
kind, D, taps, ln, fs, r, M, N, basis = ('emagls2', 1016, 16, 184, 96000.0, 0.008520696789501618, 42, 2, 'complex')
azi, zen = synth.fibonacci_grid(D)
hL, hR = synth.rigid_sphere_hrirs(azi, zen, fs=fs, taps=taps, centre_delay=taps / 4)
ma, mz = SC.mics(M, D + M)
for MM in (42, 32, 36, 48):
    ma, mz = SC.mics(MM, D + MM)
    w = E.getEMagLs2Filters(hL, hR, azi, zen, r, ma, mz, N, fs, ln, basis)
    o = O.getEMagLs2Filters(hL, hR, azi, zen, r, ma, mz, N, fs, ln, basis)
    print(MM, "mics: rel L", SC.rel(w[0], o[0]), "R", SC.rel(w[1], o[1]))
    if MM == 42:
        Wg, Wo = np.fft.fft(w[0], axis=0), np.fft.fft(o[0], axis=0)
        e = np.linalg.norm(Wg - Wo, axis=1) / np.linalg.norm(Wo)
        order = np.argsort(-e)[:12]
        print("  bins with the largest share of the error (bin, share):", [(int(k), float("%.2e" % e[k])) for k in order])
        print("  k_cut (1-based) =", int(np.ceil(max(1000.0, 500.0 * N) / (fs / 2 / (ln)))), " nfft =", 2 * ln)
