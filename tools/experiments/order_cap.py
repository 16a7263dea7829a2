"""Simulation orders above 47 (array radii above 10.9 cm at 48 kHz; 5.5 cm at 96 kHz): designs against the oracle.
    python tools/experiments/order_cap.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
import numpy as np
import emagls_amd as E
from emagls_amd import synth
from oracle import emagls_oracle as O
import shape_cases as SC
g = np.load(os.path.join(ROOT, "tests", "golden", "ref_fixtures.npz"))
maz, mzn = g["grid/micGridAziRad"].ravel(), g["grid/micGridZenRad"].ravel()
azi, zen = synth.fibonacci_grid(1500)
hL, hR = synth.rigid_sphere_hrirs(azi, zen, taps=64)
for fn, oracle, name in ((E.getEMagLsFilters, O.getEMagLsFilters, "eMagLS"), (E.getEMagLs2Filters, O.getEMagLs2Filters, "eMagLS2")):
    for r in (0.146, 0.16, 0.18, 0.193):
        t = time.time()
        try:
            w = fn(hL, hR, azi, zen, r, maz, mzn, 4, 48000.0, 96, "real")
            o = oracle(hL, hR, azi, zen, r, maz, mzn, 4, 48000.0, 96, "real")
            print(f"{name} r = {100 * r:.1f} cm (simulation order {int(np.ceil(48000 * np.pi * r / 343))}): rel L {SC.rel(w[0], o[0]):.2e} R {SC.rel(w[1], o[1]):.2e} ({time.time() - t:.1f} s)", flush=True)
        except Exception as e:
            print(f"{name} r = {100 * r:.1f} cm: {type(e).__name__}: {str(e)[:200]}", flush=True)
