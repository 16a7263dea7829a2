"""64-microphone eMagLS2 designs at array radii beyond 5.9 cm (simulation order above 26) against the oracle.
    EMAGLS_WIDE_SIM_ORDER_MAX=47 python tools/experiments/wide_radius.py [radius_cm ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import emagls_amd as E
from emagls_amd import synth
from oracle import emagls_oracle as O
from tools.bench_secondary import _grids
azi, zen, _, _ = _grids()
hL, hR = synth.rigid_sphere_hrirs(azi, zen, taps=64)
maz, mzn = synth.fibonacci_grid(64)
rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
for r in [float(x) / 100 for x in sys.argv[1:]] or [0.07, 0.08, 0.10]:
    args = (hL, hR, azi, zen, r, maz, mzn, 4, 48000.0, 128, "real")
    t0 = time.time()
    try:
        w = E.getEMagLs2Filters(*args)
        o = O.getEMagLs2Filters(*args)
        print(f"r = {100 * r:.1f} cm: rel L {rel(w[0], o[0]):.2e} R {rel(w[1], o[1]):.2e} ({time.time() - t0:.1f} s)", flush=True)
    except Exception as e:
        print(f"r = {100 * r:.1f} cm: {type(e).__name__}: {e}", flush=True)
