"""How a lane batch of n designs sweeps: persistent launch or the launch-per-bin fallback, and how long it takes."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emagls_amd import Batch, Plan, synth, _lib as L

g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_fixtures.npz"))
azi, zen, maz, mzn = g["grid/hrirGridAziRad"], g["grid/hrirGridZenRad"], g["grid/micGridAziRad"], g["grid/micGridZenRad"]
for n in [int(x) for x in (sys.argv[1:] or ["8", "12", "16"])]:
    plans = []
    for j in range(n):
        hL, hR = synth.rigid_sphere_hrirs(azi, zen, seed=100 + j)
        p = Plan(L.KIND_EMAGLS, "complex", 4, 48000.0, 512, 128, 2702, 0.042, 32)
        p.set_hrir_grid(azi, zen); p.set_mic_grid(maz, mzn); p.set_hrirs(hL, hR)
        plans.append(p)
    b = Batch(plans)
    for it in range(3):
        t0 = time.perf_counter(); b.execute(); b.synchronize(); t1 = time.perf_counter()
        out = b.get_filters()
        print(f"n={n} execute {it}: {(t1 - t0) * 1e3:.2f} ms; sweep launches per design: {plans[0].info().num_sweep_launches}", flush=True)
    b.set_profiling(1)
    b.execute(); b.synchronize()
    try:
        print(f"n={n} sweep kernel: {b.sweep_time_ms():.3f} ms")
    except Exception as e:
        print(f"n={n} sweep time unavailable: {e}")
    b.close()
    for p in plans: p.close()
