"""HRIR sets on a 64-capsule array (2702 directions, 1024 taps): time per set of emagls_design_hrir_sets on the first call and on a repeat
(the plans of the 33..64-channel path keep their geometry stages between sets of one geometry)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
from tools.bench_secondary import _grids  # noqa: E402
import emagls_amd as E  # noqa: E402
from emagls_amd import synth, _lib as L  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
azi, zen, _, _ = _grids()
sets = [synth.rigid_sphere_hrirs(azi, zen, seed=5 + j) for j in range(n)]
hL = np.stack([s[0] for s in sets], axis=2)
hR = np.stack([s[1] for s in sets], axis=2)
maz, mzn = synth.fibonacci_grid(64)
L.check(L.load().emagls_cache_clear())
for call in range(3):
    t0 = time.perf_counter()
    wL, wR = E.designHrirSets("emagls2", hL, hR, azi, zen, 0.042, maz, mzn, 4, 48000.0, 1024, "real")
    dt = time.perf_counter() - t0
    print("call %d: %d sets in %.1f ms = %.2f ms per set (checksum %.10e)" % (call, n, dt * 1e3, dt * 1e3 / n, float(np.abs(wL).sum())))
