"""The 64-capsule eMagLS2 design (tools/bench_secondary.em64): time per design and the Jacobi sweeps its bins take
(EMAGLS_WA_JACOBI_FLAG sets the rule for another sweep)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
from tools.bench_secondary import _grids  # noqa: E402
from emagls_amd import Plan, synth, _lib as L  # noqa: E402

azi, zen, _, _ = _grids()
hL, hR = synth.rigid_sphere_hrirs(azi, zen)
maz, mzn = synth.fibonacci_grid(64)
p = Plan(L.KIND_EMAGLS2, "real", 4, 48000.0, 1024, hL.shape[0], hL.shape[1], 0.042, 64)
p.set_hrir_grid(azi, zen)
p.set_mic_grid(maz, mzn)
p.set_hrirs(hL, hR)
for _ in range(2):
    p.execute()
p.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    p.execute()
    p.synchronize()
dt = (time.perf_counter() - t0) / 3
js = p.debug("jsweeps", np.int32)
n = p.info().num_pos_freqs - 1
js = js[:n]
wl, wr = p.get_filters()
print("flag %s: %.2f ms per design; Jacobi sweeps over %d bins: min %d median %d max %d mean %.1f; checksum %.12e" %
      (os.environ.get("EMAGLS_WA_JACOBI_FLAG", "1e-14"), dt * 1e3, n, js.min(), int(np.median(js)), js.max(), js.mean(), float(np.abs(wl).sum() + np.abs(wr).sum())))
p.close()
