"""getEMagLsFiltersEMAinSH at the config-3 size (2702 directions, 512 taps, 16 equatorial microphones): time per design by order, and the
stage times of one profiled run."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
from tools.bench_secondary import _grids  # noqa: E402
from emagls_amd import Plan, synth, _lib as L  # noqa: E402

azi, zen, _, _ = _grids()
hL, hR = synth.rigid_sphere_hrirs(azi, zen)
for order, M in ((1, 8), (4, 16), (6, 20)):
    maz = np.linspace(0, 2 * np.pi, M, endpoint=False) + 0.1
    p = Plan(L.KIND_EMA_SH, "real", order, 48000.0, 512, hL.shape[0], hL.shape[1], 0.05, M)
    p.set_hrir_grid(azi, zen)
    p.set_mic_grid(maz, None)
    p.set_hrirs(hL, hR)
    for _ in range(3):
        p.execute()
    p.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        p.execute()
        p.synchronize()
    dt = (time.perf_counter() - t0) / 3
    p.set_profiling(1)
    p.execute()
    p.synchronize()
    st = p.stage_times()
    print("EMAinSH order %d, %d microphones: %.2f ms per design; stages: %s" % (order, M, dt * 1e3, ", ".join("%s %.2f" % (k, v) for k, v in st)))
    p.close()
