"""One EMAinSH design of order 6 (49 channels, 20 microphones, 2702 directions, 512 taps) a few times (for rocprofv3 --kernel-trace --stats)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
from tools.bench_secondary import _grids  # noqa: E402
from emagls_amd import Plan, synth, _lib as L  # noqa: E402

azi, zen, _, _ = _grids()
hL, hR = synth.rigid_sphere_hrirs(azi, zen)
order, M = 6, 20
maz = np.linspace(0, 2 * np.pi, M, endpoint=False) + 0.1
p = Plan(L.KIND_EMA_SH, "real", order, 48000.0, 512, hL.shape[0], hL.shape[1], 0.05, M)
p.set_hrir_grid(azi, zen)
p.set_mic_grid(maz, None)
p.set_hrirs(hL, hR)
for _ in range(4):
    p.execute()
p.synchronize()
p.close()
