"""Stage times and kernel averages of the 64-capsule eMagLS2 design (the plain S-space path of wide_array.hip)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools.bench_secondary import _grids
from emagls_amd import Plan, synth, _lib as L
azi, zen, _, _ = _grids()
hL, hR = synth.rigid_sphere_hrirs(azi, zen)
maz, mzn = synth.fibonacci_grid(64)
p = Plan(L.KIND_EMAGLS2, "real", 4, 48000.0, 1024, hL.shape[0], hL.shape[1], 0.042, 64)
p.set_hrir_grid(azi, zen); p.set_mic_grid(maz, mzn); p.set_hrirs(hL, hR)
for _ in range(2):
    p.execute()
p.synchronize()
p.set_profiling(1)
p.execute(); p.synchronize()
print("stages (ms):", [(k, round(v, 3)) for k, v in p.stage_times()])
i = p.info(); print("sim order", i.sim_order, "S", i.num_sh_sim, "bins", i.num_pos_freqs)
