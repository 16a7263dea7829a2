"""emagls_from_atf_hrir_sets at BASELINE config 5's size: 8 subjects of one ATF set (16 384 directions x 8 microphones, 2048 taps)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import emagls_amd as E
from emagls_amd import synth
from tools.bench_secondary import _grids
azi, zen, _, _ = _grids()
atf, aazi, azen = synth.glasses_atfs(natf=16384, nmics=8, taps=256)
hg, ag = np.column_stack([azi, zen]), np.column_stack([aazi, azen])
subs = [synth.rigid_sphere_hrirs(azi, zen, seed=100 + j, head_radius=0.075 + 0.002 * j) for j in range(8)]
hL = np.asfortranarray(np.stack([s[0] for s in subs], axis=2)); hR = np.asfortranarray(np.stack([s[1] for s in subs], axis=2))
for rep in range(3):
    t0 = time.perf_counter(); wL, wR, dev = E.fromAtfHrirSets(hL, hR, hg, atf, ag, 48000.0, 2048, 2000.0); dt = time.perf_counter() - t0
t0 = time.perf_counter()
for j in range(3):
    E.getEMagLsFiltersFromAtf(hL[:, :, j], hR[:, :, j], hg, atf, ag, 48000.0, 2048, 2000.0, verbose=False)
ds = (time.perf_counter() - t0) / 3
print("config 5: 8 subjects in one call %.1f ms (%.2f ms per subject, host arrays incl. the 268 MB ATF set); single calls %.1f ms each" % (dt * 1e3, dt * 1e3 / 8, ds * 1e3))
