import os, sys, numpy as np
sys.path.insert(0, "/root/repo")
import emagls_amd as E
from emagls_amd import synth
g = np.load("/root/repo/tests/golden/ref_fixtures.npz")
azi, zen = g["grid/hrirGridAziRad"][0:2702:3], g["grid/hrirGridZenRad"][0:2702:3]
hL, hR = synth.rigid_sphere_hrirs(azi, zen)
mic_azi = np.linspace(0.0, 2 * np.pi, 16, endpoint=False) + 0.1
order = int(sys.argv[1]) if len(sys.argv) > 1 else 4
basis = sys.argv[2] if len(sys.argv) > 2 else "real"
print("start", order, basis, flush=True)
wL, wR = E.getEMagLsFiltersEMAinCH(hL, hR, azi, zen, 0.042, mic_azi, order, 48000.0, 128, basis)
print("ok", wL.shape, np.abs(wL).max(), flush=True)
