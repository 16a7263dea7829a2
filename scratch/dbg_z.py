import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emagls_amd import Plan, _lib as L, synth
g = np.load('tests/golden/ref_fixtures.npz')
azi, zen = g['grid/hrirGridAziRad'], g['grid/hrirGridZenRad']
maz, mzn = g['grid/micGridAziRad'], g['grid/micGridZenRad']
hL, hR = synth.rigid_sphere_hrirs(azi, zen)
sub = slice(0, 2702, 3)
for prof in (0, 1):
    p = Plan(L.KIND_EMAGLS, 'complex', 4, 48000.0, 128, hL.shape[0], hL[:, sub].shape[1], 0.042, 32)
    p.set_hrir_grid(azi[sub], zen[sub]); p.set_mic_grid(maz, mzn); p.set_hrirs(hL[:, sub], hR[:, sub])
    if prof: p.set_profiling(1)
    p.execute(); p.synchronize()
    i = p.info(); P = i.num_pos_freqs; C = 25; ldS = 448
    Z = p.debug("Z", np.complex128).reshape(P, C, ldS)
    nz = [kb for kb in range(P) if np.abs(Z[kb]).max() > 0]
    print('prof', prof, 'k_cut', i.k_cut, 'nonzero Z bins', nz[:40], len(nz))
    ok = p.debug("cond_ok", np.float64)
    print('cond_ok[:16]', ok[:16])
