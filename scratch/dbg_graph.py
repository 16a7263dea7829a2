import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emagls_amd import Plan, _lib as L, synth
g = np.load('tests/golden/ref_fixtures.npz')
azi, zen = g['grid/hrirGridAziRad'], g['grid/hrirGridZenRad']
maz, mzn = g['grid/micGridAziRad'], g['grid/micGridZenRad']
hL, hR = synth.rigid_sphere_hrirs(azi, zen)
p = Plan(L.KIND_EMAGLS, 'complex', 4, 48000.0, 512, 128, 2702, 0.042, 32)
p.set_hrir_grid(azi, zen); p.set_mic_grid(maz, mzn); p.set_hrirs(hL, hR)
ref = None
for it in range(5):
    p.execute(); p.synchronize()
    flag = p.debug('flag', np.int32)
    S=400
    R = p.debug('R', np.complex128, (S, S))
    Yc = p.debug('Yc', np.complex128)
    print(it, 'flag', flag, 'Rdiag', np.abs(np.diag(R))[:3], np.isnan(R).sum(), 'Yc nan', np.isnan(Yc).sum(), 'Yc absmax', np.abs(Yc).max())
    try:
        wL, wR = p.get_filters()
        if ref is None: ref = wL
        print('   filters maxabs', np.abs(wL).max(), 'diff vs first', np.abs(wL-ref).max())
    except Exception as e:
        print('   ERR', e)
