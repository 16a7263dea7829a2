import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1]
if 'torch' in mode:
    import torch
    torch.cuda.set_device(0)
    x = torch.zeros(10, device='cuda')
from emagls_amd import Plan, _lib as L, synth
g = np.load('tests/golden/ref_fixtures.npz')
azi, zen = g['grid/hrirGridAziRad'], g['grid/hrirGridZenRad']
maz, mzn = g['grid/micGridAziRad'], g['grid/micGridZenRad']
hL, hR = synth.rigid_sphere_hrirs(azi, zen)
p = Plan(L.KIND_EMAGLS, 'complex', 4, 48000.0, 512, 128, 2702, 0.042, 32)
p.set_hrir_grid(azi, zen); p.set_mic_grid(maz, mzn); p.set_hrirs(hL, hR)
if 'nosync' in mode:
    for it in range(3): p.execute()
    p.synchronize()
    print('after 3 back-to-back executes: flag', p.debug('flag', np.int32))
for it in range(int(os.environ.get("NIT","4"))):
    p.execute()
    try:
        wL, wR = p.get_filters()
        print(it, 'ok', np.abs(wL).max())
    except Exception as e:
        print(it, 'ERR', e, p.debug('flag', np.int32))
