import sys, os, numpy as np, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emagls_amd import Plan, Batch, _lib as L, synth
azi, zen = synth.fibonacci_grid(900)
maz, mzn = synth.em32_grid()
hL, hR = synth.rigid_sphere_hrirs(azi, zen, taps=64)
plans=[]
for r in (0.042, 0.040):
    p = Plan(L.KIND_EMAGLS, 'complex', 4, 48000.0, 64, 64, 900, r, 32)
    p.set_hrir_grid(azi, zen); p.set_mic_grid(maz, mzn); p.set_hrirs(hL, hR)
    plans.append(p)
print('plans ok', flush=True)
b = Batch(plans)
print('batch created', flush=True)
for it in range(3):
    b.execute(); print('executed', it, flush=True)
    b.synchronize(); print('synced', it, flush=True)
    res = b.get_filters(); print('got', it, np.abs(res[0][0]).max(), np.abs(res[1][0]).max(), flush=True)
