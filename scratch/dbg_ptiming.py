import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['EMAGLS_SWEEP_TIMING']='1'; os.environ['EMAGLS_SWEEP_PERSIST']='1'
import torch
from emagls_amd import Plan, _lib as L, synth
nd = int(sys.argv[1]) if len(sys.argv) > 1 else 2702
azi, zen = synth.fibonacci_grid(nd)
maz, mzn = synth.em32_grid()
hL, hR = synth.rigid_sphere_hrirs(azi, zen)
p = Plan(L.KIND_EMAGLS, 'complex', 4, 48000.0, 512, 128, nd, 0.042, 32)
p.set_hrir_grid(azi, zen); p.set_mic_grid(maz, mzn); p.set_hrirs(hL, hR)
for it in range(4):
    p.execute(); p.synchronize()
t = p.debug('sweep_timing', np.int64).reshape(-1,16)
m = t[60:500].astype(np.float64) / 100.0
print('per-bin period us', np.diff(m[:, 0]).mean().round(3))
print('comm: hop1 wait+reduce', (m[:,1]-m[:,0]).mean().round(3), 'hop2', (m[:,2]-m[:,1]).mean().round(3))
print('compute: B1+Mapply+B2', (m[:,3]-m[:,2]).mean().round(3), 'p/t + B3', (m[:,4]-m[:,3]).mean().round(3), 'partial+publish', (m[:,5]-m[:,4]).mean().round(3))
print('publish(kb) -> hop1 done (kb+1)', (m[1:,1]-m[:-1,5]).mean().round(3))
for k in (100, 200, 300):
    print(k, ((t[k,:6]-t[k,0])/100.0).round(2).tolist())
print('local flag', t[0,15]); print('hop1 failed polls per bin', t[60:500,9].mean())
print('first hop-1 poll round trip us', ((t[60:500,6]-t[60:500,0])/100.0).mean())
