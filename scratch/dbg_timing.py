import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ['EMAGLS_SWEEP_TIMING']='1'
from emagls_amd import Plan, _lib as L, synth
g = np.load('tests/golden/ref_fixtures.npz')
azi, zen = g['grid/hrirGridAziRad'], g['grid/hrirGridZenRad']
maz, mzn = g['grid/micGridAziRad'], g['grid/micGridZenRad']
hL, hR = synth.rigid_sphere_hrirs(azi, zen)
p = Plan(L.KIND_EMAGLS, 'complex', 4, 48000.0, 512, 128, 2702, 0.042, 32)
p.set_hrir_grid(azi, zen); p.set_mic_grid(maz, mzn); p.set_hrirs(hL, hR)
for it in range(4):
    p.execute(); p.synchronize()
t = p.debug('sweep_timing', np.int64).reshape(-1,16)
ks=[100,101,200,300,400]
for k in ks:
    r=t[k]; d=np.diff(r[:10])
    print(k, 'phase cycles', d.tolist(), 'total', r[9]-r[0], 'wall(100MHz ticks)', r[14]-r[15])
tot=(t[60:500,9]-t[60:500,0]); wall=(t[60:500,14]-t[60:500,15])
print('mean cycles', tot.mean(), 'mean wall us', wall.mean()/100.0, '=> clock GHz', tot.mean()/ (wall.mean()/100.0)/1e3)
print('mean phases', np.diff(t[60:500,:10],axis=1).mean(axis=0).round(0).tolist())
m=t[60:500]; print('gather loads', (m[:,10]-m[:,0]).mean(), 'B loads', (m[:,11]-m[:,10]).mean(), 'q loads', (m[:,12]-m[:,11]).mean())
print('launch-to-launch wall us', np.diff(t[60:500,15]).mean()/100.0)
