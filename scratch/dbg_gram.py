import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["EMAGLS_GRAM_COND_EST"] = "1e30"
import torch
from emagls_amd import Plan, _lib as L, synth
from oracle import emagls_oracle as O
azi, zen = synth.fibonacci_grid(900)
maz, mzn = synth.em32_grid()
hL, hR = synth.rigid_sphere_hrirs(azi, zen, taps=64)
p = Plan(L.KIND_EMAGLS, 'complex', 4, 48000.0, 128, 64, 900, 0.007, 32)
p.set_hrir_grid(azi, zen); p.set_mic_grid(maz, mzn); p.set_hrirs(hL, hR)
p.execute(); p.synchronize()
print('flags after first run', p.debug("flag", np.int32), 'route sum', p.debug("route", np.int32).sum(), 'k_cut', p.info().k_cut)
wL, wR = p.get_filters()
print('flags after get', p.debug("flag", np.int32), 'route sum', p.debug("route", np.int32).sum())
oL, oR = O.getEMagLsFilters(hL, hR, azi, zen, 0.007, maz, mzn, 4, 48000.0, 128, 'complex')
print('rel', np.abs(wL-oL).max()/np.abs(oL).max())
ok = p.debug("cond_ok", np.float64); print('cond_ok', ok[:40])
r = p.debug("route", np.int32); print('route nz', np.nonzero(r)[0], 'P', p.info().num_pos_freqs)
FL, FO = np.fft.fft(wL, axis=0), np.fft.fft(oL, axis=0)
err = np.abs(FL-FO).max(axis=1)/np.abs(FO).max()
print('err by bin', np.round(err[:40], 9))
os.environ["EMAGLS_GRAM_ROUTE"] = "0"
q = Plan(L.KIND_EMAGLS, 'complex', 4, 48000.0, 128, 64, 900, 0.007, 32)
q.set_hrir_grid(azi, zen); q.set_mic_grid(maz, mzn); q.set_hrirs(hL, hR)
q.execute(); w2L, w2R = q.get_filters()
print('no-gram plan rel', np.abs(w2L-oL).max()/np.abs(oL).max(), 'vs fallback', np.abs(w2L-wL).max()/np.abs(oL).max())
