import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emagls_amd import Plan, _lib as L, synth
from oracle import emagls_oracle as O
azi, zen = synth.fibonacci_grid(900)
maz, mzn = synth.em32_grid()
hL, hR = synth.rigid_sphere_hrirs(azi, zen, taps=64)
flen = 256
p = Plan(L.KIND_EMAGLS, 'complex', 4, 48000.0, flen, 64, 900, 0.042, 32)
p.set_hrir_grid(azi, zen); p.set_mic_grid(maz, mzn); p.set_hrirs(hL, hR)
p.execute(); wL, wR = p.get_filters()
oL, oR = O.getEMagLsFilters(hL, hR, azi, zen, 0.042, maz, mzn, 4, 48000.0, flen, 'complex')
print('rel', np.abs(wL-oL).max()/np.abs(oL).max())
FL, FO = np.fft.fft(wL, axis=0), np.fft.fft(oL, axis=0)
err = np.abs(FL-FO).max(axis=1)/np.abs(FO).max()
print('k_cut', p.info().k_cut, 'worst bins', np.argsort(err)[-12:][::-1], np.sort(err)[-12:][::-1])
print('err by bin (first 24)', np.round(err[:24], 10))
