"""designHrirSets (emagls_design_hrir_sets) with host arrays: 64 HRIR sets of config 3's / config 2's design in one call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import emagls_amd as E
from emagls_amd import synth
from tools.bench_secondary import _grids
azi, zen, maz, mzn = _grids()
n = 64
sets = [synth.rigid_sphere_hrirs(azi, zen, seed=300 + j) for j in range(n)]
hL = np.asfortranarray(np.stack([s[0] for s in sets], axis=2)); hR = np.asfortranarray(np.stack([s[1] for s in sets], axis=2))
for kind, kw in (("emagls", dict(micRadius=0.042, micGridAziRad=maz, micGridZenRad=mzn, order=4, fs=48000.0, len=512, shDefinition="complex")),
                 ("magls", dict(order=4, fs=48000.0, len=512, shDefinition="real")), ("ls", dict(order=4))):
    for rep in range(3):
        t0 = time.perf_counter()
        wL, wR = E.designHrirSets(kind, hL, hR, azi, zen, **kw)
        dt = time.perf_counter() - t0
    t0 = time.perf_counter()
    single = {"emagls": lambda a, b: E.getEMagLsFilters(a, b, azi, zen, 0.042, maz, mzn, 4, 48000.0, 512, "complex"),
              "magls": lambda a, b: E.getMagLsFilters(a, b, azi, zen, 4, 48000.0, 512, "real"),
              "ls": lambda a, b: E.getLsFilters(a, b, azi, zen, 4, "real")}[kind]
    for j in range(8):
        single(hL[:, :, j], hR[:, :, j])
    ds = (time.perf_counter() - t0) / 8
    print("%s: %d sets in one call %.2f ms = %.0f sets/s (host arrays in and out); single calls %.2f ms each = %.0f sets/s" % (kind, n, dt * 1e3, n / dt, ds * 1e3, 1 / ds))
