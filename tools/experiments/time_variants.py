"""Warm one-shot latency of the design variants at the config-3 size (2702 directions, 512 taps, em32 / 16-microphone equatorial array)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import emagls_amd as E
from emagls_amd import synth

g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "ref_fixtures.npz"))
azi, zen, maz, mzn = g["grid/hrirGridAziRad"], g["grid/hrirGridZenRad"], g["grid/micGridAziRad"], g["grid/micGridZenRad"]
hL, hR = synth.rigid_sphere_hrirs(azi, zen)
eq = np.linspace(0, 2 * np.pi, 16, endpoint=False)


def t(fn, *a, **k):
    for _ in range(3):
        fn(*a, **k)
    t0 = time.perf_counter()
    for _ in range(5):
        fn(*a, **k)
    return (time.perf_counter() - t0) / 5 * 1e3


print("getEMagLsFilters complex           %.2f ms" % t(E.getEMagLsFilters, hL, hR, azi, zen, 0.042, maz, mzn, 4, 48000.0, 512, "complex"))
print("getEMagLsFilters real + diffuseness %.2f ms" % t(E.getEMagLsFilters, hL, hR, azi, zen, 0.042, maz, mzn, 4, 48000.0, 512, "real", applyDiffusenessConst=True))
print("getMagLsFilters real               %.2f ms" % t(E.getMagLsFilters, hL, hR, azi, zen, 4, 48000.0, 512, "real"))
print("getMagLsFilters real + diffuseness %.2f ms" % t(E.getMagLsFilters, hL, hR, azi, zen, 4, 48000.0, 512, "real", applyDiffusenessConst=True))
print("getEMagLsFiltersEMAinCH real       %.2f ms" % t(E.getEMagLsFiltersEMAinCH, hL, hR, azi, zen, 0.042, eq, 4, 48000.0, 512, "real"))
print("getEMagLsFiltersEMAinSH real       %.2f ms" % t(E.getEMagLsFiltersEMAinSH, hL, hR, azi, zen, 0.042, eq, 4, 48000.0, 512, "real"))
print("getEMagLsFiltersEMAinSH complex    %.2f ms" % t(E.getEMagLsFiltersEMAinSH, hL, hR, azi, zen, 0.042, eq, 4, 48000.0, 512, "complex"))
