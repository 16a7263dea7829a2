"""Designs on HRIR grids larger than the 2702-point one (lib/*.m take any direction count): which entry points accept them and how
they compare with the oracle.   python tools/experiments/big_grids.py [D ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def main():
    import emagls_amd as E
    from emagls_amd import synth
    from oracle import emagls_oracle as O
    for D in [int(x) for x in sys.argv[1:]] or [6000, 12000]:
        azi, zen = synth.fibonacci_grid(D)
        hL, hR = synth.rigid_sphere_hrirs(azi, zen, taps=64)
        maz, mzn = synth.em32_grid()
        atf, aazi, azen = synth.glasses_atfs(natf=D + 500, nmics=8, taps=64)
        hg, ag = np.column_stack([azi, zen]), np.column_stack([aazi, azen])
        cases = [
            ("getLsFilters", (hL, hR, azi, zen, 4), {}),
            ("getMagLsFilters", (hL, hR, azi, zen, 4, 48000.0, 128), {}),
            ("getEMagLsFilters", (hL, hR, azi, zen, 0.042, maz, mzn, 4, 48000.0, 128), {}),
            ("getEMagLs2Filters", (hL, hR, azi, zen, 0.042, maz, mzn, 4, 48000.0, 128), {}),
        ]
        for name, args, kw in cases:
            t0 = time.time()
            try:
                w = getattr(E, name)(*args, **kw)
                o = getattr(O, name)(*args, **kw)
                print(f"D = {D} {name}: rel L {rel(w[0], o[0]):.2e} R {rel(w[1], o[1]):.2e}  ({time.time() - t0:.1f} s)", flush=True)
            except Exception as e:
                print(f"D = {D} {name}: {type(e).__name__}: {e}", flush=True)
        try:
            w = E.getEMagLsFiltersFromAtf(hL, hR, hg, atf, ag, 48000.0, 128, 2000.0, verbose=False)
            o = O.getEMagLsFiltersFromAtf(hL, hR, hg, atf, ag, 48000.0, 128, 2000.0)
            print(f"D = {D} getEMagLsFiltersFromAtf: rel L {rel(w[0], o[0]):.2e} R {rel(w[1], o[1]):.2e}", flush=True)
        except Exception as e:
            print(f"D = {D} getEMagLsFiltersFromAtf: {type(e).__name__}: {e}", flush=True)


if __name__ == "__main__":
    main()
