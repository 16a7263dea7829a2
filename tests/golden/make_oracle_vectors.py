"""Expected outputs of the CPU oracle for cases that are too slow to recompute inside the GPU test suite.

    python tests/golden/make_oracle_vectors.py        (about 10 minutes on 8 cores)

Inputs are synthetic and seeded (emagls_amd.synth on the reference's own grids), so the tests regenerate them and only
the oracle's outputs are stored.  These are vectors of the build's own restatement (oracle/emagls_oracle.py), not of the
MATLAB reference: they pin the GPU path to the oracle at sizes the oracle needs minutes for.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from emagls_amd import synth  # noqa: E402
from oracle import emagls_oracle as O  # noqa: E402


def main():
    g = np.load(os.path.join(ROOT, "tests", "golden", "ref_fixtures.npz"))
    azi, zen = g["grid/hrirGridAziRad"], g["grid/hrirGridZenRad"]
    maz, mzn = g["grid/micGridAziRad"], g["grid/micGridZenRad"]
    hL, hR = synth.rigid_sphere_hrirs(azi, zen)
    path = os.path.join(ROOT, "tests", "golden", "oracle_vectors.npz")
    out = {}
    if os.path.exists(path) and "--all" not in sys.argv:   # (only the cases the file does not hold yet; --all recomputes everything)
        with np.load(path) as old:
            out = {k: old[k] for k in old}
    # BASELINE config 4 at full size (1024 taps): the far end of the radius batch, r = 10 cm, simulation order 44; r = 5 cm, order 22
    for name, radius in (("config4_r100mm_len1024", 0.10), ("config4_r50mm_len1024", 0.05)):
        if name + "/wL" in out:
            continue
        wL, wR = O.getEMagLs2Filters(hL, hR, azi, zen, radius, maz, mzn, 4, 48000.0, 1024, "real")
        out[name + "/wL"], out[name + "/wR"] = wL, wR
    # BASELINE config 5 at full size (FromAtf, 16 384 ATF directions x 8 microphones, 2048 taps): the oracle needs 30 s
    if "config5_full/wL" not in out:
        atf, aazi, azen = synth.glasses_atfs(natf=16384, nmics=8, taps=256)
        wL, wR, _ = O.getEMagLsFiltersFromAtf(hL, hR, np.column_stack([azi, zen]), atf, np.column_stack([aazi, azen]), 48000.0, 2048, 2000.0)
        out["config5_full/wL"], out["config5_full/wR"] = wL, wR
    # the 64-capsule array at r = 8 cm (simulation order 35), 64-tap HRIRs, 128-tap filters: 35 s
    if "wide64_r80mm_len128/wL" not in out:
        h64L, h64R = synth.rigid_sphere_hrirs(azi, zen, taps=64)
        m64a, m64z = synth.fibonacci_grid(64)
        wL, wR = O.getEMagLs2Filters(h64L, h64R, azi, zen, 0.08, m64a, m64z, 4, 48000.0, 128, "real")
        out["wide64_r80mm_len128/wL"], out["wide64_r80mm_len128/wR"] = wL, wR
    # simulation orders above 47 (round 5): the em32's layout at r = 12 cm (order 53), 14.2 cm (order 63) and 19.3 cm (order 85, the last one the
    # reference's factorial-based getSH can form) on a 1500-point grid, 20-40 s each
    fa, fz = synth.fibonacci_grid(1500)
    fL, fR = synth.rigid_sphere_hrirs(fa, fz, taps=64)
    for name, fn, radius in (("order53_emagls", O.getEMagLsFilters, 0.12), ("order63_emagls2", O.getEMagLs2Filters, 0.142), ("order85_emagls2", O.getEMagLs2Filters, 0.193)):
        if name + "/wL" in out:
            continue
        wL, wR = fn(fL, fR, fa, fz, radius, maz, mzn, 4, 48000.0, 96, "real")
        out[name + "/wL"], out[name + "/wR"] = wL, wR
    np.savez_compressed(path, **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
