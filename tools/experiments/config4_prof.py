"""One lane batch of BASELINE config 4 (8 array radii, 1024 taps) executed a few times: run under rocprofv3 --kernel-trace --stats.
    python tools/experiments/config4_prof.py <r_lo_cm> <r_hi_cm> [pad_order]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    lo, hi = float(sys.argv[1]) / 100.0, float(sys.argv[2]) / 100.0
    pad = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    from tools.bench_secondary import _grids
    from emagls_amd import Batch, Plan, synth, _lib as L
    azi, zen, maz, mzn = _grids()
    hL, hR = synth.rigid_sphere_hrirs(azi, zen)
    plans = []
    for r in np.linspace(lo, hi, 8):
        p = Plan(L.KIND_EMAGLS2, "real", 4, 48000.0, 1024, hL.shape[0], hL.shape[1], float(r), 32, sim_order_pad=pad)
        p.set_hrir_grid(azi, zen)
        p.set_mic_grid(maz, mzn)
        p.set_hrirs(hL, hR)
        plans.append(p)
    b = Batch(plans)
    for _ in range(6):
        b.execute()
    b.synchronize()
    i = plans[0].info()
    print("lane mode", b.lane_mode(), "sweep form", i.sweep_form, "sim order", i.sim_order, "gram_from", i.gram_from, "hh_end", i.hh_end, "k_cut", i.k_cut)


if __name__ == "__main__":
    main()
