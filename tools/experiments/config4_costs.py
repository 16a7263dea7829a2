"""Time of every lane batch of BASELINE config 4's job list (256 radii on 2..10 cm, 1024 taps) alone on the GPU, for the cost model of
emagls_amd.batch.batch_cost.      python tools/experiments/config4_costs.py [max_batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes
import numpy as np
from emagls_amd import Batch, Plan, synth, _lib as L
from emagls_amd.batch import padded_lane_batches, simulation_order
from tools.bench_secondary import _grids
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
azi, zen, maz, mzn = _grids()
hL, hR = synth.rigid_sphere_hrirs(azi, zen)
radii = np.linspace(0.02, 0.10, 256)
so = [simulation_order(4, 48000.0, r, raw=True) for r in radii]
prev = ctypes.c_int(0)
L.check(L.load().emagls_set_batch_max(max(mb, 8), ctypes.byref(prev)))
for idx, pad in padded_lane_batches(so, mb):
    plans = []
    for j in idx:
        p = Plan(L.KIND_EMAGLS2, "real", 4, 48000.0, 1024, hL.shape[0], hL.shape[1], float(radii[j]), 32, sim_order_pad=pad)
        p.set_hrir_grid(azi, zen); p.set_mic_grid(maz, mzn); p.set_hrirs(hL, hR)
        plans.append(p)
    b = Batch(plans)
    for _ in range(3):
        b.execute()
    b.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        b.execute()
    b.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    i = plans[0].info()
    print(f"n {len(idx)} pad {pad} own orders {so[idx[0]]}-{so[idx[-1]]} ms {ms:.2f} sweep_form {i.sweep_form} hh_end {i.hh_end} gram_from {i.gram_from} k_cut {i.k_cut}", flush=True)
    b.close()
    for p in plans:
        p.close()
