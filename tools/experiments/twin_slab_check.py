"""Does a 16-design launch of the slab form (EMAGLS_SWEEP_REG=0: two workgroups per CU) still fit after the ring grew to 96 orders?"""
import os, sys, time, ctypes
os.environ["EMAGLS_SWEEP_REG"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from emagls_amd import Batch, Plan, _lib as L
lib = L.load()
L.check(lib.emagls_set_batch_max(16, None))
plans = []
for j in range(16):
    azi, zen, maz, mzn, hL, hR = bench.load_inputs(seed_offset=j)
    p = Plan(L.KIND_EMAGLS, "complex", 4, 48000.0, 512, hL.shape[0], hL.shape[1], 0.042, 32)
    p.set_hrir_grid(azi, zen); p.set_mic_grid(maz, mzn); p.set_hrirs(hL, hR)
    plans.append(p)
b = Batch(plans)
for _ in range(3):
    b.execute()
b.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    b.execute()
b.synchronize()
print("16-design batch, slab form: sweep_form", plans[0].info().sweep_form, "sweep launches", plans[0].info().num_sweep_launches, "%.2f ms per batch" % ((time.perf_counter() - t0) / 5 * 1e3))
