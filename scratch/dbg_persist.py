"""Persistent sweep vs launch-per-bin sweep: same filters, timing (config 3 and a small case)."""
import sys, os, time, numpy as np, faulthandler
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from emagls_amd import Plan, Batch, _lib as L, synth


def mkplan(nd, taps, flen, persist, radius=0.042):
    os.environ["EMAGLS_SWEEP_PERSIST"] = "1" if persist else "0"
    azi, zen = synth.fibonacci_grid(nd)
    maz, mzn = synth.em32_grid()
    hL, hR = synth.rigid_sphere_hrirs(azi, zen, taps=taps)
    p = Plan(L.KIND_EMAGLS, 'complex', 4, 48000.0, flen, taps, nd, radius, 32)
    p.set_hrir_grid(azi, zen); p.set_mic_grid(maz, mzn); p.set_hrirs(hL, hR)
    return p


def timeit(fn, sync, n=20):
    fn(); sync()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    sync()
    return (time.perf_counter() - t0) / n * 1e3


for nd, taps, flen in ((900, 64, 64), (2702, 128, 512)):
    ref = mkplan(nd, taps, flen, False)
    ref.execute(); ref.synchronize(); wl0, wr0 = ref.get_filters()
    per = mkplan(nd, taps, flen, True)
    for it in range(3):
        per.execute(); per.synchronize(); wl, wr = per.get_filters()
        err = max(np.abs(wl - wl0).max(), np.abs(wr - wr0).max()) / np.abs(wl0).max()
        print(f"nd={nd} flen={flen} iter {it}: persist vs per-bin rel err {err:.2e}", flush=True)
    t_ref = timeit(ref.execute, ref.synchronize)
    t_per = timeit(per.execute, per.synchronize)
    print(f"  single design: per-bin {t_ref:.3f} ms, persistent {t_per:.3f} ms", flush=True)
    per.set_profiling(1); per.execute(); per.synchronize()
    st = per.stage_times()
    print("  stages:", {k: round(v, 3) for k, v in st}, flush=True)
    if nd == 2702:
        plans = [mkplan(nd, taps, flen, True, 0.042 - 0.0005 * j) for j in range(8)]
        b = Batch(plans)
        b.execute(); b.synchronize()
        w8 = plans[0].get_filters()
        err = max(np.abs(w8[0] - wl0).max(), np.abs(w8[1] - wr0).max()) / np.abs(wl0).max()
        print(f"  batch plan 0 vs single rel err {err:.2e}", flush=True)
        t_b = timeit(b.execute, b.synchronize, 10)
        print(f"  batch of 8: {t_b:.3f} ms -> {8e3 / t_b:.1f} sets/s", flush=True)
