"""Two lane batches of n designs each: sequential versus overlapped execution (does a 9..16-design sweep become resident
while another batch's kernels are running?)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from emagls_amd import Batch, Plan, synth, _lib as L

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ref_fixtures.npz"))
azi, zen, maz, mzn = g["grid/hrirGridAziRad"], g["grid/hrirGridZenRad"], g["grid/micGridAziRad"], g["grid/micGridZenRad"]
hL, hR = synth.rigid_sphere_hrirs(azi, zen)
def mk():
    plans = []
    for j in range(n):
        p = Plan(L.KIND_EMAGLS, "complex", 4, 48000.0, 512, 128, 2702, 0.042, 32)
        p.set_hrir_grid(azi, zen); p.set_mic_grid(maz, mzn); p.set_hrirs(hL, hR)
        plans.append(p)
    return Batch(plans), plans
A, pa = mk(); B, pb = mk()
for b in (A, B):
    for _ in range(3):
        b.execute(); b.synchronize()
def launches(): return pa[0].info().num_sweep_launches, pb[0].info().num_sweep_launches
print("after sequential warm-up: sweep launches per design (1 = persistent):", launches(), flush=True)
for rep in range(3):
    t0 = time.perf_counter(); A.execute(); A.synchronize(); B.execute(); B.synchronize(); t1 = time.perf_counter()
    print(f"sequential A;B: {(t1 - t0) * 1e3:.2f} ms", launches(), flush=True)
for rep in range(4):
    t0 = time.perf_counter(); A.execute(); B.execute(); A.synchronize(); B.synchronize(); t1 = time.perf_counter()
    A.get_filters(); B.get_filters()
    print(f"overlapped A||B: {(t1 - t0) * 1e3:.2f} ms", launches(), flush=True)
# bench-like flow: three executes back to back per batch, then a sliding window over both batches
for b in (A, B):
    for _ in range(3):
        b.execute()
    b.synchronize()
print("after back-to-back triple executes:", launches(), flush=True)
t0 = time.perf_counter()
inflight = []
units = [A, B]
for it in range(12):
    if len(inflight) == 2:
        u = inflight.pop(0); u.get_filters()
    u = units[it % 2]; u.execute(); inflight.append(u)
for u in inflight: u.get_filters()
dt = time.perf_counter() - t0
print(f"sliding window, 12 batches of {n}: {dt * 1e3:.1f} ms = {12 * n / dt:.0f} sets/s", launches(), flush=True)
