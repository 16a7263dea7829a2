"""HRIR sets on one geometry (secondary figure of bench.py) with batches of 16 against batches of 32 sets."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools import bench_secondary as S
for nb, pb in ((3, 16), (3, 32), (4, 32), (2, 32)):
    try:
        r = S.config3_hrir_sets(nb, pb, rounds=6)
        print(f"{nb} batches of {pb} sets in flight: {r['filter_sets_per_s']} sets/s, shared {r['geometry_shared']}", flush=True)
    except Exception as e:
        print(f"{nb} x {pb}: {type(e).__name__}: {str(e)[:200]}", flush=True)
