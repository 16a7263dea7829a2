"""Config 5 (FromAtf, 16 384 directions x 8 microphones, 2048 taps): subjects of one ATF set per batch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools import bench_secondary as S
import ctypes
from emagls_amd import _lib as L
L.check(L.load().emagls_set_batch_max(16, None))
for n in (8, 12, 16):
    try:
        r = S.config5(reps=4, subjects=n)
        b = r["batch"]
        print(f"{n} subjects per batch: {b['ms_per_batch']} ms per batch, {b['ms_per_subject']} ms per subject, {b['filter_sets_per_s']} sets/s (single subject {r['ms_per_subject']} ms)", flush=True)
    except Exception as e:
        print(f"{n} subjects: {type(e).__name__}: {str(e)[:300]}", flush=True)
