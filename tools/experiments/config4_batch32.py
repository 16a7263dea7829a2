"""Config 4, one rank's share of the 256 radii: two lane batches of about 14 designs (max_batch 16) against one of about 28 (max_batch 32)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tools import bench_secondary as S
for mb in (16, 32):
    try:
        a = S.config4_rank_share(max_batch=mb)
        print(f"max_batch {mb}: harness {a['filter_sets_per_s']} sets/s, batches {a['lane_batches']}, pad orders {a['pad_orders']}, spread {a['rank_load_spread']}, ms {a['ms_per_share']}", flush=True)
        b = S.config4_rank_share_runner(reps=4, max_batch=mb)
        print(f"max_batch {mb}: runner {b['filter_sets_per_s']} sets/s resident {b['resident_s']}", flush=True)
    except Exception as e:
        print(f"max_batch {mb}: {type(e).__name__}: {str(e)[:300]}", flush=True)
