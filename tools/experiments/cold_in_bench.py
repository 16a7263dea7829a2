"""The first-call figures of tools/bench_secondary.py in the order bench.py runs them (after other workloads of the same process), with
the scheduler's trace on stderr:  EMAGLS_JOBS_TRACE=1 python tools/experiments/cold_in_bench.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from tools import bench_secondary as B
out = {}
out["config4_r5cm"] = B.config4(np.linspace(0.0480, 0.0500, 8), roofline_key="config4_r5cm").get("filter_sets_per_s")
out["config4_r10cm"] = B.config4(np.linspace(0.0980, 0.1000, 8), roofline_key="config4_r10cm").get("filter_sets_per_s")
out["rank_share"] = B.config4_rank_share().get("filter_sets_per_s")
sys.stderr.write("==== runner\n")
out["runner"] = B.config4_rank_share_runner()
sys.stderr.write("==== host arrays\n")
out["host_arrays"] = B.config3_host_arrays()
print(json.dumps(out))
