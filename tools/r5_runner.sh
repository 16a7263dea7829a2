#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
tag=${1:-r05aj}
timeout 600 python -m pytest tests/test_gpu_jobs.py tests/test_gpu_config4.py -q -x -m gpu -k "jobs or job_list or radius_sweep" > gpurun_out/${tag}_tests_sel.log 2>&1; tail -2 gpurun_out/${tag}_tests_sel.log
timeout 900 python - <<'PY' 2>&1 | tee gpurun_out/${tag}_runner.log
import json, sys, os
sys.path.insert(0, os.getcwd())
from tools import bench_secondary as S
for rep in range(2):
    b = S.config4_rank_share_runner(reps=5)
    print("runner", b["filter_sets_per_s"], b["resident_s"], "new radii", b["filter_sets_per_s_new_radii"], b["new_radii_s"], "first", b["first_call_s"])
PY
