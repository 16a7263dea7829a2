#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
tag=${1:-r05m}
timeout 900 python -m pytest tests/test_batch_gloo.py tests/test_gpu_jobs.py -q -x -m gpu > gpurun_out/${tag}_tests_sel.log 2>&1; tail -3 gpurun_out/${tag}_tests_sel.log
EMAGLS_JOBS_TRACE=1 timeout 900 python - > gpurun_out/${tag}_config4_runner.json 2> gpurun_out/${tag}_config4_runner.err <<'PY'
import json, sys, os
sys.path.insert(0, os.getcwd())
from tools import bench_secondary as S
print(json.dumps({"runner16": S.config4_rank_share_runner(reps=5)}))
PY
cut -c1-500 gpurun_out/${tag}_config4_runner.json; grep "emagls jobs" gpurun_out/${tag}_config4_runner.err | tail -24
