#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
tag=${1:-r05o}
EMAGLS_JOBS_TRACE=1 timeout 900 python - > gpurun_out/${tag}_config4_runner.json 2> gpurun_out/${tag}_config4_runner.err <<'PY'
import json, sys, os
sys.path.insert(0, os.getcwd())
from tools import bench_secondary as S
print(json.dumps({"runner16": S.config4_rank_share_runner(reps=2)}))
PY
cut -c1-600 gpurun_out/${tag}_config4_runner.json; grep "emagls" gpurun_out/${tag}_config4_runner.err | sed -n 20,75p
