#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
tag=${1:-r05k}
B="--no-cpu-baseline --no-sh-roofline --no-secondary"
timeout 1200 python -m pytest tests/test_gpu_jobs.py tests/test_gpu_stages.py tests/test_gpu_config4.py tests/test_batch_gloo.py -q -x -m gpu > gpurun_out/${tag}_tests_sel.log 2>&1; tail -3 gpurun_out/${tag}_tests_sel.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "fallback or recover or redo or gram_route or lane or batch or hrir_sets or one_call" > gpurun_out/${tag}_parity_sel.log 2>&1; tail -3 gpurun_out/${tag}_parity_sel.log
EMAGLS_JOBS_TRACE=1 timeout 900 python - > gpurun_out/${tag}_config4_runner.json 2> gpurun_out/${tag}_config4_runner.err <<'PY'
import json, sys, os
sys.path.insert(0, os.getcwd())
from tools import bench_secondary as S
print(json.dumps({"runner16": S.config4_rank_share_runner(reps=5)}))
PY
cut -c1-500 gpurun_out/${tag}_config4_runner.json; grep "emagls jobs" gpurun_out/${tag}_config4_runner.err | tail -10
run() { name=$1; shift; timeout 900 "$@" > gpurun_out/${tag}_$name.json 2> gpurun_out/${tag}_$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${tag}_$name.json").read().strip().splitlines()[-1]); print("$name", round(d["value"],1), d["ms_per_step"], round(d["roofline"]["frac"],3), round(d["roofline"]["avg_launch_us"]), d["roofline"]["designs_per_launch"], d.get("one_shot_ms"))
except Exception as e: print("$name FAILED", e); print(open("gpurun_out/${tag}_$name.err").read()[-900:])
PY
}
run b20a python bench.py --steps 20 --warmup 5 $B
run b128 python bench.py --steps 128 --warmup 32 $B
