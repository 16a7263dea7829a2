#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
tag=${1:-r05j}
B="--no-cpu-baseline --no-sh-roofline --no-secondary"
run() { name=$1; shift; timeout 900 "$@" > gpurun_out/${tag}_$name.json 2> gpurun_out/${tag}_$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${tag}_$name.json").read().strip().splitlines()[-1]); print("$name", round(d["value"],1), d["ms_per_step"], round(d["roofline"]["frac"],3), round(d["roofline"]["avg_launch_us"]), d["roofline"]["designs_per_launch"])
except Exception as e: print("$name FAILED", e); print(open("gpurun_out/${tag}_$name.err").read()[-900:])
PY
}
run b512 python bench.py --steps 512 --warmup 64 $B
run b512_s6 python bench.py --steps 512 --warmup 64 --slots 6 $B
run b512_s3 python bench.py --steps 512 --warmup 64 --slots 3 $B
run b128 python bench.py --steps 128 --warmup 32 $B
EMAGLS_JOBS_TRACE=1 timeout 900 python - > gpurun_out/${tag}_config4_runner.json 2> gpurun_out/${tag}_config4_runner.err <<'PY'
import json, sys, os
sys.path.insert(0, os.getcwd())
from tools import bench_secondary as S
print(json.dumps({"runner16": S.config4_rank_share_runner(reps=2)}))
PY
cut -c1-500 gpurun_out/${tag}_config4_runner.json; grep "emagls jobs" gpurun_out/${tag}_config4_runner.err | tail -10
