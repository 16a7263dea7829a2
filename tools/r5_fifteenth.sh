#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
tag=${1:-r05s}
B="--no-cpu-baseline --no-sh-roofline --no-secondary"
run() { name=$1; shift; timeout 900 "$@" > gpurun_out/${tag}_$name.json 2> gpurun_out/${tag}_$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${tag}_$name.json").read().strip().splitlines()[-1]); print("$name", round(d["value"],1), d["ms_per_step"], round(d["roofline"]["frac"],3), round(d["roofline"]["avg_launch_us"]), d["roofline"]["designs_per_launch"])
except Exception as e: print("$name FAILED", e); print(open("gpurun_out/${tag}_$name.err").read()[-900:])
PY
}
run b20 python bench.py --steps 20 --warmup 5 $B
EMAGLS_BATCH_GROUPS=1 run b20_g1 python bench.py --steps 20 --warmup 5 $B
EMAGLS_BATCH_GROUPS=1 EMAGLS_JOBS_FORK=3 run b20_g1f3 python bench.py --steps 20 --warmup 5 $B
EMAGLS_BATCH_GROUPS=1 EMAGLS_JOBS_FORK=4 run b20_g1f4 python bench.py --steps 20 --warmup 5 $B
EMAGLS_BATCH_GROUPS=3 run b20_g3 python bench.py --steps 20 --warmup 5 $B
EMAGLS_STAGGER=0 run b20_nostagger python bench.py --steps 20 --warmup 5 $B
