#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
tag=${1:-r05ak}
B="--no-cpu-baseline --no-sh-roofline --no-secondary"
run() { name=$1; shift; timeout 900 "$@" > gpurun_out/${tag}_$name.json 2> gpurun_out/${tag}_$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${tag}_$name.json").read().strip().splitlines()[-1]); print("$name", round(d["value"],1), d["ms_per_step"], round(d["roofline"]["avg_launch_us"]), d["roofline"]["designs_per_launch"], d["config"]["timed_schedule"])
except Exception as e: print("$name FAILED", e); print(open("gpurun_out/${tag}_$name.err").read()[-800:])
PY
}
for rep in 1 2; do
run b20_32_$rep python bench.py --steps 20 --warmup 5 $B
run b20_10_$rep python bench.py --steps 20 --warmup 5 --batch 10 $B
run b20_11_$rep python bench.py --steps 20 --warmup 5 --batch 11 $B
run b20_12_$rep python bench.py --steps 20 --warmup 5 --batch 12 $B
done
