#!/bin/bash
# chunk sizes of the job scheduler with the spread layout: 23 designs = 253 workgroups of 8 waves against 32 = 256 of 12
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
tag=${1:-r05ah}
B="--no-cpu-baseline --no-sh-roofline --no-secondary"
run() { name=$1; shift; timeout 900 "$@" > gpurun_out/${tag}_$name.json 2> gpurun_out/${tag}_$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${tag}_$name.json").read().strip().splitlines()[-1]); print("$name", round(d["value"],1), d["ms_per_step"], round(d["roofline"]["avg_launch_us"]), d["roofline"]["designs_per_launch"], round(d["roofline"]["frac"],3))
except Exception as e: print("$name FAILED", e); print(open("gpurun_out/${tag}_$name.err").read()[-800:])
PY
}
for bs in 32 23 20 16; do
  for sl in 4 6; do
    run b512_b${bs}_s$sl python bench.py --steps $((bs*16)) --warmup $((bs*2)) --slots $sl --batch $bs $B
  done
done
run b512_b23_s3 python bench.py --steps 368 --warmup 46 --slots 3 --batch 23 $B
run b512_b23_s8 python bench.py --steps 368 --warmup 46 --slots 8 --batch 23 $B
