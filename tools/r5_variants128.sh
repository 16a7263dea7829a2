#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
tag=${1:-r05x}
B="--no-cpu-baseline --no-sh-roofline --no-secondary"
run() { name=$1; shift; timeout 900 "$@" > gpurun_out/${tag}_$name.json 2> gpurun_out/${tag}_$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${tag}_$name.json").read().strip().splitlines()[-1]); print("$name", round(d["value"],1), d["ms_per_step"], round(d["roofline"]["avg_launch_us"]), d["roofline"]["designs_per_launch"])
except Exception as e: print("$name FAILED", e); print(open("gpurun_out/${tag}_$name.err").read()[-1500:])
PY
}
for rep in 1 2; do
for jr in 4 1 2; do
EMAGLS_JACOBI_RUN=$jr run b128_jr${jr}_$rep python bench.py --steps 128 --warmup 32 $B
EMAGLS_JACOBI_RUN=$jr run b512_jr${jr}_$rep python bench.py --steps 512 --warmup 64 $B
done
done
