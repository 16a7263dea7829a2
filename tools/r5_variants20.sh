#!/bin/bash
# 20-step variants of the job scheduler's single-chunk mode (forks, Jacobi run length)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
tag=${1:-r05w}
B="--no-cpu-baseline --no-sh-roofline --no-secondary"
run() { name=$1; shift; timeout 900 "$@" > gpurun_out/${tag}_$name.json 2> gpurun_out/${tag}_$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${tag}_$name.json").read().strip().splitlines()[-1]); print("$name", round(d["value"],1), d["ms_per_step"], round(d["roofline"]["avg_launch_us"]), d["roofline"]["designs_per_launch"])
except Exception as e: print("$name FAILED", e); print(open("gpurun_out/${tag}_$name.err").read()[-1500:])
PY
}
for rep in 1 2; do
run base_$rep python bench.py --steps 20 --warmup 5 $B
EMAGLS_JOBS_FORK=4 run fork4_$rep python bench.py --steps 20 --warmup 5 $B
EMAGLS_JACOBI_RUN=1 run jr1_$rep python bench.py --steps 20 --warmup 5 $B
EMAGLS_JACOBI_RUN=2 run jr2_$rep python bench.py --steps 20 --warmup 5 $B
EMAGLS_JOBS_FORK=4 EMAGLS_JACOBI_RUN=1 run fork4_jr1_$rep python bench.py --steps 20 --warmup 5 $B
EMAGLS_JOBS_FORK=2 run fork2_$rep python bench.py --steps 20 --warmup 5 $B
done
