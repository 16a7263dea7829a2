#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
tag=${1:-r05t}
B="--no-cpu-baseline --no-sh-roofline --no-secondary"
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_jobs.py -q -x -k "sweep_variants or other_arrays or batch or lane or job_list or config3_full" > gpurun_out/${tag}_parity_sel.log 2>&1; tail -3 gpurun_out/${tag}_parity_sel.log
for n in 20 32; do timeout 300 python tools/sweep_timing.py $n 2>&1 | grep "bin period\|sweep span" | cut -c1-140; done
run() { name=$1; shift; timeout 900 "$@" > gpurun_out/${tag}_$name.json 2> gpurun_out/${tag}_$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${tag}_$name.json").read().strip().splitlines()[-1]); print("$name", round(d["value"],1), d["ms_per_step"], round(d["roofline"]["frac"],3), round(d["roofline"]["avg_launch_us"]), d["roofline"]["designs_per_launch"])
except Exception as e: print("$name FAILED", e); print(open("gpurun_out/${tag}_$name.err").read()[-900:])
PY
}
run b20 python bench.py --steps 20 --warmup 5 $B
run b20b python bench.py --steps 20 --warmup 5 $B
EMAGLS_BATCH_GROUPS=1 EMAGLS_JOBS_FORK=3 run b20_g1f3 python bench.py --steps 20 --warmup 5 $B
run b128 python bench.py --steps 128 --warmup 32 $B
EMAGLS_BATCH_GROUPS=1 EMAGLS_JOBS_FORK=3 run b128_g1f3 python bench.py --steps 128 --warmup 32 $B
run b512 python bench.py --steps 512 --warmup 64 $B
EMAGLS_BATCH_GROUPS=1 EMAGLS_JOBS_FORK=3 run b512_g1f3 python bench.py --steps 512 --warmup 64 $B
