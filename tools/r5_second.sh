#!/bin/bash
# round 5: per-bin timing, bench matrix (waves / sliding window), a kernel trace of the 128-step run
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
tag=${1:-r05b}
B="--no-cpu-baseline --no-sh-roofline --no-secondary"
for n in 1 16; do timeout 300 python tools/sweep_timing.py $n > gpurun_out/${tag}_timing_$n.log 2>&1; cut -c1-160 gpurun_out/${tag}_timing_$n.log; done
run() { name=$1; shift; timeout 600 "$@" > gpurun_out/${tag}_$name.json 2> gpurun_out/${tag}_$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${tag}_$name.json").read().strip().splitlines()[-1]); print("$name", round(d["value"],1), d["ms_per_step"])
except Exception as e: print("$name FAILED", e)
PY
}
run b20a python bench.py --steps 20 --warmup 5 $B
run b20b python bench.py --steps 20 --warmup 5 $B
run b128w python bench.py --steps 128 --warmup 32 $B
EMAGLS_BENCH_WAVES=0 run b128s python bench.py --steps 128 --warmup 32 $B
EMAGLS_BENCH_WAVES=0 run b128s8 python bench.py --steps 128 --warmup 32 --slots 8 $B
run b512w python bench.py --steps 512 --warmup 64 $B
EMAGLS_BENCH_WAVES=0 run b512s python bench.py --steps 512 --warmup 64 $B
EMAGLS_SWEEP_SERIAL=1 run b128serial python bench.py --steps 128 --warmup 32 $B
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/${tag}_prof128 -o bench -- python3 $R/bench.py --steps 128 --warmup 0 $B > $R/gpurun_out/${tag}_prof128.log 2>&1
cd $R
python tools/timeline.py gpurun_out/${tag}_prof128 10 > gpurun_out/${tag}_timeline128.md 2>&1; head -40 gpurun_out/${tag}_timeline128.md
python tools/sweep_launches.py gpurun_out/${tag}_prof128 gpurun_out/${tag}_prof128.log > gpurun_out/${tag}_sweeps128.md 2>&1; head -20 gpurun_out/${tag}_sweeps128.md
rm -rf gpurun_out/${tag}_prof128
