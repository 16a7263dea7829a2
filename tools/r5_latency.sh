#!/bin/bash
# the register-resident sweep with its two global-memory round trips off the critical path: parity, stamps, bench
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
tag=${1:-r05al}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_jobs.py tests/test_gpu_config4.py -q -x -m gpu -k "sweep_variants or other_arrays or residency or spread_over or jobs or job_list or one_ranks_share or config3" > gpurun_out/${tag}_tests_sel.log 2>&1; tail -2 gpurun_out/${tag}_tests_sel.log
for n in 20 32 16; do timeout 300 python tools/sweep_timing.py $n 2>&1 | grep -v amdgpu.ids | head -8 | tee gpurun_out/${tag}_sweep_timing_$n.log; done
B="--no-cpu-baseline --no-sh-roofline --no-secondary"
run() { name=$1; shift; timeout 900 "$@" > gpurun_out/${tag}_$name.json 2> gpurun_out/${tag}_$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${tag}_$name.json").read().strip().splitlines()[-1]); print("$name", round(d["value"],1), d["ms_per_step"], round(d["roofline"]["avg_launch_us"]), d["roofline"]["designs_per_launch"], round(d["roofline"]["frac"],3))
except Exception as e: print("$name FAILED", e); print(open("gpurun_out/${tag}_$name.err").read()[-800:])
PY
}
run b20a python bench.py --steps 20 --warmup 5 $B
run b20b python bench.py --steps 20 --warmup 5 $B
run b20c python bench.py --steps 20 --warmup 5 $B
run b128a python bench.py --steps 128 --warmup 32 $B
run b128b python bench.py --steps 128 --warmup 32 $B
run b512 python bench.py --steps 512 --warmup 64 $B
