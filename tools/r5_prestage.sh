#!/bin/bash
# round 5, stages before the sweep: parity of the routes the rewritten kernels serve, then the bench figures and a kernel profile
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
tag=${1:-r05u}
B="--no-cpu-baseline --no-sh-roofline --no-secondary"
[ -n "$SKIP_TESTS" ] || { timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stages.py tests/test_gpu_config4.py -q -x -m gpu -k "config3 or thin or stage or large_radius or config4_shape or config4_full or other_arrays or simulation_order or ema_in or magls_filters" > gpurun_out/${tag}_tests_sel.log 2>&1; tail -3 gpurun_out/${tag}_tests_sel.log; }
run() { name=$1; shift; timeout 900 "$@" > gpurun_out/${tag}_$name.json 2> gpurun_out/${tag}_$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${tag}_$name.json").read().strip().splitlines()[-1]); print("$name", round(d["value"],1), d["ms_per_step"], d["roofline"]["kernel"], round(d["roofline"]["frac"],3), round(d["roofline"]["avg_launch_us"]), d["roofline"]["designs_per_launch"])
except Exception as e: print("$name FAILED", e); print(open("gpurun_out/${tag}_$name.err").read()[-1500:])
PY
}
run b20a python bench.py --steps 20 --warmup 5 $B
run b20b python bench.py --steps 20 --warmup 5 $B
run b20c python bench.py --steps 20 --warmup 5 $B
run b128 python bench.py --steps 128 --warmup 32 $B
run b512 python bench.py --steps 512 --warmup 64 $B
cd /tmp && export TMPDIR=/tmp
# kernel times: one 32-design chunk at a time (two lane groups of 16), and the driver's 20-step command
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof16 -o p -- python3 $R/bench.py --steps 128 --warmup 0 --slots 1 --batch 32 $B > $R/gpurun_out/${tag}_prof16.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof20 -o p -- python3 $R/bench.py --steps 20 --warmup 5 $B > $R/gpurun_out/${tag}_prof20.log 2>&1
cd $R
python tools/kernel_avgs.py gpurun_out/${tag}_prof16 16 > gpurun_out/${tag}_kernels16.md 2>&1
python tools/kernel_avgs.py gpurun_out/${tag}_prof20 20 > gpurun_out/${tag}_kernels20.md 2>&1
python tools/fill_timeline.py gpurun_out/${tag}_prof20 2 > gpurun_out/${tag}_timeline20.md 2>&1
rm -rf gpurun_out/${tag}_prof16 gpurun_out/${tag}_prof20
head -22 gpurun_out/${tag}_kernels16.md | cut -c1-150
