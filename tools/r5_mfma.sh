#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
tag=${1:-r05y}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/experiments/mfma_peak.hip -o /tmp/mfma_peak 2>/dev/null && timeout 300 /tmp/mfma_peak | tee gpurun_out/${tag}_mfma_peak.log
B="--no-cpu-baseline --no-sh-roofline --no-secondary"
run() { name=$1; shift; timeout 900 "$@" > gpurun_out/${tag}_$name.json 2> gpurun_out/${tag}_$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${tag}_$name.json").read().strip().splitlines()[-1]); print("$name", round(d["value"],1), d["ms_per_step"], round(d["roofline"]["avg_launch_us"]), d["roofline"]["designs_per_launch"])
except Exception as e: print("$name FAILED", e); print(open("gpurun_out/${tag}_$name.err").read()[-1500:])
PY
}
run b20a python bench.py --steps 20 --warmup 5 $B
run b20b python bench.py --steps 20 --warmup 5 $B
run b128 python bench.py --steps 128 --warmup 32 $B
run b512 python bench.py --steps 512 --warmup 64 $B
