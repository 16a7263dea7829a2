#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
tag=${1:-r05r}
B="--no-cpu-baseline --no-sh-roofline --no-secondary"
run() { name=$1; shift; timeout 900 "$@" > gpurun_out/${tag}_$name.json 2> gpurun_out/${tag}_$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${tag}_$name.json").read().strip().splitlines()[-1]); print("$name", round(d["value"],1), d["ms_per_step"], round(d["roofline"]["frac"],3), round(d["roofline"]["avg_launch_us"]), d["roofline"]["designs_per_launch"])
except Exception as e: print("$name FAILED", e); print(open("gpurun_out/${tag}_$name.err").read()[-900:])
PY
}
run b20_32 python bench.py --steps 20 --warmup 5 $B
run b20_16 python bench.py --steps 20 --warmup 5 --batch 16 $B
run b20_16b python bench.py --steps 20 --warmup 5 --batch 16 $B
EMAGLS_SWEEP_REG=2 run b20_16_reg python bench.py --steps 20 --warmup 5 --batch 16 $B
run b20_12 python bench.py --steps 20 --warmup 5 --batch 12 $B
run b20_10 python bench.py --steps 20 --warmup 5 --batch 10 $B
run b128_s2 python bench.py --steps 128 --warmup 32 --slots 2 $B
run b128_s3 python bench.py --steps 128 --warmup 32 --slots 3 $B
run b128_s4 python bench.py --steps 128 --warmup 32 --slots 4 $B
run b128_s4b python bench.py --steps 128 --warmup 32 --slots 4 $B
