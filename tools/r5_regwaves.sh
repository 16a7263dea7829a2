#!/bin/bash
# register-resident sweep: waves per workgroup forced (EMAGLS_REG_WAVES) against the default choice, 32-design chunks
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
tag=${1:-r05ad}
B="--no-cpu-baseline --no-sh-roofline --no-secondary"
run() { name=$1; shift; timeout 900 "$@" > gpurun_out/${tag}_$name.json 2> gpurun_out/${tag}_$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${tag}_$name.json").read().strip().splitlines()[-1]); print("$name", round(d["value"],1), d["ms_per_step"], round(d["roofline"]["avg_launch_us"]), d["roofline"]["designs_per_launch"])
except Exception as e: print("$name FAILED", e); print(open("gpurun_out/${tag}_$name.err").read()[-800:])
PY
}
for w in 0 6 8 4; do
  if [ $w = 0 ]; then unset EMAGLS_REG_WAVES; else export EMAGLS_REG_WAVES=$w; fi
  run b512_w$w python bench.py --steps 512 --warmup 64 $B
  run b128_w$w python bench.py --steps 128 --warmup 32 $B
  run b20_w$w python bench.py --steps 20 --warmup 5 $B
done
