#!/bin/bash
# granule stores of the register-resident sweep: workgroup scope inside one XCD (default there) against agent scope (EMAGLS_PERSIST_GLOBAL=1)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
tag=${1:-r05an}
B="--no-cpu-baseline --no-sh-roofline --no-secondary"
for g in 0 1; do
  export EMAGLS_PERSIST_GLOBAL=$g
  echo "EMAGLS_PERSIST_GLOBAL=$g"
  timeout 300 python tools/sweep_timing.py 32 2>&1 | grep -v amdgpu.ids | head -8 | tee gpurun_out/${tag}_timing32_g$g.log
done
run() { name=$1; shift; timeout 900 "$@" > gpurun_out/${tag}_$name.json 2> gpurun_out/${tag}_$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${tag}_$name.json").read().strip().splitlines()[-1]); print("$name", round(d["value"],1), d["ms_per_step"], round(d["roofline"]["avg_launch_us"]), d["roofline"]["designs_per_launch"])
except Exception as e: print("$name FAILED", e); print(open("gpurun_out/${tag}_$name.err").read()[-800:])
PY
}
for rep in 1 2 3; do
for g in 0 1; do
  export EMAGLS_PERSIST_GLOBAL=$g
  run b512_g${g}_$rep python bench.py --steps 512 --warmup 64 $B
  run b128_g${g}_$rep python bench.py --steps 128 --warmup 32 $B
done
done
