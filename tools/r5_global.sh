#!/bin/bash
# the exchange of the register-resident sweep through write-through granules (EMAGLS_PERSIST_GLOBAL=1: what a design spread over
# several XCDs would need) against workgroup-scope stores in one XCD's L2; then config 4's lane batches on both sweep forms
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
tag=${1:-r05ae}
B="--no-cpu-baseline --no-sh-roofline --no-secondary"
run() { name=$1; shift; timeout 900 "$@" > gpurun_out/${tag}_$name.json 2> gpurun_out/${tag}_$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${tag}_$name.json").read().strip().splitlines()[-1]); print("$name", round(d["value"],1), d["ms_per_step"], round(d["roofline"]["avg_launch_us"]), d["roofline"]["designs_per_launch"])
except Exception as e: print("$name FAILED", e); print(open("gpurun_out/${tag}_$name.err").read()[-800:])
PY
}
for g in 0 1; do
  export EMAGLS_PERSIST_GLOBAL=$g
  run b20_g$g python bench.py --steps 20 --warmup 5 $B
  run b128_g$g python bench.py --steps 128 --warmup 32 $B
  run b16_g$g python bench.py --steps 64 --warmup 16 --slots 1 --batch 16 $B
done
unset EMAGLS_PERSIST_GLOBAL
timeout 1500 python tools/experiments/config4_forms.py 2>&1 | tee gpurun_out/${tag}_config4_forms.log
