#!/bin/bash
# register-resident sweep: designs spread over all XCDs (EMAGLS_REG_SPREAD default: when it needs fewer waves per workgroup) against
# the XCD-local layout (=0); parity of the job-list tests first
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
tag=${1:-r05af}
B="--no-cpu-baseline --no-sh-roofline --no-secondary"
timeout 900 python -m pytest tests/test_gpu_jobs.py tests/test_gpu_parity.py -q -x -m gpu -k "jobs or job_list or residency or other_arrays or gram_tile or config5_shape or wide_array_at" > gpurun_out/${tag}_tests_sel.log 2>&1; tail -3 gpurun_out/${tag}_tests_sel.log
run() { name=$1; shift; timeout 900 "$@" > gpurun_out/${tag}_$name.json 2> gpurun_out/${tag}_$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${tag}_$name.json").read().strip().splitlines()[-1]); print("$name", round(d["value"],1), d["ms_per_step"], round(d["roofline"]["avg_launch_us"]), d["roofline"]["designs_per_launch"], round(d["roofline"]["frac"],3), d.get("parity",{}).get("rel_complex_error"))
except Exception as e: print("$name FAILED", e); print(open("gpurun_out/${tag}_$name.err").read()[-800:])
PY
}
for rep in 1 2 3; do
  for m in 2 0 1; do
    export EMAGLS_REG_SPREAD=$m
    run b20_s${m}_$rep python bench.py --steps 20 --warmup 5 $B
  done
done
for m in 2 1; do
  export EMAGLS_REG_SPREAD=$m
  run b128_s$m python bench.py --steps 128 --warmup 32 $B
  run b10_s$m python bench.py --steps 40 --warmup 10 --slots 1 --batch 10 $B
done
unset EMAGLS_REG_SPREAD
timeout 600 python bench.py --steps 20 --warmup 5 --no-secondary --no-sh-roofline > gpurun_out/${tag}_b20_full.json 2> gpurun_out/${tag}_b20_full.err; python - <<PY
import json
d=json.loads(open("gpurun_out/${tag}_b20_full.json").read().strip().splitlines()[-1]); print("full20", round(d["value"],1), d["parity"])
PY
timeout 1200 python tools/experiments/config4_forms.py 2>&1 | tee gpurun_out/${tag}_config4_forms.log
