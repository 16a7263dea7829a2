#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
tag=${1:-r05q}
B="--no-cpu-baseline --no-sh-roofline --no-secondary"
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -rP -k "sweep_variants or residency or other_arrays or batch or lane" > gpurun_out/${tag}_parity_sel.log 2>&1; tail -3 gpurun_out/${tag}_parity_sel.log; grep -h "register-resident form\|variant" gpurun_out/${tag}_parity_sel.log | cut -c1-160
run() { name=$1; shift; timeout 900 "$@" > gpurun_out/${tag}_$name.json 2> gpurun_out/${tag}_$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${tag}_$name.json").read().strip().splitlines()[-1]); print("$name", round(d["value"],1), d["ms_per_step"], round(d["roofline"]["frac"],3), round(d["roofline"]["avg_launch_us"]), d["roofline"]["designs_per_launch"], d["single_design_latency_ms"])
except Exception as e: print("$name FAILED", e); print(open("gpurun_out/${tag}_$name.err").read()[-900:])
PY
}
run b20a python bench.py --steps 20 --warmup 5 $B
run b128 python bench.py --steps 128 --warmup 32 $B
python - <<'PY'
import json, sys, os
sys.path.insert(0, os.getcwd())
from tools import bench_secondary as S
import numpy as np
print(json.dumps({k: v.get("filter_sets_per_s") for k, v in (("r5", S.config4(np.linspace(0.0480, 0.0500, 8))), ("r10", S.config4(np.linspace(0.0980, 0.1000, 8))), ("share", S.config4_rank_share()))}))
PY
