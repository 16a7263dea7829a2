#!/bin/bash
# from how many designs per launch the register-resident form pays: config 4's rank share (two batches of 14) and config 3 job lists
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
tag=${1:-r05ag}
timeout 600 python -m pytest tests/test_gpu_stages.py tests/test_gpu_parity.py -q -x -m gpu -k "gram_tile" > gpurun_out/${tag}_tests_sel.log 2>&1; tail -2 gpurun_out/${tag}_tests_sel.log
for m in 9 17; do
export EMAGLS_SWEEP_REG_MIN=$m
timeout 900 python - <<'PY'
import json, sys, os
sys.path.insert(0, os.getcwd())
from tools import bench_secondary as S
a = S.config4_rank_share()
b = S.config4_rank_share_runner(reps=4)
print("REG_MIN", os.environ["EMAGLS_SWEEP_REG_MIN"], "share", a["filter_sets_per_s"], a["ms_per_batch_alone"], "runner", b["filter_sets_per_s"], b["resident_s"])
PY
B="--no-cpu-baseline --no-sh-roofline --no-secondary"
for bs in 12 16; do
timeout 600 python bench.py --steps 192 --warmup 48 --slots 4 --batch $bs $B > gpurun_out/${tag}_b_$bs_$m.json 2>/dev/null; python - <<PY
import json
d=json.loads(open("gpurun_out/${tag}_b_$bs_$m.json").read().strip().splitlines()[-1]); print("REG_MIN $m batch $bs x 4 in flight:", round(d["value"],1), round(d["roofline"]["avg_launch_us"]), d["roofline"]["kernel"])
PY
done
done
