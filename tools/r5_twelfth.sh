#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
tag=${1:-r05n}
timeout 900 python -m pytest tests/test_batch_gloo.py tests/test_gpu_jobs.py -q -x -m gpu -rP > gpurun_out/${tag}_tests_sel.log 2>&1; tail -3 gpurun_out/${tag}_tests_sel.log; grep -h "worst rel" gpurun_out/${tag}_tests_sel.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "residency or batch or lane or atf" > gpurun_out/${tag}_parity_sel.log 2>&1; tail -3 gpurun_out/${tag}_parity_sel.log
timeout 2400 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench20_full.json 2> gpurun_out/${tag}_bench20_full.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/r05n_bench20_full.json").read().strip().splitlines()[-1])
print("value", round(d["value"],1), "frac", round(d["roofline"]["frac"],3), "parity", d["parity"]["rel_complex_error"], d["parity"]["max_abs_db_diff"], "cpu", d["cpu_baseline"]["value"])
s=d["secondary"]
for k,v in s.items():
    if isinstance(v, dict):
        print(k, {kk: vv for kk, vv in v.items() if kk in ("filter_sets_per_s","ms_per_share","ms_per_execute","first_call_s","new_radii_s","resident_s","filter_sets_per_s_new_radii","error","ms","designs")})
PY
tail -3 gpurun_out/${tag}_bench20_full.err
