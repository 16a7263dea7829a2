#!/bin/bash
# second session of round 3: the whole gpu suite (no -x), the bench line with secondaries, fill timeline of the driver's run
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; tag=${1:-r03_c}
timeout 2400 python -m pytest tests -m gpu -q -rP > gpurun_out/${tag}_tests_full.log 2>&1
tail -15 gpurun_out/${tag}_tests_full.log > gpurun_out/${tag}_tests.log
grep -h "norm_diff=\|rel = \|^case (\|rel L\|worst rel" gpurun_out/${tag}_tests_full.log > gpurun_out/${tag}_parity.log
grep -n "^FAILED\|^ERROR" gpurun_out/${tag}_tests_full.log | head -20
tail -3 gpurun_out/${tag}_tests.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench20.json 2> gpurun_out/${tag}_bench20.err
cut -c1-200 gpurun_out/${tag}_bench20.json; echo
python - <<'P'
import json,sys
try:
    d=json.load(open("gpurun_out/%s_bench20.json" % sys.argv[1] if len(sys.argv)>1 else "gpurun_out/r03_c_bench20.json"))
    print(json.dumps(d.get("secondary"))[:3000]); print(json.dumps(d.get("flops"))[:1200]); print(d.get("one_shot_ms"))
except Exception as e: print("no json", e)
P
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/${tag}_prof20 -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sh-roofline --no-secondary > $R/gpurun_out/${tag}_prof20.log 2>&1
cd $R
python tools/fill_timeline.py gpurun_out/${tag}_prof20 3 > gpurun_out/${tag}_fill_timeline20.md 2>&1
rm -rf gpurun_out/${tag}_prof20
