#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
tag=${1:-r05e}
B="--no-cpu-baseline --no-sh-roofline --no-secondary"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "sweep_variants or residency_is_decided or config3_full or lane or batch" > gpurun_out/${tag}_parity_sel.log 2>&1; tail -3 gpurun_out/${tag}_parity_sel.log
run() { name=$1; shift; timeout 600 "$@" > gpurun_out/${tag}_$name.json 2> gpurun_out/${tag}_$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${tag}_$name.json").read().strip().splitlines()[-1]); print("$name", round(d["value"],1), d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["avg_launch_us"])
except Exception as e: print("$name FAILED", e); print(open("gpurun_out/${tag}_$name.err").read()[-600:])
PY
}
run b20a python bench.py --steps 20 --warmup 5 $B
run b20b python bench.py --steps 20 --warmup 5 $B
EMAGLS_BATCH_GROUPS=2 run b20_g2 python bench.py --steps 20 --warmup 5 $B
EMAGLS_BATCH_GROUPS=2 run b20_g2b python bench.py --steps 20 --warmup 5 $B
run b128 python bench.py --steps 128 --warmup 32 $B
EMAGLS_BATCH_GROUPS=2 run b128_g2 python bench.py --steps 128 --warmup 32 $B
run b512 python bench.py --steps 512 --warmup 64 $B
EMAGLS_BATCH_GROUPS=2 run b512_g2 python bench.py --steps 512 --warmup 64 $B
run b512_s6 python bench.py --steps 512 --warmup 64 --slots 6 $B
run b512_s3 python bench.py --steps 512 --warmup 64 --slots 3 $B
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/${tag}_prof20 -o bench -- python3 $R/bench.py --steps 20 --warmup 5 $B > $R/gpurun_out/${tag}_prof20.log 2>&1
cd $R
python tools/fill_timeline.py gpurun_out/${tag}_prof20 1 > gpurun_out/${tag}_fill_timeline20.md 2>&1; head -80 gpurun_out/${tag}_fill_timeline20.md
rm -rf gpurun_out/${tag}_prof20
