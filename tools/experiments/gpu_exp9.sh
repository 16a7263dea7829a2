#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; tag=${1:-r03_j}
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sixteen or lane_batch or batch_of" > gpurun_out/${tag}_tests.log 2>&1; tail -2 gpurun_out/${tag}_tests.log
for g in 2 1; do
  for st in "20 5" "20 5" "128 32"; do set -- $st
    EMAGLS_BATCH_GROUPS=$g timeout 200 python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-sh-roofline --no-secondary > gpurun_out/${tag}_g${g}_k$1.json 2> gpurun_out/${tag}_g${g}_k$1.err
    echo "groups $g steps $1: $(sed 's/.*"value": \([0-9.]*\).*/\1/' gpurun_out/${tag}_g${g}_k$1.json | cut -c1-12)"
  done
done
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/${tag}_prof20 -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sh-roofline --no-secondary > $R/gpurun_out/${tag}_prof20.log 2>&1
python3 $R/tools/fill_timeline.py $R/gpurun_out/${tag}_prof20 2 > $R/gpurun_out/${tag}_fill20.md 2>&1
rm -rf $R/gpurun_out/${tag}_prof20
grep -n "^-- queue\|sweep_persist\|zero_fill" $R/gpurun_out/${tag}_fill20.md | tail -24
