#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for rep in 1 2 3; do for tw in 1 0; do
  v128=$(EMAGLS_SWEEP_TWIN=$tw timeout 600 python bench.py --steps 128 --warmup 32 --no-cpu-baseline --no-sh-roofline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f sweep %.0f us' % (d['value'], d['roofline']['avg_launch_us']))")
  v20=$(EMAGLS_SWEEP_TWIN=$tw timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sh-roofline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f sweep %.0f us' % (d['value'], d['roofline']['avg_launch_us']))")
  echo "rep $rep twin=$tw  128: $v128   20: $v20"
done; done
