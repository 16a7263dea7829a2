#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 1500 python -m pytest tests -m gpu -q -x -k "stage or config3 or config4 or config1 or from_atf or ema or orders or ill or lane_batch" 2>&1 | tail -3
for rep in 1 2 3; do
  v20=$(timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sh-roofline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f sweep %.0f us' % (d['value'], d['roofline']['avg_launch_us']))")
  v128=$(timeout 600 python bench.py --steps 128 --warmup 32 --no-cpu-baseline --no-sh-roofline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f sweep %.0f us' % (d['value'], d['roofline']['avg_launch_us']))")
  echo "rep $rep  20: $v20   128: $v128"
done
