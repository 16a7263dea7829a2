#!/bin/bash
# one scatter launch for a lane batch's results instead of two copies per design
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "lane_batch or sixteen or one_geometry or config4 or bench" 2>&1 | tail -2
for rep in 1 2 3; do
  v20=$(timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sh-roofline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f' % d['value'])")
  v128=$(timeout 600 python bench.py --steps 256 --warmup 64 --no-cpu-baseline --no-sh-roofline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f' % d['value'])")
  echo "rep $rep  20: $v20   256: $v128"
done
