#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for sl in 3 4 5 6 8; do
  v128=$(timeout 600 python bench.py --steps 256 --warmup 64 --slots $sl --no-cpu-baseline --no-sh-roofline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f sweep %.0f us' % (d['value'], d['roofline']['avg_launch_us']))")
  v20=$(timeout 600 python bench.py --steps 20 --warmup 5 --slots $sl --no-cpu-baseline --no-sh-roofline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f' % d['value'])")
  echo "slots $sl  256: $v128   20: $v20"
done
