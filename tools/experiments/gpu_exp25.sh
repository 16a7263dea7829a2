#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
for occ in 0 1 2 3; do
  v128=$(EMAGLS_DSPACE_OCC=$occ timeout 600 python bench.py --steps 256 --warmup 64 --no-cpu-baseline --no-sh-roofline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f sweep %.0f us' % (d['value'], d['roofline']['avg_launch_us']))")
  v20=$(EMAGLS_DSPACE_OCC=$occ timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sh-roofline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f' % d['value'])")
  echo "dspace occupancy cap $occ  256: $v128   20: $v20"
done
