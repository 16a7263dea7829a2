#!/bin/bash
# eager launches against hipGraph replay at 20 and 128 steps
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do for ng in 0 1; do
  v20=$(EMAGLS_NO_GRAPH=$ng timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sh-roofline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f' % d['value'])")
  v128=$(EMAGLS_NO_GRAPH=$ng timeout 600 python bench.py --steps 128 --warmup 32 --no-cpu-baseline --no-sh-roofline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f' % d['value'])")
  echo "rep $rep no_graph=$ng  20: $v20  128: $v128"
done; done
