#!/bin/bash
# the 20-design run of bench.py with the partial batch and one full batch re-divided (EMAGLS_BENCH_SPLIT: designs in the first batch)
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2 3; do
for sp in 0 8 10 12; do
  v=$(EMAGLS_BENCH_SPLIT=$sp timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sh-roofline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['config'].get('timed_schedule'))")
  echo "split=$sp: $v"
done; done
