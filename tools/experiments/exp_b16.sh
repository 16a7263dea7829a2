#!/bin/bash
# batches of more than 8 designs with several batches in flight: reproduces the stall described in DESIGN.md section 5
# (expect missing / tiny values for some configurations: the sweep falls back to launch-per-bin after its spin limit)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
val() { python -c "import sys,json; [print(round(json.loads(l)['value'],1), end=' ') for l in sys.stdin if l.startswith('{')]"; }
export EMAGLS_BATCH_MAX=16
for cfg in "2 16" "3 16" "4 16" "3 12" "4 10"; do
  set -- $cfg
  echo "== slots $1 batch $2"
  for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --slots $1 --batch $2 --no-cpu-baseline --no-sh-roofline --no-secondary 2>&1 | val; done; echo
  timeout 300 python bench.py --steps 128 --warmup 32 --slots $1 --batch $2 --no-cpu-baseline --no-sh-roofline --no-secondary 2>&1 | val; echo
done
