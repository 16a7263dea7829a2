#!/bin/bash
# throughput against the number of batches in flight (and the batch size)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
val() { python -c "import sys,json; [print(round(json.loads(l)['value'],1), end=' ') for l in sys.stdin if l.startswith('{')]"; }
for cfg in "3 8" "4 8" "5 8" "6 8" "8 8" "6 4" "4 6"; do
  set -- $cfg
  echo -n "slots $1 batch $2: "
  for i in 1 2; do timeout 300 python bench.py --steps $((16 * $1 * $2 / 4)) --warmup $(($1 * $2)) --slots $1 --batch $2 --no-cpu-baseline --no-sh-roofline --no-secondary 2>/dev/null | val; done; echo
done
