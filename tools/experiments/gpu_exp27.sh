#!/bin/bash
# geometry-sharing batches: batches in flight x sets per batch x sweep form
R=$GRAFT_REPO_ROOT; cd $R
for tw in 1 0; do for cfg in "4 16" "6 16" "4 8" "6 8" "8 8" "4 12"; do
  set -- $cfg
  v=$(EMAGLS_SWEEP_TWIN=$tw timeout 300 python -c "
from tools import bench_secondary as S
d=S.config3_hrir_sets($1, $2, rounds=8); print(d['filter_sets_per_s'])" 2>/dev/null | tail -1)
  echo "twin=$tw batches=$1 sets=$2 -> $v sets/s"
done; done
