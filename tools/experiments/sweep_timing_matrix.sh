#!/bin/bash
# in-kernel stamps of the resident sweep (tools/sweep_timing.py) for 1, 8 and 16 designs per launch over environment settings:
#   bash tools/experiments/sweep_timing_matrix.sh <tag> "<VAR=a>" "<VAR=b VAR2=c>" ...      ("" = the default configuration)
tag=$1; shift
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
i=0
for cfg in "$@"; do
  i=$((i+1))
  for n in ${DESIGNS:-1 8 16}; do
    env $cfg timeout 200 python tools/sweep_timing.py $n > gpurun_out/${tag}_c${i}_n$n.txt 2>&1
    echo "== [$cfg] designs $n"; grep "sweep span\|bin period\|hop 1 total\|hop 2\|M phase\|p phase\|partial phase" gpurun_out/${tag}_c${i}_n$n.txt
  done
done
