#!/bin/bash
# experiment: batches above 8 designs with the design-major block order of the resident sweep (slots x designs per batch)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; tag=${1:-r03_d}
export EMAGLS_BATCH_MAX=16
for cfg in "4 8" "2 16" "3 16" "4 16" "4 12" "3 12"; do set -- $cfg; sl=$1; bs=$2
  for st in "20 5" "128 32"; do set -- $st
    timeout 200 python bench.py --steps $1 --warmup $2 --slots $sl --batch $bs --no-cpu-baseline --no-sh-roofline --no-secondary > gpurun_out/${tag}_s${sl}_b${bs}_k$1.json 2> gpurun_out/${tag}_s${sl}_b${bs}_k$1.err
    echo "slots $sl batch $bs steps $1: $(sed 's/.*"value": \([0-9.]*\).*/\1/' gpurun_out/${tag}_s${sl}_b${bs}_k$1.json | cut -c1-12) rc $?"
  done
done
