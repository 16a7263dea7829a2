#!/bin/bash
# experiment: 16-design batches x forks; soak run for stalls
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; tag=${1:-r03_e}
export EMAGLS_BATCH_MAX=16
for fork in 1 2 4; do
  for st in "20 5" "128 32"; do set -- $st
    timeout 200 python bench.py --steps $1 --warmup $2 --slots 4 --batch 16 --fork $fork --no-cpu-baseline --no-sh-roofline --no-secondary > gpurun_out/${tag}_f${fork}_k$1.json 2> gpurun_out/${tag}_f${fork}_k$1.err
    echo "batch 16 fork $fork steps $1: $(sed 's/.*"value": \([0-9.]*\).*/\1/' gpurun_out/${tag}_f${fork}_k$1.json | cut -c1-12)"
  done
done
timeout 300 python bench.py --steps 4096 --warmup 64 --slots 4 --batch 16 --no-cpu-baseline --no-sh-roofline --no-secondary > gpurun_out/${tag}_soak4096.json 2> gpurun_out/${tag}_soak.err
echo "soak 4096 (4 x 16): $(sed 's/.*"value": \([0-9.]*\).*/\1/' gpurun_out/${tag}_soak4096.json | cut -c1-12)"
timeout 300 python bench.py --steps 4100 --warmup 37 --slots 3 --batch 13 --no-cpu-baseline --no-sh-roofline --no-secondary > gpurun_out/${tag}_soak4100.json 2>> gpurun_out/${tag}_soak.err
echo "soak 4100 (3 x 13): $(sed 's/.*"value": \([0-9.]*\).*/\1/' gpurun_out/${tag}_soak4100.json | cut -c1-12)"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sixteen or lane_batch or batch_of" > gpurun_out/${tag}_tests.log 2>&1; tail -3 gpurun_out/${tag}_tests.log
