#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; tag=${1:-r03_k}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config4.py -m gpu -q -x -k "gram_matrix or sixteen or lane_batch or batch_of or config3_full or rank" > gpurun_out/${tag}_tests.log 2>&1; tail -3 gpurun_out/${tag}_tests.log; grep "Gram matrix" gpurun_out/${tag}_tests.log
for g in 1 0; do for st in "20 5" "128 32" "20 5" "128 32"; do set -- $st
  EMAGLS_GRAM_LDS=$g timeout 200 python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-sh-roofline --no-secondary > gpurun_out/${tag}_gram${g}_k$1.json 2> gpurun_out/${tag}_gram${g}_k$1.err
  echo "gram_lds $g steps $1: $(sed 's/.*"value": \([0-9.]*\).*/\1/' gpurun_out/${tag}_gram${g}_k$1.json | cut -c1-12)"
done; done
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof_batch -o bench -- python3 $R/bench.py --steps 32 --warmup 0 --slots 1 --batch 8 --no-cpu-baseline --no-sh-roofline --no-secondary > $R/gpurun_out/${tag}_prof_batch.log 2>&1
cd $R; python tools/kernel_avgs.py gpurun_out/${tag}_prof_batch 8 > gpurun_out/${tag}_kernels_batch.md 2>&1; rm -rf gpurun_out/${tag}_prof_batch
grep -i "gram" gpurun_out/${tag}_kernels_batch.md | cut -c1-160
