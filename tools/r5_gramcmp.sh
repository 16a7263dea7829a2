#!/bin/bash
# gram_lds_kernel (16 x 16 x 4) against gram_lds4_kernel (4 x 4 x 4, four blocks): one lane group of 16 designs at a time
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
tag=${1:-r05ab}
B="--no-cpu-baseline --no-sh-roofline --no-secondary"
cd /tmp && export TMPDIR=/tmp
for m in 0 1; do
  export EMAGLS_GRAM_MFMA4=$m EMAGLS_GEMM_MFMA4=$m EMAGLS_BATCH_GROUPS=1
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof_m$m -o p -- python3 $R/bench.py --steps 64 --warmup 0 --slots 1 --batch 16 $B > $R/gpurun_out/${tag}_prof_m$m.log 2>&1
  (cd $R; python tools/kernel_avgs.py gpurun_out/${tag}_prof_m$m > gpurun_out/${tag}_kernels_m$m.md 2>&1; rm -rf gpurun_out/${tag}_prof_m$m; echo "MFMA4=$m"; grep "gram_lds\|gemm_tn" gpurun_out/${tag}_kernels_m$m.md | cut -c1-140)
done
