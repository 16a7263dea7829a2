#!/bin/bash
# footprint of every kernel of a config-3 batch (LDS / registers / grid) and what fits next to the resident sweep
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/fp_prof -o bench -- python3 $R/bench.py --steps 32 --warmup 0 --slots 1 --batch 8 --no-cpu-baseline --no-sh-roofline --no-secondary > $R/gpurun_out/fp_prof.log 2>&1
cd $R
python tools/kernel_footprint.py gpurun_out/fp_prof > gpurun_out/r03_kernel_footprint.md 2>&1
rm -rf gpurun_out/fp_prof
cat gpurun_out/r03_kernel_footprint.md
