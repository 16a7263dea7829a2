#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
python tools/experiments/em64_stages.py 2>&1 | tail -3
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/em64_prof -o e -- python3 $R/tools/experiments/em64_stages.py > $R/gpurun_out/em64_prof.log 2>&1
cd $R; python tools/kernel_avgs.py gpurun_out/em64_prof > gpurun_out/em64_kernels.md 2>&1; rm -rf gpurun_out/em64_prof
head -14 gpurun_out/em64_kernels.md | cut -c1-170
