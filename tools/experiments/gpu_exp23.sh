#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/dec_prof -o dec -- python3 $R/tools/experiments/decode_prof.py > $R/gpurun_out/dec_prof.log 2>&1
cd $R
python tools/kernel_avgs.py gpurun_out/dec_prof > gpurun_out/dec_kernels.md 2>&1
rm -rf gpurun_out/dec_prof
head -20 gpurun_out/dec_kernels.md | cut -c1-200; tail -2 gpurun_out/dec_prof.log | cut -c1-300
