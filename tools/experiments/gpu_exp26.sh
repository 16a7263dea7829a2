#!/bin/bash
# geometry-sharing batches: kernel averages and steady-state timeline (how long the chained sweeps run next to the other stages)
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/hs_prof -o hs -- python3 $R/tools/experiments/hrir_sets_prof.py 4 > $R/gpurun_out/hs_prof.log 2>&1
cd $R
python tools/kernel_avgs.py gpurun_out/hs_prof > gpurun_out/r03_hrir_sets_kernels.md 2>&1
python tools/timeline.py gpurun_out/hs_prof 12 > gpurun_out/r03_hrir_sets_timeline.md 2>&1
rm -rf gpurun_out/hs_prof
head -24 gpurun_out/r03_hrir_sets_kernels.md | cut -c1-160; head -14 gpurun_out/r03_hrir_sets_timeline.md | cut -c1-160; tail -1 gpurun_out/hs_prof.log | cut -c1-300
