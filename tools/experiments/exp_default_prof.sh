#!/bin/bash
# rocprofv3 kernel trace of the driver's command itself, and the sweep kernel's launch durations next to the bench's own figure
tag=${1:-r}; R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof_default -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-secondary > $R/gpurun_out/${tag}_default_bench.json 2> $R/gpurun_out/${tag}_default_bench.err
cd $R
python tools/sweep_launches.py gpurun_out/${tag}_prof_default gpurun_out/${tag}_default_bench.json > gpurun_out/${tag}_sweep_launches.md 2>&1
python tools/kernel_avgs.py gpurun_out/${tag}_prof_default 8 > gpurun_out/${tag}_default_kernels_z8.md 2>&1
rm -rf gpurun_out/${tag}_prof_default
cat gpurun_out/${tag}_sweep_launches.md; head -8 gpurun_out/${tag}_default_kernels_z8.md
