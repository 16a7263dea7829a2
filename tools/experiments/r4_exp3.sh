#!/bin/bash
# round 4, experiment 3: kernel timelines of the 20-step run with and without the synthesising sweep
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp; cd /tmp
for sy in 1 0; do
  export EMAGLS_SWEEP_SYNTH=$sy
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r4e3_prof_sy$sy -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sh-roofline --no-secondary > $R/gpurun_out/r4e3_prof_sy$sy.log 2>&1
done
cd $R
for sy in 1 0; do
  python tools/fill_timeline.py gpurun_out/r4e3_prof_sy$sy 2 > gpurun_out/r4e3_fill_sy$sy.md 2>&1
  python tools/kernel_avgs.py gpurun_out/r4e3_prof_sy$sy > gpurun_out/r4e3_kernels_sy$sy.md 2>&1
  rm -rf gpurun_out/r4e3_prof_sy$sy
done
head -5 gpurun_out/r4e3_fill_sy1.md
