#!/bin/bash
# fill timeline of the driver's run at the default configuration (4 x 16) and with forks
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out; tag=${1:-r03_h}
export TMPDIR=/tmp; cd /tmp
for fork in 1 4; do
  timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/${tag}_prof20_f$fork -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --fork $fork --no-cpu-baseline --no-sh-roofline --no-secondary > $R/gpurun_out/${tag}_prof20_f$fork.log 2>&1
  python3 $R/tools/fill_timeline.py $R/gpurun_out/${tag}_prof20_f$fork 2 > $R/gpurun_out/${tag}_fill20_f$fork.md 2>&1
  rm -rf $R/gpurun_out/${tag}_prof20_f$fork
  tail -1 $R/gpurun_out/${tag}_prof20_f$fork.log | cut -c1-160
done
