#!/bin/bash
# rocprofv3 kernel trace of one bench.py command and its summaries (queue-by-queue timeline, per-kernel averages, steady state):
#   bash tools/experiments/timeline_prof.sh <tag> "<VAR=a ...>" <bench.py arguments ...>
tag=$1; cfg=$2; shift 2
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out
export TMPDIR=/tmp; cd /tmp
for kv in $cfg; do export $kv; done
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof -o bench -- python3 $R/bench.py "$@" --no-cpu-baseline --no-sh-roofline --no-secondary > $R/gpurun_out/${tag}_prof.log 2>&1
cd $R
python tools/fill_timeline.py gpurun_out/${tag}_prof 2 > gpurun_out/${tag}_fill_timeline.md 2>&1
python tools/kernel_avgs.py gpurun_out/${tag}_prof > gpurun_out/${tag}_kernels.md 2>&1
python tools/timeline.py gpurun_out/${tag}_prof 10 > gpurun_out/${tag}_timeline.md 2>&1
rm -rf gpurun_out/${tag}_prof
grep -o '"value": [0-9.]*' gpurun_out/${tag}_prof.log; head -12 gpurun_out/${tag}_timeline.md
