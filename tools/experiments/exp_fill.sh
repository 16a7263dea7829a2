#!/bin/bash
# rocprofv3 kernel trace of the short (driver-style) bench run, analysed by tools/fill_timeline.py
tag=${1:-fill}; shift
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/${tag}_prof -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sh-roofline --no-secondary "$@" > $R/gpurun_out/${tag}_prof.log 2>&1
cd $R
python tools/fill_timeline.py gpurun_out/${tag}_prof 3 > gpurun_out/${tag}_fill.md 2>&1
rm -rf gpurun_out/${tag}_prof
cat gpurun_out/${tag}_fill.md; tail -2 gpurun_out/${tag}_prof.log | cut -c1-300
