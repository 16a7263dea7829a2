#!/bin/bash
# short-run experiments: plain bench lines at 20 and 128 steps, then the traced 20-step run
tag=${1:-fill}; shift
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sh-roofline --no-secondary "$@" 2>/dev/null | cut -c1-180; done
timeout 300 python bench.py --steps 128 --warmup 32 --no-cpu-baseline --no-sh-roofline --no-secondary "$@" 2>/dev/null | cut -c1-180
bash tools/exp_fill.sh $tag "$@" | grep -v "^   \|^$" 
