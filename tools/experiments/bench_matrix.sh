#!/bin/bash
# bench.py at 20 and 128 steps over a matrix of environment settings, each REPS times (run on the GPU box through gpurun):
#   bash tools/experiments/bench_matrix.sh <tag> "<VAR=a VAR2=b>" "<VAR=c>" ...        (one quoted setting list per configuration;
#   "" is the default configuration; BENCH_ARGS adds arguments, e.g. BENCH_ARGS="--batch 8"; REPS defaults to 2)
# Prints value (filter sets/s) and the sweep launch's average duration per run; the JSON lines land in gpurun_out/<tag>_*.json.
tag=$1; shift
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
B="--no-cpu-baseline --no-sh-roofline --no-secondary $BENCH_ARGS"
i=0
for cfg in "$@"; do
  i=$((i+1))
  for rep in $(seq 1 ${REPS:-2}); do
    for sw in "20 5" "128 32"; do
      set -- $sw
      env $cfg timeout 300 python bench.py --steps $1 --warmup $2 $B > gpurun_out/${tag}_c${i}_s$1_$rep.json 2> gpurun_out/${tag}_c${i}_s$1_$rep.err
      echo "[$cfg] steps $1 rep $rep: $(python -c "import json; d=json.load(open('gpurun_out/${tag}_c${i}_s$1_$rep.json')); print(round(d['value'],1), 'sets/s, sweep', round(d['roofline']['avg_launch_us'],1), 'us')" 2>&1 | tail -1)"
    done
  done
done
