#!/bin/bash
# Run-to-run spread of the two bench figures on one box: N runs each of `bench.py --steps 20 --warmup 5` and `--steps 128 --warmup 32`,
# interleaved.      bash tools/experiments/bench_spread.sh [N]      (through gpurun; prints one line per run and the means)
N=${1:-8}
cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-sh-roofline --no-secondary"
for rep in $(seq 1 $N); do
  for sw in "20 5" "128 32"; do
    set -- $sw
    v=$(timeout 300 python bench.py --steps $1 --warmup $2 $B 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), round(d['roofline']['avg_launch_us'],1))")
    echo "steps $1 run $rep: $v"
  done
done
