#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_stages.py -m gpu -q -x 2>&1 | tail -15 | cut -c1-200
for prio in 0 1; do
  echo "== prio $prio"
  for i in 1 2 3; do timeout 300 python bench.py --steps 20 --warmup 5 --prio $prio --no-cpu-baseline --no-sh-roofline --no-secondary 2>/dev/null | cut -c80-140; done
  timeout 300 python bench.py --steps 128 --warmup 32 --prio $prio --no-cpu-baseline --no-sh-roofline --no-secondary 2>/dev/null | cut -c80-140
done
