#!/bin/bash
# sweep-kernel iteration: operand fetch placement (EMAGLS_SWEEP_FETCH 0..3), in-kernel stamps and the 8 / 16-design launch durations
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; tag=${1:-r03_g}
for m in 0 1 2 3; do
  EMAGLS_SWEEP_FETCH=$m python tools/sweep_timing.py 8 > gpurun_out/${tag}_m${m}_timing8.log 2>&1
  echo "== fetch mode $m"; sed -n 2,10p gpurun_out/${tag}_m${m}_timing8.log
  EMAGLS_SWEEP_FETCH=$m timeout 300 python bench.py --steps 128 --warmup 32 --no-cpu-baseline --no-sh-roofline --no-secondary > gpurun_out/${tag}_m${m}_k128.json 2> gpurun_out/${tag}_m${m}_k128.err
  echo "mode $m steps 128: $(sed 's/.*"value": \([0-9.]*\).*/\1/' gpurun_out/${tag}_m${m}_k128.json | cut -c1-12) us/bin(16 designs per launch, 4 in flight) $(sed 's/.*"us_per_bin": \([0-9.]*\).*/\1/' gpurun_out/${tag}_m${m}_k128.json | cut -c1-6)"
done
