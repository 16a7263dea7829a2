#!/bin/bash
# twin workgroups for 9-16 design launches: parity of the 16-design tests, per-bin timing, bench at 20 / 128 steps, both forms
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x -k "sixteen or lane_batch or batch" > gpurun_out/tw_tests.log 2>&1; tail -5 gpurun_out/tw_tests.log
timeout 200 python tools/sweep_timing.py 16 > gpurun_out/tw_t16.log 2>&1; tail -22 gpurun_out/tw_t16.log
for tw in 1 0; do
  EMAGLS_SWEEP_TWIN=$tw timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sh-roofline --no-secondary > gpurun_out/tw_b20_$tw.json 2> gpurun_out/tw_b20_$tw.err
  EMAGLS_SWEEP_TWIN=$tw timeout 600 python bench.py --steps 128 --warmup 32 --no-cpu-baseline --no-sh-roofline --no-secondary > gpurun_out/tw_b128_$tw.json 2> gpurun_out/tw_b128_$tw.err
  echo "twin=$tw"; cut -c1-420 gpurun_out/tw_b20_$tw.json; echo; cut -c1-200 gpurun_out/tw_b128_$tw.json; echo; tail -2 gpurun_out/tw_b20_$tw.err
done
