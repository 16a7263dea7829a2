#!/bin/bash
# round 4, experiment 4: synthesising sweep with its own least-squares bins (no G at all) -- parity subset, bench
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config4.py -m gpu -q -x -rP -k "emagls_filters_thin or emagls2_filters_thin or config3_full or emagls_low_orders or ema_in_ch or sweep_variants or custom_sh or geometry or lane_batch or sixteen or one_ranks_share or large_radius" > gpurun_out/r4e4_tests.log 2>&1
tail -5 gpurun_out/r4e4_tests.log
grep -h "rel = \|norm_diff\|worst rel" gpurun_out/r4e4_tests.log | head -40
B="--no-cpu-baseline --no-sh-roofline --no-secondary"
for rep in 1 2; do
for sy in 1 0; do
  EMAGLS_SWEEP_SYNTH=$sy timeout 300 python bench.py --steps 20 --warmup 5 $B > gpurun_out/r4e4_s20_sy${sy}_$rep.json 2> gpurun_out/r4e4_s20_sy${sy}_$rep.err
  EMAGLS_SWEEP_SYNTH=$sy timeout 300 python bench.py --steps 128 --warmup 32 $B > gpurun_out/r4e4_s128_sy${sy}_$rep.json 2> gpurun_out/r4e4_s128_sy${sy}_$rep.err
done
done
for f in gpurun_out/r4e4_*.json; do echo "$f $(python -c "import json,sys; d=json.load(open('$f')); print(round(d['value'],1), round(d['roofline']['avg_launch_us'],1), d['single_design_latency_ms'])" 2>&1 | tail -1)"; done
tail -3 gpurun_out/r4e4_s20_sy1_1.err
