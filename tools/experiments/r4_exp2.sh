#!/bin/bash
# round 4, experiment 2: the synthesising sweep (sweep_synth.hip) -- parity, per-bin timing, bench
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -rP -k "emagls_filters_thin or emagls2_filters_thin or config3_full or emagls_low_orders or ema_in_ch" > gpurun_out/r4e2_tests.log 2>&1
tail -5 gpurun_out/r4e2_tests.log
grep -h "rel = \|norm_diff" gpurun_out/r4e2_tests.log | head -30
for n in 1 8 16; do
  for sp in 0 50 67; do
    EMAGLS_SYNTH_SPLIT=$sp timeout 200 python tools/sweep_timing.py $n > gpurun_out/r4e2_timing_n${n}_sp${sp}.txt 2>&1
    echo "== designs $n split $sp"; grep "bin period\|hop 1 total\|hop 2\|M phase\|p phase\|partial phase\|sweep span" gpurun_out/r4e2_timing_n${n}_sp${sp}.txt | head -8
  done
done
EMAGLS_SWEEP_SYNTH=0 timeout 200 python tools/sweep_timing.py 8 > gpurun_out/r4e2_timing_n8_nosynth.txt 2>&1; grep "bin period" gpurun_out/r4e2_timing_n8_nosynth.txt | head -1
B="--no-cpu-baseline --no-sh-roofline --no-secondary"
for sy in 1 0; do
  EMAGLS_SWEEP_SYNTH=$sy timeout 300 python bench.py --steps 20 --warmup 5 $B > gpurun_out/r4e2_s20_sy$sy.json 2> gpurun_out/r4e2_s20_sy$sy.err
  EMAGLS_SWEEP_SYNTH=$sy timeout 300 python bench.py --steps 128 --warmup 32 $B > gpurun_out/r4e2_s128_sy$sy.json 2> gpurun_out/r4e2_s128_sy$sy.err
  EMAGLS_SWEEP_SYNTH=$sy timeout 300 python bench.py --steps 128 --warmup 32 --batch 8 $B > gpurun_out/r4e2_b8_sy$sy.json 2> gpurun_out/r4e2_b8_sy$sy.err
done
for f in gpurun_out/r4e2_*.json; do echo "$f $(python -c "import json,sys; d=json.load(open('$f')); print(round(d['value'],1), round(d['roofline']['avg_launch_us'],1))" 2>&1 | tail -1)"; done
tail -3 gpurun_out/r4e2_s20_sy1.err
