#!/bin/bash
# round 4, experiment 1: complementary stage orders of the lane groups (EMAGLS_STAGGER) at 20 and 128 steps
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
B="--no-cpu-baseline --no-sh-roofline --no-secondary"
for rep in 1 2; do
for sg in 0 1 21 11; do
  EMAGLS_STAGGER=$sg timeout 300 python bench.py --steps 20 --warmup 5 $B > gpurun_out/r4e1_s20_sg${sg}_$rep.json 2> gpurun_out/r4e1_s20_sg${sg}_$rep.err
  EMAGLS_STAGGER=$sg timeout 300 python bench.py --steps 128 --warmup 32 $B > gpurun_out/r4e1_s128_sg${sg}_$rep.json 2> gpurun_out/r4e1_s128_sg${sg}_$rep.err
done
done
# batches of 8, four in flight, alternating orders against one order
for so in 0 alt; do
  EMAGLS_BENCH_STAGE_ORDER=$so timeout 300 python bench.py --steps 128 --warmup 32 --batch 8 $B > gpurun_out/r4e1_b8_so${so}.json 2> gpurun_out/r4e1_b8_so${so}.err
done
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "batch or config3" > gpurun_out/r4e1_tests.log 2>&1
tail -3 gpurun_out/r4e1_tests.log
for f in gpurun_out/r4e1_*.json; do echo "$f $(python -c "import json,sys; d=json.load(open('$f')); print(round(d['value'],1), round(d['roofline']['avg_launch_us'],1))" 2>&1 | tail -1)"; done
