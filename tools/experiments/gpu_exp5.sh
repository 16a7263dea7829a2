#!/bin/bash
# sweep-kernel iteration: in-kernel stamps, parity of the sweep tests, bench at the default configuration
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; tag=${1:-r03_f}
python tools/sweep_timing.py 1 > gpurun_out/${tag}_sweep_timing1.log 2>&1; head -11 gpurun_out/${tag}_sweep_timing1.log
python tools/sweep_timing.py 8 > gpurun_out/${tag}_sweep_timing8.log 2>&1; head -3 gpurun_out/${tag}_sweep_timing8.log
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config4.py -m gpu -q -x -k "config3_full or lane_batch or sixteen or sweep_variants or tiny_array or from_atf or magls_filters_config2 or rank" > gpurun_out/${tag}_tests.log 2>&1; tail -3 gpurun_out/${tag}_tests.log
for st in "20 5" "128 32"; do set -- $st
  timeout 300 python bench.py --steps $1 --warmup $2 --no-cpu-baseline --no-sh-roofline --no-secondary > gpurun_out/${tag}_k$1.json 2> gpurun_out/${tag}_k$1.err
  echo "steps $1: $(sed 's/.*"value": \([0-9.]*\).*/\1/' gpurun_out/${tag}_k$1.json | cut -c1-12) us/bin $(sed 's/.*"us_per_bin": \([0-9.]*\).*/\1/' gpurun_out/${tag}_k$1.json | cut -c1-6) frac $(sed 's/.*"frac": \([0-9.]*\).*/\1/' gpurun_out/${tag}_k$1.json | cut -c1-6)"
done
timeout 300 python bench.py --steps 20 --warmup 5 --batch 8 --no-cpu-baseline --no-sh-roofline --no-secondary > gpurun_out/${tag}_b8_k20.json 2> gpurun_out/${tag}_b8_k20.err
echo "batch 8 steps 20: $(sed 's/.*"value": \([0-9.]*\).*/\1/' gpurun_out/${tag}_b8_k20.json | cut -c1-12) us/bin $(sed 's/.*"us_per_bin": \([0-9.]*\).*/\1/' gpurun_out/${tag}_b8_k20.json | cut -c1-6)"
