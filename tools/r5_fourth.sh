#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
tag=${1:-r05d}
B="--no-cpu-baseline --no-sh-roofline --no-secondary"
timeout 300 python -m pytest tests/test_gpu_stages.py -q -rP -k wave_reduction 2>&1 | grep "self test\|passed\|failed"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "sweep_variants or residency_is_decided or config3_full or synthesising_sweep_on_other" > gpurun_out/${tag}_parity_sel.log 2>&1; tail -2 gpurun_out/${tag}_parity_sel.log
for n in 1 16 32; do timeout 300 python tools/sweep_timing.py $n > gpurun_out/${tag}_timing_$n.log 2>&1; cut -c1-160 gpurun_out/${tag}_timing_$n.log | grep -v "^ *$" ; done
for w in 4 8 12; do echo "EMAGLS_REG_WAVES=$w, 8 designs"; EMAGLS_REG_WAVES=$w timeout 300 python tools/sweep_timing.py 8 2>&1 | grep "bin period\|sweep span" | cut -c1-140; done
run() { name=$1; shift; timeout 600 "$@" > gpurun_out/${tag}_$name.json 2> gpurun_out/${tag}_$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${tag}_$name.json").read().strip().splitlines()[-1]); print("$name", round(d["value"],1), d["ms_per_step"])
except Exception as e: print("$name FAILED", e)
PY
}
run b20a python bench.py --steps 20 --warmup 5 $B
run b20_b20 python bench.py --steps 20 --warmup 5 --batch 20 --slots 2 $B
run b20_b10 python bench.py --steps 20 --warmup 5 --batch 10 $B
run b128w python bench.py --steps 128 --warmup 32 $B
EMAGLS_BENCH_WAVES=0 run b128s python bench.py --steps 128 --warmup 32 $B
run b128_b32 python bench.py --steps 128 --warmup 32 --batch 32 --slots 4 $B
EMAGLS_BENCH_WAVES=0 run b128_b32s python bench.py --steps 128 --warmup 32 --batch 32 --slots 4 $B
run b512w python bench.py --steps 512 --warmup 64 $B
run b512_b32 python bench.py --steps 512 --warmup 64 --batch 32 --slots 4 $B
EMAGLS_BENCH_WAVES=0 run b512_b32s python bench.py --steps 512 --warmup 64 --batch 32 --slots 4 $B
