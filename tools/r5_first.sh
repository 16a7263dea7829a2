#!/bin/bash
# round 5, first GPU session of the register-resident sweep: self test, the parity tests that exercise it, per-bin timing, bench
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
tag=${1:-r05a}
timeout 300 python -m pytest tests/test_gpu_stages.py -q -rP -k wave_reduction > gpurun_out/${tag}_selftest.log 2>&1; tail -3 gpurun_out/${tag}_selftest.log
timeout 900 python -m pytest tests/test_gpu_parity.py -q -rP -x -k "sweep_variants or synthesising_sweep_on_other or config3_full or residency_is_decided" > gpurun_out/${tag}_parity_sel.log 2>&1; tail -4 gpurun_out/${tag}_parity_sel.log; grep -h "rel vs\|variant\|batch on 64" gpurun_out/${tag}_parity_sel.log | cut -c1-220
for n in 1 8 16; do timeout 300 python tools/sweep_timing.py $n > gpurun_out/${tag}_timing_$n.log 2>&1; cat gpurun_out/${tag}_timing_$n.log | cut -c1-200; done
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sh-roofline --no-secondary > gpurun_out/${tag}_bench20.json 2> gpurun_out/${tag}_bench20.err; cut -c1-400 gpurun_out/${tag}_bench20.json; tail -2 gpurun_out/${tag}_bench20.err
timeout 600 python bench.py --steps 128 --warmup 32 --no-cpu-baseline --no-sh-roofline --no-secondary > gpurun_out/${tag}_bench128.json 2> gpurun_out/${tag}_bench128.err; cut -c1-400 gpurun_out/${tag}_bench128.json; tail -2 gpurun_out/${tag}_bench128.err
EMAGLS_SWEEP_REG=0 timeout 600 python bench.py --steps 128 --warmup 32 --no-cpu-baseline --no-sh-roofline --no-secondary > gpurun_out/${tag}_bench128_slab.json 2> gpurun_out/${tag}_bench128_slab.err; cut -c1-300 gpurun_out/${tag}_bench128_slab.json
