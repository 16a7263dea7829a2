#!/bin/bash
# end-of-round evidence: tests, smoke, bench line, kernel stats (single design and batch), PMC traffic passes
R=$GRAFT_REPO_ROOT; cd $R
python -m pytest tests -m gpu -q -rP > gpurun_out/final_tests_full.log 2>&1; tail -3 gpurun_out/final_tests_full.log > gpurun_out/final_tests.log
grep -h "norm_diff=\|rel = \|^case (" gpurun_out/final_tests_full.log > gpurun_out/final_parity.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final_smoke.log 2>&1
python bench.py > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_single -o bench -- python3 $R/bench.py --steps 8 --warmup 2 --concurrent 1 --batch 1 --no-cpu-baseline --no-sh-roofline > $R/gpurun_out/prof_single.log 2>&1
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_batch -o bench -- python3 $R/bench.py --steps 32 --warmup 2 --concurrent 8 --batch 8 --no-cpu-baseline --no-sh-roofline > $R/gpurun_out/prof_batch.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_fetch2 -o pmc -- python3 $R/bench.py --steps 8 --warmup 2 --concurrent 1 --batch 1 --no-cpu-baseline > $R/gpurun_out/pmc_fetch2.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_write2 -o pmc -- python3 $R/bench.py --steps 8 --warmup 2 --concurrent 1 --batch 1 --no-cpu-baseline > $R/gpurun_out/pmc_write2.log 2>&1
cd $R; cat gpurun_out/final_tests.log gpurun_out/final_smoke.log; cut -c1-400 gpurun_out/bench_final.json
