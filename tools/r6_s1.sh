#!/bin/bash
# round 6, GPU session 1: new full-size test of the timed kernel, case-27 family against 40-digit rows, new-radii trace, bench
mkdir -p gpurun_out/r6_s1
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r6_s1
timeout 900 python -m pytest tests/test_gpu_timed_kernel.py tests/test_gpu_jobs.py -x -q -s > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
timeout 600 python tools/experiments/case27_gpu_vs_exact.py > $O/case27.log 2>&1
EMAGLS_JOBS_TRACE=1 timeout 600 python -c "
from tools import bench_secondary as B
import json
print(json.dumps(B.config4_rank_share_runner(reps=2)))
" > $O/config4_runner.log 2> $O/config4_runner.trace
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench20.json 2> $O/bench20.err; echo "bench rc=$?" >> $O/bench20.err
tail -3 $O/tests.log; cat $O/case27.log; tail -2 $O/config4_runner.log | cut -c1-600; tail -3 $O/bench20.err
