#!/bin/bash
# round 6, GPU session 2: new tests, gather copy effect, cold-path traces, sweep timing
O=gpurun_out/r6_s2; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_timed_kernel.py tests/test_gpu_wide_arrays.py tests/test_gpu_jobs.py tests/test_mex_gateway.py -m gpu -x -q -s > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
for i in 1 2; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sh-roofline --no-secondary > $O/bench20_$i.json 2> $O/bench20_$i.err; done
timeout 600 python bench.py --steps 128 --warmup 32 --no-cpu-baseline --no-sh-roofline --no-secondary > $O/bench128.json 2> $O/bench128.err
EMAGLS_JOBS_TRACE=1 timeout 600 python tools/experiments/jobs_host_arrays.py > $O/jobs_host.log 2> $O/jobs_host.trace
timeout 300 python tools/sweep_timing.py 20 > $O/sweep_timing20.log 2>&1
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/prof_default20 -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sh-roofline --no-secondary > $R/$O/prof_default20.log 2>&1
cd $R
python tools/fill_timeline.py $O/prof_default20 2 > $O/fill_timeline20.md 2>&1
python tools/kernel_avgs.py $O/prof_default20 > $O/default20_kernels.md 2>&1
rm -rf $O/prof_default20
tail -4 $O/tests.log; grep -h "rel = \|microphones at" $O/tests.log | head -20
for f in $O/bench20_1.json $O/bench20_2.json $O/bench128.json; do python -c "
import json,sys
d=json.load(open('$f')); print('$f', round(d['value'],1), d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"; done
cat $O/jobs_host.log | head -12
