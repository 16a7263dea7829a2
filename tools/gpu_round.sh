#!/bin/bash
# One GPU-box session: parity tests, smoke, the bench line, rocprofv3 kernel traces.  Usage (through gpurun):
#   bash tools/gpu_round.sh <tag> [tests|notests|testsonly] [pmc]      (testsonly: the whole gpu suite without -x, nothing else)
# Everything lands in gpurun_out/<tag>_*; copy what should be judged into profiles/.  PYTEST_K='expr' selects tests (-k).
tag=${1:-r}; what=${2:-tests}; pmc=${3:-}
export EMAGLS_BUILD_TAG=$tag
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
if [ "$what" = "tests" ]; then
  timeout 1500 python -m pytest tests -m gpu -q -rP -x --durations=40 ${PYTEST_K:+-k "$PYTEST_K"} > gpurun_out/${tag}_tests_full.log 2>&1
  tail -5 gpurun_out/${tag}_tests_full.log > gpurun_out/${tag}_tests.log
  grep -h "norm_diff=\|rel = \|^case (\|rel L\|worst rel" gpurun_out/${tag}_tests_full.log > gpurun_out/${tag}_parity.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1
fi
if [ "$what" = "testsonly" ]; then
  timeout 2400 python -m pytest tests -m gpu -q -rP --durations=40 ${PYTEST_K:+-k "$PYTEST_K"} > gpurun_out/${tag}_tests_full.log 2>&1
  tail -15 gpurun_out/${tag}_tests_full.log > gpurun_out/${tag}_tests.log
  grep -h "norm_diff=\|rel = \|^case (\|rel L\|worst rel" gpurun_out/${tag}_tests_full.log > gpurun_out/${tag}_parity.log
  cat gpurun_out/${tag}_tests.log; exit 0
fi
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench20.json 2> gpurun_out/${tag}_bench20.err
timeout 600 python bench.py --steps 128 --warmup 32 --no-cpu-baseline --no-sh-roofline --no-secondary > gpurun_out/${tag}_bench128.json 2> gpurun_out/${tag}_bench128.err
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof_single -o bench -- python3 $R/bench.py --steps 8 --warmup 0 --slots 1 --batch 1 --no-cpu-baseline --no-sh-roofline --no-secondary > $R/gpurun_out/${tag}_prof_single.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof_batch -o bench -- python3 $R/bench.py --steps 32 --warmup 0 --slots 1 --batch 8 --no-cpu-baseline --no-sh-roofline --no-secondary > $R/gpurun_out/${tag}_prof_batch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof_batch16 -o bench -- python3 $R/bench.py --steps 128 --warmup 0 --slots 1 --batch 32 --no-cpu-baseline --no-sh-roofline --no-secondary > $R/gpurun_out/${tag}_prof_batch16.log 2>&1
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/${tag}_prof_slots4 -o bench -- python3 $R/bench.py --steps 128 --warmup 0 --no-cpu-baseline --no-sh-roofline --no-secondary > $R/gpurun_out/${tag}_prof_slots4.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof_hrirsets -o hs -- python3 $R/tools/experiments/hrir_sets_prof.py 4 > $R/gpurun_out/${tag}_prof_hrirsets.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${tag}_prof_default20 -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sh-roofline --no-secondary > $R/gpurun_out/${tag}_prof_default20.log 2>&1
if [ "$pmc" = "pmc" ]; then
  # counters in passes of their own (kernel-trace only next to --pmc); one chunk of 32 designs in flight (the bench's default shape:
  # two lane groups of 16 before one 32-design sweep launch), 4 chunks executed after the three set-up runs
  PMCCMD="python3 $R/bench.py --steps 128 --warmup 0 --slots 1 --batch 32 --no-cpu-baseline --no-sh-roofline --no-secondary"
  for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU; do
    timeout 400 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/${tag}_pmc_$c -o pmc -- $PMCCMD > $R/gpurun_out/${tag}_pmc_$c.log 2>&1
  done
  cd $R
  python tools/pmc_summary.py gpurun_out/${tag}_pmc_traffic.json gpurun_out/${tag}_pmc.md gpurun_out/${tag}_pmc_* > gpurun_out/${tag}_pmc_summary.log 2>&1
  rm -rf gpurun_out/${tag}_pmc_FETCH_SIZE gpurun_out/${tag}_pmc_WRITE_SIZE gpurun_out/${tag}_pmc_SQ_*
fi
cd $R
python tools/kernel_avgs.py gpurun_out/${tag}_prof_single 1 > gpurun_out/${tag}_kernels_single.md 2>&1
python tools/kernel_avgs.py gpurun_out/${tag}_prof_batch 8 > gpurun_out/${tag}_kernels_batch.md 2>&1
python tools/kernel_avgs.py gpurun_out/${tag}_prof_batch > gpurun_out/${tag}_kernels_batch_all.md 2>&1
python tools/kernel_avgs.py gpurun_out/${tag}_prof_batch16 16 > gpurun_out/${tag}_kernels_batch16_groups.md 2>&1
python tools/kernel_avgs.py gpurun_out/${tag}_prof_batch16 > gpurun_out/${tag}_kernels_batch16_all.md 2>&1
python tools/sweep_launches.py gpurun_out/${tag}_prof_default20 gpurun_out/${tag}_prof_default20.log > gpurun_out/${tag}_default20_sweep_launches.md 2>&1
python tools/kernel_avgs.py gpurun_out/${tag}_prof_default20 > gpurun_out/${tag}_default20_kernels.md 2>&1
python tools/fill_timeline.py gpurun_out/${tag}_prof_default20 2 > gpurun_out/${tag}_fill_timeline20.md 2>&1
python tools/timeline.py gpurun_out/${tag}_prof_slots4 10 > gpurun_out/${tag}_timeline_slots4.md 2>&1
python tools/timeline.py gpurun_out/${tag}_prof_batch 3 > gpurun_out/${tag}_timeline_slots1.md 2>&1
python tools/kernel_avgs.py gpurun_out/${tag}_prof_hrirsets > gpurun_out/${tag}_hrir_sets_kernels.md 2>&1
python tools/timeline.py gpurun_out/${tag}_prof_hrirsets 12 > gpurun_out/${tag}_hrir_sets_timeline.md 2>&1
# the raw databases are large: keep the summaries
rm -rf gpurun_out/${tag}_prof_hrirsets gpurun_out/${tag}_prof_single gpurun_out/${tag}_prof_batch gpurun_out/${tag}_prof_slots4 gpurun_out/${tag}_prof_batch16 gpurun_out/${tag}_prof_default20
cat gpurun_out/${tag}_tests.log gpurun_out/${tag}_smoke.log 2>/dev/null; cut -c1-600 gpurun_out/${tag}_bench20.json; echo; cut -c1-300 gpurun_out/${tag}_bench128.json; echo; tail -3 gpurun_out/${tag}_bench20.err; head -32 gpurun_out/${tag}_kernels_batch.md; head -12 gpurun_out/${tag}_timeline_slots4.md
