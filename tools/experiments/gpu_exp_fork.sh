#!/bin/bash
# experiment: forked pre-sweep (--fork) x hardware queues, driver-sized and long runs; fill timeline of the 20-step run
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; tag=${1:-fork}
timeout 900 python -m pytest tests/test_gpu_render.py tests/test_gpu_stages.py tests/test_gpu_shapes.py -m gpu -q -x > gpurun_out/${tag}_tests_rest.log 2>&1; tail -3 gpurun_out/${tag}_tests_rest.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "lane_batch_matches or ill_conditioned_basis or config2_with or binaural_decode_complex" > gpurun_out/${tag}_tests_new.log 2>&1; tail -3 gpurun_out/${tag}_tests_new.log
for fork in 1 4; do for q in 4 24; do
  for st in "20 5" "128 32"; do set -- $st
    GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --steps $1 --warmup $2 --fork $fork --no-cpu-baseline --no-sh-roofline --no-secondary > gpurun_out/${tag}_f${fork}_q${q}_s$1.json 2> gpurun_out/${tag}_f${fork}_q${q}_s$1.err
    echo "fork $fork queues $q steps $1: $(cut -c1-110 gpurun_out/${tag}_f${fork}_q${q}_s$1.json | sed 's/.*"value": \([0-9.]*\).*/\1/')"
  done
done; done
export TMPDIR=/tmp; cd /tmp
timeout 300 rocprofv3 --kernel-trace -d $R/gpurun_out/${tag}_prof20 -o bench -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sh-roofline --no-secondary > $R/gpurun_out/${tag}_prof20.log 2>&1
cd $R
python tools/fill_timeline.py gpurun_out/${tag}_prof20 3 > gpurun_out/${tag}_fill_timeline20.md 2>&1
rm -rf gpurun_out/${tag}_prof20
head -5 gpurun_out/${tag}_fill_timeline20.md
