#!/bin/bash
# what the round-end driver runs on the GPU box, in its order
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 2400 python -m pytest tests/ -x -q -m gpu > gpurun_out/driver_tests.log 2>&1; echo "pytest rc $?"; tail -2 gpurun_out/driver_tests.log | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/driver_bench.json 2> gpurun_out/driver_bench.err; echo "bench rc $?"; cut -c1-260 gpurun_out/driver_bench.json
