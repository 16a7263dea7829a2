#!/bin/bash
# the N > 1 code path on one GPU: process group of one rank (RCCL barrier / gather / all-reduce), and the driver's torchrun form
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; tag=${1:-r03_m}
EMAGLS_BENCH_FORCE_PG=1 timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sh-roofline --no-secondary > gpurun_out/${tag}_forcepg.json 2> gpurun_out/${tag}_forcepg.err; echo "force_pg rc $? $(cut -c1-120 gpurun_out/${tag}_forcepg.json)"
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-sh-roofline --no-secondary > gpurun_out/${tag}_torchrun.json 2> gpurun_out/${tag}_torchrun.err; echo "torchrun rc $? $(grep -h '^{' gpurun_out/${tag}_torchrun.json | cut -c1-120)"
timeout 600 python -m pytest tests/test_gpu_config4.py -m gpu -q -x -k "job_lists" > gpurun_out/${tag}_tests.log 2>&1; tail -2 gpurun_out/${tag}_tests.log
tail -3 gpurun_out/${tag}_forcepg.err gpurun_out/${tag}_torchrun.err | cut -c1-200
