#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/experiments/alloc_cost.hip -o /tmp/alloc_cost && /tmp/alloc_cost | tee gpurun_out/r05_alloc_cost.log
