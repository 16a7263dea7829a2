#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
for fm in 4; do for n in 8 16; do
  echo "== fetch mode $fm, $n designs (twin off)"; EMAGLS_SWEEP_TWIN=0 EMAGLS_SWEEP_FETCH=$fm timeout 200 python tools/sweep_timing.py $n 2>&1 | tail -10
done; done
