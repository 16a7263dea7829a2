#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out; tag=${1:-r03_n}
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -rP -k "more_than_32_channels or wide_arrays_refuse or orders_5_to_7 or emagls2_filters_thin or emagls_filters_thin" > gpurun_out/${tag}_tests.log 2>&1; tail -25 gpurun_out/${tag}_tests.log | cut -c1-220
