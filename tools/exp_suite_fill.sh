#!/bin/bash
# full gpu suite (no -x), then the short/long bench lines and the traced 20-step run
tag=${1:-x}; shift
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -rP ${PYTEST_K:+-k "$PYTEST_K"} > gpurun_out/${tag}_tests_full.log 2>&1
tail -12 gpurun_out/${tag}_tests_full.log | cut -c1-220
grep -h "norm_diff=\|rel = \|^case (\|rel L\|worst rel" gpurun_out/${tag}_tests_full.log > gpurun_out/${tag}_parity.log
bash tools/exp_fill2.sh ${tag}_fill "$@"
