#!/bin/bash
# the gpu suite with its files in reverse order (order-dependent failures: one process, shared plan cache and stream pool)
R=$GRAFT_REPO_ROOT; cd $R
files=$(ls tests/test_gpu_*.py tests/test_mex_gateway.py tests/test_fixture_parity.py | sort -r | tr '\n' ' ')
timeout 2400 python -m pytest $files -m gpu -q 2>&1 | grep -v "^  File\|^Extension\|^$" | tail -6 | cut -c1-200
