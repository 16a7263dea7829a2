#!/bin/bash
# long runs: a stalled persistent sweep (spin limit, launch-per-bin fallback) would show as a collapse of the throughput
R=$GRAFT_REPO_ROOT; cd $R
val() { python -c "import sys,json; [print(json.loads(l)['steps'], round(json.loads(l)['value'],1), end=' | ') for l in sys.stdin if l.startswith('{')]"; }
for k in 4096 4096 8192; do timeout 600 python bench.py --steps $k --warmup 32 --no-cpu-baseline --no-sh-roofline --no-secondary 2>/dev/null | val; done; echo
for i in 1 2 3 4 5 6; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sh-roofline --no-secondary 2>/dev/null | val; done; echo
