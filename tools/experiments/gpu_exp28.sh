#!/bin/bash
# sweep kernel capped at 168 VGPRs (3 waves per SIMD): room for other kernels next to it against spills on the chain
R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests -m gpu -q -x -k "config3 or sixteen or one_geometry or from_atf_persistent or magls_filters_config2" 2>&1 | tail -2
timeout 200 python tools/sweep_timing.py 8 2>&1 | tail -10 | head -2
timeout 200 python tools/sweep_timing.py 16 2>&1 | tail -10 | head -2
for rep in 1 2; do
  v20=$(timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sh-roofline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f sweep %.0f us' % (d['value'], d['roofline']['avg_launch_us']))")
  v128=$(timeout 600 python bench.py --steps 256 --warmup 64 --no-cpu-baseline --no-sh-roofline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f sweep %.0f us' % (d['value'], d['roofline']['avg_launch_us']))")
  hs=$(timeout 300 python -c "
from tools import bench_secondary as S
print(S.config3_hrir_sets(4, 16, rounds=8)['filter_sets_per_s'])" 2>/dev/null | tail -1)
  echo "rep $rep  20: $v20   256: $v128   hrir sets: $hs"
done
