#!/bin/bash
# HRIR transform with 4 / 2 / 1 directions per sub-tile (82 / 49 / 33 KB of LDS: 1 / 3 / 4 workgroups per CU)
R=$GRAFT_REPO_ROOT; cd $R
for kb in 83 50 34; do
  v20=$(EMAGLS_HRIR_FFT_LDS_KB=$kb timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sh-roofline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f' % d['value'])")
  v256=$(EMAGLS_HRIR_FFT_LDS_KB=$kb timeout 600 python bench.py --steps 512 --warmup 64 --no-cpu-baseline --no-sh-roofline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f' % d['value'], d['stages_ms'].get('hrir_prologue', d['stages_ms']))")
  hs=$(EMAGLS_HRIR_FFT_LDS_KB=$kb timeout 300 python -c "
from tools import bench_secondary as S
print(S.config3_hrir_sets(4, 16, rounds=8)['filter_sets_per_s'])" 2>/dev/null | tail -1)
  echo "lds budget $kb KB  20: $v20   512: $v256   hrir sets: $hs"
done
