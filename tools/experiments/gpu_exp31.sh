#!/bin/bash
# HRIR transform with 8 / 4 directions per sub-tile (148 / 82 KB of LDS)
R=$GRAFT_REPO_ROOT; cd $R
for kb in 150 83 150 83; do
  v20=$(EMAGLS_HRIR_FFT_LDS_KB=$kb timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-sh-roofline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f' % d['value'])")
  v256=$(EMAGLS_HRIR_FFT_LDS_KB=$kb timeout 600 python bench.py --steps 512 --warmup 64 --no-cpu-baseline --no-sh-roofline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f' % d['value'])")
  echo "lds budget $kb KB  20: $v20   512: $v256"
done
