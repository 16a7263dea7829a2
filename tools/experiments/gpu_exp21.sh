#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "config3 or sh_basis or getsh or get_sh" 2>&1 | tail -2
for rep in 1 2 3; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null > /tmp/b20.json
  v20=$(python -c "import sys,json; d=json.loads(open('/tmp/b20.json').read()); print('%.0f' % d['value'], {k:v for k,v in d.items() if 'sh' in k.lower()})")
  v128=$(timeout 600 python bench.py --steps 128 --warmup 32 --no-cpu-baseline --no-sh-roofline --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f sweep %.0f us' % (d['value'], d['roofline']['avg_launch_us']))")
  echo "rep $rep  20: $v20   128: $v128"
done
