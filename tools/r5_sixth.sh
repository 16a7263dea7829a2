#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
tag=${1:-r05f}
B="--no-cpu-baseline --no-sh-roofline --no-secondary"
for n in 16 20 24; do timeout 300 python tools/sweep_timing.py $n 2>&1 | grep "bin period\|sweep span" | cut -c1-140; done
echo "16 designs, 8 waves"; EMAGLS_REG_WAVES=8 timeout 300 python tools/sweep_timing.py 16 2>&1 | grep "bin period\|sweep span" | cut -c1-140
echo "20 designs, 12 waves"; EMAGLS_REG_WAVES=12 timeout 300 python tools/sweep_timing.py 20 2>&1 | grep "bin period\|sweep span" | cut -c1-140
run() { name=$1; shift; timeout 600 "$@" > gpurun_out/${tag}_$name.json 2> gpurun_out/${tag}_$name.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${tag}_$name.json").read().strip().splitlines()[-1]); print("$name", round(d["value"],1), d["ms_per_step"], round(d["roofline"]["frac"],3), round(d["roofline"]["avg_launch_us"]))
except Exception as e: print("$name FAILED", e); print(open("gpurun_out/${tag}_$name.err").read()[-600:])
PY
}
run b20a python bench.py --steps 20 --warmup 5 $B
run b20b python bench.py --steps 20 --warmup 5 $B
run b20c python bench.py --steps 20 --warmup 5 $B
run b128 python bench.py --steps 128 --warmup 32 $B
run b512 python bench.py --steps 512 --warmup 64 $B
export TMPDIR=/tmp; cd /tmp
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS"; do
  d=$R/gpurun_out/${tag}_pmc_$(echo $c | cut -d' ' -f1)
  timeout 400 rocprofv3 --kernel-trace --pmc $c -d $d -o pmc -- python3 $R/tools/experiments/sweep_only.py 32 4 > $d.log 2>&1
done
cd $R
python - <<'PY'
import glob, sqlite3, os, sys
from collections import defaultdict
tag = "r05f"
for d in sorted(glob.glob("gpurun_out/%s_pmc_*" % tag)):
    if not os.path.isdir(d): continue
    dbs = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
    if not dbs: print(d, "no db"); continue
    cur = sqlite3.connect(dbs[0]).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    pe = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
    pi = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
    picols = [r[1] for r in cur.execute(f"pragma table_info({pi})")]
    namecol = "name" if "name" in picols else "symbol"
    q = (f"select s.kernel_name, i.{namecol}, d.id, d.end - d.start, sum(e.value) from {pe} e join {pi} i on e.pmc_id = i.id "
         f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id where s.kernel_name like '%sweep_reg%' group by d.id, i.{namecol}")
    acc = defaultdict(list)
    for name, c, did, dur, v in cur.execute(q):
        acc[c].append((dur, v))
    for c, vals in acc.items():
        print(c, "launches", len(vals), "last: dur %.1f us value %.4g" % (vals[-1][0] / 1e3, vals[-1][1]))
PY
rm -rf gpurun_out/${tag}_pmc_*/
