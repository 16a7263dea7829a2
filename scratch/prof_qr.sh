#!/bin/bash
# usage: prof_qr.sh <tag> [ENV=VAL ...]; kernel averages of the 8-design launches only
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; tag=$1; shift
for kv in "$@"; do export "$kv"; done
cd /tmp
timeout 200 rocprofv3 --kernel-trace -d $R/gpurun_out/q_$tag -o t -- python3 $R/bench.py --steps 16 --warmup 2 --concurrent 8 --batch 8 --no-cpu-baseline --no-sh-roofline > $R/gpurun_out/q_$tag.log 2>&1
python3 - <<PY
import sqlite3, glob
con=sqlite3.connect("$R/gpurun_out/q_$tag/t_results.db"); cur=con.cursor()
tabs=[r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if t.startswith('rocpd_kernel_dispatch')][0]; ks=[t for t in tabs if t.startswith('rocpd_info_kernel_symbol')][0]
q=f"select s.kernel_name, d.grid_size_z, count(*), avg(d.end-d.start)/1000.0 from {kd} d join {ks} s on d.kernel_id=s.id group by 1,2"
for n,z,c,a in cur.execute(q):
    if (z==8 or 'sweep_persist' in n) and any(k in n for k in ("dspace_g","sweep_persist","jacobi","factor_qr","qt_kernel","hy_partial","gram_mfma","hrir_fft","qform")): print("$tag", n[:60], c, round(a,1))
PY
