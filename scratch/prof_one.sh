#!/bin/bash
# usage: prof_one.sh <tag> ; runs a lane-batch bench under rocprofv3 and prints the per-batch time of selected kernels
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p_$1 -o t -- python3 $R/bench.py --steps 16 --warmup 2 --concurrent 8 --batch 8 --no-cpu-baseline --no-sh-roofline > $R/gpurun_out/p_$1.log 2>&1
python3 - <<PY
import sqlite3
con=sqlite3.connect("$R/gpurun_out/p_$1/t_results.db"); cur=con.cursor()
for n,c,t,a,p in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    if any(k in n for k in ("dspace_g","sweep_persist","jacobi","factor_qr")): print("$1", n[:50], round(a,1))
PY
