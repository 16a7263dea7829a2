"""Every dispatch of the LAST job list of a rocprofv3 --kernel-trace run of bench.py (the timed region of `--steps 20`: one chunk): start,
end, queue, kernel -- to see what precedes the first stage and what follows the sweep.

    python tools/experiments/last_step_dispatches.py <dir> [window_us]
"""
import glob
import os
import sqlite3
import sys

d = sys.argv[1]
win = float(sys.argv[2]) if len(sys.argv) > 2 else 9000.0
db = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)[0]
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = list(cur.execute(f"select d.start, d.end, d.queue_id, s.kernel_name, d.grid_size_x*d.grid_size_y*d.grid_size_z from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
sw = [r for r in rows if "sweep_reg_kernel" in r[3]][-1]     # the sweep launch of the timed region
sel = [r for r in rows if r[0] > sw[0] - 3000e3 and r[0] < sw[1] + 1000e3]
t0 = sel[0][0]
print("dispatches from 3 ms before the last sweep_reg_kernel launch to 1 ms after its end: %d" % len(sel))
prev_end = t0
for r in sel:
    name = r[3].split("(")[0][-46:]
    print("%9.1f .. %9.1f us  q%-3d gap %7.1f  %-46s grid %d" % ((r[0] - t0) / 1e3, (r[1] - t0) / 1e3, r[2], (r[0] - prev_end) / 1e3, name, r[4]))
    prev_end = max(prev_end, r[1])
