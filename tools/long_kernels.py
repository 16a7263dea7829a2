"""List the dispatches of a rocprofv3 kernel trace that run longer than a threshold and everything that overlaps the longest one."""
import glob, os, sqlite3, sys
d = sys.argv[1]; thr_us = float(sys.argv[2]) if len(sys.argv) > 2 else 20000.0
db = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)[0]
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]; ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
qcol = "d.queue_id" if "queue_id" in cols else "0"
rows = list(cur.execute(f"select s.kernel_name, d.start, d.end, {qcol}, d.grid_size_x, d.grid_size_z from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
long_ = [r for r in rows if (r[2] - r[1]) / 1e3 > thr_us]
print("dispatches longer than", thr_us, "us:")
for r in long_: print(f"  {r[0][:70]}  {(r[2]-r[1])/1e3:.0f} us  queue {r[3]} grid.x {r[4]} z {r[5]}  start {r[1]/1e6:.3f} ms")
if long_:
    L = max(long_, key=lambda r: r[2] - r[1])
    print("overlapping the longest one (", L[0][:40], "):")
    for r in rows:
        if r is not L and r[1] < L[2] and r[2] > L[1]:
            print(f"  {r[0][:70]}  start +{(r[1]-L[1])/1e3:.0f} us  dur {(r[2]-r[1])/1e3:.0f} us  queue {r[3]} grid.x {r[4]} z {r[5]}")
