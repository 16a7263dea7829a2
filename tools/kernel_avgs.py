"""Per-kernel launch counts and average durations from a rocprofv3 --kernel-trace output directory
(the rocpd sqlite database).  Usage:  python tools/kernel_avgs.py <dir> [grid_z]  -> markdown table on stdout."""
import glob
import os
import sqlite3
import sys


def main():
    d = sys.argv[1]
    want_z = int(sys.argv[2]) if len(sys.argv) > 2 else None
    dbs = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
    if not dbs:
        raise SystemExit("no *_results.db under " + d)
    con = sqlite3.connect(dbs[0])
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    q = (f"select s.kernel_name, d.grid_size_z, count(*), avg(d.end-d.start)/1000.0, sum(d.end-d.start)/1000.0 "
         f"from {kd} d join {ks} s on d.kernel_id=s.id group by 1,2 order by 5 desc")
    rows = [r for r in cur.execute(q) if want_z is None or r[1] == want_z]
    tot = sum(r[4] for r in rows)
    print("| kernel | grid.z | launches | avg us | total us | % |")
    print("|---|---|---|---|---|---|")
    for n, z, c, a, s in rows:
        print(f"| `{n[:90]}` | {z} | {c} | {a:.1f} | {s:.0f} | {100 * s / tot:.1f} |")


if __name__ == "__main__":
    main()
