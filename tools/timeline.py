"""Timeline analysis of a rocprofv3 --kernel-trace run (rocpd sqlite): how busy the GPU is and what overlaps.

    python tools/timeline.py <dir> [n_sweeps]

Prints, for the steady-state part of the run (the window spanned by the last `n_sweeps` sweep launches, default 12,
i.e. the end of bench.py's timed region):
wall time, time with >= 1 kernel running, time with the sweep running, time with the sweep running ALONE, and the
per-kernel busy time (union of its dispatch intervals) as a share of the wall time."""
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def union(iv):
    iv = sorted(iv)
    tot, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def main():
    d = sys.argv[1]
    nsw = int(float(sys.argv[2])) if len(sys.argv) > 2 else 12
    db = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    rows = list(cur.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
    sw_all = [r for r in rows if ("sweep_persist" in r[0] or "sweep_synth" in r[0] or "sweep_reg" in r[0])]
    if len(sw_all) > nsw + 1:   # leave the very last sweep out (drain of the pipeline)
        t0, t1 = sw_all[-nsw - 1][1], sw_all[-2][2]
        rows = [r for r in rows if r[1] >= t0 and r[2] <= t1]
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    wall = t1 - t0
    allv = [(s, e) for _, s, e in rows]
    sweep = [(s, e) for n, s, e in rows if ("sweep_persist" in n or "sweep_synth" in n or "sweep_reg" in n)]
    others = [(s, e) for n, s, e in rows if ("sweep_persist" not in n and "sweep_synth" not in n and "sweep_reg" not in n)]
    busy, sw, ot = union(allv), union(sweep), union(others)
    both = sw + ot - busy
    print(f"steady-state window {wall / 1e6:.2f} ms, {len(rows)} dispatches")
    print(f"GPU has >= 1 kernel running: {100 * busy / wall:.1f} %   idle: {100 * (1 - busy / wall):.1f} %")
    print(f"sweep running: {100 * sw / wall:.1f} %   other kernels running: {100 * ot / wall:.1f} %   both at once: {100 * both / wall:.1f} %")
    per = defaultdict(list)
    for n, s, e in rows:
        per[n].append((s, e))
    print("| kernel | dispatches | busy % of wall | sum of durations % |")
    print("|---|---|---|---|")
    for n, iv in sorted(per.items(), key=lambda kv: -union(kv[1]))[:25]:
        print(f"| `{n[:70]}` | {len(iv)} | {100 * union(iv) / wall:.1f} | {100 * sum(e - s for s, e in iv) / wall:.1f} |")


if __name__ == "__main__":
    main()
