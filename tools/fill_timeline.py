"""Fill/drain timeline of a short bench run (rocprofv3 --kernel-trace of `bench.py --steps 20 --warmup 5`):
what every queue does between the start of the timed region and the end of its last sweep.

    python tools/fill_timeline.py <dir> [n_sweeps_in_timed_region=3]
"""
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def short(n):
    n = n.replace("_ZN6emagls", "").replace("12_GLOBAL__N_1", "")
    return n[:34]


def main():
    d = sys.argv[1]
    nsw = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    db = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({kd})")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    print("dispatch columns:", cols)
    sel = f"select s.kernel_name, d.start, d.end, d.{qcol} from {kd} d join {ks} s on d.kernel_id=s.id order by d.start" if qcol else \
        f"select s.kernel_name, d.start, d.end, 0 from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"
    rows = list(cur.execute(sel))
    sw = [r for r in rows if ("sweep_persist" in r[0] or "sweep_synth" in r[0] or "sweep_reg" in r[0])]
    last = sw[-nsw:]
    # the timed region starts after the idle gap that precedes the first kernel of the unit owning the first of these sweeps
    t_end = max(r[2] for r in last)
    before = [r for r in rows if r[2] <= last[0][1]]
    # walk back from the first timed sweep to the largest idle gap in the preceding 30 ms
    t0 = last[0][1]
    ends = sorted(set([r[2] for r in rows if r[2] < last[0][1] and r[2] > last[0][1] - 40e6]))
    best_gap, t_start = 0, None
    for r in rows:
        if r[1] >= last[0][1] or r[1] < last[0][1] - 40e6:
            continue
        prev_end = max([x[2] for x in rows if x[1] < r[1]] or [r[1]])
        gap = r[1] - prev_end
        if gap > best_gap:
            best_gap, t_start = gap, r[1]
    print(f"timed region (from the idle gap of {best_gap / 1e3:.0f} us): {(t_end - t_start) / 1e6:.2f} ms")
    reg = [r for r in rows if r[1] >= t_start and r[2] <= t_end + 1]
    perq = defaultdict(list)
    for n, s, e, q in reg:
        perq[q].append((n, s, e))
    for q, lst in sorted(perq.items()):
        print(f"\n-- queue {q}: {len(lst)} dispatches, first {(lst[0][1] - t_start) / 1e3:.0f} us, last end {(lst[-1][2] - t_start) / 1e3:.0f} us,"
              f" busy {sum(e - s for _, s, e in lst) / 1e3:.0f} us")
        agg = []
        for n, s, e in lst:
            if agg and agg[-1][0] == short(n):
                agg[-1][2] = e
                agg[-1][3] += 1
            else:
                agg.append([short(n), s, e, 1])
        for n, s, e, c in agg:
            if (e - s) > 60e3 or "sweep" in n:
                print(f"   {(s - t_start) / 1e3:8.0f} .. {(e - t_start) / 1e3:8.0f} us  {n} x{c}")


if __name__ == "__main__":
    main()
