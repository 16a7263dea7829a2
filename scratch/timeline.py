"""Timeline analysis of a rocprofv3 kernel trace: busy union, per-kernel overlap, gaps.  python scratch/timeline.py db [t_from_frac]"""
import sqlite3, sys, numpy as np
con = sqlite3.connect(sys.argv[1]); cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
print(cols)
rows = list(cur.execute("select name, start, end, stream_id, queue_id from kernels order by start"))
print(len(rows), 'kernels')
t0 = rows[0][1]
st = np.array([r[1] - t0 for r in rows], dtype=np.float64) / 1e3
en = np.array([r[2] - t0 for r in rows], dtype=np.float64) / 1e3
names = [r[0] for r in rows]
# take the last 40% of the trace (steady state)
lo = st[-1] * float(sys.argv[2]) if len(sys.argv) > 2 else st[-1] * 0.6
sel = st >= lo
s, e = st[sel], en[sel]
nm = [n for n, k in zip(names, sel) if k]
span = e.max() - s.min()
# union of busy intervals
order = np.argsort(s); cur_e = -1; busy = 0.0; cs = None
for i in order:
    if s[i] > cur_e:
        if cs is not None: busy += cur_e - cs
        cs = s[i]; cur_e = e[i]
    else:
        cur_e = max(cur_e, e[i])
busy += cur_e - cs
print(f"window {span/1e3:.2f} ms, busy(union) {busy/1e3:.2f} ms, sum of kernel durations {np.sum(e-s)/1e3:.2f} ms")
from collections import defaultdict
agg = defaultdict(float); cnt = defaultdict(int)
for n, a, b in zip(nm, s, e):
    k = n.split('(')[0][-40:]
    agg[k] += b - a; cnt[k] += 1
for k, v in sorted(agg.items(), key=lambda x: -x[1])[:16]:
    print(f"  {k:42s} n={cnt[k]:6d} total {v/1e3:8.2f} ms avg {v/cnt[k]:8.2f} us")
sw = np.array(['sweep_half_kernel' in n for n in nm])
if sw.any():
    ss, se = s[sw], e[sw]
    gaps = ss[1:] - se[:-1]
    print(f"sweep launches: {sw.sum()}, mean dur {np.mean(se-ss):.2f} us, median gap to next {np.median(gaps):.2f} us")
