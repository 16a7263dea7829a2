"""Per-kernel summary of rocprofv3 passes of ONE command (rocpd sqlite): dispatches, average duration and -- from the --pmc passes
of FETCH_SIZE / WRITE_SIZE -- HBM bytes per launch, (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (the gfx950 correction of
MI355X_MICROARCH.md), and the bandwidth that makes against the 8 TB/s peak.

    python tools/pmc_simple.py <out.md> <out.json> <dir> [<dir> ...]     (a kernel-trace directory and one per counter)"""
import glob
import json
import os
import sqlite3
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import demangle  # noqa: E402


def read(d):
    dbs = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
    if not dbs:
        return {}, {}
    cur = sqlite3.connect(dbs[0]).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    dur = defaultdict(list)
    for name, ns in cur.execute(f"select s.kernel_name, d.end - d.start from {kd} d join {ks} s on d.kernel_id = s.id"):
        dur[name].append(ns)
    cnt = defaultdict(lambda: defaultdict(list))
    pe = [t for t in tabs if t.startswith("rocpd_pmc_event")]
    pi = [t for t in tabs if t.startswith("rocpd_info_pmc")]
    if pe and pi:
        picols = [r[1] for r in cur.execute(f"pragma table_info({pi[0]})")]
        namecol = "name" if "name" in picols else ("symbol" if "symbol" in picols else picols[-1])
        q = (f"select s.kernel_name, i.{namecol}, d.id, sum(e.value) from {pe[0]} e join {pi[0]} i on e.pmc_id = i.id "
             f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id group by d.id, i.{namecol}")
        try:
            for name, c, _, v in cur.execute(q):
                cnt[name][c].append(v)
        except sqlite3.Error:
            pass
    return dur, cnt


def main():
    out_md, out_json, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    durs, cnts = {}, defaultdict(dict)
    for d in dirs:
        du, cn = read(d)
        if du and not cn:
            durs = du    # the kernel-trace pass without counters: undisturbed durations
        elif du and not durs:
            durs = du
        for k, cs in cn.items():
            for c, v in cs.items():
                cnts[k][c] = sum(v) / len(v)
    res = {}
    for k, v in durs.items():
        e = {"dispatches": len(v), "avg_us": sum(v) / len(v) / 1e3}
        e.update(cnts.get(k, {}))
        if "FETCH_SIZE" in e or "WRITE_SIZE" in e:
            e["bytes"] = int((2 * e.get("FETCH_SIZE", 0.0) + e.get("WRITE_SIZE", 0.0)) * 1024)
            e["GBps"] = e["bytes"] / (e["avg_us"] * 1e-6) / 1e9
            e["frac_of_hbm_peak"] = e["GBps"] / 8000.0
        res[demangle(k)] = e
    with open(out_json, "w") as f:
        json.dump(res, f, indent=1)
    with open(out_md, "w") as f:
        f.write("| kernel | dispatches | avg us | total us | HBM MB / launch | GB/s | of 8 TB/s |\n|---|---|---|---|---|---|---|\n")
        for k, e in sorted(res.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["dispatches"]):
            f.write(f"| `{k[:80]}` | {e['dispatches']} | {e['avg_us']:.1f} | {e['avg_us'] * e['dispatches']:.0f} | "
                    f"{e.get('bytes', 0) / 1e6:.1f} | {e.get('GBps', float('nan')):.0f} | {e.get('frac_of_hbm_peak', float('nan')):.3f} |\n")


if __name__ == "__main__":
    main()
