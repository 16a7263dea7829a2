"""Summarise rocprofv3 --pmc passes (rocpd sqlite) per kernel: mean counter value per dispatch.

    python tools/pmc_summary.py <out.json> <out.md> <dir> [<dir> ...]

Each <dir> is the -d directory of one `rocprofv3 --kernel-trace --pmc <COUNTERS>` run of the same command.  Writes the
per-kernel means of every counter found, and for FETCH_SIZE / WRITE_SIZE the HBM bytes per launch
    bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024
(FETCH_SIZE doubled per the gfx950 correction in MI355X_MICROARCH.md: 128-byte requests are counted as 64 bytes),
plus `per_set`: the sum over all kernels of one design (dispatch counts normalised by the number of designs profiled)."""
import glob
import json
import os
import re
import sqlite3
import subprocess
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.csrc_hash import csrc_sha16 as _csrc_sha16  # noqa: E402


def demangle(n):
    try:
        d = subprocess.run(["c++filt", n.replace(".kd", "")], capture_output=True, text=True).stdout.strip()
    except Exception:
        d = n
    d = d.replace("(anonymous namespace)::", "").replace("emagls::", "").replace("void ", "")
    d = re.sub(r"\(.*", "", d)
    return d or n


LANES = int(os.environ.get("EMAGLS_PMC_LANES", "16"))   # grid.z of the stages before the sweep in the profiled command (a 32-design chunk runs them as two lane groups of 16)
SWEEP_DESIGNS = int(os.environ.get("EMAGLS_PMC_SWEEP_DESIGNS", "32"))   # designs one sweep launch of the profiled command covers
ONCE_PER_GROUP = "hrir_fft"   # a kernel every lane group launches exactly once per execute (counts the executes of the run)


def read(d):
    dbs = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)
    if not dbs:
        return {}
    cur = sqlite3.connect(dbs[0]).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    pe = [t for t in tabs if t.startswith("rocpd_pmc_event")]
    pi = [t for t in tabs if t.startswith("rocpd_info_pmc")]
    if not pe or not pi:
        print("no pmc tables in", dbs[0], tabs, file=sys.stderr)
        return {}
    pe, pi = pe[0], pi[0]
    picols = [r[1] for r in cur.execute(f"pragma table_info({pi})")]
    namecol = "name" if "name" in picols else ("symbol" if "symbol" in picols else picols[-1])
    q = (f"select s.kernel_name, i.{namecol}, e.value from {pe} e join {pi} i on e.pmc_id = i.id "
         f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id")
    acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))   # kernel -> counter -> dispatch-less sum
    cnt = defaultdict(lambda: defaultdict(int))
    # one row per (dispatch, counter instance/dimension): sum over instances, then average over dispatches
    q2 = (f"select s.kernel_name, i.{namecol}, d.id, sum(e.value), d.grid_size_z / d.workgroup_size_z, d.end - d.start "
          f"from {pe} e join {pi} i on e.pmc_id = i.id "
          f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id group by d.id, i.{namecol} order by d.id")
    out = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    sweeps = defaultdict(list)
    for kname, cname, _, val, gz, ns in cur.execute(q2):
        if "sweep_persist" in kname or "sweep_synth" in kname or "sweep_reg" in kname:      # (design = blockIdx.x there, not grid.z: the batch launches are the LAST ones of the run)
            sweeps[(kname, cname)].append((val, ns))
            continue
        if gz != LANES:
            continue          # (the single-design plan bench.py also runs: not part of the per-batch figures)
        out[kname][cname].append(val)
        dur[kname].append(ns)
    n_exec = max([len(v) for k, cs in out.items() if ONCE_PER_GROUP in k for v in cs.values()] or [1])
    n_sweeps = max(1, n_exec * LANES // SWEEP_DESIGNS)   # (batch executes of the run: the last launches are the batch's)
    for (kname, cname), lst in sweeps.items():
        lst = lst[-n_sweeps:]
        out[kname][cname] = [v for v, _ in lst]
        dur[kname] = [ns for _, ns in lst]
    res = {k: {c: (sum(v) / len(v), len(v)) for c, v in cs.items()} for k, cs in out.items()}
    for k in res:
        res[k]["avg_us"] = (sum(dur[k]) / len(dur[k]) / 1e3, len(dur[k]))
    return res


def main():
    out_json, out_md, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    merged = defaultdict(dict)
    for d in dirs:
        for k, cs in read(d).items():
            merged[k].update(cs)
    res = {"note": "means per dispatch from separate rocprofv3 --pmc passes; bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (FETCH_SIZE "
                   "doubled per the gfx950 correction of MI355X_MICROARCH.md; WRITE_SIZE matched the algorithmic bytes on sh_basis_kernel)",
           "designs_per_launch": LANES, "designs_per_sweep_launch": SWEEP_DESIGNS, "build": os.environ.get("EMAGLS_BUILD_TAG", "?"), "csrc_sha16": _csrc_sha16()}
    per_set = 0.0
    sweep_set = 0.0
    rows = []
    launches_per_batch = {}
    for k, cs in merged.items():
        name = demangle(k)
        e = {c: v for c, (v, n) in cs.items()}
        nd = max(n for _, n in cs.values())
        e["dispatches"] = nd
        if e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) > 0 and e.get("avg_us", 0) > 0:
            # gfx94x MfmaUtil formula (no gfx950 derived counters ship): busy cycles summed over SIMDs / (cycles x CUs x 4)
            e["mfma_util_pct"] = 100.0 * e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["avg_us"] * 1e-6 * 2.4e9 * 256 * 4)
            e["mfma_f64_tflops"] = e.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0) * 512 / (e["avg_us"] * 1e-6) / 1e12
        if "FETCH_SIZE" in e or "WRITE_SIZE" in e:
            e["fetch_kb"] = e.get("FETCH_SIZE", 0.0)
            e["write_kb"] = e.get("WRITE_SIZE", 0.0)
            e["bytes"] = int((2 * e["fetch_kb"] + e["write_kb"]) * 1024)
            launches_per_batch[name] = nd
            if "sweep_persist" in name or "sweep_synth" in name or "sweep_reg" in name:
                sweep_set += e["bytes"] / float(SWEEP_DESIGNS)   # (one launch per batch execute, SWEEP_DESIGNS designs each)
            else:
                per_set += e["bytes"] * nd
        base = name.split("<")[0]
        key = base if base in ("sweep_persist_kernel", "sweep_synth_kernel", "sweep_reg_kernel", "sweep_half_kernel", "dspace_g_kernel") else name
        res[key] = e
        rows.append((name, e))
    # batch executions in the run = dispatches of a kernel that runs once per batch
    n_exec = max([n for k, n in launches_per_batch.items() if k.startswith(ONCE_PER_GROUP)] or [1])
    per_set = per_set / (n_exec * LANES) + sweep_set
    res["per_set"] = {"bytes": int(per_set), "lane_group_executions": n_exec, "sweep_bytes_per_set": int(sweep_set),
                      "note": "sum over the lane-group dispatches of the run / (lane-group executions x %d designs) + the sweep launch's bytes / "
                              "the designs it covers" % LANES}
    with open(out_json, "w") as f:
        json.dump(res, f, indent=1)
    counters = sorted({c for _, e in rows for c in e if c not in ("dispatches", "fetch_kb", "write_kb", "bytes")})
    with open(out_md, "w") as f:
        f.write("| kernel | dispatches | " + " | ".join(counters) + " | HBM bytes / launch |\n|---|---|" + "---|" * (len(counters) + 1) + "\n")
        for name, e in sorted(rows, key=lambda r: -r[1].get("bytes", 0)):
            f.write(f"| `{name[:70]}` | {e['dispatches']} | " + " | ".join(f"{e.get(c, float('nan')):.4g}" for c in counters) +
                    f" | {e.get('bytes', 0) / 1e6:.1f} MB |\n")
        f.write(f"\nHBM traffic per filter set (all kernels): {per_set / 1e6:.0f} MB\n")


if __name__ == "__main__":
    main()
