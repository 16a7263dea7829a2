"""Turn the outputs of scratch/final_run.sh (gpurun_out/) into the tracked summaries under profiles/."""
import sqlite3, json, collections, os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(R, "gpurun_out")
def pmc(db, counter):
    con = sqlite3.connect(db); cur = con.cursor()
    agg = collections.defaultdict(list)
    for n, c, val in cur.execute("select name, counter_name, counter_value from pmc_events"):
        if c == counter: agg[n].append(val)
    return {k: (len(x), sum(x) / len(x)) for k, x in agg.items()}
f = pmc(os.path.join(G, "pmc_fetch2/pmc_results.db"), "FETCH_SIZE")
w = pmc(os.path.join(G, "pmc_write2/pmc_results.db"), "WRITE_SIZE")
keys = sorted(set(f) | set(w), key=lambda k: -(f.get(k, (0, 0))[1] + w.get(k, (0, 0))[1]))
rows = [(k, f.get(k, (0, 0))[0], f.get(k, (0, 0))[1], w.get(k, (0, 0))[1]) for k in keys[:12]]
note = {'sweep_persist': '535 MB (471 bins x 1.14 MB: G_k 1.08 MB + M_k + |H|)', 'dspace_g': '518 MB out, 11 MB of real order terms in (re-read by each of the 8 bin chunks)',
        'sh_basis_kernel<false>': '3.37 GB out (D=2^20, N=19)', 'gram_mfma': '8.8 MB in (real conj(Y) once per K split), 20 MB of split-K partials out',
        'factor_qr_kernel<double, 32, 13': 'Tn 1.8 MB in (real), reflectors 7 MB out (Householder-route bins only)',
        'hrir_fft': '5.5 MB in, 24 MB out', 'hy_partial': 'Yc 8.8 MB (real) + Hc 3.7 MB in'}
md = ["# Round 1 PMC counters, end-of-round build (separate rocprofv3 --pmc passes: FETCH_SIZE, then WRITE_SIZE)", "",
      "`rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 bench.py --steps 8 --warmup 2 --concurrent 1 --batch 1 --no-cpu-baseline` (and the same with WRITE_SIZE).",
      "Values are KB per dispatch as reported; per the MI355X guide FETCH_SIZE under-counts wide coalesced streaming reads by 2x on gfx950 (the `x2` column), WRITE_SIZE is uncalibrated.", "",
      "| kernel | dispatches | FETCH_SIZE KB (mean) | x2 | WRITE_SIZE KB (mean) | algorithmic bytes/launch |", "|---|---|---|---|---|---|"]
for k, n, fk, wk in rows:
    nt = next((v for kk, v in note.items() if kk in k), '')
    md.append(f"| `{k[:70]}` | {n} | {fk:.1f} | {2 * fk:.1f} | {wk:.1f} | {nt} |")
open(os.path.join(R, "profiles/r01_pmc_traffic.md"), "w").write("\n".join(md) + "\n")
sp = next(r for r in rows if 'sweep_persist' in r[0]); dg = next(r for r in rows if 'dspace_g' in r[0])
pj = os.path.join(R, "profiles/pmc_traffic.json")
j = json.load(open(pj))
j["sweep_persist_kernel"] = {"fetch_kb": sp[2], "write_kb": sp[3], "bytes": int((2 * sp[2] + sp[3]) * 1024)}
j["dspace_g_kernel"] = {"fetch_kb": dg[2], "write_kb": dg[3], "bytes": int((2 * dg[2] + dg[3]) * 1024)}
json.dump(j, open(pj, "w"), indent=1)
# single-design kernel stats
out = subprocess.run([sys.executable, os.path.join(R, "scratch/prof_md.py"), os.path.join(G, "prof_single/bench_results.db"),
                      "Round 1, end-of-round build: one design in flight (persistent sweep)",
                      "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 8 --warmup 2 --concurrent 1 --batch 1 --no-cpu-baseline --no-sh-roofline"],
                     capture_output=True, text=True).stdout
open(os.path.join(R, "profiles/r01_c_final_kernel_stats.md"), "w").write(out)
# batch kernel stats
con = sqlite3.connect(os.path.join(G, "prof_batch/bench_results.db")); cur = con.cursor()
rows = list(cur.execute("select name,start,end,grid_z,grid_x from kernels order by start"))
agg = collections.defaultdict(lambda: [0, 0.0])
for n, s, e, gz, gx in rows:
    if gz == 8:   # launches that cover 8 designs
        agg[n][0] += 1; agg[n][1] += (e - s) / 1e3
nb = agg[[k for k in agg if 'filter_epilogue' in k][0]][0]
sw = [(n, s, e) for n, s, e, gz, gx in rows if 'sweep_persist' in n][-nb:]   # the batch sweeps are the last nb sweep launches
for n, s, e in sw:
    agg[n][0] += 1; agg[n][1] += (e - s) / 1e3
o = ["# Round 1, end-of-round build: a batch of 8 designs in lane mode (grid.z = design), one batch in flight", "",
     "`rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 32 --warmup 2 --concurrent 8 --batch 8 --no-cpu-baseline --no-sh-roofline`", "",
     f"Only the launches that cover 8 designs are listed ({nb} batch executes in the trace). Durations in microseconds.", "",
     "| kernel | launches / batch | avg us | us / batch |", "|---|---|---|---|"]
tot = 0
for k, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1]):
    per = t / nb; tot += per
    o.append(f"| `{k[:100]}` | {c / nb:.1f} | {t / c:.2f} | {per:.1f} |")
o.append(f"\nSum: {tot:.0f} us per batch of 8 = {tot / 8:.0f} us per design (kernels back to back on one stream).")
open(os.path.join(R, "profiles/r01_d_batch_kernel_stats.md"), "w").write("\n".join(o) + "\n")
import shutil; shutil.copy(os.path.join(G, "bench_final.json"), os.path.join(R, "profiles/r01_bench.json"))
print("\n".join(o[6:26])); print(o[-1]); print(j["sweep_persist_kernel"], j["dspace_g_kernel"])
# parity report lines printed by the GPU tests (reference assertAllClose metrics and relative errors)
pl = os.path.join(G, "final_parity.log")
if os.path.exists(pl):
    lines = open(pl).read().strip().splitlines()
    tests = open(os.path.join(G, "final_tests.log")).read().strip().splitlines()[-1]
    open(os.path.join(R, "profiles/r01_parity.md"), "w").write(
        "# Round 1 GPU parity report (python -m pytest tests -m gpu -q -rP on the MI355X box)\n\n" + tests + "\n\n```\n" + "\n".join(lines) + "\n```\n")
