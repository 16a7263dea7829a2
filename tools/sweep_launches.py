"""Durations of the dominant kernel's launches in a rocprofv3 --kernel-trace run of bench.py, next to the figure the bench
measured live with HIP events (roofline.avg_launch_us in the JSON line of the same run).

    python tools/sweep_launches.py <rocprof dir> <bench stdout of the same run> [n_timed_launches]

bench.py averages the sweep launches of the FULL batches of its timed region (K // B launches of B designs each, B = designs per batch); in the trace
these are the last K // 8 launches with the full grid (the earlier full-grid launches belong to the set-up executes of the
batches and the warm-up; smaller grids to the single-design plan, the partial batch, the one-shot and parity checks)."""
import glob
import json
import os
import sqlite3
import sys


def main():
    d, out = sys.argv[1], sys.argv[2]
    line = [l for l in open(out) if l.startswith("{")][-1]
    js = json.loads(line)
    bsz = int(js.get("config", {}).get("designs_per_batch", 8))
    n_timed = int(sys.argv[3]) if len(sys.argv) > 3 else max(js["steps"] // bsz, 1)
    db = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    kname = js.get("roofline", {}).get("kernel", "sweep_persist_kernel")
    allrows = [(n, s, e, gx) for n, s, e, gx in cur.execute(f"select s.kernel_name, d.start, d.end, d.grid_size_x / d.workgroup_size_x from {kd} d "
                                                            f"join {ks} s on d.kernel_id=s.id order by d.start") if ("sweep_persist" in n or "sweep_synth" in n or "sweep_reg" in n)]
    print(f"command: python3 bench.py --steps {js['steps']} --warmup {js['warmup']} --no-secondary   ->  {js['value']:.1f} {js['unit']}")
    if "sweep_reg" in kname:
        # round 5: the timed region is ONE job list (chunks of `bsz` designs, the partial chunk last); its chunks' sweeps are the last
        # launches of the register-resident kernel in the trace (set-up runs and warm-up come before; launches of up to 8 designs take
        # the slab kernel)
        rows = [(s, e, g) for n, s, e, g in allrows if "sweep_reg" in n]
        n_chunks = -(-js["steps"] // bsz)
        last_rows = rows[-n_chunks:]
        big = max(g for _, _, g in last_rows)
        last = [(e - s) / 1e3 for s, e, g in last_rows if g == big]
        dur = [(e - s) / 1e3 for s, e, _ in rows]
        print(f"{kname} launches in the trace: {len(dur)}; all: avg {sum(dur) / len(dur):.1f} us, min {min(dur):.1f}, max {max(dur):.1f}")
        print(f"the {len(last)} launches of the timed region's largest chunks ({big} workgroups each; the last launches of the trace): "
              + ", ".join(f"{x:.1f}" for x in last) + f" us; avg {sum(last) / len(last):.1f} us")
    else:
        rows = [(s, e, gx) for n, s, e, gx in allrows]
        dur = [(e - s) / 1e3 for s, e, _ in rows]
        gmax = max(g for _, _, g in rows)
        full = [(e - s) / 1e3 for s, e, g in rows if g == gmax]
        print(f"sweep launches in the trace: {len(dur)}; all: avg {sum(dur) / len(dur):.1f} us, min {min(dur):.1f}, max {max(dur):.1f}")
        print(f"launches with the full grid of {gmax} workgroups ({bsz} designs): {len(full)}; avg {sum(full) / len(full):.1f} us")
        last = full[-n_timed:]
        print(f"the {n_timed} full-batch launches of the timed region (the last ones): " + ", ".join(f"{x:.1f}" for x in last) + f" us; avg {sum(last) / len(last):.1f} us")
    r = js["roofline"]
    print(f"bench.py, HIP events on the batch streams, same run: roofline.avg_launch_us = {r['avg_launch_us']:.1f} us "
          f"(achieved {r['achieved']:.0f} {r['unit']}, frac {r['frac']:.3f})")


if __name__ == "__main__":
    main()
