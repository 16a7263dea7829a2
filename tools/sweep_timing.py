"""Per-bin critical path of the persistent sweep from its in-kernel wall-clock stamps (EMAGLS_SWEEP_TIMING=1; 100 MHz counter).

    python tools/sweep_timing.py [designs]        (config 3; `designs` plans in one lane batch, default 1)

Stamps of workgroup 1 of a design, per bin kb (sweep_persist.hip, PSTAMP): 0 communication wave starts on bin kb, 6 first
poll of hop 1 answered, 1 hop 1 done (totals published), 2 hop 2 done (totals in LDS), 3 p phase starts, 4 partial phase
starts, 5 partials published."""
import os
import sys

os.environ["EMAGLS_SWEEP_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    import bench
    from emagls_amd import Batch, Plan, _lib as L
    if n > 8:
        L.check(L.load().emagls_set_batch_max(n, None))
    plans = []
    for j in range(n):
        azi, zen, maz, mzn, hL, hR = bench.load_inputs(seed_offset=j)
        p = Plan(L.KIND_EMAGLS, "complex", 4, 48000.0, 512, hL.shape[0], hL.shape[1], 0.042, 32)
        p.set_streams(1)
        p.set_hrir_grid(azi, zen)
        p.set_mic_grid(maz, mzn)
        p.set_hrirs(hL, hR)
        plans.append(p)
    b = Batch(plans) if n > 1 else None
    if b and os.environ.get("SWEEP_TIMING_FORK"):   # the lone chunk of a job list: one lane group (EMAGLS_BATCH_GROUPS=1), stages forked
        b.set_streams(int(os.environ["SWEEP_TIMING_FORK"]))
    for _ in range(4):
        (b.execute() if b else plans[0].execute())
    (b.synchronize() if b else plans[0].synchronize())
    info = plans[0].info()
    P, k0 = info.num_pos_freqs, max(info.k_cut - 1, 1)
    for j, p in [(0, plans[0]), (n - 1, plans[-1])][:min(n, 2)]:
        t = p.debug("sweep_timing", np.int64).reshape(P, 16).astype(np.float64) * 0.01   # microseconds
        kb = np.arange(k0 + 2, P - 2)
        if info.sweep_form == 3:   # sweep_reg.hip: 7 iteration starts (operand synthesis), 0 poll starts, 2 totals in LDS, 3 B2 passed, 4 B3 passed, 5 published
            rows = {
                "bin period (stamp 7 -> next bin's stamp 7)": t[kb + 1, 7] - t[kb, 7],
                "operand synthesis (7->0)": t[kb, 0] - t[kb, 7],
                "wait for the other workgroups' partials + sum (0->2)": t[kb, 2] - t[kb, 0],
                "B1 + M phase + B2 (2->3)": t[kb, 3] - t[kb, 2],
                "p phase (own units) + Bp (3->8)": t[kb, 8] - t[kb, 3],
                "t + partial + wave reduction + B3 (8->4)": t[kb, 4] - t[kb, 8],
                "workgroup sum + publish (4->5)": t[kb, 5] - t[kb, 4],
            }
            print("design %d of %d (register-resident form): %d swept bins, sweep span %.1f us, one XCD: %s" % (j, n, P - k0, t[P - 1, 5] - t[k0, 7], bool(p.debug("sweep_timing", np.int64)[15])))
            for name, v in rows.items():
                print("  %-56s median %6.2f  mean %6.2f  p90 %6.2f us" % (name, np.median(v), v.mean(), np.percentile(v, 90)))
            if os.environ.get("SWEEP_TIMING_WINDOWS"):   # the phases over the run of the launch (what runs NEXT to it ends somewhere inside)
                nwin = int(os.environ["SWEEP_TIMING_WINDOWS"])
                print("  windows of the swept bins (us from the first bin: mean of period / synthesis / wait / M / p / products / publish):")
                for w in np.array_split(np.arange(len(kb)), nwin):
                    print("    %7.0f .. %7.0f us: " % (t[kb[w[0]], 7] - t[k0, 7], t[kb[w[-1]], 7] - t[k0, 7]) + "  ".join("%5.2f" % v[w].mean() for v in rows.values()))
            continue
        rows = {
            "bin period (stamp 0 -> next bin's stamp 0)": t[kb + 1, 0] - t[kb, 0],
            "comm start -> first poll answered (0->6)": t[kb, 6] - t[kb, 0],
            "hop 1: poll answered -> totals published (6->1)": t[kb, 1] - t[kb, 6],
            "hop 1 total (0->1)": t[kb, 1] - t[kb, 0],
            "hop 2 (1->2)": t[kb, 2] - t[kb, 1],
            "B1 + M phase + B2 (2->3)": t[kb, 3] - t[kb, 2],
            "p phase + B3 (3->4)": t[kb, 4] - t[kb, 3],
            "partial phase + publish (4->5)": t[kb, 5] - t[kb, 4],
            "publish -> next bin's comm start (5->0')": t[kb + 1, 0] - t[kb, 5],
        }
        print("design %d of %d: %d swept bins, sweep span %.1f us, one XCD: %s" % (j, n, P - k0, t[P - 1, 5] - t[k0, 0], bool(t[0, 15] if False else p.debug("sweep_timing", np.int64)[15])))
        for name, v in rows.items():
            print("  %-52s median %6.2f  mean %6.2f  p90 %6.2f us" % (name, np.median(v), v.mean(), np.percentile(v, 90)))


if __name__ == "__main__":
    main()
