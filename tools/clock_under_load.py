"""Shader clock and per-bin period of the resident sweep ALONE and NEXT TO three other 16-design batches that keep executing
(the steady state of bench.py): is the pipeline clock / power bound?

    python tools/clock_under_load.py

In-kernel stamps (EMAGLS_SWEEP_TIMING=1): s_memrealtime (100 MHz) and s_memtime (shader clock) at the start of every bin."""
import os
import sys
import threading

os.environ["EMAGLS_SWEEP_TIMING"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    import torch
    import bench
    from emagls_amd import Batch, Plan, _lib as L
    L.check(L.load().emagls_set_batch_max(16, None))

    def make_batch(seed0):
        plans = []
        for j in range(16):
            azi, zen, maz, mzn, hL, hR = bench.load_inputs(seed_offset=seed0 + j)
            p = Plan(L.KIND_EMAGLS, "complex", 4, 48000.0, 512, hL.shape[0], hL.shape[1], 0.042, 32)
            p.set_streams(1)
            p.set_hrir_grid(azi, zen)
            p.set_mic_grid(maz, mzn)
            p.set_hrirs(hL, hR)
            plans.append(p)
        b = Batch(plans)
        b.set_stream(torch.cuda.Stream().cuda_stream)
        b.set_side_stream(torch.cuda.Stream().cuda_stream)
        return plans, b

    streams_keep = []
    units = [make_batch(100 * i) for i in range(4)]
    for plans, b in units:
        for _ in range(3):
            b.execute()
        b.synchronize()

    def report(tag):
        plans, b = units[0]
        p = plans[0]
        info = p.info()
        P, k0 = info.num_pos_freqs, max(info.k_cut - 1, 1)
        raw = p.debug("sweep_timing", np.int64).reshape(P, 16)
        kb = np.arange(k0 + 2, P - 2)
        wall = (raw[kb + 1, 0] - raw[kb, 0]) * 0.01          # us
        cyc = (raw[kb + 1, 10] - raw[kb, 10]).astype(np.float64)
        mhz = cyc / wall
        print("%-28s bin period median %.2f us (p90 %.2f), sweep span %.0f us, shader clock median %.0f MHz (p10 %.0f, p90 %.0f)"
              % (tag, np.median(wall), np.percentile(wall, 90), (raw[P - 1, 0] - raw[k0, 0]) * 0.01, np.median(mhz),
                 np.percentile(mhz, 10), np.percentile(mhz, 90)))

    plans, b = units[0]
    b.execute()
    b.synchronize()
    report("alone")
    stop = threading.Event()

    def churn(i):
        pl, bb = units[i]
        while not stop.is_set():
            bb.execute()
            bb.synchronize()

    th = [threading.Thread(target=churn, args=(i,)) for i in (1, 2, 3)]
    for t in th:
        t.start()
    for it in range(6):
        b.execute()
        b.synchronize()
        report("next to 3 batches, run %d" % it)
    stop.set()
    for t in th:
        t.join()
    for pl, bb in units:
        bb.close()
        for p in pl:
            p.close()


if __name__ == "__main__":
    main()
