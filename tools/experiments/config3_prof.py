"""BASELINE config 3 (em32, N = 4, 2702 directions, 512 taps): one design and a lane batch of 8 executed a few times, each batch
alone on the GPU (no other batch in flight: the per-kernel durations are the undisturbed ones) -- run under rocprofv3
--kernel-trace --stats.   python tools/experiments/config3_prof.py [single|batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "batch"
    import bench
    from emagls_amd import Batch, Plan, _lib as L

    def mk(j):
        azi, zen, maz, mzn, hL, hR = bench.load_inputs(j)
        p = Plan(L.KIND_EMAGLS, "complex", 4, 48000.0, 512, hL.shape[0], hL.shape[1], 0.042, 32)
        p.set_hrir_grid(azi, zen)
        p.set_mic_grid(maz, mzn)
        p.set_hrirs(hL, hR)
        return p
    if what == "single":
        p = mk(0)
        for _ in range(6):
            p.execute()
            p.synchronize()
    else:
        b = Batch([mk(j) for j in range(8)])
        for _ in range(6):
            b.execute()
            b.synchronize()


if __name__ == "__main__":
    main()
