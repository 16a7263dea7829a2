"""A lane batch of `n` config-3 designs executed `reps` times (for rocprofv3 passes over the resident sweep alone).

    python tools/experiments/sweep_only.py [n=32] [reps=6]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
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
    for _ in range(reps):
        (b.execute() if b else plans[0].execute())
        (b.synchronize() if b else plans[0].synchronize())
    print("sweep form", plans[0].info().sweep_form, "designs", n)


if __name__ == "__main__":
    main()
