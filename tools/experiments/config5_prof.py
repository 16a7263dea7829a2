"""BASELINE config 5 (FromAtf: 16 384 ATF directions x 8 microphones, 2702 HRIR directions, 2048 taps): one subject and the batch
of 8 subjects executed a few times -- run under rocprofv3 --kernel-trace --stats.   python tools/experiments/config5_prof.py [single|batch]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "single"
    from tools.bench_secondary import _grids
    from emagls_amd import Batch, Plan, synth, _lib as L
    azi, zen, _, _ = _grids()
    atf, aazi, azen = synth.glasses_atfs(natf=16384, nmics=8, taps=256)

    def mk(j):
        hL, hR = synth.rigid_sphere_hrirs(azi, zen, seed=100 + j, head_radius=0.075 + 0.02 * j / 7)
        p = Plan(L.KIND_FROM_ATF, "real", 0, 48000.0, 2048, hL.shape[0], hL.shape[1], nmics=8, f_trans=2000.0, atf_taps=256, natf=16384)
        p.set_hrir_grid(azi, zen)
        p.set_hrirs(hL, hR)
        p.set_atfs(atf, aazi, azen)
        return p
    if what == "single":
        p = mk(0)
        for _ in range(6):
            p.execute()
        p.synchronize()
    else:
        plans = [mk(j) for j in range(8)]
        b = Batch(plans)
        for _ in range(6):
            b.execute()
        b.synchronize()


if __name__ == "__main__":
    main()
