"""Config 4 lane batches on the two forms of the synthesising sweep (slab form: EMAGLS_SWEEP_REG=0 / default up to 8 designs;
register-resident form: EMAGLS_SWEEP_REG=2) at 8 and 16 radii per batch, r ~ 5 cm and r ~ 10 cm.
    python tools/experiments/config4_forms.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = """
import json, sys, numpy as np
sys.path.insert(0, %r)
from tools import bench_secondary as S
from emagls_amd import _lib as L
lib = L.load()
import ctypes as C; lib.emagls_set_batch_max(32, C.byref(C.c_int(0)))
n = int(sys.argv[1]); r0 = float(sys.argv[2])
radii = np.linspace(r0 - 0.002, r0, n)
print(json.dumps(S.config4(radii, reps=4)))
""" % ROOT


def main():
    for r0 in (0.05, 0.10):
        for n in (8, 16):
            for mode in ("0", "1", "2"):
                if n == 16 and mode == "1":
                    continue   # (the default takes the register-resident form from 9 designs on: the same as 2)
                env = dict(os.environ, EMAGLS_SWEEP_REG=mode)
                out = subprocess.run([sys.executable, "-c", CHILD, str(n), str(r0)], capture_output=True, text=True, env=env, timeout=600)
                line = [l for l in out.stdout.splitlines() if l.startswith("{")]
                if not line:
                    print(f"r0 {r0} n {n} mode {mode}: FAILED", out.stderr[-400:])
                    continue
                d = json.loads(line[-1])
                print(f"r ~ {100 * r0:.0f} cm, {n:2d} radii per batch, EMAGLS_SWEEP_REG={mode}: {d['ms_per_batch']:8.3f} ms per batch, {d['ms_per_batch'] / n:6.3f} ms per design, "
                      f"{d['filter_sets_per_s']:7.1f} sets/s  (sim order {d['sim_order']})", flush=True)


if __name__ == "__main__":
    main()
