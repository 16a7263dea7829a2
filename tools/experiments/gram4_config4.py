"""The Gram tile on the two FP64 MFMA shapes at config 4's sizes (S = 529 ... 2025: many tiles per design):
    python tools/experiments/gram4_config4.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = """
import json, sys, numpy as np
sys.path.insert(0, %r)
from tools import bench_secondary as S
r0 = float(sys.argv[1])
print(json.dumps(S.config4(np.linspace(r0 - 0.002, r0, 8), reps=6)))
""" % ROOT

for r0 in (0.05, 0.075, 0.10):
    for mode in ("0", "1"):
        env = dict(os.environ, EMAGLS_GRAM_MFMA4=mode)
        out = subprocess.run([sys.executable, "-c", CHILD, str(r0)], capture_output=True, text=True, env=env, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(f"r0 {r0} mode {mode}: FAILED", out.stderr[-400:])
            continue
        d = json.loads(line[-1])
        print(f"r ~ {100 * r0:.1f} cm (sim order {d['sim_order']}), 8 radii, EMAGLS_GRAM_MFMA4={mode}: {d['ms_per_batch']:8.3f} ms per batch, {d['filter_sets_per_s']:7.1f} sets/s", flush=True)
