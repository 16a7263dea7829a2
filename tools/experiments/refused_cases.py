"""The five job lists of tools/fuzz_jobs.py 40 7 that the build of that run refused (simulation order above 47; more than 27 orders
on the orthonormal route), on the current build."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests"); sys.path.insert(0, ROOT + "/tools")
import numpy as np
import fuzz_jobs as F
from emagls_amd._lib import EmaglsError
rng = np.random.default_rng(7)
cases = [F.draw(rng) for _ in range(40)]
for i in (3, 9, 22, 28, 33):
    c = dict(cases[i]); c["n"] = min(c["n"], 10)
    t = time.time()
    try:
        a, b = F.run(c)
        print(f"case {i} {c} -> list vs single calls {a:.2e}, vs oracle {b:.2e} ({time.time() - t:.1f} s)", flush=True)
    except EmaglsError as e:
        print(f"case {i} {c} -> refused: {str(e)[:200]}", flush=True)
    except Exception as e:
        print(f"case {i} {c} -> ERROR {type(e).__name__}: {str(e)[:300]}", flush=True)
