"""Shape sweep on the GPU box: all cases of tests/shape_cases.py.
Run:  python tools/fuzz_shapes.py --driver   (prints one line per case; a crash names the case and the sweep resumes behind it)"""
import sys, time, traceback
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from emagls_amd._lib import EmaglsError
from shape_cases import CASES, run


def driver():
    """Children run ranges of cases; after a crash the next child resumes behind the crashed case."""
    import subprocess
    nxt = 0
    while nxt < len(CASES):
        p = subprocess.run([sys.executable, __file__, str(nxt), str(len(CASES))], capture_output=True, text=True, timeout=1500)
        sys.stdout.write(p.stdout)
        started = [int(l.split()[1]) for l in p.stdout.splitlines() if l.startswith("case ")]
        if p.returncode == 0:
            break
        last = started[-1] if started else nxt
        print(f"   -> CRASH in case {last} (rc {p.returncode})\n{p.stderr[-1500:]}", flush=True)
        nxt = last + 1


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--driver":
    driver()
elif __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    last = int(sys.argv[2]) if len(sys.argv) > 2 else len(CASES)
    bad = 0
    for i in range(first, min(last, len(CASES))):
        c = CASES[i]
        print(f"case {i} {c} ...", flush=True)
        t = time.time()
        try:
            e = run(c)
            tag = "ok" if e < 1e-6 else "MISMATCH"
            bad += e >= 1e-6
            print(f"   -> {tag} rel={e:.2e}  ({time.time() - t:.1f} s)", flush=True)
        except EmaglsError as ex:
            print(f"   -> refused: {ex}", flush=True)
        except Exception:
            bad += 1
            print("   -> EXCEPTION\n" + traceback.format_exc(), flush=True)
    print("bad cases:", bad)
