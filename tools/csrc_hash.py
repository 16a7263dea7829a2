"""A stamp of the kernel sources a profile was taken on: sha256 over emagls_amd/csrc/* (names and contents, sorted).  The GPU box has
no .git, so the stamp is computed from the tree itself; the profile summaries (tools/pmc_summary.py, tools/secondary_traffic.py)
store it, and bench.py / tools/bench_secondary.py mark a roofline block "stale" when the sources have changed since."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha16():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "emagls_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp", ".h")):
            h.update(name.encode())
            with open(os.path.join(d, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def stale(stamp):
    """True when a profile stamped `stamp` (None: never stamped) was taken on other kernel sources than the tree's."""
    return stamp != csrc_sha16()


if __name__ == "__main__":
    print(csrc_sha16())
