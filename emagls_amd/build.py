"""Build the HIP library in-tree:  emagls_amd/lib/libemagls.so  (gfx950 only).

    python -m emagls_amd.build [-j N] [--force]

hipcc cross-compiles without a GPU.  Objects are rebuilt when a source or header is newer.
"""
from __future__ import annotations

import argparse
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libemagls.so")
SOURCES = ["sh_basis.hip", "modal.hip", "fft.hip", "gram_chol.hip", "factor.hip", "gramroute.hip", "sweep.hip", "sweep_persist.hip", "sweep_synth.hip", "sweep_reg.hip", "dspace.hip", "atf.hip", "decode.hip", "render.hip", "render_api.hip", "emash.hip", "wide.hip", "wide_array.hip", "microbench.hip",
           "capi.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result", "-Wno-unused-value"]


def _newest_header():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "emagls.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src, force):
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    srcp = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj) and os.path.getmtime(obj) > os.path.getmtime(srcp)
            and os.path.getmtime(obj) > _newest_header()):
        return obj, False
    cmd = [HIPCC, *FLAGS, "-c", srcp, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout[-4000:], r.stderr[-8000:]))
    if r.stderr.strip():
        sys.stderr.write(r.stderr[-3000:])
    return obj, True


def build(jobs=4, force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        res = list(ex.map(lambda s: _compile(s, force), SOURCES))
    objs = [o for o, _ in res]
    if any(c for _, c in res) or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-L/opt/rocm/lib", "-lhipfft", "-pthread",
               "-Wl,-rpath,/opt/rocm/lib"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-4000:])
        if verbose:
            print("built", LIB)
    elif verbose:
        print("up to date:", LIB)
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("-j", type=int, default=4)
    ap.add_argument("--force", action="store_true")
    a = ap.parse_args()
    build(a.j, a.force)
