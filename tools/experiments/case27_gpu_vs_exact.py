"""Case 27 family on the GPU: the library's least-squares rows W(k,:) of the lowest bins (plan buffer "W") against the 40-digit rows
of tests/golden/case27_truth.npz and against the oracle's FP64 rows, 32 / 36 / 42 / 48 microphones."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import emagls_amd as E  # noqa: E402
from emagls_amd import Plan, _lib as L  # noqa: E402
from oracle import emagls_oracle as O  # noqa: E402
import shape_cases as SC  # noqa: E402
import importlib.util  # noqa: E402

spec = importlib.util.spec_from_file_location("mk", os.path.join(ROOT, "tests", "golden", "make_case27_truth.py"))
mk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mk)
T = np.load(mk.OUT)
nrm = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
for M in (32, 36, 42, 48):
    azi, zen, hL, hR, ma, mz = mk.inputs(M)
    p = Plan(L.KIND_EMAGLS2, mk.basis, mk.N, mk.fs, mk.ln, hL.shape[0], hL.shape[1], mk.r, M)
    p.set_hrir_grid(azi, zen)
    p.set_mic_grid(ma, mz)
    p.set_hrirs(hL, hR)
    p.execute()
    wl, wr = p.get_filters()
    info = p.info()
    P = info.num_pos_freqs
    W = p.debug("W", np.complex128).reshape(2, P, -1)[:, :, :M]
    o = O.getEMagLs2Filters(hL, hR, azi, zen, mk.r, ma, mz, mk.N, mk.fs, mk.ln, mk.basis)
    print(f"M={M}: filters GPU vs oracle: rel L {SC.rel(wl, o[0]):.2e} R {SC.rel(wr, o[1]):.2e}; |W| of all bins = {np.linalg.norm(W[0]):.3e}")
    for k in mk.BINS:
        xl, xr = T[f"m{M}_k{k}_wl"], T[f"m{M}_k{k}_wr"]
        ol, orr = T[f"m{M}_k{k}_wl_fp64"], T[f"m{M}_k{k}_wr_fp64"]
        gl, gr = W[0, k - 1], W[1, k - 1]
        print(f"   bin {k}: GPU vs 40-digit L {nrm(gl, xl):.2e} R {nrm(gr, xr):.2e} | oracle FP64 vs 40-digit L {nrm(ol, xl):.2e} R {nrm(orr, xr):.2e} | GPU vs oracle FP64 L {nrm(gl, ol):.2e} R {nrm(gr, orr):.2e}")
    p.close()
