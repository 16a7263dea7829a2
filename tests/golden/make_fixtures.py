#!/usr/bin/env python3
"""Convert the reference's shipped golden filter sets into one compressed .npz.

Run in the build container only (needs /root/reference):
    python tests/golden/make_fixtures.py

The .mat files under /root/reference/resources/ are the reference's own regression fixtures
(written by verifyEMagLs.m:203-227, compared at verifyEMagLs.m:152-200).  They are DATA (grids,
scalars, expected filter taps), not source.  They stay under the reference's non-commercial
academic licence (see /root/reference/LICENSE); this notice travels with the derived file.

Output: tests/golden/ref_fixtures.npz with keys  "<basis>_<method>/<variable>".
"""
import glob
import os
import sys

import numpy as np
import scipy.io as sio

SRC = "/root/reference/resources"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_fixtures.npz")
PREFIX = "HRIR_L2702_512samples_32channels_sh4_"


def main():
    out = {}
    files = sorted(glob.glob(os.path.join(SRC, PREFIX + "*.mat")))
    if not files:
        sys.exit("reference fixtures not found under " + SRC)
    grids_saved = False
    for f in files:
        tag = os.path.basename(f)[len(PREFIX):-4]  # e.g. real_eMagLS_woDC
        d = sio.loadmat(f)
        for k, v in d.items():
            if k.startswith("__"):
                continue
            if k in ("hrirGridAziRad", "hrirGridZenRad", "micGridAziRad", "micGridZenRad"):
                # identical in every file: keep one copy (checked below)
                key = "grid/" + k
                v = np.asarray(v, dtype=np.float64).ravel()
                if key in out:
                    # the real/complex files differ by 1 ulp (deg2rad on different MATLAB releases)
                    assert np.allclose(out[key], v, rtol=0, atol=1e-15), (f, k)
                else:
                    out[key] = v
                continue
            if v.size == 1:
                out[f"{tag}/{k}"] = np.float64(v.ravel()[0])
            else:
                out[f"{tag}/{k}"] = np.ascontiguousarray(v)
    # the 8-channel room IR used by testEMagLsFromAtfs.m:30,67 (input data for the render path)
    rir = sio.loadmat(os.path.join(SRC, "meetingRoom_leftLsp.mat"))
    out["meetingRoom/roomIRs_first4096"] = np.ascontiguousarray(rir["roomIRs"][:4096].astype(np.float32))
    out["meetingRoom/fs"] = np.float64(rir["fs"].ravel()[0])
    np.savez_compressed(DST, **out)
    print("wrote", DST, os.path.getsize(DST) / 1e6, "MB", len(out), "arrays")


if __name__ == "__main__":
    main()
