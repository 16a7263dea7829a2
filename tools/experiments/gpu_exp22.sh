#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "decode or harness or stage_prologue or config3" 2>&1 | tail -3
python - <<'P'
import json, os
from tools import bench_secondary as S
for fused in ("1", "0"):
    os.environ["EMAGLS_DECODE_FUSED"] = fused
    a, b = S.binaural_decode(), S.binaural_decode_long()
    print("fused", fused, {k: a[k]["ms"] for k in ("real", "complex")}, {k: b[k]["ms"] for k in ("real", "complex")})
P
