#!/bin/bash
# fused overlap-save decode: register-FFT form / LDS-pass form / hipFFT passes
R=$GRAFT_REPO_ROOT; cd $R
timeout 900 python -m pytest tests -m gpu -q -x -k "decode or harness or mex" 2>&1 | tail -3
python - <<'P'
import json, os
from tools import bench_secondary as S
for tag, env in (("register FFT", {}), ("LDS passes", {"EMAGLS_DECODE_REGFFT": "0"}), ("hipFFT", {"EMAGLS_DECODE_FUSED": "0"})):
    pass
P
for mode in reg lds hipfft; do
  case $mode in reg) e="";; lds) e="EMAGLS_DECODE_REGFFT=0";; hipfft) e="EMAGLS_DECODE_FUSED=0";; esac
  env $e python -c "
from tools import bench_secondary as S
a, b = S.binaural_decode(), S.binaural_decode_long()
print('$mode', {k: a[k]['ms'] for k in ('real', 'complex')}, {k: b[k]['ms'] for k in ('real', 'complex')})" 2>/dev/null | tail -1
done
