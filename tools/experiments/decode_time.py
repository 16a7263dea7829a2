"""binauralDecode timings (harness size and 100 s) for the forms of the fused kernel:  python tools/experiments/decode_time.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools import bench_secondary as S
for wave in ("1", "0"):
    os.environ["EMAGLS_DECODE_WAVE"] = wave
    a, b = S.binaural_decode(), S.binaural_decode_long()
    print("wave", wave, "harness real/complex ms", a["real"]["ms"], a["complex"]["ms"], "frac", a["real"]["frac_of_hbm_peak"], a["complex"]["frac_of_hbm_peak"],
          "| 100 s real/complex ms", b["real"]["ms"], b["complex"]["ms"], "frac", b["real"]["frac_of_hbm_peak"], b["complex"]["frac_of_hbm_peak"], flush=True)
