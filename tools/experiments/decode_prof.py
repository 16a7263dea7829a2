"""rocprofv3 target: six binauralDecode calls on device buffers (100 s x 25 channels, 512 taps).
    python tools/experiments/decode_prof.py [real|complex]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools import bench_secondary as S
kind = sys.argv[1] if len(sys.argv) > 1 else "real"
print(json.dumps(S.binaural_decode_long(reps=3, kinds=(kind,))))
