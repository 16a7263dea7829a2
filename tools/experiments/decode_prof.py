"""rocprofv3 target: a few binauralDecode calls on device buffers (fused overlap-save kernel against the hipFFT passes)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools import bench_secondary as S
print(json.dumps(S.binaural_decode_long(reps=3)))
