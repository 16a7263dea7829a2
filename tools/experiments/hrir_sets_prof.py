"""rocprofv3 target: config 3's design as batches of HRIR sets on one geometry (tools/bench_secondary.config3_hrir_sets)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools import bench_secondary as S
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4
print(json.dumps(S.config3_hrir_sets(nb, 16, rounds=6)))
