"""The 64-capsule eMagLS2 design of tools/bench_secondary.em64 on its own (for rocprofv3 --kernel-trace --stats)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tools import bench_secondary as B  # noqa: E402

print(json.dumps(B.em64(reps=int(sys.argv[1]) if len(sys.argv) > 1 else 3)))
