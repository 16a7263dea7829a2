"""rocprofv3 target: the SH-basis assembly at D = 2^20, N = 19 (the launch bench.py's sh_basis_roofline times)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from emagls_amd import _lib as L
print(json.dumps(bench.sh_basis_roofline(L.load())))
