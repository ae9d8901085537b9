import sys, json
sys.path.insert(0, '/root/repo')
import bench
from smnngp import _lib as L
ctx = L.default_context()
r = bench.measure_predict(L, ctx, 16384, 3072, 4, "relu", 2048, 3, 1)
print(json.dumps({k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if kk != "roofline"}) for k, v in r.items()}, indent=1))
