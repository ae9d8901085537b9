import sys, json
sys.path.insert(0, '/root/repo')
import bench
from smnngp import _lib as L
ctx = L.default_context()
print(json.dumps(bench.measure_sweep(L, ctx, 2048, 256, 16, 4, "relu"), indent=1))
print(json.dumps(bench.measure_small_n(L, ctx, 245, 6, 2, "relu", 256), indent=1))
