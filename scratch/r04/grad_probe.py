import sys, json
sys.path.insert(0, '/root/repo')
import bench
from smnngp import _lib as L
ctx = L.default_context()
r = bench.measure_loss_grad(L, ctx, 16384, 3072, 4, "relu", 3, 1)
r.pop("roofline", None)
print(json.dumps(r, indent=1))
