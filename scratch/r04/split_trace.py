"""C4-shaped smn_spr_loss, three calls, for a kernel trace (split build on unless SPLIT=0)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smnngp import _lib as L
ctx = L.Context(0)
ctx.call("smn_debug_split_build", int(os.environ.get("SPLIT", "1")))
n, d = 16384, 3072
rng = np.random.default_rng(1)
x = ctx.to_device(rng.standard_normal((n, d)).astype(np.float32)); y = ctx.to_device(rng.standard_normal((n, 1)).astype(np.float32))
for _ in range(3):
    lp, q, ld, info = C.c_double(), C.c_double(), C.c_double(), C.c_int()
    ctx.call("smn_spr_loss", x.dcode, L.NET_MLP, L.ACT["relu"], 4, 1.0, 0.3, 1.0, x.ptr, n, x.ld, d, y.ptr, 1e-2, 0.0, 1.0, C.byref(lp), C.byref(q), C.byref(ld), C.byref(info))
print(lp.value, info.value)
