import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smnngp import _lib as L
ctx = L.Context(0)
for dt in (np.float32, np.float64):
    n = 2048
    rng = np.random.default_rng(0)
    a = rng.standard_normal((n, n)); a = (a @ a.T / n + np.eye(n)).astype(dt)
    d = ctx.to_device(a)
    info, ld = C.c_int(), C.c_double()
    ctx.call("smn_cholesky", L.dtype_code(dt), d.ptr, n, n, n, 0, 0.0, 0.0, C.byref(info), C.byref(ld))
    ctx.synchronize()
    print(np.dtype(dt).name, "info", info.value, "logdet", ld.value, flush=True)
