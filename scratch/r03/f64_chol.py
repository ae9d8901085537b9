"""r03: wall time of smn_cholesky at N = 8192 in fp64 and N = 4096 in fp32 (events around 10 factorisations)."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smnngp import _lib as L
ctx = L.Context(0)
for dt, n in ((np.float64, 8192), (np.float32, 4096), (np.float64, 2048)):
    rng = np.random.default_rng(0)
    g = rng.standard_normal((n, 256)); a = (g @ g.T / 256 + np.eye(n)).astype(dt)
    pass
    info, ld = C.c_int(), C.c_double()
    best = 1e9
    for it in range(6):
        d2 = ctx.to_device(a); ctx.synchronize()
        t = time.perf_counter()
        ctx.call("smn_cholesky", L.dtype_code(dt), d2.ptr, n, n, n, 0, 0.0, 0.0, C.byref(info), C.byref(ld))
        ctx.synchronize()
        best = min(best, time.perf_counter() - t)
    print("chol", np.dtype(dt).name, n, "%.3f ms" % (best * 1e3), "info", info.value, "logdet %.6f" % ld.value, flush=True)
