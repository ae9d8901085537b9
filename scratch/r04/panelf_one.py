"""One factorisation per (dtype, n) for a kernel trace: SMN_PANEL_LEAF=1|2 python scratch/r04/panelf_one.py"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smnngp import _lib as L
ctx = L.Context(0)
for dt, n in [(np.float32, 8192), (np.float64, 4096)]:
    rng = np.random.default_rng(3)
    g = rng.standard_normal((n, 64)).astype(dt)
    a = (g @ g.T / 64 + np.diag(rng.uniform(1.0, 2.0, n))).astype(dt)
    for rep in range(3):
        ad = ctx.to_device(a)
        info, logdet = C.c_int(), C.c_double()
        ctx.call("smn_cholesky", L.dtype_code(dt), ad.ptr, n, n, n, 0, 0.0, 0.0, C.byref(info), C.byref(logdet))
        ctx.synchronize()
