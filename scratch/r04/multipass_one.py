"""f64 N = 8192 and 16384 factorisations for a kernel trace: PASSES=4|1 python scratch/r04/multipass_one.py"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smnngp import _lib as L
ctx = L.Context(0)
ctx.call("smn_debug_panel_passes", int(os.environ.get("PASSES", "4")))
for n in (8192, 16384):
    rng = np.random.default_rng(n)
    g = rng.standard_normal((n, 32))
    a = g @ g.T / 32
    a[np.arange(n), np.arange(n)] += rng.uniform(1.0, 2.0, n)
    for rep in range(2):
        ad = ctx.to_device(a)
        info, logdet = C.c_int(), C.c_double()
        ctx.call("smn_cholesky", L.F64, ad.ptr, n, n, n, 0, 0.0, 0.0, C.byref(info), C.byref(logdet))
        ctx.synchronize()
