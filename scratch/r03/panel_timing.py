"""r03: in-kernel timeline of the panel kernel (needs the -DSMN_PANEL_TIMING variant: SMNNGP_LIB=.../libsmnngp_timing.so)."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smnngp import _lib as L
ctx = L.Context(0)
cases = ((np.float32, 2048), (np.float32, 8192), (np.float64, 2048))
if len(sys.argv) > 1:
    cases = [c for c in cases if str(c[1]) == sys.argv[1] and (len(sys.argv) < 3 or np.dtype(c[0]).name == sys.argv[2])]
for dt, n in cases:
    rng = np.random.default_rng(0)
    g = rng.standard_normal((n, 256)); a = (g @ g.T / 256 + np.eye(n)).astype(dt)
    d = ctx.to_device(a)
    info, ld = C.c_int(), C.c_double()
    print(np.dtype(dt).name, "n", n, flush=True)
    ctx.call("smn_cholesky", L.dtype_code(dt), d.ptr, n, n, n, 0, 0.0, 0.0, C.byref(info), C.byref(ld))
    ctx.synchronize()
    print("   info", info.value, "logdet", ld.value, flush=True)
