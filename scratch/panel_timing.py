"""Timeline of workgroup 0 of the first sub-panel (the -DSMN_PANEL_TIMING build prints it):
    python build.py --variant timing -DSMN_PANEL_TIMING
    SMNNGP_LIB=.../libsmnngp_timing.so SMN_PANEL_LEAF=1|2 python scratch/panel_timing.py [n ...]"""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smnngp import _lib as L
ctx = L.Context(0)
sizes = [int(v) for v in sys.argv[1:]] or [2048]
for n in sizes:
    for dt in (np.float32, np.float64):
        rng = np.random.default_rng(0)
        g = rng.standard_normal((n, 64)).astype(dt)
        a = (g @ g.T / 64 + np.eye(n)).astype(dt)
        d = ctx.to_device(a)
        info, ld = C.c_int(), C.c_double()
        print("==", np.dtype(dt).name, "n", n, flush=True)
        ctx.call("smn_cholesky", L.dtype_code(dt), d.ptr, n, n, n, 0, 0.0, 0.0, C.byref(info), C.byref(ld))
        ctx.synchronize()
        print(np.dtype(dt).name, "info", info.value, "logdet", ld.value, flush=True)
