"""Split build (the kernel matrix's corner built on the bulk stream beside the first panel chain) against the single launch:
same bits, time per smn_spr_loss.  python scratch/r04/split_build_probe.py"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smnngp import _lib as L

ctx = L.Context(0)
def loss(x, y, L_, act, eps):
    lp, q, ld, info = C.c_double(), C.c_double(), C.c_double(), C.c_int()
    ctx.call("smn_spr_loss", x.dcode, L.NET_MLP, L.ACT[act], L_, 1.0, 0.3, 1.0, x.ptr, x.shape[0], x.ld, x.shape[1], y.ptr, eps, 0.0, 1.0,
             C.byref(lp), C.byref(q), C.byref(ld), C.byref(info))
    return lp.value, q.value, ld.value, info.value

for dt, n, d, nl, act in [(np.float32, 8000, 256, 2, "relu"), (np.float32, 12345, 128, 3, "erf"), (np.float64, 8192, 64, 2, "relu"),
                          (np.float32, 16384, 3072, 4, "relu"), (np.float32, 32768, 1024, 6, "erf")]:
    rng = np.random.default_rng(n)
    x = ctx.to_device((rng.standard_normal((n, d)) ).astype(dt))
    y = ctx.to_device(rng.standard_normal((n, 1)).astype(dt))
    res, tms = {}, {}
    for on in (1, 0, 1, 0):
        ctx.call("smn_debug_split_build", on)
        r = loss(x, y, nl, act, 1e-2)
        res.setdefault(on, []).append(r)
        ts = []
        for rep in range(5):
            ctx.synchronize(); t0 = time.perf_counter(); loss(x, y, nl, act, 1e-2); ts.append(time.perf_counter() - t0)
        tms.setdefault(on, []).append(1e3 * float(np.median(ts)))
    same = all(r == res[1][0] for r in res[1] + res[0])
    print(np.dtype(dt).name, n, d, nl, act, "identical:", same, res[1][0], " ms split %s  single %s" % (["%.3f" % t for t in tms[1]], ["%.3f" % t for t in tms[0]]), flush=True)
ctx.call("smn_debug_split_build", 1)
