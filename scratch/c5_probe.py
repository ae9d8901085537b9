"""C5 shape (N=32768 d=1024, 6-layer erf, NNGP+NTK, fp32): bare Gram vs fused builds."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smnngp import _lib as L
n, d, nl = 32768, 1024, 6
ctx = L.Context(0)
rng = np.random.default_rng(0)
x = ctx.to_device(rng.standard_normal((n, d)).astype(np.float32))
k = ctx.empty((n, n), np.float32); t = ctx.empty((n, n), np.float32)
def timed(fn, reps=3):
    fn(); ctx.synchronize()
    ctx.call("smn_timer_start")
    for _ in range(reps): fn()
    ms = C.c_double(); ctx.call("smn_timer_stop_ms", C.byref(ms))
    return ms.value / reps
g = timed(lambda: ctx.call("smn_gram", L.F32, x.ptr, n, d, None, 0, 0, d, k.ptr, n, None, None))
print("bare Gram (full mirror)      %.2f ms" % g)
for act in ("erf", "relu"):
    for mask, name in ((L.GET_NNGP, "nngp"), (L.GET_NNGP | L.GET_NTK, "nngp+ntk")):
        for fill, fname in ((L.FILL_LOWER, "lower"), (L.FILL_FULL, "full")):
            ms = timed(lambda: ctx.call("smn_kernel_mlp", L.F32, L.NET_MLP, L.ACT[act], nl, 1.5, 0.3, 1.0, x.ptr, n, d, None, 0, 0, d,
                                        mask, fill, k.ptr, t.ptr if mask & L.GET_NTK else None, n))
            print("fused %-4s L=%d %-9s %-5s  %.2f ms" % (act, nl, name, fname, ms), flush=True)
