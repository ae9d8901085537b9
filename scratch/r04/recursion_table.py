"""Stand-alone layer recursion (a3) over a stored K0, N = 16384 fp32: symmetric (lower tiles + mirror) and cross form,
L = 1, 2, 4, 6, ReLU and erf.  Algorithmic bytes = read N^2 + write N^2."""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smnngp import _lib as L
n, d = 16384, 256
ctx = L.Context(0)
rng = np.random.default_rng(0)
x = ctx.to_device(rng.standard_normal((n, d)).astype(np.float32))
k0 = ctx.empty((n, n), np.float32); kk = ctx.empty((n, n), np.float32); q1 = ctx.empty((n,), np.float32)
ctx.call("smn_gram", L.F32, x.ptr, n, d, None, 0, 0, d, k0.ptr, n, q1.ptr, None)
for sym in (1, 0):
    for act in ("relu", "erf"):
        for nl in (1, 2, 4, 6):
            def rec():
                ctx.call("smn_recursion", L.F32, L.NET_MLP, L.ACT[act], nl, 1.0, 1e-8, 1.0, k0.ptr, n, n, n, q1.ptr, q1.ptr, sym,
                         L.GET_NNGP, kk.ptr, None, n)
            for _ in range(2): rec()
            ctx.call("smn_profile_enable", 1)
            for _ in range(5): rec()
            ms, cnt = C.c_double(), C.c_int()
            ctx.call("smn_profile_read", 2, C.byref(ms), C.byref(cnt))
            ctx.call("smn_profile_enable", 0)
            per = ms.value / cnt.value
            print("sym=%d recursion N=%d L=%d %s: %.3f ms  %.0f GB/s (%.1f%% of 8 TB/s)" % (sym, n, nl, act, per, 2.0*n*n*4/per/1e6, 2.0*n*n*4/per/1e6/80), flush=True)
