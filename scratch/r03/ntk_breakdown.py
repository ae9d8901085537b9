import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, "/root/repo")
from smnngp import nt_kernels, _lib as L
ctx = L.default_context()
n, d, tt = 8192, 1024, 1024
rng = np.random.default_rng(0)
m = n + tt
xa = ctx.to_device(rng.standard_normal((m, d)).astype(np.float32))
kf = nt_kernels.get_mlp_kernel(4, 1, act="relu", w_std=1.2, b_std=0.3, last_w_std=1.0)
def T(name, fn, reps=3):
    fn(); ctx.synchronize(); best = 1e9
    for _ in range(reps):
        ctx.synchronize(); t0 = time.perf_counter(); r = fn(); ctx.synchronize(); best = min(best, time.perf_counter() - t0)
    print("%-40s %8.3f ms" % (name, best * 1e3), flush=True); return r
xh = T("download x (m x d)", lambda: xa.numpy())
T("upload x", lambda: ctx.to_device(xh))
both = T("joint build nngp+ntk", lambda: kf(xa, None, ("nngp", "ntk")))
kj, tj = both
code = xa.dcode; es = 4
info, logdet = C.c_int(), C.c_double()
tj0 = tj.numpy()
def chol():
    t = ctx.to_device(tj0); ctx.synchronize(); t0 = time.perf_counter()
    ctx.call("smn_cholesky", code, t.ptr, n, n, m, n, 0.0, 1e-3, C.byref(info), C.byref(logdet)); ctx.synchronize()
    return t, time.perf_counter() - t0
t, dtc = chol(); t, dtc = chol(); print("cholesky Theta_dd %.3f ms info %d" % (dtc * 1e3, info.value))
bptr = C.c_void_p(t.ptr.value + n * es)
for trans in (0, 1):
    T("smn_trsm trans=%d (N x T rhs, ld m)" % trans, lambda: ctx.call("smn_trsm", code, t.ptr, n, m, bptr, tt, m, trans))
a_t = ctx.empty((tt, n), np.float32)
T("smn_transpose", lambda: ctx.call("smn_transpose", code, a_t.ptr, n, bptr, m, n, tt))
ka_t = ctx.empty((tt, n), np.float32)
T("smn_gram a^T K_dd (T x N x N)", lambda: ctx.call("smn_gram", code, a_t.ptr, tt, n, kj.ptr, n, m, n, ka_t.ptr, n, None, None))
q = ctx.empty((tt, tt), np.float32)
T("smn_gram a^T (K a) (T x T x N)", lambda: ctx.call("smn_gram", code, a_t.ptr, tt, n, ka_t.ptr, tt, n, n, q.ptr, tt, None, None))
T("download T x T", lambda: q.numpy())
