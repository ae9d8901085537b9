import ctypes as C, numpy as np, sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smnngp import _lib as L
ctx = L.Context(0)
sizes = [(256, 128), (512, 0), (1024, 128), (2048, 128), (4096, 128)] if len(sys.argv) < 2 else [(int(sys.argv[1]), 128)]
for n, m in sizes:
    rng = np.random.default_rng(n)
    g = rng.standard_normal((n + m, 64)); a = (g @ g.T / 64 + np.diag(rng.uniform(1.0, 2.0, n + m))).astype(np.float32)
    ad = ctx.to_device(a)
    info, logdet = C.c_int(), C.c_double()
    print("calling n=%d" % n, flush=True)
    t0 = time.time()
    ctx.call("smn_cholesky", L.F32, ad.ptr, n + m, n, n + m, 0, 0.0, 0.0, C.byref(info), C.byref(logdet))
    dt = time.time() - t0
    l = np.linalg.cholesky(a[:n, :n].astype(np.float64))
    got = ad.numpy().astype(np.float64)
    e1 = np.abs(np.tril(got[:n, :n]) - l).max() / np.abs(l).max()
    w = np.linalg.solve(l, a[:n, n:].astype(np.float64)).T if m else np.zeros((0, n))
    e2 = np.abs(got[n:, :n] - w).max() / max(1e-30, np.abs(w).max()) if m else 0.0
    print("n=%d m=%d info=%d logdet err %.2e  L err %.2e  rows err %.2e (%.1f ms)" % (n, m, info.value, abs(logdet.value - 2 * np.log(np.diag(l)).sum()) / abs(logdet.value), e1, e2, dt * 1e3), flush=True)
