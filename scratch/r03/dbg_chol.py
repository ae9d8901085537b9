"""r03 debug: factor of a small SPD matrix through smn_cholesky against numpy, entry by entry."""
import ctypes as C
import sys
import numpy as np
sys.path.insert(0, ".")
from smnngp import _lib as L

ctx = L.default_context()
for dtype in (np.float32, np.float64):
    for n in (128, 256):
        rng = np.random.default_rng(n)
        g = rng.standard_normal((n, n + 32))
        a = g @ g.T / n + np.eye(n)
        ad = ctx.to_device(a.astype(dtype))
        info, logdet = C.c_int(), C.c_double()
        ctx.call("smn_cholesky", L.dtype_code(dtype), ad.ptr, n, n, n, 0, 0.0, 0.0, C.byref(info), C.byref(logdet))
        got = np.tril(ad.numpy().astype(np.float64))
        l = np.linalg.cholesky(a)
        err = np.abs(got - l)
        bad = np.argwhere(~(err < 1e-3))
        print(dtype.__name__, n, "info", info.value, "logdet", logdet.value, 2 * np.log(np.diag(l)).sum(), "max err", np.nanmax(err),
              "n bad", len(bad), "first bad", bad[:6].tolist())
        if len(bad):
            r, c = bad[0]
            print("  got", got[r, max(0, c - 2):c + 3], "want", l[r, max(0, c - 2):c + 3])
            cols = sorted(set(bad[:, 1].tolist()))
            rows = sorted(set(bad[:, 0].tolist()))
            print("  bad cols", cols[:40], " bad rows (first 40)", rows[:40])
