"""panelf_kernel (SMN_PANEL_LEAF=2: the leaf split between a factor wave and the row waves) against panelr_kernel (=1):
bit-identity of the factor, then time per factorisation.  python scratch/r04/panelf_check.py [quick]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smnngp import _lib as L


def ctx_with(leaf):
    old = os.environ.get("SMN_PANEL_LEAF")
    os.environ["SMN_PANEL_LEAF"] = leaf
    try:
        return L.Context(0)
    finally:
        if old is None:
            del os.environ["SMN_PANEL_LEAF"]
        else:
            os.environ["SMN_PANEL_LEAF"] = old


def factor(c, a, n, m, dt):
    ad = c.to_device(a)
    info, logdet = C.c_int(), C.c_double()
    c.call("smn_cholesky", L.dtype_code(dt), ad.ptr, n + m, n, n + m, 0, 0.0, 0.0, C.byref(info), C.byref(logdet))
    return info.value, logdet.value, ad


cs = {k: ctx_with(k) for k in ("1", "2")}
ok = True
for dt, n, m in [(np.float32, 256, 0), (np.float32, 2048, 128), (np.float32, 4352, 0), (np.float64, 1152, 128), (np.float32, 8192, 128)]:
    rng = np.random.default_rng(21)
    g = rng.standard_normal((n + m, 64)).astype(dt)
    a = (g @ g.T / 64 + np.diag(rng.uniform(1.0, 2.0, n + m))).astype(dt)
    out = {}
    for k in ("1", "2", "2"):
        i, ld, ad = factor(cs[k], a, n, m, dt)
        out.setdefault(k, []).append((i, ld, ad.numpy()))
    il = np.tril_indices(n + m)
    same = out["1"][0][1] == out["2"][0][1] and np.array_equal(out["1"][0][2][il], out["2"][0][2][il])
    rep = out["2"][0][1] == out["2"][1][1] and np.array_equal(out["2"][0][2][il], out["2"][1][2][il])
    ref = np.linalg.cholesky(a[:n, :n].astype(np.float64))
    err = np.abs(np.tril(out["2"][0][2][:n, :n]) - ref).max()
    print(np.dtype(dt).name, n, m, "info", out["2"][0][0], "bit-identical to panelr:", same, " repeatable:", rep, " max|L - L_ref|", err, flush=True)
    ok &= same and rep
print("ALL IDENTICAL" if ok else "MISMATCH", flush=True)
if len(sys.argv) > 1 and sys.argv[1] == "quick":
    sys.exit(0 if ok else 1)
for dt, n in [(np.float32, 2048), (np.float32, 4096), (np.float32, 8192), (np.float32, 16384), (np.float64, 4096), (np.float64, 8192)]:
    rng = np.random.default_rng(3)
    g = rng.standard_normal((n, 64)).astype(dt)
    a = (g @ g.T / 64 + np.diag(rng.uniform(1.0, 2.0, n))).astype(dt)
    for k in ("1", "2"):
        c = cs[k]
        ad = c.to_device(a)
        src = c.to_device(a)
        info, logdet = C.c_int(), C.c_double()
        ts = []
        for rep in range(8):
            ad.copy_from(src) if hasattr(ad, "copy_from") else None
            c.synchronize()
            t0 = time.perf_counter()
            c.call("smn_cholesky", L.dtype_code(dt), ad.ptr, n, n, n, 0, 0.0, 0.0, C.byref(info), C.byref(logdet))
            c.synchronize()
            ts.append(time.perf_counter() - t0)
            if not hasattr(ad, "copy_from"):
                ad = c.to_device(a)
        print(np.dtype(dt).name, n, "leaf", k, "min %.3f ms  median %.3f ms" % (1e3 * min(ts[1:]), 1e3 * float(np.median(ts[1:]))), flush=True)
sys.exit(0 if ok else 1)
