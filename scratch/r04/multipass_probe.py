"""Multi-pass panel workgroups (more row groups than CUs: each workgroup carries several, the later ones through the solve alone)
against one group per workgroup (smn_debug_panel_passes 1): same bits, time per factorisation.  python scratch/r04/multipass_probe.py"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smnngp import _lib as L

ctx = L.Context(0)
def factor(a, n, m, dt):
    ad = ctx.to_device(a)
    info, logdet = C.c_int(), C.c_double()
    ctx.synchronize(); t0 = time.perf_counter()
    ctx.call("smn_cholesky", L.dtype_code(dt), ad.ptr, n + m, n, n + m, 0, 0.0, 0.0, C.byref(info), C.byref(logdet))
    ctx.synchronize(); dtm = time.perf_counter() - t0
    return info.value, logdet.value, ad, dtm

ok = True
for dt, n, m in [(np.float64, 4352, 128), (np.float64, 6144, 128), (np.float64, 8192, 0), (np.float64, 16384, 0), (np.float32, 36864, 128)]:
    rng = np.random.default_rng(n)
    g = rng.standard_normal((n + m, 32)).astype(dt)
    a = g @ g.T
    a /= 32
    a[np.arange(n + m), np.arange(n + m)] += rng.uniform(1.0, 2.0, n + m).astype(dt)
    res = {}
    for mp in (4, 1, 4, 1):
        ctx.call("smn_debug_panel_passes", mp)
        i, ld, ad, tm = factor(a, n, m, dt)
        if mp not in res:
            res[mp] = (i, ld, ad.numpy(), [tm])
        else:
            res[mp][3].append(tm)
        del ad
    il = np.tril_indices(n + m, k=0) if n + m <= 9000 else None
    if il is not None:
        same = res[4][1] == res[1][1] and np.array_equal(res[4][2][il], res[1][2][il])
    else:   # the lower triangle in row blocks (an index array of the whole triangle would not fit)
        same = res[4][1] == res[1][1]
        for r0 in range(0, n + m, 2048):
            blk4, blk1 = res[4][2][r0:r0 + 2048], res[1][2][r0:r0 + 2048]
            cols = np.arange(n + m)[None, :] <= (r0 + np.arange(blk4.shape[0]))[:, None]
            same &= bool(np.array_equal(blk4[cols], blk1[cols]))
    print(np.dtype(dt).name, n, m, "info", res[4][0], "identical:", same, " ms multi-pass %s  single %s" % (["%.2f" % (1e3 * t) for t in res[4][3]], ["%.2f" % (1e3 * t) for t in res[1][3]]), flush=True)
    ok &= bool(same)
    del res, a, g
ctx.call("smn_debug_panel_passes", 4)
print("ALL IDENTICAL" if ok else "MISMATCH")
sys.exit(0 if ok else 1)
