"""Largest-size sanity run: N = 65536 (fp32): fused loss, sampled kernel rows against the oracle, log-pdf identity."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nngp_oracle as O
from smnngp import _lib as L
n, d, nl = int(os.environ.get("PN", 65536)), 256, 2
ctx = L.Context(0)
rng = np.random.default_rng(0)
xh = rng.standard_normal((n, d)).astype(np.float32); yh = rng.standard_normal(n).astype(np.float32)
x = ctx.to_device(xh); y = ctx.to_device(yh)
lp, quad, logdet, info = C.c_double(), C.c_double(), C.c_double(), C.c_int()
t0 = time.perf_counter()
ctx.call("smn_spr_loss", L.F32, L.NET_MLP, L.ACT["relu"], nl, 1.2, 0.3, 1.0, x.ptr, n, d, d, y.ptr, 1e-2, 0.0, 1.0,
         C.byref(lp), C.byref(quad), C.byref(logdet), C.byref(info))
print("N=%d: smn_spr_loss %.1f ms, info %d, logpdf %.6g, identity residual %.2e" % (
    n, (time.perf_counter() - t0) * 1e3, info.value, lp.value,
    abs(lp.value - (-0.5 * quad.value - 0.5 * n * np.log(2 * np.pi) - 0.5 * logdet.value)) / abs(lp.value)), flush=True)
rows = np.sort(rng.choice(n, 16, replace=False))
out = ctx.empty((16, n), np.float32)
for i, r in enumerate(rows):
    ctx.call("smn_kernel_mlp_rows", L.F32, L.NET_MLP, L.ACT["relu"], nl, 1.2, 0.3, 1.0, x.ptr, n, d, d, int(r), int(r) + 1, L.GET_NNGP,
             C.c_void_p(out.ptr.value + i * n * 4), None, n)
got = out.numpy().astype(np.float64)
x64 = xh.astype(np.float64)
ref = O.mlp_kernel(x64[rows], x64, nl, "relu", 1.2, 0.3, 1.0)
for i, r in enumerate(rows):
    ref[i, r] = O.diag_recursion((x64[r] ** 2).sum() / d, nl, "relu", 1.2, 0.3, 1.0)
print("sampled kernel rows: max rel err %.2e" % (np.abs(got - ref).max() / np.abs(ref).max()), flush=True)
