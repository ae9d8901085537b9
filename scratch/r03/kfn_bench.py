import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from smnngp import nt_kernels, _lib as L
ctx = L.default_context()
rng = np.random.default_rng(0)
for n, d in ((4096, 512), (6144, 1024), (8192, 1024), (8192, 3072), (12288, 1024)):
    x = ctx.to_device(rng.standard_normal((n, d)).astype(np.float32))
    kf = nt_kernels.get_mlp_kernel(4, 1, act="relu", w_std=1.2, b_std=0.3, last_w_std=1.0)
    kf(x, None, "nngp"); ctx.synchronize(); best = 1e9
    for _ in range(5):
        ctx.synchronize(); t0 = time.perf_counter(); kf(x, None, "nngp"); ctx.synchronize(); best = min(best, time.perf_counter() - t0)
    t = (n // 128) * (n // 128 + 1) // 2
    print("N=%d d=%d (%d lower tiles): %.3f ms  %.1f TF" % (n, d, t, best * 1e3, n * n * d / best / 1e12), flush=True)
