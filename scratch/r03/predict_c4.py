"""r03: SPR.test_nll (Gaussian head) at N = 16384, T = 2048: wall per call, for a kernel trace."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smnngp import nt_kernels, _lib as L
from smnngp.spax.kernels import NNGPKernel
from smnngp.spax.likelihoods import GaussianLikelihood
from smnngp.spax.models import SPR
ctx = L.default_context()
n, d, t = 16384, 3072, 2048
rng = np.random.default_rng(0)
xd = ctx.to_device(rng.standard_normal((n, d)).astype(np.float32)); yh = rng.standard_normal(n).astype(np.float32)
xt = ctx.to_device(rng.standard_normal((t, d)).astype(np.float32)); yt = rng.standard_normal(t)
kernel = NNGPKernel(lambda w, b, l: nt_kernels.get_mlp_kernel(4, act="relu", w_std=w, b_std=b, last_w_std=l), 1.0, 1e-8, 1.0)
m = SPR(kernel, GaussianLikelihood(), xd, yh, 0.0, 1.0, eps=1e-3)
m.test_nll(xt, yt)
for _ in range(3):
    ctx.synchronize(); t0 = time.perf_counter(); v = m.test_nll(xt, yt); ctx.synchronize()
    print("test_nll %.6f  %.2f ms" % (v, (time.perf_counter() - t0) * 1e3), flush=True)
