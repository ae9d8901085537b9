"""r03: SPR.loss_and_grad at C4 (N=16384, d=3072, 4-layer relu, fp32) against SPR.loss: wall time of each, best of 3."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smnngp import nt_kernels
from smnngp.spax.kernels import NNGPKernel
from smnngp.spax.likelihoods import GaussianLikelihood, StudentTLikelihood
from smnngp.spax.models import SPR
n, d, dt = 16384, 3072, np.float32
rng = np.random.default_rng(0)
x = rng.standard_normal((n, d)).astype(dt); y = rng.standard_normal(n).astype(dt)
k = NNGPKernel(lambda w, b, l: nt_kernels.get_mlp_kernel(4, 1, act="relu", w_std=w, b_std=b, last_w_std=l), 1.0, 0.3, 1.0)
m = SPR(k, StudentTLikelihood(2.0, 2.0), x, y, 0.0, 1.0, eps=1e-2)
m.loss_and_grad(); m.loss()
ta = tl = 1e9
for _ in range(3):
    t0 = time.perf_counter(); l, g = m.loss_and_grad(); ta = min(ta, time.perf_counter() - t0)
    t0 = time.perf_counter(); l2 = m.loss(); tl = min(tl, time.perf_counter() - t0)
print("N=%d d=%d %s: loss %.2f ms, loss+analytic grad %.2f ms (= %.2f loss evals)  loss %.6f / %.6f  grad %s" % (
    n, d, np.dtype(dt).name, tl * 1e3, ta * 1e3, ta / tl, l, l2, {k_: float(v) for k_, v in g.items()} if isinstance(g, dict) else g), flush=True)
