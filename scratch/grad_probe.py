"""Time of one analytic loss+gradient call vs the finite-difference gradient (13 loss evaluations) it replaces."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smnngp import nt_kernels, train
from smnngp.spax.kernels import NNGPKernel
from smnngp.spax.likelihoods import StudentTLikelihood
from smnngp.spax.models import SPR
for n, d, dt in ((1024, 64, np.float64), (4096, 512, np.float32), (8192, 512, np.float32), (16384, 3072, np.float32)):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((n, d)).astype(dt); y = rng.standard_normal(n).astype(dt)
    k = NNGPKernel(lambda w, b, l: nt_kernels.get_mlp_kernel(4, 1, act="relu", w_std=w, b_std=b, last_w_std=l), 1.0, 0.3, 1.0)
    m = SPR(k, StudentTLikelihood(2.0, 2.0), x, y, 0.0, 1.0, eps=1e-2)
    m.loss_and_grad(); m.loss()
    t0 = time.perf_counter(); l, g = m.loss_and_grad(); ta = time.perf_counter() - t0
    t0 = time.perf_counter(); m.loss(); tl = time.perf_counter() - t0
    print("N=%d d=%d %s: loss %.2f ms, loss+analytic grad %.2f ms (= %.1f loss evals; FD needs 13)" % (n, d, np.dtype(dt).name, tl * 1e3, ta * 1e3, ta / tl), flush=True)
    del m
