"""Kernel sequence of one reference-sized SPR.loss / loss_and_grad / test_nll (N=245, d=6): run under rocprofv3 --kernel-trace."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smnngp import nt_kernels
from smnngp.spax.kernels import NNGPKernel
from smnngp.spax.likelihoods import StudentTLikelihood
from smnngp.spax.models import SPR
rng = np.random.default_rng(0)
n, d, dt = 245, 6, np.float32
x = rng.standard_normal((n, d)).astype(dt); y = rng.standard_normal(n).astype(dt)
xt = rng.standard_normal((32, d)).astype(dt); yt = rng.standard_normal(32).astype(dt)
k = NNGPKernel(lambda w, b, l: nt_kernels.get_mlp_kernel(2, 1, act="relu", w_std=w, b_std=b, last_w_std=l), 1.0, 0.3, 1.0)
m = SPR(k, StudentTLikelihood(2.0, 2.0), x, y, 0.0, 1.0, eps=1e-2)
for _ in range(3): m.loss()
m.loss_and_grad(); m.loss_and_grad()
m.test_nll(xt, yt); m.test_nll(xt, yt)
