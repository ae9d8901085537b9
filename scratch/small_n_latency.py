"""Latency of the reference-sized workloads (C1: yacht-shaped N=245 d=6 L=2 Student-t; a few UCI-like sizes)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smnngp import nt_kernels, train
from smnngp.spax.kernels import NNGPKernel
from smnngp.spax.likelihoods import StudentTLikelihood
from smnngp.spax.models import SPR
for n, d, dt in ((245, 6, np.float64), (245, 6, np.float32), (1000, 8, np.float64), (4096, 16, np.float32)):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((n, d)).astype(dt); y = rng.standard_normal(n).astype(dt)
    xt = rng.standard_normal((32, d)).astype(dt); yt = rng.standard_normal(32).astype(dt)
    k = NNGPKernel(lambda w, b, l: nt_kernels.get_mlp_kernel(2, 1, act="relu", w_std=w, b_std=b, last_w_std=l), 1.0, 0.3, 1.0)
    m = SPR(k, StudentTLikelihood(2.0, 2.0), x, y, 0.0, 1.0, eps=1e-2)
    step = train.build_train_step(m, method="analytic")
    for f, name, reps in ((m.loss, "loss", 200), (m.loss_and_grad, "loss_and_grad", 100), (lambda: m.test_nll(xt, yt), "test_nll", 50),
                          (lambda: step(1e-3), "train_step", 100)):
        f(); f()
        t0 = time.perf_counter()
        for _ in range(reps): f()
        print("N=%d d=%d %s %-14s %8.1f us" % (n, d, np.dtype(dt).name, name, (time.perf_counter() - t0) / reps * 1e6), flush=True)
