import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from smnngp import nt_kernels
from smnngp.spax.kernels import NNGPKernel
from smnngp.spax.likelihoods import StudentTLikelihood
from smnngp.spax.models import SPR
for n, d in ((8192, 256), (16384, 3072)):
    dt = np.float32
    rng = np.random.default_rng(0)
    x = rng.standard_normal((n, d)).astype(dt); y = rng.standard_normal(n).astype(dt)
    k = NNGPKernel(lambda w, b, l: nt_kernels.get_mlp_kernel(4, 1, act="relu", w_std=w, b_std=b, last_w_std=l), 1.0, 0.3, 1.0)
    m = SPR(k, StudentTLikelihood(2.0, 2.0), x, y, 0.0, 1.0, eps=1e-2)
    for env in (None,):
        if env: os.environ["SMN_NO_ID"] = env
        else: os.environ.pop("SMN_NO_ID", None)
        m.loss_and_grad()
        t0 = time.perf_counter(); l, g = m.loss_and_grad(); ta = time.perf_counter() - t0
        print(n, "SMN_NO_ID", env, "%.2f ms" % (ta * 1e3), l, flush=True)
    del m
