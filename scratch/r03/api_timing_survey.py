"""r03: wall time of the public entry points at one moderate size, beside a flop / byte floor, to catch silent slow paths
(the gradient's dead-tile launches were found this way).  N = 8192, d = 1024, T = 1024, fp32 unless stated."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smnngp import nt_kernels, _lib as L, predict, sweeps
from smnngp.spax.kernels import NNGPKernel
from smnngp.spax.likelihoods import GaussianLikelihood, StudentTLikelihood
from smnngp.spax.models import SPR
ctx = L.default_context()
n, d, t = 8192, 1024, 1024
rng = np.random.default_rng(0)
xh = rng.standard_normal((n, d)).astype(np.float32); yh = rng.standard_normal(n).astype(np.float32)
xth = rng.standard_normal((t, d)).astype(np.float32); yth = rng.standard_normal(t)
x, xt = ctx.to_device(xh), ctx.to_device(xth)
PEAK = 157.3e12

def bench(name, fn, flops=None, reps=3):
    fn(); ctx.synchronize()
    best = 1e9
    for _ in range(reps):
        ctx.synchronize(); t0 = time.perf_counter(); r = fn(); ctx.synchronize()
        best = min(best, time.perf_counter() - t0)
    msg = "%-58s %9.3f ms" % (name, best * 1e3)
    if flops: msg += "   %6.1f TF (%4.1f %% of f32 MFMA peak)" % (flops / best / 1e12, 100 * flops / best / PEAK)
    print(msg, flush=True)
    return r

gram = 2.0 * n * n * d
for act in ("relu", "erf"):
    kf = nt_kernels.get_mlp_kernel(4, 1, act=act, w_std=1.2, b_std=0.3, last_w_std=1.0)
    bench("mlp %s kernel_fn(x, None, 'nngp') [full matrix]" % act, lambda: kf(x, None, "nngp"), gram / 2)
    bench("mlp %s kernel_fn(x, None, ('nngp','ntk'))" % act, lambda: kf(x, None, ("nngp", "ntk")), gram / 2)
    bench("mlp %s kernel_fn(x, xt, 'nngp') [N x T]" % act, lambda: kf(x, xt, "nngp"), 2.0 * n * t * d)
kr = nt_kernels.get_dense_resnet_kernel(4, 1, act="relu", w_std=1.2, b_std=0.3, last_w_std=1.0)
bench("dense_resnet relu kernel_fn(x, None, 'nngp')", lambda: kr(x, None, "nngp"), gram / 2)
kern = NNGPKernel(lambda w, b, l: nt_kernels.get_mlp_kernel(4, 1, act="relu", w_std=w, b_std=b, last_w_std=l), 1.2, 0.3, 1.0)
chol = n ** 3 / 3.0
for nm, lik in (("gaussian", GaussianLikelihood()), ("student-t", StudentTLikelihood(2.0, 2.0))):
    m = SPR(kern, lik, x, yh, 0.0, 1.0, eps=1e-2)
    bench("SPR.loss %s" % nm, m.loss, gram / 2 + chol)
    bench("SPR.loss_and_grad %s" % nm, m.loss_and_grad, gram / 2 + 3 * chol)
    bench("SPR.test_nll %s (T = %d)" % (nm, t), lambda: m.test_nll(xt, yth), gram / 2 + chol + float(n) * n * t + float(n) * t * t)
kd = kern.get_kernel_fn()(x, None, "nngp")
post = predict.gradient_descent_mse_ensemble(kern.get_kernel_fn(), x, yh.reshape(-1, 1), diag_reg=1e-3)
bench("gradient_descent_mse_ensemble predict nngp (T = %d)" % t, lambda: post(x_test=xt, get="nngp", compute_cov=True), gram / 2 + chol + float(n) * n * t)
bench("gradient_descent_mse_ensemble predict ntk  (T = %d)" % t, lambda: post(x_test=xt, get="ntk", compute_cov=True), gram / 2 + chol + float(n) * n * t, reps=1)
from smnngp.spax.utils import multivariate_normal_logpdf
bench("multivariate_normal_logpdf(y, 0, K + 1e-2 I) from a device K", lambda: multivariate_normal_logpdf(yh, np.zeros(n), kd + L.jitter(n, 1e-2) if hasattr(L, "jitter") else kd), chol)
ns = 2048
bench("sweeps.find_grid 2x2 (w,b) x 2 eps x 3x3 (alpha,beta), N = %d T = 256" % ns,
      lambda: sweeps.find_grid(xh[:ns, :64], yh[:ns], xth[:256, :64], yth[:256], w_std_list=(1.0, 1.4), b_std_list=(0.0, 0.3), eps_list=(1e-4, 1e-2),
                               num_samples=200), reps=1)
