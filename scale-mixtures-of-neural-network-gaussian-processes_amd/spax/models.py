"""spax/models.py mirror — SPR (exact GP / Student-t process regression).  SVSP is out of scope
(sparse variational classifier: SURVEY.md section 2, row 7)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _lib
from .._lib import DeviceArray, as_device
from ..nt_kernels import KernelFn
from .base import ConstraintTrainVar, Module
from .bijectors import positive
from .utils import jitter

__all__ = ["SPR"]


class SPR(Module):
    def __init__(self, kernel, likelihood, x_data, y_data, y_mean, y_std, *, eps: float = 1e-6):
        super().__init__()
        self.kernel = kernel
        self.likelihood = likelihood
        self.x_data = as_device(x_data)              # resident in HBM for the life of the model
        self.y_host = np.asarray(y_data, dtype=np.float64).reshape(-1)
        self.y_data = as_device(self.y_host, self.x_data.ctx, dtype=self.x_data.dtype)
        self.y_mean = float(np.asarray(y_mean))
        self.y_std = float(np.asarray(y_std))
        self.num_data = self.x_data.shape[0]
        self.eps = ConstraintTrainVar(eps, constraint=positive())

    def _f64_data(self):
        if self.x_data.dtype == np.float64:
            return self.x_data, self.y_data
        if getattr(self, "_x64", None) is None:
            ctx = self.x_data.ctx
            self._x64 = ctx.to_device(self.x_data.numpy().astype(np.float64))
            self._y64 = ctx.to_device(self.y_host)
        return self._x64, self._y64

    def _student_quad_f64(self, kernel_fn, scale):
        """y^T (K + (1e-6 / scale) I)^-1 y / scale  =  y^T (scale K + 1e-6 I)^-1 y in fp64 (likelihoods.py:60-61).
        It depends on the training data and the hyper-parameters only -- not on the test points -- and the reference's
        evaluation loop calls test_nll twice per check point, on the validation and on the test split
        (experiments/regression/train.py:203-212, test.py:89-99): the value of the last parameter setting is kept, so the
        second call costs the fp32 posterior alone.  (Measured at N = 16384: this fp64 build + factorisation is 44 ms
        against 29 ms for the posterior, and running the two concurrently on two contexts buys 3 % -- both are bound by the
        matrix pipes, not by latency; profiles/r04_two_context_probe.txt.)"""
        key = (tuple(kernel_fn.params), float(scale))
        hit = getattr(self, "_quad64_cache", None)
        if hit is not None and hit[0] == key:
            return hit[1]
        xd, yd = self._f64_data()
        net, act, L, w, b, lw = kernel_fn.params
        quad, info = C.c_double(), C.c_int()
        xd.ctx.call("smn_spr_loss", xd.dcode, net, act, L, w, b, lw, xd.ptr, xd.shape[0], xd.shape[1],
                    xd.shape[1], yd.ptr, 1e-6 / scale, 0.0, 1.0, None, C.byref(quad), None, C.byref(info))
        val = float("nan") if info.value else quad.value / scale
        self._quad64_cache = (key, val)
        return val

    # ---- spax/models.py:93-98
    def loss(self):
        eps = self.eps.safe_value
        kernel_fn = self.kernel.get_kernel_fn()
        if isinstance(kernel_fn, KernelFn) and hasattr(self.likelihood, "lml_params"):
            # fused: build K(X,X)+eps I in the factorisation workspace, factor, carry y through
            df, scale = self.likelihood.lml_params()
            x, ctx = self.x_data, self.x_data.ctx
            net, act, L, w, b, lw = kernel_fn.params
            lp, info = C.c_double(), C.c_int()
            ctx.call("smn_spr_loss", x.dcode, net, act, L, w, b, lw, x.ptr, x.shape[0], x.shape[1], x.shape[1],
                     self.y_data.ptr, eps, df, scale, C.byref(lp), None, None, C.byref(info))
            log_prob = lp.value
        else:
            cov = self.kernel.K(kernel_fn, self.x_data) + jitter(self.num_data, eps=eps)
            log_prob = self.likelihood.prior_logpdf(self.y_host, cov)
        return -log_prob / self.num_data

    def loss_and_grad(self):
        """(loss, {variable name: d loss / d RAW value}) -- the analytic counterpart of
        objax.GradValues(model.loss, model.vars()) in experiments/regression/train.py:61-67.
        One augmented factorisation gives alpha = K~^-1 y, K~^-1, the quadratic form and logdet; one pass over
        the lower triangle of X X^T / d contracts G = coef alpha alpha^T - K~^-1 with the forward-mode
        dK/d(w_std, b_std, last_w_std) (csrc/grad.hip).  The (a, b) derivatives of the Student-t head and the
        softplus chain rule are closed forms on the host.  MLP / dense-ResNet kernels only."""
        import math
        from .utils import digamma
        kernel_fn = self.kernel.get_kernel_fn()
        if not (isinstance(kernel_fn, KernelFn) and hasattr(self.likelihood, "lml_params")):
            raise NotImplementedError("analytic gradients need an MLP / dense-ResNet KernelFn and a Gaussian or "
                                      "Student-t likelihood; use train.value_and_grad_fd")
        eps = self.eps.safe_value
        df, scale = self.likelihood.lml_params()
        x, ctx = self.x_data, self.x_data.ctx
        net, act, L, w, b, lw = kernel_fn.params
        n = self.num_data
        quad, logdet, info = C.c_double(), C.c_double(), C.c_int()
        terms = (C.c_double * 4)()
        ctx.call("smn_spr_loss_grad", x.dcode, net, act, L, w, b, lw, x.ptr, n, x.shape[1], x.shape[1],
                 self.y_data.ptr, eps, df, scale, C.byref(quad), C.byref(logdet), C.byref(info), terms)
        names = {id(v): k for k, v in self.vars().items()}
        nan = float("nan")
        if info.value != 0:
            return nan, {k: nan for k in self.vars()}
        q, ld = quad.value, logdet.value
        dlp = {}                                              # d logpdf / d constrained value
        for var, t in ((self.kernel.w_std, terms[0]), (self.kernel.b_std, terms[1]),
                       (self.kernel.last_w_std, terms[2]), (self.eps, terms[3])):
            dlp[id(var)] = 0.5 * t
        if df <= 0.0:                                         # likelihoods.py:25-28
            lp = -0.5 * q - 0.5 * n * math.log(2.0 * math.pi) - 0.5 * ld
        else:                                                 # likelihoods.py:45-50, utils.py:178-183
            a_, b_ = self.likelihood.a.safe_value, self.likelihood.b.safe_value
            t = 0.5 * (df + n)
            qs = q / scale
            lp = (-t * math.log1p(qs / df) - 0.5 * n * math.log(df * math.pi) + math.lgamma(t) - math.lgamma(0.5 * df)
                  - 0.5 * (ld + n * math.log(scale)))
            d_scale = t * qs / ((df + qs) * scale) - 0.5 * n / scale
            d_df = (-0.5 * math.log1p(qs / df) + t * qs / (df * (df + qs)) - 0.5 * n / df
                    + 0.5 * digamma(t) - 0.5 * digamma(0.5 * df))
            dlp[id(self.likelihood.a)] = 2.0 * d_df - d_scale * b_ / (a_ * a_)     # df = 2a, scale = b/a
            dlp[id(self.likelihood.b)] = d_scale / a_
        grads = {}
        for vid, g in dlp.items():
            var = next(v for v in self.vars().values() if id(v) == vid)
            grads[names[vid]] = float(-g / n * var.constraint.grad(var.value))
        return -lp / n, grads

    # ---- spax/models.py:100-120
    def test_nll(self, x, y):
        eps = self.eps.safe_value
        kernel_fn = self.kernel.get_kernel_fn()
        mean, cov = self.kernel.predict(kernel_fn, self.x_data, self.y_data, x, eps=eps)
        require = self.likelihood.require
        if require:
            if "cov_data" in require:
                if isinstance(kernel_fn, KernelFn):
                    # likelihoods.py:60-61 needs y^T (b/a K + 1e-6 I)^-1 y with K WITHOUT the eps jitter
                    # (models.py:107 "TODO: check"); hand the quadratic form over instead of an N x N matrix.
                    # That matrix carries only a 1e-6 jitter: in fp32 it is not numerically PD (the
                    # reference's fp32 inv() returns noise there), so this one quadratic form always runs
                    # in fp64 on upcast copies of the training data.
                    cov_data = self._student_quad_f64(kernel_fn, self.likelihood.lml_params()[1])
                else:
                    cov_data = self.kernel.K(kernel_fn, self._f64_data()[0])   # fp64 for the same reason
            aux_dict = dict(cov_data=cov_data, y_data=self.y_host)
            aux = tuple(aux_dict[k] for k in require)
        else:
            aux = None

        y = np.asarray(y, dtype=np.float64)
        log_prob = self.likelihood.logpdf(
            (y * self.y_std) + self.y_mean,
            (np.asarray(mean, dtype=np.float64).flatten() * self.y_std) + self.y_mean,
            cov * self.y_std ** 2 if isinstance(cov, DeviceArray) else np.asarray(cov, dtype=np.float64) * self.y_std ** 2,
            aux,
        )
        ll = np.mean(log_prob)
        return -ll
