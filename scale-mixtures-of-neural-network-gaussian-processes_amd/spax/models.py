"""spax/models.py mirror — SPR (exact GP / Student-t process regression).  SVSP is out of scope
(sparse variational classifier: SURVEY.md section 2, row 7)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from .. import _lib
from .._lib import as_device
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
                    _, scale = self.likelihood.lml_params()
                    xd, yd = self._f64_data()
                    ctx = xd.ctx
                    net, act, L, w, b, lw = kernel_fn.params
                    quad, info = C.c_double(), C.c_int()
                    ctx.call("smn_spr_loss", xd.dcode, net, act, L, w, b, lw, xd.ptr, xd.shape[0], xd.shape[1],
                             xd.shape[1], yd.ptr, 1e-6 / scale, 0.0, 1.0, None, C.byref(quad), None,
                             C.byref(info))
                    cov_data = float("nan") if info.value else quad.value / scale
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
            np.asarray(cov, dtype=np.float64) * self.y_std ** 2,
            aux,
        )
        ll = np.mean(log_prob)
        return -ll
