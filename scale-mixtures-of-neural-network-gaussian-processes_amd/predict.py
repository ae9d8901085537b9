"""neural_tangents.predict.gradient_descent_mse_ensemble look-alike (t = infinity, NNGP posterior).

Called by spax/kernels.py:30-31 and experiments/regression/find.py:75-76 as
    predict_fn = gradient_descent_mse_ensemble(kernel_fn, x_train, y_train, diag_reg=eps)
    mean, cov = predict_fn(x_test=x_test, get="nngp", compute_cov=True)
The ridge is RELATIVE: K~ = K_dd + diag_reg * tr(K_dd)/N * I.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import as_device, default_context
from .nt_kernels import KernelFn

__all__ = ["gradient_descent_mse_ensemble", "PredictResult"]


class PredictResult(tuple):
    """(mean, cov) plus the by-products of the factorisation."""
    quad = None
    logdet = None
    info = 0


def gradient_descent_mse_ensemble(kernel_fn, x_train, y_train, diag_reg=0.0, diag_reg_absolute_scale=False):
    ctx = getattr(kernel_fn, "ctx", None) or default_context()
    x = as_device(x_train, ctx)
    y = as_device(np.asarray(y_train).reshape(x.shape[0], -1) if not isinstance(y_train, _lib.DeviceArray) else y_train,
                  ctx, dtype=x.dtype)
    n = x.shape[0]
    c = y.shape[1] if len(y.shape) > 1 else 1

    def predict_fn(t=None, x_test=None, get="nngp", compute_cov=True):
        if t is not None:
            raise NotImplementedError("only the t = infinity posterior is on the hot path")
        if get != "nngp":
            raise NotImplementedError("only get='nngp' is on the hot path (NTK predict: sample.ipynb only)")
        xt = x if x_test is None else as_device(x_test, ctx, dtype=x.dtype)
        tt = xt.shape[0]
        mean = ctx.empty((tt, c), x.dtype)
        cov = ctx.empty((tt, tt), x.dtype)
        quad = (C.c_double * c)()
        logdet = C.c_double()
        info = C.c_int()
        rel, ab = (0.0, float(diag_reg)) if diag_reg_absolute_scale else (float(diag_reg), 0.0)
        if isinstance(kernel_fn, KernelFn):
            net, act, L, w, b, lw = kernel_fn.params
            ctx.call("smn_spr_predict", x.dcode, net, act, L, w, b, lw, x.ptr, n, x.shape[1], xt.ptr, tt, xt.shape[1],
                     x.shape[1], y.ptr, c, rel, ab, mean.ptr, cov.ptr, tt, quad, C.byref(logdet), C.byref(info))
        else:  # any other kernel_fn: build the joint kernel with it, then the same factorisation
            xa = np.concatenate([np.asarray(x), np.asarray(xt)], axis=0)
            kj = as_device(kernel_fn(xa, None, "nngp"), ctx, dtype=x.dtype)
            ctx.call("smn_predict", x.dcode, kj.ptr, n, tt, n + tt, y.ptr, c, rel, ab, mean.ptr, cov.ptr, tt,
                     quad, C.byref(logdet), C.byref(info))
        if info.value != 0:       # JAX semantics: a failed Cholesky is silent NaN
            mean = ctx.to_device(np.full((tt, c), np.nan, dtype=x.dtype))
            cov = ctx.to_device(np.full((tt, tt), np.nan, dtype=x.dtype))
        res = PredictResult((mean, cov) if compute_cov else (mean,))
        res.quad = np.array(list(quad))
        res.logdet = logdet.value
        res.info = info.value
        return res if compute_cov else res[0]

    return predict_fn
