"""neural_tangents.predict.gradient_descent_mse_ensemble look-alike (t = infinity; NNGP posterior fused on the
device, NTK posterior composed from the public entry points).

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
    ctx = getattr(kernel_fn, "ctx", None) or (x_train.ctx if isinstance(x_train, _lib.DeviceArray) else default_context())
    x = as_device(x_train, ctx)
    y = as_device(np.asarray(y_train).reshape(x.shape[0], -1) if not isinstance(y_train, _lib.DeviceArray) else y_train,
                  ctx, dtype=x.dtype)
    n = x.shape[0]
    c = y.shape[1] if len(y.shape) > 1 else 1

    def predict_fn(t=None, x_test=None, get="nngp", compute_cov=True):
        if t is not None:
            raise NotImplementedError("only the t = infinity posterior is on the hot path")
        if get == "ntk":
            return _predict_ntk(x_test, compute_cov)
        if get != "nngp":
            raise NotImplementedError("get must be 'nngp' or 'ntk'")
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

    def _predict_ntk(x_test, compute_cov):
        """get='ntk' (the reference uses it in sample.ipynb only):
            mean = Theta_td Theta~^-1 y = a^T y,   a = Theta~^-1 Theta_dt,
            cov  = K_tt + a^T K_dd a - (a^T K_dt + K_td a),      Theta~ = Theta_dd + diag_reg tr(Theta_dd)/N I.
        Composed from the public pieces, on the device: one joint (NNGP, NTK) kernel build, smn_cholesky of Theta_dd in place,
        two smn_trsm (= cho_solve) on the Theta_dt block where it lies, one transpose, four products through smn_gram
        (a general A B^T / d); only the T x T sum is host arithmetic."""
        xt = x if x_test is None else as_device(x_test, ctx, dtype=x.dtype)
        tt = xt.shape[0]
        dt = x.dtype
        xa = ctx.to_device(np.concatenate([x.numpy().reshape(n, -1), xt.numpy().reshape(tt, -1)], axis=0))
        both = kernel_fn(xa, None, ("nngp", "ntk"))
        kj, tj = as_device(both[0], ctx, dtype=dt), as_device(both[1], ctx, dtype=dt)
        m = n + tt
        es = dt.itemsize
        code = x.dcode

        def at(arr, row, col):                                                       # address of arr[row, col], ld = m
            return C.c_void_p(arr.ptr.value + (row * m + col) * es)

        rel, ab = (0.0, float(diag_reg)) if diag_reg_absolute_scale else (float(diag_reg), 0.0)
        info, logdet = C.c_int(), C.c_double()
        ctx.call("smn_cholesky", code, tj.ptr, n, n, m, n, ab, rel, C.byref(info), C.byref(logdet))   # Theta_dd -> L
        if info.value != 0:
            nanm = np.full((tt, c), np.nan, dtype=dt)
            return (nanm, np.full((tt, tt), np.nan, dtype=dt)) if compute_cov else nanm
        for trans in (0, 1):                                                         # a, in place of Theta_dt
            ctx.call("smn_trsm", code, tj.ptr, n, m, at(tj, 0, n), tt, m, trans)
        a_t = ctx.empty((tt, n), dt)
        ctx.call("smn_transpose", code, a_t.ptr, n, at(tj, 0, n), m, n, tt)
        if c == 1:
            y_t = y                                                                  # [N, 1] is [1, N]
        else:
            y_t = ctx.empty((c, n), dt)
            ctx.call("smn_transpose", code, y_t.ptr, n, y.ptr, c, n, c)
        mean_d = ctx.empty((tt, c), dt)
        ctx.call("smn_gram", code, a_t.ptr, tt, n, y_t.ptr, c, n, n, mean_d.ptr, c, None, None)      # a^T y / N
        mean = mean_d.numpy().astype(np.float64) * n
        if not compute_cov:
            return mean.astype(dt)
        ka_t = ctx.empty((tt, n), dt)
        ctx.call("smn_gram", code, a_t.ptr, tt, n, kj.ptr, n, m, n, ka_t.ptr, n, None, None)         # a^T K_dd / N
        quad_d, cross_d = ctx.empty((tt, tt), dt), ctx.empty((tt, tt), dt)
        ctx.call("smn_gram", code, a_t.ptr, tt, n, ka_t.ptr, tt, n, n, quad_d.ptr, tt, None, None)    # a^T K_dd a / N^2
        ctx.call("smn_gram", code, at(kj, n, 0), tt, m, a_t.ptr, tt, n, n, cross_d.ptr, tt, None, None)   # K_td a / N
        k_tt = np.empty((tt, tt), dt)
        ctx.call("smn_memcpy2d_d2h", k_tt.ctypes.data_as(C.c_void_p), tt * es, at(kj, n, n), m * es, tt * es, tt)
        cross = cross_d.numpy().astype(np.float64) * n
        cov = k_tt.astype(np.float64) + quad_d.numpy().astype(np.float64) * (float(n) * n) - (cross.T + cross)
        return mean.astype(dt), cov.astype(dt)

    return predict_fn
