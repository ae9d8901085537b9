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
            mean = Theta_td Theta~^-1 y,   a = Theta~^-1 Theta_dt,
            cov  = K_tt + a^T K_dd a - (a^T K_dt + K_td a),      Theta~ = Theta_dd + diag_reg tr(Theta_dd)/N I.
        Composed from the public pieces: one joint (NNGP, NTK) kernel build, smn_cholesky of Theta_dd, two smn_trsm
        (= cho_solve) on [Theta_dt | y], one device GEMM K_dd a (smn_gram is a general A B^T / d), T x T host algebra."""
        xt = x if x_test is None else as_device(x_test, ctx, dtype=x.dtype)
        tt = xt.shape[0]
        dt = x.dtype
        xa = ctx.to_device(np.concatenate([x.numpy().reshape(n, -1), xt.numpy().reshape(tt, -1)], axis=0))
        both = kernel_fn(xa, None, ("nngp", "ntk"))
        kj, tj = as_device(both[0], ctx, dtype=dt), as_device(both[1], ctx, dtype=dt)
        m = n + tt
        es = dt.itemsize
        kj_h_td = np.empty((tt, m), dt); tj_h_td = np.empty((tt, m), dt)          # rows n.. of both joint kernels
        ctx.call("smn_memcpy_d2h", kj_h_td.ctypes.data_as(C.c_void_p), C.c_void_p(kj.ptr.value + n * m * es), tt * m * es)
        ctx.call("smn_memcpy_d2h", tj_h_td.ctypes.data_as(C.c_void_p), C.c_void_p(tj.ptr.value + n * m * es), tt * m * es)
        k_td, k_tt, t_td = kj_h_td[:, :n].astype(np.float64), kj_h_td[:, n:].astype(np.float64), tj_h_td[:, :n]
        rel, ab = (0.0, float(diag_reg)) if diag_reg_absolute_scale else (float(diag_reg), 0.0)
        info, logdet = C.c_int(), C.c_double()
        ctx.call("smn_cholesky", x.dcode, tj.ptr, n, n, m, n, ab, rel, C.byref(info), C.byref(logdet))   # Theta_dd -> L
        if info.value != 0:
            nanm = np.full((tt, c), np.nan, dtype=dt)
            return (nanm, np.full((tt, tt), np.nan, dtype=dt)) if compute_cov else nanm
        rhs = ctx.to_device(np.ascontiguousarray(np.concatenate([t_td.T, y.numpy().reshape(n, c)], axis=1), dtype=dt))
        for trans in (0, 1):
            ctx.call("smn_trsm", x.dcode, tj.ptr, n, m, rhs.ptr, tt + c, tt + c, trans)
        sol = rhs.numpy().astype(np.float64)                                         # [a | Theta~^-1 y]
        a, v = sol[:, :tt], sol[:, tt:]
        mean = t_td.astype(np.float64) @ v
        if not compute_cov:
            return mean.astype(dt)
        at = ctx.to_device(np.ascontiguousarray(a.T, dtype=dt))                      # [T, N]
        ka = ctx.empty((n, tt), dt)
        ctx.call("smn_gram", x.dcode, kj.ptr, n, m, at.ptr, tt, n, n, ka.ptr, tt, None, None)   # K_dd a / n
        kda = ka.numpy().astype(np.float64) * n
        cross = k_td @ a
        cov = k_tt + a.T @ kda - (cross.T + cross)
        return mean.astype(dt), cov.astype(dt)

    return predict_fn
