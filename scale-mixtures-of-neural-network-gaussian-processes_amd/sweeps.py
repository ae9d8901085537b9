"""Batched hyper-parameter sweeps on one dataset — the grid search of experiments/regression/find.py:134-199
restructured around what the device can do at once:

  * every (w_std, b_std, eps) cell is one small problem of identical shape; ALL of them go through the library's batched
    entry points (smn_spr_predict_batch, smn_spr_loss_batch): one sequence of launches with grid.y = the number of cells
    -- fused build under each cell's layer program, factorisation, read-out -- so 99 problems of N = 2048 fill the chip
    instead of queueing on a few workgroups each;
  * each eps costs two factorisations of the cell's kernel (the reference does the same two: find.py:141 predict with
    neural_tangents' relative ridge, find.py:151-159 inv / logdet with the absolute eps): two batches;
  * every (alpha, beta) of the Burr-XII scale mixture is host arithmetic on the scalars and T-vectors that
    came back (find.py:165-187, reproduced as written, including prob_prior == prob_q and random_state=101).
"""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from . import _lib
from ._lib import as_device, default_context

__all__ = ["find_grid", "loss_batch", "predict_batch"]

_NET = {"mlp": _lib.NET_MLP, None: _lib.NET_MLP, "resnet": _lib.NET_DENSE_RESNET}


def _norm_logpdf(x, mean, sigma):
    z = (x - mean) / sigma
    return -0.5 * z * z - np.log(sigma) - 0.5 * math.log(2 * math.pi)


def _darr(values):
    arr = np.ascontiguousarray(values, dtype=np.float64)
    return arr, arr.ctypes.data_as(C.POINTER(C.c_double))


def loss_batch(ctx, x, y, *, network="mlp", num_hiddens=4, activation="relu", w_std, b_std, last_w_std=1.0, eps, df=0.0, scale=1.0):
    """G x SPR.loss on one data set in one batched pass (smn_spr_loss_batch).  w_std, b_std, last_w_std, eps, df, scale:
    scalars or sequences of one common length G.  x, y: DeviceArrays of ctx (float32 or float64, the compute type).
    Returns (logpdf[G], quad[G], logdet[G], info[G]) as NumPy arrays; logpdf is NaN where info != 0."""
    g = max(np.size(v) for v in (w_std, b_std, last_w_std, eps, df, scale))
    cols = [_darr(np.broadcast_to(np.asarray(v, dtype=np.float64), (g,))) for v in (w_std, b_std, last_w_std, eps, df, scale)]
    n, d = x.shape
    lp, quad, logdet = np.empty(g), np.empty(g), np.empty(g)
    info = np.zeros(g, dtype=np.int32)
    pd = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))    # noqa: E731
    ctx.call("smn_spr_loss_batch", x.dcode, _NET[network], _lib.ACT[activation], num_hiddens, g, cols[0][1], cols[1][1], cols[2][1],
             x.ptr, n, x.ld, d, y.ptr, cols[3][1], cols[4][1], cols[5][1], pd(lp), pd(quad), pd(logdet),
             info.ctypes.data_as(C.POINTER(C.c_int)))
    return lp, quad, logdet, info


def predict_batch(ctx, x, y, x_test, *, network="mlp", num_hiddens=4, activation="relu", w_std, b_std, last_w_std=1.0,
                  diag_reg, full_cov=False, on_device=False):
    """G x NNGPKernel.predict (relative ridge diag_reg, spax/kernels.py:29-32) on one data set in one batched pass
    (smn_spr_predict_batch).  y: DeviceArray [n, c].  Returns (mean [G,t,c], var [G,t] or cov [G,t,t], info[G]); the first
    two as DeviceArrays when on_device."""
    g = max(np.size(v) for v in (w_std, b_std, last_w_std, diag_reg))
    cols = [_darr(np.broadcast_to(np.asarray(v, dtype=np.float64), (g,))) for v in (w_std, b_std, last_w_std, diag_reg)]
    n, d = x.shape
    t = x_test.shape[0]
    c = y.shape[1] if len(y.shape) > 1 else 1
    mean = ctx.empty((g, t, c), x.dtype)
    out = ctx.empty((g, t, t) if full_cov else (g, t), x.dtype)
    info = np.zeros(g, dtype=np.int32)
    ctx.call("smn_spr_predict_batch", x.dcode, _NET[network], _lib.ACT[activation], num_hiddens, g, cols[0][1], cols[1][1], cols[2][1],
             x.ptr, n, x.ld, x_test.ptr, t, x_test.ld, d, y.ptr, c, cols[3][1], None, mean.ptr,
             out.ptr if full_cov else None, t, None if full_cov else out.ptr, None, None, info.ctypes.data_as(C.POINTER(C.c_int)))
    return (mean, out, info) if on_device else (mean.numpy(), out.numpy(), info)


def find_grid(x_train, y_train, x_test, y_test, y_mean=0.0, y_std=1.0, *, network="mlp", num_hiddens=4,
              activation="relu", w_std_list=(1.0, 1.4, 2.0), b_std_list=(0.0, 0.3, 1.0),
              eps_list=(1e-6, 1e-4, 1e-2), alpha_list=(1.0, 2.0, 3.0), beta_list=(1.0, 2.0, 3.0),
              num_samples=1000, ctx=None):
    """Returns dict(gnll[i,j,k], tnll[i,j,k,a,b], best_gaussian, best_student) — the tables find.py logs."""
    from scipy import stats as scipy_stats       # the Burr-XII draws, as in the reference's find.py

    if network not in _NET:
        raise ValueError(f"Unsupported network '{network}'")
    if activation not in _lib.ACT:
        raise KeyError("Unsupported act '{}'".format(activation))
    ctx = ctx or default_context()
    xt = as_device(x_train, ctx)
    dt = xt.dtype
    n, d = xt.shape
    xs = as_device(np.asarray(x_test, dtype=dt), ctx)
    t = xs.shape[0]
    y = np.asarray(y_train, dtype=np.float64).reshape(-1)
    yd = ctx.to_device(y.astype(dt).reshape(n, 1))
    y_ = np.asarray(y_test, dtype=np.float64) * y_std + y_mean
    minus_log_two_pi = -(n / 2) * math.log(2 * math.pi)
    shape = (len(w_std_list), len(b_std_list), len(eps_list))
    gnll = np.full(shape, np.nan)
    tnll = np.full(shape + (len(alpha_list), len(beta_list)), np.nan)
    # every cell of the grid is one problem of the two batches
    wi, bj, ek = np.meshgrid(np.asarray(w_std_list, float), np.asarray(b_std_list, float), np.asarray(eps_list, float), indexing="ij")
    kw = dict(network=network, num_hiddens=num_hiddens, activation=activation, w_std=wi.ravel(), b_std=bj.ravel(), last_w_std=1.0)
    mean_dev, var_dev, info_p = predict_batch(ctx, xt, yd, xs, diag_reg=ek.ravel(), on_device=True, **kw)   # find.py:141 (relative ridge)
    _, quad_all, logdet_all, info_l = loss_batch(ctx, xt, yd, eps=ek.ravel(), **kw)              # find.py:151-159 (absolute eps)
    mean_all, var_all = mean_dev.numpy(), var_dev.numpy()
    for g, (i, j, k) in enumerate(np.ndindex(*shape)):
        if info_p[g]:
            continue
        mean_ = mean_all[g].astype(np.float64).ravel() * y_std + y_mean
        std_diag = np.sqrt(var_all[g].astype(np.float64))       # only the marginal variances are used (find.py:50-55)
        gnll[i, j, k] = -np.mean(_norm_logpdf(y_, mean_, std_diag * y_std))             # find.py:50-55,145
    # find.py:165-187, every (cell, alpha, beta) at once on the device (smn_mixture_nll).  find.py:165-170 redraws the
    # Burr-XII sample inside the innermost loop with a FIXED random_state=101, i.e. the same numbers every time: one draw
    # per (alpha, beta); prob_prior == prob_q there, so the importance ratio is 1.
    mixes = [(a, bb) for a in alpha_list for bb in beta_list]
    sample_q = np.ascontiguousarray([scipy_stats.burr12.rvs(c=a, d=bb, loc=0., scale=1., size=num_samples, random_state=101)
                                     for a, bb in mixes], dtype=np.float64)
    g_all = int(np.prod(shape))
    skip = np.ascontiguousarray((info_p != 0) | (info_l != 0), dtype=np.int32)
    tn = np.empty((g_all, len(mixes)))
    pd = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))    # noqa: E731
    quad_c, logdet_c = np.nan_to_num(quad_all), np.nan_to_num(logdet_all)
    y_c = np.ascontiguousarray(y_, dtype=np.float64)
    ctx.call("smn_mixture_nll", xt.dcode, g_all, t, mean_dev.ptr, var_dev.ptr, pd(quad_c), pd(logdet_c),
             skip.ctypes.data_as(C.POINTER(C.c_int)), pd(y_c), float(y_mean), float(y_std), n, len(mixes), int(num_samples),
             pd(sample_q), None, pd(tn))
    tnll[...] = tn.reshape(shape + (len(alpha_list), len(beta_list)))
    out = dict(gnll=gnll, tnll=tnll, best_gaussian=None, best_student=None)
    if np.isfinite(gnll).any():
        i, j, k = np.unravel_index(np.nanargmin(gnll), gnll.shape)
        out["best_gaussian"] = ((w_std_list[i], b_std_list[j], eps_list[k]), float(gnll[i, j, k]))
    if np.isfinite(tnll).any():
        i, j, k, ia, ib = np.unravel_index(np.nanargmin(tnll), tnll.shape)
        out["best_student"] = ((w_std_list[i], b_std_list[j], alpha_list[ia], beta_list[ib], eps_list[k]),
                               float(tnll[i, j, k, ia, ib]))
    return out
