"""spax/utils.py mirror — the GP part only (jitter, multivariate_t_logpdf) plus the factorisation helper
the likelihoods share.  The SVSP helpers of the reference (utils.py:22-74,94-140) are out of scope."""
from __future__ import annotations

import ctypes as C
import math

import numpy as np

from .. import _lib
from .._lib import DeviceArray, ScaledIdentity, as_device

__all__ = ["jitter", "multivariate_t_logpdf", "multivariate_normal_logpdf", "factor_stats"]


def jitter(num, eps=1e-6):
    """spax/utils.py:26-27 — eps * I, kept symbolic (adding it to a DeviceArray is free)."""
    return ScaledIdentity(num, eps)


def factor_stats(x, cov):
    """(quad, logdet, info) of  x^T cov^-1 x  and  log det cov  for cov = scale*A + shift*I on the device.

    One smn_lml call (jittered Cholesky + carried solve); replaces lax.linalg.cholesky +
    triangular_solve of spax/utils.py:179-180 and the Cholesky inside jax's MVN logpdf."""
    if not isinstance(cov, DeviceArray):
        cov = as_device(np.asarray(cov))
    if not cov.scale > 0.0:      # the lazy form factors A + (shift / scale) I: only for a positive scale; anything else is
        cov = as_device(cov)     # materialised as it stands (as_device resolves the view through the host)
    ctx = cov.ctx
    n = cov.shape[0]
    xv = as_device(np.asarray(x).reshape(-1), ctx, dtype=cov.dtype)
    if xv.shape[0] != n or cov.shape[1] != n:
        raise ValueError("multivariate logpdf got incompatible shapes")
    quad, logdet, info = C.c_double(), C.c_double(), C.c_int()
    ctx.call("smn_lml", cov.dcode, cov.ptr, n, cov.ld, xv.ptr, cov.shift / cov.scale, 0.0, 1.0,
             None, C.byref(quad), C.byref(logdet), C.byref(info))
    if info.value != 0:
        return float("nan"), float("nan"), info.value
    return quad.value / cov.scale, logdet.value + n * math.log(cov.scale), 0


def multivariate_normal_logpdf(x, mean, cov):
    """jax.scipy.stats.multivariate_normal.logpdf as used at spax/likelihoods.py:27."""
    x = np.asarray(x, dtype=np.float64) - np.asarray(mean, dtype=np.float64)
    n = x.shape[-1]
    quad, logdet, _ = factor_stats(x, cov)
    return -0.5 * quad - n / 2 * math.log(2 * math.pi) - 0.5 * logdet


def multivariate_t_logpdf(x, loc, shape, df, allow_singular=None):
    """spax/utils.py:160-183 (same error behaviour for the unsupported argument forms)."""
    if allow_singular is not None:
        raise NotImplementedError("allow_singular argument of multivariate_t.logpdf")
    loc = np.asarray(loc, dtype=np.float64)
    if not loc.shape:
        raise NotImplementedError("scalar loc: use a Student-t logpdf on the host")
    n = loc.shape[-1]
    if not np.shape(shape) and not isinstance(shape, DeviceArray):
        raise NotImplementedError("multivariate_t.logpdf doesn't support scalar shape")
    if len(shape.shape) < 2 or tuple(shape.shape[-2:]) != (n, n):
        raise ValueError("multivariate_t.logpdf got incompatible shapes")
    df = float(df)
    t = 0.5 * (df + n)
    quad, logdet, _ = factor_stats(np.asarray(x, dtype=np.float64) - loc, shape)
    return (-t * math.log1p(quad / df) - n / 2 * math.log(df * math.pi) + math.lgamma(t) - math.lgamma(0.5 * df)
            - 0.5 * logdet)


def digamma(x):
    """psi(x) for x > 0: upward recurrence to x >= 10, then the asymptotic series (|error| < 1e-13)."""
    x = float(x)
    if not x > 0.0:
        raise ValueError("digamma: x must be positive")
    r = 0.0
    while x < 10.0:
        r -= 1.0 / x
        x += 1.0
    f = 1.0 / (x * x)
    return r + math.log(x) - 0.5 / x - f * (1.0 / 12 - f * (1.0 / 120 - f * (1.0 / 252 - f * (1.0 / 240 - f / 132))))

