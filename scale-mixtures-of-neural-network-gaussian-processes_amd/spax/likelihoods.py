"""spax/likelihoods.py mirror — Gaussian and Student-t (inverse-gamma scale mixture) heads."""
from __future__ import annotations

import math

import numpy as np

from .._lib import DeviceArray, as_device
from .base import ConstraintTrainVar, Module
from .bijectors import positive
from .utils import factor_stats, jitter, multivariate_normal_logpdf, multivariate_t_logpdf

__all__ = ["Likelihood", "GaussianLikelihood", "StudentTLikelihood"]


def _diag64(cov):
    """diag(cov) in float64; a device matrix hands over its diagonal only (the marginal heads below read nothing else)."""
    if isinstance(cov, DeviceArray):
        return np.asarray(cov.diagonal(), dtype=np.float64)
    return np.diagonal(np.asarray(cov, dtype=np.float64))


def _norm_logpdf(x, mean, sigma):
    z = (x - mean) / sigma
    return -0.5 * z * z - np.log(sigma) - 0.5 * math.log(2 * math.pi)


def _t_logpdf(x, df, loc, scale):
    z = (x - loc) / scale
    return (math.lgamma(0.5 * (df + 1)) - math.lgamma(0.5 * df) - 0.5 * math.log(df * math.pi)
            - np.log(scale) - 0.5 * (df + 1) * np.log1p(z * z / df))


class Likelihood(Module):
    pass


class GaussianLikelihood(Likelihood):
    require = None

    def lml_params(self):
        """(df, scale) for the fused smn_spr_loss call: df <= 0 selects the Gaussian form."""
        return 0.0, 1.0

    def prior_logpdf(self, x, cov):
        """spax/likelihoods.py:25-28."""
        zero = np.zeros_like(np.asarray(x, dtype=np.float64))
        return multivariate_normal_logpdf(x, zero, cov)

    def logpdf(self, x, mean, cov, aux):
        """spax/likelihoods.py:30-33."""
        sigma = np.sqrt(_diag64(cov))
        return _norm_logpdf(np.asarray(x, dtype=np.float64), np.asarray(mean, dtype=np.float64), sigma)


class StudentTLikelihood(Likelihood):
    require = ["cov_data", "y_data"]

    def __init__(self, alpha, beta):
        super().__init__()
        self.a = ConstraintTrainVar(alpha, constraint=positive())
        self.b = ConstraintTrainVar(beta, constraint=positive())

    def lml_params(self):
        a, b = self.a.safe_value, self.b.safe_value
        return 2.0 * a, b / a

    def prior_logpdf(self, x, cov):
        """spax/likelihoods.py:45-50 — MVT(nu = 2a, shape = (b/a) cov)."""
        a = self.a.safe_value
        b = self.b.safe_value
        zero = np.zeros_like(np.asarray(x, dtype=np.float64))
        return multivariate_t_logpdf(x, zero, (b / a) * cov, 2 * a)

    def logpdf(self, x, mean, cov, aux):
        """spax/likelihoods.py:52-65.  The reference forms inv(b/a K + 1e-6 I) explicitly (:60); here the
        quadratic form comes from one jittered Cholesky.  `cov_data` may be a device matrix or a
        precomputed quadratic form (float) handed over by SPR.test_nll's fused path."""
        a = self.a.safe_value
        b = self.b.safe_value
        cov_data, y_data = aux
        y_data = np.asarray(y_data, dtype=np.float64)
        num_data = y_data.shape[-1]
        df = 2 * a
        cond_df = df + num_data
        if isinstance(cov_data, float):
            quad = cov_data
        else:
            if getattr(cov_data, "dtype", None) == np.float32:
                # 1e-6 jitter is below fp32 resolution of an O(1) kernel: factor this one in fp64
                cov_data = as_device(np.asarray(cov_data, dtype=np.float64), getattr(cov_data, "ctx", None))
            quad, _, _ = factor_stats(y_data, (b / a) * cov_data + jitter(num_data))
        d = df + quad
        sigma = np.sqrt(d / cond_df * b / a * _diag64(cov))
        return _t_logpdf(np.asarray(x, dtype=np.float64), cond_df, np.asarray(mean, dtype=np.float64), sigma)
