"""spax/bijectors.py mirror (positive constraint: softplus default, exp alternative)."""
from __future__ import annotations

import numpy as np

__all__ = ["positive", "triangular", "Softplus", "Exp"]


class PositiveBijector:
    def __init__(self, lower=0.0):
        self.lower = lower

    def __call__(self, x):
        return self.lower + self.base(x)

    def inverse(self, x):
        return self.base_inv(np.asarray(x, dtype=np.float64) - self.lower)

    def grad(self, x):
        """d constrained / d raw at raw value x (chain rule of the analytic hyper-parameter gradients)."""
        return self.base_grad(np.asarray(x, dtype=np.float64))


class Exp(PositiveBijector):
    def base(self, x):
        return np.exp(x)

    def base_inv(self, x):
        return np.log(x)

    def base_grad(self, x):
        return np.exp(x)


class Softplus(PositiveBijector):
    def base(self, x):
        return np.logaddexp(x, 0.0)

    def base_inv(self, x):                       # spax/bijectors.py:53 — identity from 20 upwards
        x = np.asarray(x, dtype=np.float64)
        return np.where(x < 20.0, np.log(np.expm1(np.minimum(x, 20.0))), x)

    def base_grad(self, x):                      # d softplus = sigmoid
        return 0.5 * (1.0 + np.tanh(0.5 * x))


_TYPES = {"exp": Exp, "softplus": Softplus}


def positive(lower=None, base=None):
    lower_bound = lower if lower is not None else 0.0
    name = base if base is not None else "softplus"
    return _TYPES[name.lower()](lower_bound)


def triangular():
    raise NotImplementedError
