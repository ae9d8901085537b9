"""Host-side holder of the NNGP kernel's three positive hyper-parameters — the counterpart of the class the
reference defines at spax/kernels.py:9-41 (same public names: K, predict, get_params, get_kernel_fn; the trainables
w_std / b_std / last_w_std are the names its checkpoint reader looks for, experiments/regression/test.py:38-43).

Nothing numerical happens here: a kernel function built by `nt_kernels` is a handle on the fused device kernels, and
`predict` hands (kernel_fn, x, y) to the augmented-Cholesky posterior of `predict.py`.
"""
from __future__ import annotations

from ..predict import gradient_descent_mse_ensemble
from .base import ConstraintTrainVar, Module
from .bijectors import positive

__all__ = ["NNGPKernel"]

_TRAINABLES = ("w_std", "b_std", "last_w_std")


class NNGPKernel(Module):
    """`get_kernel_fn(w_std, b_std, last_w_std) -> kernel_fn` is any of the `nt_kernels.get_*_kernel` factories with the
    architecture arguments bound (experiments/regression/train.py:128-134)."""

    def __init__(self, get_kernel_fn, w_std: float = 1.0, b_std: float = 1.0, last_w_std: float = 1.0):
        super().__init__()
        if not callable(get_kernel_fn):
            raise TypeError("get_kernel_fn must be callable: (w_std, b_std, last_w_std) -> kernel_fn")
        self._get_kernel_fn = get_kernel_fn
        for name, value in zip(_TRAINABLES, (w_std, b_std, last_w_std)):
            setattr(self, name, ConstraintTrainVar(value, constraint=positive()))   # stored softplus-inverse

    # -- current (constrained) values -------------------------------------------------------------------------
    def get_params(self):
        """(w_std, b_std, last_w_std) as positive floats — spax/kernels.py:34-35."""
        return tuple(getattr(self, name).safe_value for name in _TRAINABLES)

    def get_kernel_fn(self):
        """A kernel function at the current hyper-parameters; rebuilt on every call because the trainables may have
        moved since the last one (spax/kernels.py:37-41)."""
        return self._get_kernel_fn(*self.get_params())

    # -- the two uses the model makes of a kernel function ----------------------------------------------------
    def K(self, kernel_fn, x, x2=None):
        """NNGP kernel matrix of x against x2 (against itself when x2 is None) — spax/kernels.py:23-27.  Passing x
        twice, as the reference does, would hide the symmetry from the device kernel; None keeps the lower-tile path."""
        other = None if (x2 is None or x2 is x) else x2
        return kernel_fn(x, other, get="nngp")

    def predict(self, kernel_fn, x, y, x_test, eps=1e-6):
        """Posterior mean [T, C] and covariance [T, T] at x_test — spax/kernels.py:29-32.  `eps` is neural_tangents'
        RELATIVE ridge (diag_reg), not the absolute jitter of the likelihood heads."""
        posterior = gradient_descent_mse_ensemble(kernel_fn, x, y, diag_reg=eps)
        return tuple(posterior(x_test=x_test, get="nngp", compute_cov=True))
