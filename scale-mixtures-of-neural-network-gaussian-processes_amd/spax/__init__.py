"""spax — host-side mirror of the hot-path surface of the reference library of the same name.

Sub-modules (import them by name, as the reference's experiments do):
    spax.models       SPR
    spax.kernels      NNGPKernel
    spax.likelihoods  GaussianLikelihood, StudentTLikelihood
    spax.bijectors    positive
    spax.utils        jitter, multivariate_normal_logpdf, multivariate_t_logpdf, ...
The reference's `priors` module belongs to its sparse variational classifier and is not part of this path.
"""
import importlib as _importlib

from .base import ConstraintTrainVar, Module, TrainVar

for _name in ("bijectors", "utils", "likelihoods", "kernels", "models"):
    globals()[_name] = _importlib.import_module("." + _name, __name__)
del _name

__all__ = ["Module", "TrainVar", "ConstraintTrainVar", "bijectors", "utils", "likelihoods", "kernels", "models"]
