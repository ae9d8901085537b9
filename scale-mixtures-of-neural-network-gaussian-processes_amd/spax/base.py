"""spax/base.py mirror: Module / TrainVar / ConstraintTrainVar without objax (host scalars only)."""
from __future__ import annotations

import numpy as np

__all__ = ["Module", "TrainVar", "ConstraintTrainVar"]


class TrainVar:
    def __init__(self, tensor):
        self._value = np.asarray(tensor, dtype=np.float64)

    @property
    def value(self):
        return self._value

    @value.setter
    def value(self, v):
        self._value = np.asarray(v, dtype=np.float64)

    def assign(self, v):
        self.value = v

    def __repr__(self):
        return "%s(%r)" % (type(self).__name__, self._value)


class ConstraintTrainVar(TrainVar):
    """Stores constraint.inverse(x), exposes safe_value = constraint(raw) — spax/base.py:15-25."""

    def __init__(self, tensor, constraint):
        super().__init__(constraint.inverse(np.asarray(tensor, dtype=np.float64)))
        self.constraint = constraint

    @property
    def safe_value(self):
        return float(self.constraint(self._value))

    def __repr__(self):
        return super().__repr__()[:-1] + ", constraint=%s)" % type(self.constraint).__name__


class Module:
    """Minimal objax.Module stand-in.  vars() names follow objax's scoping rule -- "(SPR).kernel(NNGPKernel).w_std",
    "(SPR).eps", "(SPR).likelihood(StudentTLikelihood).a" -- so a collection saved here carries the names a
    reference run would have written, and the reference's checkpoint reader (which matches by the last dotted
    component, experiments/regression/test.py:38-43) finds the same keys."""

    def vars(self, scope=""):
        out = {}
        scope += "(%s)." % type(self).__name__
        for k, v in vars(self).items():
            if isinstance(v, TrainVar):
                out[scope + k] = v
            elif isinstance(v, Module):
                out.update(v.vars(scope + k))
        return out
