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
    """Minimal objax.Module stand-in: vars() collects TrainVars by dotted name (the keys the
    reference's checkpoint reader matches by last component, experiments/regression/test.py:38-43)."""

    def vars(self, prefix=""):
        out = {}
        for k, v in vars(self).items():
            name = "%s.%s" % (prefix, k) if prefix else k
            if isinstance(v, TrainVar):
                out["(%s).%s" % (type(self).__name__, name)] = v
            elif isinstance(v, Module):
                out.update(v.vars(name))
        return out
