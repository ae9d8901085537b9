"""Training step for SPR — the counterpart of experiments/regression/train.py:61-67
(`objax.GradValues(model.loss, vars)` + `objax.optimizer.Adam`) without autodiff.

The model has at most six trainable scalars (w_std, b_std, last_w_std, eps and, for the Student-t
likelihood, a, b — the names experiments/regression/test.py:38-43 matches checkpoints by), all stored
as softplus-inverse raw values (spax/base.py:15-25).  The gradient of the loss with respect to those raw
values is taken by central differences: 2 loss evaluations (= 2 fused build + Cholesky passes on the GPU)
per variable.  This is SURVEY.md section 8f.1's finite-difference fallback (use float64 data: in float32 the loss
carries ~1e-6 relative noise and the quotient is dominated by it).  The analytic form, 1/2 tr((c aa^T - K^-1)
dK/dtheta) with forward-mode dK/dtheta through the layer recursion, is SPR.loss_and_grad (csrc/grad.hip);
build_train_step prefers it when the model supports it.
"""
from __future__ import annotations

import math

import numpy as np

from .spax.base import TrainVar

__all__ = ["train_vars", "value_and_grad", "value_and_grad_fd", "Adam", "PlateauSchedule", "build_train_step"]


def train_vars(model):
    """Dotted-name -> TrainVar for every trainable of the model (kernel, likelihood, eps)."""
    return {k: v for k, v in model.vars().items() if isinstance(v, TrainVar)}


def value_and_grad(model, variables=None):
    """(loss, {name: dloss/draw}) from the analytic gradient (SPR.loss_and_grad): ONE augmented factorisation
    and one contraction pass instead of 2 loss evaluations per variable, and usable in float32."""
    value, grads = model.loss_and_grad()
    if variables is not None:
        grads = {k: g for k, g in grads.items() if k in variables}
    return value, grads


def value_and_grad_fd(loss_fn, variables, h=1e-4):
    """(loss, {name: dloss/draw}) by central differences on the raw (unconstrained) values."""
    value = float(loss_fn())
    grads = {}
    for name, var in variables.items():
        raw = float(var.value)
        step = h * max(1.0, abs(raw))
        var.assign(raw + step)
        up = float(loss_fn())
        var.assign(raw - step)
        dn = float(loss_fn())
        var.assign(raw)
        grads[name] = (up - dn) / (2.0 * step)
    return value, grads


class Adam:
    """objax.optimizer.Adam defaults (beta1=0.9, beta2=0.999, eps=1e-8), called as optimizer(lr, grads)."""

    def __init__(self, variables, beta1=0.9, beta2=0.999, eps=1e-8):
        self.vars, self.b1, self.b2, self.eps = variables, beta1, beta2, eps
        self.m = {k: 0.0 for k in variables}
        self.v = {k: 0.0 for k in variables}
        self.t = 0

    def __call__(self, lr, grads):
        self.t += 1
        lr_t = lr * math.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        for k, g in grads.items():
            if not np.isfinite(g):
                continue
            self.m[k] = self.b1 * self.m[k] + (1 - self.b1) * g
            self.v[k] = self.b2 * self.v[k] + (1 - self.b2) * g * g
            self.vars[k].assign(float(self.vars[k].value) - lr_t * self.m[k] / (math.sqrt(self.v[k]) + self.eps))


def build_train_step(model, variables=None, optimizer=None, h=1e-4, method="auto"):
    """train_step(learning_rate) -> loss before the update  (regression/train.py:61-67).
    method: "analytic" (SPR.loss_and_grad), "fd" (central differences) or "auto" (analytic when the model's
    kernel / likelihood support it, else finite differences)."""
    if method not in ("auto", "analytic", "fd"):
        raise ValueError("method must be 'auto', 'analytic' or 'fd'")
    variables = variables if variables is not None else train_vars(model)
    optimizer = optimizer or Adam(variables)
    state = {"analytic": method != "fd" and hasattr(model, "loss_and_grad")}

    def train_step(learning_rate):
        if state["analytic"]:
            try:
                value, grads = value_and_grad(model, variables)
            except NotImplementedError:
                if method == "analytic":
                    raise
                state["analytic"] = False
        if not state["analytic"]:
            value, grads = value_and_grad_fd(model.loss, variables, h=h)
        optimizer(learning_rate, grads)
        return value

    return train_step


class PlateauSchedule:
    """Learning-rate decay on a validation plateau, with the interface the regression run uses
    (experiments/regression/train.py:159,198,208-211; experiments/utils.py:153-231): `.lr` is the current rate,
    `.step(metric)` records one validation result and returns True when it just cut the rate by `factor`.
    A result counts as an improvement when it beats the best one by more than `threshold` (relative by default);
    after more than `patience` results without improvement the rate is multiplied by `factor`, not below `min_lr`."""

    def __init__(self, lr, mode="min", factor=0.1, patience=10, threshold=1e-4, threshold_mode="rel", min_lr=0.0,
                 eps=1e-8):
        if mode not in ("min", "max"):
            raise ValueError("mode " + str(mode) + " is unknown!")
        if threshold_mode not in ("rel", "abs"):
            raise ValueError("threshold mode " + str(threshold_mode) + " is unknown!")
        self.lr, self.factor, self.patience, self.min_lr, self.eps = float(lr), factor, patience, min_lr, eps
        sign = 1.0 if mode == "min" else -1.0               # work on sign * metric: smaller is always better
        if threshold_mode == "rel":
            # min: a < best (1 - t);  max: a > best (1 + t)  <=>  -a < -best (1 + t)
            self._margin = lambda best: best * (1.0 - sign * threshold)
        else:
            self._margin = lambda best: best - threshold
        self._sign = sign
        self._best = float("inf")
        self.stale = 0
        self.calls = 0

    @property
    def best(self):
        return self._sign * self._best

    def step(self, metric):
        value = self._sign * float(metric)
        self.calls += 1
        if value < self._margin(self._best):
            self._best, self.stale = value, 0
            return False
        self.stale += 1
        if self.stale <= self.patience:
            return False
        self.stale = 0
        lowered = max(self.lr * self.factor, self.min_lr)
        if self.lr - lowered > self.eps:
            self.lr = lowered
        return True

