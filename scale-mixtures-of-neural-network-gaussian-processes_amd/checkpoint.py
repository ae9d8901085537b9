"""Checkpoint compatibility with the reference's regression runs.

A reference run directory holds
  * ``NNN.npz``   -- ``objax.io.save_var_collection``: ``names`` (array of dotted variable names) and one
                    entry per variable keyed "0", "1", ... holding the RAW tensor (for a
                    ``ConstraintTrainVar`` that is softplus-inverse of the hyper-parameter);
                    written by ``Checkpointer.save`` (experiments/utils.py:98-127);
  * ``meta.npy``  -- a pickled ``dict(args=vars(args))`` (experiments/regression/train.py:167).
``experiments/regression/test.py:38-53,89-130`` reads them back, matching variables by the LAST dotted
component of their name (a, b, w_std, b_std, last_w_std, eps | diag_reg), and rebuilds the SPR model.
This module reads and writes the same layout, so hyper-parameters trained by the reference can be
evaluated by this engine and vice versa.  Host-only except ``restore_spr`` (the model uploads X).
"""
from __future__ import annotations

import glob
import os

import numpy as np

__all__ = ["save_var_collection", "load_var_collection", "get_from_vars", "Checkpointer", "save_meta",
           "load_meta", "latest_index", "read_run", "restore_spr"]

FILE_MATCH = "*.npz"
FILE_FORMAT = "{:03d}.npz"
HYPER_KEYS = ("a", "b", "w_std", "b_std", "last_w_std", "eps")


def save_var_collection(path, vc):
    """``vc``: dotted name -> TrainVar (``Module.vars()``).  Same file layout as objax's writer: shared
    variables are stored once, values are the raw tensors."""
    names, data, seen = [], {}, set()
    for name, var in vc.items():
        if id(var) in seen:
            continue
        seen.add(id(var))
        data[str(len(names))] = np.asarray(var.value)
        names.append(name)
    with open(path, "wb") as f:            # a file object: np.savez would append ".npz" to a bare name
        np.savez(f, names=np.array(names), **data)


def load_var_collection(path):
    """The stored arrays as a plain dict (``names`` plus "0", "1", ...)."""
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def get_from_vars(saved_vars, key):
    """experiments/regression/test.py:38-43 -- first variable whose last dotted component equals ``key``."""
    for i, name in enumerate(saved_vars["names"]):
        if key == str(name).split(".")[-1]:
            return saved_vars[str(i)]
    return None


class Checkpointer:
    """experiments/utils.py:98-127: keep the ``keep_ckpts`` most recent files, save on a new best loss."""

    FILE_MATCH = FILE_MATCH
    FILE_FORMAT = FILE_FORMAT

    def __init__(self, logdir, keep_ckpts=10, makedir=True):
        self.logdir = logdir
        self.keep_ckpts = keep_ckpts
        if makedir:
            os.makedirs(logdir, exist_ok=True)
        self.best_loss = float("inf")

    def save(self, idx, vc):
        if not isinstance(vc, dict):
            raise TypeError("Must pass a variable collection (Module.vars()) to save; received %s" % type(vc))
        save_var_collection(os.path.join(self.logdir, self.FILE_FORMAT.format(idx)), vc)
        for ckpt in sorted(glob.glob(os.path.join(self.logdir, self.FILE_MATCH)))[:-self.keep_ckpts]:
            os.remove(ckpt)

    def step(self, idx, loss, vc):
        if loss < self.best_loss:
            self.best_loss = loss
            self.save(idx, vc)
            return True
        return False


def save_meta(ckpt_dir, args):
    """``args``: the run's argument dict (method, network, num_hiddens, activation, data_name, last_w_std ...)."""
    np.save(os.path.join(ckpt_dir, "meta.npy"), dict(args=dict(args)))


def load_meta(ckpt_dir):
    """The ``args`` dict of the run.  meta.npy is a pickle (that is the reference's format): only load run
    directories you trust."""
    return np.load(os.path.join(ckpt_dir, "meta.npy"), allow_pickle=True).item()["args"]


def latest_index(ckpt_dir):
    """experiments/regression/test.py:47-49 -- the highest NNN among NNN.npz."""
    idx = []
    for ckpt in glob.glob(os.path.join(ckpt_dir, FILE_MATCH)):
        stem = "".join(os.path.basename(ckpt).split(".")[:-1])
        if stem.isdigit():
            idx.append(int(stem))
    if not idx:
        raise FileNotFoundError("no NNN.npz checkpoint under %r" % ckpt_dir)
    return max(idx)


def read_run(ckpt_dir, ckpt_index=None):
    """(raw hyper-parameter dict, context args) of a run directory, with test.py's fallbacks: ``eps`` may be
    stored as ``diag_reg`` (:100-101), a missing ``last_w_std`` comes from the run's arguments, already
    constrained there (:103-104)."""
    if ckpt_index is None:
        ckpt_index = latest_index(ckpt_dir)
    saved = load_var_collection(os.path.join(ckpt_dir, FILE_FORMAT.format(ckpt_index)))
    context = load_meta(ckpt_dir)
    raw = {k: get_from_vars(saved, k) for k in HYPER_KEYS}
    if raw["eps"] is None:
        raw["eps"] = get_from_vars(saved, "diag_reg")
    if raw["last_w_std"] is None:
        # test.py assigns np.array(context["last_w_std"]) as the raw value as it stands; keep that reading
        raw["last_w_std"] = np.array(context["last_w_std"], dtype=np.float64)
    return raw, context


def restore_spr(ckpt_dir, x_train, y_train, y_mean, y_std, ckpt_index=None, dtype=np.float32):
    """experiments/regression/test.py:89-130: rebuild SPR from a run directory and assign the stored raw values."""
    from . import nt_kernels
    from .spax.kernels import NNGPKernel
    from .spax.likelihoods import GaussianLikelihood, StudentTLikelihood
    from .spax.models import SPR
    raw, context = read_run(ckpt_dir, ckpt_index)
    network = context.get("network") or "mlp"
    if network == "mlp":
        base_kernel_fn = nt_kernels.get_mlp_kernel
    elif network == "resnet":
        base_kernel_fn = nt_kernels.get_dense_resnet_kernel
    else:
        raise ValueError("Unsupported network '%s'" % network)
    num_hiddens, activation = context["num_hiddens"], context["activation"]

    def get_kernel_fn(w_std, b_std, last_w_std):
        return base_kernel_fn(num_hiddens, act=activation, w_std=w_std, b_std=b_std, last_w_std=last_w_std)

    kernel = NNGPKernel(get_kernel_fn, 1.0, 1.0, 1.0)
    method = context["method"]
    if method == "gp":
        likelihood = GaussianLikelihood()
    elif method == "tp":
        likelihood = StudentTLikelihood(1, 1)
    else:
        raise ValueError("Unsupported method '%s'" % method)
    model = SPR(kernel, likelihood, np.asarray(x_train, dtype=dtype), np.asarray(y_train, dtype=dtype),
                y_mean, y_std, eps=1)
    model.eps.assign(raw["eps"])
    model.kernel.w_std.assign(raw["w_std"])
    model.kernel.b_std.assign(raw["b_std"])
    model.kernel.last_w_std.assign(raw["last_w_std"])
    if method == "tp":                       # the Gaussian likelihood has no a / b (test.py assigns them blindly)
        if raw["a"] is not None:
            model.likelihood.a.assign(raw["a"])
        if raw["b"] is not None:
            model.likelihood.b.assign(raw["b"])
    return model, context
