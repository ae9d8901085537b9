"""Kernel factories — same names, arguments and defaults as experiments/nt_kernels.py:21-103.

Each factory returns ``kernel_fn(x1, x2=None, get="nngp")`` like neural_tangents' stax.serial does;
here it is a small object that launches the fused HIP build (smn_kernel_mlp) and returns device
arrays.  ``get`` accepts "nngp", "ntk" or a tuple of both (a namedtuple-like Kernels result).
"""
from __future__ import annotations

import collections

import numpy as np

from . import _lib
from ._lib import DeviceArray, as_device, default_context

__all__ = ["get_mlp_kernel", "get_cnn_kernel", "get_conv_resnet_kernel", "get_dense_resnet_kernel"]

Kernels = collections.namedtuple("Kernels", ["nngp", "ntk"])


def get_act_class(act):
    """experiments/nt_kernels.py:12-18 — KeyError for anything but relu / erf."""
    if act == "relu":
        return _lib.ACT["relu"]
    elif act == "erf":
        return _lib.ACT["erf"]
    else:
        raise KeyError("Unsupported act '{}'".format(act))


def _parse_get(get):
    if isinstance(get, str):
        names = (get,)
    else:
        names = tuple(get)
    mask = 0
    for g in names:
        if g == "nngp":
            mask |= _lib.GET_NNGP
        elif g == "ntk":
            mask |= _lib.GET_NTK
        else:
            raise ValueError("get must be 'nngp', 'ntk' or a tuple of them, got %r" % (g,))
    return names, mask


class KernelFn:
    """kernel_fn of an MLP-family architecture (net = MLP or dense ResNet)."""

    def __init__(self, net, num_hiddens, act, w_std, b_std, last_w_std, ctx=None):
        self.net, self.num_hiddens = int(net), int(num_hiddens)
        self.act_name, self.act = act, get_act_class(act)
        self.w_std, self.b_std, self.last_w_std = float(w_std), float(b_std), float(last_w_std)
        self.ctx = ctx

    @property
    def params(self):
        return (self.net, self.act, self.num_hiddens, self.w_std, self.b_std, self.last_w_std)

    def __call__(self, x1, x2=None, get="nngp", fill="full"):
        ctx = self.ctx or (x1.ctx if isinstance(x1, DeviceArray) else default_context())
        names, mask = _parse_get(get)
        a = as_device(x1, ctx)
        if len(a.shape) != 2:
            a_host = a.numpy().reshape(a.shape[0], -1)
            a = ctx.to_device(a_host)
        b = None
        if x2 is not None and x2 is not x1:
            b = as_device(x2, ctx, dtype=a.dtype)
            if len(b.shape) != 2:
                b = ctx.to_device(b.numpy().reshape(b.shape[0], -1))
            if b.shape[1] != a.shape[1]:
                raise ValueError("x1 and x2 feature dimensions differ: %s vs %s" % (a.shape, b.shape))
            if b.dtype != a.dtype:
                raise TypeError("x1 and x2 dtypes differ")
        n1, d = a.shape
        n2 = n1 if b is None else b.shape[0]
        k = ctx.empty((n1, n2), a.dtype) if mask & _lib.GET_NNGP else None
        t = ctx.empty((n1, n2), a.dtype) if mask & _lib.GET_NTK else None
        ctx.call("smn_kernel_mlp", a.dcode, self.net, self.act, self.num_hiddens, self.w_std, self.b_std,
                 self.last_w_std, a.ptr, n1, d, None if b is None else b.ptr, n2, d, d, mask,
                 _lib.FILL_FULL if fill == "full" else _lib.FILL_LOWER,
                 None if k is None else k.ptr, None if t is None else t.ptr, n2)
        out = {"nngp": k, "ntk": t}
        if isinstance(get, str):
            return out[get]
        if len(names) == 2:
            return Kernels(**{n: out[n] for n in names}) if set(names) == {"nngp", "ntk"} else tuple(out[n] for n in names)
        return tuple(out[n] for n in names)


def get_mlp_kernel(num_hiddens, num_class=1, act="relu", w_std=1., b_std=0., last_w_std=1.):
    """experiments/nt_kernels.py:21-31.  `num_class` and the hidden width (512) do not enter the kernel."""
    return KernelFn(_lib.NET_MLP, num_hiddens, act, w_std, b_std, last_w_std)


def get_dense_resnet_kernel(num_hiddens, num_class=1, act="relu", w_std=1., b_std=0., last_w_std=1.):
    """experiments/nt_kernels.py:83-103."""
    return KernelFn(_lib.NET_DENSE_RESNET, num_hiddens, act, w_std, b_std, last_w_std)


class CnnKernelFn:
    """kernel_fn of get_cnn_kernel (experiments/nt_kernels.py:34-45) or, with entry="smn_kernel_conv_resnet", of
    get_conv_resnet_kernel (:48-80, num_hiddens = block size); x is [N,H,W,C]; NNGP only."""

    def __init__(self, num_hiddens, act, w_std, b_std, last_w_std, ctx=None, entry="smn_kernel_cnn"):
        self.num_hiddens, self.act_name, self.act = int(num_hiddens), act, get_act_class(act)
        self.w_std, self.b_std, self.last_w_std = float(w_std), float(b_std), float(last_w_std)
        self.ctx = ctx
        self.entry = entry

    def __call__(self, x1, x2=None, get="nngp", fill="full"):
        if get != "nngp":
            raise NotImplementedError("conv kernel: only get='nngp' is on the hot path")
        ctx = self.ctx or (x1.ctx if isinstance(x1, DeviceArray) else default_context())
        a = as_device(x1, ctx)
        if len(a.shape) != 4:
            raise ValueError("conv kernel expects x of shape [N,H,W,C]")
        b = None
        if x2 is not None and x2 is not x1:
            b = as_device(x2, ctx, dtype=a.dtype)
            if b.shape[1:] != a.shape[1:]:
                raise ValueError("x1 and x2 image shapes differ")
        n1, h, w, c = a.shape
        n2 = n1 if b is None else b.shape[0]
        k = ctx.empty((n1, n2), a.dtype)
        ctx.call(self.entry, a.dcode, self.act, self.num_hiddens, self.w_std, self.b_std, self.last_w_std,
                 a.ptr, n1, None if b is None else b.ptr, n2, h, w, c,
                 _lib.FILL_FULL if fill == "full" else _lib.FILL_LOWER, k.ptr, n2)
        return k


def get_cnn_kernel(num_hiddens, num_class=1, act="relu", w_std=1., b_std=0., last_w_std=1.):
    """experiments/nt_kernels.py:34-45."""
    return CnnKernelFn(num_hiddens, act, w_std, b_std, last_w_std)


def get_conv_resnet_kernel(num_hiddens, num_class, act="relu", w_std=1., b_std=0., last_w_std=1.):
    """experiments/nt_kernels.py:48-80 — WideResnet(block_size=num_hiddens, k=1) without pooling; x [N,H,W,C] with H, W
    multiples of 8 (three stride-2 stages); NNGP only.  `num_class` does not enter the kernel."""
    return CnnKernelFn(num_hiddens, act, w_std, b_std, last_w_std, entry="smn_kernel_conv_resnet")
