"""ctypes binding of libsmnngp.so (include/smnngp.h) + the device-array type of the facade.

The binding fails loudly: a missing library raises ImportError, a failing call raises
SmnError with the library's message.  There is no CPU path behind any of these calls.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SMNNGP_LIB") or os.path.join(_HERE, "libsmnngp.so")   # override: A/B builds only

OK, EINVAL, EHIP, ENOMEM, ENOTSUP, ECOMM = 0, -1, -2, -3, -4, -5
F32, F64 = 0, 1
ACT = {"relu": 0, "erf": 1}
GET_NNGP, GET_NTK = 1, 2
FILL_FULL, FILL_LOWER = 0, 1
NET_MLP, NET_DENSE_RESNET = 0, 1


class SmnError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libsmnngp error %d: %s" % (code, msg))
        self.code = code


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libsmnngp.so is not built (%s). Build it with "
            "`python scale-mixtures-of-neural-network-gaussian-processes_amd/build.py`; "
            "there is no CPU fallback." % LIB_PATH)
    return C.CDLL(LIB_PATH)


_lib = _load()

_vp, _i, _i64, _d, _sz = C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_size_t
_pi, _pd, _pvp = C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_void_p)
_pi64 = C.POINTER(C.c_int64)

# name -> argtypes  (restype is always int).  Mirrors include/smnngp.h one to one.
PROTOTYPES = {
    "smn_version": [],
    "smn_device_count": [_pi],
    "smn_ctx_create": [_i, _pvp],
    "smn_ctx_destroy": [_vp],
    "smn_last_error": [_vp, C.c_char_p, _sz],
    "smn_synchronize": [_vp],
    "smn_malloc": [_vp, _sz, _pvp],
    "smn_free": [_vp, _vp],
    "smn_memset": [_vp, _vp, _i, _sz],
    "smn_memcpy_h2d": [_vp, _vp, _vp, _sz],
    "smn_memcpy_d2h": [_vp, _vp, _vp, _sz],
    "smn_memcpy_d2d": [_vp, _vp, _vp, _sz],
    "smn_memcpy2d_h2d": [_vp, _vp, _sz, _vp, _sz, _sz, _sz],
    "smn_memcpy2d_d2h": [_vp, _vp, _sz, _vp, _sz, _sz, _sz],
    "smn_timer_start": [_vp],
    "smn_timer_stop_ms": [_vp, _pd],
    "smn_profile_enable": [_vp, _i],
    "smn_profile_read": [_vp, _i, _pd, _pi],
    "smn_profile_flops": [_vp, _i, _pd],
    "smn_kernel_mlp": [_vp, _i, _i, _i, _i, _d, _d, _d, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _i, _i, _vp, _vp, _i64],
    "smn_kernel_mlp_rows": [_vp, _i, _i, _i, _i, _d, _d, _d, _vp, _i64, _i64, _i64, _i64, _i64, _i, _vp, _vp, _i64],
    "smn_kernel_mlp_lower_rows": [_vp, _i, _i, _i, _i, _d, _d, _d, _vp, _i64, _i64, _i64, _i64, _i64, _i, _vp, _vp, _i64],
    "smn_kernel_mlp_shard": [_vp, _i, _i, _i, _i, _d, _d, _d, _vp, _i64, _i64, _i64, _i, _i, _i64, _i, _vp, _vp],
    "smn_gram": [_vp, _i, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _vp, _vp],
    "smn_recursion": [_vp, _i, _i, _i, _i, _d, _d, _d, _vp, _i64, _i64, _i64, _vp, _vp, _i, _i, _vp, _vp, _i64],
    "smn_kernel_cnn": [_vp, _i, _i, _i, _d, _d, _d, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _i, _vp, _i64],
    "smn_kernel_conv_resnet": [_vp, _i, _i, _i, _d, _d, _d, _vp, _i64, _vp, _i64, _i64, _i64, _i64, _i, _vp, _i64],
    "smn_cholesky": [_vp, _i, _vp, _i64, _i64, _i64, _i64, _d, _d, _pi, _pd],
    "smn_trsm": [_vp, _i, _vp, _i64, _i64, _vp, _i64, _i64, _i],
    "smn_transpose": [_vp, _i, _vp, _i64, _vp, _i64, _i64, _i64],
    "smn_lml": [_vp, _i, _vp, _i64, _i64, _vp, _d, _d, _d, _pd, _pd, _pd, _pi],
    "smn_predict": [_vp, _i, _vp, _i64, _i64, _i64, _vp, _i64, _d, _d, _vp, _vp, _i64, _pd, _pd, _pi],
    "smn_spr_loss": [_vp, _i, _i, _i, _i, _d, _d, _d, _vp, _i64, _i64, _i64, _vp, _d, _d, _d, _pd, _pd, _pd, _pi],
    "smn_spr_predict": [_vp, _i, _i, _i, _i, _d, _d, _d, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _d, _d,
                        _vp, _vp, _i64, _pd, _pd, _pi],
    "smn_spr_loss_batch": [_vp, _i, _i, _i, _i, _i, _pd, _pd, _pd, _vp, _i64, _i64, _i64, _vp, _pd, _pd, _pd, _pd, _pd, _pd, _pi],
    "smn_spr_predict_batch": [_vp, _i, _i, _i, _i, _i, _pd, _pd, _pd, _vp, _i64, _i64, _vp, _i64, _i64, _i64, _vp, _i64, _pd, _pd,
                              _vp, _vp, _i64, _vp, _pd, _pd, _pi],
    "smn_debug_batch_bytes": [_vp, _sz],
    "smn_debug_split_build": [_vp, C.c_int],
    "smn_debug_panel_passes": [_vp, C.c_int],
    "smn_mixture_nll": [_vp, _i, _i, _i64, _vp, _vp, _pd, _pd, _pi, _pd, _d, _d, _i64, _i, _i, _pd, _pd, _pd],
    "smn_comm_unique_id": [C.c_char_p],
    "smn_lml_grad_terms": [_vp, _i, _i, _i, _i, _d, _d, _d, _vp, _i64, _i64, _vp, _vp, _i64, _vp, _d, _pd],
    "smn_spr_loss_grad": [_vp, _i, _i, _i, _i, _d, _d, _d, _vp, _i64, _i64, _i64, _vp, _d, _d, _d, _pd, _pd, _pi, _pd],
    "smn_comm_init": [_vp, _i, _i, C.c_char_p],
    "smn_comm_destroy": [_vp],
    "smn_allgather": [_vp, _i, _i, _vp, _vp, _i64],
    "smn_unpack_lower_blocks": [_vp, _i, _vp, _i64, _i, _i64, _vp, _i64],
    "smn_lml_from_blocks": [_vp, _i, _vp, _i64, _i, _i64, _vp, _d, _d, _d, _pd, _pd, _pd, _pi],
    "smn_comm_info": [_vp, _pi, _pi],
    "smn_shard_begin": [_vp, _i, _i64, _d],
    "smn_kernel_mlp_shard_cols": [_vp, _i, _i, _i, _i, _d, _d, _d, _vp, _i64, _i64, _i64, _i, _i, _i, _pi64, _i, _vp, _vp],
    "smn_shard_exchange_cols": [_vp, _i, _vp, _vp, _i64, _i, _i, _pi64, _i],
    "smn_shard_exchange_cols_to": [_vp, _i, _vp, _vp, _i64, _i, _i, _pi64, _i, _vp, _i64],
    "smn_shard_scatter_cols": [_vp, _i, _vp, _i64, _i, _i, _pi64, _i, _vp, _i64],
    "smn_shard_wait": [_vp],
    "smn_lml_from_shards": [_vp, _i, _i64, _vp, _d, _d, _pd, _pd, _pd, _pi],
    "smn_debug_delay": [_vp, _i, _i64],
}
for _name, _args in PROTOTYPES.items():
    _fn = getattr(_lib, _name)          # AttributeError here == a symbol the header declares is missing
    _fn.argtypes = _args
    _fn.restype = C.c_int


def np_dtype(code):
    return np.float64 if code == F64 else np.float32


def dtype_code(dt):
    dt = np.dtype(dt)
    if dt == np.float32:
        return F32
    if dt == np.float64:
        return F64
    raise TypeError("smnngp supports float32 / float64 arrays, got %s" % dt)


class Context:
    """One HIP device context (stream + workspaces).  Not thread-safe."""

    def __init__(self, device=None):
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        n = C.c_int(0)
        _lib.smn_device_count(C.byref(n))
        if n.value <= 0:
            raise SmnError(EHIP, "no HIP device visible: smnngp has no CPU path")
        h = C.c_void_p()
        rc = _lib.smn_ctx_create(int(device), C.byref(h))
        if rc != OK:
            raise SmnError(rc, "smn_ctx_create(device=%d) failed" % device)
        self.handle = h
        self.device = int(device)

    def call(self, name, *args):
        rc = getattr(_lib, name)(self.handle, *args)
        if rc != OK:
            buf = C.create_string_buffer(512)
            _lib.smn_last_error(self.handle, buf, 512)
            raise SmnError(rc, "%s: %s" % (name, buf.value.decode(errors="replace")))

    def synchronize(self):
        self.call("smn_synchronize")

    def close(self):
        if getattr(self, "handle", None):
            _lib.smn_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- arrays
    def empty(self, shape, dtype):
        return DeviceArray(self, tuple(int(s) for s in shape), np.dtype(dtype))

    def to_device(self, host, dtype=None):
        host = np.ascontiguousarray(host, dtype=dtype if dtype is not None else None)
        if host.dtype not in (np.float32, np.float64):
            host = host.astype(np.float64 if host.dtype.itemsize >= 8 else np.float32)
        arr = DeviceArray(self, host.shape, host.dtype)
        if host.nbytes:
            self.call("smn_memcpy_h2d", arr.ptr, host.ctypes.data_as(C.c_void_p), host.nbytes)
        return arr


_default_ctx = None


def default_context():
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context()
    return _default_ctx


class ScaledIdentity:
    """eps * I without the N x N array — what spax.utils.jitter returns (spax/utils.py:26-27)."""

    def __init__(self, num, eps):
        self.num, self.eps = int(num), float(eps)

    def __array__(self, dtype=None, copy=None):
        return self.eps * np.eye(self.num, dtype=dtype or np.float64)

    def __rmul__(self, s):
        return ScaledIdentity(self.num, self.eps * float(s))

    __mul__ = __rmul__


class DeviceArray:
    """Row-major dense device array (1-D or 2-D).  Represents  scale * A + shift * I  lazily so that
    `K + jitter(n, eps)` and `(b/a) * cov` (spax/models.py:96, spax/likelihoods.py:49) cost nothing."""

    def __init__(self, ctx, shape, dtype, ptr=None, owner=True):
        self.ctx, self.shape, self.dtype = ctx, tuple(shape), np.dtype(dtype)
        self.size = int(np.prod(self.shape)) if self.shape else 1
        self.nbytes = self.size * self.dtype.itemsize
        self.scale, self.shift = 1.0, 0.0
        self._owner = owner
        if ptr is None:
            p = C.c_void_p()
            ctx.call("smn_malloc", max(self.nbytes, 16), C.byref(p))
            self.ptr = p
        else:
            self.ptr = ptr
        self._base = None

    @property
    def dcode(self):
        return dtype_code(self.dtype)

    @property
    def ld(self):
        return self.shape[-1] if len(self.shape) > 1 else 1

    def __del__(self):
        try:
            if self._owner and self.ptr is not None and self.ctx.handle:
                _lib.smn_free(self.ctx.handle, self.ptr)
        except Exception:
            pass

    def _view(self, scale, shift):
        v = DeviceArray(self.ctx, self.shape, self.dtype, ptr=self.ptr, owner=False)
        v._base = self if self._base is None else self._base   # keep the buffer alive
        v.scale, v.shift = scale, shift
        return v

    def __mul__(self, s):
        return self._view(self.scale * float(s), self.shift * float(s))

    __rmul__ = __mul__

    def __add__(self, other):
        if isinstance(other, ScaledIdentity):
            return self._view(self.scale, self.shift + other.eps)
        return NotImplemented

    __radd__ = __add__

    def raw_numpy(self):
        out = np.empty(self.shape, dtype=self.dtype)
        if self.nbytes:
            self.ctx.call("smn_memcpy_d2h", out.ctypes.data_as(C.c_void_p), self.ptr, self.nbytes)
        return out

    def numpy(self):
        out = self.raw_numpy()
        if self.scale != 1.0:
            out *= self.dtype.type(self.scale)
        if self.shift != 0.0:
            out[np.diag_indices(min(out.shape))] += self.dtype.type(self.shift)
        return out

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)

    def diagonal(self):
        """The main diagonal, fetched as one strided copy (not through a download of the whole matrix)."""
        if len(self.shape) != 2:
            return np.diagonal(self.numpy())
        m = min(self.shape)
        out = np.empty(m, dtype=self.dtype)
        if m:
            isz = self.dtype.itemsize
            self.ctx.call("smn_memcpy2d_d2h", out.ctypes.data_as(C.c_void_p), isz, self.ptr, (self.ld + 1) * isz, isz, m)
        if self.scale != 1.0:
            out *= self.dtype.type(self.scale)
        if self.shift != 0.0:
            out += self.dtype.type(self.shift)
        return out

    def flatten(self):
        return self.numpy().flatten()

    def __getitem__(self, idx):
        return self.numpy()[idx]

    def __repr__(self):
        return "DeviceArray(shape=%s, dtype=%s, device=%d)" % (self.shape, self.dtype, self.ctx.device)


def as_device(x, ctx=None, dtype=None):
    """numpy / DeviceArray -> a plain DeviceArray whose raw pointer may be handed to the library (float32 stays
    float32, everything else becomes float64 unless `dtype` says otherwise).

    A DeviceArray argument is checked, not trusted: one that lives in another Context raises (its pointer belongs to
    that context's device and stream); a lazy view `scale * A + shift * I` is materialised (the library would read A);
    a dtype other than the requested one is converted (it would be read with the wrong element size).  Both
    conversions go through the host: they are for small operands (x_test, y), large ones should arrive right."""
    if isinstance(x, DeviceArray):
        if ctx is not None and x.ctx is not ctx:
            raise ValueError("DeviceArray belongs to another Context (device %d) than the one in use (device %d)"
                             % (x.ctx.device, ctx.device))
        want = x.dtype if dtype is None else np.dtype(dtype)
        if x.scale != 1.0 or x.shift != 0.0 or want != x.dtype:
            return x.ctx.to_device(np.ascontiguousarray(x.numpy(), dtype=want))
        return x
    ctx = ctx or default_context()
    x = np.asarray(x)
    if dtype is None:
        dtype = np.float32 if x.dtype == np.float32 else np.float64
    return ctx.to_device(np.ascontiguousarray(x, dtype=dtype))
