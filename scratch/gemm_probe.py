"""Times the bare Gram kernel (build_kernel<NET_NONE>) through smn_gram: TF/s of the tile engine alone."""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, ".")
from smnngp import _lib as L
n = int(os.environ.get("PN", 16384)); d = int(os.environ.get("PD", 3072))
ctx = L.Context(0)
rng = np.random.default_rng(0)
x = ctx.to_device(rng.standard_normal((n, d)).astype(np.float32))
k0 = ctx.empty((n, n), np.float32)
for _ in range(2):
    ctx.call("smn_gram", L.F32, x.ptr, n, d, None, 0, 0, d, k0.ptr, n, None, None)
ctx.call("smn_profile_enable", 1)
reps = 5
for _ in range(reps):
    ctx.call("smn_gram", L.F32, x.ptr, n, d, None, 0, 0, d, k0.ptr, n, None, None)
ms, cnt = C.c_double(), C.c_int()
ctx.call("smn_profile_read", 1, C.byref(ms), C.byref(cnt))
t = n // 128
fl = t * (t + 1) // 2 * 128 * 128 * 2.0 * d
per = ms.value / cnt.value
print("%s map=%s lds=%s  N=%d d=%d  %.3f ms  %.1f TF (executed lower tiles, mirrored store)" % (
    os.path.basename(L.LIB_PATH), os.environ.get("SMN_XCD_MAP", "1"), os.environ.get("SMN_DEBUG_LDS", "-"), n, d, per, fl / per / 1e9))
