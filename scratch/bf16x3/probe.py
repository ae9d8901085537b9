"""R&D probe: the bf16 x 3 split GEMM (six products) against the exact-f32 MFMA Gram and an fp64 reference."""
import ctypes as C, os, subprocess, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from smnngp import _lib as L
so = os.path.join(HERE, "libbf16x3.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-std=c++17",
                           os.path.join(HERE, "gemm_bf16x3.hip"), "-o", so])
lib = C.CDLL(so)
lib.bf16x3_gemm.restype = C.c_double
lib.bf16x3_gemm.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
sop = os.path.join(HERE, "libbf16x3p.so")
if not os.path.exists(sop):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-std=c++17",
                           os.path.join(HERE, "gemm_bf16x3_presplit.hip"), "-o", sop])
libp = C.CDLL(sop)
libp.bf16x3_gemm_presplit.restype = C.c_double
libp.bf16x3_gemm_presplit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
ctx = L.Context(0)
rng = np.random.default_rng(0)
for n, d in ((2048, 512), (8192, 3072), (16384, 3072)):
    xh = rng.standard_normal((n, d)).astype(np.float32)
    x = ctx.to_device(xh)
    c = ctx.empty((n, n), np.float32)
    ctx.synchronize()
    ms = lib.bf16x3_gemm(x.ptr, x.ptr, c.ptr, n, n, d, 3)
    assert ms > 0, ms
    planes = ctx.empty((3 * n * d // 2,), np.float32)                   # 3 planes of bf16
    c2 = ctx.empty((n, n), np.float32)
    sm = C.c_double()
    msp = libp.bf16x3_gemm_presplit(x.ptr, planes.ptr, c2.ptr, n, d, 3, C.byref(sm))
    assert msp > 0, msp
    g = ctx.empty((n, n), np.float32)
    ctx.call("smn_timer_start")
    for _ in range(3):
        ctx.call("smn_gram", L.F32, x.ptr, n, d, x.ptr, n, d, d, g.ptr, n, None, None)     # cross form: every tile computed
    t = C.c_double(); ctx.call("smn_timer_stop_ms", C.byref(t)); ms32 = t.value / 3
    rows = np.sort(rng.choice(n, 64, replace=False))
    got = np.stack([np.frombuffer(memoryview(bytearray(n * 4)), np.float32) for _ in rows])
    ref32 = np.empty_like(got)
    for i, r in enumerate(rows):
        ctx.call("smn_memcpy_d2h", got[i].ctypes.data_as(C.c_void_p), C.c_void_p(c.ptr.value + int(r) * n * 4), n * 4)
        ctx.call("smn_memcpy_d2h", ref32[i].ctypes.data_as(C.c_void_p), C.c_void_p(g.ptr.value + int(r) * n * 4), n * 4)
    x64 = xh.astype(np.float64)
    ref = x64[rows] @ x64.T / d
    scale = np.abs(ref).max()
    fl = 2.0 * n * n * d
    got2 = np.empty_like(got)
    for i, r in enumerate(rows):
        ctx.call("smn_memcpy_d2h", got2[i].ctypes.data_as(C.c_void_p), C.c_void_p(c2.ptr.value + int(r) * n * 4), n * 4)
    print("N=%d d=%d: bf16x3 %.3f ms = %.1f TFLOP/s | f32 MFMA %.3f ms = %.1f TFLOP/s | speed-up %.2fx" % (n, d, ms, fl / ms / 1e9, ms32, fl / ms32 / 1e9, ms32 / ms))
    print("    pre-split planes: GEMM %.3f ms = %.1f TFLOP/s (%.2fx), one-off split %.3f ms; identical to the on-the-fly split: %s" % (
        msp, fl / msp / 1e9, ms32 / msp, sm.value, bool(np.array_equal(got, got2))))
    print("    max |err| / max|C| vs fp64:  bf16x3 %.3e   f32 MFMA %.3e ;  rms  bf16x3 %.3e   f32 MFMA %.3e" % (
        np.abs(got - ref).max() / scale, np.abs(ref32 - ref).max() / scale,
        np.sqrt(((got - ref) ** 2).mean()) / scale, np.sqrt(((ref32 - ref) ** 2).mean()) / scale), flush=True)
