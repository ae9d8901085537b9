import sys, json, time, ctypes as C
sys.path.insert(0, '/root/repo')
import numpy as np
import bench
from smnngp import _lib as L
ctx = L.default_context()
n, d, nl = 16384, 3072, 4
rng = np.random.default_rng(0)
x = ctx.to_device(rng.standard_normal((n, d)).astype(np.float32)); y = ctx.to_device(rng.standard_normal(n).astype(np.float32))
quad, logdet, info = C.c_double(), C.c_double(), C.c_int()
terms = (C.c_double * 4)()
def call():
    ctx.call("smn_spr_loss_grad", L.F32, L.NET_MLP, L.ACT["relu"], nl, 1.0, 0.3, 1.0, x.ptr, n, d, d, y.ptr, 1e-2, 4.0, 1.0,
             C.byref(quad), C.byref(logdet), C.byref(info), terms)
call(); call()
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(3): call()
ctx.synchronize()
print("ms per call", (time.perf_counter() - t0) / 3 * 1e3)
ctx.call("smn_profile_enable", 1)
call()
print(json.dumps(bench.read_profile(ctx, 1), indent=0))
fl = C.c_double()
ctx.call("smn_profile_flops", 5, C.byref(fl)); print("trail flops", fl.value)
ctx.call("smn_profile_enable", 0)
