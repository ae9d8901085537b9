"""One conv-NNGP kernel call (C3-shaped images, N from argv) for counter passes."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smnngp import _lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
dt = np.float64 if (len(sys.argv) < 3 or sys.argv[2] == "f64") else np.float32
ctx = L.Context(0)
rng = np.random.default_rng(0)
x = ctx.to_device(rng.uniform(0, 1, (n, 32, 32, 3)).astype(dt))
k = ctx.empty((n, n), dt)
for _ in range(2):
    ctx.call("smn_kernel_cnn", L.dtype_code(dt), L.ACT["relu"], 4, 1.0, 0.1, 1.0, x.ptr, n, None, 0, 32, 32, 3, L.FILL_FULL, k.ptr, n)
ctx.synchronize()
