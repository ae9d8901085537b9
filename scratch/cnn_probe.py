"""Times the conv-NNGP kernel (C3 shape: 32x32x3 images, 4 layers, ReLU) at a few N, fp32 and fp64."""
import ctypes as C, os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smnngp import _lib as L
ctx = L.Context(0)
rng = np.random.default_rng(0)
for dt in (np.float32, np.float64):
    for n in (256, 1024, 2048):
        x = ctx.to_device(rng.uniform(0, 1, (n, 32, 32, 3)).astype(dt))
        k = ctx.empty((n, n), dt)
        def run():
            ctx.call("smn_kernel_cnn", L.dtype_code(dt), L.ACT["relu"], 4, 1.0, 0.1, 1.0, x.ptr, n, None, 0, 32, 32, 3, L.FILL_FULL, k.ptr, n)
        run(); ctx.synchronize()
        t0 = time.perf_counter(); run(); ctx.synchronize(); dtm = time.perf_counter() - t0
        pairs = n * (n + 1) // 2
        print("cnn %s N=%d: %.2f ms, %.3g pair-pixel-layers/s  -> N=10000 est %.1f s" % (np.dtype(dt).name, n, dtm * 1e3, pairs * 1024 * 4 / dtm, dtm * (10000 * 10001 / 2) / pairs))
