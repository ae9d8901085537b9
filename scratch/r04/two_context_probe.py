"""Do two contexts on one GPU overlap?  k threads, each with its own context, each running `reps` SPR.loss calls."""
import sys, time, threading, ctypes as C, os
sys.path.insert(0, '/root/repo')
import numpy as np
from smnngp import _lib as L

def run(n, d, nl, dtypes, reps):
    ctxs = [L.Context(0) for _ in dtypes]
    rng = np.random.default_rng(0)
    xh = rng.standard_normal((n, d)); yh = rng.standard_normal(n)
    data = [(c.to_device(xh.astype(dt)), c.to_device(yh.astype(dt))) for c, dt in zip(ctxs, dtypes)]
    def work(i, reps):
        c, (x, y) = ctxs[i], data[i]
        lp, info = C.c_double(), C.c_int()
        for _ in range(reps):
            c.call("smn_spr_loss", L.dtype_code(dtypes[i]), L.NET_MLP, L.ACT["relu"], nl, 1.0, 1e-8, 1.0, x.ptr, n, d, d, y.ptr,
                   1e-3 if dtypes[i] == np.float32 else 1e-6, 0.0, 1.0, C.byref(lp), None, None, C.byref(info))
    for i in range(len(ctxs)):
        work(i, 1)
    single = []
    for i in range(len(ctxs)):
        t0 = time.perf_counter(); work(i, reps); single.append((time.perf_counter() - t0) / reps * 1e3)
    th = [threading.Thread(target=work, args=(i, reps)) for i in range(len(ctxs))]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    both = (time.perf_counter() - t0) / reps * 1e3
    print("N=%d d=%d %s: alone %s ms, together %.3f ms per round (sum of alone %.3f)" % (n, d, [np.dtype(t).name for t in dtypes],
          ["%.3f" % s for s in single], both, sum(single)), flush=True)

print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
run(4096, 512, 3, [np.float32, np.float32], 20)
run(2048, 64, 3, [np.float32, np.float32, np.float32, np.float32], 20)
run(16384, 3072, 4, [np.float32, np.float64], 3)
