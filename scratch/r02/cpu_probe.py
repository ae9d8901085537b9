"""What the GPU box's host really gives the CPU baseline: affinity, cgroup quota, LAPACK spotrf time by thread count."""
import os, sys, time
import numpy as np, scipy.linalg as sla
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import host_parallel as HP
print("affinity", len(os.sched_getaffinity(0)), "cpu_count", os.cpu_count(), "host_cores()", HP.host_cores())
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
    try: print(p, open(p).read().strip())
    except OSError as e: print(p, "-", e)
n = 8192
rng = np.random.default_rng(0)
g = rng.standard_normal((n, 256)).astype(np.float32)
for thr in (64, 32, 16, 8):
    a = g @ g.T / 256 + np.eye(n, dtype=np.float32)
    with HP.blas_threads(thr):
        t0 = time.perf_counter(); sla.cholesky(a, lower=True, overwrite_a=True, check_finite=False); dt = time.perf_counter() - t0
    print("spotrf N=%d with %d BLAS threads: %.2f s = %.0f GFLOP/s" % (n, thr, dt, n ** 3 / 3 / dt / 1e9), flush=True)
