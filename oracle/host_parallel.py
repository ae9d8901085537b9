"""The CPU oracle at BASELINE.json's full sizes: the same oracle functions (oracle/nngp_oracle.py), driven over
row blocks on a host thread pool so that N = 16384 finishes in tens of seconds on the GPU box's host cores.

THIS FILE IS TEST INFRASTRUCTURE (like the rest of oracle/): only tests/ and the `cpu_baseline` leg of bench.py
import it.  Nothing is restated here: the Gram is `O.input_gram` (BLAS), every layer is `O._dense` + the
activation map of `O.get_act` applied to a block of rows (NumPy ufuncs release the GIL), the factorisation is
LAPACK through scipy.  Follows experiments/nt_kernels.py:21-31 (layer order), spax/models.py:93-98 (absolute
jitter, -logpdf / N) and spax/likelihoods.py:25-28 (MVN log-pdf) exactly as the small-size oracle does.
"""
from __future__ import annotations

import os
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import scipy.linalg as sla

from . import nngp_oracle as O


def host_cores(limit=64):
    """CPUs this process may really use: the affinity mask, cut by the cgroup CPU quota where one is set (a container that
    sees 64 cores through its affinity mask but holds a 16-CPU quota runs 64 BLAS threads slower than 16)."""
    try:
        c = len(os.sched_getaffinity(0))
    except AttributeError:
        c = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    c = min(c, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                quota = int(parts[0])
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = int(f.read().split()[0])
                if quota > 0:
                    c = min(c, max(1, quota // period))
            break
        except (OSError, ValueError, IndexError, ZeroDivisionError):
            continue
    return max(1, min(c, limit))


def blas_threads(n):
    """Context manager: BLAS / LAPACK thread pools limited to n threads (no-op without threadpoolctl)."""
    try:
        from threadpoolctl import threadpool_limits
        return threadpool_limits(limits=int(n))
    except Exception:
        import contextlib
        return contextlib.nullcontext()


def mlp_kernel_rows_threaded(x, num_hiddens, act, w_std, b_std, last_w_std, dtype=np.float64, block=256, cores=None):
    """Symmetric O.mlp_kernel(x, None, ..., "nngp") for large N: returns (K [N,N] in `dtype`, seconds).
    The diagonal is forced to the variance recursion exactly as O.mlp_kernel's symmetric case does."""
    cores = cores or host_cores()
    x = np.ascontiguousarray(x, dtype=dtype)
    n = x.shape[0]
    amap = O.get_act(act)
    t0 = time.perf_counter()
    with blas_threads(cores):
        k, q, _ = O.input_gram(x, None)
    q = q.astype(dtype)

    def rows(r0):
        r1 = min(n, r0 + block)
        kb, q1, q2 = k[r0:r1], q[r0:r1], q
        for _ in range(num_hiddens):
            kb, q1, q2, _t = O._dense(kb, q1, q2, None, w_std, b_std)
            kb, q1, q2, _t = amap(kb, q1, q2, None)
        kb, q1, q2, _t = O._dense(kb, q1, q2, None, last_w_std, 0.0)
        k[r0:r1] = kb
        return q1

    with ThreadPoolExecutor(max_workers=cores) as ex:
        qd = np.concatenate(list(ex.map(rows, range(0, n, block))))
    k[np.diag_indices(n)] = qd
    return k, time.perf_counter() - t0


def gaussian_lml(k, y, eps, cores=None):
    """(logpdf, quad, logdet, seconds) of N(y; 0, K + eps I): spax/models.py:96-97, spax/likelihoods.py:25-28.
    K is overwritten by its factor."""
    n = k.shape[0]
    t0 = time.perf_counter()
    k[np.diag_indices(n)] += k.dtype.type(eps)
    with blas_threads(cores or host_cores()):
        l = sla.cholesky(k, lower=True, overwrite_a=True, check_finite=False)
        z = sla.solve_triangular(l, np.asarray(y, dtype=k.dtype), lower=True, check_finite=False)
    quad = float(np.dot(z.astype(np.float64), z.astype(np.float64)))
    logdet = 2.0 * float(np.log(np.diag(l).astype(np.float64)).sum())
    lp = -0.5 * quad - 0.5 * n * np.log(2.0 * np.pi) - 0.5 * logdet
    return lp, quad, logdet, time.perf_counter() - t0
