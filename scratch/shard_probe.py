"""Per-rank build time of the paired layout, measured on ONE GPU by running each rank's launch in turn
(no communicator): what phases_ms.build will be on every rank of a P-GPU run."""
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smnngp import _lib as L, sharding as S
n, d, nl = 16384, 3072, 4
ctx = L.Context(0)
rng = np.random.default_rng(0)
x = ctx.to_device(rng.standard_normal((n, d)).astype(np.float32))
out = {}
def timed(fn, reps=5):
    fn(); ctx.synchronize()
    ctx.call("smn_timer_start")
    for _ in range(reps): fn()
    ms = C.c_double(); ctx.call("smn_timer_stop_ms", C.byref(ms))
    return ms.value / reps
k = ctx.empty((n, n), np.float32)
full = timed(lambda: ctx.call("smn_kernel_mlp", L.F32, L.NET_MLP, 0, nl, 1.0, 1e-8, 1.0, x.ptr, n, d, None, 0, 0, d, L.GET_NNGP, L.FILL_LOWER, k.ptr, None, n))
out["single_gpu_lower_ms"] = full
for P in (2, 4, 8):
    chunk, h = S.paired_chunk_elems(n, P), S.block_rows(n, P)
    stage = ctx.empty((chunk,), np.float32)
    ts = [timed(lambda r=r: ctx.call("smn_kernel_mlp_shard", L.F32, L.NET_MLP, 0, nl, 1.0, 1e-8, 1.0, x.ptr, n, d, d, P, r, h, L.GET_NNGP, stage.ptr, None)) for r in range(P)]
    full_stage = ctx.empty((P * chunk,), np.float32)
    un = timed(lambda: ctx.call("smn_unpack_lower_blocks", L.F32, full_stage.ptr, n, P, h, k.ptr, n))
    out["P%d" % P] = {"per_rank_ms": [round(t, 3) for t in ts], "max_ms": max(ts), "speedup_vs_single": full / max(ts), "unpack_ms": un,
                      "chunk_MB": chunk * 4 / 1e6}
print(json.dumps(out))
