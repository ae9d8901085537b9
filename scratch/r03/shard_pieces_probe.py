"""r03: what one rank's share of the PIPELINED sharded build costs, played on one GPU without a communicator: the launches of
sharding.part_tile_rows (one per piece) for every rank of P = 2, 4, 8, with the default number of pieces and with fewer, on the
main stream and (SMN_COMM_CUS_FORCE=1 in the environment) on the stream that leaves SMN_COMM_CUS CUs to RCCL."""
import ctypes as C, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smnngp import _lib as L, sharding as S
n, d, nl = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (16384, 3072, 4)
ctx = L.Context(0)
rng = np.random.default_rng(0)
x = ctx.to_device(rng.standard_normal((n, d)).astype(np.float32))
spec = (L.NET_MLP, 0, nl, 1.0, 1e-8, 1.0)
be = S.DeviceBackend(ctx)
def timed(fn, reps=5):
    fn(); ctx.synchronize()
    ctx.call("smn_timer_start")
    for _ in range(reps): fn()
    ms = C.c_double(); ctx.call("smn_timer_stop_ms", C.byref(ms))
    return ms.value / reps
out = {"n": n, "d": d, "layers": nl, "masked": os.environ.get("SMN_COMM_CUS_FORCE") == "1", "comm_cus": os.environ.get("SMN_COMM_CUS", "16")}
pipeline = os.environ.get("SMN_PROBE_PIPELINE") == "1"   # pieces launched as inside smn_shard_begin ... smn_lml_from_shards
out["pipeline_mode"] = pipeline
k = ctx.empty((n, n), np.float32)
out["single_gpu_lower_ms"] = timed(lambda: ctx.call("smn_kernel_mlp", L.F32, L.NET_MLP, 0, nl, 1.0, 1e-8, 1.0, x.ptr, n, d, None, 0, 0, d, L.GET_NNGP, L.FILL_LOWER, k.ptr, None, n))
if pipeline:
    be.begin(L.F32, n)
for P in (2, 4, 8):
    chunk, h = S.paired_chunk_elems(n, P), S.block_rows(n, P)
    mine = ctx.empty((chunk,), np.float32)
    res = {}
    gdef = S.default_parts(n, P)
    extra = {int(g) for g in os.environ.get("SMN_PROBE_G", "").split(",") if g}
    for G in sorted({gdef, max(1, gdef // 2), 1} | extra, reverse=True):
        per_rank = []
        for r in range(P):
            rows_list = S.part_tile_rows(n, P, r, G)
            def run():
                padded = False
                for rows in rows_list:
                    if rows[1] > rows[0] or rows[3] > rows[2]:
                        be.build_rows(L.F32, spec, x.ptr, n, d, d, P, r, h, rows, padded, mine.ptr)
                        padded = True
                if pipeline:
                    ctx.call("smn_shard_wait")      # the main stream joins the build streams
            per_rank.append(round(timed(run), 4))
        res["G%d" % G] = {"per_rank_ms": per_rank, "max_ms": max(per_rank), "speedup_vs_single": out["single_gpu_lower_ms"] / max(per_rank)}
    res["default_parts"] = gdef
    out["P%d" % P] = res
print(json.dumps(out))
