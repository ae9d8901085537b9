"""Per-rank cost of the cyclic column-first shard, every rank of P = 2 / 4 / 8 played on ONE GPU (C4 and C5 shapes):
the rank's single build launch, the scatter of all gathered pieces (P x the rank's bytes), and for reference the fused
one-GPU build."""
import ctypes as C, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smnngp import _lib as L, sharding as S

def cat(ctx, c):
    ms, cnt = C.c_double(), C.c_int()
    ctx.call("smn_profile_read", c, C.byref(ms), C.byref(cnt))
    return ms.value, cnt.value

def probe(n, d, nl, act, with_ntk):
    ctx = L.Context(0)
    rng = np.random.default_rng(0)
    x = ctx.to_device(rng.standard_normal((n, d)).astype(np.float32))
    y = ctx.to_device(rng.standard_normal(n).astype(np.float32))
    spec = (L.NET_MLP, L.ACT[act], nl, 1.0, 1e-8, 1.0)
    lp, info = C.c_double(), C.c_int()
    out = {"n": n, "d": d, "layers": nl, "act": act, "ntk": with_ntk}
    for _ in range(2):
        ctx.call("smn_spr_loss", L.F32, *spec, x.ptr, n, d, d, y.ptr, 1e-3, 0.0, 1.0, C.byref(lp), None, None, C.byref(info))
    ctx.call("smn_profile_enable", 2 << 1)
    for _ in range(3):
        ctx.call("smn_spr_loss", L.F32, *spec, x.ptr, n, d, d, y.ptr, 1e-3, 0.0, 1.0, C.byref(lp), None, None, C.byref(info))
    ms, cnt = cat(ctx, 1)
    out["one_gpu_fused_build_ms"] = ms / cnt
    ctx.call("smn_profile_enable", 0)
    for world in (2, 4, 8):
        cols = S.default_col_pieces(n, world)
        lay = S.col_layout(n, world, cols)
        mine = ctx.empty((lay["elems"],), np.float32)
        mine_t = ctx.empty((lay["elems"],), np.float32) if with_ntk else None
        be = S.DeviceBackend(ctx)
        per_rank = []
        for r in range(world):
            be.build_cols(L.F32, spec, x.ptr, n, d, d, world, r, cols, mine.ptr, mine_t.ptr if with_ntk else None)
            ctx.call("smn_profile_enable", 2 << 1)
            for _ in range(3):
                be.build_cols(L.F32, spec, x.ptr, n, d, d, world, r, cols, mine.ptr, mine_t.ptr if with_ntk else None)
            ms, cnt = cat(ctx, 1)
            ctx.call("smn_profile_enable", 0)
            per_rank.append(ms / cnt)
        # the scatter of every piece of a full staging buffer (what each rank does per step), main-stream filled
        stage = ctx.empty((world * lay["elems"],), np.float32)
        ca = S.cols_array(cols)
        ctx.call("smn_shard_begin", L.F32, n, 1e-3)
        ctx.call("smn_profile_enable", 2 << 6)
        for g in range(len(cols) - 1):
            ctx.call("smn_shard_scatter_cols", L.F32, stage.ptr, n, world, len(cols) - 1, ca, g, None, 0)
        ctx.call("smn_shard_wait")
        ctx.synchronize()
        ms, cnt = cat(ctx, 6)
        ctx.call("smn_profile_enable", 0)
        ctx.call("smn_shard_begin", L.F32, n, 1e-3)      # (drops the arrivals of the probe)
        out["P%d" % world] = {"pieces": len(cols) - 1, "rank_build_ms": [round(v, 4) for v in per_rank], "rank_build_ms_max": max(per_rank),
                              "tiles_per_rank": sum(t + 1 for t in S.rank_tile_rows(n, world, 0)),
                              "bytes_in_per_rank_MB": (world - 1) * lay["elems"] * 4 / 1e6, "chunk_MB": lay["elems"] * 4 / 1e6,
                              "piece0_bytes_in_per_rank_MB": (world - 1) * lay["count"][0] * 4 / 1e6,
                              "scatter_all_pieces_ms": ms, "scatter_piece0_ms_est": ms * lay["count"][0] / lay["elems"]}
        del mine, mine_t, stage
    return out

res = [probe(16384, 3072, 4, "relu", False), probe(32768, 1024, 6, "erf", True)]
print(json.dumps(res, indent=1))
