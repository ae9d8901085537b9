"""Test helper: the ranks of a multi-GPU build played one after the other on ONE GPU (the GPU boxes have one device)."""
import ctypes as C

import numpy as np


def play_ranks(L, ctx, spec, x, n, d, world, cols, dtype=np.float32, with_ntk=True):
    """`world` ranks played on one GPU: each builds its whole share with ONE smn_kernel_mlp_shard_cols launch into a
    NaN-poisoned chunk, and its pieces are copied to where the all-gather of piece g would leave them
    (stage[world * off[g] + rank * count[g]]).  Returns (stage, stage_ntk)."""
    from smnngp import sharding as S
    lay = S.col_layout(n, world, cols)
    isz = np.dtype(dtype).itemsize
    code = L.dtype_code(dtype)
    stage = ctx.to_device(np.full(world * lay["elems"], np.nan, dtype))
    stage_t = ctx.to_device(np.full(world * lay["elems"], np.nan, dtype)) if with_ntk else None
    be = S.DeviceBackend(ctx)
    for r in range(world):
        mine = ctx.to_device(np.full(lay["elems"], np.nan, dtype))
        mine_t = ctx.to_device(np.full(lay["elems"], np.nan, dtype)) if with_ntk else None
        be.build_cols(code, spec, x.ptr, n, d, d, world, r, cols, mine.ptr, mine_t.ptr if with_ntk else None)
        for g in range(len(cols) - 1):
            for src, dst in ((mine, stage), (mine_t, stage_t)):
                if src is None:
                    continue
                ctx.call("smn_memcpy_d2d", C.c_void_p(dst.ptr.value + isz * (world * lay["off"][g] + r * lay["count"][g])),
                         C.c_void_p(src.ptr.value + isz * lay["off"][g]), isz * lay["count"][g])
        ctx.synchronize()
        del mine, mine_t
    return stage, stage_t
