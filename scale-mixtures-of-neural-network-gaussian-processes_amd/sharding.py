"""Row-block sharding of the kernel build across the GPUs of one node.

Every K entry depends only on (x_i, x_j, q_i, q_j), so rows are independent units: rank r of P owns the
row block [r*R, min(n, (r+1)*R)) with R = ceil(n / P), X is replicated, and ONE all-gather of equal-sized
chunks (R * ld elements per rank, the last one zero-padded) assembles K.  Pure host logic: no device calls
here, so the same functions drive the RCCL path (bench.py) and the gloo CPU tests.
"""
from __future__ import annotations


def rows_per_rank(n: int, world: int) -> int:
    if n <= 0 or world <= 0:
        raise ValueError("n and world must be positive")
    return -(-n // world)


def row_shard(n: int, world: int, rank: int):
    """(begin, end) of the rows rank `rank` builds; end == begin for ranks past the data."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    r = rows_per_rank(n, world)
    b = min(n, rank * r)
    return b, min(n, b + r)


def gathered_rows(n: int, world: int) -> int:
    """Row count of the all-gather receive buffer (>= n; the tail rows are padding)."""
    return rows_per_rank(n, world) * world


def chunk_elems(n: int, world: int, ld: int) -> int:
    """Elements each rank contributes to the all-gather."""
    return rows_per_rank(n, world) * ld


# ------------------------------------------------------------------ paired lower-trapezoid layout
# K is symmetric and the Cholesky reads its lower triangle only, so a rank that builds full rows does
# twice the necessary work and the all-gather moves twice the necessary bytes.  Cutting the rows into 2P
# blocks of h rows and giving rank r the blocks r and 2P-1-r balances the lower triangle exactly: block b
# spans columns [0, (b+1)h), so the pair covers (r+1)h + (2P-r)h = (2P+1)h columns per row -- the same for
# every rank -- and, packed densely, every rank contributes exactly h*h*(2P+1) elements to ONE all-gather.

TILE = 128


def block_rows(n: int, world: int) -> int:
    """h: rows per block of the paired layout (a multiple of the 128-row tile)."""
    if n <= 0 or world <= 0:
        raise ValueError("n and world must be positive")
    h = -(-n // (2 * world))
    return -(-h // TILE) * TILE


def paired_blocks(world: int, rank: int):
    """The two row blocks (low, high) rank `rank` builds."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    return rank, 2 * world - 1 - rank


def block_owner(world: int, b: int) -> int:
    if not 0 <= b < 2 * world:
        raise ValueError("block out of range")
    return b if b < world else 2 * world - 1 - b


def block_range(n: int, world: int, b: int):
    """(row_begin, row_end) of block b, clipped to n (empty when the block lies past the data)."""
    h = block_rows(n, world)
    lo = min(n, b * h)
    return lo, min(n, lo + h)


def paired_chunk_elems(n: int, world: int) -> int:
    """Elements each rank contributes to the all-gather in the paired layout."""
    h = block_rows(n, world)
    return h * h * (2 * world + 1)


def block_offset(n: int, world: int, b: int):
    """(element offset inside the gathered staging buffer, leading dimension) of packed block b."""
    h = block_rows(n, world)
    owner = block_owner(world, b)
    off = owner * paired_chunk_elems(n, world) + (0 if b < world else h * (owner + 1) * h)
    return off, (b + 1) * h


def build_lower_sharded(ctx, dtype_code, itemsize, net, act, num_hiddens, w_std, b_std, last_w_std,
                        x_ptr, n, ldx, d, rank, world, stage_ptr, k_ptr, ldk, ntk_stage_ptr=None, ntk_ptr=None):
    """This rank's share of the symmetric kernel build + the exchange: one smn_kernel_mlp_shard launch into the
    rank's chunk of `stage_ptr` (world * paired_chunk_elems elements), one in-place smn_allgather and, when k_ptr
    is given, one smn_unpack_lower_blocks into the lower triangle of k_ptr [n,n] (pass None and hand the staging
    buffer to smn_lml_from_blocks to skip the separate copy of K).  With `ntk_stage_ptr` (a second staging buffer of
    the same size) the NTK is built, gathered and -- into ntk_ptr -- unpacked alongside (BASELINE config 5).
    All on the context's stream."""
    import ctypes as C
    chunk = paired_chunk_elems(n, world)
    h = block_rows(n, world)
    off = rank * chunk * itemsize
    mine = C.c_void_p(stage_ptr.value + off)
    mine_t = C.c_void_p(ntk_stage_ptr.value + off) if ntk_stage_ptr is not None else None
    ctx.call("smn_kernel_mlp_shard", dtype_code, net, act, num_hiddens, w_std, b_std, last_w_std,
             x_ptr, n, ldx, d, world, rank, h, 1 | (2 if mine_t is not None else 0), mine, mine_t)
    ctx.call("smn_allgather", dtype_code, mine, stage_ptr, chunk)           # in place
    if mine_t is not None:
        ctx.call("smn_allgather", dtype_code, mine_t, ntk_stage_ptr, chunk)
    if k_ptr is not None:
        ctx.call("smn_unpack_lower_blocks", dtype_code, stage_ptr, n, world, h, k_ptr, ldk)
    if mine_t is not None and ntk_ptr is not None:
        ctx.call("smn_unpack_lower_blocks", dtype_code, ntk_stage_ptr, n, world, h, ntk_ptr, ldk)
