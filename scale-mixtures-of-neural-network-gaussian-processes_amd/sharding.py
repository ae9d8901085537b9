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
