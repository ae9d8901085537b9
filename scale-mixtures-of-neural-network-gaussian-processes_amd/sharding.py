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
    _check_communicator(ctx, world)
    chunk = paired_chunk_elems(n, world)
    h = block_rows(n, world)
    off = rank * chunk * itemsize
    mine = C.c_void_p(stage_ptr.value + off)
    mine_t = C.c_void_p(ntk_stage_ptr.value + off) if ntk_stage_ptr is not None else None
    ctx.call("smn_kernel_mlp_shard", dtype_code, net, act, num_hiddens, w_std, b_std, last_w_std,
             x_ptr, n, ldx, d, world, rank, h, 1 | (2 if mine_t is not None else 0), mine, mine_t)
    ctx.call("smn_allgather", world, dtype_code, mine, stage_ptr, chunk)    # in place
    if mine_t is not None:
        ctx.call("smn_allgather", world, dtype_code, mine_t, ntk_stage_ptr, chunk)
    if k_ptr is not None:
        ctx.call("smn_unpack_lower_blocks", dtype_code, stage_ptr, n, world, h, k_ptr, ldk)
    if mine_t is not None and ntk_ptr is not None:
        ctx.call("smn_unpack_lower_blocks", dtype_code, ntk_stage_ptr, n, world, h, ntk_ptr, ldk)


def _check_communicator(ctx, world):
    """A world > 1 exchange on a context without a communicator would quietly degrade to a local copy and hand back a
    kernel whose other ranks' blocks are garbage: refuse."""
    import ctypes as C
    nr, rk = C.c_int(0), C.c_int(0)
    ctx.call("smn_comm_info", C.byref(nr), C.byref(rk))
    if nr.value != world:
        raise RuntimeError("sharded build over %d ranks, but the context's communicator has %d (smn_comm_init first)"
                           % (world, nr.value))


# ------------------------------------------------------------------ pipelined exchange (pieces of the chunks)
# The all-gather rides behind the build: a rank's chunk is cut into `parts` equal element ranges; as soon as the rank has
# built the tile rows that complete piece g, piece g of ALL ranks is gathered (and scattered into the factorisation
# workspace) on the communication stream while the tile rows of piece g+1 are being built.  What remains exposed is the
# last piece.  Host logic only; the device steps go through a backend object so that the gloo CPU tests drive exactly
# this control flow.

def default_parts(n: int, world: int) -> int:
    """Pieces per chunk.  A piece's tile rows are one build launch, and a launch of fewer tiles than the chip has CUs
    leaves CUs idle for a whole tile time (0.2-0.4 ms at d = 3072): at least ~256 tiles per piece, at most 8 pieces (16 on
    two ranks, where the exchange over the single link takes as long as the build and what is exposed is one piece of it),
    a power of two that cuts the chunk into multiples of 4 elements."""
    t = -(-n // TILE)
    tiles_per_rank = t * (t + 1) // 2 // world
    p = 16 if world == 2 else 8
    while p > 1 and tiles_per_rank // p < 256:
        p //= 2
    chunk = paired_chunk_elems(n, world)
    while p > 1 and (chunk % p or (chunk // p) % 4):
        p //= 2
    return p


def part_tile_rows(n: int, world: int, rank: int, parts: int):
    """For each piece g: (lo_t0, lo_t1, hi_t0, hi_t1) = the 128-row tile rows of the rank's low / high block that must be
    built before piece g can be gathered and were not built for an earlier piece.  Tile rows are built whole and in
    packed order (low block top to bottom, then the high block), so piece g needs every tile row that STARTS before the
    piece ends."""
    h = block_rows(n, world)
    chunk = paired_chunk_elems(n, world)
    if parts <= 0 or chunk % parts or (chunk // parts) % 4:
        raise ValueError("parts=%d must divide the chunk (%d elements) into multiples of 4" % (parts, chunk))
    piece = chunk // parts
    tpb = h // TILE
    lo, hi = paired_blocks(world, rank)
    ld_lo, ld_hi = (lo + 1) * h, (hi + 1) * h
    starts = [t * TILE * ld_lo for t in range(tpb)] + [h * ld_lo + t * TILE * ld_hi for t in range(tpb)]
    out, built = [], 0
    for g in range(parts):
        end = (g + 1) * piece
        upto = built
        while upto < 2 * tpb and starts[upto] < end:
            upto += 1
        a, b = built, upto                      # packed tile rows [a, b)
        out.append((min(a, tpb), min(b, tpb), max(a, tpb) - tpb, max(b, tpb) - tpb))
        built = upto
    assert built == 2 * tpb
    return out


class DeviceBackend:
    """The product path: every step is one libsmnngp call on the rank's context."""

    def __init__(self, ctx):
        self.ctx = ctx

    def comm_size(self):
        import ctypes as C
        nr, rk = C.c_int(0), C.c_int(0)
        self.ctx.call("smn_comm_info", C.byref(nr), C.byref(rk))
        return nr.value

    def begin(self, dtype_code, n):
        self.ctx.call("smn_shard_begin", dtype_code, n)

    def build_rows(self, dtype_code, spec, x_ptr, n, ldx, d, world, rank, h, rows, reuse, mine_ptr, ntk_mine_ptr=None):
        net, act, num_hiddens, w_std, b_std, last_w_std = spec
        self.ctx.call("smn_kernel_mlp_shard_rows", dtype_code, net, act, num_hiddens, w_std, b_std, last_w_std,
                      x_ptr, n, ldx, d, world, rank, h, rows[0], rows[1], rows[2], rows[3], 1 if reuse else 0,
                      1 | (2 if ntk_mine_ptr is not None else 0), mine_ptr, ntk_mine_ptr)

    def exchange_part(self, dtype_code, mine_ptr, stage_ptr, n, world, h, parts, part, ntk=None):
        """Piece `part` of the NNGP chunks into the factorisation workspace; with ntk = (mine, stage, out, ld) also the
        same piece of the NTK chunks into the caller's matrix `out`.  Both on the communication stream."""
        self.ctx.call("smn_shard_exchange_part", dtype_code, mine_ptr, stage_ptr, n, world, h, parts, part)
        if ntk is not None:
            self.ctx.call("smn_shard_exchange_part_to", dtype_code, ntk[0], ntk[1], n, world, h, parts, part, ntk[2], ntk[3])

    def lml(self, dtype_code, n, y_ptr, eps_abs, df, scale):
        import ctypes as C
        lp, quad, logdet, info = C.c_double(), C.c_double(), C.c_double(), C.c_int()
        self.ctx.call("smn_lml_from_shards", dtype_code, n, y_ptr, eps_abs, df, scale, C.byref(lp), C.byref(quad),
                      C.byref(logdet), C.byref(info))
        return lp.value, quad.value, logdet.value, info.value


def lml_sharded_pipelined(backend, dtype_code, spec, x_ptr, n, ldx, d, y_ptr, rank, world, mine_ptr, stage_ptr,
                          eps_abs, df=0.0, scale=1.0, parts=None, ntk=None):
    """One SPR.loss evaluation with the kernel build sharded over `world` ranks and the exchange pipelined behind it
    (`spec` = (net, act, num_hiddens, w_std, b_std, last_w_std); mine: the rank's chunk, paired_chunk_elems elements;
    stage: world * that).  Returns (logpdf, quad, logdet, info); every rank computes the same values.
    ntk = (mine, stage, out, ld): the build also produces the NTK (one joint launch per piece: BASELINE config 5's "erf NNGP
    + NTK"), whose pieces ride the same pipeline and are assembled in the caller's matrix `out` (lower triangle by
    128-column tiles) by the time the likelihood has been read."""
    if backend.comm_size() != world:
        raise RuntimeError("sharded build over %d ranks, but the communicator has %d (smn_comm_init first)"
                           % (world, backend.comm_size()))
    parts = parts or default_parts(n, world)
    h = block_rows(n, world)
    backend.begin(dtype_code, n)
    padded = False
    for g, rows in enumerate(part_tile_rows(n, world, rank, parts)):
        if rows[1] > rows[0] or rows[3] > rows[2]:
            if ntk is None:
                backend.build_rows(dtype_code, spec, x_ptr, n, ldx, d, world, rank, h, rows, padded, mine_ptr)
            else:
                backend.build_rows(dtype_code, spec, x_ptr, n, ldx, d, world, rank, h, rows, padded, mine_ptr, ntk[0])
            padded = True
        if ntk is None:
            backend.exchange_part(dtype_code, mine_ptr, stage_ptr, n, world, h, parts, g)
        else:
            backend.exchange_part(dtype_code, mine_ptr, stage_ptr, n, world, h, parts, g, ntk)
    return backend.lml(dtype_code, n, y_ptr, eps_abs, df, scale)
