"""Sharding of the kernel build across the GPUs of one node.

Every K entry depends only on (x_i, x_j, q_i, q_j), so rows are independent units and X is replicated.  Three layouts, from
plain to what `bench.py --gpus P` runs:

* full row blocks (`row_shard`): rank r owns rows [r*R, (r+1)*R) x all columns; one all-gather of R*ld elements per rank;
* paired lower blocks (`paired_blocks`): 2P row blocks, rank r owns blocks r and 2P-1-r, lower trapezoids only -- half the
  flops and bytes, exactly balanced; one monolithic all-gather, then the factorisation;
* cyclic column-first (`col_layout`, `lml_sharded_cols`): 128-row tile rows dealt in boustrophedon order, lower tiles only;
  every rank holds an equal share of EVERY column range, so the exchange goes out column range by column range and the
  single-GPU factorisation starts on the first super-panel's columns while the rest is still on the links.

Pure host logic: no device calls here, so the same functions drive the RCCL path (bench.py) and the gloo CPU tests.
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


# ------------------------------------------------------------------ cyclic column-first layout
# The factorisation is right-looking: its first super-panel needs only the kernel's first 1024 COLUMNS (all rows), the
# update that follows the next 1024, and so on.  Row-block shards deliver whole rows, i.e. every column range last.  Here the
# 128-row tile rows are dealt in boustrophedon order with period 2P -- group j = tile rows [jP, (j+1)P), rank r owns
# t_j(r) = jP + (r if j is even else P-1-r) -- so that (i) tile row t holds t+1 lower tiles and every pair of groups gives
# every rank the same number of them: the build is balanced; (ii) every aligned group of P tile rows holds exactly one tile
# row per rank: a tile-column range [c0, c1) is ceil(T/P) - floor(c0/P) strips of 128 x (c1-c0)*128 elements on EVERY rank
# (its tile rows from group floor(c0/P) down), one equal-count all-gather.  (Tile rows that start inside the range carry
# their above-diagonal tiles as padding: 5.4 % of the bytes at N = 16384 with 1024-column ranges.)  Mirrors
# csrc/internal.hpp ColPieces.

MAX_COL_PIECES = 32


def tile_rows(n: int) -> int:
    return -(-n // TILE)


def tile_row_owner(world: int, t: int) -> int:
    u = t % (2 * world)
    return u if u < world else 2 * world - 1 - u


def rank_tile_row(world: int, rank: int, j: int) -> int:
    """The tile row rank `rank` owns in group j (may lie past the kernel: the caller compares with tile_rows(n))."""
    return j * world + (rank if j % 2 == 0 else world - 1 - rank)


def rank_tile_rows(n: int, world: int, rank: int):
    t_all = tile_rows(n)
    out = []
    for j in range(-(-t_all // world)):
        t = rank_tile_row(world, rank, j)
        if t < t_all:
            out.append(t)
    return out


def default_col_pieces(n: int, world: int, first_cols: int = 1024, head_cols: int = 256, max_pieces: int = 20):
    """Tile-column boundaries [0, c1, ..., T] of the exchange.  The first super-panel of the factorisation (1024 columns)
    goes out in `head_cols` = 256-column ranges: its first two sub-panels wait for 256 columns only -- 15 MB per rank at
    N = 16384 on 8 GPUs instead of 59 MB -- and the update behind them for the rest of the super-panel.  Everything beyond
    goes out in ranges of one super-panel (1024 columns) each, widened so that there are at most `max_pieces` in all
    (N = 32768: 2048 columns); a narrow range carries little above-diagonal padding ((w-1)/(T+1) of the bytes for w tile
    columns: 5.4 % at N = 16384).  Any ascending boundaries are legal (a range that does not start on a multiple of `world`
    carries at most one padding strip per rank)."""
    if n <= 0 or world <= 0:
        raise ValueError("n and world must be positive")
    t_all = tile_rows(n)
    first = max(1, first_cols // TILE)
    head = max(1, head_cols // TILE)
    cols = [0]
    if world > 1 and t_all > first:
        cols = list(range(0, first, head))                       # [0, 2, 4, 6] at the defaults
        start = first
    else:
        start = 0
    left = max(1, min(max_pieces, MAX_COL_PIECES) - (len(cols) if start else 0))
    w = max(first, -(-(t_all - start) // left))
    cols = (cols if start else []) + list(range(start, t_all, w)) + [t_all]
    return cols


def col_layout(n: int, world: int, cols):
    """Per piece g: slots (strips per rank), width (elements per strip row), count (elements per rank), off (element offset
    of the piece in a rank's chunk); plus `elems` = elements per rank.  The staging buffer holds piece g of all ranks at
    world * off[g], rank-major."""
    t_all = tile_rows(n)
    cols = [int(c) for c in cols]
    if len(cols) < 2 or len(cols) - 1 > MAX_COL_PIECES or cols[0] != 0 or cols[-1] != t_all:
        raise ValueError("column pieces must span the tile columns [0, %d) in at most %d pieces" % (t_all, MAX_COL_PIECES))
    slots, width, count, off = [], [], [], [0]
    for g in range(len(cols) - 1):
        if cols[g + 1] <= cols[g]:
            raise ValueError("piece %d = [%d, %d): boundaries must ascend" % (g, cols[g], cols[g + 1]))
        slots.append(-(-t_all // world) - cols[g] // world)
        width.append((cols[g + 1] - cols[g]) * TILE)
        count.append(slots[-1] * TILE * width[-1])
        off.append(off[-1] + count[-1])
    return {"cols": cols, "slots": slots, "width": width, "count": count, "off": off[:-1], "elems": off[-1]}


def cols_array(cols):
    """The boundaries as the int64 array the C-ABI takes."""
    import ctypes as C
    return (C.c_int64 * len(cols))(*[int(c) for c in cols])


class DeviceBackend:
    """The product path: every step is one libsmnngp call on the rank's context."""

    def __init__(self, ctx):
        self.ctx = ctx

    def comm_size(self):
        import ctypes as C
        nr, rk = C.c_int(0), C.c_int(0)
        self.ctx.call("smn_comm_info", C.byref(nr), C.byref(rk))
        return nr.value

    def begin(self, dtype_code, n, eps_abs):
        self.ctx.call("smn_shard_begin", dtype_code, n, eps_abs)

    def build_cols(self, dtype_code, spec, x_ptr, n, ldx, d, world, rank, cols, mine_ptr, ntk_mine_ptr=None):
        net, act, num_hiddens, w_std, b_std, last_w_std = spec
        self.ctx.call("smn_kernel_mlp_shard_cols", dtype_code, net, act, num_hiddens, w_std, b_std, last_w_std,
                      x_ptr, n, ldx, d, world, rank, len(cols) - 1, cols_array(cols),
                      1 | (2 if ntk_mine_ptr is not None else 0), mine_ptr, ntk_mine_ptr)

    def exchange_cols(self, dtype_code, mine_ptr, stage_ptr, n, world, cols, piece, ntk=None):
        """Piece `piece` of the NNGP chunks into the factorisation workspace; with ntk = (mine, stage, out, ld) also the
        same piece of the NTK chunks into the caller's matrix `out`."""
        ca = cols_array(cols)
        self.ctx.call("smn_shard_exchange_cols", dtype_code, mine_ptr, stage_ptr, n, world, len(cols) - 1, ca, piece)
        if ntk is not None:
            self.ctx.call("smn_shard_exchange_cols_to", dtype_code, ntk[0], ntk[1], n, world, len(cols) - 1, ca, piece,
                          ntk[2], ntk[3])

    def lml(self, dtype_code, n, y_ptr, df, scale):
        import ctypes as C
        lp, quad, logdet, info = C.c_double(), C.c_double(), C.c_double(), C.c_int()
        self.ctx.call("smn_lml_from_shards", dtype_code, n, y_ptr, df, scale, C.byref(lp), C.byref(quad),
                      C.byref(logdet), C.byref(info))
        return lp.value, quad.value, logdet.value, info.value


def lml_sharded_cols(backend, dtype_code, spec, x_ptr, n, ldx, d, y_ptr, rank, world, mine_ptr, stage_ptr,
                     eps_abs, df=0.0, scale=1.0, cols=None, ntk=None, progress=None):
    """One SPR.loss evaluation with the kernel build sharded over `world` ranks in the cyclic column-first layout
    (`spec` = (net, act, num_hiddens, w_std, b_std, last_w_std); mine: the rank's chunk, col_layout(...)["elems"] elements;
    stage: world * that).  The rank builds its whole share in one launch; the pieces then go out column range by column
    range (all-gather + scatter on the context's side streams) and the factorisation, issued right behind them, waits for
    each piece only when it reaches its columns.  Returns (logpdf, quad, logdet, info); every rank computes the same values.
    ntk = (mine, stage, out, ld): the build also produces the NTK (one joint launch: BASELINE config 5's "erf NNGP + NTK"),
    whose pieces ride the same streams and are assembled in the caller's matrix `out` (lower triangle by 128-column tiles)
    by the time the likelihood has been read.  progress(phase): called with a short phase name before every step (the
    watchdog of bench.py)."""
    if backend.comm_size() != world:
        raise RuntimeError("sharded build over %d ranks, but the communicator has %d (smn_comm_init first)"
                           % (world, backend.comm_size()))
    cols = list(cols) if cols is not None else default_col_pieces(n, world)
    say = progress or (lambda phase: None)
    say("begin")
    backend.begin(dtype_code, n, eps_abs)
    say("build")
    backend.build_cols(dtype_code, spec, x_ptr, n, ldx, d, world, rank, cols, mine_ptr, None if ntk is None else ntk[0])
    for g in range(len(cols) - 1):
        say("gather %d/%d" % (g, len(cols) - 1))
        backend.exchange_cols(dtype_code, mine_ptr, stage_ptr, n, world, cols, g, ntk)
    say("factor")
    return backend.lml(dtype_code, n, y_ptr, df, scale)
