"""N > 1 path on CPU: the row-shard / all-gather host logic of bench.py with world_size-2 gloo processes.
Each rank builds its rows with the oracle (stand-in for smn_kernel_mlp_rows), the ranks all-gather, and the
assembled kernel and its LML must equal the single-process result."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_row_shard_partitions_every_row_once():
    from smnngp import sharding as S
    for n, w in [(16384, 8), (301, 2), (7, 8), (1, 1), (1000, 3)]:
        seen = np.zeros(n, dtype=int)
        for r in range(w):
            b, e = S.row_shard(n, w, r)
            assert 0 <= b <= e <= n and e - b <= S.rows_per_rank(n, w)
            seen[b:e] += 1
        assert (seen == 1).all()
        assert S.gathered_rows(n, w) >= n and S.chunk_elems(n, w, n) == S.rows_per_rank(n, w) * n
    with pytest.raises(ValueError):
        S.row_shard(10, 2, 2)


def _worker(rank, world, port, n, d, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from oracle import nngp_oracle as O
    from smnngp import sharding as S
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(0)                     # replicated X, like bench.py
        x = rng.standard_normal((n, d)); y = rng.standard_normal(n)
        b, e = S.row_shard(n, world, rank)
        rows = O.mlp_kernel(x[b:e], x, 2, "relu", 1.2, 0.3, 1.0)          # this rank's row block
        if e > b:                                                           # exact diagonal, like the HIP shard
            rows[np.arange(e - b), np.arange(b, e)] = O.diag_recursion((x[b:e] ** 2).sum(1) / d, 2, "relu", 1.2, 0.3, 1.0)
        r = S.rows_per_rank(n, world)
        send = torch.zeros(r, n, dtype=torch.float64)
        send[: e - b] = torch.from_numpy(rows)
        recv = torch.zeros(S.gathered_rows(n, world), n, dtype=torch.float64)
        dist.all_gather_into_tensor(recv, send)
        k = recv[:n].numpy()
        lml = O.mvn_logpdf(y, k + 1e-3 * np.eye(n))
        t = torch.tensor([lml], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)           # bench.py's max-over-ranks reduction
        if rank == 0:
            q.put((k, lml, float(t.item())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [64, 37])
def test_world_size_2_gloo_build_gather_lml(n):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    from oracle import nngp_oracle as O
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    d = 5
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n, d, q)) for r in range(2)]
    for p in procs:
        p.start()
    k, lml, lml_max = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(0)
    x = rng.standard_normal((n, d)); y = rng.standard_normal(n)
    ref = O.mlp_kernel(x, None, 2, "relu", 1.2, 0.3, 1.0)
    assert np.allclose(k, ref, rtol=1e-12, atol=1e-14)
    ref_lml = O.mvn_logpdf(y, ref + 1e-3 * np.eye(n))
    assert abs(lml - ref_lml) < 1e-9 * abs(ref_lml) and abs(lml_max - lml) < 1e-9 * abs(lml)


# ------------------------------------------------------------------ paired lower-trapezoid layout
def _unpack_numpy(stage, n, world, S):
    """NumPy restatement of smn_unpack_lower_blocks (tile-granular lower triangle)."""
    k = np.full((n, n), np.nan)
    for b in range(2 * world):
        rb, re = S.block_range(n, world, b)
        off, ld = S.block_offset(n, world, b)
        for r in range(rb, re):
            cend = min(n, (r // S.TILE + 1) * S.TILE)
            k[r, :cend] = stage[off + (r - rb) * ld: off + (r - rb) * ld + cend]
    return k


def test_paired_layout_covers_rows_once_is_balanced_and_packs_without_overlap():
    from smnngp import sharding as S
    for n, w in [(16384, 8), (16384, 2), (32768, 8), (4096, 4), (301, 2), (1000, 3), (128, 8), (1, 1)]:
        h = S.block_rows(n, w)
        assert h % S.TILE == 0 and 2 * w * h >= n
        seen = np.zeros(n, dtype=int)
        used = np.zeros(w * S.paired_chunk_elems(n, w), dtype=np.int8) if n <= 4096 else None
        work = []
        for r in range(w):
            lo, hi = S.paired_blocks(w, r)
            assert S.block_owner(w, lo) == r and S.block_owner(w, hi) == r and lo + hi == 2 * w - 1
            tiles = 0
            for b in (lo, hi):
                rb, re = S.block_range(n, w, b)
                seen[rb:re] += 1
                off, ld = S.block_offset(n, w, b)
                assert ld == (b + 1) * h and ld >= re                       # a packed row holds columns [0, re)
                assert r * S.paired_chunk_elems(n, w) <= off
                assert off + h * ld <= (r + 1) * S.paired_chunk_elems(n, w)  # stays inside the owner's chunk
                if used is not None:
                    used[off: off + h * ld] += 1
                tiles += sum(t + 1 for t in range(rb // S.TILE, -(-re // S.TILE)))
            work.append(tiles)
        assert (seen == 1).all()
        if used is not None:
            assert used.max() <= 1
        if n % (2 * w * S.TILE) == 0:                                       # exact split: perfectly balanced
            assert max(work) == min(work)
            total = (n // S.TILE) * (n // S.TILE + 1) // 2
            assert sum(work) == total
    with pytest.raises(ValueError):
        S.paired_blocks(2, 2)
    with pytest.raises(ValueError):
        S.block_owner(2, 4)


def _paired_worker(rank, world, port, n, d, q):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from oracle import nngp_oracle as O
    from smnngp import sharding as S
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(0)
        x = rng.standard_normal((n, d)); y = rng.standard_normal(n)
        chunk = S.paired_chunk_elems(n, world)
        stage = torch.full((world * chunk,), float("nan"), dtype=torch.float64)
        sn = stage.numpy()
        for b in S.paired_blocks(world, rank):
            rb, re = S.block_range(n, world, b)
            if re <= rb:
                continue
            rows = O.mlp_kernel(x[rb:re], x[:re], 2, "relu", 1.2, 0.3, 1.0)   # stand-in for smn_kernel_mlp_lower_rows
            rows[np.arange(re - rb), np.arange(rb, re)] = O.diag_recursion((x[rb:re] ** 2).sum(1) / d, 2, "relu", 1.2, 0.3, 1.0)
            off, ld = S.block_offset(n, world, b)
            for i in range(re - rb):
                sn[off + i * ld: off + i * ld + re] = rows[i]
        send = stage[rank * chunk: (rank + 1) * chunk].clone()
        dist.all_gather_into_tensor(stage, send)
        k = _unpack_numpy(stage.numpy(), n, world, S)
        kl = np.tril(k)
        ks = kl + np.tril(kl, -1).T
        lml = O.mvn_logpdf(y, ks + 1e-3 * np.eye(n))
        if rank == 0:
            q.put((ks, lml))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n", [300, 513])
def test_world_size_2_gloo_paired_lower_blocks(n):
    pytest.importorskip("torch")
    import torch.multiprocessing as mp
    from oracle import nngp_oracle as O
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    d = 5
    procs = [ctx.Process(target=_paired_worker, args=(r, 2, port, n, d, q)) for r in range(2)]
    for p in procs:
        p.start()
    k, lml = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(0)
    x = rng.standard_normal((n, d)); y = rng.standard_normal(n)
    ref = O.mlp_kernel(x, None, 2, "relu", 1.2, 0.3, 1.0)
    assert np.allclose(k, ref, rtol=1e-12, atol=1e-14)
    ref_lml = O.mvn_logpdf(y, ref + 1e-3 * np.eye(n))
    assert abs(lml - ref_lml) < 1e-9 * abs(ref_lml)


# ------------------------------------------------------------------ cyclic column-first exchange: the real driver, a CPU backend
def test_cyclic_layout_is_balanced_and_every_column_range_is_an_equal_count_gather():
    from smnngp import sharding as S
    for n, w in [(16384, 8), (16384, 2), (32768, 8), (4096, 4), (301, 2), (1000, 3), (128, 8), (20000, 8), (5000, 5)]:
        t_all = S.tile_rows(n)
        owned = sorted(t for r in range(w) for t in S.rank_tile_rows(n, w, r))
        assert owned == list(range(t_all))                                      # every tile row exactly once
        for t in range(t_all):
            assert t in S.rank_tile_rows(n, w, S.tile_row_owner(w, t))
        tiles = [sum(t + 1 for t in S.rank_tile_rows(n, w, r)) for r in range(w)]
        if t_all % (2 * w) == 0:
            assert len(set(tiles)) == 1                                         # exactly balanced over whole periods
        else:
            assert max(tiles) - min(tiles) <= 2 * t_all                         # at most ~one period's worth apart
        cols = S.default_col_pieces(n, w)
        lay = S.col_layout(n, w, cols)
        assert cols[0] == 0 and cols[-1] == t_all and len(cols) - 1 <= S.MAX_COL_PIECES
        assert cols == sorted(set(cols))
        seen = set()
        for g in range(len(cols) - 1):
            base = cols[g] // w * w                              # first tile row of the group the piece starts in
            for r in range(w):
                # the rank's tile rows from that group down fit the piece's slots, in order
                mine = [t for t in S.rank_tile_rows(n, w, r) if t >= base]
                assert len(mine) <= lay["slots"][g]
                for i, t in enumerate(mine):
                    assert t == S.rank_tile_row(w, r, cols[g] // w + i)
                    for tc in range(cols[g], min(cols[g + 1], t + 1)):
                        assert (t, tc) not in seen
                        seen.add((t, tc))
            assert lay["count"][g] == lay["slots"][g] * S.TILE * lay["width"][g]
        assert len(seen) == t_all * (t_all + 1) // 2                             # every lower tile in exactly one piece
        assert lay["elems"] == sum(lay["count"])
    # the first super-panel in 256-column ranges (its first sub-panels wait for 15 MB per rank, not 59), then one per super-panel
    assert S.default_col_pieces(16384, 8) == [0, 2, 4, 6] + list(range(8, 129, 8))
    assert S.default_col_pieces(32768, 8) == [0, 2, 4, 6] + list(range(8, 256, 16)) + [256]
    assert S.default_col_pieces(16384, 1) == list(range(0, 129, 8))              # one rank: nothing to wait for
    lay = S.col_layout(16384, 8, S.default_col_pieces(16384, 8))
    assert lay["elems"] * 8 < 1.06 * 128 * 129 // 2 * 128 * 128                  # 5.4 % above-diagonal padding
    assert 7 * lay["count"][0] * 4 < 15e6                                        # bytes a rank receives for range 0
    with pytest.raises(ValueError):
        S.col_layout(1000, 3, [0, 4, 4, 8])          # boundaries must ascend
    with pytest.raises(ValueError):
        S.col_layout(1000, 2, [0, 4])                # ... and span every tile column


class _CpuBackend:
    """The device steps of sharding.lml_sharded_cols restated on the host: oracle rows for the build, a gloo all-gather
    per piece, NumPy for the scatter (the mapping of csrc/comm.hip scatter_piece_kernel, jitter on the diagonal included)
    and the head.  `mine` and `stage` are NumPy arrays handed through the driver untouched."""

    def __init__(self, dist, torch, x, y, world, rank):
        self.dist, self.torch, self.x, self.y, self.world, self.rank = dist, torch, x, y, world, rank
        self.calls = []

    def comm_size(self):
        return self.dist.get_world_size()

    def begin(self, dtype_code, n, eps_abs):
        self.k = np.full((n, n), np.nan)
        self.eps = eps_abs
        self.calls.append("begin")

    def build_cols(self, dtype_code, spec, x_ptr, n, ldx, d, world, rank, cols, mine, ntk_mine=None):
        from oracle import nngp_oracle as O
        from smnngp import sharding as S
        _net, act, nh, w_std, b_std, lw = spec
        act = "relu" if act == 0 else "erf"
        lay = S.col_layout(n, world, cols)
        for t in S.rank_tile_rows(n, world, rank):
            rb, re = t * S.TILE, min(n, (t + 1) * S.TILE)
            both = O.mlp_kernel(self.x[rb:re], self.x[:re], nh, act, w_std, b_std, lw, get=("nngp", "ntk"))
            full = O.mlp_kernel(self.x[rb:re], None, nh, act, w_std, b_std, lw, get=("nngp", "ntk"))
            kr, kt = both[0], both[1]
            kr[np.arange(re - rb), np.arange(rb, re)] = np.diag(full[0])        # exact diagonal, as the device kernel writes it
            kt[np.arange(re - rb), np.arange(rb, re)] = np.diag(full[1])
            for g in range(len(cols) - 1):
                if cols[g] > t:
                    break
                slot = t // world - cols[g] // world          # groups below the one the piece starts in
                assert S.rank_tile_row(world, rank, cols[g] // world + slot) == t
                c0, c1 = cols[g] * S.TILE, min(re, cols[g + 1] * S.TILE)
                wd = lay["width"][g]
                for i in range(re - rb):
                    o = lay["off"][g] + (slot * S.TILE + i) * wd
                    mine[o: o + c1 - c0] = kr[i, c0:c1]
                    if ntk_mine is not None:
                        ntk_mine[o: o + c1 - c0] = kt[i, c0:c1]
        self.calls.append(("build", tuple(cols)))

    def exchange_cols(self, dtype_code, mine, stage, n, world, cols, piece, ntk=None):
        if ntk is not None:                                   # the same piece of the NTK chunks into the caller's matrix
            self._exchange(ntk[0], ntk[1], n, world, cols, piece, ntk[2], 0.0)
        self._exchange(mine, stage, n, world, cols, piece, self.k, self.eps)
        self.calls.append(("exchange", piece))

    def _exchange(self, mine, stage, n, world, cols, g, out, diag_add):
        from smnngp import sharding as S
        lay = S.col_layout(n, world, cols)
        cnt, off, wd = lay["count"][g], lay["off"][g], lay["width"][g]
        recv = self.torch.from_numpy(stage[world * off: world * (off + cnt)])
        self.dist.all_gather_into_tensor(recv, self.torch.from_numpy(mine[off: off + cnt].copy()))
        t_all = S.tile_rows(n)
        for r in range(world):                                   # scatter_piece_kernel, strip row by strip row
            for srow in range(lay["slots"][g] * S.TILE):
                j = cols[g] // world + srow // S.TILE
                t = j * world + (world - 1 - r if j % 2 else r)
                row = t * S.TILE + srow % S.TILE
                if t >= t_all or row >= n:
                    continue
                c0 = cols[g] * S.TILE
                cend = min(n, (t + 1) * S.TILE, c0 + wd)
                if cend <= c0:
                    continue
                src = stage[world * off + r * cnt + srow * wd: world * off + r * cnt + srow * wd + cend - c0]
                out[row, c0:cend] = src
                if diag_add and c0 <= row < cend:
                    out[row, row] += diag_add

    def lml(self, dtype_code, n, y_ptr, df, scale):
        from oracle import nngp_oracle as O
        kl = np.tril(self.k)
        self.ks = kl + np.tril(kl, -1).T                        # (the jitter is in it already: the scatter added it)
        lp = O.mvn_logpdf(self.y, self.ks)
        return lp, 0.0, 0.0, 0


def _cols_worker(rank, world, port, n, d, cols, q, with_ntk=False):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    from smnngp import sharding as S
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(0)
        x = rng.standard_normal((n, d)); y = rng.standard_normal(n)
        cols = cols or S.default_col_pieces(n, world)
        elems = S.col_layout(n, world, cols)["elems"]
        mine = np.full(elems, np.nan); stage = np.full(world * elems, np.nan)
        be = _CpuBackend(dist, torch, x, y, world, rank)
        spec = (0, 0, 2, 1.2, 0.3, 1.0)
        ntk = None
        if with_ntk:
            tk = np.full((n, n), np.nan)
            ntk = (np.full(elems, np.nan), np.full(world * elems, np.nan), tk, n)
        phases = []
        lp, _, _, info = S.lml_sharded_cols(be, 1, spec, None, n, d, d, None, rank, world, mine, stage, 1e-3, cols=cols, ntk=ntk,
                                            progress=phases.append)
        # the driver's order: begin, ONE build, then the pieces in column order, then the factorisation
        assert be.calls[0] == "begin" and be.calls[1][0] == "build"
        assert [c[1] for c in be.calls[2:]] == list(range(len(cols) - 1))
        assert phases[0] == "begin" and phases[1] == "build" and phases[-1] == "factor" and len(phases) == len(cols) + 2
        try:
            S.lml_sharded_cols(be, 1, spec, None, n, d, d, None, rank, world + 1, mine, stage, 1e-3, cols=cols)
            raised = False
        except RuntimeError:
            raised = True
        if rank == 0:
            q.put((be.ks, lp, raised, None if ntk is None else np.tril(ntk[2])))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,cols,with_ntk", [(300, None, False), (513, [0, 2, 4, 5], False), (200, [0, 2], False),
                                              (300, [0, 2, 3], True), (700, [0, 1, 3, 6], False)])
def test_world_size_2_gloo_column_first_driver_with_cpu_backend(n, cols, with_ntk):
    """Two gloo processes run sharding.lml_sharded_cols itself -- the function bench.py --gpus N runs on the GPUs -- with the
    device steps replaced by a CPU backend: one build per rank in the cyclic layout, an all-gather per column range, the
    scatter (jitter on the diagonal), the head."""
    pytest.importorskip("torch")
    import torch.multiprocessing as mp
    from oracle import nngp_oracle as O
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    d = 5
    procs = [ctx.Process(target=_cols_worker, args=(r, 2, port, n, d, cols, q, with_ntk)) for r in range(2)]
    for p in procs:
        p.start()
    k, lml, raised, tk = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rng = np.random.default_rng(0)
    x = rng.standard_normal((n, d)); y = rng.standard_normal(n)
    ref = O.mlp_kernel(x, None, 2, "relu", 1.2, 0.3, 1.0)
    assert np.allclose(k, ref + 1e-3 * np.eye(n), rtol=1e-12, atol=1e-14)
    ref_lml = O.mvn_logpdf(y, ref + 1e-3 * np.eye(n))
    assert abs(lml - ref_lml) < 1e-9 * abs(ref_lml)
    assert raised            # a world the communicator does not have is refused
    if with_ntk:             # BASELINE config 5: the NTK rides the same exchange, column range by column range
        ref_t = O.mlp_kernel(x, None, 2, "relu", 1.2, 0.3, 1.0, get=("nngp", "ntk"))[1]
        rr, cc = np.indices((n, n))
        low = cc <= rr
        assert np.allclose(tk[low], ref_t[low], rtol=1e-12, atol=1e-14)
