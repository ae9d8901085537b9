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
