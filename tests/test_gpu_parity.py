"""GPU parity tests: the HIP path (through the C-ABI / the spax facade) against the CPU oracle on the
same seeded inputs.  Tolerances are the north-star's: 1e-5 relative in fp64 (asserted tighter where the
arithmetic allows), 1e-2 relative in fp32 (asserted at 2e-3 or better)."""
import ctypes as C

import numpy as np
import pytest
import scipy.linalg as sla

pytestmark = pytest.mark.gpu

from oracle import nngp_oracle as O  # noqa: E402  (test infrastructure only)

RTOL = {np.float32: 2e-3, np.float64: 1e-8}


@pytest.fixture(scope="module")
def L():
    from smnngp import _lib
    return _lib


@pytest.fixture(scope="module")
def ctx(L):
    return L.default_context()


from _tol import relerr, relerr_norm  # noqa: E402  norm-wise AND element-wise (|a-b| <= rtol |b| + rtol 1e-3 max|b|): tests/_tol.py


# ----------------------------------------------------------------------------- kernel build
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("act", ["relu", "erf"])
@pytest.mark.parametrize("net", ["mlp", "resnet"])
@pytest.mark.parametrize("n,d,layers", [(33, 6, 2), (200, 50, 4), (129, 3072, 1)])
def test_symmetric_kernel_nngp_ntk(dtype, act, net, n, d, layers):
    from smnngp import nt_kernels
    rng = np.random.default_rng(10 + n)
    x = rng.standard_normal((n, d)).astype(dtype)
    w, b, lw = 1.4, 0.3, 0.8
    fac = nt_kernels.get_mlp_kernel if net == "mlp" else nt_kernels.get_dense_resnet_kernel
    ofn = O.mlp_kernel if net == "mlp" else O.dense_resnet_kernel
    kfn = fac(layers, act=act, w_std=w, b_std=b, last_w_std=lw)
    got = kfn(x, None, get=("nngp", "ntk"))
    rk, rt = ofn(x.astype(np.float64), None, layers, act, w, b, lw, ("nngp", "ntk"))
    k, t = np.asarray(got.nngp), np.asarray(got.ntk)
    assert k.dtype == dtype and k.shape == (n, n)
    assert relerr(k, rk) < RTOL[dtype]
    assert relerr(t, rt) < RTOL[dtype] * 5
    assert np.array_equal(k, k.T)                      # mirrored store is exact
    # single get: the f32 MLP takes the correlation-space fast path when no NTK is asked for (layer_prog.hpp: single-sqrt
    # J / one-branch asin, |error| <= 3e-7 of the map's range), so the two results agree to that, not bit for bit
    k2 = np.asarray(kfn(x, x, get="nngp"))
    assert relerr(k2, rk) < RTOL[dtype]
    if dtype == np.float64:
        assert relerr(k2, k) < 1e-12
    else:
        assert np.abs(k2 - k).max() < 1e-6 * np.abs(k).max() and relerr(k2, k) < 5e-4
    assert np.array_equal(k2, k2.T)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("act", ["relu", "erf"])
def test_cross_kernel_and_zero_bias_defaults(dtype, act):
    from smnngp import nt_kernels
    rng = np.random.default_rng(3)
    x1 = rng.standard_normal((300, 17)).astype(dtype)
    x2 = rng.standard_normal((130, 17)).astype(dtype)
    kfn = nt_kernels.get_mlp_kernel(3, act=act)          # defaults w=1, b=0, lw=1 (nt_kernels.py:21)
    k = np.asarray(kfn(x1, x2, get="nngp"))
    ref = O.mlp_kernel(x1.astype(np.float64), x2.astype(np.float64), 3, act, 1.0, 0.0, 1.0)
    assert k.shape == (300, 130)
    assert relerr(k, ref) < RTOL[dtype]
    # reference default b_std = 1e-8 (regression/train.py:43)
    kfn = nt_kernels.get_mlp_kernel(2, act=act, w_std=1.0, b_std=1e-8, last_w_std=1.0)
    k = np.asarray(kfn(x2, None, get="nngp"))
    assert relerr(k, O.mlp_kernel(x2.astype(np.float64), None, 2, act, 1.0, 1e-8, 1.0)) < RTOL[dtype]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("act", ["relu", "erf"])
@pytest.mark.parametrize("net", ["mlp", "resnet"])
def test_zero_input_rows_with_zero_bias(dtype, act, net):
    """q = 0 at every layer for an all-zero input when b_std = 0: the correlation K/sqrt(q q') is 0/0 there.  No NaN
    may appear and the entries must be the oracle's (which defines that correlation as 0, like the reference's
    safe division), symmetric and cross, NNGP and NTK."""
    from smnngp import nt_kernels
    rng = np.random.default_rng(0)
    x = rng.standard_normal((130, 5)); x[3] = 0.0; x[129] = 0.0
    x2 = rng.standard_normal((4, 5)); x2[1] = 0.0
    fac = nt_kernels.get_mlp_kernel if net == "mlp" else nt_kernels.get_dense_resnet_kernel
    of = O.mlp_kernel if net == "mlp" else O.dense_resnet_kernel
    kfn = fac(3, act=act, w_std=1.3, b_std=0.0, last_w_std=1.0)
    with np.errstate(all="ignore"):
        rk, rt = of(x, None, 3, act, 1.3, 0.0, 1.0, ("nngp", "ntk"))
        ck, ct = of(x, x2, 3, act, 1.3, 0.0, 1.0, ("nngp", "ntk"))
    g = kfn(x.astype(dtype), None, get=("nngp", "ntk")); gc = kfn(x.astype(dtype), x2.astype(dtype), get=("nngp", "ntk"))
    for got, ref, scale in ((g.nngp, rk, 1), (g.ntk, rt, 5), (gc.nngp, ck, 1), (gc.ntk, ct, 5)):
        got = np.asarray(got, np.float64)
        assert np.isfinite(ref).all() and np.isfinite(got).all()
        assert relerr(got, ref) < RTOL[dtype] * scale


def test_bad_arguments_raise(L, ctx):
    from smnngp import nt_kernels
    with pytest.raises(KeyError):
        nt_kernels.get_mlp_kernel(2, act="tanh")
    kfn = nt_kernels.get_mlp_kernel(2)
    with pytest.raises(ValueError):
        kfn(np.zeros((4, 3)), np.zeros((4, 5)))
    with pytest.raises(ValueError):
        kfn(np.zeros((4, 3)), None, get="bogus")
    with pytest.raises(L.SmnError):
        nt_kernels.get_mlp_kernel(40)(np.ones((4, 3)))      # more activation layers than supported


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_gram_then_recursion_equals_fused(L, ctx, dtype):
    rng = np.random.default_rng(5)
    n1, n2, d = 260, 132, 40
    x1 = ctx.to_device(rng.standard_normal((n1, d)).astype(dtype))
    x2 = ctx.to_device(rng.standard_normal((n2, d)).astype(dtype))
    code = L.dtype_code(dtype)
    k0 = ctx.empty((n1, n2), dtype); q1 = ctx.empty((n1,), dtype); q2 = ctx.empty((n2,), dtype)
    ctx.call("smn_gram", code, x1.ptr, n1, d, x2.ptr, n2, d, d, k0.ptr, n2, q1.ptr, q2.ptr)
    rk0, rq1, rq2 = O.input_gram(x1.numpy().astype(np.float64), x2.numpy().astype(np.float64))
    assert relerr(k0.numpy(), rk0) < RTOL[dtype] and relerr(q1.numpy(), rq1) < RTOL[dtype]
    assert relerr(q2.numpy(), rq2) < RTOL[dtype]
    for act in ("relu", "erf"):
        k = ctx.empty((n1, n2), dtype); t = ctx.empty((n1, n2), dtype)
        ctx.call("smn_recursion", code, L.NET_MLP, L.ACT[act], 3, 1.3, 0.2, 0.9, k0.ptr, n1, n2, n2, q1.ptr, q2.ptr,
                 0, L.GET_NNGP | L.GET_NTK, k.ptr, t.ptr, n2)
        rk, rt = O.mlp_kernel(x1.numpy().astype(np.float64), x2.numpy().astype(np.float64), 3, act, 1.3, 0.2, 0.9,
                              ("nngp", "ntk"))
        assert relerr(k.numpy(), rk) < RTOL[dtype]
        assert relerr(t.numpy(), rt) < RTOL[dtype] * 5


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n", [64, 203, 516])
@pytest.mark.parametrize("net,act,layers", [("mlp", "relu", 4), ("mlp", "erf", 2), ("resnet", "relu", 2)])
def test_symmetric_recursion_lower_tiles_and_mirror(L, ctx, dtype, n, net, act, layers):
    """smn_recursion on a symmetric K0 (find.py's sweep form): the lower-tile + LDS-mirror kernel must fill BOTH
    triangles, match the oracle, be exactly symmetric, and agree with the cross form of the same kernel."""
    rng = np.random.default_rng(n)
    d, ld = 24, (n + 3) // 4 * 4                                         # 16-byte aligned rows (API contract)
    xh = rng.standard_normal((n, d)).astype(dtype)
    code = L.dtype_code(dtype)
    netc = L.NET_MLP if net == "mlp" else L.NET_DENSE_RESNET
    ofn = O.mlp_kernel if net == "mlp" else O.dense_resnet_kernel
    rk, rt = ofn(xh.astype(np.float64), None, layers, act, 1.2, 0.3, 0.9, ("nngp", "ntk"))
    got = {}
    c = ctx
    x = c.to_device(xh)
    k0 = c.empty((n, ld), dtype); q = c.empty((n,), dtype)
    c.call("smn_gram", code, x.ptr, n, d, None, 0, 0, d, k0.ptr, ld, q.ptr, None)
    for sym in (1, 0):                 # symmetric = 0: the same K0 through the cross form (every tile of the square, no mirror)
        k = c.to_device(np.full((n, ld), np.nan, dtype)); t = c.to_device(np.full((n, ld), np.nan, dtype))
        c.call("smn_recursion", code, netc, L.ACT[act], layers, 1.2, 0.3, 0.9, k0.ptr, n, n, ld, q.ptr, q.ptr,
               sym, L.GET_NNGP | L.GET_NTK, k.ptr, t.ptr, ld)
        got[sym] = (k.numpy()[:, :n], t.numpy()[:, :n])
        assert relerr(got[sym][0], rk) < RTOL[dtype] and relerr(got[sym][1], rt) < RTOL[dtype] * 5
        k1 = c.to_device(np.full((n, ld), np.nan, dtype))                    # NNGP only (the f32 ReLU fast path)
        c.call("smn_recursion", code, netc, L.ACT[act], layers, 1.2, 0.3, 0.9, k0.ptr, n, n, ld, q.ptr, q.ptr,
               sym, L.GET_NNGP, k1.ptr, None, ld)
        k1h = k1.numpy()[:, :n]
        assert relerr(k1h, rk) < RTOL[dtype]
        if sym == 1:
            assert (got[sym][0] == got[sym][0].T).all() and (got[sym][1] == got[sym][1].T).all()
            assert (k1h == k1h.T).all()
    il = np.tril_indices(n, -1)        # (the cross form has no closed-form diagonal)
    assert relerr(got[1][0][il], got[0][0][il]) < (1e-6 if dtype == np.float32 else 1e-14)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_row_shard_equals_rows_of_full_kernel(L, ctx, dtype):
    rng = np.random.default_rng(6)
    n, d = 301, 24
    xh = rng.standard_normal((n, d)).astype(dtype)
    x = ctx.to_device(xh)
    ref = O.mlp_kernel(xh.astype(np.float64), None, 2, "relu", 1.2, 0.1, 1.0)
    for rb, re in ((0, 76), (76, 200), (200, 301)):
        out = ctx.empty((re - rb, n), dtype)
        ctx.call("smn_kernel_mlp_rows", L.dtype_code(dtype), L.NET_MLP, L.ACT["relu"], 2, 1.2, 0.1, 1.0, x.ptr, n, d, d,
                 rb, re, L.GET_NNGP, out.ptr, None, n)
        assert relerr(out.numpy(), ref[rb:re]) < RTOL[dtype]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("method", ["lower_rows", "shard"])
@pytest.mark.parametrize("n,world", [(301, 1), (1000, 2), (1500, 3), (2048, 8), (130, 4)])
def test_paired_lower_block_shards_assemble_the_lower_triangle(L, ctx, dtype, n, world, method):
    """Multi-GPU build protocol rehearsed on one GPU: every rank's two smn_kernel_mlp_lower_rows
    calls write its chunk of the staging buffer (the all-gather is then the identity), smn_unpack_lower_blocks
    assembles K; its lower triangle must be the oracle's and smn_lml must not look at anything else."""
    from smnngp import sharding as S
    rng = np.random.default_rng(n + world)
    d = 24
    xh = rng.standard_normal((n, d)).astype(dtype)
    yh = rng.standard_normal(n).astype(dtype)
    x = ctx.to_device(xh); y = ctx.to_device(yh)
    code, es = L.dtype_code(dtype), np.dtype(dtype).itemsize
    chunk = S.paired_chunk_elems(n, world)
    stage = ctx.to_device(np.full(world * chunk, np.nan, dtype))            # NaN poison: unwritten = visible
    k = ctx.to_device(np.full((n, n), np.nan, dtype))
    for r in range(world):
        if method == "shard":                                                # one launch per rank (bench.py's form)
            ctx.call("smn_kernel_mlp_shard", code, L.NET_MLP, L.ACT["relu"], 2, 1.2, 0.1, 1.0, x.ptr, n, d, d,
                     world, r, S.block_rows(n, world), L.GET_NNGP, C.c_void_p(stage.ptr.value + r * chunk * es), None)
            continue
        for b in S.paired_blocks(world, r):
            rb, re = S.block_range(n, world, b)
            if re <= rb:
                continue
            off, ld = S.block_offset(n, world, b)
            ctx.call("smn_kernel_mlp_lower_rows", code, L.NET_MLP, L.ACT["relu"], 2, 1.2, 0.1, 1.0, x.ptr, n, d, d,
                     rb, re, L.GET_NNGP, C.c_void_p(stage.ptr.value + off * es), None, ld)
    ctx.call("smn_unpack_lower_blocks", code, stage.ptr, n, world, S.block_rows(n, world), k.ptr, n)
    ref = O.mlp_kernel(xh.astype(np.float64), None, 2, "relu", 1.2, 0.1, 1.0)
    got = k.numpy().astype(np.float64)
    il = np.tril_indices(n)
    assert np.isfinite(got[il]).all()
    assert relerr(got[il], ref[il]) < RTOL[dtype]
    iu = np.triu_indices(n, S.TILE)                                          # beyond the diagonal tile: untouched
    assert np.isnan(got[iu]).all()
    eps = 1e-3 if dtype == np.float32 else 1e-6
    lp, quad, logdet, info = C.c_double(), C.c_double(), C.c_double(), C.c_int()
    ctx.call("smn_lml", code, k.ptr, n, n, y.ptr, eps, 0.0, 1.0, C.byref(lp), C.byref(quad), C.byref(logdet), C.byref(info))
    want = O.mvn_logpdf(yh.astype(np.float64), ref + eps * np.eye(n))
    assert info.value == 0 and abs(lp.value - want) < (2e-3 if dtype == np.float32 else 1e-8) * abs(want)
    # the same likelihood straight from the staging buffer (no separate K): identical bits
    lp2, q2, ld2, i2 = C.c_double(), C.c_double(), C.c_double(), C.c_int()
    ctx.call("smn_lml_from_blocks", code, stage.ptr, n, world, S.block_rows(n, world), y.ptr, eps, 0.0, 1.0,
             C.byref(lp2), C.byref(q2), C.byref(ld2), C.byref(i2))
    assert i2.value == 0 and lp2.value == lp.value and q2.value == quad.value and ld2.value == logdet.value


def test_build_lower_sharded_single_rank_and_bad_geometry(L, ctx):
    from smnngp import sharding as S
    rng = np.random.default_rng(3)
    n, d = 700, 16
    xh = rng.standard_normal((n, d))
    x = ctx.to_device(xh)
    stage = ctx.empty((S.paired_chunk_elems(n, 1),), np.float64)
    k = ctx.to_device(np.zeros((n, n)))
    S.build_lower_sharded(ctx, L.F64, 8, L.NET_MLP, L.ACT["erf"], 3, 1.1, 0.2, 0.9, x.ptr, n, d, d, 0, 1, stage.ptr, k.ptr, n)
    ref, reft = O.mlp_kernel(xh, None, 3, "erf", 1.1, 0.2, 0.9, ("nngp", "ntk"))
    il = np.tril_indices(n)
    assert relerr(k.numpy()[il], ref[il]) < RTOL[np.float64]
    # NNGP + NTK together (config 5's flow): a second staging buffer, both gathered and unpacked
    stage_t = ctx.empty((S.paired_chunk_elems(n, 1),), np.float64)
    k2 = ctx.to_device(np.zeros((n, n))); t2 = ctx.to_device(np.zeros((n, n)))
    S.build_lower_sharded(ctx, L.F64, 8, L.NET_MLP, L.ACT["erf"], 3, 1.1, 0.2, 0.9, x.ptr, n, d, d, 0, 1, stage.ptr, k2.ptr, n,
                          ntk_stage_ptr=stage_t.ptr, ntk_ptr=t2.ptr)
    assert relerr(k2.numpy()[il], ref[il]) < RTOL[np.float64] and relerr(t2.numpy()[il], reft[il]) < 5 * RTOL[np.float64]
    with pytest.raises(L.SmnError):
        ctx.call("smn_unpack_lower_blocks", L.F64, stage.ptr, n, 1, 100, k.ptr, n)        # block_rows not a tile multiple
    with pytest.raises(L.SmnError):
        ctx.call("smn_unpack_lower_blocks", L.F64, stage.ptr, n, 1, 128, k.ptr, n)        # 2*1*128 rows < n
    with pytest.raises(L.SmnError):
        ctx.call("smn_kernel_mlp_lower_rows", L.F64, L.NET_MLP, L.ACT["erf"], 3, 1.1, 0.2, 0.9, x.ptr, n, d, d,
                 0, 256, L.GET_NNGP, stage.ptr, None, 128)                                 # ldk < row_end


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_shard_build_ntk_erf(L, ctx, dtype):
    """smn_kernel_mlp_shard with both outputs (C5's shape of work: erf NNGP + NTK), three ranks on one GPU."""
    from smnngp import sharding as S
    rng = np.random.default_rng(11)
    n, d, world = 900, 40, 3
    xh = rng.standard_normal((n, d)).astype(dtype)
    x = ctx.to_device(xh)
    code, es = L.dtype_code(dtype), np.dtype(dtype).itemsize
    chunk, h = S.paired_chunk_elems(n, world), S.block_rows(n, world)
    sk = ctx.to_device(np.full(world * chunk, np.nan, dtype)); st = ctx.to_device(np.full(world * chunk, np.nan, dtype))
    for r in range(world):
        ctx.call("smn_kernel_mlp_shard", code, L.NET_MLP, L.ACT["erf"], 3, 1.3, 0.2, 0.8, x.ptr, n, d, d, world, r, h,
                 L.GET_NNGP | L.GET_NTK, C.c_void_p(sk.ptr.value + r * chunk * es), C.c_void_p(st.ptr.value + r * chunk * es))
    k = ctx.to_device(np.zeros((n, n), dtype)); t = ctx.to_device(np.zeros((n, n), dtype))
    ctx.call("smn_unpack_lower_blocks", code, sk.ptr, n, world, h, k.ptr, n)
    ctx.call("smn_unpack_lower_blocks", code, st.ptr, n, world, h, t.ptr, n)
    rk, rt = O.mlp_kernel(xh.astype(np.float64), None, 3, "erf", 1.3, 0.2, 0.8, get=("nngp", "ntk"))
    il = np.tril_indices(n)
    assert relerr(k.numpy()[il], rk[il]) < RTOL[dtype]
    assert relerr(t.numpy()[il], rt[il]) < RTOL[dtype] * 5
    with pytest.raises(L.SmnError):
        ctx.call("smn_kernel_mlp_shard", code, L.NET_MLP, L.ACT["erf"], 3, 1.3, 0.2, 0.8, x.ptr, n, d, d, world, 3, h,
                 L.GET_NNGP, sk.ptr, None)                                   # rank out of range
    with pytest.raises(L.SmnError):
        ctx.call("smn_kernel_mlp_shard", code, L.NET_MLP, L.ACT["erf"], 3, 1.3, 0.2, 0.8, x.ptr, n, d, d, world, 0, h,
                 L.GET_NNGP | L.GET_NTK, sk.ptr, None)                       # NTK asked for, no buffer


def test_rccl_single_rank_communicator_allgather(L):
    """librccl is dlopen'ed and driven through its C ABI (unique id by value, comm init, all-gather, destroy).
    One rank is all a 1-GPU box can host; it still exercises every RCCL entry point the N>1 path uses."""
    c = L.Context(0)
    uid = C.create_string_buffer(128)
    assert L._lib.smn_comm_unique_id(uid) == 0
    c.call("smn_comm_init", 1, 0, uid)
    with pytest.raises(L.SmnError):
        c.call("smn_comm_init", 1, 0, uid)                                   # already initialised
    src = np.arange(4096, dtype=np.float32)
    a = c.to_device(src); b = c.to_device(np.zeros_like(src))
    c.call("smn_allgather", 1, L.F32, a.ptr, b.ptr, src.size)                # out of place
    assert (b.numpy() == src).all()
    c.call("smn_allgather", 1, L.F32, a.ptr, a.ptr, src.size)                # in place (the bench's form)
    assert (a.numpy() == src).all()
    a64 = c.to_device(src.astype(np.float64)); b64 = c.to_device(np.zeros(src.size))
    c.call("smn_allgather", 1, L.F64, a64.ptr, b64.ptr, src.size)
    with pytest.raises(L.SmnError):
        c.call("smn_allgather", 2, L.F32, a.ptr, b.ptr, src.size)            # a world the communicator does not have
    assert (b64.numpy() == src).all()
    c.call("smn_comm_destroy")
    c.call("smn_allgather", 1, L.F32, a.ptr, b.ptr, src.size)                # no communicator, a world of one: plain copy
    with pytest.raises(L.SmnError):
        c.call("smn_allgather", 2, L.F32, a.ptr, b.ptr, src.size)            # world > 1 without a communicator: refused, not copied
    assert (b.numpy() == src).all()


@pytest.mark.parametrize("torch_first", [True, False])
def test_rccl_communicator_in_a_process_that_also_holds_torch(torch_first):
    """bench.py's N>1 path imports torch (gloo rendezvous) in the process that then opens the RCCL communicator.
    PyTorch's ROCm wheel carries its own librccl.so; comm.hip must use the copy already in the process (a second one
    beside it made ncclCommInitRank fail).  Own process: the import order is the point."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    imports = ["import torch, torch.distributed", "from smnngp import _lib as L"]
    if not torch_first:
        imports.reverse()
    code = "\n".join(["import sys, ctypes as C, numpy as np", "sys.path.insert(0, %r)" % root] + imports + [
        "import torch",
        "c = L.Context(0)",
        "uid = C.create_string_buffer(128)",
        "assert L._lib.smn_comm_unique_id(uid) == 0",
        "c.call('smn_comm_init', 1, 0, uid)",
        "src = np.arange(4096, dtype=np.float32)",
        "a = c.to_device(src); b = c.to_device(np.zeros_like(src))",
        "c.call('smn_allgather', 1, L.F32, a.ptr, b.ptr, src.size)",
        "assert (b.numpy() == src).all()",
        "c.call('smn_comm_destroy')",
        "print('ok')"])
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("act", ["relu", "erf"])
@pytest.mark.parametrize("shape,layers", [((7, 5, 4, 3), 2), ((9, 8, 8, 1), 4), ((5, 1, 1, 6), 3), ((3, 32, 32, 3), 1),
                                          # every pixels-per-lane form of the pair kernel, exact and ragged:
                                          ((4, 32, 32, 2), 4),      # 32 x 32: the register-only stencil, all 4 layers
                                          ((4, 16, 16, 2), 3),      # 256 px = 64 x 4, exact
                                          ((3, 28, 28, 1), 2),      # 784 px: ragged 16-per-lane form (dummy-slot lanes)
                                          ((3, 20, 20, 2), 3),      # 400 px: ragged, most of the last rounds empty
                                          ((2, 40, 40, 1), 2),      # 1600 px: ragged 64-per-lane form
                                          ((2, 64, 64, 1), 1)])     # 4096 px: the largest image, exact
def test_cnn_kernel(dtype, act, shape, layers):
    """get_cnn_kernel (nt_kernels.py:34-45): symmetric, cross and odd image shapes vs the oracle."""
    from smnngp import nt_kernels
    rng = np.random.default_rng(sum(shape))
    x = rng.standard_normal(shape).astype(dtype)
    x2 = rng.standard_normal((4,) + shape[1:]).astype(dtype)
    kfn = nt_kernels.get_cnn_kernel(layers, act=act, w_std=1.3, b_std=0.2, last_w_std=0.9)
    k = np.asarray(kfn(x, None, get="nngp"))
    ref = O.cnn_kernel(x.astype(np.float64), None, layers, act, 1.3, 0.2, 0.9)
    assert k.shape == (shape[0], shape[0]) and relerr(k, ref) < RTOL[dtype]
    assert np.array_equal(k, k.T)
    kc = np.asarray(kfn(x, x2, get="nngp"))
    refc = O.cnn_kernel(x.astype(np.float64), x2.astype(np.float64), layers, act, 1.3, 0.2, 0.9)
    assert kc.shape == (shape[0], 4) and relerr(kc, refc) < RTOL[dtype]
    with pytest.raises(NotImplementedError):
        kfn(x, None, get="ntk")


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("act", ["relu", "erf"])
def test_cnn_kernel_32x32_forms_agree_at_the_image_borders(L, ctx, dtype, act):
    """32 x 32 x 3 images: fp64 takes the 4x4-patch kernel (stencil in registers, halo by ds_bpermute), fp32 the LDS-map
    kernel; both against the oracle and against each other to fp32 accuracy.  Images with structure at the borders (a
    constant image makes every border and corner pixel a distinct case)."""
    rng = np.random.default_rng(11)
    n = 6
    x = rng.standard_normal((n, 32, 32, 3))
    x[0] = 1.0                      # constant image
    x[1, 0, :, :] = 5.0             # loud top row
    x[2, :, 31, :] = -4.0           # loud right column
    x[3, 31, 0, :] = 7.0            # loud corner

    def run(dt):
        xd = ctx.to_device(x.astype(dt))
        k = ctx.empty((n, n), dt)
        ctx.call("smn_kernel_cnn", L.dtype_code(dt), L.ACT[act], 4, 1.2, 0.3, 0.9, xd.ptr, n, None, 0, 32, 32, 3, L.FILL_FULL, k.ptr, n)
        return k.numpy()

    got = run(dtype)
    ref = O.cnn_kernel(x.astype(dtype).astype(np.float64), None, 4, act, 1.2, 0.3, 0.9)
    assert relerr(got, ref) < RTOL[dtype]
    other = run(np.float32 if dtype == np.float64 else np.float64)
    assert np.abs(got - other).max() < 2e-5 * np.abs(ref).max()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_cnn_kernel_tiled_pair_order(L, ctx, dtype):
    """Large pair counts take the XCD-tiled pair order (cnn.hip: from 4 tiles of pairs per XCD-resident workgroup on):
    symmetric (ragged last tiles, half-empty diagonal tiles) and cross kernels against the oracle."""
    rng = np.random.default_rng(3)
    n1, n2, shape = 1531, 1203, (6, 6, 1)
    x = rng.standard_normal((n1,) + shape).astype(dtype)
    x2 = rng.standard_normal((n2,) + shape).astype(dtype)
    xd, x2d = ctx.to_device(x), ctx.to_device(x2)
    ksd, kcd = ctx.empty((n1, n1), dtype), ctx.empty((n1, n2), dtype)
    code = L.dtype_code(dtype)
    ctx.call("smn_kernel_cnn", code, L.ACT["relu"], 2, 1.3, 0.2, 0.9, xd.ptr, n1, None, 0, 6, 6, 1, L.FILL_FULL, ksd.ptr, n1)
    ctx.call("smn_kernel_cnn", code, L.ACT["relu"], 2, 1.3, 0.2, 0.9, xd.ptr, n1, x2d.ptr, n2, 6, 6, 1, L.FILL_FULL, kcd.ptr, n2)
    ks, kc = ksd.numpy(), kcd.numpy()
    assert np.array_equal(ks, ks.T)
    ref = O.cnn_kernel(x.astype(np.float64), None, 2, "relu", 1.3, 0.2, 0.9)
    refc = O.cnn_kernel(x.astype(np.float64), x2.astype(np.float64), 2, "relu", 1.3, 0.2, 0.9)
    assert relerr(ks, ref) < RTOL[dtype] and relerr(kc, refc) < RTOL[dtype]
    assert np.abs(ks - ref).max() < RTOL[dtype] * np.abs(ref).max()           # no pair skipped or misplaced


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("act", ["relu", "erf"])
@pytest.mark.parametrize("shape,block", [((6, 8, 8, 3), 1), ((5, 16, 8, 2), 2), ((3, 32, 32, 3), 1), ((4, 8, 24, 1), 3)])
def test_conv_resnet_kernel(dtype, act, shape, block):
    """get_conv_resnet_kernel (nt_kernels.py:48-80): WideResnet blocks with strided convolutions and Conv / Identity
    shortcuts, symmetric and cross, square and non-square images, vs the oracle."""
    from smnngp import nt_kernels
    rng = np.random.default_rng(sum(shape) + block)
    x = rng.standard_normal(shape).astype(dtype)
    x2 = rng.standard_normal((3,) + shape[1:]).astype(dtype)
    kfn = nt_kernels.get_conv_resnet_kernel(block, 10, act=act, w_std=1.2, b_std=0.3, last_w_std=0.9)
    k = np.asarray(kfn(x, None, get="nngp"))
    ref = O.conv_resnet_kernel(x.astype(np.float64), None, block, act, 1.2, 0.3, 0.9)
    assert k.shape == (shape[0], shape[0]) and relerr(k, ref) < RTOL[dtype]
    assert np.array_equal(k, k.T)
    kc = np.asarray(kfn(x, x2, get="nngp"))
    refc = O.conv_resnet_kernel(x.astype(np.float64), x2.astype(np.float64), block, act, 1.2, 0.3, 0.9)
    assert kc.shape == (shape[0], 3) and relerr(kc, refc) < RTOL[dtype]


def test_conv_resnet_kernel_rejects_what_it_cannot_do(L, ctx):
    from smnngp import nt_kernels
    kfn = nt_kernels.get_conv_resnet_kernel(1, 10)
    with pytest.raises(L.SmnError):
        kfn(np.zeros((2, 12, 12, 3)), None)                                   # 12 is not a multiple of 8
    with pytest.raises(L.SmnError):
        nt_kernels.get_conv_resnet_kernel(7, 10)(np.zeros((2, 8, 8, 3)), None)  # op list too long
    with pytest.raises(KeyError):
        nt_kernels.get_conv_resnet_kernel(1, 10, act="tanh")
    with pytest.raises(NotImplementedError):
        kfn(np.zeros((2, 8, 8, 3)), None, get="ntk")


def test_cnn_kernel_feeds_the_same_inference_heads():
    """SVSP-style consumers (spax/models.py:38-43) and NNGPKernel.predict use the conv kernel through the
    generic kernel_fn path: joint kernel -> smn_predict."""
    from smnngp import nt_kernels, predict
    rng = np.random.default_rng(77)
    x = rng.standard_normal((20, 6, 6, 2)); xt = rng.standard_normal((5, 6, 6, 2)); y = rng.standard_normal((20, 2))
    kfn = nt_kernels.get_cnn_kernel(2, act="relu", w_std=1.2, b_std=0.1, last_w_std=1.0)
    mean, cov = predict.gradient_descent_mse_ensemble(kfn, x, y, diag_reg=1e-3)(x_test=xt)
    kw = dict(num_hiddens=2, act="relu", w_std=1.2, b_std=0.1, last_w_std=1.0)
    rm, rc = O.predict(O.cnn_kernel(x, None, **kw), O.cnn_kernel(xt, x, **kw), O.cnn_kernel(xt, None, **kw), y, 1e-3)
    assert relerr_norm(np.asarray(mean), rm) < 1e-7 and relerr_norm(np.asarray(cov), rc) < 1e-7


def test_cifar_shaped_conv_kernel_student_t_lml_and_test_nll_against_the_oracle():
    """BASELINE config 3 at a size the oracle finishes in seconds: 176 training + 24 test images of 32x32x3 (the shape
    that takes the 4x4-patch-per-lane pair kernel and its patch-order tables), fp64, 4 conv layers.  The whole chain --
    conv-NNGP kernel -> Student-t log-marginal likelihood (smn_lml with df > 0) and SPR.test_nll (posterior through the
    joint conv kernel + the b/a K + 1e-6 I quadratic form) -- against O.cnn_kernel + O.mvt_logpdf and the oracle's
    restatement of spax/models.py:100-120 / likelihoods.py:52-65 on those kernels."""
    import scipy.linalg as sla_
    from smnngp import nt_kernels
    from smnngp.spax.kernels import NNGPKernel
    from smnngp.spax.likelihoods import StudentTLikelihood
    from smnngp.spax.models import SPR
    rng = np.random.default_rng(303)
    n, t, nl = 176, 24, 4
    xa = rng.uniform(0.0, 1.0, (n + t, 32, 32, 3))
    xa = (xa - np.array([0.4914, 0.4822, 0.4465])) / np.array([0.247, 0.243, 0.261])      # classification/data.py:138-139
    x, xt = xa[:n], xa[n:]
    y = (rng.integers(0, 10, n) == 3).astype(np.float64) - 0.1
    yt = (rng.integers(0, 10, t) == 3).astype(np.float64) - 0.1
    alpha, beta, eps = 2.0, 2.0, 1e-4
    kw = dict(num_hiddens=nl, act="relu", w_std=1.3, b_std=0.2, last_w_std=1.0)
    kall = O.cnn_kernel(xa, None, **kw)                                                   # one oracle pass: all blocks
    k_dd, k_td, k_tt = kall[:n, :n], kall[n:, :n], kall[n:, n:]
    # --- the kernel itself (patch-order path) and the Student-t LML on it
    kfn = nt_kernels.get_cnn_kernel(nl, act="relu", w_std=1.3, b_std=0.2, last_w_std=1.0)
    got = np.asarray(kfn(x, None, "nngp"))
    assert relerr(got, k_dd) < RTOL[np.float64]
    kernel = NNGPKernel(lambda w, b, l: nt_kernels.get_cnn_kernel(nl, act="relu", w_std=w, b_std=b, last_w_std=l), 1.3, 0.2, 1.0)
    model = SPR(kernel, StudentTLikelihood(alpha, beta), x, y, 0.0, 1.0, eps=eps)
    want_lp = O.mvt_logpdf(y, (beta / alpha) * (k_dd + eps * np.eye(n)), 2.0 * alpha)
    assert abs(model.loss() - (-want_lp / n)) < 1e-7 * abs(want_lp / n)
    # --- test_nll: posterior (relative ridge) + Student-t predictive with the K-without-eps quadratic form
    mean, cov = O.predict(k_dd, k_td, k_tt, y[:, None], diag_reg=eps)
    df = 2.0 * alpha
    khat = (beta / alpha) * k_dd + 1e-6 * np.eye(n)
    d = df + y @ sla_.cho_solve(sla_.cho_factor(khat, lower=True), y)
    sigma = np.sqrt(np.diag(d / (df + n) * (beta / alpha) * cov))
    want_nll = -float(np.mean(O.student_t_logpdf(yt, df + n, mean.ravel(), sigma)))
    got_nll = model.test_nll(xt, yt)
    assert abs(got_nll - want_nll) < 1e-5 * max(1.0, abs(want_nll))                      # cond ~1e9 quadratic form: DESIGN section 1


# ----------------------------------------------------------------------------- factorisation
def _spd(rng, n, dtype, cond=1e3):
    a = rng.standard_normal((n, n))
    q, _ = np.linalg.qr(a)
    ev = np.geomspace(1.0, cond, n)
    return ((q * ev) @ q.T).astype(dtype)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,m", [(128, 0), (300, 0), (256, 128), (391, 37), (1024, 0)])
def test_cholesky_and_schur(L, ctx, dtype, n, m):
    rng = np.random.default_rng(n + m)
    a = _spd(rng, n + m, np.float64)
    ad = ctx.to_device(a.astype(dtype))
    info, logdet = C.c_int(), C.c_double()
    ctx.call("smn_cholesky", L.dtype_code(dtype), ad.ptr, n + m, n, n + m, 0, 0.0, 0.0, C.byref(info), C.byref(logdet))
    got = ad.numpy().astype(np.float64)
    l = np.linalg.cholesky(a[:n, :n])
    tol = 1e-9 if dtype == np.float64 else 5e-3
    assert info.value == 0
    assert abs(logdet.value - 2 * np.log(np.diag(l)).sum()) < tol * max(1.0, abs(logdet.value))
    assert relerr_norm(np.tril(got[:n, :n]), l) < tol
    if m:
        w = sla.solve_triangular(l, a[:n, n:], lower=True).T        # B L^-T
        assert relerr_norm(got[n:, :n], w) < tol
        s = a[n:, n:] - w @ w.T
        assert relerr_norm(np.tril(got[n:, n:]), np.tril(s)) < tol


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,m", [(1024, 0), (896, 128), (9216, 0)])
def test_cholesky_reads_the_lower_triangle_only(L, ctx, dtype, n, m):
    """LAPACK 'L' semantics: whatever sits strictly above the diagonal (here NaN) must not reach the factor, the
    solved rows, the Schur complement or logdet.  Sizes are tile multiples so the matrix is factored in place."""
    rng = np.random.default_rng(n + m + 1)
    if n + m > 4096:                                                         # cheap SPD matrix for the large case
        a = rng.standard_normal((n + m, 64)); a = a @ a.T / 64 + np.eye(n + m)
    else:
        a = _spd(rng, n + m, np.float64)
    poisoned = a.copy()
    poisoned[np.triu_indices(n + m, 1)] = np.nan
    info, logdet = C.c_int(), C.c_double()
    ad = ctx.to_device(poisoned.astype(dtype))
    ctx.call("smn_cholesky", L.dtype_code(dtype), ad.ptr, n + m, n, n + m, 0, 0.0, 0.0, C.byref(info), C.byref(logdet))
    got = ad.numpy().astype(np.float64)
    ref = ctx.to_device(a.astype(dtype))
    info2, logdet2 = C.c_int(), C.c_double()
    ctx.call("smn_cholesky", L.dtype_code(dtype), ref.ptr, n + m, n, n + m, 0, 0.0, 0.0, C.byref(info2), C.byref(logdet2))
    want = ref.numpy().astype(np.float64)
    il = np.tril_indices(n + m)
    assert info.value == 0 and info2.value == 0 and logdet.value == logdet2.value
    assert np.isfinite(got[il]).all() and (got[il] == want[il]).all()        # bit-identical lower triangle


@pytest.mark.parametrize("dtype,n,m", [(np.float64, 6144, 128), (np.float32, 8192, 0)])
def test_cholesky_more_workgroups_than_cus_and_lookahead(L, ctx, dtype, n, m):
    """Sizes where a panel launch has more workgroups than the chip has CUs (f64: 16 rows per workgroup)
    and where the look-ahead stream runs beside the trailing update: late workgroups must still see the
    un-factored diagonal block."""
    rng = np.random.default_rng(99)
    g = rng.standard_normal((n + m, 64))
    a = g @ g.T / 64 + np.diag(rng.uniform(1.0, 2.0, n + m))
    ad = ctx.to_device(a.astype(dtype))
    info, logdet = C.c_int(), C.c_double()
    ctx.call("smn_cholesky", L.dtype_code(dtype), ad.ptr, n + m, n, n + m, 0, 0.0, 0.0, C.byref(info), C.byref(logdet))
    l = np.linalg.cholesky(a[:n, :n])
    tol = 1e-9 if dtype == np.float64 else 2e-3
    assert info.value == 0
    assert abs(logdet.value - 2 * np.log(np.diag(l)).sum()) < tol * abs(logdet.value)
    got = ad.numpy().astype(np.float64)
    assert relerr_norm(np.tril(got[:n, :n]), l) < tol
    if m:
        w = sla.solve_triangular(l, a[:n, n:], lower=True).T
        assert relerr_norm(got[n:, :n], w) < tol
        assert relerr_norm(np.tril(got[n:, n:]), np.tril(a[n:, n:] - w @ w.T)) < tol


@pytest.mark.parametrize("env", [
    {"SMN_CHAIN_CUS": "0"},                                    # no look-ahead: one stream, serial
    {"SMN_CHAIN_CUS": "64"},                                   # masked look-ahead, other reservation
    {"SMN_CHAIN_CUS": "8"},
    {"SMN_SUPER": "256"},                                      # a super-panel = one outer panel: no near updates
    {"SMN_SUPER": "512"},
    {"SMN_SUPER": "2048"},
    {"SMN_SUPER": "512", "SMN_CHAIN_CUS": "0"},
    {"SMN_XCD_MAP": "0"},                                      # linear tile order instead of the XCD patch order
    {"SMN_SUPER_WIDE_ROWS": "4096"},                           # 2048-column super-panels while 4096 rows are left, 1024 below
    {"SMN_SUPER_WIDE_ROWS": "0"},                              # 2048-column super-panels throughout
    {"SMN_PANEL_LEAF": "0"},                                   # the in-LDS micro-panel kernel (panel_kernel) instead of the register leaf
])
def test_cholesky_schedule_variants_agree(L, env):
    """Every schedule the environment switches select factors the same matrix to the same result (the default
    one is checked against LAPACK above): n = 8192 + 128 appended rows, fp32."""
    import os
    n, m = 8192, 128
    rng = np.random.default_rng(5)
    g = rng.standard_normal((n + m, 96)).astype(np.float32)
    a = (g @ g.T / 96 + np.diag(rng.uniform(1.0, 2.0, n + m))).astype(np.float32)

    def run(extra):
        old = {k: os.environ.get(k) for k in extra}
        os.environ.update(extra)
        try:
            c = L.Context(0)
        finally:
            for k, v in old.items():
                if v is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v
        ad = c.to_device(a)
        info, logdet = C.c_int(), C.c_double()
        c.call("smn_cholesky", L.F32, ad.ptr, n + m, n, n + m, 0, 0.0, 0.0, C.byref(info), C.byref(logdet))
        return info.value, logdet.value, ad.numpy()

    i0, ld0, f0 = run({})
    i1, ld1, f1 = run(env)
    # different K groupings round differently in fp32: agreement to fp32 accumulation accuracy, not bit for bit
    assert i0 == 0 and i1 == 0 and abs(ld1 - ld0) < 1e-5 * abs(ld0)
    il = np.tril_indices(n + m)
    assert relerr_norm(f1[il], f0[il]) < 1e-4


@pytest.mark.parametrize("dtype,n,m", [(np.float32, 2048, 128), (np.float32, 4352, 0), (np.float64, 1152, 128)])
def test_panel_register_leaf_agrees_with_the_lds_micro_panel_kernel(L, dtype, n, m):
    """panelr_kernel (register-resident 16x16 leaf on DPP broadcasts, block updates behind each leaf on all waves, image
    streamed in beside the first leaves) against panel_kernel (8-column micro-panels in LDS): two different summation
    orders of the same factorisation, so agreement to accumulation accuracy -- and twice the leaf kernel bit for bit (a
    hand-off between the helper waves and the row threads that came too early or too late would show as a run-to-run
    difference).  Sizes on both sides of the 16-row / 128-row workgroup switch (4096 rows)."""
    import os
    rng = np.random.default_rng(21)
    g = rng.standard_normal((n + m, 64)).astype(dtype)
    a = (g @ g.T / 64 + np.diag(rng.uniform(1.0, 2.0, n + m))).astype(dtype)
    out = []
    for leaf in ("0", "1", "1"):
        old = os.environ.get("SMN_PANEL_LEAF")
        os.environ["SMN_PANEL_LEAF"] = leaf
        try:
            c = L.Context(0)
        finally:
            if old is None:
                del os.environ["SMN_PANEL_LEAF"]
            else:
                os.environ["SMN_PANEL_LEAF"] = old
        ad = c.to_device(a)
        info, logdet = C.c_int(), C.c_double()
        c.call("smn_cholesky", L.dtype_code(dtype), ad.ptr, n + m, n, n + m, 0, 0.0, 0.0, C.byref(info), C.byref(logdet))
        assert info.value == 0
        out.append((logdet.value, ad.numpy()))
    il = np.tril_indices(n + m)
    tol = 1e-5 if dtype == np.float32 else 1e-12
    assert abs(out[0][0] - out[1][0]) < tol * abs(out[0][0])
    assert relerr_norm(out[1][1][il], out[0][1][il]) < tol
    assert out[1][0] == out[2][0] and np.array_equal(out[1][1][il], out[2][1][il])


@pytest.mark.parametrize("n,m", [(6144, 128), (4352, 0)])
def test_multi_pass_panel_workgroups_give_the_one_group_bits(L, ctx, n, m):
    """fp64 panels carry 16 rows per workgroup; with more row groups than CUs a workgroup takes several, the later ones through
    the MFMA update of their tile and the leaf's v-steps alone (cholesky.hip panelr_kernel `passes`).  Same instructions on the
    same operands: the factor of the one-group-per-workgroup launch bit for bit (384 groups -> two passes; 264 -> two, the
    second one partly empty), and LAPACK's to rounding."""
    rng = np.random.default_rng(n)
    g = rng.standard_normal((n + m, 48))
    a = g @ g.T / 48 + np.diag(rng.uniform(1.0, 2.0, n + m))
    out = []
    try:
        for passes in (4, 1, 4):
            ctx.call("smn_debug_panel_passes", passes)
            ad = ctx.to_device(a)
            info, logdet = C.c_int(), C.c_double()
            ctx.call("smn_cholesky", L.F64, ad.ptr, n + m, n, n + m, 0, 0.0, 0.0, C.byref(info), C.byref(logdet))
            assert info.value == 0
            out.append((logdet.value, ad.numpy()))
    finally:
        ctx.call("smn_debug_panel_passes", 4)
    il = np.tril_indices(n + m)
    assert out[0][0] == out[1][0] == out[2][0]
    assert np.array_equal(out[0][1][il], out[1][1][il]) and np.array_equal(out[0][1][il], out[2][1][il])
    ref = np.linalg.cholesky(a[:n, :n])
    assert np.abs(np.tril(out[0][1][:n, :n]) - ref).max() < 1e-11


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_split_build_of_the_fused_loss_gives_the_single_launch_bits(L, ctx, dtype):
    """smn_spr_loss from 112 tile rows on (look-ahead on) builds the bottom-right corner of the kernel matrix as a second launch
    on the bulk stream, beside the first super-panel's panel chain; the factorisation waits for it where it first touches those
    columns (kernel_build.hip run_build_t, heads.hip aug_finish, cholesky.hip need_columns).  Same tiles, same arithmetic: the
    single launch's bits, twice; and a matrix that is not positive definite in its FIRST columns (the factorisation gives up
    while the corner is still in flight) comes back as info != 0 with the context usable afterwards."""
    n, d = 14400, 48
    rng = np.random.default_rng(14)
    x = ctx.to_device(rng.standard_normal((n, d)).astype(dtype))
    y = ctx.to_device(rng.standard_normal((n, 1)).astype(dtype))

    def loss(eps):
        lp, quad, logdet, info = C.c_double(), C.c_double(), C.c_double(), C.c_int()
        ctx.call("smn_spr_loss", L.dtype_code(dtype), L.NET_MLP, L.ACT["relu"], 2, 1.0, 0.3, 1.0, x.ptr, n, d, d, y.ptr, eps, 0.0, 1.0,
                 C.byref(lp), C.byref(quad), C.byref(logdet), C.byref(info))
        return lp.value, quad.value, logdet.value, info.value

    split = [loss(1e-2), loss(1e-2)]
    try:
        ctx.call("smn_debug_split_build", 0)
        single = loss(1e-2)
    finally:
        ctx.call("smn_debug_split_build", 1)
    assert single[3] == 0 and split[0] == single and split[1] == single
    bad = loss(-5.0)                       # K - 5 I: the first pivot is already negative
    assert bad[3] == 1 and np.isnan(bad[0])
    assert loss(1e-2) == single


@pytest.mark.parametrize("dtype,n,m", [(np.float32, 9216, 128), (np.float64, 8192, 0)])
def test_cholesky_lookahead_is_bitwise_reproducible(L, ctx, dtype, n, m):
    """The look-ahead runs the block updates on a second stream beside the next block's panel chain.  A missing
    dependency would show as run-to-run differences: eight factorizations of the same matrix must agree bit for bit."""
    rng = np.random.default_rng(8)
    g = rng.standard_normal((n + m, 80)).astype(dtype)
    a = (g @ g.T / 80 + np.diag(rng.uniform(1.0, 2.0, n + m))).astype(dtype)
    first = None
    for rep in range(8):
        ad = ctx.to_device(a)
        info, logdet = C.c_int(), C.c_double()
        ctx.call("smn_cholesky", L.dtype_code(dtype), ad.ptr, n + m, n, n + m, 0, 0.0, 0.0, C.byref(info), C.byref(logdet))
        got = ad.numpy()
        assert info.value == 0
        if first is None:
            first = (got, logdet.value)
        else:
            il = np.tril_indices(n + m)
            assert logdet.value == first[1] and np.array_equal(got[il], first[0][il]), rep


def test_cholesky_shift_and_not_pd(L, ctx):
    rng = np.random.default_rng(11)
    n = 200
    a = _spd(rng, n, np.float64)
    ad = ctx.to_device(a)
    info, logdet = C.c_int(), C.c_double()
    ctx.call("smn_cholesky", L.F64, ad.ptr, n, n, n, n, 0.5, 0.25, C.byref(info), C.byref(logdet))
    ref = a + (0.5 + 0.25 * np.trace(a) / n) * np.eye(n)
    assert abs(logdet.value - np.linalg.slogdet(ref)[1]) < 1e-9 * abs(logdet.value)
    bad = a.copy(); bad[150, 150] = -1.0
    bd = ctx.to_device(bad)
    ctx.call("smn_cholesky", L.F64, bd.ptr, n, n, n, 0, 0.0, 0.0, C.byref(info), C.byref(logdet))
    assert info.value == 151 and np.isnan(logdet.value)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_trsm_lower(L, ctx, dtype):
    rng = np.random.default_rng(12)
    n, r = 333, 45
    l = np.linalg.cholesky(_spd(rng, n, np.float64, cond=100.0))
    b = rng.standard_normal((n, r))
    ld = ctx.to_device(l.astype(dtype)); bd = ctx.to_device(b.astype(dtype))
    ctx.call("smn_trsm", L.dtype_code(dtype), ld.ptr, n, n, bd.ptr, r, r, 0)
    ref = sla.solve_triangular(l, b, lower=True)
    assert relerr_norm(bd.numpy(), ref) < (1e-9 if dtype == np.float64 else 2e-3)
    # trans = 1: L^T X = B; together the two solves are cho_solve (K^-1 B)
    ctx.call("smn_trsm", L.dtype_code(dtype), ld.ptr, n, n, bd.ptr, r, r, 1)
    ref2 = sla.solve_triangular(l, ref, lower=True, trans="T")
    assert relerr_norm(bd.numpy(), ref2) < (1e-8 if dtype == np.float64 else 5e-3)
    assert relerr_norm(bd.numpy(), sla.cho_solve((l, True), b)) < (1e-8 if dtype == np.float64 else 5e-3)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_trsm_across_super_panels(L, ctx, dtype):
    """The solve-only sweep is two-level (left-looking inside a 1024-column super-panel, one far update of every later column
    behind it): a matrix of several super-panels, ragged in both dimensions, a strided right-hand side."""
    rng = np.random.default_rng(112)
    n, r, ldb = 2700, 205, 260
    l = np.linalg.cholesky(_spd(rng, n, np.float64, cond=50.0))
    b = rng.standard_normal((n, ldb))
    ld = ctx.to_device(l.astype(dtype)); bd = ctx.to_device(b.astype(dtype))
    ctx.call("smn_trsm", L.dtype_code(dtype), ld.ptr, n, n, bd.ptr, r, ldb, 0)
    got = bd.numpy()
    ref = sla.solve_triangular(l, b[:, :r], lower=True)
    assert relerr_norm(got[:, :r], ref) < (1e-9 if dtype == np.float64 else 2e-3)
    assert np.array_equal(got[:, r:], b[:, r:].astype(dtype))                 # columns beyond nrhs are not touched
    ctx.call("smn_trsm", L.dtype_code(dtype), ld.ptr, n, n, bd.ptr, r, ldb, 1)
    assert relerr_norm(bd.numpy()[:, :r], sla.cho_solve((l, True), b[:, :r])) < (1e-8 if dtype == np.float64 else 5e-3)


def test_transpose_entry_point(L, ctx):
    rng = np.random.default_rng(113)
    for dtype, rows, cols, lds, ldd in ((np.float32, 37, 70, 75, 40), (np.float64, 130, 33, 33, 130), (np.float32, 1, 1, 1, 1)):
        a = rng.standard_normal((rows, lds)).astype(dtype)
        out = np.full((cols, ldd), 7.0, dtype=dtype)
        ad, od = ctx.to_device(a), ctx.to_device(out)
        ctx.call("smn_transpose", L.dtype_code(dtype), od.ptr, ldd, ad.ptr, lds, rows, cols)
        got = od.numpy()
        assert np.array_equal(got[:, :rows], a[:, :cols].T)
        assert np.all(got[:, rows:] == 7.0)
    with pytest.raises(L.SmnError):
        ctx.call("smn_transpose", L.dtype_code(np.float32), od.ptr, 0, ad.ptr, 1, 1, 1)


# ----------------------------------------------------------------------------- heads
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_lml_gaussian_and_student_t(L, ctx, dtype):
    rng = np.random.default_rng(13)
    n = 270
    x = rng.standard_normal((n, 5)); y = rng.standard_normal(n)
    k = O.mlp_kernel(x, None, 2, "relu", 1.0, 0.5, 1.0)
    eps = 1e-2
    kd = ctx.to_device(k.astype(dtype)); yd = ctx.to_device(y.astype(dtype))
    lp, quad, logdet, info = C.c_double(), C.c_double(), C.c_double(), C.c_int()
    code = L.dtype_code(dtype)
    tol = 1e-8 if dtype == np.float64 else 2e-3
    ctx.call("smn_lml", code, kd.ptr, n, n, yd.ptr, eps, 0.0, 1.0, C.byref(lp), C.byref(quad), C.byref(logdet), C.byref(info))
    ref = O.mvn_logpdf(y, k + eps * np.eye(n))
    assert info.value == 0 and abs(lp.value - ref) < tol * abs(ref)
    kd = ctx.to_device(k.astype(dtype))
    ctx.call("smn_lml", code, kd.ptr, n, n, yd.ptr, eps, 4.0, 1.5, C.byref(lp), None, None, C.byref(info))
    ref = O.mvt_logpdf(y, 1.5 * (k + eps * np.eye(n)), 4.0)
    assert abs(lp.value - ref) < tol * abs(ref)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("n,t,c", [(245, 30, 1), (130, 140, 3)])
def test_predict_joint_and_fused(L, ctx, dtype, n, t, c):
    from smnngp import nt_kernels, predict
    rng = np.random.default_rng(14 + n)
    d = 6
    x = rng.standard_normal((n, d)).astype(dtype); xt = rng.standard_normal((t, d)).astype(dtype)
    y = rng.standard_normal((n, c)).astype(dtype)
    kw = dict(num_hiddens=2, act="relu", w_std=1.1, b_std=0.4, last_w_std=1.0)
    x64, xt64 = x.astype(np.float64), xt.astype(np.float64)
    kdd = O.mlp_kernel(x64, None, **kw); ktd = O.mlp_kernel(xt64, x64, **kw); ktt = O.mlp_kernel(xt64, None, **kw)
    eps = 1e-3 if dtype == np.float64 else 1e-2
    rmean, rcov = O.predict(kdd, ktd, ktt, y.astype(np.float64), diag_reg=eps)
    kfn = nt_kernels.get_mlp_kernel(2, act="relu", w_std=1.1, b_std=0.4, last_w_std=1.0)
    pf = predict.gradient_descent_mse_ensemble(kfn, x, y, diag_reg=eps)
    res = pf(x_test=xt, get="nngp", compute_cov=True)
    mean, cov = np.asarray(res[0]), np.asarray(res[1])
    tol = 1e-7 if dtype == np.float64 else 1e-2
    assert mean.shape == (t, c) and cov.shape == (t, t)
    assert relerr_norm(mean, rmean) < tol and relerr_norm(cov, rcov) < tol
    kt = kdd + eps * np.trace(kdd) / n * np.eye(n)
    rquad = [float(y[:, k].astype(np.float64) @ np.linalg.solve(kt, y[:, k].astype(np.float64))) for k in range(c)]
    assert np.allclose(res.quad, rquad, rtol=tol)
    # generic kernel_fn path (joint kernel handed to smn_predict)
    pf2 = predict.gradient_descent_mse_ensemble(lambda a, b, g: kfn(a, b, g), x, y, diag_reg=eps)
    m2, c2 = pf2(x_test=xt)
    assert relerr_norm(np.asarray(m2), rmean) < tol and relerr_norm(np.asarray(c2), rcov) < tol


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("act", ["relu", "erf"])
def test_predict_ntk_posterior(dtype, act):
    """predict_fn(get='ntk') (sample.ipynb's use of gradient_descent_mse_ensemble) against the oracle's Appendix A.5."""
    from smnngp import nt_kernels, predict
    rng = np.random.default_rng(31)
    n, t, d, eps = 260, 37, 7, 1e-2
    x = rng.standard_normal((n, d)); xt = rng.standard_normal((t, d)); y = rng.standard_normal((n, 2))
    kfn = nt_kernels.get_mlp_kernel(2, act=act, w_std=1.3, b_std=0.2, last_w_std=1.1)
    pf = predict.gradient_descent_mse_ensemble(kfn, x.astype(dtype), y.astype(dtype), diag_reg=eps)
    mean, cov = pf(x_test=xt.astype(dtype), get="ntk")
    xa = np.concatenate([x, xt])
    kj, tj = O.mlp_kernel(xa, None, 2, act, 1.3, 0.2, 1.1, ("nngp", "ntk"))
    rmean, rcov = O.predict_ntk(kj[:n, :n], kj[n:, :n], kj[n:, n:], tj[:n, :n], tj[n:, :n], y, diag_reg=eps)
    tol = 1e-7 if dtype == np.float64 else 1e-2
    assert relerr_norm(mean, rmean) < tol and relerr_norm(cov, rcov) < tol
    only_mean = pf(x_test=xt.astype(dtype), get="ntk", compute_cov=False)
    assert relerr_norm(only_mean, rmean) < tol
    with pytest.raises(NotImplementedError):
        pf(x_test=xt.astype(dtype), get="bogus")


# ----------------------------------------------------------------------------- spax facade == the reference's call sequence
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("method", ["gp", "tp"])
@pytest.mark.parametrize("network", ["mlp", "resnet"])
def test_spr_loss_and_test_nll(dtype, method, network):
    """Mirrors experiments/regression/train.py:126-142,61-74 on the syn-t generator (data.py:229-236)."""
    from smnngp import nt_kernels
    from smnngp.spax.kernels import NNGPKernel
    from smnngp.spax.likelihoods import GaussianLikelihood, StudentTLikelihood
    from smnngp.spax.models import SPR
    num = 300
    rs = np.random.RandomState(761)
    xx = np.linspace(-num / 2, num / 2, num)[:, None]
    cov = np.exp(-0.5 * (xx - xx.T) ** 2)
    yy = rs.multivariate_normal(mean=np.zeros(num), cov=cov, size=1).flatten() + rs.standard_t(df=1, size=num) * 0.8
    idx = np.random.RandomState(10).permutation(num)
    xx, yy = xx[idx], yy[idx]
    ntr = 240
    xm, xs = xx[:ntr].mean(0), xx[:ntr].std(0); ym, ys = yy[:ntr].mean(), yy[:ntr].std()
    xtr, xte = (xx[:ntr] - xm) / xs, (xx[ntr:] - xm) / xs
    ytr, yte = (yy[:ntr] - ym) / ys, (yy[ntr:] - ym) / ys
    nh, act, ws, bs, ls, eps, al, be = 2, "relu", 1.0, 0.3, 1.0, 1e-2, 2.0, 2.0
    base = nt_kernels.get_mlp_kernel if network == "mlp" else nt_kernels.get_dense_resnet_kernel

    def get_kernel_fn(w_std, b_std, last_w_std):
        return base(nh, 1, act=act, w_std=w_std, b_std=b_std, last_w_std=last_w_std)

    kernel = NNGPKernel(get_kernel_fn, ws, bs, ls)
    lik = GaussianLikelihood() if method == "gp" else StudentTLikelihood(al, be)
    model = SPR(kernel, lik, xtr.astype(dtype), ytr.astype(dtype), ym, ys, eps=eps)
    okw = dict(kernel=network, num_hiddens=nh, act=act, w_std=ws, b_std=bs, last_w_std=ls, eps=eps, method=method,
               alpha=al, beta=be)
    rl = O.spr_loss(xtr, ytr, **okw)
    rn = O.spr_test_nll(xtr, ytr, xte, yte, ym, ys, **okw)
    tol = 1e-7 if dtype == np.float64 else 1e-2
    # Student-t test_nll holds y^T (b/a K + 1e-6 I)^-1 y (likelihoods.py:60): cond ~1e9 here, so two
    # correct fp64 factorizations differ at ~1e-6; the north-star bar for fp64 is 1e-5.
    tol_nll = 1e-5 if (dtype == np.float64 and method == "tp") else tol
    assert abs(model.loss() - rl) < tol * max(1.0, abs(rl))
    assert abs(model.test_nll(xte.astype(dtype), yte.astype(dtype)) - rn) < tol_nll * max(1.0, abs(rn))
    # trainables round-trip through the softplus constraint (spax/base.py:15-25)
    assert abs(kernel.w_std.safe_value - ws) < 1e-12 and abs(model.eps.safe_value - eps) < 1e-12
    # the un-fused path (generic kernel_fn -> K + jitter -> prior_logpdf) gives the same number
    kernel2 = NNGPKernel(lambda w, b, l: (lambda a, c, get: get_kernel_fn(w, b, l)(a, c, get)), ws, bs, ls)
    model2 = SPR(kernel2, lik, xtr.astype(dtype), ytr.astype(dtype), ym, ys, eps=eps)
    assert abs(model2.loss() - rl) < tol * max(1.0, abs(rl))
    assert abs(model2.test_nll(xte.astype(dtype), yte.astype(dtype)) - rn) < tol_nll * max(1.0, abs(rn))


@pytest.mark.parametrize("n,t", [(1, 1), (2, 3), (127, 1), (128, 2), (129, 5), (255, 1), (257, 130)])
def test_loss_and_test_nll_at_the_smallest_sizes_and_the_tile_edges(n, t):
    """One data point, one test point, and the sizes on either side of the 128-row tile edge (the padded rows are an identity block
    the factorisation carries along; the test block straddles a tile at t = 130): SPR.loss and SPR.test_nll, both likelihoods,
    fp64, against the oracle."""
    from smnngp import nt_kernels
    from smnngp.spax.kernels import NNGPKernel
    from smnngp.spax.likelihoods import GaussianLikelihood, StudentTLikelihood
    from smnngp.spax.models import SPR
    rng = np.random.default_rng(1000 * n + t)
    d = 4
    x, y = rng.standard_normal((n, d)), rng.standard_normal(n)
    xt, yt = rng.standard_normal((t, d)), rng.standard_normal(t)
    hyp = dict(w_std=1.2, b_std=0.4, last_w_std=0.9, eps=5e-2, alpha=2.5, beta=1.5)
    for method in ("gp", "tp"):
        kernel = NNGPKernel(lambda w, b, l: nt_kernels.get_mlp_kernel(2, 1, act="erf", w_std=w, b_std=b, last_w_std=l),
                            hyp["w_std"], hyp["b_std"], hyp["last_w_std"])
        lik = GaussianLikelihood() if method == "gp" else StudentTLikelihood(hyp["alpha"], hyp["beta"])
        model = SPR(kernel, lik, x, y, 0.3, 1.7, eps=hyp["eps"])
        okw = dict(kernel="mlp", num_hiddens=2, act="erf", method=method, **hyp)
        rl = O.spr_loss(x, y, **okw)
        rn = O.spr_test_nll(x, y, xt, yt, 0.3, 1.7, **okw)
        assert abs(model.loss() - rl) < 1e-9 * max(1.0, abs(rl)), (method, model.loss(), rl)
        assert abs(model.test_nll(xt, yt) - rn) < 1e-6 * max(1.0, abs(rn)), (method, model.test_nll(xt, yt), rn)


def test_not_pd_gives_nan_like_the_reference(L, ctx):
    """JAX's Cholesky returns NaN on a non-PD matrix and the driver notices later (train.py:211): no exception, NaN out,
    through the fused call (info = first bad pivot), the facade's likelihood on a device matrix, and the batched call."""
    from smnngp import nt_kernels
    from smnngp.spax.kernels import NNGPKernel
    from smnngp.spax.likelihoods import GaussianLikelihood, StudentTLikelihood
    rng = np.random.default_rng(3)
    x = rng.standard_normal((40, 3)).astype(np.float32)
    y = np.linspace(-1, 1, 40).astype(np.float32)
    k = NNGPKernel(lambda w, b, l: nt_kernels.get_mlp_kernel(1, act="relu", w_std=w, b_std=b, last_w_std=l), 1., 1e-8, 1.)
    kmat = k.K(k.get_kernel_fn(), L.as_device(x, ctx))
    neg = (-1.0) * kmat + L.ScaledIdentity(40, 1e-3)        # negative definite: the first pivot already fails
    assert np.isnan(GaussianLikelihood().prior_logpdf(y, neg))
    assert np.isnan(StudentTLikelihood(2.0, 2.0).prior_logpdf(y, neg))
    xd, yd = ctx.to_device(x), ctx.to_device(y)
    lp, info = C.c_double(), C.c_int()
    ctx.call("smn_spr_loss", L.F32, L.NET_MLP, L.ACT["relu"], 1, 1.0, 1e-8, 1.0, xd.ptr, 40, 3, 3, yd.ptr, -5.0, 0.0, 1.0,
             C.byref(lp), None, None, C.byref(info))         # K - 5 I: not positive definite
    assert info.value >= 1 and np.isnan(lp.value)


# ----------------------------------------------------------------------------- "next" rows
def test_find_grid_matches_reference_formulas():
    """experiments/regression/find.py:134-199 with K0 reused across (w,b): every table entry vs the same
    formulas evaluated with the CPU oracle."""
    from scipy import stats as sst
    from scipy.special import logsumexp
    from smnngp import sweeps
    rng = np.random.default_rng(21)
    n, t, d = 61, 9, 4
    x = rng.standard_normal((n, d)); y = rng.standard_normal(n)
    xt = rng.standard_normal((t, d)); yt = rng.standard_normal(t)
    ws, bs, es, als, bes = (1.0, 1.4), (0.0, 0.3), (1e-4, 1e-2), (1.0, 2.0), (1.0, 3.0)
    got = sweeps.find_grid(x, y, xt, yt, 0.2, 1.3, network="mlp", num_hiddens=2, activation="relu", w_std_list=ws,
                           b_std_list=bs, eps_list=es, alpha_list=als, beta_list=bes)
    y_ = yt * 1.3 + 0.2
    for i, w in enumerate(ws):
        for j, b in enumerate(bs):
            kw = dict(num_hiddens=2, act="relu", w_std=w, b_std=b, last_w_std=1.0)
            kdd = O.mlp_kernel(x, None, **kw); ktd = O.mlp_kernel(xt, x, **kw); ktt = O.mlp_kernel(xt, None, **kw)
            for k, eps in enumerate(es):
                mean, cov = O.predict(kdd, ktd, ktt, y[:, None], diag_reg=eps)
                mean_ = mean.ravel() * 1.3 + 0.2
                sd = np.sqrt(np.diag(cov))
                g = -np.mean(O.normal_logpdf(y_, mean_, sd * 1.3))
                assert abs(got["gnll"][i, j, k] - g) < 1e-6 * max(1, abs(g))
                ke = kdd + eps * np.eye(n)
                quad = y @ np.linalg.solve(ke, y); logdet = np.linalg.slogdet(ke)[1]
                for ia, a in enumerate(als):
                    for ib, be in enumerate(bes):
                        sq = sst.burr12.rvs(c=a, d=be, loc=0., scale=1., size=1000, random_state=101)
                        lpd = -(n / 2) * np.log(2 * np.pi) - 0.5 * logdet - 0.5 * quad / sq - 0.5 * n * np.log(sq)
                        wgt = np.exp(lpd - lpd.max()); wb = wgt / wgt.sum()
                        lps = np.log(wb + 1e-24)[:, None] + O.normal_logpdf(y_, mean_, np.sqrt(sq[:, None]) * sd[None, :] * 1.3)
                        tn = -np.mean(logsumexp(lps, axis=0))
                        assert abs(got["tnll"][i, j, k, ia, ib] - tn) < 1e-6 * max(1, abs(tn)), (w, b, eps, a, be)
    assert got["best_gaussian"] is not None and got["best_student"] is not None


def test_batched_loss_with_the_look_ahead_schedule_is_bit_identical_too(L, ctx):
    """A batch whose problems are large enough for the look-ahead schedule (two streams, CU-masked far updates): grid.y rides
    through those launches as well."""
    from smnngp import sweeps
    n, d = 8300, 6
    rng = np.random.default_rng(8300)
    x = ctx.to_device(rng.standard_normal((n, d)).astype(np.float32))
    y = ctx.to_device(rng.standard_normal((n, 1)).astype(np.float32))
    ws, bs, eps = np.array([1.0, 1.5, 0.8]), np.array([0.1, 0.4, 0.0]), np.array([1e-1, 3e-2, 2e-1])
    lp, quad, logdet, info = C.c_double(), C.c_double(), C.c_double(), C.c_int()
    want = []
    for b in range(3):
        ctx.call("smn_spr_loss", L.F32, L.NET_MLP, L.ACT["relu"], 2, ws[b], bs[b], 1.0, x.ptr, n, d, d, y.ptr, eps[b], 0.0, 1.0,
                 C.byref(lp), C.byref(quad), C.byref(logdet), C.byref(info))
        want.append((lp.value, quad.value, logdet.value, info.value))
    got = sweeps.loss_batch(ctx, x, y, network="mlp", num_hiddens=2, activation="relu", w_std=ws, b_std=bs, last_w_std=1.0, eps=eps)
    for b in range(3):
        assert (got[0][b], got[1][b], got[2][b], got[3][b]) == want[b] and want[b][3] == 0


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_mixture_nll_matches_the_reference_formulas(L, ctx, dtype):
    """smn_mixture_nll against find.py:165-187 written out in NumPy: self-normalised importance weights of the sigma^2 draws,
    the mixture of normals per test point, with and without an importance ratio, a skipped problem, weights spanning
    hundreds of nats."""
    from scipy.special import logsumexp
    rng = np.random.default_rng(77)
    g, t, nmix, S, n = 5, 37, 3, 500, 300
    mean = rng.standard_normal((g, t)).astype(dtype)
    var = (0.2 + rng.random((g, t))).astype(dtype)
    quad = 200.0 + 300.0 * rng.random(g)
    logdet = -50.0 + 100.0 * rng.random(g)
    skip = np.array([0, 0, 1, 0, 0], dtype=np.int32)
    y = rng.standard_normal(t)
    y_mean, y_std = 0.3, 1.7
    sq = np.ascontiguousarray(0.05 + 3.0 * rng.random((nmix, S)))
    ratio = np.ascontiguousarray(0.5 + rng.random((nmix, S)))
    md, vd = ctx.to_device(mean), ctx.to_device(var)
    pd = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))    # noqa: E731
    for rt in (None, ratio):
        out = np.empty((g, nmix))
        ctx.call("smn_mixture_nll", L.dtype_code(dtype), g, t, md.ptr, vd.ptr, pd(quad), pd(logdet), skip.ctypes.data_as(C.POINTER(C.c_int)),
                 pd(y), y_mean, y_std, n, nmix, S, pd(sq), None if rt is None else pd(rt), pd(out))
        for b in range(g):
            for m in range(nmix):
                if skip[b]:
                    assert np.isnan(out[b, m])
                    continue
                q = sq[m]
                lpd = -(n / 2) * np.log(2 * np.pi) - 0.5 * logdet[b] - 0.5 * quad[b] / q - 0.5 * n * np.log(q)
                w = np.exp(lpd - lpd.max()) * (1.0 if rt is None else rt[m])
                wb = w / w.sum()
                mu = mean[b].astype(np.float64) * y_std + y_mean
                sd = np.sqrt(q[:, None]) * np.sqrt(var[b].astype(np.float64))[None, :] * y_std
                lps = np.log(wb + 1e-24)[:, None] + O.normal_logpdf(y, mu, sd)
                want = -np.mean(logsumexp(lps, axis=0))
                assert abs(out[b, m] - want) < 1e-10 * max(1.0, abs(want)), (b, m)
    with pytest.raises(L.SmnError):
        ctx.call("smn_mixture_nll", L.dtype_code(dtype), g, t, md.ptr, vd.ptr, pd(quad), pd(logdet), None, pd(y), y_mean, y_std, n, nmix, 9000,
                 pd(sq), None, pd(out))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("n,d,net,act,layers", [(245, 6, "mlp", "relu", 2), (700, 12, "mlp", "erf", 3), (1300, 20, "resnet", "relu", 2),
                                                 (2048, 8, "mlp", "relu", 4)])
def test_batched_loss_and_predict_are_bit_identical_to_the_serial_calls(L, ctx, dtype, n, d, net, act, layers):
    """smn_spr_loss_batch / smn_spr_predict_batch: G problems of one data set under their own (w_std, b_std, last_w_std,
    shift) in one sequence of launches (grid.y = G).  Every problem must reproduce the serial smn_spr_loss /
    smn_spr_predict call BIT FOR BIT (same kernels, same tiles, same order), also when the batch runs in chunks, and the
    oracle to tolerance; a problem that is not positive definite reports its own info and leaves the others alone."""
    from smnngp import sweeps
    rng = np.random.default_rng(500 + n)
    xh = rng.standard_normal((n, d)).astype(dtype)
    xh[7] = xh[3]                                              # two equal rows: singular without a shift
    yh = rng.standard_normal((n, 1)).astype(dtype)
    t = 37
    xth = rng.standard_normal((t, d)).astype(dtype)
    x, y, xt = ctx.to_device(xh), ctx.to_device(yh), ctx.to_device(xth)
    code = L.dtype_code(dtype)
    netc = L.NET_MLP if net == "mlp" else L.NET_DENSE_RESNET
    ws = np.array([1.0, 1.4, 2.0, 0.7, 1.0, 1.2, 1.0])
    bs = np.array([0.0, 0.3, 1.0, 0.1, 1e-8, 0.5, 0.2])
    lws = np.array([1.0, 1.0, 0.5, 2.0, 1.0, 1.0, 1.0])
    eps = np.array([1e-2, 1e-1, 1e-3, 1e-2, 1e-2, 3e-2, -50.0])  # the last one: a negative shift -> not positive definite
    dfs = np.array([0.0, 4.0, 0.0, 2.0, 6.0, 0.0, 0.0])
    scs = np.array([1.0, 1.5, 1.0, 0.5, 2.0, 1.0, 1.0])
    g = len(ws)
    lp, quad, logdet, info = C.c_double(), C.c_double(), C.c_double(), C.c_int()
    want = []
    for b in range(g):
        ctx.call("smn_spr_loss", code, netc, L.ACT[act], layers, ws[b], bs[b], lws[b], x.ptr, n, d, d, y.ptr, eps[b], dfs[b], scs[b],
                 C.byref(lp), C.byref(quad), C.byref(logdet), C.byref(info))
        want.append((lp.value, quad.value, logdet.value, info.value))
    assert want[-1][3] > 0 and all(w[3] == 0 for w in want[:-1])
    for budget in (1 << 40, 3 * (n + 256) ** 2 * np.dtype(dtype).itemsize):      # one pass; chunks of at most three problems
        ctx.call("smn_debug_batch_bytes", int(budget))
        got = sweeps.loss_batch(ctx, x, y, network=net, num_hiddens=layers, activation=act, w_std=ws, b_std=bs, last_w_std=lws,
                                eps=eps, df=dfs, scale=scs)
        for b in range(g):
            if want[b][3]:
                assert got[3][b] == want[b][3] and np.isnan(got[0][b]) and np.isnan(got[2][b])
            else:
                assert (got[0][b], got[1][b], got[2][b], got[3][b]) == want[b], (b, budget)
    ofn = O.mlp_kernel if net == "mlp" else O.dense_resnet_kernel
    x64, y64 = xh.astype(np.float64), yh.astype(np.float64).ravel()
    for b in (0, 1, 3):
        kdd = ofn(x64, None, layers, act, ws[b], bs[b], lws[b]) + eps[b] * np.eye(n)
        ref = O.mvn_logpdf(y64, kdd) if dfs[b] == 0 else O.mvt_logpdf(y64, scs[b] * kdd, dfs[b])
        assert abs(want[b][0] - ref) < (1e-2 if dtype == np.float32 else 1e-7) * abs(ref)
    # predict: relative ridge, mean / covariance / variance per problem
    rel = np.array([1e-2, 1e-1, 1e-3, 1e-2, 1e-2, 3e-2, 1e-2])
    mean_d, cov_d = ctx.empty((t, 1), dtype), ctx.empty((t, t), dtype)
    wantp = []
    for b in range(g):
        ctx.call("smn_spr_predict", code, netc, L.ACT[act], layers, ws[b], bs[b], lws[b], x.ptr, n, d, xt.ptr, t, d, d, y.ptr, 1,
                 rel[b], 0.0, mean_d.ptr, cov_d.ptr, t, None, None, C.byref(info))
        assert info.value == 0
        wantp.append((mean_d.numpy().copy(), cov_d.numpy().copy()))
    for budget in (1 << 40, 2 * (n + 256) ** 2 * np.dtype(dtype).itemsize):
        ctx.call("smn_debug_batch_bytes", int(budget))
        kw = dict(network=net, num_hiddens=layers, activation=act, w_std=ws, b_std=bs, last_w_std=lws, diag_reg=rel)
        mean, var, infop = sweeps.predict_batch(ctx, x, y, xt, **kw)
        mean2, cov, _ = sweeps.predict_batch(ctx, x, y, xt, full_cov=True, **kw)
        assert not infop.any() and np.array_equal(mean, mean2)
        for b in range(g):
            assert np.array_equal(mean[b], wantp[b][0]) and np.array_equal(cov[b], wantp[b][1])
            assert np.array_equal(var[b], np.diag(wantp[b][1]))
    ctx.call("smn_debug_batch_bytes", 48 << 30)


def test_finite_difference_train_step():
    """regression/train.py:61-67 (GradValues + Adam) on the facade: the FD gradient equals the FD gradient of the
    oracle loss, and a few Adam steps lower the loss."""
    from smnngp import nt_kernels, train
    from smnngp.spax.kernels import NNGPKernel
    from smnngp.spax.likelihoods import StudentTLikelihood
    from smnngp.spax.models import SPR
    rng = np.random.default_rng(31)
    n, d = 80, 3
    x = rng.standard_normal((n, d)); y = np.sin(x[:, 0]) + 0.1 * rng.standard_normal(n)
    kernel = NNGPKernel(lambda w, b, l: nt_kernels.get_mlp_kernel(2, act="relu", w_std=w, b_std=b, last_w_std=l), 1.0, 0.5, 1.0)
    lik = StudentTLikelihood(2.0, 2.0)
    model = SPR(kernel, lik, x, y, 0.0, 1.0, eps=1e-2)
    tv = train.train_vars(model)
    assert sorted(k.split(".")[-1] for k in tv) == ["a", "b", "b_std", "eps", "last_w_std", "w_std"]
    val, grads = train.value_and_grad_fd(model.loss, tv, h=1e-5)

    def oracle_loss(raw):
        p = {k.split(".")[-1]: float(O.softplus(v)) for k, v in raw.items()}
        return O.spr_loss(x, y, num_hiddens=2, act="relu", w_std=p["w_std"], b_std=p["b_std"], last_w_std=p["last_w_std"],
                          eps=p["eps"], method="tp", alpha=p["a"], beta=p["b"])
    raw0 = {k: float(v.value) for k, v in tv.items()}
    assert abs(val - oracle_loss(raw0)) < 1e-9
    for k in tv:
        hh = 1e-5 * max(1.0, abs(raw0[k]))
        up = dict(raw0); up[k] += hh; dn = dict(raw0); dn[k] -= hh
        ref = (oracle_loss(up) - oracle_loss(dn)) / (2 * hh)
        assert abs(grads[k] - ref) < 1e-5 * max(1.0, abs(ref)), (k, grads[k], ref)
    step = train.build_train_step(model)
    losses = [step(1e-2) for _ in range(12)]
    assert model.loss() < losses[0] - 1e-4 and all(np.isfinite(losses))


# ----------------------------------------------------------------------------- reference checkpoints
@pytest.mark.parametrize("method,network", [("tp", None), ("gp", "resnet")])
def test_restore_spr_from_a_reference_style_run_directory(tmp_path, method, network):
    """A run directory laid out like the reference's (objax names, RAW softplus-inverse tensors, pickled args,
    eps stored as diag_reg, last_w_std only in the args) is rebuilt the way regression/test.py:89-130 does, and
    evaluates to the oracle's numbers at the constrained hyper-parameters; saving our own model round-trips."""
    import os
    from smnngp import checkpoint as CK
    rng = np.random.default_rng(21)
    n, t, d = 180, 25, 5
    x, xt = rng.standard_normal((n, d)), rng.standard_normal((t, d))
    y, yt = rng.standard_normal(n), rng.standard_normal(t)
    hyp = dict(w_std=1.3, b_std=0.4, eps=2e-2, a=1.7, b=2.4)
    raw = {k: np.array(O.softplus_inverse(v), np.float32) for k, v in hyp.items()}
    names = ["(SPR).kernel(NNGPKernel).w_std", "(SPR).kernel(NNGPKernel).b_std", "(SPR).diag_reg"]
    vals = [raw["w_std"], raw["b_std"], raw["eps"]]
    if method == "tp":
        names += ["(SPR).likelihood(StudentTLikelihood).a", "(SPR).likelihood(StudentTLikelihood).b"]
        vals += [raw["a"], raw["b"]]
    d1 = str(tmp_path / "ref_run"); os.makedirs(d1)
    lw_raw = 0.8                                                            # test.py assigns the arg as the raw value
    np.savez(os.path.join(d1, "012.npz"), names=np.array(names), **{str(i): v for i, v in enumerate(vals)})
    np.save(os.path.join(d1, "meta.npy"), dict(args=dict(method=method, network=network, num_hiddens=2, activation="erf",
                                                         data_name="syn", last_w_std=lw_raw)))
    model, ctx_args = CK.restore_spr(d1, x, y, 0.3, 1.9, dtype=np.float64)
    assert ctx_args["activation"] == "erf"
    okw = dict(kernel=network or "mlp", num_hiddens=2, act="erf", w_std=float(O.softplus(raw["w_std"])),
               b_std=float(O.softplus(raw["b_std"])), last_w_std=float(O.softplus(lw_raw)), eps=float(O.softplus(raw["eps"])),
               method=method, alpha=float(O.softplus(raw["a"])), beta=float(O.softplus(raw["b"])))
    rl = O.spr_loss(x, y, **okw)
    rn = O.spr_test_nll(x, y, xt, yt, 0.3, 1.9, **okw)
    assert abs(model.loss() - rl) < 1e-7 * max(1.0, abs(rl))
    assert abs(model.test_nll(xt, yt) - rn) < 1e-5 * max(1.0, abs(rn))
    # our own writer -> our own reader: identical raw values, identical loss
    d2 = str(tmp_path / "own_run")
    ck = CK.Checkpointer(d2)
    assert ck.step(1, model.loss(), model.vars())
    CK.save_meta(d2, ctx_args)
    model2, _ = CK.restore_spr(d2, x, y, 0.3, 1.9, dtype=np.float64)
    for k, v in model.vars().items():
        assert float(model2.vars()[k].value) == float(v.value)
    assert model2.loss() == model.loss()


# ----------------------------------------------------------------------------- analytic LML gradients
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("method", ["gp", "tp"])
@pytest.mark.parametrize("network,act,nh", [("mlp", "relu", 2), ("mlp", "erf", 3), ("resnet", "relu", 2), ("resnet", "erf", 1)])
def test_analytic_loss_gradient_matches_finite_differences_of_the_oracle(dtype, method, network, act, nh):
    """SPR.loss_and_grad (one augmented factorisation + one contraction pass) against central differences of the
    fp64 oracle loss, for every trainable of regression/train.py (w_std, b_std, last_w_std, eps, a, b)."""
    from smnngp import nt_kernels
    from smnngp.spax.kernels import NNGPKernel
    from smnngp.spax.likelihoods import GaussianLikelihood, StudentTLikelihood
    from smnngp.spax.models import SPR
    rng = np.random.default_rng(17)
    n, d = 150, 6
    x = rng.standard_normal((n, d))
    y = np.sin(x[:, 0]) + 0.3 * rng.standard_normal(n)
    hyp = dict(w_std=1.3, b_std=0.4, last_w_std=0.9, eps=5e-2, alpha=1.7, beta=2.4)
    base = nt_kernels.get_mlp_kernel if network == "mlp" else nt_kernels.get_dense_resnet_kernel

    def get_kernel_fn(w_std, b_std, last_w_std):
        return base(nh, 1, act=act, w_std=w_std, b_std=b_std, last_w_std=last_w_std)

    kernel = NNGPKernel(get_kernel_fn, hyp["w_std"], hyp["b_std"], hyp["last_w_std"])
    lik = GaussianLikelihood() if method == "gp" else StudentTLikelihood(hyp["alpha"], hyp["beta"])
    model = SPR(kernel, lik, x.astype(dtype), y.astype(dtype), 0.0, 1.0, eps=hyp["eps"])
    loss, grads = model.loss_and_grad()
    okw = dict(kernel=network, num_hiddens=nh, act=act, method=method, **hyp)
    keys = ("w_std", "b_std", "last_w_std", "eps") + (("alpha", "beta") if method == "tp" else ())
    ref = O.spr_loss_grad_fd(x, y, keys=keys, **okw)
    rl = O.spr_loss(x, y, **okw)
    tol = 2e-6 if dtype == np.float64 else 1e-2
    assert abs(loss - rl) < (1e-9 if dtype == np.float64 else 1e-3) * max(1.0, abs(rl))
    assert abs(loss - model.loss()) < (1e-10 if dtype == np.float64 else 1e-4) * max(1.0, abs(rl))
    vmap = {"w_std": kernel.w_std, "b_std": kernel.b_std, "last_w_std": kernel.last_w_std, "eps": model.eps}
    if method == "tp":
        vmap.update(alpha=lik.a, beta=lik.b)
    names = {id(v): k for k, v in model.vars().items()}
    assert set(grads) == set(model.vars())
    scale = max(abs(v) for v in ref.values())
    for k in keys:
        var = vmap[k]
        got = grads[names[id(var)]] / float(var.constraint.grad(var.value))     # undo the softplus chain rule
        assert abs(got - ref[k]) < tol * max(scale, abs(ref[k])), (k, got, ref[k])


@pytest.mark.parametrize("dtype,n", [(np.float64, 700), (np.float64, 1280), (np.float32, 1100)])
def test_analytic_gradient_at_sizes_that_skip_identity_tiles(dtype, n):
    """The gradient's factorisation carries an N x N identity block whose structurally zero tiles are skipped
    (several tile rows here, with and without a ragged last tile); values must not change."""
    from smnngp import nt_kernels
    from smnngp.spax.kernels import NNGPKernel
    from smnngp.spax.likelihoods import StudentTLikelihood
    from smnngp.spax.models import SPR
    rng = np.random.default_rng(n)
    d = 5
    x = rng.standard_normal((n, d)); y = np.cos(x[:, 0]) + 0.3 * rng.standard_normal(n)
    hyp = dict(w_std=1.1, b_std=0.3, last_w_std=1.2, eps=5e-2, alpha=2.0, beta=1.5)
    kernel = NNGPKernel(lambda w, b, l: nt_kernels.get_mlp_kernel(2, 1, act="relu", w_std=w, b_std=b, last_w_std=l),
                        hyp["w_std"], hyp["b_std"], hyp["last_w_std"])
    lik = StudentTLikelihood(hyp["alpha"], hyp["beta"])
    model = SPR(kernel, lik, x.astype(dtype), y.astype(dtype), 0.0, 1.0, eps=hyp["eps"])
    loss, grads = model.loss_and_grad()
    okw = dict(kernel="mlp", num_hiddens=2, act="relu", method="tp", **hyp)
    ref = O.spr_loss_grad_fd(x, y, **okw)
    assert abs(loss - O.spr_loss(x, y, **okw)) < (1e-9 if dtype == np.float64 else 1e-3)
    vmap = {"w_std": kernel.w_std, "b_std": kernel.b_std, "last_w_std": kernel.last_w_std, "eps": model.eps,
            "alpha": lik.a, "beta": lik.b}
    names = {id(v): k for k, v in model.vars().items()}
    scale = max(abs(v) for v in ref.values())
    tol = 5e-6 if dtype == np.float64 else 2e-2
    for k, var in vmap.items():
        got = grads[names[id(var)]] / float(var.constraint.grad(var.value))
        assert abs(got - ref[k]) < tol * max(scale, abs(ref[k])), (k, got, ref[k])


@pytest.mark.parametrize("n", [4096, 8192])
def test_gradient_factorisation_launches_only_the_live_tiles_of_the_identity_block(n):
    """Both routes of the gradient's factorisation (heads.hip factor_with_identity): below 8192 rows the joint matrix
    [[K~, .], [I, 0], [y^T, 0, 0]], whose Schur complement is -K~^-1; from 8192 on the rectangle [[K~], [I], [y^T]] factored
    without its Schur block, then -L^-T L^-1 as one launch.  Either way ~N^3 update flops (chol N^3/3 + L^-T N^3/3 +
    L^-T L^-1 N^3/3) when the structural zeros of the identity rows are left out of every launch (a dense 2N x 2N
    factorisation would execute ~2.3 N^3).  The library counts the flops of the tiles it launches (smn_profile_flops), so
    this pins the launch shapes, not a timing."""
    from smnngp import nt_kernels, _lib as L
    from smnngp.spax.kernels import NNGPKernel
    from smnngp.spax.likelihoods import GaussianLikelihood
    from smnngp.spax.models import SPR
    d = 64
    rng = np.random.default_rng(5)
    x = rng.standard_normal((n, d)).astype(np.float32); y = rng.standard_normal(n).astype(np.float32)
    kernel = NNGPKernel(lambda w, b, l: nt_kernels.get_mlp_kernel(2, 1, act="relu", w_std=w, b_std=b, last_w_std=l), 1.0, 0.3, 1.0)
    model = SPR(kernel, GaussianLikelihood(), x, y, 0.0, 1.0, eps=1e-2)
    ctx = model.x_data.ctx
    model.loss_and_grad()
    ctx.call("smn_profile_enable", 0)          # resets the counters
    loss, grads = model.loss_and_grad()
    fl = 0.0
    for cat in (4, 5):
        v = C.c_double()
        ctx.call("smn_profile_flops", cat, C.byref(v))
        fl += v.value
    assert np.isfinite(loss) and all(np.isfinite(g) for g in grads.values())
    assert 0.9 * n ** 3 < fl < 1.45 * n ** 3, fl / n ** 3      # whole 128 x 128 tiles: a little above N^3 at this size
    # at these sizes the factorisation runs with the look-ahead (two streams, CU mask) AND the launch split: its loss must
    # be SPR.loss's, and its w_std gradient the central difference of SPR.loss (the un-hinted factorisation)
    assert abs(loss - model.loss()) < 2e-5 * abs(loss)
    name = [k for k in grads if k.endswith("w_std") and "last" not in k][0]
    raw, h = kernel.w_std.value.copy(), 2e-2
    kernel.w_std.value = raw + h
    lp = model.loss()
    kernel.w_std.value = raw - h
    lm = model.loss()
    kernel.w_std.value = raw
    fd = (lp - lm) / (2 * h)
    assert abs(grads[name] - fd) < 0.03 * abs(fd) + 2e-4, (grads[name], fd)
    if n == 8192:   # the rectangle route against the oracle's central differences (two of the six; fp64 on the host, ~10 s)
        ref = O.spr_loss_grad_fd(x.astype(np.float64), y.astype(np.float64), keys=("w_std", "eps"), kernel="mlp", num_hiddens=2,
                                 act="relu", method="gp", w_std=1.0, b_std=0.3, last_w_std=1.0, eps=1e-2)
        names = {id(v): k for k, v in model.vars().items()}
        scale = max(abs(v) for v in ref.values())
        for key, var in (("w_std", kernel.w_std), ("eps", model.eps)):
            got = grads[names[id(var)]] / float(var.constraint.grad(var.value))
            assert abs(got - ref[key]) < 2e-2 * max(scale, abs(ref[key])), (key, got, ref[key])


def test_device_matrix_hands_over_its_diagonal_without_a_full_download():
    from smnngp import _lib as L
    ctx = L.default_context()
    rng = np.random.default_rng(2)
    for dtype, shape in ((np.float32, (37, 37)), (np.float64, (20, 33)), (np.float32, (1, 1))):
        a = rng.standard_normal(shape).astype(dtype)
        d = ctx.to_device(a)
        assert np.array_equal(d.diagonal(), np.diagonal(a))
        v = d * 2.5
        np.testing.assert_allclose(v.diagonal(), 2.5 * np.diagonal(a), rtol=1e-6)


def test_analytic_train_step_descends_and_agrees_with_fd_step():
    """regression/train.py:61-67 with the analytic gradient: same first Adam update as finite differences."""
    from smnngp import nt_kernels, train
    from smnngp.spax.kernels import NNGPKernel
    from smnngp.spax.likelihoods import StudentTLikelihood
    from smnngp.spax.models import SPR
    rng = np.random.default_rng(4)
    n, d = 120, 4
    x = rng.standard_normal((n, d)); y = np.tanh(x[:, 0] - x[:, 1]) + 0.2 * rng.standard_normal(n)

    def make():
        kernel = NNGPKernel(lambda w, b, l: nt_kernels.get_mlp_kernel(2, 1, act="relu", w_std=w, b_std=b, last_w_std=l),
                            1.0, 1.0, 1.0)
        return SPR(kernel, StudentTLikelihood(2.0, 2.0), x, y, 0.0, 1.0, eps=1e-1)

    ma, mf = make(), make()
    sa = train.build_train_step(ma, method="analytic")
    sf = train.build_train_step(mf, method="fd", h=1e-5)
    la, lf = sa(0.05), sf(0.05)
    assert abs(la - lf) < 1e-10
    for k, v in ma.vars().items():
        assert abs(float(v.value) - float(mf.vars()[k].value)) < 1e-4
    losses = [la] + [sa(0.05) for _ in range(15)]
    assert losses[-1] < losses[0]
    with pytest.raises(ValueError):
        train.build_train_step(ma, method="bogus")


# ----------------------------------------------------------------------------- host-side contracts (round-2 advice)
def test_context_driven_from_a_thread_that_did_not_create_it(L, ctx):
    """The current HIP device belongs to the calling THREAD: every entry point makes the context's device current
    itself (SMN_ENTER), so a context made in the main thread works from a worker (sweeps.py does exactly this)."""
    import threading
    rng = np.random.default_rng(70)
    x = rng.standard_normal((200, 12)); y = rng.standard_normal(200)
    want = O.spr_loss(x, y, num_hiddens=2, act="relu", w_std=1.2, b_std=0.3, last_w_std=1.0, eps=1e-3, method="gp")
    c2 = L.Context(ctx.device)              # created here ...
    got, err = [], []

    def work():                             # ... driven there (allocations, attributes, launches, events)
        try:
            xd = c2.to_device(x); yd = c2.to_device(y)
            lp, quad, logdet, info = C.c_double(), C.c_double(), C.c_double(), C.c_int()
            c2.call("smn_profile_enable", 1)
            c2.call("smn_spr_loss", L.F64, L.NET_MLP, L.ACT["relu"], 2, 1.2, 0.3, 1.0, xd.ptr, 200, 12, 12, yd.ptr, 1e-3,
                    0.0, 1.0, C.byref(lp), C.byref(quad), C.byref(logdet), C.byref(info))
            c2.call("smn_profile_enable", 0)
            got.append(-lp.value / 200)
            del xd, yd
        except Exception as e:              # noqa: BLE001
            err.append(e)
    t = threading.Thread(target=work); t.start(); t.join()
    assert not err, err
    assert abs(got[0] - want) < 1e-9 * max(1.0, abs(want))
    c2.close()


def test_two_contexts_on_two_devices_in_one_thread(L):
    n = C.c_int(0)
    L._lib.smn_device_count(C.byref(n))
    if n.value < 2:
        pytest.skip("one visible device")
    rng = np.random.default_rng(71)
    x = rng.standard_normal((150, 8)); y = rng.standard_normal(150)
    want = O.spr_loss(x, y, num_hiddens=1, act="erf", w_std=1.0, b_std=0.2, last_w_std=1.0, eps=1e-3, method="gp")
    ctxs = [L.Context(0), L.Context(1)]
    for c in (ctxs[1], ctxs[0], ctxs[1]):   # alternate: the guard must switch every call
        xd = c.to_device(x); yd = c.to_device(y)
        lp, quad, logdet, info = C.c_double(), C.c_double(), C.c_double(), C.c_int()
        c.call("smn_spr_loss", L.F64, L.NET_MLP, L.ACT["erf"], 1, 1.0, 0.2, 1.0, xd.ptr, 150, 8, 8, yd.ptr, 1e-3, 0.0, 1.0,
               C.byref(lp), C.byref(quad), C.byref(logdet), C.byref(info))
        assert abs(-lp.value / 150 - want) < 1e-9 * max(1.0, abs(want))
        del xd, yd


def test_as_device_checks_context_dtype_and_views(L, ctx):
    from smnngp import nt_kernels, predict
    rng = np.random.default_rng(72)
    a = ctx.to_device(rng.standard_normal((6, 6)))
    assert L.as_device(a, ctx) is a
    v = 2.0 * a + L.ScaledIdentity(6, 0.5)                 # a lazy view: the raw pointer still holds A
    m = L.as_device(v, ctx)
    assert m is not v and m.scale == 1.0 and m.shift == 0.0
    assert np.allclose(m.raw_numpy(), 2.0 * a.raw_numpy() + 0.5 * np.eye(6))
    f = L.as_device(a, ctx, dtype=np.float32)              # converted, not reinterpreted
    assert f.dtype == np.float32 and np.allclose(f.raw_numpy(), a.raw_numpy().astype(np.float32))
    other = L.Context(ctx.device)
    with pytest.raises(ValueError):
        L.as_device(a, other)
    # an f64 DeviceArray x_test with f32 training data is converted (it used to be read as f32)
    x = rng.standard_normal((40, 5)).astype(np.float32); y = rng.standard_normal(40).astype(np.float32)
    xt64 = rng.standard_normal((7, 5))
    kfn = nt_kernels.get_mlp_kernel(2, act="relu", w_std=1.1, b_std=0.2)
    pf = predict.gradient_descent_mse_ensemble(kfn, x, y[:, None], diag_reg=1e-2)
    m_dev, c_dev = pf(x_test=ctx.to_device(xt64), get="nngp", compute_cov=True)
    m_np, c_np = pf(x_test=xt64.astype(np.float32), get="nngp", compute_cov=True)
    assert np.allclose(np.asarray(m_dev), np.asarray(m_np), rtol=1e-5, atol=1e-6)
    assert np.allclose(np.asarray(c_dev), np.asarray(c_np), rtol=1e-5, atol=1e-6)
    other.close()


def test_too_many_output_columns_is_rejected_before_anything_runs(L, ctx):
    rng = np.random.default_rng(73)
    x = ctx.to_device(rng.standard_normal((64, 4))); y = ctx.to_device(rng.standard_normal((64, 49)))
    mean = ctx.empty((1, 49), np.float64)
    with pytest.raises(L.SmnError) as e:
        ctx.call("smn_spr_predict", L.F64, L.NET_MLP, L.ACT["relu"], 1, 1.0, 0.1, 1.0, x.ptr, 64, 4, x.ptr, 1, 4, 4, y.ptr, 49,
                 1e-3, 0.0, mean.ptr, None, 1, None, None, None)
    assert e.value.code == L.ENOTSUP


# ----------------------------------------------------------------------------- column-first exchange (cyclic layout)
from _played import play_ranks as _play_ranks  # noqa: E402  (tests/_played.py: P ranks played on one GPU)


@pytest.mark.parametrize("n,d,world,cols,dtype", [(1000, 24, 3, [0, 3, 6, 8], np.float32), (2048, 16, 2, None, np.float32),
                                                    (700, 10, 1, [0, 2, 6], np.float32), (1536, 8, 4, [0, 12], np.float32),
                                                    (1100, 12, 8, [0, 8, 9], np.float64), (3000, 8, 5, None, np.float32),
                                                    (2000, 8, 4, [0, 1, 3, 6, 7, 16], np.float32), (1300, 6, 8, [0, 2, 4, 6, 8, 11], np.float64)])
def test_column_first_pieces_assemble_the_same_kernel(L, ctx, n, d, world, cols, dtype):
    """`world` ranks played on one GPU in the cyclic column-first layout: the scattered pieces (smn_shard_scatter_cols into a
    matrix of the caller's) equal the one-launch kernel bit for bit on the lower triangle by 128-column tiles, NNGP and
    NTK, and everything the scatter does not own stays untouched (NaN-poisoned target and staging)."""
    from smnngp import sharding as S
    rng = np.random.default_rng(90 + n)
    x = ctx.to_device(rng.standard_normal((n, d)).astype(dtype))
    code = L.dtype_code(dtype)
    spec = (L.NET_MLP, L.ACT["relu"], 2, 1.3, 0.2, 1.0)
    cols = cols or S.default_col_pieces(n, world)
    stage, stage_t = _play_ranks(L, ctx, spec, x, n, d, world, cols, dtype)
    k = ctx.to_device(np.full((n, n), np.nan, dtype))
    kt = ctx.to_device(np.full((n, n), np.nan, dtype))
    ca = S.cols_array(cols)
    for g in reversed(range(len(cols) - 1)):                  # any order: the pieces are disjoint
        ctx.call("smn_shard_scatter_cols", code, stage.ptr, n, world, len(cols) - 1, ca, g, k.ptr, n)
        ctx.call("smn_shard_scatter_cols", code, stage_t.ptr, n, world, len(cols) - 1, ca, g, kt.ptr, n)
    ctx.call("smn_shard_wait")
    ref = ctx.empty((n, n), dtype)
    ref_t = ctx.empty((n, n), dtype)
    ctx.call("smn_kernel_mlp", code, *spec, x.ptr, n, d, None, 0, 0, d, L.GET_NNGP | L.GET_NTK, L.FILL_LOWER, ref.ptr, ref_t.ptr, n)
    rr, cc = np.indices((n, n))
    own = cc < np.minimum(n, (rr // 128 + 1) * 128)          # the lower triangle by 128-column tiles
    for got, want in ((k.numpy(), ref.numpy()), (kt.numpy(), ref_t.numpy())):
        assert np.array_equal(got[own], want[own])
        assert np.isnan(got[~own]).all()
    with pytest.raises(L.SmnError):                           # boundaries must ascend and span every tile column
        ctx.call("smn_shard_scatter_cols", code, stage.ptr, n, world, 2, S.cols_array([0, S.tile_rows(n), S.tile_rows(n)]), 0, k.ptr, n)


@pytest.mark.parametrize("world,order,delay", [(2, "forward", None), (4, "reverse", (2, 20000)), (8, "forward", (2, 30000)),
                                                (8, "odd-first", (1, 10000)), (3, "reverse", None)])
def test_column_first_lml_is_bit_identical_whatever_the_arrival_order(L, world, order, delay):
    """P ranks played on one GPU, pieces scattered into the factorisation workspace in a forced order and behind a held
    stream (smn_debug_delay), with the look-ahead schedule on (SMN_CHAIN_MIN_N lowered): the factorisation waits for each
    piece where it first touches its columns and the log-pdf, quadratic form and logdet equal the fused single-GPU
    smn_spr_loss bit for bit."""
    import os
    from smnngp import sharding as S
    old = os.environ.get("SMN_CHAIN_MIN_N")
    os.environ["SMN_CHAIN_MIN_N"] = "1024"
    try:
        c2 = L.Context(0)
    finally:
        if old is None:
            del os.environ["SMN_CHAIN_MIN_N"]
        else:
            os.environ["SMN_CHAIN_MIN_N"] = old
    n, d, eps = 4000, 16, 1e-2
    rng = np.random.default_rng(17)
    x = c2.to_device(rng.standard_normal((n, d)).astype(np.float32)); y = c2.to_device(rng.standard_normal(n).astype(np.float32))
    spec = (L.NET_MLP, L.ACT["relu"], 3, 1.2, 0.25, 1.0)
    lp, quad, logdet, info = C.c_double(), C.c_double(), C.c_double(), C.c_int()
    c2.call("smn_spr_loss", L.F32, *spec, x.ptr, n, d, d, y.ptr, eps, 4.0, 1.5, C.byref(lp), C.byref(quad), C.byref(logdet), C.byref(info))
    want = (lp.value, quad.value, logdet.value, info.value)
    assert want[3] == 0
    cols = S.default_col_pieces(n, world)
    assert len(cols) - 1 >= 3
    stage, _ = _play_ranks(L, c2, spec, x, n, d, world, cols, with_ntk=False)
    ca = S.cols_array(cols)
    pieces = list(range(len(cols) - 1))
    if order == "reverse":
        pieces.reverse()
    elif order == "odd-first":
        pieces = pieces[1::2] + pieces[0::2]
    for rep in range(2):                                       # twice: the events of the pieces are reused
        c2.call("smn_shard_begin", L.F32, n, eps)
        for i, g in enumerate(pieces):
            if delay is not None and i in (0, len(pieces) - 1):
                c2.call("smn_debug_delay", delay[0], delay[1])
            c2.call("smn_shard_scatter_cols", L.F32, stage.ptr, n, world, len(cols) - 1, ca, g, None, 0)
        c2.call("smn_lml_from_shards", L.F32, n, y.ptr, 4.0, 1.5, C.byref(lp), C.byref(quad), C.byref(logdet), C.byref(info))
        assert (lp.value, quad.value, logdet.value, info.value) == want
    # a column range that never went out is refused, not factored
    c2.call("smn_shard_begin", L.F32, n, eps)
    for g in pieces[1:]:
        c2.call("smn_shard_scatter_cols", L.F32, stage.ptr, n, world, len(cols) - 1, ca, g, None, 0)
    with pytest.raises(L.SmnError):
        c2.call("smn_lml_from_shards", L.F32, n, y.ptr, 0.0, 1.0, C.byref(lp), C.byref(quad), C.byref(logdet), C.byref(info))
    c2.call("smn_shard_begin", L.F32, n, eps)                  # the abandoned pipeline is joined and forgotten
    c2.synchronize()
    del x, y, stage
    c2.close()


def test_column_first_route_on_a_one_rank_communicator(L, ctx):
    """smn_shard_begin -> one build -> smn_shard_exchange_cols per piece (RCCL, one rank, communication + scatter streams)
    -> smn_lml_from_shards equals the fused smn_spr_loss bit for bit; a world the communicator does not have is refused."""
    from smnngp import sharding as S
    n, d = 1500, 20
    rng = np.random.default_rng(91)
    x = ctx.to_device(rng.standard_normal((n, d)).astype(np.float32)); y = ctx.to_device(rng.standard_normal(n).astype(np.float32))
    spec = (L.NET_MLP, L.ACT["erf"], 3, 1.4, 0.3, 0.9)
    lp, quad, logdet, info = C.c_double(), C.c_double(), C.c_double(), C.c_int()
    ctx.call("smn_spr_loss", L.F32, *spec, x.ptr, n, d, d, y.ptr, 1e-2, 0.0, 1.0, C.byref(lp), C.byref(quad),
             C.byref(logdet), C.byref(info))
    assert info.value == 0
    c2 = L.Context(ctx.device)
    uid = C.create_string_buffer(128)
    assert L._lib.smn_comm_unique_id(uid) == 0
    c2.call("smn_comm_init", 1, 0, uid)
    x2 = c2.to_device(x.numpy()); y2 = c2.to_device(y.numpy())
    t_all = S.tile_rows(n)
    for cols in ([0, t_all], [0, 4, t_all], list(range(0, t_all, 2)) + [t_all]):
        elems = S.col_layout(n, 1, cols)["elems"]
        mine = c2.empty((elems,), np.float32); stage = c2.empty((elems,), np.float32)
        phases = []
        got = S.lml_sharded_cols(S.DeviceBackend(c2), L.F32, spec, x2.ptr, n, d, d, y2.ptr, 0, 1, mine.ptr, stage.ptr,
                                 1e-2, cols=cols, progress=phases.append)
        assert got[3] == 0 and got[0] == lp.value and got[2] == logdet.value and got[1] == quad.value
        assert phases[0] == "begin" and phases[-1] == "factor"
        del mine, stage
    cols = [0, 4, 8, t_all]
    elems = S.col_layout(n, 1, cols)["elems"]
    mine = c2.empty((elems,), np.float32); stage = c2.empty((elems,), np.float32)
    with pytest.raises(RuntimeError):
        S.lml_sharded_cols(S.DeviceBackend(c2), L.F32, spec, x2.ptr, n, d, d, y2.ptr, 0, 2, mine.ptr, stage.ptr, 1e-2)
    # BASELINE config 5: NNGP + NTK through the same exchange (smn_shard_exchange_cols_to)
    mine_t = c2.empty((elems,), np.float32); stage_t = c2.empty((elems,), np.float32)
    tk = c2.to_device(np.full((n, n), np.nan, np.float32))
    got = S.lml_sharded_cols(S.DeviceBackend(c2), L.F32, spec, x2.ptr, n, d, d, y2.ptr, 0, 1, mine.ptr, stage.ptr,
                             1e-2, cols=cols, ntk=(mine_t.ptr, stage_t.ptr, tk.ptr, n))
    # (the joint NNGP + NTK build takes the generic maps, the NNGP-only smn_spr_loss above the f32 fast ones: rounding apart)
    assert got[3] == 0 and abs(got[0] - lp.value) < 1e-5 * abs(lp.value)
    ref = ctx.empty((n, n), np.float32); ref_t = ctx.empty((n, n), np.float32)
    ctx.call("smn_kernel_mlp", L.F32, *spec, x.ptr, n, d, None, 0, 0, d, L.GET_NNGP | L.GET_NTK, L.FILL_LOWER, ref.ptr, ref_t.ptr, n)
    rr, cc = np.indices((n, n))
    own = cc < np.minimum(n, (rr // 128 + 1) * 128)
    assert np.array_equal(tk.numpy()[own], ref_t.numpy()[own]) and np.isnan(tk.numpy()[~own]).all()
    ca = S.cols_array(cols)
    with pytest.raises(L.SmnError) as e:                      # a world the communicator does not have
        c2.call("smn_shard_exchange_cols_to", L.F32, mine_t.ptr, stage_t.ptr, n, 2, len(cols) - 1, S.cols_array([0, 4, 8, t_all]), 0, tk.ptr, n)
    assert e.value.code == L.ECOMM
    with pytest.raises(L.SmnError):                           # the exchange without smn_shard_begin
        c2.call("smn_shard_exchange_cols", L.F32, mine.ptr, stage.ptr, n, 1, len(cols) - 1, ca, 0)
    with pytest.raises(RuntimeError):       # the monolithic paired route checks the communicator too
        S.build_lower_sharded(c2, L.F32, 4, *spec, x2.ptr, n, d, d, 0, 2, stage.ptr, None, 0)
    c2.call("smn_comm_destroy")
    nr, rk = C.c_int(-1), C.c_int(-1)
    c2.call("smn_comm_info", C.byref(nr), C.byref(rk))
    assert (nr.value, rk.value) == (1, 0)
    del x2, y2, mine, stage, mine_t, stage_t, tk
    c2.close()
