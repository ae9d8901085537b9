"""Golden-vector tests.  CPU: the oracle still reproduces the committed vectors (guards oracle drift) and
the closed-form known answers.  GPU: the HIP path against the same vectors."""
import os

import numpy as np
import pytest

from oracle import nngp_oracle as O
from _tol import relerr  # norm-wise AND element-wise: tests/_tol.py

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nngp_golden.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def _case(name):
    # mlp_n33_d6_relu_L2_w1.4_b0.3_lw0.8
    parts = name.split("_")
    return dict(n=int(parts[1][1:]), d=int(parts[2][1:]), act=parts[3], L=int(parts[4][1:]), w=float(parts[5][1:]),
                b=float(parts[6][1:]), lw=float(parts[7][2:]))


def _inputs(c):
    rng = np.random.default_rng(1000 * c["n"] + c["d"])
    return rng.standard_normal((c["n"], c["d"])), rng.standard_normal((5, c["d"]))


def test_known_answers(gold):
    q1, q2, k = gold["kat_inputs"]
    kn, _, _, th = O.relu_map(np.array([[k]]), np.array([q1]), np.array([q2]), np.ones((1, 1)))
    assert abs(kn[0, 0] - gold["kat_relu"][0]) < 1e-15 and abs(th[0, 0] - gold["kat_relu"][1]) < 1e-15
    kn, _, _, th = O.erf_map(np.array([[k]]), np.array([q1]), np.array([q2]), np.ones((1, 1)))
    assert abs(kn[0, 0] - gold["kat_erf"][0]) < 1e-15 and abs(th[0, 0] - gold["kat_erf"][1]) < 1e-15


def test_oracle_reproduces_golden_kernels(gold):
    names = [str(n) for n in gold["names"]]
    assert len(names) >= 40
    for name in names:
        c = _case(name)
        x, x2 = _inputs(c)
        k, t = O.mlp_kernel(x, None, c["L"], c["act"], c["w"], c["b"], c["lw"], ("nngp", "ntk"))
        assert np.allclose(k, gold[name + "/k"], rtol=1e-12, atol=1e-14), name
        if name + "/t" in gold:
            assert np.allclose(t, gold[name + "/t"], rtol=1e-12, atol=1e-14), name
        kc = O.mlp_kernel(x2, x, c["L"], c["act"], c["w"], c["b"], c["lw"])
        assert np.allclose(kc, gold[name + "/kc"], rtol=1e-12, atol=1e-14), name


def test_oracle_reproduces_golden_heads(gold):
    x, y, xt, yt = gold["heads/x"], gold["heads/y"], gold["heads/xt"], gold["heads/yt"]
    for eps, al, be, tp, res, loss, nll in gold["heads/table"]:
        kw = dict(kernel="resnet" if res else "mlp", num_hiddens=2, act="relu", w_std=1.4, b_std=0.3, last_w_std=1.0,
                  eps=eps, method="tp" if tp else "gp", alpha=al, beta=be)
        assert abs(O.spr_loss(x, y, **kw) - loss) < 1e-9 * max(1, abs(loss))
        assert abs(O.spr_test_nll(x, y, xt, yt, 0.25, 1.5, **kw) - nll) < 1e-7 * max(1, abs(nll))
    for act in ("relu", "erf"):
        assert np.allclose(O.cnn_kernel(gold["cnn/x"], None, 3, act, 1.3, 0.2, 0.9), gold["cnn/k_" + act], rtol=1e-12)
        for bs in (1, 2):
            assert np.allclose(O.conv_resnet_kernel(gold["resnet/x"], None, bs, act, 1.2, 0.3, 0.9),
                               gold["resnet/k_%s_%d" % (act, bs)], rtol=1e-12)


# ----------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-8), (np.float32, 2e-3)])
def test_hip_kernels_match_golden(gold, dtype, tol):
    from smnngp import nt_kernels
    for name in [str(n) for n in gold["names"]]:
        c = _case(name)
        x, x2 = _inputs(c)
        kfn = nt_kernels.get_mlp_kernel(c["L"], act=c["act"], w_std=c["w"], b_std=c["b"], last_w_std=c["lw"])
        got = kfn(x.astype(dtype), None, get=("nngp", "ntk"))
        ref = gold[name + "/k"]
        assert relerr(np.asarray(got.nngp), ref) < tol, name
        if name + "/t" in gold:
            rt = gold[name + "/t"]
            assert relerr(np.asarray(got.ntk), rt) < 5 * tol, name
        kc = np.asarray(kfn(x2.astype(dtype), x.astype(dtype), get="nngp"))
        assert relerr(kc, gold[name + "/kc"]) < tol, name


@pytest.mark.gpu
def test_hip_heads_match_golden(gold):
    from smnngp import nt_kernels
    from smnngp.spax.kernels import NNGPKernel
    from smnngp.spax.likelihoods import GaussianLikelihood, StudentTLikelihood
    from smnngp.spax.models import SPR
    x, y, xt, yt = gold["heads/x"], gold["heads/y"], gold["heads/xt"], gold["heads/yt"]
    for eps, al, be, tp, res, loss, nll in gold["heads/table"]:
        fac = nt_kernels.get_dense_resnet_kernel if res else nt_kernels.get_mlp_kernel
        kernel = NNGPKernel(lambda w, b, l: fac(2, act="relu", w_std=w, b_std=b, last_w_std=l), 1.4, 0.3, 1.0)
        lik = StudentTLikelihood(al, be) if tp else GaussianLikelihood()
        model = SPR(kernel, lik, x, y, 0.25, 1.5, eps=eps)
        assert abs(model.loss() - loss) < 1e-7 * max(1, abs(loss)), (eps, al, be, tp, res)
        # eps = 1e-6 makes the predictive system cond ~1e8: 1e-5 is the north-star fp64 bar
        assert abs(model.test_nll(xt, yt) - nll) < 1e-5 * max(1, abs(nll)), (eps, al, be, tp, res)


@pytest.mark.gpu
def test_hip_cnn_matches_golden(gold):
    from smnngp import nt_kernels
    for act in ("relu", "erf"):
        k = np.asarray(nt_kernels.get_cnn_kernel(3, act=act, w_std=1.3, b_std=0.2, last_w_std=0.9)(gold["cnn/x"]))
        assert relerr(k, gold["cnn/k_" + act]) < 1e-9
        for bs in (1, 2):
            kr = np.asarray(nt_kernels.get_conv_resnet_kernel(bs, 10, act=act, w_std=1.2, b_std=0.3, last_w_std=0.9)(gold["resnet/x"]))
            ref = gold["resnet/k_%s_%d" % (act, bs)]
            assert relerr(kr, ref) < 1e-9
