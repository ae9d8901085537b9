"""Pins the CPU oracle (oracle/nngp_oracle.py) without the reference:
closed forms vs quadrature, composition vs a finite-width network, log-pdfs vs
scipy.stats, posterior vs scipy.linalg, known-answer values of SURVEY.md section 4."""
import numpy as np
import pytest
import scipy.linalg as sla
import scipy.stats as st
from scipy import integrate
from scipy.special import erf

from oracle import nngp_oracle as O

Q1, Q2, K12 = 1.3, 0.7, 0.45


def _pdf(u, v, q1, q2, k):
    det = q1 * q2 - k * k
    return np.exp(-0.5 * (q2 * u * u - 2 * k * u * v + q1 * v * v) / det) / (2 * np.pi * np.sqrt(det))


def test_relu_kat_and_quadrature():
    k = np.array([[K12]]); q1 = np.array([Q1]); q2 = np.array([Q2])
    kn, q1n, q2n, th = O.relu_map(k, q1, q2, theta=np.ones_like(k))
    assert abs(kn[0, 0] - 0.28155319635922454) < 1e-15
    assert abs(th[0, 0] - 0.3281848189079046) < 1e-15
    assert q1n[0] == Q1 / 2 and q2n[0] == Q2 / 2
    val, _ = integrate.dblquad(lambda v, u: u * v * _pdf(u, v, Q1, Q2, K12), 0, 12, 0, 12,
                               epsabs=1e-12, epsrel=1e-12)
    assert abs(val - kn[0, 0]) < 1e-9
    dval, _ = integrate.dblquad(lambda v, u: _pdf(u, v, Q1, Q2, K12), 0, 12, 0, 12,
                                epsabs=1e-12, epsrel=1e-12)
    assert abs(dval - th[0, 0]) < 1e-9


def test_erf_kat_and_gauss_hermite():
    k = np.array([[K12]]); q1 = np.array([Q1]); q2 = np.array([Q2])
    kn, q1n, q2n, th = O.erf_map(k, q1, q2, theta=np.ones_like(k))
    assert abs(kn[0, 0] - 0.19810604275632643) < 1e-15
    assert abs(th[0, 0] - 0.45501869192057837) < 1e-15
    x, w = np.polynomial.hermite_e.hermegauss(120)
    l = np.linalg.cholesky(np.array([[Q1, K12], [K12, Q2]]))
    u = l[0, 0] * x[:, None] + 0 * x[None, :]
    v = l[1, 0] * x[:, None] + l[1, 1] * x[None, :]
    ww = (w[:, None] * w[None, :]) / (2 * np.pi)
    assert abs((ww * erf(u) * erf(v)).sum() - kn[0, 0]) < 1e-12
    d = lambda z: 2 / np.sqrt(np.pi) * np.exp(-z * z)
    assert abs((ww * d(u) * d(v)).sum() - th[0, 0]) < 1e-12
    # diagonal map: q -> 2/pi asin(2q/(1+2q))
    assert abs(q1n[0] - (2 / np.pi) * np.arcsin(2 * Q1 / (1 + 2 * Q1))) < 1e-16
    assert abs((ww * erf(u) ** 2).sum() - q1n[0]) < 1e-12


@pytest.mark.parametrize("act", ["relu", "erf"])
def test_diag_recursion_matches_kernel_diagonal(act):
    rng = np.random.default_rng(1)
    x = rng.standard_normal((9, 5))
    k = O.mlp_kernel(x, None, 3, act, 1.4, 0.3, 0.7)
    q0 = (x * x).sum(1) / 5
    assert np.allclose(np.diag(k), O.diag_recursion(q0, 3, act, 1.4, 0.3, 0.7), rtol=1e-13)
    assert np.allclose(k, k.T, rtol=1e-14)
    k12 = O.mlp_kernel(x[:4], x[2:], 3, act, 1.4, 0.3, 0.7)
    assert np.allclose(k12, k[:4, 2:], rtol=1e-12, atol=1e-14)


@pytest.mark.parametrize("act", ["relu", "erf"])
def test_mlp_nngp_and_ntk_vs_finite_width_network(act):
    """Layer composition (nt_kernels.py:21-31) vs an empirical width-2048 network (32 draws) in the NTK
    parameterisation z = w/sqrt(n_in) W h + b beta; NNGP = E[f f'], NTK = <df/dtheta, df'/dtheta>."""
    torch = pytest.importorskip("torch")
    torch.manual_seed(0)
    rng = np.random.default_rng(3)
    x = torch.tensor(rng.standard_normal((4, 6)), dtype=torch.float64)
    w_std, b_std, lw, L, width, draws = 1.3, 0.4, 0.8, 2, 2048, 32
    phi = torch.relu if act == "relu" else torch.erf
    nngp = np.zeros((4, 4)); ntk = np.zeros((4, 4))
    for _ in range(draws):
        params = []
        n_in = 6
        for _l in range(L):
            params += [torch.randn(n_in, width, dtype=torch.float64, requires_grad=True),
                       torch.randn(width, dtype=torch.float64, requires_grad=True)]
            n_in = width
        params += [torch.randn(n_in, 2048, dtype=torch.float64, requires_grad=True)]

        def f(xx):
            h = xx; n = 6
            for l in range(L):
                h = phi(w_std / np.sqrt(n) * h @ params[2 * l] + b_std * params[2 * l + 1]); n = width
            return lw / np.sqrt(n) * h @ params[-1]
        out = f(x)                                   # [4, 2048] independent output units
        nngp += (out @ out.T).detach().numpy() / 2048 / draws
        g = []
        for i in range(4):
            gi = torch.autograd.grad(out[i, 0], params, retain_graph=True)
            g.append(torch.cat([t.reshape(-1) for t in gi]))
        g = torch.stack(g)
        # the 2048 output heads share the trunk; the NTK of ONE head uses only its own read-out column
        ntk += (g @ g.T).numpy() / draws
    k, t = O.mlp_kernel(x.numpy(), None, L, act, w_std, b_std, lw, ("nngp", "ntk"))
    assert np.max(np.abs(nngp - k)) / np.max(np.abs(k)) < 0.04
    assert np.max(np.abs(ntk - t)) / np.max(np.abs(t)) < 0.04


def test_logpdfs_vs_scipy_stats():
    rng = np.random.default_rng(4)
    a = rng.standard_normal((7, 7)); cov = a @ a.T + 0.5 * np.eye(7)
    y = rng.standard_normal(7)
    assert abs(O.mvn_logpdf(y, cov) - st.multivariate_normal.logpdf(y, np.zeros(7), cov)) < 1e-12
    for df in (2.0, 4.0, 7.5):
        assert abs(O.mvt_logpdf(y, cov, df) - st.multivariate_t.logpdf(y, np.zeros(7), cov, df)) < 1e-12
    x = rng.standard_normal(5); m = rng.standard_normal(5); s = rng.uniform(0.5, 2, 5)
    assert np.allclose(O.normal_logpdf(x, m, s), st.norm.logpdf(x, m, s), rtol=1e-13)
    assert np.allclose(O.student_t_logpdf(x, 9.0, m, s), st.t.logpdf(x, 9.0, m, s), rtol=1e-13)
    assert np.isnan(O.mvn_logpdf(y, -cov))


def test_predict_vs_explicit_inverse():
    rng = np.random.default_rng(5)
    x = rng.standard_normal((12, 4)); xt = rng.standard_normal((5, 4)); y = rng.standard_normal((12, 1))
    kw = dict(num_hiddens=2, act="relu", w_std=1.2, b_std=0.2, last_w_std=1.0)
    kdd = O.mlp_kernel(x, None, **kw); ktd = O.mlp_kernel(xt, x, **kw); ktt = O.mlp_kernel(xt, None, **kw)
    mean, cov = O.predict(kdd, ktd, ktt, y, diag_reg=1e-3)
    kt = kdd + 1e-3 * np.trace(kdd) / 12 * np.eye(12)
    inv = np.linalg.inv(kt)
    assert np.allclose(mean, ktd @ inv @ y, rtol=1e-9)
    assert np.allclose(cov, ktt - ktd @ inv @ ktd.T, rtol=1e-8, atol=1e-12)
    # joint-matrix identity the HIP path relies on: the Schur complement of the joint kernel
    xa = np.concatenate([x, xt]); kj = O.mlp_kernel(xa, None, **kw)
    assert np.allclose(kj[:12, :12], kdd) and np.allclose(kj[12:, :12], ktd) and np.allclose(kj[12:, 12:], ktt)


def test_softplus_roundtrip():
    for v in (1e-8, 1e-6, 0.3, 1.0, 19.9, 25.0):
        assert abs(O.softplus(O.softplus_inverse(v)) - v) < 1e-9 * max(1.0, v) + 1e-17 or v >= 20
    assert O.softplus_inverse(25.0) == 25.0      # bijectors.py:53 guard


def test_spr_paths_run_and_are_consistent():
    rng = np.random.default_rng(6)
    x = rng.standard_normal((20, 3)); y = rng.standard_normal(20)
    xt = rng.standard_normal((6, 3)); yt = rng.standard_normal(6)
    kw = dict(num_hiddens=2, act="relu", w_std=1.0, b_std=0.5, last_w_std=1.0, eps=1e-2)
    lg = O.spr_loss(x, y, method="gp", **kw); lt = O.spr_loss(x, y, method="tp", alpha=2., beta=2., **kw)
    assert np.isfinite(lg) and np.isfinite(lt)
    # Student-t with nu -> inf and b/a = 1 tends to the Gaussian LML
    linf = O.spr_loss(x, y, method="tp", alpha=1e6, beta=1e6, **kw)
    assert abs(linf - lg) < 1e-2      # converges like 1/nu
    ng = O.spr_test_nll(x, y, xt, yt, 0.3, 1.7, method="gp", **kw)
    nt = O.spr_test_nll(x, y, xt, yt, 0.3, 1.7, method="tp", alpha=2., beta=2., **kw)
    assert np.isfinite(ng) and np.isfinite(nt)


def test_cnn_kernel_reduces_to_mlp_for_1x1_images_and_is_psd():
    rng = np.random.default_rng(7)
    x = rng.standard_normal((6, 1, 1, 5))
    kc = O.cnn_kernel(x, None, 2, "relu", 3.0, 0.2, 0.9)           # 1x1 image: box mean = value/9
    km = O.mlp_kernel(x.reshape(6, 5), None, 2, "relu", 1.0, 0.2, 0.9)
    assert np.allclose(kc, km, rtol=1e-12)
    x = rng.standard_normal((5, 4, 4, 3))
    for act in ("relu", "erf"):
        k = O.cnn_kernel(x, None, 3, act, 1.3, 0.1, 1.0)
        assert np.allclose(k, k.T) and np.linalg.eigvalsh(k).min() > -1e-12
        assert np.allclose(O.cnn_kernel(x[:2], x[1:], 3, act, 1.3, 0.1, 1.0), k[:2, 1:], rtol=1e-12)


def test_dense_resnet_kernel_basic():
    rng = np.random.default_rng(8)
    x = rng.standard_normal((6, 4))
    k, t = O.dense_resnet_kernel(x, None, 2, "relu", 1.1, 0.2, 0.9, ("nngp", "ntk"))
    assert np.allclose(k, k.T) and np.linalg.eigvalsh(k).min() > 0 and np.linalg.eigvalsh(t).min() > 0
    with pytest.raises(KeyError):
        O.mlp_kernel(x, None, 1, "tanh")


@pytest.mark.parametrize("act", ["relu", "erf"])
def test_dense_resnet_nngp_and_ntk_vs_finite_width_network(act):
    """dense_resnet_kernel (experiments/nt_kernels.py:83-103: Dense; L x {FanOut; (act; Dense) + Identity; FanInSum};
    act; Dense(last_w, b=0)) against an empirical width-2048 network in the NTK parameterisation
    z = w/sqrt(n_in) W h + b beta (32 draws): NNGP = E[f f'] over 2048 read-out heads, NTK = <df/dtheta, df'/dtheta>
    of one head.  Monte-Carlo error at this width / draw count is ~1-2 %; asserted at 4 %, element-wise."""
    torch = pytest.importorskip("torch")
    torch.manual_seed(2)
    rng = np.random.default_rng(11)
    n, d, L, width, heads, draws = 4, 6, 2, 2048, 2048, 32
    w_std, b_std, lw = 1.1, 0.4, 0.8
    x = torch.tensor(rng.standard_normal((n, d)), dtype=torch.float64)
    phi = torch.relu if act == "relu" else torch.erf
    nngp = np.zeros((n, n)); ntk = np.zeros((n, n))
    for _ in range(draws):
        params = [torch.randn(d, width, dtype=torch.float64, requires_grad=True),
                  torch.randn(width, dtype=torch.float64, requires_grad=True)]
        for _l in range(L):
            params += [torch.randn(width, width, dtype=torch.float64, requires_grad=True),
                       torch.randn(width, dtype=torch.float64, requires_grad=True)]
        params += [torch.randn(width, heads, dtype=torch.float64, requires_grad=True)]
        h = w_std / np.sqrt(d) * x @ params[0] + b_std * params[1]
        for l in range(L):                                   # ResBlock: Dense(act(h)) + h
            h = w_std / np.sqrt(width) * phi(h) @ params[2 + 2 * l] + b_std * params[3 + 2 * l] + h
        out = lw / np.sqrt(width) * phi(h) @ params[-1]      # [n, heads]
        nngp += (out @ out.T).detach().numpy() / heads / draws
        g = []
        for i in range(n):
            gi = torch.autograd.grad(out[i, 0], params, retain_graph=True)
            g.append(torch.cat([t.reshape(-1) for t in gi]))
        g = torch.stack(g)
        ntk += (g @ g.T).numpy() / draws
    k, t = O.dense_resnet_kernel(x.numpy(), None, L, act, w_std, b_std, lw, ("nngp", "ntk"))
    assert np.max(np.abs(nngp - k) / np.abs(k)) < 0.04, (nngp, k)
    assert np.max(np.abs(ntk - t) / np.abs(t)) < 0.04, (ntk, t)
    # cross form is the off-diagonal block of the joint kernel
    kc, tc = O.dense_resnet_kernel(x.numpy()[:2], x.numpy()[1:], L, act, w_std, b_std, lw, ("nngp", "ntk"))
    assert np.allclose(kc, k[:2, 1:], rtol=1e-9, atol=1e-12) and np.allclose(tc, t[:2, 1:], rtol=1e-6, atol=1e-9)


@pytest.mark.parametrize("act", ["relu", "erf"])
def test_cnn_kernel_vs_finite_width_network(act):
    """cnn_kernel (experiments/nt_kernels.py:34-45: L x [Conv 3x3 stride 1 SAME; act]; Flatten; Dense(last_w)) against an
    empirical 512-channel CNN in the NTK parameterisation (weights w/sqrt(9 c_in), ZERO padding so the divisor stays 9 at
    the border, bias b beta per channel), flattened over (h, w, channel) into 512 read-out heads scaled
    last_w/sqrt(H W ch): E[f f'] over 48 draws.  This is the check of the three conventions the oracle takes from NT's
    documentation: SAME zero padding, the /9 divisor and Flatten = mean over pixels.  MC error 1-2 % (2.1 % worst entry at 96 draws, both activations); asserted 5 %, element-wise."""
    torch = pytest.importorskip("torch")
    F = torch.nn.functional
    torch.manual_seed(3)
    rng = np.random.default_rng(12)
    n, hw, cin, ch, heads, draws, L = 3, 6, 3, 512, 512, 48, 2
    w_std, b_std, lw = 1.3, 0.3, 0.9
    xh = rng.standard_normal((n, hw, hw, cin))
    x = torch.tensor(np.transpose(xh, (0, 3, 1, 2)))                    # NCHW
    phi = torch.relu if act == "relu" else torch.erf
    emp = np.zeros((n, n))
    for _ in range(draws):
        h = x
        for _l in range(L):
            c_in = h.shape[1]
            wgt = torch.randn(ch, c_in, 3, 3, dtype=torch.float64)
            bias = torch.randn(ch, dtype=torch.float64)
            h = phi(F.conv2d(h, w_std / np.sqrt(9 * c_in) * wgt, b_std * bias, stride=1, padding=1))
        flat = h.reshape(n, -1)
        out = lw / np.sqrt(flat.shape[1]) * flat @ torch.randn(flat.shape[1], heads, dtype=torch.float64)
        emp += (out @ out.T).numpy() / heads / draws
    k = O.cnn_kernel(xh, None, L, act, w_std, b_std, lw)
    assert np.max(np.abs(emp - k) / np.abs(k)) < 0.05, (emp, k)
    # a network WITHOUT the border convention (divisor = number of taps inside the image) must NOT match: the test can tell
    k_wrong = O.cnn_kernel(np.pad(xh, ((0, 0), (1, 1), (1, 1), (0, 0)), mode="edge"), None, L, act, w_std, b_std, lw)
    assert np.max(np.abs(emp - k_wrong) / np.abs(k)) > 0.05


@pytest.mark.parametrize("act", ["relu", "erf"])
def test_conv_resnet_kernel_vs_finite_width_network(act):
    """conv_resnet_kernel (nt_kernels.py:48-80) against an empirical WideResnet of 192 channels (NTK
    parameterisation, 3x3 convolutions with stax's SAME padding, strides 1/2/2/2, Conv shortcut in the first block of
    every group, Identity after) on 8x8x3 images: E[f f'] over 24 draws x 512 read-out heads.  Also the structural
    checks: symmetric, PSD, joint-vs-cross consistency."""
    torch = pytest.importorskip("torch")
    F = torch.nn.functional
    torch.manual_seed(1)
    rng = np.random.default_rng(2)
    n, hw, cin, ch, heads, draws, bs = 3, 8, 3, 192, 512, 24, 1
    w_std, b_std, lw = 1.2, 0.3, 0.9
    xh = rng.standard_normal((n, hw, hw, cin))
    x = torch.tensor(np.transpose(xh, (0, 3, 1, 2)))                    # NCHW
    phi = torch.relu if act == "relu" else torch.erf

    def conv(h, stride):
        c_in = h.shape[1]
        wgt = torch.randn(ch, c_in, 3, 3, dtype=torch.float64)
        bias = torch.randn(ch, dtype=torch.float64)
        size = h.shape[-1]
        out = -(-size // stride)
        pad = max((out - 1) * stride + 3 - size, 0)
        h = F.pad(h, (pad // 2, pad - pad // 2, pad // 2, pad - pad // 2))
        return F.conv2d(h, w_std / np.sqrt(9 * c_in) * wgt, b_std * bias, stride=stride)

    emp = np.zeros((n, n))
    for _ in range(draws):
        h = conv(x, 1)
        for stride in (1, 2, 2, 2):
            for blk in range(bs):
                s = stride if blk == 0 else 1
                main = conv(phi(conv(phi(h), s)), 1)
                h = main + (conv(h, s) if blk == 0 else h)
        flat = h.reshape(n, -1)
        out = lw / np.sqrt(flat.shape[1]) * flat @ torch.randn(flat.shape[1], heads, dtype=torch.float64)
        emp += (out @ out.T).numpy() / heads / draws
    k = O.conv_resnet_kernel(xh, None, bs, act, w_std, b_std, lw)
    assert np.max(np.abs(emp - k)) / np.max(np.abs(k)) < 0.06
    assert np.allclose(k, k.T, rtol=1e-13) and np.linalg.eigvalsh(k).min() > 0
    x2 = rng.standard_normal((2, hw, hw, cin))
    kj = O.conv_resnet_kernel(np.concatenate([xh, x2]), None, 2, act, w_std, b_std, lw)
    kc = O.conv_resnet_kernel(xh, x2, 2, act, w_std, b_std, lw)
    assert np.allclose(kc, kj[:n, n:], rtol=1e-12, atol=1e-14)
