"""CPU oracle (NumPy/SciPy) for the scale-mixture NNGP hot path.

THIS FILE IS TEST INFRASTRUCTURE.  Only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may import it.  The product
(``smnngp``) never does: it calls the HIP library through the C-ABI and fails
loudly when that library is missing.

PARITY UNPINNED.  The reference (Hyungi-Lee/Scale-Mixtures-of-Neural-Network-
Gaussian-Processes) holds no tests, no golden vectors and no fixtures for this
path, and its arithmetic lives in un-vendored, un-pinned third-party packages
(neural_tangents ~0.3.x, jax ~0.2.2x, objax ~1.4) that are not importable in
this container (ordinary ModuleNotFoundError; nothing was denied).  This file
therefore restates the *published* algorithms of those packages and anchors on
the reference's own call sites.  What pins it instead (tests/test_oracle_*.py):
closed forms vs orthant / Gauss-Hermite quadrature, layer composition vs a
finite-width Monte-Carlo network, the log-pdfs vs scipy.stats, the posterior
vs scipy.linalg.cho_solve, and the known-answer values of SURVEY.md section 4.

Each function cites the reference file:line whose behaviour it follows
(paths relative to /root/reference).
"""
from __future__ import annotations

import math

import numpy as np
import scipy.linalg as sla
from scipy.special import gammaln

__all__ = [
    "softplus", "softplus_inverse", "get_act", "input_gram", "diag_recursion",
    "mlp_kernel", "dense_resnet_kernel", "cnn_kernel", "predict", "predict_ntk",
    "mvn_logpdf", "mvt_logpdf", "normal_logpdf", "student_t_logpdf", "spr_loss",
    "spr_test_nll", "jitter", "relu_map", "erf_map",
]


# --------------------------------------------------------------------------
# spax/bijectors.py:31-63, spax/base.py:15-25  (positive constraint)
# --------------------------------------------------------------------------
def softplus(raw):
    """Softplus(lower=0).__call__ — spax/bijectors.py:39-40,52."""
    raw = np.asarray(raw, dtype=np.float64)
    return np.logaddexp(raw, 0.0)


def softplus_inverse(x):
    """Softplus.base_inv — spax/bijectors.py:53 (identity for x >= 20)."""
    x = np.asarray(x, dtype=np.float64)
    return np.where(x < 20.0, np.log(np.expm1(np.minimum(x, 20.0))), x)


def jitter(num, eps=1e-6, dtype=np.float64):
    """spax/utils.py:26-27 — eps * I (absolute)."""
    return eps * np.eye(num, dtype=dtype)


# --------------------------------------------------------------------------
# experiments/nt_kernels.py:12-18 — activation lookup
# --------------------------------------------------------------------------
def get_act(act):
    if act == "relu":
        return relu_map
    if act == "erf":
        return erf_map
    raise KeyError("Unsupported act '{}'".format(act))


def relu_map(k, q1, q2, theta=None):
    """neural_tangents stax.Relu kernel transform (arc-cosine, Cho & Saul 2009).

    k [N,M] pre-activation covariance, q1 [N], q2 [M] pre-activation variances.
    Returns (k_new, q1_new, q2_new, theta_new).  SURVEY.md Appendix A.2.
    """
    dt = k.dtype
    p = np.outer(q1, q2)
    sp = np.sqrt(p)
    with np.errstate(divide="ignore", invalid="ignore"):
        c = np.where(sp > 0, k / sp, 0.0)
    c = np.clip(c, -1.0, 1.0)
    ang = np.arccos(c)
    kdot = (np.pi - ang) / (2.0 * np.pi)
    k_new = (np.sqrt(np.maximum(p - k * k, 0.0)) + (np.pi - ang) * k) / (2.0 * np.pi)
    th = None if theta is None else (theta * kdot).astype(dt)
    return k_new.astype(dt), (q1 / 2.0).astype(dt), (q2 / 2.0).astype(dt), th


def erf_map(k, q1, q2, theta=None):
    """neural_tangents stax.Erf kernel transform (Williams 1997). Appendix A.2."""
    dt = k.dtype
    p = np.outer(1.0 + 2.0 * q1, 1.0 + 2.0 * q2)
    kdot = 4.0 / (np.pi * np.sqrt(np.maximum(p - 4.0 * k * k, 0.0)))
    k_new = (2.0 / np.pi) * np.arcsin(np.clip(2.0 * k / np.sqrt(p), -1.0, 1.0))
    q1n = (2.0 / np.pi) * np.arcsin(2.0 * q1 / (1.0 + 2.0 * q1))
    q2n = (2.0 / np.pi) * np.arcsin(2.0 * q2 / (1.0 + 2.0 * q2))
    th = None if theta is None else (theta * kdot).astype(dt)
    return k_new.astype(dt), q1n.astype(dt), q2n.astype(dt), th


def input_gram(x1, x2=None):
    """K0 = x1 x2^T / d and the two diagonals (NT _inputs_to_kernel; called at
    spax/kernels.py:25,27).  Appendix A.1."""
    x1 = np.asarray(x1)
    x1f = x1.reshape(x1.shape[0], -1)
    x2f = x1f if x2 is None else np.asarray(x2).reshape(np.asarray(x2).shape[0], -1)
    d = x1f.shape[1]
    k0 = (x1f @ x2f.T) / d
    q1 = np.einsum("ij,ij->i", x1f, x1f) / d
    q2 = np.einsum("ij,ij->i", x2f, x2f) / d
    return k0, q1, q2


def _dense(k, q1, q2, theta, w, b):
    """stax.Dense(width, W_std=w, b_std=b) kernel transform (width does not enter).
    NTK parameterisation: Theta <- K_new + w^2 Theta.  Appendix A.1."""
    k = w * w * k + b * b
    q1 = w * w * q1 + b * b
    q2 = w * w * q2 + b * b
    if theta is not None:
        theta = k + w * w * theta
    return k, q1, q2, theta


def _fix_diag(k, q, sym):
    """Symmetric case: the diagonal of K IS the variance recursion q.  Forcing it removes the
    sqrt(rounding) noise that K_ii / sqrt(q_i q_i) = 1 - 1e-16 injects into acos at c = 1 (the generic
    formula loses half the digits there; at fp64 that shows up as ~1e-9 on K_ii and ~1e-6 on Kdot_ii
    after two layers).  The restated math is unchanged; only its evaluation at c = 1 is made exact."""
    if sym:
        np.fill_diagonal(k, q)


def _ret(k, theta, get):
    if get == "nngp":
        return k
    if get == "ntk":
        return theta
    if tuple(get) == ("nngp", "ntk"):
        return k, theta
    raise ValueError("get must be 'nngp', 'ntk' or ('nngp','ntk')")


def mlp_kernel(x1, x2=None, num_hiddens=1, act="relu", w_std=1.0, b_std=0.0,
               last_w_std=1.0, get="nngp", dtype=np.float64):
    """experiments/nt_kernels.py:21-31 — L x [Dense(512,w,b); act] ; Dense(C, last_w, b=0)."""
    amap = get_act(act)
    x1 = np.asarray(x1, dtype=dtype)
    x2 = None if x2 is None else np.asarray(x2, dtype=dtype)
    sym = x2 is None
    k, q1, q2 = input_gram(x1, x2)
    _fix_diag(k, q1, sym)
    theta = None if get == "nngp" else np.zeros_like(k)
    for _ in range(num_hiddens):
        k, q1, q2, theta = _dense(k, q1, q2, theta, w_std, b_std)
        _fix_diag(k, q1, sym)
        k, q1, q2, theta = amap(k, q1, q2, theta)
        _fix_diag(k, q1, sym)
    k, q1, q2, theta = _dense(k, q1, q2, theta, last_w_std, 0.0)
    _fix_diag(k, q1, sym)
    return _ret(k, theta, get)


def diag_recursion(q0, num_hiddens, act, w_std, b_std, last_w_std):
    """Per-row variance through the MLP stack (the diagonal of mlp_kernel(x, x))."""
    q = np.asarray(q0, dtype=np.float64)
    for _ in range(num_hiddens):
        q = w_std * w_std * q + b_std * b_std
        if act == "relu":
            q = q / 2.0
        elif act == "erf":
            q = (2.0 / np.pi) * np.arcsin(2.0 * q / (1.0 + 2.0 * q))
        else:
            raise KeyError("Unsupported act '{}'".format(act))
    return last_w_std * last_w_std * q


def dense_resnet_kernel(x1, x2=None, num_hiddens=1, act="relu", w_std=1.0, b_std=0.0,
                        last_w_std=1.0, get="nngp", dtype=np.float64):
    """experiments/nt_kernels.py:83-103 — Dense; L x {FanOut; (act;Dense) + Identity; FanInSum}; act; Dense."""
    amap = get_act(act)
    x1 = np.asarray(x1, dtype=dtype)
    x2 = None if x2 is None else np.asarray(x2, dtype=dtype)
    sym = x2 is None
    k, q1, q2 = input_gram(x1, x2)
    theta = None if get == "nngp" else np.zeros_like(k)
    k, q1, q2, theta = _dense(k, q1, q2, theta, w_std, b_std)
    _fix_diag(k, q1, sym)
    for _ in range(num_hiddens):
        kb, q1b, q2b, tb = amap(k, q1, q2, theta)
        kb, q1b, q2b, tb = _dense(kb, q1b, q2b, tb, w_std, b_std)
        k, q1, q2 = kb + k, q1b + q1, q2b + q2
        _fix_diag(k, q1, sym)
        if theta is not None:
            theta = tb + theta
    k, q1, q2, theta = amap(k, q1, q2, theta)
    k, q1, q2, theta = _dense(k, q1, q2, theta, last_w_std, 0.0)
    _fix_diag(k, q1, sym)
    return _ret(k, theta, get)


def _box3(a):
    """3x3 zero-padded box SUM over the last two axes (divisor applied by caller)."""
    p = np.pad(a, [(0, 0)] * (a.ndim - 2) + [(1, 1), (1, 1)])
    h, w = a.shape[-2:]
    out = np.zeros_like(a)
    for dh in range(3):
        for dw in range(3):
            out += p[..., dh:dh + h, dw:dw + w]
    return out


def cnn_kernel(x1, x2=None, num_hiddens=1, act="relu", w_std=1.0, b_std=0.0,
               last_w_std=1.0, dtype=np.float64):
    """experiments/nt_kernels.py:34-45 — L x [Conv(1,3x3,SAME,w,b); act]; Flatten; Dense(last_w).
    x [N,H,W,C].  NNGP only.  Appendix A.4 (documented NT behaviour, small sizes only)."""
    x1 = np.asarray(x1, dtype=dtype)
    x2 = x1 if x2 is None else np.asarray(x2, dtype=dtype)
    c = x1.shape[-1]
    k = np.einsum("nhwc,mhwc->nmhw", x1, x2) / c
    q1 = np.einsum("nhwc,nhwc->nhw", x1, x1) / c
    q2 = np.einsum("mhwc,mhwc->mhw", x2, x2) / c
    for _ in range(num_hiddens):
        k = w_std ** 2 * _box3(k) / 9.0 + b_std ** 2
        q1 = w_std ** 2 * _box3(q1) / 9.0 + b_std ** 2
        q2 = w_std ** 2 * _box3(q2) / 9.0 + b_std ** 2
        if act == "relu":
            p = q1[:, None] * q2[None, :]
            sp = np.sqrt(p)
            with np.errstate(divide="ignore", invalid="ignore"):
                cc = np.where(sp > 0, k / sp, 0.0)
            ang = np.arccos(np.clip(cc, -1.0, 1.0))
            k = (np.sqrt(np.maximum(p - k * k, 0.0)) + (np.pi - ang) * k) / (2 * np.pi)
            q1, q2 = q1 / 2.0, q2 / 2.0
        elif act == "erf":
            p = (1 + 2 * q1)[:, None] * (1 + 2 * q2)[None, :]
            k = (2 / np.pi) * np.arcsin(np.clip(2 * k / np.sqrt(p), -1.0, 1.0))
            q1 = (2 / np.pi) * np.arcsin(2 * q1 / (1 + 2 * q1))
            q2 = (2 / np.pi) * np.arcsin(2 * q2 / (1 + 2 * q2))
        else:
            raise KeyError("Unsupported act '{}'".format(act))
    return (last_w_std ** 2 * k.mean(axis=(2, 3))).astype(dtype)


# --------------------------------------------------------------------------
# spax/kernels.py:29-32 -> neural_tangents.predict.gradient_descent_mse_ensemble
# --------------------------------------------------------------------------
def predict(k_dd, k_td, k_tt, y, diag_reg=0.0, diag_reg_absolute_scale=False):
    """t = infinity NNGP posterior.  Appendix A.5.

    K~ = K_dd + diag_reg * (tr K_dd / N) * I  (NT's *relative* ridge; absolute when
    diag_reg_absolute_scale), mean = K_td K~^-1 y [T,C], cov = K_tt - K_td K~^-1 K_dt."""
    k_dd = np.asarray(k_dd)
    n = k_dd.shape[0]
    scale = 1.0 if diag_reg_absolute_scale else np.trace(k_dd) / n
    kt = k_dd + diag_reg * scale * np.eye(n, dtype=k_dd.dtype)
    cf = sla.cho_factor(kt, lower=True)
    y = np.asarray(y, dtype=k_dd.dtype)
    y2 = y[:, None] if y.ndim == 1 else y
    mean = k_td @ sla.cho_solve(cf, y2)
    cov = k_tt - k_td @ sla.cho_solve(cf, k_td.T)
    return mean, cov


def predict_ntk(k_dd, k_td, k_tt, t_dd, t_td, y, diag_reg=0.0):
    """get='ntk' posterior (sample.ipynb:186-195 only).  Appendix A.5."""
    n = k_dd.shape[0]
    tt = t_dd + diag_reg * (np.trace(t_dd) / n) * np.eye(n, dtype=t_dd.dtype)
    cf = sla.cho_factor(tt, lower=True)
    y2 = y[:, None] if y.ndim == 1 else y
    mean = t_td @ sla.cho_solve(cf, y2)
    a = sla.cho_solve(cf, t_td.T)          # Theta~^-1 Theta_dt   [N,T]
    cov = k_tt + a.T @ k_dd @ a - (a.T @ k_td.T + k_td @ a)
    return mean, cov


# --------------------------------------------------------------------------
# spax/likelihoods.py:25-28 (jax MVN logpdf), spax/utils.py:160-183 (MVT)
# --------------------------------------------------------------------------
def mvn_logpdf(x, cov):
    """Zero-mean MVN log-pdf: -1/2 z'z - N/2 log 2pi - sum log L_ii, z = L^-1 x.
    NaN when cov is not PD (JAX Cholesky semantics: silent NaN)."""
    n = x.shape[-1]
    try:
        l = sla.cholesky(cov, lower=True)
    except sla.LinAlgError:
        return np.float64("nan")
    z = sla.solve_triangular(l, x, lower=True)
    return float(-0.5 * z @ z - n / 2 * np.log(2 * np.pi) - np.log(np.diag(l)).sum())


def mvt_logpdf(x, shape, df):
    """spax/utils.py:178-183 with loc = 0."""
    n = x.shape[-1]
    t = 0.5 * (df + n)
    try:
        l = sla.cholesky(shape, lower=True)
    except sla.LinAlgError:
        return np.float64("nan")
    z = sla.solve_triangular(l, x, lower=True)
    return float(-t * np.log(1 + (z @ z) / df) - n / 2 * np.log(df * np.pi)
                 + gammaln(t) - gammaln(0.5 * df) - np.log(np.diag(l)).sum())


def normal_logpdf(x, mean, sigma):
    """jax.scipy.stats.norm.logpdf — spax/likelihoods.py:32."""
    z = (x - mean) / sigma
    return -0.5 * z * z - np.log(sigma) - 0.5 * np.log(2 * np.pi)


def student_t_logpdf(x, df, loc, scale):
    """jax.scipy.stats.t.logpdf — spax/likelihoods.py:64."""
    z = (x - loc) / scale
    return (gammaln(0.5 * (df + 1)) - gammaln(0.5 * df) - 0.5 * np.log(df * np.pi)
            - np.log(scale) - 0.5 * (df + 1) * np.log1p(z * z / df))


# --------------------------------------------------------------------------
# spax/models.py:81-120 — SPR
# --------------------------------------------------------------------------
def spr_loss(x, y, *, kernel="mlp", num_hiddens=1, act="relu", w_std=1.0, b_std=1.0,
             last_w_std=1.0, eps=1e-6, method="gp", alpha=2.0, beta=2.0, dtype=np.float64):
    """SPR.loss — spax/models.py:93-98.  -logpdf / N with cov = K + eps I (absolute)."""
    kfn = {"mlp": mlp_kernel, "resnet": dense_resnet_kernel}[kernel]
    k = kfn(x, None, num_hiddens, act, w_std, b_std, last_w_std, "nngp", dtype)
    n = k.shape[0]
    cov = k + jitter(n, eps, k.dtype)
    y = np.asarray(y, dtype=k.dtype)
    if method == "gp":
        lp = mvn_logpdf(y, cov)                                   # likelihoods.py:25-28
    elif method == "tp":
        lp = mvt_logpdf(y, (beta / alpha) * cov, 2.0 * alpha)     # likelihoods.py:45-50
    else:
        raise KeyError(method)
    return -lp / n


def spr_test_nll(x, y, x_test, y_test, y_mean=0.0, y_std=1.0, *, kernel="mlp", num_hiddens=1,
                 act="relu", w_std=1.0, b_std=1.0, last_w_std=1.0, eps=1e-6, method="gp",
                 alpha=2.0, beta=2.0, dtype=np.float64, return_parts=False):
    """SPR.test_nll — spax/models.py:100-120 (+ likelihoods.py:30-33,52-65)."""
    kfn = {"mlp": mlp_kernel, "resnet": dense_resnet_kernel}[kernel]
    args = (num_hiddens, act, w_std, b_std, last_w_std, "nngp", dtype)
    k_dd = kfn(x, None, *args)
    k_td = kfn(x_test, x, *args)
    k_tt = kfn(x_test, None, *args)
    y = np.asarray(y, dtype=k_dd.dtype)
    mean, cov = predict(k_dd, k_td, k_tt, y[:, None], diag_reg=eps)   # relative ridge
    xs = np.asarray(y_test) * y_std + y_mean
    ms = mean.ravel() * y_std + y_mean
    covs = cov * y_std ** 2
    if method == "gp":
        lp = normal_logpdf(xs, ms, np.sqrt(np.diag(covs)))
    elif method == "tp":
        n = k_dd.shape[0]
        df = 2.0 * alpha
        khat = (beta / alpha) * k_dd + jitter(n, 1e-6, k_dd.dtype)   # likelihoods.py:60 (K without eps)
        d = df + y @ sla.cho_solve(sla.cho_factor(khat, lower=True), y)
        sigma = np.sqrt(np.diag(d / (df + n) * (beta / alpha) * covs))
        lp = student_t_logpdf(xs, df + n, ms, sigma)
    else:
        raise KeyError(method)
    nll = -float(np.mean(lp))
    if return_parts:
        return nll, mean, cov
    return nll


def spr_loss_grad_fd(x, y, keys=("w_std", "b_std", "last_w_std", "eps", "alpha", "beta"), h=1e-5, **kw):
    """d spr_loss / d (constrained hyper-parameter) by central differences in fp64 — the quantity
    objax.GradValues(model.loss, vars) differentiates (experiments/regression/train.py:61-67), before the
    softplus chain rule.  Checker for the analytic gradient (csrc/grad.hip); the step is h * |value|."""
    out = {}
    for k in keys:
        v = float(kw[k])
        step = h * abs(v) if v != 0.0 else h          # relative: eps ~ 1e-3 needs a step far below 1e-5
        up = dict(kw); up[k] = v + step
        dn = dict(kw); dn[k] = v - step
        out[k] = (spr_loss(x, y, **up) - spr_loss(x, y, **dn)) / (2.0 * step)
    return out


# --------------------------------------------------------------------------
# experiments/nt_kernels.py:48-80 — get_conv_resnet_kernel (WideResnet, k = 1, no pooling)
# --------------------------------------------------------------------------
def _conv3_same(a, stride):
    """Kernel transform of stax.Conv(_, (3,3), strides=(s,s), padding='SAME') on the last two axes of a same-pixel
    covariance map: the 3x3 zero-padded window SUM (the caller scales by w^2/9 and adds b^2).  SAME follows
    lax.padtype_to_pads: out = ceil(in/s), pad_total = max((out-1) s + 3 - in, 0), pad_lo = pad_total // 2."""
    h, w = a.shape[-2:]
    oh, ow = -(-h // stride), -(-w // stride)
    ph, pw = max((oh - 1) * stride + 3 - h, 0), max((ow - 1) * stride + 3 - w, 0)
    pad = [(0, 0)] * (a.ndim - 2) + [(ph // 2, ph - ph // 2), (pw // 2, pw - pw // 2)]
    ap = np.pad(a, pad)
    out = np.zeros(a.shape[:-2] + (oh, ow), dtype=a.dtype)
    for dy in range(3):
        for dx in range(3):
            out += ap[..., dy: dy + (oh - 1) * stride + 1: stride, dx: dx + (ow - 1) * stride + 1: stride]
    return out


def _act_maps(k, q1, q2, act):
    """The per-pixel activation transform of cnn_kernel on (K [N,M,H,W], q1 [N,H,W], q2 [M,H,W])."""
    if act == "relu":
        p = q1[:, None] * q2[None, :]
        sp = np.sqrt(p)
        with np.errstate(divide="ignore", invalid="ignore"):
            cc = np.where(sp > 0, k / sp, 0.0)
        ang = np.arccos(np.clip(cc, -1.0, 1.0))
        return (np.sqrt(np.maximum(p - k * k, 0.0)) + (np.pi - ang) * k) / (2 * np.pi), q1 / 2.0, q2 / 2.0
    if act == "erf":
        p = (1 + 2 * q1)[:, None] * (1 + 2 * q2)[None, :]
        kk = (2 / np.pi) * np.arcsin(np.clip(2 * k / np.sqrt(p), -1.0, 1.0))
        return kk, (2 / np.pi) * np.arcsin(2 * q1 / (1 + 2 * q1)), (2 / np.pi) * np.arcsin(2 * q2 / (1 + 2 * q2))
    raise KeyError("Unsupported act '{}'".format(act))


def conv_resnet_kernel(x1, x2=None, num_hiddens=1, act="relu", w_std=1.0, b_std=0.0, last_w_std=1.0,
                       dtype=np.float64):
    """experiments/nt_kernels.py:48-80 — WideResnet(block_size=num_hiddens, k=1): Conv; four groups of residual
    blocks (strides 1, 2, 2, 2; the first block of a group has a Conv shortcut, the others Identity; a block's main
    path is act, Conv(stride), act, Conv); Flatten; Dense(last_w).  The AvgPool is commented out in the reference,
    so only same-pixel covariances enter.  x [N,H,W,C].  NNGP only; small sizes only (documented NT behaviour)."""
    x1 = np.asarray(x1, dtype=dtype)
    sym = x2 is None
    x2 = x1 if sym else np.asarray(x2, dtype=dtype)
    c = x1.shape[-1]
    w2, b2 = w_std ** 2, b_std ** 2
    state = (np.einsum("nhwc,mhwc->nmhw", x1, x2) / c, np.einsum("nhwc,nhwc->nhw", x1, x1) / c,
             np.einsum("mhwc,mhwc->mhw", x2, x2) / c)

    def conv(st, stride):
        return tuple(w2 * _conv3_same(a, stride) / 9.0 + b2 for a in st)

    def block(st, stride, mismatch):
        main = conv(_act_maps(*st, act), stride)
        main = conv(_act_maps(*main, act), 1)
        short = conv(st, stride) if mismatch else st
        return tuple(m + s for m, s in zip(main, short))

    state = conv(state, 1)
    for stride in (1, 2, 2, 2):
        state = block(state, stride, True)
        for _ in range(num_hiddens - 1):
            state = block(state, 1, False)
    return (last_w_std ** 2 * state[0].mean(axis=(2, 3))).astype(dtype)
