"""Generates tests/golden/nngp_golden.npz from the CPU oracle (oracle/nngp_oracle.py).

The reference cannot run here (jax / neural_tangents / objax are not installed) and holds no vectors of
its own, so these are the oracle's outputs on seeded inputs over the reference's own hyper-parameter
grids (experiments/regression/find.py:18-22, train.py:37-45) plus the closed-form known answers of
SURVEY.md section 4.  They pin the ORACLE against drift (tests/test_golden.py, CPU) and the HIP path
against the oracle-at-commit-time (GPU).  Run:  python tests/golden/make_golden.py
"""
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import nngp_oracle as O  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nngp_golden.npz")


def cases():
    """(name, kwargs) — kept small: N <= 64."""
    out = []
    grid = [(8, 6), (33, 6), (64, 50)]
    params = [(1.0, 1e-8, 1.0), (1.4, 0.3, 0.8), (2.0, 1.0, 1.0), (1.0, 0.0, 1.0)]
    for (n, d), (w, b, lw), act, L in itertools.product(grid, params, ("relu", "erf"), (1, 2, 4, 6)):
        if n == 33 and (L in (1, 6) or w in (1.0,)):      # keep the fixture file small: full grid only at N=8
            continue
        if n == 64 and not (L == 4 and w == 1.4):
            continue
        out.append(("mlp_n%d_d%d_%s_L%d_w%g_b%g_lw%g" % (n, d, act, L, w, b, lw),
                    dict(n=n, d=d, w=w, b=b, lw=lw, act=act, L=L)))
    return out


def main():
    data = {}
    # known answers
    data["kat_inputs"] = np.array([1.3, 0.7, 0.45])
    data["kat_relu"] = np.array([0.28155319635922454, 0.3281848189079046])
    data["kat_erf"] = np.array([0.19810604275632643, 0.45501869192057837])
    names = []
    for name, c in cases():
        rng = np.random.default_rng(1000 * c["n"] + c["d"])
        x = rng.standard_normal((c["n"], c["d"]))
        x2 = rng.standard_normal((5, c["d"]))
        k, t = O.mlp_kernel(x, None, c["L"], c["act"], c["w"], c["b"], c["lw"], ("nngp", "ntk"))
        kc = O.mlp_kernel(x2, x, c["L"], c["act"], c["w"], c["b"], c["lw"])
        data[name + "/k"] = k; data[name + "/kc"] = kc
        if c["n"] < 64:
            data[name + "/t"] = t
        names.append(name)
    # inference heads on the reference's eps / alpha / beta grids
    rng = np.random.default_rng(7)
    x = rng.standard_normal((40, 6)); y = rng.standard_normal(40)
    xt = rng.standard_normal((9, 6)); yt = rng.standard_normal(9)
    data["heads/x"] = x; data["heads/y"] = y; data["heads/xt"] = xt; data["heads/yt"] = yt
    rows = []
    for eps, (al, be), method, net in itertools.product((1e-6, 1e-2), ((1., 1.), (2., 2.), (3., 1.)), ("gp", "tp"),
                                                        ("mlp", "resnet")):
        kw = dict(kernel=net, num_hiddens=2, act="relu", w_std=1.4, b_std=0.3, last_w_std=1.0, eps=eps,
                  method=method, alpha=al, beta=be)
        loss = O.spr_loss(x, y, **kw)
        nll, mean, cov = O.spr_test_nll(x, y, xt, yt, 0.25, 1.5, return_parts=True, **kw)
        rows.append((eps, al, be, 0.0 if method == "gp" else 1.0, 0.0 if net == "mlp" else 1.0, loss, nll))
        data["heads/mean_%s_%g" % (net, eps)] = mean
        data["heads/cov_%s_%g" % (net, eps)] = cov
    data["heads/table"] = np.array(rows)
    # conv kernel
    xi = np.random.default_rng(11).standard_normal((6, 5, 4, 3))
    data["cnn/x"] = xi
    for act in ("relu", "erf"):
        data["cnn/k_" + act] = O.cnn_kernel(xi, None, 3, act, 1.3, 0.2, 0.9)
    # conv-resnet (WideResnet) kernel, block sizes 1 and 2
    xr = np.random.default_rng(12).standard_normal((5, 8, 16, 2))
    data["resnet/x"] = xr
    for act in ("relu", "erf"):
        for bs in (1, 2):
            data["resnet/k_%s_%d" % (act, bs)] = O.conv_resnet_kernel(xr, None, bs, act, 1.2, 0.3, 0.9)
    data["names"] = np.array(names)
    np.savez_compressed(OUT, **data)
    print("wrote %s: %d arrays, %.1f KB" % (OUT, len(data), os.path.getsize(OUT) / 1024))


if __name__ == "__main__":
    main()
