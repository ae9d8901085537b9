#!/usr/bin/env python3
"""The reference's regression experiment (experiments/regression/train.py + test.py) on its two offline datasets,
run on the device path.  Not a CLI replica: a short script that shows the drop-in surface end to end.

    python examples/regression_synthetic.py [syn-t|syn-normal] [gp|tp]
"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from smnngp import checkpoint, nt_kernels, train                      # noqa: E402
from smnngp.spax.kernels import NNGPKernel                            # noqa: E402
from smnngp.spax.likelihoods import GaussianLikelihood, StudentTLikelihood  # noqa: E402
from smnngp.spax.models import SPR                                    # noqa: E402


def dataset(name):
    """experiments/regression/data.py:219-236, then permute_dataset(seed=10) and an 80/10/10 split."""
    if name == "syn-normal":
        num, rs = 100, np.random.RandomState(829)
        noise = lambda: rs.standard_normal(size=num) * 0.2
    elif name == "syn-t":
        num, rs = 300, np.random.RandomState(761)
        noise = lambda: rs.standard_t(df=1, size=num) * 0.8
    else:
        raise KeyError("Unsupported dataset '{}'".format(name))
    x = np.linspace(-num / 2, num / 2, num)[:, None]
    y = rs.multivariate_normal(mean=np.zeros(num), cov=np.exp(-0.5 * (x - x.T) ** 2), size=1).flatten() + noise()
    idx = np.random.RandomState(10).permutation(num)
    x, y = x[idx], y[idx]
    a, b = int(0.8 * num), int(0.9 * num)
    xm, xs, ym, ys = x[:a].mean(0), x[:a].std(0), y[:a].mean(), y[:a].std()
    f = lambda u, v: ((u - xm) / xs, (v - ym) / ys)
    return f(x[:a], y[:a]), f(x[a:b], y[a:b]), f(x[b:], y[b:]), (ym, ys)


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "syn-t"
    method = sys.argv[2] if len(sys.argv) > 2 else "tp"
    (xtr, ytr), (xva, yva), (xte, yte), (ym, ys) = dataset(name)
    args = dict(method=method, network="mlp", num_hiddens=2, activation="relu", data_name=name, last_w_std=1.0)
    kernel = NNGPKernel(lambda w, b, l: nt_kernels.get_mlp_kernel(2, act="relu", w_std=w, b_std=b, last_w_std=l), 1.0, 1.0, 1.0)
    likelihood = GaussianLikelihood() if method == "gp" else StudentTLikelihood(2.0, 2.0)
    model = SPR(kernel, likelihood, xtr, ytr, ym, ys, eps=1e-2)
    step = train.build_train_step(model)                              # analytic gradient + Adam
    scheduler = train.PlateauSchedule(lr=0.03, factor=0.5, patience=2)
    run_dir = tempfile.mkdtemp(prefix="smnngp_run_")
    ck = checkpoint.Checkpointer(run_dir)
    checkpoint.save_meta(run_dir, args)
    print("[%5d] NLL: %.5f  TEST: %.5f" % (0, model.test_nll(xva, yva), model.test_nll(xte, yte)))
    for i in range(1, 301):
        nll = step(scheduler.lr)
        if i % 50 == 0:
            valid, test = model.test_nll(xva, yva), model.test_nll(xte, yte)
            ws, bs, ls = kernel.get_params()
            mark = "  (saved)" if ck.step(i, valid, model.vars()) else ""
            print("[%5d] nll: %.5f  ws: %.4f  bs: %.3E  ls: %.4f  e: %.3E  NLL: %.5f  TEST: %.5f%s"
                  % (i, nll, ws, bs, ls, model.eps.safe_value, valid, test, mark))
            if scheduler.step(valid):
                print("LR reduced to %.6f" % scheduler.lr)
                if scheduler.lr < 1e-3:
                    break
    restored, _ = checkpoint.restore_spr(run_dir, xtr, ytr, ym, ys, dtype=np.float64)
    print("restored from %s (step %d): TEST NLL %.5f" % (run_dir, checkpoint.latest_index(run_dir), restored.test_nll(xte, yte)))


if __name__ == "__main__":
    main()
