"""The reference's regression run, end to end, on the device path: experiments/regression/train.py:126-215
(model, train step, validation / checkpoint loop) followed by experiments/regression/test.py:38-134 (restore from
the run directory, test NLL), on the offline-reproducible `syn-normal` / `syn-t` generators (data.py:219-236, seeds 829 / 761)."""
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import nngp_oracle as O  # noqa: E402  (test infrastructure only)


def _dataset(name):
    """experiments/regression/data.py:219-236 (the two generators that need no download), then the run's own
    permute_dataset(seed=10) / split_dataset(0.8, 0.1, 0.1) with train statistics."""
    if name == "syn-normal":
        num, rs = 100, np.random.RandomState(829)
        noise = lambda: rs.standard_normal(size=num) * 0.2
    else:
        num, rs = 300, np.random.RandomState(761)
        noise = lambda: rs.standard_t(df=1, size=num) * 0.8
    x = np.linspace(-num / 2, num / 2, num)[:, None]
    cov = np.exp(-0.5 * (x - x.T) ** 2)
    y = rs.multivariate_normal(mean=np.zeros(num), cov=cov, size=1).flatten() + noise()
    idx = np.random.RandomState(10).permutation(num)
    x, y = x[idx], y[idx]
    ntr, nva = int(0.8 * num), int(0.1 * num)
    xm, xs, ym, ys = x[:ntr].mean(0), x[:ntr].std(0), y[:ntr].mean(), y[:ntr].std()
    std = lambda a, b: ((a - xm) / xs, (b - ym) / ys)
    return std(x[:ntr], y[:ntr]), std(x[ntr:ntr + nva], y[ntr:ntr + nva]), std(x[ntr + nva:], y[ntr + nva:]), (ym, ys)


@pytest.mark.parametrize("data_name,method", [("syn-t", "gp"), ("syn-t", "tp"), ("syn-normal", "gp"), ("syn-normal", "tp")])
def test_regression_run_train_checkpoint_restore(tmp_path, data_name, method):
    from smnngp import checkpoint, nt_kernels, train
    from smnngp.spax.kernels import NNGPKernel
    from smnngp.spax.likelihoods import GaussianLikelihood, StudentTLikelihood
    from smnngp.spax.models import SPR
    (xtr, ytr), (xva, yva), (xte, yte), (ym, ys) = _dataset(data_name)
    args = dict(method=method, network="mlp", num_hiddens=2, activation="relu", data_name=data_name, last_w_std=1.0)

    def get_kernel_fn(w_std, b_std, last_w_std):
        return nt_kernels.get_mlp_kernel(args["num_hiddens"], act=args["activation"], w_std=w_std, b_std=b_std,
                                         last_w_std=last_w_std)

    kernel = NNGPKernel(get_kernel_fn, 1.0, 1.0, 1.0)                   # train.py defaults
    likelihood = GaussianLikelihood() if method == "gp" else StudentTLikelihood(2.0, 2.0)
    model = SPR(kernel, likelihood, xtr, ytr, ym, ys, eps=1e-2)
    model_vars = model.vars()
    train_step = train.build_train_step(model, method="analytic")
    ckpt_dir = str(tmp_path / "run")
    ck = checkpoint.Checkpointer(ckpt_dir, keep_ckpts=3)
    checkpoint.save_meta(ckpt_dir, args)
    valid0, test0 = model.test_nll(xva, yva), model.test_nll(xte, yte)
    loss0 = model.loss()
    best = (0, valid0, test0)
    ck.step(0, valid0, model_vars)
    lr = 0.05
    for i in range(1, 121):
        nll = train_step(lr)
        assert math.isfinite(nll)
        if i % 20 == 0:
            valid = model.test_nll(xva, yva)
            if ck.step(i, valid, model_vars):
                best = (i, valid, model.test_nll(xte, yte))
    assert model.loss() < loss0 - 1e-3                                   # the marginal likelihood went up
    assert best[0] > 0 and best[1] < valid0                              # and validation NLL improved on the way
    # experiments/regression/test.py: newest file of the run directory -> model -> test NLL
    restored, ctx_args = checkpoint.restore_spr(ckpt_dir, xtr, ytr, ym, ys, dtype=np.float64)
    assert ctx_args["method"] == method and checkpoint.latest_index(ckpt_dir) == best[0]
    got = restored.test_nll(xte, yte)
    assert abs(got - best[2]) < 1e-9 * max(1.0, abs(best[2]))
    # and the restored hyper-parameters evaluate to the oracle's numbers
    ws, bs, ls = restored.kernel.get_params()
    okw = dict(kernel="mlp", num_hiddens=2, act="relu", w_std=ws, b_std=bs, last_w_std=ls, eps=restored.eps.safe_value,
               method=method)
    if method == "tp":
        okw.update(alpha=restored.likelihood.a.safe_value, beta=restored.likelihood.b.safe_value)
    ref = O.spr_test_nll(xtr, ytr, xte, yte, ym, ys, **okw)
    assert abs(got - ref) < 1e-5 * max(1.0, abs(ref))
    assert sorted(os.listdir(ckpt_dir))[-1] == "meta.npy"
