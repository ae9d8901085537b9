"""Randomised parity sweep (not a test: a bug hunt).  Random shapes / hyper-parameters / dtypes through the public
Python surface, every result against the oracle.  Prints only disagreements and a final tally."""
import os, sys, time, traceback
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nngp_oracle as O
from smnngp import nt_kernels, predict
from smnngp.spax.kernels import NNGPKernel
from smnngp.spax.likelihoods import GaussianLikelihood, StudentTLikelihood
from smnngp.spax.models import SPR

KINDS = sys.argv[3].split(",") if len(sys.argv) > 3 else ["kernel", "kernel", "heads", "grad", "cnn", "chol", "predict"]
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
rng = np.random.default_rng(seed)
tol = {np.float32: 3e-3, np.float64: 2e-8}
bad = 0; done = 0
def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)
t_end = time.time() + budget
while time.time() < t_end:
    case = None
    try:
        dt = [np.float32, np.float64][rng.integers(2)]
        kind = rng.choice(KINDS)
        act = ["relu", "erf"][rng.integers(2)]
        net = ["mlp", "resnet"][rng.integers(2)]
        L = int(rng.integers(1, 6))
        w, b, lw = float(rng.uniform(0.5, 2.0)), float(rng.choice([0.0, 1e-8, 0.1, 0.5, 1.0])), float(rng.uniform(0.5, 1.5))
        fac = nt_kernels.get_mlp_kernel if net == "mlp" else nt_kernels.get_dense_resnet_kernel
        ofn = O.mlp_kernel if net == "mlp" else O.dense_resnet_kernel
        if kind == "shard":
            import ctypes as C
            from smnngp import _lib as LL, sharding as S
            ctx = LL.default_context()
            n, d, world = int(rng.integers(1, 1800)), int(rng.integers(1, 60)), int(rng.integers(1, 9))
            case = (kind, dt.__name__, net, act, L, n, d, world, w, b, lw)
            xh = rng.standard_normal((n, d)).astype(dt)
            x = ctx.to_device(xh)
            code, es = LL.dtype_code(dt), np.dtype(dt).itemsize
            chunk, h = S.paired_chunk_elems(n, world), S.block_rows(n, world)
            stage = ctx.to_device(np.full(world * chunk, np.nan, dt)); st2 = ctx.to_device(np.full(world * chunk, np.nan, dt))
            netc = LL.NET_MLP if net == "mlp" else LL.NET_DENSE_RESNET
            for r in range(world):
                ctx.call("smn_kernel_mlp_shard", code, netc, LL.ACT[act], L, w, b, lw, x.ptr, n, d, d, world, r, h, 3,
                         C.c_void_p(stage.ptr.value + r * chunk * es), C.c_void_p(st2.ptr.value + r * chunk * es))
            k = ctx.to_device(np.zeros((n, n), dt)); t2 = ctx.to_device(np.zeros((n, n), dt))
            ctx.call("smn_unpack_lower_blocks", code, stage.ptr, n, world, h, k.ptr, n)
            ctx.call("smn_unpack_lower_blocks", code, st2.ptr, n, world, h, t2.ptr, n)
            rk, rt = ofn(xh.astype(np.float64), None, L, act, w, b, lw, ("nngp", "ntk"))
            il = np.tril_indices(n)
            errs = [rel(k.numpy()[il], rk[il]), rel(t2.numpy()[il], rt[il]) / 5]
        elif kind == "chol":
            import ctypes as C, scipy.linalg as sla
            from smnngp import _lib as LL
            ctx = LL.default_context()
            n, m = int(rng.integers(1, 1400)), int(rng.integers(0, 260))
            case = (kind, dt.__name__, n, m)
            g = rng.standard_normal((n + m, max(4, (n + m) // 3)))
            a = g @ g.T / g.shape[1] + np.diag(rng.uniform(0.5, 1.5, n + m))
            ad = ctx.to_device(a.astype(dt))
            info, logdet = C.c_int(), C.c_double()
            ctx.call("smn_cholesky", LL.dtype_code(dt), ad.ptr, n + m, n, n + m, 0, 0.0, 0.0, C.byref(info), C.byref(logdet))
            gotm = ad.numpy().astype(np.float64)
            l = np.linalg.cholesky(a[:n, :n])
            errs = [rel(np.tril(gotm[:n, :n]), l), abs(logdet.value - 2 * np.log(np.diag(l)).sum()) / max(1.0, abs(logdet.value)), float(info.value)]
            if m:
                wmat = sla.solve_triangular(l, a[:n, n:], lower=True).T
                errs += [rel(gotm[n:, :n], wmat), rel(np.tril(gotm[n:, n:]), np.tril(a[n:, n:] - wmat @ wmat.T))]
            # trsm both ways on a few right-hand sides
            r = int(rng.integers(1, 40))
            bmat = rng.standard_normal((n, r))
            ld = ctx.to_device(l.astype(dt)); bd = ctx.to_device(bmat.astype(dt))
            ctx.call("smn_trsm", LL.dtype_code(dt), ld.ptr, n, n, bd.ptr, r, r, 0)
            ctx.call("smn_trsm", LL.dtype_code(dt), ld.ptr, n, n, bd.ptr, r, r, 1)
            errs.append(rel(bd.numpy(), sla.cho_solve((l, True), bmat)) / 10)
        elif kind == "predict":
            n, t, d, c = int(rng.integers(2, 500)), int(rng.integers(1, 200)), int(rng.integers(1, 30)), int(rng.integers(1, 5))
            eps = float(rng.choice([1e-3, 1e-2, 1e-1])); b = max(b, 0.1)
            case = (kind, dt.__name__, net, act, L, n, t, d, c, w, b, lw, eps)
            x = rng.standard_normal((n, d)); y = rng.standard_normal((n, c)); xt = rng.standard_normal((t, d))
            kfn = fac(L, act=act, w_std=w, b_std=b, last_w_std=lw)
            pf = predict.gradient_descent_mse_ensemble(kfn, x.astype(dt), y.astype(dt), diag_reg=eps)
            mean, cov = pf(x_test=xt.astype(dt))
            kw = (L, act, w, b, lw)
            rm, rc = O.predict(ofn(x, None, *kw), ofn(xt, x, *kw), ofn(xt, None, *kw), y, diag_reg=eps)
            scale = 50 if dt == np.float32 else 1e3
            errs = [rel(mean, rm) / scale, rel(cov, rc) / scale]
        elif kind == "kernel":
            n, m, d = int(rng.integers(1, 600)), int(rng.integers(1, 300)), int(rng.integers(1, 200))
            case = (kind, dt.__name__, net, act, L, n, m, d, w, b, lw)
            x = rng.standard_normal((n, d)).astype(dt); x2 = rng.standard_normal((m, d)).astype(dt)
            kfn = fac(L, act=act, w_std=w, b_std=b, last_w_std=lw)
            got = kfn(x, None, get=("nngp", "ntk")); gc = kfn(x, x2, get=("nngp", "ntk"))
            rk, rt = ofn(x.astype(np.float64), None, L, act, w, b, lw, ("nngp", "ntk"))
            ck, ct = ofn(x.astype(np.float64), x2.astype(np.float64), L, act, w, b, lw, ("nngp", "ntk"))
            errs = [rel(got.nngp, rk), rel(got.ntk, rt) / 5, rel(gc.nngp, ck), rel(gc.ntk, ct) / 5]
        elif kind == "cnn":
            n, m = int(rng.integers(1, 24)), int(rng.integers(1, 12))
            h, wd, c = int(rng.integers(1, 20)), int(rng.integers(1, 20)), int(rng.integers(1, 5))
            resn = rng.integers(2) == 1
            if not resn and rng.integers(3) == 0:   # r03: the 32x32 kernels (4x4 patch per lane for 1 / 3 channels, row pairs otherwise)
                h = wd = 32; n, m = int(rng.integers(1, 9)), int(rng.integers(1, 6)); L = int(rng.integers(1, 5))
            if resn:
                h, wd = 8 * int(rng.integers(1, 4)), 8 * int(rng.integers(1, 4)); L = int(rng.integers(1, 3))
            case = (kind, dt.__name__, "resnet" if resn else "cnn", act, L, n, m, h, wd, c, w, b, lw)
            x = rng.standard_normal((n, h, wd, c)).astype(dt); x2 = rng.standard_normal((m, h, wd, c)).astype(dt)
            kfn = (nt_kernels.get_conv_resnet_kernel(L, 10, act=act, w_std=w, b_std=b, last_w_std=lw) if resn
                   else nt_kernels.get_cnn_kernel(L, act=act, w_std=w, b_std=b, last_w_std=lw))
            of = O.conv_resnet_kernel if resn else O.cnn_kernel
            errs = [rel(kfn(x, None), of(x.astype(np.float64), None, L, act, w, b, lw)),
                    rel(kfn(x, x2), of(x.astype(np.float64), x2.astype(np.float64), L, act, w, b, lw))]
        else:
            n, t, d = int(rng.integers(2, 500)), int(rng.integers(1, 80)), int(rng.integers(1, 40))
            method = ["gp", "tp"][rng.integers(2)]
            eps = float(rng.choice([1e-3, 1e-2, 1e-1])); al, be = float(rng.uniform(0.8, 3)), float(rng.uniform(0.8, 3))
            b = max(b, 0.1)                      # keep the systems reasonably conditioned for the tolerance used
            case = (kind, dt.__name__, net, act, L, n, t, d, w, b, lw, method, eps, al, be)
            x = rng.standard_normal((n, d)); y = rng.standard_normal(n); xt = rng.standard_normal((t, d)); yt = rng.standard_normal(t)
            kernel = NNGPKernel(lambda a1, a2, a3: fac(L, act=act, w_std=a1, b_std=a2, last_w_std=a3), w, b, lw)
            lik = GaussianLikelihood() if method == "gp" else StudentTLikelihood(al, be)
            model = SPR(kernel, lik, x.astype(dt), y.astype(dt), 0.1, 1.3, eps=eps)
            okw = dict(kernel=net, num_hiddens=L, act=act, w_std=w, b_std=b, last_w_std=lw, eps=eps, method=method, alpha=al, beta=be)
            if kind == "heads":
                rl = O.spr_loss(x, y, **okw); rn = O.spr_test_nll(x, y, xt, yt, 0.1, 1.3, **okw)
                scale = 50 if dt == np.float32 else 1e3          # loss / nll amplify kernel error by the conditioning
                errs = [abs(model.loss() - rl) / max(1, abs(rl)) / scale, abs(model.test_nll(xt.astype(dt), yt.astype(dt)) - rn) / max(1, abs(rn)) / scale]
            else:
                loss, grads = model.loss_and_grad()
                keys = ("w_std", "b_std", "last_w_std", "eps") + (("alpha", "beta") if method == "tp" else ())
                ref = O.spr_loss_grad_fd(x, y, keys=keys, **okw)
                vmap = {"w_std": kernel.w_std, "b_std": kernel.b_std, "last_w_std": kernel.last_w_std, "eps": model.eps}
                if method == "tp":
                    vmap.update(alpha=lik.a, beta=lik.b)
                names = {id(v): k for k, v in model.vars().items()}
                sc = max(abs(v) for v in ref.values())
                scale = 20 if dt == np.float32 else 500
                errs = [abs(grads[names[id(vmap[k])]] / float(vmap[k].constraint.grad(vmap[k].value)) - ref[k]) / max(sc, 1e-12) / scale for k in keys]
        done += 1
        worst = max(errs)
        if not np.isfinite(worst) or worst > tol[dt]:
            bad += 1
            print("MISMATCH", case, ["%.2e" % e for e in errs], flush=True)
    except Exception as e:
        bad += 1
        print("EXCEPTION", case, repr(e), flush=True)
        traceback.print_exc()
print("fuzz seed %d: %d cases, %d flagged" % (seed, done, bad), flush=True)
