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
        elif kind == "cols":
            # cyclic column-first shard, `world` ranks played on one GPU, random (also unaligned) column ranges, random arrival order
            import ctypes as C
            from smnngp import _lib as LL, sharding as S
            ctx = LL.default_context()
            n, d, world = int(rng.integers(2, 1800)), int(rng.integers(1, 40)), int(rng.integers(1, 9))
            t_all = S.tile_rows(n)
            cuts = sorted(set(int(v) for v in rng.integers(1, max(2, t_all), size=int(rng.integers(0, 6))) if 0 < v < t_all))
            cols = [0] + cuts + [t_all]
            case = (kind, dt.__name__, net, act, L, n, d, world, cols, w, b, lw)
            xh = rng.standard_normal((n, d)).astype(dt); yh = rng.standard_normal(n).astype(dt)
            x, y = ctx.to_device(xh), ctx.to_device(yh)
            code, es = LL.dtype_code(dt), np.dtype(dt).itemsize
            netc = LL.NET_MLP if net == "mlp" else LL.NET_DENSE_RESNET
            lay = S.col_layout(n, world, cols)
            stage = ctx.to_device(np.full(world * lay["elems"], np.nan, dt))
            ca = S.cols_array(cols)
            for r in range(world):
                mine = ctx.to_device(np.full(lay["elems"], np.nan, dt))
                ctx.call("smn_kernel_mlp_shard_cols", code, netc, LL.ACT[act], L, w, b, lw, x.ptr, n, d, d, world, r, len(cols) - 1, ca, 1, mine.ptr, None)
                for g in range(len(cols) - 1):
                    ctx.call("smn_memcpy_d2d", C.c_void_p(stage.ptr.value + es * (world * lay["off"][g] + r * lay["count"][g])),
                             C.c_void_p(mine.ptr.value + es * lay["off"][g]), es * lay["count"][g])
                ctx.synchronize()
                del mine
            eps = 1e-2 if dt == np.float32 else 1e-6
            lp, quad, logdet, info = C.c_double(), C.c_double(), C.c_double(), C.c_int()
            ctx.call("smn_spr_loss", code, netc, LL.ACT[act], L, w, b, lw, x.ptr, n, d, d, y.ptr, eps, 0.0, 1.0, C.byref(lp), C.byref(quad), C.byref(logdet), C.byref(info))
            want = (lp.value, logdet.value, info.value)
            ctx.call("smn_shard_begin", code, n, eps)
            for g in rng.permutation(len(cols) - 1):
                ctx.call("smn_shard_scatter_cols", code, stage.ptr, n, world, len(cols) - 1, ca, int(g), None, 0)
            ctx.call("smn_lml_from_shards", code, n, y.ptr, 0.0, 1.0, C.byref(lp), C.byref(quad), C.byref(logdet), C.byref(info))
            got = (lp.value, logdet.value, info.value)
            same = got == want or (np.isnan(got[0]) and np.isnan(want[0]) and got[2] == want[2])
            errs = [0.0 if same else 1.0]
        elif kind == "batch":
            import ctypes as C
            from smnngp import _lib as LL, sweeps
            ctx = LL.default_context()
            n, d, t, g = int(rng.integers(1, 900)), int(rng.integers(1, 30)), int(rng.integers(1, 70)), int(rng.integers(1, 9))
            case = (kind, dt.__name__, net, act, L, n, d, t, g)
            x = ctx.to_device(rng.standard_normal((n, d)).astype(dt)); y = ctx.to_device(rng.standard_normal((n, 1)).astype(dt))
            xt = ctx.to_device(rng.standard_normal((t, d)).astype(dt))
            ws, bs, lws = rng.uniform(0.5, 2.0, g), rng.choice([0.0, 1e-8, 0.1, 0.5, 1.0], g), rng.uniform(0.5, 1.5, g)
            eps = rng.choice([1e-3, 1e-2, 1e-1], g); dfs = rng.choice([0.0, 3.0, 5.0], g); scs = rng.uniform(0.5, 2.0, g)
            code = LL.dtype_code(dt); netc = LL.NET_MLP if net == "mlp" else LL.NET_DENSE_RESNET
            lp, quad, logdet, info = C.c_double(), C.c_double(), C.c_double(), C.c_int()
            mean_d, cov_d = ctx.empty((t, 1), dt), ctx.empty((t, t), dt)
            kw = dict(network=net, num_hiddens=L, activation=act, w_std=ws, b_std=bs, last_w_std=lws)
            gl = sweeps.loss_batch(ctx, x, y, eps=eps, df=dfs, scale=scs, **kw)
            gm, gc, gi = sweeps.predict_batch(ctx, x, y, xt, diag_reg=eps, full_cov=True, **kw)
            mism = 0
            for i in range(g):
                ctx.call("smn_spr_loss", code, netc, LL.ACT[act], L, ws[i], bs[i], lws[i], x.ptr, n, d, d, y.ptr, eps[i], dfs[i], scs[i],
                         C.byref(lp), C.byref(quad), C.byref(logdet), C.byref(info))
                if info.value != gl[3][i] or (info.value == 0 and (lp.value, quad.value, logdet.value) != (gl[0][i], gl[1][i], gl[2][i])):
                    mism += 1
                ctx.call("smn_spr_predict", code, netc, LL.ACT[act], L, ws[i], bs[i], lws[i], x.ptr, n, d, xt.ptr, t, d, d, y.ptr, 1, eps[i], 0.0,
                         mean_d.ptr, cov_d.ptr, t, None, None, C.byref(info))
                if info.value != gi[i] or (info.value == 0 and not (np.array_equal(mean_d.numpy(), gm[i]) and np.array_equal(cov_d.numpy(), gc[i]))):
                    mism += 1
            errs = [float(mism)]
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
