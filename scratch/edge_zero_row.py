"""Edge case: an all-zero input row with b_std = 0 (q = 0 at every layer for that row)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nngp_oracle as O
from smnngp import nt_kernels
rng = np.random.default_rng(0)
x = rng.standard_normal((9, 5)); x[3] = 0.0
x2 = rng.standard_normal((4, 5)); x2[1] = 0.0
for dt in (np.float64, np.float32):
    for net, fac, of in (("mlp", nt_kernels.get_mlp_kernel, O.mlp_kernel), ("resnet", nt_kernels.get_dense_resnet_kernel, O.dense_resnet_kernel)):
        for act in ("relu", "erf"):
            kfn = fac(3, act=act, w_std=1.3, b_std=0.0, last_w_std=1.0)
            with np.errstate(all="ignore"):
                rk, rt = of(x, None, 3, act, 1.3, 0.0, 1.0, ("nngp", "ntk"))
                ck, ct = of(x, x2, 3, act, 1.3, 0.0, 1.0, ("nngp", "ntk"))
            g = kfn(x.astype(dt), None, get=("nngp", "ntk")); gc = kfn(x.astype(dt), x2.astype(dt), get=("nngp", "ntk"))
            def cmp(a, b):
                a = np.asarray(a, np.float64)
                fin = np.isfinite(b)
                return ("nan_in_hip=%d nan_in_oracle=%d maxerr=%.2e" % (int((~np.isfinite(a)).sum()), int((~fin).sum()),
                        np.abs(a[fin] - b[fin]).max() / max(np.abs(b[fin]).max(), 1e-300)))
            print(np.dtype(dt).name, net, act, "| nngp", cmp(g.nngp, rk), "| ntk", cmp(g.ntk, rt), "| cross nngp", cmp(gc.nngp, ck), "| cross ntk", cmp(gc.ntk, ct), flush=True)
