"""Edge case: duplicated input rows (correlation exactly 1 off the diagonal)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nngp_oracle as O
from smnngp import nt_kernels
rng = np.random.default_rng(0)
x = rng.standard_normal((40, 7)); x[5] = x[2]; x[39] = x[2]
for dt in (np.float64, np.float32):
    for act in ("relu", "erf"):
        kfn = nt_kernels.get_mlp_kernel(4, act=act, w_std=1.3, b_std=0.2, last_w_std=1.0)
        rk, rt = O.mlp_kernel(x, None, 4, act, 1.3, 0.2, 1.0, ("nngp", "ntk"))
        g = kfn(x.astype(dt), None, get=("nngp", "ntk"))
        k = np.asarray(g.nngp, np.float64); t = np.asarray(g.ntk, np.float64)
        print(np.dtype(dt).name, act, "K[5,2]-K[2,2]: hip %.3e oracle %.3e | T[5,2]-T[2,2]: hip %.3e oracle %.3e | max rel err K %.2e T %.2e" % (
            k[5, 2] - k[2, 2], rk[5, 2] - rk[2, 2], t[5, 2] - t[2, 2], rt[5, 2] - rt[2, 2],
            np.abs(k - rk).max() / np.abs(rk).max(), np.abs(t - rt).max() / np.abs(rt).max()), flush=True)
