import numpy as np, sys
sys.path.insert(0, '.')
from oracle import nngp_oracle as O
from smnngp import nt_kernels
rng = np.random.default_rng(43)
x = rng.standard_normal((33, 6))
for act in ("relu","erf"):
    kfn = nt_kernels.get_mlp_kernel(2, act=act, w_std=1.4, b_std=0.3, last_w_std=0.8)
    got = kfn(x, None, get=("nngp","ntk"))
    rk, rt = O.mlp_kernel(x, None, 2, act, 1.4, 0.3, 0.8, ("nngp","ntk"))
    k, t = np.asarray(got.nngp), np.asarray(got.ntk)
    ek = np.abs(k-rk)/np.abs(rk).max(); et = np.abs(t-rt)/np.abs(rt).max()
    print(act, "nngp max", ek.max(), "at", np.unravel_index(ek.argmax(), ek.shape), "offdiag max", (ek-np.diag(np.diag(ek))).max())
    print(act, "ntk  max", et.max(), "at", np.unravel_index(et.argmax(), et.shape), "offdiag max", (et-np.diag(np.diag(et))).max())
    i,j = np.unravel_index(et.argmax(), et.shape)
    print("  got %.15g ref %.15g" % (t[i,j], rt[i,j]))
