"""One small conv-resnet kernel call against the oracle (first GPU contact of a new kernel: run alone)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import nngp_oracle as O
from smnngp import nt_kernels
rng = np.random.default_rng(0)
for dt in (np.float32, np.float64):
    x = rng.standard_normal((4, 8, 8, 3)).astype(dt)
    k = np.asarray(nt_kernels.get_conv_resnet_kernel(1, 10, act="relu", w_std=1.2, b_std=0.3, last_w_std=0.9)(x, None))
    ref = O.conv_resnet_kernel(x.astype(np.float64), None, 1, "relu", 1.2, 0.3, 0.9)
    print(np.dtype(dt).name, "max rel err", np.abs(k - ref).max() / np.abs(ref).max(), flush=True)
