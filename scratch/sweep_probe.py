"""find_grid throughput on a UCI-sized problem: 1 vs 4 workers."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smnngp import sweeps
rng = np.random.default_rng(0)
n, t, d = 1000, 120, 8
x = rng.standard_normal((n, d)); y = rng.standard_normal(n); xt = rng.standard_normal((t, d)); yt = rng.standard_normal(t)
kw = dict(network="mlp", num_hiddens=4, activation="relu", w_std_list=(1.0, 1.2, 1.4, 1.7, 2.0), b_std_list=(0.0, 0.1, 0.3, 0.6, 1.0),
          eps_list=(1e-6, 1e-4, 1e-2), alpha_list=(1.0, 2.0, 3.0), beta_list=(1.0, 2.0, 3.0))
sweeps.find_grid(x, y, xt, yt, **kw)
for w in (1, 2, 4, 8):
    t0 = time.perf_counter(); r = sweeps.find_grid(x, y, xt, yt, workers=w, **kw); dt = time.perf_counter() - t0
    print("workers=%d: %.3f s for %d (w,b) x %d eps x 9 (alpha,beta); best %s" % (w, dt, 25, 3, r["best_student"][0]), flush=True)
