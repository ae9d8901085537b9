#!/usr/bin/env python3
"""The reference's grid search (experiments/regression/find.py:134-199) on its offline `syn-t` data set, on the device path:
every (w_std, b_std, eps) cell of the default grid goes through two batched passes (grid.y = cell) and the Burr-XII
mixture NLL of every (cell, alpha, beta) through one kernel.  Prints the two lines find.py logs last.

    python examples/grid_search_synthetic.py [syn-t|syn-normal] [mlp|resnet]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from regression_synthetic import dataset          # noqa: E402  (the reference's generators and split)
from smnngp import sweeps                         # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "syn-t"
    network = sys.argv[2] if len(sys.argv) > 2 else "mlp"
    (x, y), _valid, (xt, yt), (y_mean, y_std) = dataset(name)
    grid = dict(w_std_list=(1.0, 1.4, 2.0), b_std_list=(0.0, 0.3, 1.0), eps_list=tuple(float("1e%d" % v) for v in range(-6, 5)),
                alpha_list=(1.0, 2.0, 3.0), beta_list=(1.0, 2.0, 3.0))          # find.py:18-22
    t0 = time.perf_counter()
    out = sweeps.find_grid(x, y, xt, yt, y_mean, y_std, network=network, num_hiddens=4, activation="relu", **grid)
    dt = time.perf_counter() - t0
    cells = out["gnll"].size
    print("%d cells x %d mixtures in %.1f ms" % (cells, out["tnll"].size // cells, dt * 1e3))
    print("(%s): %.4f" % (out["best_student"][0], out["best_student"][1]))       # find.py:196-197
    print("(%s): %.4f" % (out["best_gaussian"][0], out["best_gaussian"][1]))


if __name__ == "__main__":
    main()
