"""Coefficients of the single-sqrt f64 form of J(c) = sqrt(1-c^2) + (pi - acos c) c used by nngp_math.hpp:
    J(c) = (pi/2)(c + |c|) + d^(3/2) R(d),   d = 1 - |c|,
    R(d) = (sqrt(2 - d) - (1 - d) acos(1 - d)/sqrt(d)) / d      (analytic on [0, 1]; nearest singularity d = 2)
Chebyshev interpolation of R on [0, 1] at 60 digits (mpmath), converted to monomials in d (terms decay like 2^-k, so
Horner in d is well conditioned), then the f64 Horner form is checked against the exact J on 2e5 points."""
import sys
import mpmath as mp
import numpy as np

mp.mp.dps = 60
DEG = int(sys.argv[1]) if len(sys.argv) > 1 else 22


def R(d):
    d = mp.mpf(d)
    if d < mp.mpf("1e-20"):
        return 2 * mp.sqrt(2) / 3
    a = 1 - d
    return (mp.sqrt(1 + a) - a * mp.acos(a) / mp.sqrt(d)) / d


def J(c):
    c = mp.mpf(c)
    return mp.sqrt(1 - c * c) + (mp.pi - mp.acos(c)) * c


n = DEG + 1
nodes = [(mp.cos(mp.pi * (2 * k + 1) / (2 * n)) + 1) / 2 for k in range(n)]   # Chebyshev nodes on [0, 1]
vals = [R(x) for x in nodes]
# monomial coefficients through the Vandermonde system at 60 digits
A = mp.matrix(n, n)
for i, x in enumerate(nodes):
    for j in range(n):
        A[i, j] = x ** j
coef = mp.lu_solve(A, mp.matrix(vals))
c64 = np.array([float(coef[j]) for j in range(n)])


def j_f64(c):
    a = np.abs(c)
    d = 1.0 - a
    s = np.sqrt(d)
    r = np.full_like(c, c64[-1])
    for k in range(n - 2, -1, -1):
        r = r * d + c64[k]           # numpy has no fma: the GPU form is at least this accurate
    return (d * s) * r + (np.pi / 2) * (c + a)


rng = np.random.default_rng(0)
cs = np.concatenate([rng.uniform(-1, 1, 200000), 1 - np.logspace(-16, 0, 2000), -1 + np.logspace(-16, 0, 2000), [0.0, 1.0, -1.0]])
exact = np.array([float(J(float(c))) for c in cs[::20]])
got = j_f64(cs[::20])
err = np.abs(got - exact)
print("degree", DEG, "max |J - exact| =", err.max(), " max relative to max(J, 1e-300):", (err / np.maximum(exact, 1e-300))[exact > 1e-12].max())
print("coefficients (d^0 ... d^%d):" % DEG)
for v in c64:
    print("  %.17e," % v)
