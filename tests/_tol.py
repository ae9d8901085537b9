"""The error metric of the parity tests.

`relerr(a, b)` is the LARGER of two readings, so one assert of `relerr(a, b) < rtol` states both:

  norm-wise      max|a - b| / max|b|
  element-wise   max_ij |a_ij - b_ij| / (|b_ij| + FLOOR * max|b|)
                 i.e. the usual  |a - b| <= rtol * |b| + atol  with  atol = rtol * FLOOR * max|b|.

FLOOR = 1e-3: entries down to a thousandth of the largest one are held to the RELATIVE tolerance of the test
(2e-3 in fp32, 1e-8 in fp64 on kernels), smaller ones (erf kernels near zero, NTK off-diagonals, cross kernels) to an
absolute tolerance a thousand times tighter than the norm-wise reading alone would allow.  A pure relative test is not
meaningful below that: an fp32 kernel entry of 1e-6 beside O(1) intermediates carries an absolute rounding error of
~1e-7 whatever the implementation.
"""
import numpy as np

FLOOR = 1e-3


def relerr(a, b, floor=FLOOR):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        raise AssertionError("shape mismatch %s vs %s" % (a.shape, b.shape))
    if b.size == 0:
        return 0.0
    scale = max(float(np.max(np.abs(b))), 1e-300)
    diff = np.abs(a - b)
    if not np.isfinite(diff).all():
        return float("inf")
    norm = float(np.max(diff)) / scale
    elem = float(np.max(diff / (np.abs(b) + floor * scale)))
    return max(norm, elem)


def relerr_norm(a, b):
    """The norm-wise reading alone, for the OUTPUTS OF SOLVES (Cholesky factors, triangular solves, posterior means and
    covariances, Schur complements): a backward-stable factorisation bounds max|a - b| / max|b| by cond * unit roundoff,
    but an entry that is small because large terms cancel (a posterior mean near zero) carries the same ABSOLUTE error
    as its neighbours -- LAPACK's own fp32 result fails an element-wise relative test there.  Kernel entries are
    computed entry by entry and take `relerr`."""
    return relerr(a, b, floor=1.0)
