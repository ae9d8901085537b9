// nngp_math.hpp — the per-element nonlinear maps of the NNGP / NTK layer recursion.
//
// Math (neural_tangents stax.Relu / stax.Erf kernel transforms composed
// by experiments/nt_kernels.py:21-31):
//   ReLU:  c = K / sqrt(q_i q_j);  K' = sqrt(q_i q_j)/(2 pi) * J(c),  J(c) = sqrt(1-c^2) + (pi - acos c) c
//          Kdot = (pi - acos c) / (2 pi)
//   Erf:   c = 2K / sqrt((1+2q_i)(1+2q_j));  K' = (2/pi) asin c;  Kdot = 4 / (pi sqrt((1+2q_i)(1+2q_j) - 4K^2))
// The per-row factors r_i, s_i are precomputed per layer (diag_tables kernel), so per element:
//   ReLU:  c = clamp(K r_i r_j),  K' = s_i s_j J(c)          r = 1/sqrt(q),  s = sqrt(q / (2 pi))
//   Erf:   c = clamp(2 K r_i r_j), K' = (2/pi) asin(c)        r = 1/sqrt(1+2q)
// J is written with asin so that one odd/even split serves both signs without cancellation:
//   J(c) = (pi/2) c + sqrt(1-c^2) + |c| asin|c|,   pi - acos c = pi/2 + asin c.
#pragma once
#include <hip/hip_runtime.h>

namespace nngp {

constexpr double kPi = 3.14159265358979323846;

template <typename T>
struct ActOut {
  T k;     // K'
  T kdot;  // Kdot (only meaningful when requested)
};

// asin(|c|) for |c| <= 1, f32: fdlibm-style split, branch-free, ~1.2 ulp.
//   |c| <= 0.5 : asin(a) = a + a t P(t), t = a^2
//   |c| >  0.5 : asin(a) = pi/2 - 2 asin(s), s = sqrt((1-a)/2), same polynomial with t = s^2
// P fitted on [0, 0.25] (minimax-refined least squares), max rel. err 6.8e-8 in f32 arithmetic.
__device__ __forceinline__ float asin_abs(float a, float c2) {
  const float z = fmaf(-0.5f, a, 0.5f);
  const float s = __builtin_amdgcn_sqrtf(z);
  const bool big = a > 0.5f;
  const float t = big ? z : c2;
  const float p = big ? s : a;
  float r = 0.041802484542131424f;
  r = fmaf(r, t, 0.02439034730195999f);
  r = fmaf(r, t, 0.04542740806937218f);
  r = fmaf(r, t, 0.07495657354593277f);
  r = fmaf(r, t, 0.1666674166917801f);
  r = fmaf(p * t, r, p);
  return big ? fmaf(-2.0f, r, 1.57079632679489662f) : r;
}
__device__ __forceinline__ double fast_sqrt(double x);   // below

// asin(|c|) in f64: the same split, branch-free (libm's asin takes BOTH of its branches in a wave whose lanes
// straddle 0.5, which they always do here), with the classic rational approximation of asin on [0, 0.5]
//   asin(p) = p + p t P5(t)/Q4(t),  t = p^2        (coefficients: Sun fdlibm e_asin.c, |error| < 1 ulp there)
// and a Newton-refined v_rcp_f64 for the one division.  The same formula in NumPy fp64 (true division) against
// numpy.arcsin over 2e6 points of [0, 1]: |error| <= 2.3e-16 absolute, 4.3e-16 relative.
__device__ __forceinline__ double asin_abs(double a, double c2) {
  const double z = fma(-0.5, a, 0.5);
  const double s = fast_sqrt(z);
  const bool big = a > 0.5;
  const double t = big ? z : c2;
  const double p = big ? s : a;
  double pn = 3.47933107596021167570e-05;
  pn = fma(pn, t, 7.91534994289814532176e-04);
  pn = fma(pn, t, -4.00555345006794114027e-02);
  pn = fma(pn, t, 2.01212532134862925881e-01);
  pn = fma(pn, t, -3.25565818622400915405e-01);
  pn = fma(pn, t, 1.66666666666666657415e-01);
  pn *= t;
  double qn = 7.70381505559019352791e-02;
  qn = fma(qn, t, -6.88283971605453293030e-01);
  qn = fma(qn, t, 2.02094576023350569471e+00);
  qn = fma(qn, t, -2.40339491173441421878e+00);
  qn = fma(qn, t, 1.0);
  double r = __builtin_amdgcn_rcp(qn);
  r = fma(fma(-qn, r, 1.0), r, r);
  r = fma(fma(-qn, r, 1.0), r, r);
  const double v = fma(p, pn * r, p);
  return big ? fma(-2.0, v, 1.57079632679489661923) : v;
}

__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
// f64 square roots of arguments in [0, 4] (1 - c^2, (1 - |c|)/2, ...): v_rsq_f64 seed + one coupled Goldschmidt
// step + one residual correction.  libm's sqrt spends twice the instructions on range scaling and special cases
// that cannot occur here; the zero argument (c = +-1) is handled explicitly.
__device__ __forceinline__ double fast_sqrt(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  const double d = fma(-g, g, x);
  g = fma(d, h, g);
  return x > 0.0 ? g : 0.0;
}
__device__ __forceinline__ double fast_rsqrt(double x) {   // +inf at 0, as 1/sqrt(0)
  double y = __builtin_amdgcn_rsq(x);
  const double e = fma(-x * y, y, 1.0);                     // 1 - x y^2
  y = fma(y * e, fma(0.375, e, 0.5), y);                    // y (1 + e/2 + 3 e^2/8)
  const double e2 = fma(-x * y, y, 1.0);
  y = fma(0.5 * y, e2, y);
  return x > 0.0 ? y : (1.0 / 0.0);
}

template <typename T>
__device__ __forceinline__ T clamp1(T c) {
  return fmin(fmax(c, T(-1)), T(1));
}

// J(c) = sqrt(1-c^2) + (pi - acos c) c in f64 with ONE square root and no division (NNGP-only paths):
//   J(c) = (pi/2)(c + |c|) + d^(3/2) R(d),  d = 1 - |c|,  R(d) = (sqrt(2-d) - (1-d) acos(1-d)/sqrt(d)) / d
// R is analytic on [0, 1] (nearest singularity d = 2, so the monomial terms decay like 2^-k and Horner in d is well
// conditioned).  Degree-17 Chebyshev interpolant computed at 60 digits (scratch/relu_j_f64/fit.py); the f64 Horner form
// against the exact J on 2e5 points of [-1, 1] (+ both end regions down to 1e-16): |error| <= 4.4e-16 (one ulp of pi),
// relative error <= 3.9e-16 also where J -> 0 (c -> -1), because the form is a product there.  35 f64 instructions
// against ~60 for the asin form above (two square roots, a rational and a Newton-refined reciprocal).
__device__ __forceinline__ double relu_j_f64(double c) {
  const double a = fabs(c);
  const double d = 1.0 - a;
  // sqrt(d) for d in [0, 1]: v_rsq_f64 seed + one coupled Goldschmidt step + one residual correction; the argument
  // is kept off zero so no select is needed (d = 0 gives d * s = 0 below all the same)
  const double dm = fmax(d, 1e-290);
  const double y = __builtin_amdgcn_rsq(dm);
  double g = dm * y, h = 0.5 * y;
  const double e = fma(-h, g, 0.5);
  g = fma(g, e, g);
  h = fma(h, e, h);
  g = fma(fma(-g, g, dm), h, g);
  double r = 3.06589867858277351e-07;
  r = fma(r, d, -2.07641508550871668e-06);
  r = fma(r, d, 6.76505315467904507e-06);
  r = fma(r, d, -1.34781210293057574e-05);
  r = fma(r, d, 1.85191488354136423e-05);
  r = fma(r, d, -1.79489152436826625e-05);
  r = fma(r, d, 1.35345338816749084e-05);
  r = fma(r, d, -6.11286061738224154e-06);
  r = fma(r, d, 5.46578806810183682e-06);
  r = fma(r, d, 5.83672864724229018e-06);
  r = fma(r, d, 1.83491268408970555e-05);
  r = fma(r, d, 5.10942545148346727e-05);
  r = fma(r, d, 1.52114199372216396e-04);
  r = fma(r, d, 4.88256075393192783e-04);
  r = fma(r, d, 1.75373706836867010e-03);
  r = fma(r, d, 7.57614408386195464e-03);
  r = fma(r, d, 4.71404520791057768e-02);
  r = fma(r, d, 9.42809041582063356e-01);
  return fma(d * g, r, 1.57079632679489661923 * (c + a));
}

__device__ __forceinline__ float relu_j_fast(float c);   // the f32 counterpart, below

// ReLU map.  kt = pre-activation covariance, rr = r_i r_j, ss = s_i s_j.
template <typename T, bool WANT_DOT>
__device__ __forceinline__ ActOut<T> relu_map(T kt, T rr, T ss) {
  const T c = clamp1(kt * rr);
  if constexpr (!WANT_DOT) {   // NNGP only: the single-sqrt forms of J
    ActOut<T> o;
    if constexpr (sizeof(T) == 8) o.k = ss * relu_j_f64(c);
    else o.k = ss * (3.14159265358979323846f * relu_j_fast(c));
    o.kdot = T(0);
    return o;
  }
  const T a = fabs(c);
  const T c2 = c * c;
  const T as = asin_abs(a, c2);
  const T sq = fast_sqrt(fma(-c, c, T(1)));
  const T j = fma(T(kPi / 2), c, fma(a, as, sq));
  ActOut<T> o;
  o.k = ss * j;
  if (WANT_DOT) o.kdot = fma(copysign(as, c), T(1.0 / (2.0 * kPi)), T(0.25));
  return o;
}

// J(c) = sqrt(1-c^2) + (pi - acos c) c for the f32 NNGP-only fast path, ONE transcendental:
//   J(c) = (pi/2)(c + |c|) + (1-|c|)^(3/2) R(|c|),   R(a) = (sqrt(1+a) - a acos(a)/sqrt(1-a)) / (1-a)
// R is analytic on [0,1]; the degree-5 fit below (minimax-refined least squares) gives |J - exact| <=
// 3.2e-7 over [-1,1] in f32 arithmetic (1.0e-7 relative to J's range pi; the last ulp of pi is 2.4e-7).
// J is 1-Lipschitz-ish (J' = pi - acos c <= pi), so this error does not amplify through the layers.
__device__ __forceinline__ float clamp01(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); }   // folds into the producer's clamp modifier
// Returns J(c) / pi (the factor pi rides in the next layer's table product: diag_tables_kernel).  c is NOT clamped on entry: a
// product that rounds to |c| = 1 + 1e-7 only has to keep the square root real, which the [0, 1] clamp of 1 - |c| does for free.
__device__ __forceinline__ float relu_j_fast(float c) {
  const float a = fabsf(c);
  const float d = clamp01(1.0f - a);
  const float s = __builtin_amdgcn_sqrtf(d);
  float r = -0.00017015916819218546f;
  r = fmaf(r, a, 0.0008293232531286776f);
  r = fmaf(r, a, -0.002279274631291628f);
  r = fmaf(r, a, 0.005947876255959272f);
  r = fmaf(r, a, -0.022532213479280472f);
  r = fmaf(r, a, 0.31830984354019165f);
  return fmaf(d * s, r, fmaxf(c, 0.0f));
}

// asin(c) for |c| <= 1, f32 NNGP-only erf fast path: ONE branch-free formula,
//   asin|c| = pi/2 - sqrt(1 - |c|) P7(|c|),   P7 = a minimax fit of (pi/2 - asin a) / sqrt(1 - a) on [0, 1] with P7(0) = fl(pi/2)
// (so asin(0) = 0 exactly and the odd extension is continuous): 11 vector instructions and one sqrt against 14 and a sqrt for
// the split form above.  |error| <= 2.6e-7 absolute over [-1, 1] in f32 arithmetic (the rounding of the pi/2 - ... difference;
// the fit itself is 5e-8), i.e. 1.6e-7 of the kernel's range after the 2/pi: the accuracy class of relu_j_fast below.
__device__ __forceinline__ float asin_fast(float c) {
  const float a = fabsf(c);
  const float s = __builtin_amdgcn_sqrtf(clamp01(1.0f - a));   // (c arrives unclamped: see relu_j_fast)
  float r = -0.0015507979551330209f;
  r = fmaf(r, a, 0.007713252678513527f);
  r = fmaf(r, a, -0.01858212612569332f);
  r = fmaf(r, a, 0.031964052468538284f);
  r = fmaf(r, a, -0.050575364381074905f);
  r = fmaf(r, a, 0.08905214071273804f);
  r = fmaf(r, a, -0.21460402011871338f);
  r = fmaf(r, a, 1.5707963705062866f);
  return copysignf(fmaf(-s, r, 1.5707963705062866f), c);
}

// Erf map.  rr = r_i r_j with r = 1/sqrt(1+2q).
template <typename T, bool WANT_DOT>
__device__ __forceinline__ ActOut<T> erf_map(T kt, T rr, T /*ss*/) {
  const T c = clamp1(T(2) * kt * rr);
  const T a = fabs(c);
  const T as = asin_abs(a, c * c);
  ActOut<T> o;
  o.k = T(2.0 / kPi) * copysign(as, c);
  if (WANT_DOT) o.kdot = T(4.0 / kPi) * rr * fast_rsqrt(fma(-c, c, T(1)));
  return o;
}

}  // namespace nngp
