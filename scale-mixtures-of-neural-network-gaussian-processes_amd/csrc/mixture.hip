// mixture.hip — predictive NLL under a sampled scale mixture (self-normalised importance sampling over sigma^2 draws).
//
// Replaces the innermost block of the grid search, experiments/regression/find.py:165-187, which the reference runs once
// per (w_std, b_std, eps, alpha, beta) cell in JAX: with sigma^2 draws q_s (Burr-XII there; any proposal here),
//   log p(data | q_s) = -(n/2) log 2 pi - 1/2 logdet - 1/2 quad / q_s - (n/2) log q_s                     (find.py:172)
//   w_s = exp(log p - max) * prior_s / proposal_s,  w~_s = w_s / sum w                                       (:174-177)
//   tnll = - mean_t logsumexp_s [ log(w~_s + 1e-24) + log N(y_t; mean_t, sqrt(q_s) sd_t y_std) ]            (:178-181)
// One workgroup per (mixture, problem): the S weights go through LDS once, every thread then owns test points and runs the
// S-term logsumexp from LDS.  All arithmetic in fp64 (the weights span hundreds of nats).  99 cells x 9 mixtures x 1000
// draws x 256 test points are 2.3e8 exponentials: microseconds here, seconds in NumPy.
#include <cmath>
#include <vector>

#include "internal.hpp"

namespace {

constexpr double kPiM = 3.14159265358979323846;

template <typename T>
__global__ void __launch_bounds__(256) mixture_nll_kernel(const T* __restrict__ mean, const T* __restrict__ var, int64_t t,
                                                          const double* __restrict__ quad, const double* __restrict__ logdet,
                                                          const int* __restrict__ skip, const double* __restrict__ y_test,
                                                          double y_mean, double y_std, double n, int S,
                                                          const double* __restrict__ sample_q, const double* __restrict__ ratio,
                                                          double* __restrict__ out) {
  extern __shared__ double sm[];       // [S] A_s = log(w~_s + 1e-24) - 1/2 log q_s ; [S] 1/q_s ; [256] reduction
  double* A = sm;
  double* iq = sm + S;
  double* red = iq + S;
  const int m = blockIdx.x, g = blockIdx.y, nm = gridDim.x, tid = threadIdx.x;
  if (skip && skip[g]) {
    if (tid == 0) out[(int64_t)g * nm + m] = nan("");
    return;
  }
  const double* q = sample_q + (int64_t)m * S;
  const double* rt = ratio ? ratio + (int64_t)m * S : nullptr;
  const double c0 = -0.5 * n * log(2.0 * kPiM) - 0.5 * logdet[g], mq = -0.5 * quad[g];
  // log p(data | q_s), its maximum over s
  double mx = -INFINITY;
  for (int s = tid; s < S; s += 256) {
    const double lp = c0 + mq / q[s] - 0.5 * n * log(q[s]);
    A[s] = lp;
    mx = fmax(mx, lp);
  }
  red[tid] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] = fmax(red[tid], red[tid + o]);
    __syncthreads();
  }
  mx = red[0];
  __syncthreads();
  double sum = 0.0;
  for (int s = tid; s < S; s += 256) {
    const double w = exp(A[s] - mx) * (rt ? rt[s] : 1.0);
    A[s] = w;
    sum += w;
  }
  red[tid] = sum;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  sum = red[0];
  __syncthreads();
  for (int s = tid; s < S; s += 256) {
    iq[s] = 1.0 / q[s];
    A[s] = log(A[s] / sum + 1e-24) - 0.5 * log(q[s]);
  }
  __syncthreads();
  // every test point: logsumexp_s (A_s - 1/2 z^2 / q_s) - log(sd y_std) - 1/2 log 2 pi
  double acc = 0.0;
  for (int64_t i = tid; i < t; i += 256) {
    const double sd = sqrt((double)var[(int64_t)g * t + i]) * y_std;
    const double z = (y_test[i] - ((double)mean[(int64_t)g * t + i] * y_std + y_mean)) / sd;
    const double hz2 = 0.5 * z * z;
    double m2 = -INFINITY;
    for (int s = 0; s < S; ++s) m2 = fmax(m2, A[s] - hz2 * iq[s]);
    double e = 0.0;
    for (int s = 0; s < S; ++s) e += exp(A[s] - hz2 * iq[s] - m2);
    acc += m2 + log(e) - log(sd) - 0.5 * log(2.0 * kPiM);
  }
  red[tid] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  if (tid == 0) out[(int64_t)g * nm + m] = -red[0] / (double)t;
}

}  // namespace

extern "C" int smn_mixture_nll(smn_ctx* ctx, int dtype, int nprob, int64_t t, const void* mean_d, const void* var_d,
                               const double* quad_h, const double* logdet_h, const int* skip_h, const double* y_test_h,
                               double y_mean, double y_std, int64_t n, int nmix, int nsamples, const double* sample_q_h,
                               const double* ratio_h, double* tnll_h) {
  if (!ctx || !mean_d || !var_d || !quad_h || !logdet_h || !y_test_h || !sample_q_h || !tnll_h) return SMN_EINVAL;
  SMN_ENTER(ctx);
  if (dtype != SMN_F32 && dtype != SMN_F64) return smn_fail(ctx, SMN_EINVAL, "bad dtype");
  if (nprob <= 0 || nprob > 65535 || t <= 0 || n <= 0 || nmix <= 0 || nsamples <= 0 || nsamples > 8192)
    return smn_fail(ctx, SMN_EINVAL, "smn_mixture_nll: bad sizes (at most 65535 problems, 8192 draws)");
  const size_t nd = (size_t)nprob * 2 + (size_t)t + (size_t)nmix * (size_t)nsamples * (ratio_h ? 2 : 1) + (size_t)nprob * (size_t)nmix;
  void* wv = nullptr;
  SMN_TRY(smn_workspace(ctx, 9, sizeof(double) * nd + sizeof(int) * (size_t)nprob, &wv));
  double* quad_d = static_cast<double*>(wv);
  double* ld_d = quad_d + nprob;
  double* y_d = ld_d + nprob;
  double* q_d = y_d + t;
  double* r_d = ratio_h ? q_d + (size_t)nmix * nsamples : nullptr;
  double* out_d = q_d + (size_t)nmix * nsamples * (ratio_h ? 2 : 1);
  int* skip_d = reinterpret_cast<int*>(out_d + (size_t)nprob * nmix);
  hipStream_t st = ctx->stream;
  SMN_HIP(ctx, hipMemcpyAsync(quad_d, quad_h, sizeof(double) * (size_t)nprob, hipMemcpyHostToDevice, st));
  SMN_HIP(ctx, hipMemcpyAsync(ld_d, logdet_h, sizeof(double) * (size_t)nprob, hipMemcpyHostToDevice, st));
  SMN_HIP(ctx, hipMemcpyAsync(y_d, y_test_h, sizeof(double) * (size_t)t, hipMemcpyHostToDevice, st));
  SMN_HIP(ctx, hipMemcpyAsync(q_d, sample_q_h, sizeof(double) * (size_t)nmix * (size_t)nsamples, hipMemcpyHostToDevice, st));
  if (ratio_h) SMN_HIP(ctx, hipMemcpyAsync(r_d, ratio_h, sizeof(double) * (size_t)nmix * (size_t)nsamples, hipMemcpyHostToDevice, st));
  if (skip_h) SMN_HIP(ctx, hipMemcpyAsync(skip_d, skip_h, sizeof(int) * (size_t)nprob, hipMemcpyHostToDevice, st));
  const size_t lds = sizeof(double) * (2 * (size_t)nsamples + 256);
  dim3 grid((unsigned)nmix, (unsigned)nprob);
  if (dtype == SMN_F64) {
    SMN_TRY(smn_allow_lds(ctx, reinterpret_cast<const void*>(mixture_nll_kernel<double>), lds));
    hipLaunchKernelGGL(mixture_nll_kernel<double>, grid, dim3(256), lds, st, static_cast<const double*>(mean_d), static_cast<const double*>(var_d), t,
                       quad_d, ld_d, skip_h ? skip_d : nullptr, y_d, y_mean, y_std, (double)n, nsamples, q_d, r_d, out_d);
  } else {
    SMN_TRY(smn_allow_lds(ctx, reinterpret_cast<const void*>(mixture_nll_kernel<float>), lds));
    hipLaunchKernelGGL(mixture_nll_kernel<float>, grid, dim3(256), lds, st, static_cast<const float*>(mean_d), static_cast<const float*>(var_d), t,
                       quad_d, ld_d, skip_h ? skip_d : nullptr, y_d, y_mean, y_std, (double)n, nsamples, q_d, r_d, out_d);
  }
  SMN_CHECK_LAUNCH(ctx);
  SMN_HIP(ctx, hipMemcpyAsync(tnll_h, out_d, sizeof(double) * (size_t)nprob * (size_t)nmix, hipMemcpyDeviceToHost, st));
  SMN_HIP(ctx, hipStreamSynchronize(st));
  return SMN_OK;
}
