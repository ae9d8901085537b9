// grad.hip — analytic hyper-parameter gradients of the log-marginal likelihood.
//
// Replaces what objax.GradValues(model.loss, vars) provides to experiments/regression/train.py:61-67.
// With K~ = K(w, b, lw) + eps I,  alpha = K~^-1 y  and  log p = f(quad = y^T K~^-1 y, logdet K~):
//     d log p / d theta = 1/2 * sum_ij G_ij * dK~_ij/d theta,      G = coef * alpha alpha^T - K~^-1
// (coef = 1 for the Gaussian head, (nu+N)/((nu + quad/s) s) for the Student-t head with shape s K~).
// alpha and -K~^-1 come out of the SAME augmented factorisation the predictive head uses
// (smn_predict with K_td = I, K_tt = 0: mean = alpha, "covariance" = -K~^-1), so the only new device
// code is the contraction: one pass over the lower triangle of K0 = X X^T / d that carries
// (K, dK/dw^2, dK/db^2) through the layer stack in forward mode (fp64 arithmetic whatever the storage
// type) and accumulates  sum G dK/dw_std,  sum G dK/db_std,  sum G dK/dlast_w_std,  tr G.
//
// Forward-mode rules (SURVEY.md Appendix A.1/A.2 differentiated; q_i, q_j = variances entering the map):
//   Dense:  A = w2 K + b2           dA/dw2 = K + w2 dK/dw2          dA/db2 = 1 + w2 dK/db2
//   ReLU:   phi   = sqrt(q_i q_j)/(2 pi) * (sqrt(1-c^2) + (pi - acos c) c),   c = A / sqrt(q_i q_j)
//           phi_A = (pi - acos c)/(2 pi)        phi_qi = sqrt(1-c^2) sqrt(q_i q_j) / (4 pi q_i)
//   Erf:    phi   = (2/pi) asin s,   s = 2A / sqrt(P),  P = (1+2q_i)(1+2q_j)
//           phi_A = 4 / (pi sqrt(P) sqrt(1-s^2))     phi_qi = -(2/pi) s / (sqrt(1-s^2) (1+2q_i))
#include <cmath>

#include "gemm_nt.hpp"
#include "internal.hpp"
#include "layer_prog.hpp"

namespace {

constexpr double kPiD = 3.14159265358979323846;

struct ActD {
  double o, dA, d1, d2;
};

template <int ACT>
__device__ __forceinline__ ActD act_d(double a, double qi, double qj) {
  ActD r;
  if (ACT == ACT_RELU) {
    const double sp = sqrt(qi * qj);
    double c = a / sp;
    c = fmin(fmax(c, -1.0), 1.0);
    const double s1 = sqrt(fmax(1.0 - c * c, 0.0));
    const double pm = kPiD - acos(c);
    r.o = sp * (s1 + pm * c) * (1.0 / (2.0 * kPiD));
    r.dA = pm * (1.0 / (2.0 * kPiD));
    r.d1 = s1 * sp / (4.0 * kPiD * qi);
    r.d2 = s1 * sp / (4.0 * kPiD * qj);
  } else {
    const double ti = 1.0 + 2.0 * qi, tj = 1.0 + 2.0 * qj;
    const double sP = sqrt(ti * tj);
    double s = 2.0 * a / sP;
    s = fmin(fmax(s, -1.0), 1.0);
    const double den = sqrt(fmax(1.0 - s * s, 1e-300));
    r.o = (2.0 / kPiD) * asin(s);
    r.dA = 4.0 / (kPiD * sP * den);
    r.d1 = -(2.0 / kPiD) * s / (den * ti);
    r.d2 = -(2.0 / kPiD) * s / (den * tj);
  }
  return r;
}

// Diagonal of the same recursion: variance entering each activation and its derivatives.
// tab[(s*3 + 0)*n + i] = q,  +1: dq/dw2,  +2: dq/db2.
template <int NET, int ACT>
__global__ void grad_tables_kernel(const double* __restrict__ q0, int64_t n, int nsets, double w2, double b2,
                                   double* __restrict__ tab) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double q = q0[i], dw = 0.0, db = 0.0;
  if (NET == NET_RESNET) {
    dw = q;
    db = 1.0;
    q = w2 * q + b2;
  }
  for (int s = 0; s < nsets; ++s) {
    if (NET == NET_MLP) {   // Dense in front of every activation
      const double qa = w2 * q + b2;
      dw = q + w2 * dw;
      db = 1.0 + w2 * db;
      q = qa;
    }
    tab[(int64_t)(s * 3 + 0) * n + i] = q;
    tab[(int64_t)(s * 3 + 1) * n + i] = dw;
    tab[(int64_t)(s * 3 + 2) * n + i] = db;
    const ActD r = act_d<ACT>(q, q, q);           // on the diagonal A = q_i = q_j
    const double dq = r.dA + r.d1 + r.d2;
    double o = r.o, ow = dq * dw, ob = dq * db;
    if (NET == NET_RESNET && s != nsets - 1) {    // K <- [Dense o act](K) + K
      const double ka = w2 * o + b2;
      const double kw = o + w2 * ow, kb = 1.0 + w2 * ob;
      o = q + ka;
      ow = dw + kw;
      ob = db + kb;
    }
    q = o; dw = ow; db = ob;
  }
}

template <typename T>
struct GradArgs {
  const T* k0; int64_t ldk0;
  const T* nkinv; int64_t ldki;      // -K~^-1
  const T* alpha;
  int64_t n;
  const double* tab;                 // grad_tables_kernel output
  int nsets;
  double w2, b2, lw2, coef;
  double* partial;                   // [gridDim.x][4]
};

constexpr int GT = 64;               // tile edge of the contraction

template <typename T, int NET, int ACT>
__global__ void __launch_bounds__(256) grad_contract_kernel(GradArgs<T> a) {
  int tr, tc;
  tri_decode(blockIdx.x, tr, tc);
  const int64_t row0 = (int64_t)tr * GT, col0 = (int64_t)tc * GT, n = a.n;
  const int tid = threadIdx.x;
  const int lc = tid % GT;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};   // sum G dK/dw2, sum G dK/db2, sum G K, tr G
  const int64_t j = col0 + lc;
  for (int lr = tid / GT; lr < GT; lr += 256 / GT) {
    const int64_t i = row0 + lr;
    if (i >= n || j >= n || j > i) continue;
    double k = (double)a.k0[i * a.ldk0 + j], dw = 0.0, db = 0.0;
    if (NET == NET_RESNET) {
      dw = k;
      db = 1.0;
      k = a.w2 * k + a.b2;
    }
    for (int s = 0; s < a.nsets; ++s) {
      if (NET == NET_MLP) {
        const double ka = a.w2 * k + a.b2;
        dw = k + a.w2 * dw;
        db = 1.0 + a.w2 * db;
        k = ka;
      }
      const double qi = a.tab[(int64_t)(s * 3 + 0) * n + i], qj = a.tab[(int64_t)(s * 3 + 0) * n + j];
      if (i == j) k = qi;                         // exact diagonal
      const ActD r = act_d<ACT>(k, qi, qj);
      double o = r.o;
      double ow = r.dA * dw + r.d1 * a.tab[(int64_t)(s * 3 + 1) * n + i] + r.d2 * a.tab[(int64_t)(s * 3 + 1) * n + j];
      double ob = r.dA * db + r.d1 * a.tab[(int64_t)(s * 3 + 2) * n + i] + r.d2 * a.tab[(int64_t)(s * 3 + 2) * n + j];
      if (NET == NET_RESNET && s != a.nsets - 1) {
        const double ka = a.w2 * o + a.b2;
        const double kw = o + a.w2 * ow, kb = 1.0 + a.w2 * ob;
        o = k + ka;
        ow = dw + kw;
        ob = db + kb;
      }
      k = o; dw = ow; db = ob;
    }
    const double g = a.coef * (double)a.alpha[i] * (double)a.alpha[j] + (double)a.nkinv[i * a.ldki + j];
    const double m = (i == j) ? 1.0 : 2.0;        // the upper triangle is the mirror image
    acc[0] += m * g * a.lw2 * dw;
    acc[1] += m * g * a.lw2 * db;
    acc[2] += m * g * a.lw2 * k;
    if (i == j) acc[3] += g;
  }
  __shared__ double red[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    double v = acc[q];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((tid & 63) == 0) red[q][tid >> 6] = v;
  }
  __syncthreads();
  if (tid < 4) a.partial[(int64_t)blockIdx.x * 4 + tid] = (red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]);
}

// Second stage: fixed-order sum of the per-tile partials (bitwise reproducible).
__global__ void __launch_bounds__(256) grad_reduce_kernel(const double* __restrict__ partial, int64_t ntiles,
                                                          double* __restrict__ out) {
  __shared__ double red[4][4];
  const int tid = threadIdx.x;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  for (int64_t t = tid; t < ntiles; t += 256)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] += partial[t * 4 + q];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    double v = acc[q];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    if ((tid & 63) == 0) red[q][tid >> 6] = v;
  }
  __syncthreads();
  if (tid < 4) out[tid] = (red[tid][0] + red[tid][1]) + (red[tid][2] + red[tid][3]);
}

template <typename T>
__global__ void cast_q_kernel(const T* __restrict__ s, double* __restrict__ d, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) d[i] = (double)s[i];
}

template <typename T, int NET, int ACT>
int grad_terms_na(smn_ctx* ctx, const GradArgs<T>& a, const double* q64, int64_t ntiles, double* out_d) {
  hipLaunchKernelGGL((grad_tables_kernel<NET, ACT>), dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, ctx->stream,
                     q64, a.n, a.nsets, a.w2, a.b2, const_cast<double*>(a.tab));
  SMN_CHECK_LAUNCH(ctx);
  {
    ProfScope ps(ctx, PROF_MISC, ctx->stream);
    hipLaunchKernelGGL((grad_contract_kernel<T, NET, ACT>), dim3((unsigned)ntiles), dim3(256), 0, ctx->stream, a);
  }
  SMN_CHECK_LAUNCH(ctx);
  hipLaunchKernelGGL(grad_reduce_kernel, dim3(1), dim3(256), 0, ctx->stream, a.partial, ntiles, out_d);
  SMN_CHECK_LAUNCH(ctx);
  return SMN_OK;
}

template <typename T>
int grad_terms_t(smn_ctx* ctx, int net, int act, int num_hiddens, double w_std, double b_std, double last_w_std,
                 const void* k0, int64_t n, int64_t ldk0, const void* q, const void* nkinv, int64_t ldki,
                 const void* alpha, double coef, double out_h[4]) {
  const int nsets = net == SMN_NET_MLP ? num_hiddens : num_hiddens + 1;
  if (nsets > kMaxSets) return smn_fail(ctx, SMN_ENOTSUP, "num_hiddens too large (max %d activation layers)", kMaxSets);
  const int64_t t = (n + GT - 1) / GT, ntiles = t * (t + 1) / 2;
  void* wsv = nullptr;
  const size_t nd = (size_t)n * (1 + 3 * (size_t)(nsets > 0 ? nsets : 1)) + (size_t)ntiles * 4 + 4;
  SMN_TRY(smn_workspace(ctx, 4, sizeof(double) * nd, &wsv));
  double* q64 = static_cast<double*>(wsv);
  double* tab = q64 + n;
  double* partial = tab + (size_t)n * 3 * (size_t)(nsets > 0 ? nsets : 1);
  double* out_d = partial + (size_t)ntiles * 4;
  hipLaunchKernelGGL(cast_q_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                     static_cast<const T*>(q), q64, n);
  SMN_CHECK_LAUNCH(ctx);
  GradArgs<T> a;
  a.k0 = static_cast<const T*>(k0); a.ldk0 = ldk0;
  a.nkinv = static_cast<const T*>(nkinv); a.ldki = ldki;
  a.alpha = static_cast<const T*>(alpha);
  a.n = n; a.tab = tab; a.nsets = nsets;
  a.w2 = w_std * w_std; a.b2 = b_std * b_std; a.lw2 = last_w_std * last_w_std; a.coef = coef;
  a.partial = partial;
  int rc;
  if (net == SMN_NET_MLP && act == SMN_ACT_RELU) rc = grad_terms_na<T, NET_MLP, ACT_RELU>(ctx, a, q64, ntiles, out_d);
  else if (net == SMN_NET_MLP) rc = grad_terms_na<T, NET_MLP, ACT_ERF>(ctx, a, q64, ntiles, out_d);
  else if (act == SMN_ACT_RELU) rc = grad_terms_na<T, NET_RESNET, ACT_RELU>(ctx, a, q64, ntiles, out_d);
  else rc = grad_terms_na<T, NET_RESNET, ACT_ERF>(ctx, a, q64, ntiles, out_d);
  SMN_TRY(rc);
  double s[4];
  SMN_HIP(ctx, hipMemcpyAsync(s, out_d, sizeof s, hipMemcpyDeviceToHost, ctx->stream));
  SMN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  out_h[0] = s[0] * 2.0 * w_std;                  // d/dw_std  = 2 w  d/dw^2
  out_h[1] = s[1] * 2.0 * b_std;
  out_h[2] = s[2] * 2.0 / last_w_std;             // K = lw^2 K_L  =>  dK/dlw = 2 K / lw
  out_h[3] = s[3];                                // dK~/deps = I
  return SMN_OK;
}

}  // namespace

extern "C" int smn_lml_grad_terms(smn_ctx* ctx, int dtype, int net, int act, int num_hiddens, double w_std,
                                  double b_std, double last_w_std, const void* k0_d, int64_t n, int64_t ldk0,
                                  const void* q_d, const void* neg_kinv_d, int64_t ldkinv, const void* alpha_d,
                                  double coef, double terms_h[4]) {
  if (!ctx || !k0_d || !q_d || !neg_kinv_d || !alpha_d || !terms_h) return SMN_EINVAL;
  SMN_ENTER(ctx);
  if (dtype != SMN_F32 && dtype != SMN_F64) return smn_fail(ctx, SMN_EINVAL, "bad dtype");
  if (n <= 0 || ldk0 < n || ldkinv < n) return smn_fail(ctx, SMN_EINVAL, "smn_lml_grad_terms: bad sizes");
  if (net != SMN_NET_MLP && net != SMN_NET_DENSE_RESNET) return smn_fail(ctx, SMN_EINVAL, "unknown net %d", net);
  if (act != SMN_ACT_RELU && act != SMN_ACT_ERF) return smn_fail(ctx, SMN_EINVAL, "Unsupported act %d", act);
  if (num_hiddens < 0 || !(last_w_std != 0.0)) return smn_fail(ctx, SMN_EINVAL, "smn_lml_grad_terms: bad hyper-parameters");
  if (dtype == SMN_F64)
    return grad_terms_t<double>(ctx, net, act, num_hiddens, w_std, b_std, last_w_std, k0_d, n, ldk0, q_d, neg_kinv_d,
                                ldkinv, alpha_d, coef, terms_h);
  return grad_terms_t<float>(ctx, net, act, num_hiddens, w_std, b_std, last_w_std, k0_d, n, ldk0, q_d, neg_kinv_d, ldkinv,
                             alpha_d, coef, terms_h);
}

// Fused: K0 = X X^T / d and its diagonal, K by the stand-alone recursion straight into the factorisation workspace
// laid out as [[K, .], [I, 0]], ONE augmented factorisation (alpha, -K~^-1, quad, logdet), then the contraction.
extern "C" int smn_spr_loss_grad(smn_ctx* ctx, int dtype, int net, int act, int num_hiddens, double w_std,
                                 double b_std, double last_w_std, const void* x_d, int64_t n, int64_t ldx, int64_t d,
                                 const void* y_d, double eps_abs, double df, double scale, double* quad_h,
                                 double* logdet_h, int* info_h, double terms_h[4]) {
  if (!ctx || !x_d || !y_d || !terms_h) return SMN_EINVAL;
  SMN_ENTER(ctx);
  if (dtype != SMN_F32 && dtype != SMN_F64) return smn_fail(ctx, SMN_EINVAL, "bad dtype");
  if (n <= 0 || d <= 0) return smn_fail(ctx, SMN_EINVAL, "smn_spr_loss_grad: empty");
  if (df > 0.0 && !(scale > 0.0)) return smn_fail(ctx, SMN_EINVAL, "smn_spr_loss_grad: scale must be > 0");
  const size_t es = dtype_size(dtype);
  const int64_t al = 16 / (int64_t)es;
  const int64_t ld0 = round_up(n, al);
  void *k0 = nullptr, *post = nullptr;
  SMN_TRY(smn_workspace(ctx, 5, es * ((size_t)n * ld0 + (size_t)n), &k0));
  SMN_TRY(smn_workspace(ctx, 7, es * ((size_t)n * ld0 + (size_t)n), &post));
  void* q = static_cast<char*>(k0) + es * (size_t)n * ld0;
  void* ninv = post;
  void* alpha = static_cast<char*>(post) + es * (size_t)n * ld0;
  SMN_TRY(smn_gram(ctx, dtype, x_d, n, ldx, nullptr, 0, 0, d, k0, ld0, q, nullptr));
  double quad = 0.0, logdet = 0.0;
  int info = 0;
  SMN_TRY(factor_with_identity(ctx, dtype, net, act, num_hiddens, w_std, b_std, last_w_std, k0, ld0, q, n, y_d, eps_abs, alpha,
                               ninv, ld0, &quad, &logdet, &info));
  if (quad_h) *quad_h = quad;
  if (logdet_h) *logdet_h = logdet;
  if (info_h) *info_h = info;
  if (info != 0) {
    for (int i = 0; i < 4; ++i) terms_h[i] = std::nan("");
    return SMN_OK;
  }
  double coef = 1.0;
  if (df > 0.0) coef = (df + (double)n) / ((df + quad / scale) * scale);
  return smn_lml_grad_terms(ctx, dtype, net, act, num_hiddens, w_std, b_std, last_w_std, k0, n, ld0, q, ninv, ld0, alpha,
                            coef, terms_h);
}
