// grad.hip — analytic hyper-parameter gradients of the log-marginal likelihood.
//
// Replaces what objax.GradValues(model.loss, vars) provides to experiments/regression/train.py:61-67.
// With K~ = K(w, b, lw) + eps I,  alpha = K~^-1 y  and  log p = f(quad = y^T K~^-1 y, logdet K~):
//     d log p / d theta = 1/2 * sum_ij G_ij * dK~_ij/d theta,      G = coef * alpha alpha^T - K~^-1
// (coef = 1 for the Gaussian head, (nu+N)/((nu + quad/s) s) for the Student-t head with shape s K~).
// alpha and -K~^-1 come out of the SAME augmented factorisation the predictive head uses
// (smn_predict with K_td = I, K_tt = 0: mean = alpha, "covariance" = -K~^-1), so the only new device
// code is the contraction: one pass over the lower triangle of K0 = X X^T / d that carries
// (K, dK/dw^2, dK/db^2) through the layer stack in forward mode (in the arithmetic of the storage type,
// sums in fp64) and accumulates  sum G dK/dw_std,  sum G dK/db_std,  sum G dK/dlast_w_std,  tr G.
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

// The same maps per ELEMENT, in the arithmetic R of the storage type (float for f32 matrices: the Gram entries and
// -K~^-1 carry f32 rounding already, fp64 arithmetic on them recovers nothing; double for f64), division-free from per-row
// tables and with the branch-free asin of the forward kernels (nngp_math.hpp) instead of libm's acos:
//   ReLU: ra = 1/sqrt(q), rb = sqrt(q):   c = A ra_i ra_j,  pi - acos c = pi/2 + asin c,  sp = rb_i rb_j,
//         phi_qi = sqrt(1-c^2) sp / (4 pi q_i) = sqrt(1-c^2)/(4 pi) * rb_j ra_i
//   Erf:  ra = 1/sqrt(1+2q), rb = ra^2:   s = 2 A ra_i ra_j,  phi_A = (4/pi) ra_i ra_j / sqrt(1-s^2),
//         phi_qi = -(2/pi) s rb_i / sqrt(1-s^2)
template <typename R>
struct ActR {
  R o, dA, d1, d2;
};

template <int ACT, typename R>
__device__ __forceinline__ ActR<R> act_r(R a, R rai, R rbi, R raj, R rbj) {
  ActR<R> r;
  if (ACT == ACT_RELU) {
    const R c = nngp::clamp1(a * (rai * raj));
    const R as = nngp::asin_abs(fabs(c), c * c);
    const R s1 = nngp::fast_sqrt(fmax(fma(-c, c, R(1)), R(0)));
    const R pm = R(kPiD / 2) + copysign(as, c);
    const R sp = rbi * rbj;
    r.o = sp * fma(pm, c, s1) * R(1.0 / (2.0 * kPiD));
    r.dA = pm * R(1.0 / (2.0 * kPiD));
    const R t = s1 * R(1.0 / (4.0 * kPiD));
    r.d1 = t * (rbj * rai);
    r.d2 = t * (rbi * raj);
  } else {
    const R uu = rai * raj;
    const R sv = nngp::clamp1(R(2) * a * uu);
    const R as = nngp::asin_abs(fabs(sv), sv * sv);
    const R rden = nngp::fast_rsqrt(fmax(fma(-sv, sv, R(1)), sizeof(R) == 8 ? R(1e-300) : R(1e-30)));
    r.o = R(2.0 / kPiD) * copysign(as, sv);
    r.dA = R(4.0 / kPiD) * uu * rden;
    const R t = R(-2.0 / kPiD) * sv * rden;
    r.d1 = t * rbi;
    r.d2 = t * rbj;
  }
  return r;
}

constexpr int kTabFields = 5;   // per activation layer and row: q, dq/dw2, dq/db2, ra, rb

// Diagonal of the same recursion (fp64 arithmetic): variance entering each activation, its derivatives and the two
// per-row factors of act_r.  tab[(s*5 + f)*n + i], f = 0: q, 1: dq/dw2, 2: dq/db2, 3: ra, 4: rb.
template <int NET, int ACT, typename R>
__global__ void grad_tables_kernel(const double* __restrict__ q0, int64_t n, int nsets, double w2, double b2,
                                   R* __restrict__ tab) {
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
    R* t = tab + (int64_t)s * kTabFields * n + i;
    t[0] = (R)q;
    t[n] = (R)dw;
    t[2 * n] = (R)db;
    if (ACT == ACT_RELU) {
      t[3 * n] = (R)(1.0 / sqrt(q));
      t[4 * n] = (R)sqrt(q);
    } else {
      t[3 * n] = (R)(1.0 / sqrt(1.0 + 2.0 * q));
      t[4 * n] = (R)(1.0 / (1.0 + 2.0 * q));
    }
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
  using R = T;                       // arithmetic of the per-element chain
  const T* k0; int64_t ldk0;
  const T* nkinv; int64_t ldki;      // -K~^-1
  const T* alpha;
  int64_t n;
  const R* tab;                      // grad_tables_kernel output
  int nsets;
  double w2, b2, lw2, coef;
  double* partial;                   // [gridDim.x][4]
};

constexpr int GT = 64;               // tile edge of the contraction

// One 64 x 64 tile of the lower triangle per workgroup; a thread owns one column and 16 rows (wave w: rows w, w+4, ...)
// and carries their (K, dK/dw2, dK/db2) through the layers together: the column-side table entries are loaded once per
// layer, the row-side ones are wave-uniform (scalar loads).  Products in R, the four sums in double.
template <typename T, int NET, int ACT>
__global__ void __launch_bounds__(256) grad_contract_kernel(GradArgs<T> a) {
  using R = typename GradArgs<T>::R;
  int tr, tc;
  tri_decode(blockIdx.x, tr, tc);
  const int64_t row0 = (int64_t)tr * GT, col0 = (int64_t)tc * GT, n = a.n;
  const int tid = threadIdx.x;
  const int lc = tid % GT;
  const int w = __builtin_amdgcn_readfirstlane(tid / GT);
  constexpr int NR = GT / 4;
  const int64_t j = col0 + lc, jc = j < n ? j : n - 1;
  const R w2 = (R)a.w2, b2 = (R)a.b2;
  R k[NR], dw[NR], db[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int64_t i = row0 + w + 4 * r, ic = i < n ? i : n - 1;
    k[r] = (R)a.k0[ic * a.ldk0 + jc];
    dw[r] = R(0);
    db[r] = R(0);
    if (NET == NET_RESNET) {
      dw[r] = k[r];
      db[r] = R(1);
      k[r] = fma(w2, k[r], b2);
    }
  }
  for (int s = 0; s < a.nsets; ++s) {
    const R* ts = a.tab + (int64_t)s * kTabFields * n;
    const R cdw = ts[n + jc], cdb = ts[2 * n + jc], cra = ts[3 * n + jc], crb = ts[4 * n + jc];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      const int64_t i = row0 + w + 4 * r, ic = i < n ? i : n - 1;   // wave-uniform
      const R rq = ts[ic], rdw = ts[n + ic], rdb = ts[2 * n + ic], rra = ts[3 * n + ic], rrb = ts[4 * n + ic];
      R kk = k[r], kw = dw[r], kb = db[r];
      if (NET == NET_MLP) {
        kw = fma(w2, kw, kk);
        kb = fma(w2, kb, R(1));
        kk = fma(w2, kk, b2);
      }
      if (i == j) kk = rq;                        // exact diagonal
      const ActR<R> q = act_r<ACT, R>(kk, rra, rrb, cra, crb);
      R o = q.o;
      R ow = fma(q.dA, kw, fma(q.d1, rdw, q.d2 * cdw));
      R ob = fma(q.dA, kb, fma(q.d1, rdb, q.d2 * cdb));
      if (NET == NET_RESNET && s != a.nsets - 1) {
        const R ka = fma(w2, o, b2);
        const R aw = fma(w2, ow, o), ab = fma(w2, ob, R(1));
        o = kk + ka;
        ow = kw + aw;
        ob = kb + ab;
      }
      k[r] = o; dw[r] = ow; db[r] = ob;
    }
  }
  double acc[4] = {0.0, 0.0, 0.0, 0.0};   // sum G dK/dw2, sum G dK/db2, sum G K, tr G
  const R aj = (R)a.alpha[jc], coef = (R)a.coef, lw2 = (R)a.lw2;
#pragma unroll
  for (int r = 0; r < NR; ++r) {
    const int64_t i = row0 + w + 4 * r, ic = i < n ? i : n - 1;
    const bool valid = i < n && j < n && j <= i;
    const R g = fma(coef * (R)a.alpha[ic], aj, (R)a.nkinv[ic * a.ldki + jc]);
    const R m = !valid ? R(0) : (i == j ? R(1) : R(2));   // the upper triangle is the mirror image
    const R gm = m * g * lw2;
    acc[0] += (double)(gm * dw[r]);
    acc[1] += (double)(gm * db[r]);
    acc[2] += (double)(gm * k[r]);
    if (valid && i == j) acc[3] += (double)g;
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
  hipLaunchKernelGGL((grad_tables_kernel<NET, ACT, T>), dim3((unsigned)((a.n + 255) / 256)), dim3(256), 0, ctx->stream,
                     q64, a.n, a.nsets, a.w2, a.b2, const_cast<T*>(a.tab));
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
  const size_t ntab = (size_t)n * kTabFields * (size_t)(nsets > 0 ? nsets : 1);   // T-typed; sized as doubles
  const size_t nd = (size_t)n + ntab + (size_t)ntiles * 4 + 4;
  SMN_TRY(smn_workspace(ctx, 4, sizeof(double) * nd, &wsv));
  double* q64 = static_cast<double*>(wsv);
  double* tabd = q64 + n;
  double* partial = tabd + ntab;
  double* out_d = partial + (size_t)ntiles * 4;
  hipLaunchKernelGGL(cast_q_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                     static_cast<const T*>(q), q64, n);
  SMN_CHECK_LAUNCH(ctx);
  GradArgs<T> a;
  a.k0 = static_cast<const T*>(k0); a.ldk0 = ldk0;
  a.nkinv = static_cast<const T*>(nkinv); a.ldki = ldki;
  a.alpha = static_cast<const T*>(alpha);
  a.n = n; a.tab = reinterpret_cast<const T*>(tabd); a.nsets = nsets;
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
// laid out as the rectangle [[K~], [I], [y^T]], a no-Schur factorisation (L, L^-T, L^-1 y), -K~^-1 = -L^-T L^-1 as one
// full-rate launch, alpha = L^-T (L^-1 y) (heads.hip factor_with_identity), then the contraction.
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
  SMN_TRY(gram_lower(ctx, dtype, x_d, n, ldx, d, k0, ld0, q));   // (lower tiles: all the recursion and the contraction read)
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
