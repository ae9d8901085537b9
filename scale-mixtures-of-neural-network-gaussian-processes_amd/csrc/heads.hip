// heads.hip — the inference heads of the path, all expressed on ONE augmented matrix:
//
//        [ K + shift   .          .   ]   rows 0 .. n_pad        (training block, identity padding)
//   A =  [ K_td        K_tt       .   ]   rows n_pad .. +t       (test rows)
//        [ y^T         0          0   ]   rows n_pad+t .. +c     (one row per output column of y)
//
// A partial Cholesky over the first n_pad columns (cholesky.hip) leaves
//   rows t:  K_td L^-T            Schur(t,t)  = K_tt - K_td K~^-1 K_dt      -> predictive covariance
//   rows y:  z^T = (L^-1 y)^T     Schur(y,t)  = - y^T K~^-1 K_dt            -> -predictive mean
//                                 Schur(y,y)  = - y^T K~^-1 y               -> -quadratic form
// and sum(log pivots) = logdet.  That covers
//   SPR.loss                      spax/models.py:93-98   (+ likelihoods.py:25-28,45-50, utils.py:160-183)
//   NNGPKernel.predict            spax/kernels.py:29-32  (neural_tangents gradient_descent_mse_ensemble)
//   the quadratic form of StudentTLikelihood.logpdf      spax/likelihoods.py:60-61
// The closed-form log-pdf arithmetic on the resulting scalars is done on the host (lgamma etc.).
#include <algorithm>
#include <climits>
#include <cmath>
#include <vector>

#include "internal.hpp"

namespace {

struct Aug {
  int64_t n, n_pad, t, c, n_total, lda;
  void* a;
  size_t es;
  char* at(int64_t r, int64_t col) const { return static_cast<char*>(a) + es * (size_t)(r * lda + col); }
};

int aug_alloc(smn_ctx* ctx, int dtype, int64_t n, int64_t t, int64_t c, Aug* g) {
  g->n = n; g->t = t; g->c = c;
  g->n_pad = round_up(n, kTile);
  g->n_total = g->n_pad + round_up(t + c, kTile);
  g->lda = g->n_total;
  g->es = dtype_size(dtype);
  return smn_workspace(ctx, 2, g->es * (size_t)g->n_total * (size_t)g->n_total, &g->a);
}

// Gaussian / multivariate-t log-pdf from (quad = y^T cov^-1 y, logdet = log det cov).
double logpdf_from(double quad, double logdet, int64_t n, double df, double scale, int info) {
  if (info != 0 || std::isnan(quad) || std::isnan(logdet)) return std::nan("");
  const double nn = (double)n;
  if (df <= 0.0)  // jax.scipy.stats.multivariate_normal.logpdf, spax/likelihoods.py:27
    return -0.5 * quad - 0.5 * nn * std::log(2.0 * M_PI) - 0.5 * logdet;
  // spax/utils.py:178-183 with shape = scale * cov
  const double t = 0.5 * (df + nn);
  const double quad_s = quad / scale;
  const double logdet_s = logdet + nn * std::log(scale);
  return -t * std::log1p(quad_s / df) - 0.5 * nn * std::log(df * M_PI) + std::lgamma(t) - std::lgamma(0.5 * df) -
         0.5 * logdet_s;
}

// Fused build of the augmented matrix straight from the inputs.
// allow_split: the caller goes on to aug_finish with an absolute jitter only (smn_spr_loss): under the look-ahead the
// bottom-right corner of the matrix is then built on the bulk stream beside the first super-panel's panel chain (run_build).
int aug_build(smn_ctx* ctx, const BuildSpec& spec, const Aug& g, const void* x, int64_t ldx, const void* xt,
              int64_t ldxt, int64_t d, int nbatch = 0, const double* bw = nullptr, const double* bb = nullptr,
              const double* blw = nullptr, bool allow_split = false, bool want_trace = false) {
  const int64_t kp = k_pad(spec.dtype, d);
  void* xs = nullptr;
  SMN_TRY(smn_workspace(ctx, 0, g.es * (size_t)kp * (size_t)g.n_total + sizeof(double) * (size_t)g.n_total, &xs));
  double* q = static_cast<double*>(xs);
  char* xp = reinterpret_cast<char*>(q + g.n_total);
  SMN_TRY(pad_rows(ctx, spec.dtype, x, g.n, ldx, d, xp, g.n_total, kp, q, g.n_pad, xt, g.t, ldxt));   // both blocks in one launch
  BuildCall c{};
  c.spec = spec;
  c.x1p = xp; c.ld1 = kp; c.rows1 = g.n_total; c.q1 = q;
  c.x2p = xp; c.ld2 = kp; c.rows2 = g.n_total; c.q2 = q;
  c.kp = (int)kp; c.d = d;
  c.symmetric = 1; c.mirror = 0; c.exact_diag = 1;
  c.store_mode = STORE_PAD_IDENTITY;
  c.nv0 = g.n; c.aug0 = g.n_pad; c.nv1 = g.t;
  c.get_mask = SMN_GET_NNGP;
  c.out_k = g.a; c.ldo = g.lda;
  c.nbatch = nbatch; c.bw = bw; c.bb = bb; c.blw = blw; c.out_bs = g.n_total * g.lda;   // batched: problem b at a + b * n_total^2
  // (not while pieces of a column-first exchange are pending on this context: the Arrival list is theirs)
  c.split_corner = (allow_split && nbatch == 0 && ctx->split_build && ctx->arrivals.empty()) ? split_corner_tiles(ctx, g.n_total / kTile) : 0;
  c.want_trace = (want_trace && nbatch == 0) ? 1 : 0;   // (relative ridge: the prep launch then carries the shift, aug_finish)
  return run_build(ctx, c);
}

int aug_finish(smn_ctx* ctx, int dtype, const Aug& g, const void* y, int64_t ldy, int64_t n_shift, double jitter_abs,
               double ridge_rel, void* mean, void* cov, int64_t ldcov, double* quad_h, double* logdet_h, int* info_h,
               bool td_identity = false) {
  if (g.c > 48) return smn_fail(ctx, SMN_ENOTSUP, "more than 48 output columns");   // the mailbox holds 62 doubles
  // the right-hand-side rows, the diagonal shift and the scalar reset are one launch: with an absolute jitter only, or with the
  // relative ridge when the build in front left the trace of the kernel's diagonal (aug_build want_trace)
  const bool traced = ctx->trace_ready;
  ctx->trace_ready = false;
  const bool prepped = g.c > 0 && (ridge_rel == 0.0 || traced);
  const int64_t n_sh = (jitter_abs != 0.0 || ridge_rel != 0.0) ? n_shift : 0;
  // a split build (aug_build): the corner's columns are prepped behind the corner's own launch, on the bulk stream, and the
  // factorisation takes them as ONE arrival (cholesky.hip need_columns: whoever first touches those columns waits for it)
  const int64_t corner = ctx->corner_col;
  ctx->corner_col = 0;
  if (corner > 0 && !prepped) SMN_HIP(ctx, hipStreamSynchronize(ctx->stream_bulk));   // (no caller does this: the trace needs every column)
  const bool split = corner > 0 && prepped && !ctx->consume_arrivals;
  if (split) {
    const int64_t sh = n_sh;
    SMN_TRY(aug_prep(ctx, dtype, g.a, g.lda, g.n_pad + g.t, corner, y, g.n, g.c, ldy, std::min(sh, corner), jitter_abs, 0, nullptr, ridge_rel, n_shift));
    const int prc = aug_prep(ctx, dtype, g.a, g.lda, g.n_pad + g.t, g.n_total, y, g.n, g.c, ldy, sh, jitter_abs, corner, ctx->stream_bulk, ridge_rel, n_shift);
    if (prc != SMN_OK) {
      (void)hipStreamSynchronize(ctx->stream_bulk);
      return prc;
    }
    SMN_HIP(ctx, hipEventRecord(ctx->ev_corner, ctx->stream_bulk));
    ctx->arrivals.clear();
    ctx->arrivals.push_back({corner, g.n_total, ctx->ev_corner});
    ctx->consume_arrivals = true;
  } else if (prepped) {
    if (corner > 0) SMN_HIP(ctx, hipStreamSynchronize(ctx->stream_bulk));
    SMN_TRY(aug_prep(ctx, dtype, g.a, g.lda, g.n_pad + g.t, g.n_total, y, g.n, g.c, ldy, n_sh, jitter_abs, 0, nullptr, ridge_rel, n_shift));
  } else {
    SMN_TRY(set_aug_rows(ctx, dtype, g.a, g.lda, g.n_pad + g.t, g.n_total, y, g.n, g.c, ldy));
  }
  ctx->chol_prepped = prepped;
  // identity test rows (the small-N gradient route): whole 128-row tiles of them are skipped where they are structurally zero
  const int64_t id0 = td_identity ? g.n_pad : -1, id1 = td_identity ? g.n_pad + g.t / kTile * kTile : -1;
  const int crc = cholesky_padded(ctx, dtype, g.a, g.n_total, g.n_pad, g.lda, n_shift, jitter_abs, ridge_rel, false, id0, id1);
  ctx->chol_prepped = false;
  if (split) {   // (the factorisation has made the caller's stream wait for the corner, whatever its return code)
    ctx->arrivals.clear();
    ctx->consume_arrivals = false;
  }
  SMN_TRY(crc);
  // a column-first exchange: the factorisation has waited for the pieces of the workspace; pieces scattered elsewhere (the NTK
  // of config 5, smn_shard_exchange_cols_to) ride the same scatter stream -- the call returns behind all of them
  if (ctx->consume_arrivals) SMN_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_c1, 0));
  double* quad_dev = ctx->d_scal + 8;
  SMN_TRY(extract_posterior(ctx, dtype, g.a, g.lda, g.n_pad, g.t, g.c, mean, cov, ldcov, quad_dev, true));
  double ld = 0.0;
  int info = 0;
  SMN_TRY(fetch_mail(ctx, quad_h ? (int)g.c : 0, quad_h, &ld, &info));
  if (info != 0) {
    ld = std::nan("");
    if (quad_h)
      for (int64_t k = 0; k < g.c; ++k) quad_h[k] = std::nan("");
  }
  if (logdet_h) *logdet_h = ld;
  if (info_h) *info_h = info;
  return SMN_OK;
}

}  // namespace

extern "C" int smn_cholesky(smn_ctx* ctx, int dtype, void* a_d, int64_t n_total, int64_t n_factor, int64_t lda,
                            int64_t n_shift, double jitter_abs, double ridge_rel, int* info_h, double* logdet_h) {
  if (!ctx || !a_d) return SMN_EINVAL;
  SMN_ENTER(ctx);
  if (dtype != SMN_F32 && dtype != SMN_F64) return smn_fail(ctx, SMN_EINVAL, "bad dtype");
  if (n_factor <= 0 || n_factor > n_total || lda < n_total || n_shift < 0 || n_shift > n_factor)
    return smn_fail(ctx, SMN_EINVAL, "smn_cholesky: bad sizes");
  const size_t es = dtype_size(dtype);
  const bool inplace = n_total % kTile == 0 && n_factor % kTile == 0 && lda % (16 / (int64_t)es) == 0 &&
                       (reinterpret_cast<uintptr_t>(a_d) & 15) == 0;
  if (inplace) {
    SMN_TRY(cholesky_padded(ctx, dtype, a_d, n_total, n_factor, lda, n_shift, jitter_abs, ridge_rel, true));
  } else {
    const int64_t m = n_total - n_factor, nfp = round_up(n_factor, kTile), ntp = nfp + round_up(m, kTile);
    void* w = nullptr;
    SMN_TRY(smn_workspace(ctx, 2, es * (size_t)ntp * (size_t)ntp, &w));
    char* wb = static_cast<char*>(w);
    const char* ab = static_cast<const char*>(a_d);
    SMN_HIP(ctx, hipMemsetAsync(w, 0, es * (size_t)ntp * (size_t)ntp, ctx->stream));
    SMN_TRY(copy_matrix(ctx, dtype, wb, ntp, ab, lda, n_factor, n_factor, 1));
    SMN_TRY(copy_matrix(ctx, dtype, wb + es * (size_t)(nfp * ntp), ntp, ab + es * (size_t)(n_factor * lda), lda, m, n_factor, 0));
    SMN_TRY(copy_matrix(ctx, dtype, wb + es * (size_t)(nfp * ntp + nfp), ntp, ab + es * (size_t)(n_factor * lda + n_factor), lda, m, m, 1));
    SMN_TRY(fill_identity_pad(ctx, dtype, w, ntp, nfp, n_factor));
    SMN_TRY(cholesky_padded(ctx, dtype, w, ntp, nfp, ntp, n_shift, jitter_abs, ridge_rel, true));
    char* ao = static_cast<char*>(a_d);
    SMN_TRY(copy_matrix(ctx, dtype, ao, lda, wb, ntp, n_factor, n_factor, 1));
    SMN_TRY(copy_matrix(ctx, dtype, ao + es * (size_t)(n_factor * lda), lda, wb + es * (size_t)(nfp * ntp), ntp, m, n_factor, 0));
    SMN_TRY(copy_matrix(ctx, dtype, ao + es * (size_t)(n_factor * lda + n_factor), lda, wb + es * (size_t)(nfp * ntp + nfp), ntp, m, m, 1));
  }
  double ld = 0.0;
  int info = 0;
  SMN_TRY(fetch_logdet_info(ctx, &ld, &info));
  if (info != 0) ld = std::nan("");
  if (logdet_h) *logdet_h = ld;
  if (info_h) *info_h = info;
  return SMN_OK;
}

extern "C" int smn_trsm(smn_ctx* ctx, int dtype, const void* l_d, int64_t n, int64_t ldl, void* b_d, int64_t nrhs,
                        int64_t ldb, int trans) {
  if (!ctx || !l_d || !b_d) return SMN_EINVAL;
  SMN_ENTER(ctx);
  if (dtype != SMN_F32 && dtype != SMN_F64) return smn_fail(ctx, SMN_EINVAL, "bad dtype");
  if (n <= 0 || nrhs <= 0) return smn_fail(ctx, SMN_EINVAL, "smn_trsm: empty");
  if (trans != 0 && trans != 1) return smn_fail(ctx, SMN_EINVAL, "smn_trsm: trans must be 0 or 1");
  // X = L^-1 B  <=>  X^T = B^T L^-T: the columns of B ride through the panel sweep as appended rows.
  // trans = 1: L^T X = B  <=>  L' (J X) = J B with L' = J L^T J lower triangular (J reverses the order),
  // so the same forward sweep serves the backward substitution.
  Aug g;
  SMN_TRY(aug_alloc(ctx, dtype, n, nrhs, 0, &g));
  SMN_HIP(ctx, hipMemsetAsync(g.a, 0, g.es * (size_t)g.n_total * (size_t)g.n_total, ctx->stream));
  if (trans == 0) {
    SMN_TRY(copy_matrix(ctx, dtype, g.a, g.lda, l_d, ldl, n, n, 1));
    SMN_TRY(transpose_matrix(ctx, dtype, g.at(g.n_pad, 0), g.lda, b_d, ldb, n, nrhs));
  } else {
    SMN_TRY(flip_transpose_lower(ctx, dtype, g.a, g.lda, l_d, ldl, n));
    SMN_TRY(transpose_flip(ctx, dtype, g.at(g.n_pad, 0), g.lda, b_d, ldb, n, nrhs, 1, 0));
  }
  SMN_TRY(fill_identity_pad(ctx, dtype, g.a, g.lda, g.n_pad, n));
  SMN_TRY(solve_rows_padded(ctx, dtype, g.a, g.n_total, g.n_pad, g.lda));
  if (trans == 0)
    SMN_TRY(transpose_matrix(ctx, dtype, b_d, ldb, g.at(g.n_pad, 0), g.lda, nrhs, n));
  else
    SMN_TRY(transpose_flip(ctx, dtype, b_d, ldb, g.at(g.n_pad, 0), g.lda, nrhs, n, 0, 1));
  return SMN_OK;
}

extern "C" int smn_transpose(smn_ctx* ctx, int dtype, void* dst_d, int64_t ldd, const void* src_d, int64_t lds, int64_t rows,
                             int64_t cols) {
  if (!ctx || !dst_d || !src_d) return SMN_EINVAL;
  SMN_ENTER(ctx);
  if (dtype != SMN_F32 && dtype != SMN_F64) return smn_fail(ctx, SMN_EINVAL, "bad dtype");
  if (rows < 0 || cols < 0 || lds < cols || ldd < rows) return smn_fail(ctx, SMN_EINVAL, "smn_transpose: bad sizes");
  return transpose_matrix(ctx, dtype, dst_d, ldd, src_d, lds, rows, cols);
}

extern "C" int smn_lml(smn_ctx* ctx, int dtype, void* k_d, int64_t n, int64_t ldk, const void* y_d, double eps_abs,
                       double df, double scale, double* logpdf_h, double* quad_h, double* logdet_h, int* info_h) {
  if (!ctx || !k_d || !y_d) return SMN_EINVAL;
  SMN_ENTER(ctx);
  if (dtype != SMN_F32 && dtype != SMN_F64) return smn_fail(ctx, SMN_EINVAL, "bad dtype");
  if (n <= 0) return smn_fail(ctx, SMN_EINVAL, "smn_lml: empty");
  if (df > 0.0 && !(scale > 0.0)) return smn_fail(ctx, SMN_EINVAL, "smn_lml: scale must be > 0");
  Aug g;
  SMN_TRY(aug_alloc(ctx, dtype, n, 0, 1, &g));
  // only rows >= n (identity padding + appended rows) need clearing: above them just the lower triangle is ever
  // read (tests: test_cholesky_reads_the_lower_triangle_only), and that is copied in below
  SMN_HIP(ctx, hipMemsetAsync(g.at(n, 0), 0, g.es * (size_t)(g.n_total - n) * (size_t)g.lda, ctx->stream));
  SMN_TRY(copy_matrix(ctx, dtype, g.a, g.lda, k_d, ldk, n, n, 1));
  SMN_TRY(fill_identity_pad(ctx, dtype, g.a, g.lda, g.n_pad, n));
  double quad = 0.0, ld = 0.0;
  int info = 0;
  SMN_TRY(aug_finish(ctx, dtype, g, y_d, 1, n, eps_abs, 0.0, nullptr, nullptr, 0, &quad, &ld, &info));
  if (logpdf_h) *logpdf_h = logpdf_from(quad, ld, n, df, scale, info);
  if (quad_h) *quad_h = quad;
  if (logdet_h) *logdet_h = ld;
  if (info_h) *info_h = info;
  return SMN_OK;
}

// smn_lml fed straight from the gathered paired lower-block staging buffer (multi-GPU path): the blocks are
// scattered into the factorisation workspace itself, so the assembled kernel is never materialised a second time.
extern "C" int smn_lml_from_blocks(smn_ctx* ctx, int dtype, const void* stage_d, int64_t n, int nranks,
                                   int64_t block_rows, const void* y_d, double eps_abs, double df, double scale,
                                   double* logpdf_h, double* quad_h, double* logdet_h, int* info_h) {
  if (!ctx || !stage_d || !y_d) return SMN_EINVAL;
  SMN_ENTER(ctx);
  if (dtype != SMN_F32 && dtype != SMN_F64) return smn_fail(ctx, SMN_EINVAL, "bad dtype");
  if (n <= 0) return smn_fail(ctx, SMN_EINVAL, "smn_lml_from_blocks: empty");
  if (df > 0.0 && !(scale > 0.0)) return smn_fail(ctx, SMN_EINVAL, "smn_lml_from_blocks: scale must be > 0");
  Aug g;
  SMN_TRY(aug_alloc(ctx, dtype, n, 0, 1, &g));
  SMN_HIP(ctx, hipMemsetAsync(g.at(n, 0), 0, g.es * (size_t)(g.n_total - n) * (size_t)g.lda, ctx->stream));
  SMN_TRY(smn_unpack_lower_blocks(ctx, dtype, stage_d, n, nranks, block_rows, g.a, g.lda));
  SMN_TRY(fill_identity_pad(ctx, dtype, g.a, g.lda, g.n_pad, n));
  double quad = 0.0, ld = 0.0;
  int info = 0;
  SMN_TRY(aug_finish(ctx, dtype, g, y_d, 1, n, eps_abs, 0.0, nullptr, nullptr, 0, &quad, &ld, &info));
  if (logpdf_h) *logpdf_h = logpdf_from(quad, ld, n, df, scale, info);
  if (quad_h) *quad_h = quad;
  if (logdet_h) *logdet_h = ld;
  if (info_h) *info_h = info;
  return SMN_OK;
}

// ---- the column-first multi-GPU route ("chunk ... and overlap with compute") ----
//   smn_shard_begin            factorisation workspace up (padding rows cleared) before the first piece arrives; states the
//                              absolute jitter, which the scatter adds to the diagonal entries as it writes them
//   smn_kernel_mlp_shard_cols  (kernel_build.hip) the rank's whole share of the build: ONE launch on every CU
//   smn_shard_exchange_cols    piece g (a column range of the lower triangle, every rank's share of it): all-gather on the
//                              communication stream, scatter into the workspace on the scatter stream, and an Arrival
//                              {columns, event} for the factorisation.  Returns at once.
//   smn_lml_from_shards        factorisation + head.  The main stream waits PIECE BY PIECE (cholesky.hip need_columns): the
//                              panel chain of a super-panel for that super-panel's columns, F0 for the next one's, the bulk far
//                              update for all -- so only the first piece's gather + scatter is exposed, the rest of the exchange
//                              rides under the first panel chain.
namespace {

__global__ void delay_kernel(long long ticks) {   // test hook: holds a stream for a bounded time (100 MHz wall clock)
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

int pool_event(smn_ctx* ctx, hipEvent_t* ev) {
  if (ctx->ev_pool_used == ctx->ev_pool.size()) {
    hipEvent_t e;
    SMN_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
    ctx->ev_pool.push_back(e);
  }
  *ev = ctx->ev_pool[ctx->ev_pool_used++];
  return SMN_OK;
}

int check_exchange(smn_ctx* ctx, const char* who, int dtype, int64_t n, int nranks, bool to_workspace) {
  if (dtype != SMN_F32 && dtype != SMN_F64) return smn_fail(ctx, SMN_EINVAL, "bad dtype");
  if (to_workspace) {
    if (!ctx->shard_a || ctx->shard_n != n || ctx->shard_dtype != dtype)
      return smn_fail(ctx, SMN_EINVAL, "%s: smn_shard_begin(dtype, n, eps) first", who);
    if (ctx->ws[2] != ctx->shard_a) return smn_fail(ctx, SMN_EINVAL, "%s: the workspace moved since smn_shard_begin", who);
  }
  const int P = ctx->comm ? ctx->nranks : 1;
  if (nranks != P)   // a world > 1 call on a context without a communicator would quietly gather nothing
    return smn_fail(ctx, SMN_ECOMM, "%s: %d ranks asked for, the context's communicator has %d", who, nranks, P);
  return SMN_OK;
}

// scatter of piece g on the scatter stream behind `after` (an event on whichever stream filled the staging buffer); into the
// factorisation workspace (k_d == nullptr: jitter on the diagonal, an Arrival for the factorisation) or a matrix of the caller's
int scatter_piece(smn_ctx* ctx, int dtype, const void* stage_d, int64_t n, const ColPieces& cp, int g, hipEvent_t after,
                  void* k_d, int64_t ldk) {
  hipStream_t ss = ctx->stream_scatter;
  SMN_HIP(ctx, hipStreamWaitEvent(ss, after, 0));
  {
    ProfScope ps(ctx, PROF_MISC, ss);
    if (k_d) SMN_TRY(scatter_piece_on(ctx, ss, dtype, stage_d, n, cp, g, k_d, ldk, 0.0));
    else SMN_TRY(scatter_piece_on(ctx, ss, dtype, stage_d, n, cp, g, ctx->shard_a, ctx->shard_lda, ctx->shard_eps));
  }
  if (!k_d) {
    hipEvent_t arrived;
    SMN_TRY(pool_event(ctx, &arrived));
    SMN_HIP(ctx, hipEventRecord(arrived, ss));
    int64_t hi = cp.c[g + 1] * kTile;
    if (hi > n) hi = n;
    ctx->arrivals.push_back({cp.c[g] * kTile, hi, arrived});
  }
  SMN_HIP(ctx, hipEventRecord(ctx->ev_c1, ss));   // "everything issued so far has landed" (smn_shard_wait)
  return SMN_OK;
}

int exchange_piece(smn_ctx* ctx, const char* who, int dtype, const void* mine_d, void* stage_d, int64_t n, int nranks, int npieces,
                   const int64_t* piece_cols, int piece, void* k_d, int64_t ldk) {
  SMN_TRY(check_exchange(ctx, who, dtype, n, nranks, k_d == nullptr));
  ColPieces cp;
  SMN_TRY(col_pieces_make(ctx, n, nranks, npieces, piece_cols, &cp));
  if (piece < 0 || piece >= npieces) return smn_fail(ctx, SMN_EINVAL, "%s: piece %d of %d", who, piece, npieces);
  hipStream_t sc = ctx->stream_comm;
  SMN_HIP(ctx, hipEventRecord(ctx->ev_c0, ctx->stream));      // the rank's share is built and the workspace is up
  SMN_HIP(ctx, hipStreamWaitEvent(sc, ctx->ev_c0, 0));
  {
    ProfScope ps(ctx, PROF_COMM, sc);
    SMN_TRY(allgather_piece_on(ctx, sc, dtype, mine_d, stage_d, cp, piece));
  }
  // The scatter has its own stream: the NEXT piece's all-gather (link-bound) starts at once, not behind this scatter (HBM-bound).
  hipEvent_t gathered;
  SMN_TRY(pool_event(ctx, &gathered));
  SMN_HIP(ctx, hipEventRecord(gathered, sc));
  return scatter_piece(ctx, dtype, stage_d, n, cp, piece, gathered, k_d, ldk);
}

}  // namespace

extern "C" int smn_shard_begin(smn_ctx* ctx, int dtype, int64_t n, double eps_abs) {
  if (!ctx) return SMN_EINVAL;
  SMN_ENTER(ctx);
  if (dtype != SMN_F32 && dtype != SMN_F64) return smn_fail(ctx, SMN_EINVAL, "bad dtype");
  if (n <= 0) return smn_fail(ctx, SMN_EINVAL, "smn_shard_begin: empty");
  if (!ctx->arrivals.empty()) {   // an abandoned pipeline: its pieces may still be landing in the workspace about to be cleared
    SMN_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_c1, 0));
    ctx->arrivals.clear();
  }
  ctx->ev_pool_used = 0;
  Aug g;
  SMN_TRY(aug_alloc(ctx, dtype, n, 0, 1, &g));
  SMN_HIP(ctx, hipMemsetAsync(g.at(n, 0), 0, g.es * (size_t)(g.n_total - n) * (size_t)g.lda, ctx->stream));
  SMN_TRY(fill_identity_pad(ctx, dtype, g.a, g.lda, g.n_pad, n));
  ctx->shard_a = g.a; ctx->shard_lda = g.lda; ctx->shard_n = n; ctx->shard_dtype = dtype; ctx->shard_eps = eps_abs;
  return SMN_OK;
}

extern "C" int smn_shard_exchange_cols(smn_ctx* ctx, int dtype, const void* mine_d, void* stage_d, int64_t n, int nranks,
                                       int npieces, const int64_t* piece_cols, int piece) {
  if (!ctx || !mine_d || !stage_d) return SMN_EINVAL;
  SMN_ENTER(ctx);
  return exchange_piece(ctx, "smn_shard_exchange_cols", dtype, mine_d, stage_d, n, nranks, npieces, piece_cols, piece, nullptr, 0);
}

// The same exchange of one piece, scattered into a matrix of the caller's (the NTK of a joint NNGP + NTK shard: BASELINE
// config 5) instead of the factorisation workspace: no jitter, no Arrival; smn_shard_wait (or smn_lml_from_shards) makes the
// main stream wait for every piece issued before it.
extern "C" int smn_shard_exchange_cols_to(smn_ctx* ctx, int dtype, const void* mine_d, void* stage_d, int64_t n, int nranks,
                                          int npieces, const int64_t* piece_cols, int piece, void* k_d, int64_t ldk) {
  if (!ctx || !mine_d || !stage_d || !k_d) return SMN_EINVAL;
  SMN_ENTER(ctx);
  if (n <= 0 || ldk < n) return smn_fail(ctx, SMN_EINVAL, "smn_shard_exchange_cols_to: bad sizes");
  return exchange_piece(ctx, "smn_shard_exchange_cols_to", dtype, mine_d, stage_d, n, nranks, npieces, piece_cols, piece, k_d, ldk);
}

// The scatter half alone, for a staging buffer the caller filled on the main stream (tests that play P ranks on one GPU;
// a caller with its own transport): piece g of stage_d [P * elements per rank] into the workspace (k_d == NULL) or k_d.
extern "C" int smn_shard_scatter_cols(smn_ctx* ctx, int dtype, const void* stage_d, int64_t n, int nranks, int npieces,
                                      const int64_t* piece_cols, int piece, void* k_d, int64_t ldk) {
  if (!ctx || !stage_d) return SMN_EINVAL;
  SMN_ENTER(ctx);
  if (dtype != SMN_F32 && dtype != SMN_F64) return smn_fail(ctx, SMN_EINVAL, "bad dtype");
  if (!k_d) {
    if (!ctx->shard_a || ctx->shard_n != n || ctx->shard_dtype != dtype)
      return smn_fail(ctx, SMN_EINVAL, "smn_shard_scatter_cols: smn_shard_begin(dtype, n, eps) first");
    if (ctx->ws[2] != ctx->shard_a) return smn_fail(ctx, SMN_EINVAL, "smn_shard_scatter_cols: the workspace moved since smn_shard_begin");
  } else if (ldk < n) {
    return smn_fail(ctx, SMN_EINVAL, "smn_shard_scatter_cols: ldk < n");
  }
  ColPieces cp;
  SMN_TRY(col_pieces_make(ctx, n, nranks, npieces, piece_cols, &cp));
  if (piece < 0 || piece >= npieces) return smn_fail(ctx, SMN_EINVAL, "smn_shard_scatter_cols: piece %d of %d", piece, npieces);
  hipEvent_t filled;
  SMN_TRY(pool_event(ctx, &filled));
  SMN_HIP(ctx, hipEventRecord(filled, ctx->stream));
  return scatter_piece(ctx, dtype, stage_d, n, cp, piece, filled, k_d, ldk);
}

// The main stream waits for every piece issued so far (for callers of smn_shard_exchange_cols_to that do not finish with
// smn_lml_from_shards).
extern "C" int smn_shard_wait(smn_ctx* ctx) {
  if (!ctx) return SMN_EINVAL;
  SMN_ENTER(ctx);
  ProfScope ps(ctx, PROF_STALL, ctx->stream);
  SMN_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_c1, 0));
  return SMN_OK;
}

// Test hook: hold one of the context's streams (0 main, 1 communication, 2 scatter) for `usec` microseconds (at most 1 s), so a
// test can make pieces arrive late or out of order.
extern "C" int smn_debug_delay(smn_ctx* ctx, int stream_id, int64_t usec) {
  if (!ctx || stream_id < 0 || stream_id > 2 || usec < 0 || usec > 1000000) return SMN_EINVAL;
  SMN_ENTER(ctx);
  hipStream_t st = stream_id == 0 ? ctx->stream : (stream_id == 1 ? ctx->stream_comm : ctx->stream_scatter);
  hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, st, (long long)usec * 100);
  SMN_CHECK_LAUNCH(ctx);
  return SMN_OK;
}

extern "C" int smn_lml_from_shards(smn_ctx* ctx, int dtype, int64_t n, const void* y_d, double df, double scale,
                                   double* logpdf_h, double* quad_h, double* logdet_h, int* info_h) {
  if (!ctx || !y_d) return SMN_EINVAL;
  SMN_ENTER(ctx);
  if (!ctx->shard_a || ctx->shard_n != n || ctx->shard_dtype != dtype)
    return smn_fail(ctx, SMN_EINVAL, "smn_lml_from_shards: smn_shard_begin(dtype, n, eps) first");
  if (df > 0.0 && !(scale > 0.0)) return smn_fail(ctx, SMN_EINVAL, "smn_lml_from_shards: scale must be > 0");
  Aug g;
  SMN_TRY(aug_alloc(ctx, dtype, n, 0, 1, &g));       // the same slot and size: no reallocation
  if (g.a != ctx->shard_a) return smn_fail(ctx, SMN_EINVAL, "smn_lml_from_shards: the workspace moved since smn_shard_begin");
  // every column of the kernel must have been issued as a piece (the factorisation waits for Arrivals, not for the streams)
  {
    std::vector<char> have((size_t)g.n_pad / kTile, 0);
    for (const auto& a : ctx->arrivals)
      for (int64_t c = a.col_begin / kTile; c < (a.col_end + kTile - 1) / kTile; ++c) have[(size_t)c] = 1;
    for (int64_t c = 0; c < (n + kTile - 1) / kTile; ++c)
      if (!have[(size_t)c]) return smn_fail(ctx, SMN_EINVAL, "smn_lml_from_shards: tile column %lld was never exchanged", (long long)c);
  }
  double quad = 0.0, ld = 0.0;
  int info = 0;
  // jitter_abs = 0: the scatter added it already.  The factorisation consumes the Arrivals and leaves the main stream behind
  // all of them.
  ctx->consume_arrivals = true;
  const int rc = aug_finish(ctx, dtype, g, y_d, 1, n, 0.0, 0.0, nullptr, nullptr, 0, &quad, &ld, &info);
  ctx->consume_arrivals = false;
  ctx->arrivals.clear();
  ctx->shard_a = nullptr;
  SMN_TRY(rc);
  if (logpdf_h) *logpdf_h = logpdf_from(quad, ld, n, df, scale, info);
  if (quad_h) *quad_h = quad;
  if (logdet_h) *logdet_h = ld;
  if (info_h) *info_h = info;
  return SMN_OK;
}

namespace {

template <typename T>
__global__ void identity_block_kernel(T* __restrict__ a, int64_t lda, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i * lda + i] = T(1);
}

}  // namespace

// Analytic gradients (grad.hip): alpha = K~^-1 y and -K~^-1.  Two routes, by size:
// * n_pad >= kGradRectFromN: WITHOUT a 2N x 2N matrix.  The workspace is the RECTANGLE
//        [ K~ ]   n_pad rows        (K by the layer recursion of the Gram matrix k0, straight into it)
//   A =  [ I  ]   n rows            identity block: row i is structurally zero left of column i
//        [ y^T]   1 row (+ padding)
//   of n_pad columns.  A no-Schur factorisation (cholesky.hip: the appended rows' trailing block is never touched, and here
//   does not exist) leaves L, X = I L^-T = L^-T (upper triangular rows) and z^T = (L^-1 y)^T; then ONE full-rate launch forms
//   -K~^-1 = -X X^T (each tile's K loop starts at its row's first non-zero column) straight into the caller's matrix, and a
//   row pass gives alpha = X z and the quadratic form z^T z.  Same N^3 flops as the joint factorisation below, half its memory
//   (2 N^2 instead of 4 N^2 elements), and the N^3 / 3 of the inverse no longer rides on the panel chain: 49.6 -> 46.0 ms at
//   N = 16384.
// * below that: the joint factorisation of [[K~, .], [I, 0], [y^T, 0, 0]] (smn_predict with K_td = I, K_tt = 0, assembled in
//   the workspace): -K~^-1 and alpha fall out of the Schur complement, whose updates ride in the factorisation's own launches.
//   At these sizes the factorisation is chain-bound and those updates are free, while the rectangle route's extra launches
//   are not (N = 4096: 2.08 against 2.22 ms; N = 245: 206 against 221 us; profiles/r04_small_n_latency.txt).
constexpr int64_t kGradRectFromN = 8192;

int factor_with_identity(smn_ctx* ctx, int dtype, int net, int act, int num_hiddens, double w_std, double b_std,
                         double last_w_std, const void* k0_d, int64_t ldk0, const void* q_d, int64_t n, const void* y_d,
                         double eps_abs, void* alpha_d, void* ninv_d, int64_t ldinv, double* quad_h, double* logdet_h,
                         int* info_h) {
  if (round_up(n, kTile) < kGradRectFromN) {
    Aug g;
    SMN_TRY(aug_alloc(ctx, dtype, n, n, 1, &g));
    SMN_HIP(ctx, hipMemsetAsync(g.at(n, 0), 0, g.es * (size_t)(g.n_total - n) * (size_t)g.lda, ctx->stream));
    SMN_TRY(smn_recursion(ctx, dtype, net, act, num_hiddens, w_std, b_std, last_w_std, k0_d, n, n, ldk0, q_d, q_d, 1,
                          SMN_GET_NNGP, g.a, nullptr, g.lda));
    if (dtype == SMN_F64)
      hipLaunchKernelGGL(identity_block_kernel<double>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                         reinterpret_cast<double*>(g.at(g.n_pad, 0)), g.lda, n);
    else
      hipLaunchKernelGGL(identity_block_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                         reinterpret_cast<float*>(g.at(g.n_pad, 0)), g.lda, n);
    SMN_CHECK_LAUNCH(ctx);
    SMN_TRY(fill_identity_pad(ctx, dtype, g.a, g.lda, g.n_pad, n));
    return aug_finish(ctx, dtype, g, y_d, 1, n, eps_abs, 0.0, alpha_d, ninv_d, ldinv, quad_h, logdet_h, info_h, true);
  }
  const size_t es = dtype_size(dtype);
  const int64_t n_pad = round_up(n, kTile), n_app = round_up(n + 1, kTile), n_total = n_pad + n_app, lda = n_pad;
  void* av = nullptr;
  SMN_TRY(smn_workspace(ctx, 2, es * (size_t)n_total * (size_t)lda, &av));
  char* a = static_cast<char*>(av);
  SMN_HIP(ctx, hipMemsetAsync(a + es * (size_t)n * (size_t)lda, 0, es * (size_t)(n_total - n) * (size_t)lda, ctx->stream));
  SMN_TRY(smn_recursion(ctx, dtype, net, act, num_hiddens, w_std, b_std, last_w_std, k0_d, n, n, ldk0, q_d, q_d, 1,
                        SMN_GET_NNGP, a, nullptr, lda));
  char* xrows = a + es * (size_t)n_pad * (size_t)lda;
  if (dtype == SMN_F64)
    hipLaunchKernelGGL(identity_block_kernel<double>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       reinterpret_cast<double*>(xrows), lda, n);
  else
    hipLaunchKernelGGL(identity_block_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       reinterpret_cast<float*>(xrows), lda, n);
  SMN_CHECK_LAUNCH(ctx);
  SMN_TRY(fill_identity_pad(ctx, dtype, a, lda, n_pad, n));
  // y^T in the row behind the identity block, the absolute jitter on the diagonal, logdet / info reset: one launch
  SMN_TRY(aug_prep(ctx, dtype, a, lda, n_pad + n, n_pad, y_d, n, 1, 1, eps_abs != 0.0 ? n : 0, eps_abs));
  ctx->chol_prepped = true;
  ctx->chol_noschur = true;
  const int crc = cholesky_padded(ctx, dtype, a, n_total, n_pad, lda, n, eps_abs, 0.0, false, n_pad, n_pad + n / kTile * kTile);
  ctx->chol_prepped = false;
  ctx->chol_noschur = false;
  SMN_TRY(crc);
  double* quad_dev = ctx->d_scal + 8;
  SMN_TRY(inverse_from_rows(ctx, dtype, xrows, lda, xrows + es * (size_t)n * (size_t)lda, n_pad, n, ninv_d, ldinv, alpha_d, quad_dev));
  double ld = 0.0, quad = 0.0;
  int info = 0;
  SMN_TRY(fetch_results(ctx, quad_dev, 1, &quad, &ld, &info));
  if (info != 0) ld = quad = std::nan("");
  if (quad_h) *quad_h = quad;
  if (logdet_h) *logdet_h = ld;
  if (info_h) *info_h = info;
  return SMN_OK;
}

int predict_joint(smn_ctx* ctx, int dtype, void* kj_d, int64_t n, int64_t t, int64_t ldk, const void* y_d, int64_t c,
                  double ridge_rel, double ridge_abs, void* mean_d, void* cov_d, int64_t ldcov, double* quad_h,
                  double* logdet_h, int* info_h) {
  if (!ctx || !kj_d || !y_d) return SMN_EINVAL;
  if (dtype != SMN_F32 && dtype != SMN_F64) return smn_fail(ctx, SMN_EINVAL, "bad dtype");
  if (n <= 0 || t < 0 || c <= 0) return smn_fail(ctx, SMN_EINVAL, "smn_predict: bad sizes");
  if (c > 48) return smn_fail(ctx, SMN_ENOTSUP, "more than 48 output columns");
  Aug g;
  SMN_TRY(aug_alloc(ctx, dtype, n, t, c, &g));
  const char* kb = static_cast<const char*>(kj_d);
  SMN_HIP(ctx, hipMemsetAsync(g.at(n, 0), 0, g.es * (size_t)(g.n_total - n) * (size_t)g.lda, ctx->stream));
  SMN_TRY(copy_matrix(ctx, dtype, g.a, g.lda, kb, ldk, n, n, 1));
  SMN_TRY(copy_matrix(ctx, dtype, g.at(g.n_pad, 0), g.lda, kb + g.es * (size_t)(n * ldk), ldk, t, n, 0));
  SMN_TRY(copy_matrix(ctx, dtype, g.at(g.n_pad, g.n_pad), g.lda, kb + g.es * (size_t)(n * ldk + n), ldk, t, t, 1));
  SMN_TRY(fill_identity_pad(ctx, dtype, g.a, g.lda, g.n_pad, n));
  return aug_finish(ctx, dtype, g, y_d, c, n, ridge_abs, ridge_rel, mean_d, cov_d, ldcov, quad_h, logdet_h, info_h);
}

extern "C" int smn_predict(smn_ctx* ctx, int dtype, void* kj_d, int64_t n, int64_t t, int64_t ldk, const void* y_d,
                           int64_t c, double ridge_rel, double ridge_abs, void* mean_d, void* cov_d, int64_t ldcov,
                           double* quad_h, double* logdet_h, int* info_h) {
  if (!ctx) return SMN_EINVAL;
  SMN_ENTER(ctx);
  return predict_joint(ctx, dtype, kj_d, n, t, ldk, y_d, c, ridge_rel, ridge_abs, mean_d, cov_d, ldcov, quad_h, logdet_h,
                       info_h);
}

extern "C" int smn_spr_loss(smn_ctx* ctx, int dtype, int net, int act, int num_hiddens, double w_std, double b_std,
                            double last_w_std, const void* x_d, int64_t n, int64_t ldx, int64_t d, const void* y_d,
                            double eps_abs, double df, double scale, double* logpdf_h, double* quad_h, double* logdet_h,
                            int* info_h) {
  if (!ctx || !x_d || !y_d) return SMN_EINVAL;
  SMN_ENTER(ctx);
  if (dtype != SMN_F32 && dtype != SMN_F64) return smn_fail(ctx, SMN_EINVAL, "bad dtype");
  if (n <= 0 || d <= 0) return smn_fail(ctx, SMN_EINVAL, "smn_spr_loss: empty");
  if (df > 0.0 && !(scale > 0.0)) return smn_fail(ctx, SMN_EINVAL, "smn_spr_loss: scale must be > 0");
  Aug g;
  SMN_TRY(aug_alloc(ctx, dtype, n, 0, 1, &g));
  BuildSpec s{dtype, net, act, num_hiddens, w_std, b_std, last_w_std};
  SMN_TRY(aug_build(ctx, s, g, x_d, ldx, x_d, ldx, d, 0, nullptr, nullptr, nullptr, true));
  double quad = 0.0, ld = 0.0;
  int info = 0;
  SMN_TRY(aug_finish(ctx, dtype, g, y_d, 1, n, eps_abs, 0.0, nullptr, nullptr, 0, &quad, &ld, &info));
  if (logpdf_h) *logpdf_h = logpdf_from(quad, ld, n, df, scale, info);
  if (quad_h) *quad_h = quad;
  if (logdet_h) *logdet_h = ld;
  if (info_h) *info_h = info;
  return SMN_OK;
}

// ---- batched small problems (experiments/regression/find.py:134-199: 99 factorisations of one data set under a grid of
// (w_std, b_std, eps); train.py:178-212: thousands of steps at N = 245).  One SPR.loss at N = 245 is two workgroups on a
// 256-CU chip and a 2048-point one a hundred; G problems of identical shape -- same x, y, net, depth; their own w_std,
// b_std, last_w_std and diagonal shift -- run as ONE sequence of launches with grid.y = G: the fused build (per-problem
// layer program and tables), the prep, every panel / update launch of the factorisation, the read-out.  Problem b lives
// at workspace + b * n_total^2; per-problem logdet / info / quadratic forms come back in one copy.  Each problem executes
// exactly the arithmetic of the serial call (same kernels, same tiles, same order): bit-identical results.
namespace {

template <typename T>
__global__ void batch_trace_kernel(const T* __restrict__ a, int64_t lda, int64_t bstride, int64_t n, double* __restrict__ out) {
  __shared__ double red[256];   // one block per problem; the deterministic tree of diag_trace_kernel (cholesky.hip)
  a += (int64_t)blockIdx.x * bstride;
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += (double)a[i * lda + i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}

// aug_prep_kernel / diag_shift_kernel per problem: right-hand-side rows, diagonal shift abs[b] + rel[b] tr / n, scalar reset
template <typename T>
__global__ void batch_prep_kernel(T* __restrict__ a, int64_t lda, int64_t bstride, int64_t row0, int64_t ncols,
                                  const T* __restrict__ y, int64_t n, int64_t c, int64_t ldy, int64_t n_shift,
                                  const double* __restrict__ shift_abs, const double* __restrict__ ridge_rel,
                                  const double* __restrict__ trace, double* __restrict__ logdet, int* __restrict__ info) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t k = blockIdx.y, b = blockIdx.z;
  if (i >= ncols || k >= c) return;
  a += b * bstride;
  a[(row0 + k) * lda + i] = i < n ? y[i * ldy + k] : T(0);
  if (k == 0) {
    if (i < n_shift) {
      const double rel = ridge_rel ? ridge_rel[b] : 0.0;
      const double sh = shift_abs[b] + (rel != 0.0 ? rel * trace[b] / (double)n_shift : 0.0);
      if (sh != 0.0) a[i * lda + i] = (T)((double)a[i * lda + i] + sh);
    }
    if (i == 0) {
      logdet[b] = 0.0;
      info[b] = INT_MAX;
    }
  }
}

// extract_posterior_kernel per problem; res[b] = {logdet, info, quad[0..c)}
template <typename T>
__global__ void batch_extract_kernel(const T* __restrict__ a, int64_t lda, int64_t bstride, int64_t aug0, int64_t t, int64_t c,
                                     T* __restrict__ mean, T* __restrict__ cov, int64_t ldcov, T* __restrict__ var,
                                     double* __restrict__ res, const double* __restrict__ logdet, const int* __restrict__ info) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t b = blockIdx.z;
  a += b * bstride;
  for (int64_t i = blockIdx.y; i <= t; i += gridDim.y)
    if (i < t) {
      if (j < t && cov) {
        const int64_t hi = i > j ? i : j, lo = i > j ? j : i;
        cov[(b * t + i) * ldcov + j] = a[(aug0 + hi) * lda + aug0 + lo];
      }
      if (j == 0 && var) var[b * t + i] = a[(aug0 + i) * lda + aug0 + i];
      if (j < c && mean) mean[(b * t + i) * c + j] = -a[(aug0 + t + j) * lda + aug0 + i];
    } else {
      if (j < c) res[b * (2 + c) + 2 + j] = -(double)a[(aug0 + t + j) * lda + aug0 + t + j];
      if (j == 0) {
        res[b * (2 + c)] = logdet[b];
        res[b * (2 + c) + 1] = (double)info[b];
      }
    }
}


int spr_batch(smn_ctx* ctx, const char* who, int dtype, int net, int act, int num_hiddens, int nprob, const double* w_std,
              const double* b_std, const double* last_w_std, const void* x_d, int64_t n, int64_t ldx, const void* xt_d,
              int64_t t, int64_t ldxt, int64_t d, const void* y_d, int64_t c, const double* ridge_rel, const double* shift_abs,
              void* mean_d, void* cov_d, int64_t ldcov, void* var_d, double* quad_h, double* logdet_h, int* info_h) {
  if (dtype != SMN_F32 && dtype != SMN_F64) return smn_fail(ctx, SMN_EINVAL, "bad dtype");
  if (nprob <= 0 || !w_std || !b_std || !last_w_std || !shift_abs) return smn_fail(ctx, SMN_EINVAL, "%s: empty batch or null parameter array", who);
  if (n <= 0 || d <= 0 || t < 0 || c <= 0) return smn_fail(ctx, SMN_EINVAL, "%s: bad sizes", who);
  if (c > 48) return smn_fail(ctx, SMN_ENOTSUP, "more than 48 output columns");
  if (cov_d && ldcov < t) return smn_fail(ctx, SMN_EINVAL, "%s: ldcov < t", who);
  const size_t es = dtype_size(dtype);
  Aug g;
  g.n = n; g.t = t; g.c = c;
  g.n_pad = round_up(n, kTile);
  g.n_total = g.n_pad + round_up(t + c, kTile);
  g.lda = g.n_total;
  g.es = es;
  const size_t per = es * (size_t)g.n_total * (size_t)g.n_total;
  int chunk = (int)std::min<size_t>((size_t)nprob, std::max<size_t>(1, ctx->batch_bytes / per));
  if (chunk > 65535) chunk = 65535;   // grid.y / grid.z
  SMN_TRY(smn_workspace(ctx, 2, per * (size_t)chunk, &g.a));
  // per-problem scalars: shift_abs, ridge_rel, trace, logdet (doubles), res [2 + c], info (ints)
  const size_t nd = (size_t)chunk * (4 + 2 + (size_t)c);
  void* sv = nullptr;
  SMN_TRY(smn_workspace(ctx, 8, sizeof(double) * nd + sizeof(int) * (size_t)chunk, &sv));
  double* sh_d = static_cast<double*>(sv);
  double* rel_d = sh_d + chunk;
  double* tr_d = rel_d + chunk;
  double* ld_d = tr_d + chunk;
  double* res_d = ld_d + chunk;
  int* info_d = reinterpret_cast<int*>(res_d + (size_t)chunk * (2 + (size_t)c));
  std::vector<double> res_h((size_t)chunk * (2 + (size_t)c));
  const int64_t bstride = g.n_total * g.lda;
  BuildSpec spec{dtype, net, act, num_hiddens, w_std[0], b_std[0], last_w_std[0]};
  for (int p0 = 0; p0 < nprob; p0 += chunk) {
    const int nb = std::min(chunk, nprob - p0);
    SMN_TRY(aug_build(ctx, spec, g, x_d, ldx, t > 0 ? xt_d : x_d, t > 0 ? ldxt : ldx, d, nb, w_std + p0, b_std + p0, last_w_std + p0));
    SMN_HIP(ctx, hipMemcpyAsync(sh_d, shift_abs + p0, sizeof(double) * (size_t)nb, hipMemcpyHostToDevice, ctx->stream));
    if (ridge_rel) {
      SMN_HIP(ctx, hipMemcpyAsync(rel_d, ridge_rel + p0, sizeof(double) * (size_t)nb, hipMemcpyHostToDevice, ctx->stream));
      if (dtype == SMN_F64)
        hipLaunchKernelGGL(batch_trace_kernel<double>, dim3((unsigned)nb), dim3(256), 0, ctx->stream, static_cast<const double*>(g.a), g.lda, bstride, n, tr_d);
      else
        hipLaunchKernelGGL(batch_trace_kernel<float>, dim3((unsigned)nb), dim3(256), 0, ctx->stream, static_cast<const float*>(g.a), g.lda, bstride, n, tr_d);
    }
    {
      dim3 gp((unsigned)((g.n_total + 255) / 256), (unsigned)c, (unsigned)nb);
      if (dtype == SMN_F64)
        hipLaunchKernelGGL(batch_prep_kernel<double>, gp, dim3(256), 0, ctx->stream, static_cast<double*>(g.a), g.lda, bstride, g.n_pad + t, g.n_total,
                           static_cast<const double*>(y_d), n, c, c, n, sh_d, ridge_rel ? rel_d : nullptr, tr_d, ld_d, info_d);
      else
        hipLaunchKernelGGL(batch_prep_kernel<float>, gp, dim3(256), 0, ctx->stream, static_cast<float*>(g.a), g.lda, bstride, g.n_pad + t, g.n_total,
                           static_cast<const float*>(y_d), n, c, c, n, sh_d, ridge_rel ? rel_d : nullptr, tr_d, ld_d, info_d);
    }
    SMN_CHECK_LAUNCH(ctx);
    ctx->batch = nb; ctx->batch_stride = bstride; ctx->batch_logdet = ld_d; ctx->batch_info = info_d;
    ctx->chol_prepped = true;
    const int crc = cholesky_padded(ctx, dtype, g.a, g.n_total, g.n_pad, g.lda, n, 0.0, 0.0, false);
    ctx->chol_prepped = false;
    ctx->batch = 1; ctx->batch_stride = 0; ctx->batch_logdet = nullptr; ctx->batch_info = nullptr;
    SMN_TRY(crc);
    {
      const int64_t w = std::max<int64_t>(std::max(t, c), 1);
      dim3 ge((unsigned)((w + 255) / 256), (unsigned)std::min<int64_t>(t + 1, 4096), (unsigned)nb);
      char* mean_p = mean_d ? static_cast<char*>(mean_d) + es * (size_t)p0 * (size_t)t * (size_t)c : nullptr;
      char* cov_p = cov_d ? static_cast<char*>(cov_d) + es * (size_t)p0 * (size_t)t * (size_t)ldcov : nullptr;
      char* var_p = var_d ? static_cast<char*>(var_d) + es * (size_t)p0 * (size_t)t : nullptr;
      if (dtype == SMN_F64)
        hipLaunchKernelGGL(batch_extract_kernel<double>, ge, dim3(256), 0, ctx->stream, static_cast<const double*>(g.a), g.lda, bstride, g.n_pad, t, c,
                           reinterpret_cast<double*>(mean_p), reinterpret_cast<double*>(cov_p), ldcov, reinterpret_cast<double*>(var_p), res_d, ld_d, info_d);
      else
        hipLaunchKernelGGL(batch_extract_kernel<float>, ge, dim3(256), 0, ctx->stream, static_cast<const float*>(g.a), g.lda, bstride, g.n_pad, t, c,
                           reinterpret_cast<float*>(mean_p), reinterpret_cast<float*>(cov_p), ldcov, reinterpret_cast<float*>(var_p), res_d, ld_d, info_d);
    }
    SMN_CHECK_LAUNCH(ctx);
    SMN_HIP(ctx, hipMemcpyAsync(res_h.data(), res_d, sizeof(double) * (size_t)nb * (2 + (size_t)c), hipMemcpyDeviceToHost, ctx->stream));
    SMN_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int b = 0; b < nb; ++b) {
      const double* r = &res_h[(size_t)b * (2 + (size_t)c)];
      int info = (int)r[1];
      if (info == INT_MAX) info = 0;
      if (info_h) info_h[p0 + b] = info;
      if (logdet_h) logdet_h[p0 + b] = info ? std::nan("") : r[0];
      for (int64_t k = 0; k < c && quad_h; ++k) quad_h[(size_t)(p0 + b) * (size_t)c + (size_t)k] = info ? std::nan("") : r[2 + k];
    }
  }
  return SMN_OK;
}

}  // namespace

// Test hook: the workspace budget of one batched pass (default 48 GB; larger batches run in chunks of what fits).
extern "C" int smn_debug_panel_passes(smn_ctx* ctx, int max_passes) {
  if (!ctx || max_passes < 1 || max_passes > 64) return SMN_EINVAL;
  ctx->panel_max_passes = max_passes;
  return SMN_OK;
}

extern "C" int smn_debug_split_build(smn_ctx* ctx, int on) {
  if (!ctx) return SMN_EINVAL;
  ctx->split_build = on != 0;
  return SMN_OK;
}

extern "C" int smn_debug_batch_bytes(smn_ctx* ctx, size_t bytes) {
  if (!ctx || bytes == 0) return SMN_EINVAL;
  ctx->batch_bytes = bytes;
  return SMN_OK;
}

extern "C" int smn_spr_loss_batch(smn_ctx* ctx, int dtype, int net, int act, int num_hiddens, int nprob, const double* w_std,
                                  const double* b_std, const double* last_w_std, const void* x_d, int64_t n, int64_t ldx,
                                  int64_t d, const void* y_d, const double* eps_abs, const double* df, const double* scale,
                                  double* logpdf_h, double* quad_h, double* logdet_h, int* info_h) {
  if (!ctx || !x_d || !y_d) return SMN_EINVAL;
  SMN_ENTER(ctx);
  if (nprob <= 0 || !eps_abs) return smn_fail(ctx, SMN_EINVAL, "smn_spr_loss_batch: empty batch or null eps_abs");
  for (int b = 0; b < nprob && df; ++b)
    if (df[b] > 0.0 && !(scale && scale[b] > 0.0)) return smn_fail(ctx, SMN_EINVAL, "smn_spr_loss_batch: scale must be > 0");
  std::vector<double> quad((size_t)nprob), ld((size_t)nprob);
  std::vector<int> info((size_t)nprob);
  SMN_TRY(spr_batch(ctx, "smn_spr_loss_batch", dtype, net, act, num_hiddens, nprob, w_std, b_std, last_w_std, x_d, n, ldx, nullptr, 0, 0, d,
                    y_d, 1, nullptr, eps_abs, nullptr, nullptr, 0, nullptr, quad.data(), ld.data(), info.data()));
  for (int b = 0; b < nprob; ++b) {
    if (logpdf_h) logpdf_h[b] = logpdf_from(quad[(size_t)b], ld[(size_t)b], n, df ? df[b] : 0.0, scale ? scale[b] : 1.0, info[(size_t)b]);
    if (quad_h) quad_h[b] = quad[(size_t)b];
    if (logdet_h) logdet_h[b] = ld[(size_t)b];
    if (info_h) info_h[b] = info[(size_t)b];
  }
  return SMN_OK;
}

extern "C" int smn_spr_predict_batch(smn_ctx* ctx, int dtype, int net, int act, int num_hiddens, int nprob, const double* w_std,
                                     const double* b_std, const double* last_w_std, const void* x_d, int64_t n, int64_t ldx,
                                     const void* xt_d, int64_t t, int64_t ldxt, int64_t d, const void* y_d, int64_t c,
                                     const double* ridge_rel, const double* ridge_abs, void* mean_d, void* cov_d, int64_t ldcov,
                                     void* var_d, double* quad_h, double* logdet_h, int* info_h) {
  if (!ctx || !x_d || !y_d || (t > 0 && !xt_d)) return SMN_EINVAL;
  SMN_ENTER(ctx);
  if (nprob <= 0 || !ridge_rel) return smn_fail(ctx, SMN_EINVAL, "smn_spr_predict_batch: empty batch or null ridge_rel");
  std::vector<double> zero;
  if (!ridge_abs) {
    zero.assign((size_t)nprob, 0.0);
    ridge_abs = zero.data();
  }
  return spr_batch(ctx, "smn_spr_predict_batch", dtype, net, act, num_hiddens, nprob, w_std, b_std, last_w_std, x_d, n, ldx, xt_d, t, ldxt,
                   d, y_d, c, ridge_rel, ridge_abs, mean_d, cov_d, ldcov, var_d, quad_h, logdet_h, info_h);
}

extern "C" int smn_spr_predict(smn_ctx* ctx, int dtype, int net, int act, int num_hiddens, double w_std, double b_std,
                               double last_w_std, const void* x_d, int64_t n, int64_t ldx, const void* xt_d, int64_t t,
                               int64_t ldxt, int64_t d, const void* y_d, int64_t c, double ridge_rel, double ridge_abs,
                               void* mean_d, void* cov_d, int64_t ldcov, double* quad_h, double* logdet_h, int* info_h) {
  if (!ctx || !x_d || !y_d || (t > 0 && !xt_d)) return SMN_EINVAL;
  SMN_ENTER(ctx);
  if (dtype != SMN_F32 && dtype != SMN_F64) return smn_fail(ctx, SMN_EINVAL, "bad dtype");
  if (n <= 0 || d <= 0 || t < 0 || c <= 0) return smn_fail(ctx, SMN_EINVAL, "smn_spr_predict: bad sizes");
  if (c > 48) return smn_fail(ctx, SMN_ENOTSUP, "more than 48 output columns");
  Aug g;
  SMN_TRY(aug_alloc(ctx, dtype, n, t, c, &g));
  BuildSpec s{dtype, net, act, num_hiddens, w_std, b_std, last_w_std};
  SMN_TRY(aug_build(ctx, s, g, x_d, ldx, t > 0 ? xt_d : x_d, t > 0 ? ldxt : ldx, d, 0, nullptr, nullptr, nullptr, true, ridge_rel != 0.0));
  return aug_finish(ctx, dtype, g, y_d, c, n, ridge_abs, ridge_rel, mean_d, cov_d, ldcov, quad_h, logdet_h, info_h);
}
