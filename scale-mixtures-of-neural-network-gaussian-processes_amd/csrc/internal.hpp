// internal.hpp — C++-side interfaces between the translation units of libsmnngp.so.
#pragma once
#include "common.hpp"

constexpr int kTile = 128;   // GEMM block edge; every padded dimension is a multiple of it
constexpr int kMaxColPieces = 32;   // pieces of a column-first exchange

// Cyclic column-first shard of the symmetric kernel build over P ranks (host side: sharding.py, same formulas).
//   tile rows (128 rows) are dealt in boustrophedon order with period 2P: group j = tile rows [jP, (j+1)P), rank r owns
//   t_j(r) = jP + (j even ? r : P-1-r)  -- tile row t holds t+1 lower tiles, and every pair of groups gives every rank the same
//   number of them, so the build is balanced and EVERY aligned group of P tile rows holds exactly one tile row per rank;
//   piece g = tile columns [c[g], c[g+1]) of every tile row from the group of c[g] down (group f = floor(c[g] / P)): slots =
//   ceil(T / P) - f strips of 128 x (c[g+1]-c[g])*128 elements per rank -- the same count on every rank, so ONE equal-count
//   all-gather moves a whole column range of the lower triangle (tile rows that start inside or above it carry their
//   above-diagonal tiles as padding: at most one strip per rank and piece when c[g] is not a multiple of P).
struct ColPieces {
  int P = 1, np = 0;
  int64_t T = 0;                       // tile rows of the kernel: ceil(n / 128)
  int64_t c[kMaxColPieces + 1] = {};   // tile-column boundaries, c[0] = 0, c[np] = T
  int64_t off[kMaxColPieces + 1] = {}; // element offset of piece g in a rank's chunk; off[np] = elements per rank
  int64_t slots(int g) const { return (T + P - 1) / P - c[g] / P; }
  int64_t width(int g) const { return (c[g + 1] - c[g]) * kTile; }
  int64_t count(int g) const { return slots(g) * kTile * width(g); }
};
int col_pieces_make(smn_ctx* ctx, int64_t n, int nranks, int npieces, const int64_t* piece_cols, ColPieces* out);

struct BuildSpec {
  int dtype, net, act, num_hiddens;
  double w_std, b_std, last_w_std;
};

enum { STORE_BOUNDS = 0, STORE_PAD_IDENTITY = 1 };

// One fused Gram + layer-recursion launch.  Operands are PADDED copies (rows a multiple of 128,
// K a multiple of 32 elements, zero filled) made by pad_rows(); q1/q2 are ||x||^2/d per padded row.
struct BuildCall {
  BuildSpec spec;
  const void* x1p; int64_t ld1; int64_t rows1;
  const void* x2p; int64_t ld2; int64_t rows2;
  int kp; int64_t d;
  const double* q1; const double* q2;
  int symmetric;               // 1: x2 == x1, only tiles with tc <= tr are computed
  int mirror;                  // symmetric only: also store the transposed tile (full matrix out)
  int lower_skip;              // rectangular only: skip tiles wholly above the global diagonal (row shards)
  int64_t row_off, col_off;    // added to local indices for the row == col (exact diagonal) test
  int exact_diag;              // write the closed-form diagonal where global row == global col
  int store_mode;              // STORE_BOUNDS: write [0,out_rows) x [0,out_cols) only
  int64_t out_rows, out_cols;  // STORE_PAD_IDENTITY: write everything, identity outside `valid`
  int64_t nv0, aug0, nv1;      // valid(i) = i < nv0 || (aug0 <= i < aug0 + nv1)
  int get_mask;
  void* out_k; void* out_t; int64_t ldo;
  // paired lower-block shard (symmetric operands): two tile-aligned row blocks [rb,re), each written as
  // rows x columns [0,re) into its own packed output of leading dimension shard_ld
  int shard; int64_t shard_rb[2], shard_re[2]; void* shard_k[2]; void* shard_t[2]; int64_t shard_ld[2];
  // shard == 2: cyclic column-first shard (ColPieces below); shard_k[0] / shard_t[0] = the rank's chunk
  int cy_P, cy_rank, cy_np; int64_t cy_c[kMaxColPieces + 1]; int64_t cy_off[kMaxColPieces];
  // batched build (nbatch > 0): the same operands under nbatch layer programs that differ in (w_std, b_std, last_w_std)
  // only -- host arrays bw / bb / blw --, problem g written at out_k + g * out_bs elements
  int nbatch; const double* bw; const double* bb; const double* blw; int64_t out_bs;
  // split_corner = TB > 0 (symmetric, un-sharded, one problem): the tiles with row AND column >= T - TB go out as a second
  // launch on the bulk stream; ctx->corner_col / ev_corner tell the factorisation (cholesky.hip need_columns) when they landed
  int split_corner;
  // want_trace: leave sum_i<nv0 K_ii (from the exact-diagonal table) in ctx->d_scal[1] and set ctx->trace_ready
  int want_trace;
};
int run_build(smn_ctx* ctx, const BuildCall& c);
// TB for a split build of T tile rows on this context (0: do not split)
int split_corner_tiles(const smn_ctx* ctx, int64_t tiles);

// dst[rows_pad, kp] (ld = kp) <- zero-padded copy of src[n, d]; also q[rows_pad] = ||row||^2 / d.
// rows_a > 0: rows [rows_a, rows_pad) are src2[n2, d] (zero-padded like the first block): one launch for both blocks of an
// augmented operand
int pad_rows(smn_ctx* ctx, int dtype, const void* src, int64_t n, int64_t lds, int64_t d,
             void* dst, int64_t rows_pad, int64_t kp, double* q, int64_t rows_a = 0, const void* src2 = nullptr, int64_t n2 = 0,
             int64_t lds2 = 0);

inline int64_t k_pad(int dtype, int64_t d) { return round_up(d, dtype == SMN_F64 ? 16 : 32); }

// Partial Cholesky on a padded matrix (n_total, n_factor multiples of 128).  Device-side results:
// logdet (double) and info (int) are left in ctx->d_scal[0] / ctx->d_info[0]; no host sync.
int cholesky_padded(smn_ctx* ctx, int dtype, void* a, int64_t n_total, int64_t n_factor, int64_t lda,
                    int64_t n_shift, double jitter_abs, double ridge_rel, bool keep_factor,
                    int64_t id0 = -1, int64_t id1 = -1);   // id0/id1: appended rows [id0, id1) are an identity block
int predict_joint(smn_ctx* ctx, int dtype, void* kj_d, int64_t n, int64_t t, int64_t ldk, const void* y_d, int64_t c,
                  double ridge_rel, double ridge_abs, void* mean_d, void* cov_d, int64_t ldcov, double* quad_h,
                  double* logdet_h, int* info_h);
// alpha = K~^-1 y and -K~^-1 for the analytic gradients: a no-Schur factorisation of the rectangle [[K~], [I], [y^T]] (K from
// the Gram matrix k0 by the layer recursion, in place), then -L^-T L^-1 as one launch (heads.hip)
int factor_with_identity(smn_ctx* ctx, int dtype, int net, int act, int num_hiddens, double w_std, double b_std,
                         double last_w_std, const void* k0_d, int64_t ldk0, const void* q_d, int64_t n, const void* y_d,
                         double eps_abs, void* alpha_d, void* ninv_d, int64_t ldinv, double* quad_h, double* logdet_h,
                         int* info_h);
int fetch_logdet_info(smn_ctx* ctx, double* logdet, int* info);
int gram_lower(smn_ctx* ctx, int dtype, const void* x_d, int64_t n, int64_t ldx, int64_t d, void* k0_d, int64_t ldk, void* q_d);
// -x x^T (lower, into neg_inv [n, n] ld = ldo) and alpha = x z from the rows x [n, kcols] (ld = ldx, row i zero left of its
// 128-column tile) and the vector z [kcols]; *quad_dev = z^T z.  cholesky.hip.
int inverse_from_rows(smn_ctx* ctx, int dtype, const void* x, int64_t ldx, const void* z, int64_t kcols, int64_t n,
                      void* neg_inv, int64_t ldo, void* alpha, double* quad_dev);
// logdet, info and nq device doubles (quadratic forms) through the pinned mailbox: one tiny kernel + ONE synchronisation
int fetch_results(smn_ctx* ctx, const double* quad_dev, int nq, double* quad_h, double* logdet, int* info);

// small helpers implemented in util.hip
int fill_identity_pad(smn_ctx* ctx, int dtype, void* a, int64_t lda, int64_t n_pad, int64_t n_valid);
int copy_matrix(smn_ctx* ctx, int dtype, void* dst, int64_t ldd, const void* src, int64_t lds,
                int64_t rows, int64_t cols, int lower_only);
// a[row0 + k, i] = (i < n ? y[i*ldy + k] : 0) for k < c, i < ncols   (whole rows are written)
int set_aug_rows(smn_ctx* ctx, int dtype, void* a, int64_t lda, int64_t row0, int64_t ncols, const void* y,
                 int64_t n, int64_t c, int64_t ldy);
// predictive read-out of a factored augmented matrix (see heads.hip)
int extract_posterior(smn_ctx* ctx, int dtype, const void* a, int64_t lda, int64_t aug0, int64_t t, int64_t c,
                      void* mean, void* cov, int64_t ldcov, double* quad_dev, bool publish = false);
// set_aug_rows + absolute diagonal shift + reset of logdet / info in one launch (cholesky_padded then runs with
// ctx->chol_prepped set and skips its own two)
// (columns [col0, ncols) only, on stream st: the corner of a split build is prepped behind its own launch)
// ridge_rel != 0: the shift is jitter_abs + ridge_rel * d_scal[1] / n_trace (the trace left there by the build: want_trace)
int aug_prep(smn_ctx* ctx, int dtype, void* a, int64_t lda, int64_t row0, int64_t ncols, const void* y, int64_t n,
             int64_t c, int64_t ldy, int64_t n_shift, double jitter_abs, int64_t col0 = 0, hipStream_t st = nullptr,
             double ridge_rel = 0.0, int64_t n_trace = 0);
// the mailbox after a launch that published into it (extract_posterior(..., publish = true)): synchronise and read
int fetch_mail(smn_ctx* ctx, int nq, double* quad_h, double* logdet, int* info);
int solve_rows_padded(smn_ctx* ctx, int dtype, void* a, int64_t n_total, int64_t n_factor, int64_t lda);
int transpose_matrix(smn_ctx* ctx, int dtype, void* dst, int64_t ldd, const void* src, int64_t lds,
                     int64_t rows, int64_t cols);   // dst[c, r] = src[r, c]
// dst[i, j] = src[n-1-j, n-1-i] for j <= i  (J L^T J);  transpose with the src rows / dst rows reversed
int flip_transpose_lower(smn_ctx* ctx, int dtype, void* dst, int64_t ldd, const void* src, int64_t lds, int64_t n);
int transpose_flip(smn_ctx* ctx, int dtype, void* dst, int64_t ldd, const void* src, int64_t lds, int64_t rows,
                   int64_t cols, int flip_src_rows, int flip_dst_rows);
// column-first exchange (comm.hip): the all-gather of piece g on stream `st`, and its scatter into k (lower triangle by
// 128-column tiles; diag_add is added to the diagonal entries as they are written: the absolute jitter of SPR.loss)
int allgather_piece_on(smn_ctx* ctx, hipStream_t st, int dtype, const void* mine_d, void* stage_d, const ColPieces& cp, int g);
int scatter_piece_on(smn_ctx* ctx, hipStream_t st, int dtype, const void* stage_d, int64_t n, const ColPieces& cp, int g,
                     void* k_d, int64_t ldk, double diag_add);
