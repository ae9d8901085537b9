/* smnngp.h — C-ABI of libsmnngp.so: the MI355X (gfx950) scale-mixture NNGP hot path.
 *
 * Drop-in boundary for ONE path of Hyungi-Lee/Scale-Mixtures-of-Neural-Network-Gaussian-Processes:
 * NNGP/NTK kernel-matrix build -> jittered Cholesky -> triangular solves -> Gaussian / Student-t
 * log-marginal-likelihood and predictive mean / covariance.  The reference has no FFI of its own
 * (it is pure Python over JAX + neural_tangents); every entry point below names the reference
 * call (file:line under /root/reference) whose arithmetic it replaces.  The Python side binds this
 * header with ctypes; there are no torch / JAX types anywhere in the ABI.
 *
 * Conventions
 *   - every function returns int status: SMN_OK (0) or a negative SMN_E* code; the message is
 *     kept per context and read with smn_last_error().  Nothing throws across the ABI.
 *   - matrices are dense row-major with an explicit leading dimension (elements, not bytes).
 *   - "d" pointers are DEVICE pointers obtained from smn_malloc(); "h" pointers are host memory
 *     borrowed for the duration of the call.  dtype: SMN_F32 / SMN_F64.
 *   - calls are stream-ordered on the context's stream; functions that return host scalars
 *     synchronise that stream.  A context is not thread-safe.
 *   - numerical failure is NOT an error status: a non-positive pivot is reported through *info
 *     (1-based index of the first bad pivot, 0 = fine) and the dependent outputs are NaN, which is
 *     what the reference's JAX Cholesky does (silent NaN, experiments/regression/train.py:211).
 */
#ifndef SMNNGP_H
#define SMNNGP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct smn_ctx smn_ctx;

enum { SMN_OK = 0, SMN_EINVAL = -1, SMN_EHIP = -2, SMN_ENOMEM = -3, SMN_ENOTSUP = -4, SMN_ECOMM = -5 };
enum { SMN_F32 = 0, SMN_F64 = 1 };
enum { SMN_ACT_RELU = 0, SMN_ACT_ERF = 1 };            /* experiments/nt_kernels.py:12-18           */
enum { SMN_GET_NNGP = 1, SMN_GET_NTK = 2 };            /* kernel_fn(..., get=) bit mask             */
enum { SMN_FILL_FULL = 0, SMN_FILL_LOWER = 1 };        /* symmetric build: mirror or lower triangle */
enum { SMN_NET_MLP = 0, SMN_NET_DENSE_RESNET = 1 };    /* nt_kernels.py:21-31 / :83-103             */

/* ---- context, errors, memory (JAX array semantics: the spax modules never manage memory themselves) ---- */
int smn_version(void);
int smn_device_count(int* n);
int smn_ctx_create(int device_id, smn_ctx** out);
int smn_ctx_destroy(smn_ctx* ctx);
int smn_last_error(smn_ctx* ctx, char* buf, size_t n);
int smn_synchronize(smn_ctx* ctx);
int smn_malloc(smn_ctx* ctx, size_t bytes, void** dptr);
int smn_free(smn_ctx* ctx, void* dptr);
int smn_memset(smn_ctx* ctx, void* dptr, int value, size_t bytes);
int smn_memcpy_h2d(smn_ctx* ctx, void* dst_d, const void* src_h, size_t bytes);
int smn_memcpy_d2h(smn_ctx* ctx, void* dst_h, const void* src_d, size_t bytes);
int smn_memcpy_d2d(smn_ctx* ctx, void* dst_d, const void* src_d, size_t bytes);
/* 2-D copies (row pitch in bytes) used to move unpadded host matrices into padded device ones */
int smn_memcpy2d_h2d(smn_ctx* ctx, void* dst_d, size_t dpitch, const void* src_h, size_t spitch,
                     size_t width_bytes, size_t rows);
int smn_memcpy2d_d2h(smn_ctx* ctx, void* dst_h, size_t dpitch, const void* src_d, size_t spitch,
                     size_t width_bytes, size_t rows);
/* timing hooks (hipEvents on the context's stream) */
int smn_timer_start(smn_ctx* ctx);
int smn_timer_stop_ms(smn_ctx* ctx, double* ms);           /* synchronises */
/* per-kernel timing: while enabled kernel launches are bracketed by a hipEvent pair on their own
 * stream.  category: 0 prep (pad/tables), 1 fused Gram+recursion build, 2 stand-alone recursion,
 * 3 Cholesky panel, 4 Cholesky strip update, 5 Cholesky trailing update, 6 other (scatter of gathered blocks),
 * 7 all-gathers, 8 the wait of the main stream for the FIRST piece of a column-first exchange (its exposed part),
 * 9 later waits of the factorisation for pieces the first panel chain did not cover (stalls).
 * on: 0 off; 1 every category; (2 << c) only category c (values add up to a mask).  An event pair
 * costs a few microseconds of queue time per launch, so timing ONE category perturbs a step far less
 * than timing all ~280 launches of it. */
int smn_profile_enable(smn_ctx* ctx, int on);
int smn_profile_read(smn_ctx* ctx, int category, double* total_ms, int* launches);
/* MFMA flops the launches of a category have EXECUTED since the last smn_profile_enable (whole 128x128 tiles, counted
 * on the host as they are issued): 4 strip updates, 5 trailing updates.  What a roofline
 * fraction of those kernels is priced with. */
int smn_profile_flops(smn_ctx* ctx, int category, double* flops);

/* ---- NNGP / NTK kernel build ----
 * Replaces kernel_fn(x1, x2, get) produced by get_mlp_kernel / get_dense_resnet_kernel
 * (experiments/nt_kernels.py:21-31, :83-103; called from spax/kernels.py:23-27 and
 * experiments/regression/find.py:64-70): K0 = x1 x2^T / d on MFMA, then num_hiddens x
 * [Dense(w_std,b_std); act] and Dense(last_w_std, b=0) fused into the GEMM epilogue.
 *   x1_d [n1,d] ld=ldx1; x2_d [n2,d] or NULL (symmetric: x2 = x1, fill selects mirror/lower).
 *   get_mask: SMN_GET_NNGP | SMN_GET_NTK; nngp_d [n1,n2] ld=ldk (may be NULL if not requested),
 *   ntk_d likewise. */
int smn_kernel_mlp(smn_ctx* ctx, int dtype, int net, int act, int num_hiddens,
                   double w_std, double b_std, double last_w_std,
                   const void* x1_d, int64_t n1, int64_t ldx1,
                   const void* x2_d, int64_t n2, int64_t ldx2, int64_t d,
                   int get_mask, int fill,
                   void* nngp_d, void* ntk_d, int64_t ldk);

/* Row-sharded variant (multi-GPU build): computes rows [row_begin,row_end)
 * of the symmetric kernel of x_d against all n columns into out_d [(row_end-row_begin), n]. */
int smn_kernel_mlp_rows(smn_ctx* ctx, int dtype, int net, int act, int num_hiddens,
                        double w_std, double b_std, double last_w_std,
                        const void* x_d, int64_t n, int64_t ldx, int64_t d,
                        int64_t row_begin, int64_t row_end, int get_mask,
                        void* nngp_rows_d, void* ntk_rows_d, int64_t ldk);

/* Balanced symmetric shard: rows [row_begin,row_end) x columns [0,row_end) only (the lower
 * trapezoid of that row block; 128x128 tiles wholly above the diagonal are skipped and what lies
 * right of the diagonal inside the written range is unspecified except that the diagonal itself is
 * exact).  out_d [(row_end-row_begin), >= row_end] with ld = ldk; see smn_unpack_lower_blocks. */
int smn_kernel_mlp_lower_rows(smn_ctx* ctx, int dtype, int net, int act, int num_hiddens,
                              double w_std, double b_std, double last_w_std,
                              const void* x_d, int64_t n, int64_t ldx, int64_t d,
                              int64_t row_begin, int64_t row_end, int get_mask,
                              void* nngp_rows_d, void* ntk_rows_d, int64_t ldk);

/* One rank's whole share of the paired layout in ONE launch: the lower trapezoids of row blocks
 * `rank` and 2*nranks-1-rank (block_rows rows each, a multiple of 128), packed into
 * nngp_chunk_d / ntk_chunk_d [block_rows^2 * (2*nranks+1)] exactly as smn_unpack_lower_blocks
 * expects them after the all-gather. */
int smn_kernel_mlp_shard(smn_ctx* ctx, int dtype, int net, int act, int num_hiddens,
                         double w_std, double b_std, double last_w_std,
                         const void* x_d, int64_t n, int64_t ldx, int64_t d,
                         int nranks, int rank, int64_t block_rows, int get_mask,
                         void* nngp_chunk_d, void* ntk_chunk_d);

/* The two halves of the build, exposed separately for sweeps that reuse K0 across (w_std,b_std)
 * (experiments/regression/find.py:134-138) and for roofline measurement of the recursion alone.
 * smn_gram: k0_d = x1 x2^T / d (+ q1_d [n1], q2_d [n2] diagonals ||x||^2/d).
 * smn_recursion: applies the layer stack elementwise to k0 (HBM-streaming kernel). */
int smn_gram(smn_ctx* ctx, int dtype, const void* x1_d, int64_t n1, int64_t ldx1,
             const void* x2_d, int64_t n2, int64_t ldx2, int64_t d,
             void* k0_d, int64_t ldk, void* q1_d, void* q2_d);
int smn_recursion(smn_ctx* ctx, int dtype, int net, int act, int num_hiddens,
                  double w_std, double b_std, double last_w_std,
                  const void* k0_d, int64_t n1, int64_t n2, int64_t ldk0,
                  const void* q1_d, const void* q2_d, int symmetric, int get_mask,
                  void* nngp_d, void* ntk_d, int64_t ldk);

/* conv-NNGP (experiments/nt_kernels.py:34-45): x [n,H,W,C] NHWC, 3x3 SAME stride 1, Flatten, Dense. */
int smn_kernel_cnn(smn_ctx* ctx, int dtype, int act, int num_hiddens,
                   double w_std, double b_std, double last_w_std,
                   const void* x1_d, int64_t n1, const void* x2_d, int64_t n2,
                   int64_t H, int64_t W, int64_t C, int fill, void* nngp_d, int64_t ldk);

/* Conv-ResNet NNGP kernel: experiments/nt_kernels.py:48-80 get_conv_resnet_kernel = WideResnet(block_size, k=1)
 * without pooling: Conv; four groups of block_size residual blocks (strides 1,2,2,2; Conv shortcut in the first
 * block of a group, Identity after; main path act, Conv(stride), act, Conv); Flatten; Dense(last_w_std).
 * Same argument meaning as smn_kernel_cnn; H and W must be multiples of 8; block_size <= 6. */
int smn_kernel_conv_resnet(smn_ctx* ctx, int dtype, int act, int block_size,
                           double w_std, double b_std, double last_w_std,
                           const void* x1_d, int64_t n1, const void* x2_d, int64_t n2,
                           int64_t H, int64_t W, int64_t C, int fill, void* nngp_d, int64_t ldk);

/* ---- factorisation and solves ----
 * smn_cholesky: in-place lower Cholesky of the leading n_factor x n_factor block of the symmetric
 * matrix a_d [n_total,n_total] (lower triangle read/written), carried through the remaining
 * n_total-n_factor rows: on return rows >= n_factor hold B L^-T in their first n_factor columns
 * and the Schur complement C - B A^-1 B^T in the trailing block (lower triangle).  With
 * n_total == n_factor this is a plain potrf.  Before factoring, jitter_abs + ridge_rel*tr(A)/n
 * is added to the first n_shift diagonal entries: jitter_abs is spax/utils.py:26-27 +
 * spax/models.py:96 (absolute), ridge_rel is neural_tangents' diag_reg scaling used by
 * spax/kernels.py:29-32 (trace taken over those n_shift entries).
 * Replaces lax.linalg.cholesky + triangular_solve (spax/utils.py:179-180), the Cholesky inside
 * jax.scipy.stats.multivariate_normal.logpdf (spax/likelihoods.py:27) and cho_factor/cho_solve
 * inside gradient_descent_mse_ensemble.  *logdet_h = 2 sum log L_ii over n_factor columns. */
int smn_cholesky(smn_ctx* ctx, int dtype, void* a_d, int64_t n_total, int64_t n_factor, int64_t lda,
                 int64_t n_shift, double jitter_abs, double ridge_rel, int* info_h, double* logdet_h);

/* X = op(L)^-1 B in place, B [n,nrhs] row-major ld=ldb, L lower [n,n]; trans=0: L, 1: L^T.
 * (lax.linalg.triangular_solve, spax/utils.py:180.) */
int smn_trsm(smn_ctx* ctx, int dtype, const void* l_d, int64_t n, int64_t ldl,
             void* b_d, int64_t nrhs, int64_t ldb, int trans);
/* dst_d[c, r] = src_d[r, c] for r < rows, c < cols (row-major, leading dimensions in elements; no overlap).  The NTK
 * posterior of gradient_descent_mse_ensemble (sample.ipynb:194-195) needs Theta~^-1 Theta_dt as a left operand. */
int smn_transpose(smn_ctx* ctx, int dtype, void* dst_d, int64_t ldd, const void* src_d, int64_t lds, int64_t rows,
                  int64_t cols);

/* ---- likelihood heads (host scalars out) ----
 * smn_lml: log-marginal likelihood of y_d [n] under cov = K + eps I, K given as k_d [n,n] lower
 * (destroyed: overwritten by its factor).  df <= 0: Gaussian (spax/likelihoods.py:25-28);
 * df > 0: multivariate Student-t with shape = scale*cov, df = 2a, scale = b/a
 * (spax/likelihoods.py:45-50, spax/utils.py:178-183).  NaN + info>0 when not PD. */
int smn_lml(smn_ctx* ctx, int dtype, void* k_d, int64_t n, int64_t ldk, const void* y_d,
            double eps_abs, double df, double scale, double* logpdf_h, double* quad_h,
            double* logdet_h, int* info_h);

/* smn_predict: t=infinity NNGP posterior (spax/kernels.py:29-32 -> neural_tangents
 * gradient_descent_mse_ensemble).  kj_d is the JOINT kernel of [x_train; x_test] ([n+t, n+t],
 * lower triangle, destroyed).  y_d [n,c] row-major.  mean_d [t,c], cov_d [t,t] (full, symmetric).
 * ridge_rel = diag_reg (relative: * tr(K_dd)/n); ridge_abs adds an absolute term.
 * quad_h (may be NULL) receives y_k^T K~^-1 y_k for each output column k. */
int smn_predict(smn_ctx* ctx, int dtype, void* kj_d, int64_t n, int64_t t, int64_t ldk,
                const void* y_d, int64_t c, double ridge_rel, double ridge_abs,
                void* mean_d, void* cov_d, int64_t ldcov, double* quad_h, double* logdet_h, int* info_h);

/* ---- fused model-level calls (what the spax facade uses; nothing leaves the GPU but scalars) ----
 * smn_spr_loss: SPR.loss (spax/models.py:93-98): builds K(x,x) straight into the factorisation
 * workspace, adds eps_abs, factors, and returns the log-pdf of y (Gaussian df<=0 / Student-t). */
int smn_spr_loss(smn_ctx* ctx, int dtype, int net, int act, int num_hiddens,
                 double w_std, double b_std, double last_w_std,
                 const void* x_d, int64_t n, int64_t ldx, int64_t d, const void* y_d,
                 double eps_abs, double df, double scale,
                 double* logpdf_h, double* quad_h, double* logdet_h, int* info_h);
/* smn_spr_predict: NNGPKernel.predict (spax/kernels.py:29-32) from the raw inputs: joint kernel of
 * [x; x_test], relative/absolute ridge on the training block, posterior mean [t,c] and cov [t,t]. */
int smn_spr_predict(smn_ctx* ctx, int dtype, int net, int act, int num_hiddens,
                    double w_std, double b_std, double last_w_std,
                    const void* x_d, int64_t n, int64_t ldx, const void* xt_d, int64_t t, int64_t ldxt,
                    int64_t d, const void* y_d, int64_t c, double ridge_rel, double ridge_abs,
                    void* mean_d, void* cov_d, int64_t ldcov,
                    double* quad_h, double* logdet_h, int* info_h);

/* ---- batched small problems: G evaluations on ONE data set in one sequence of launches (grid.y = G) ----
 * The reference's real workloads are small and many: experiments/regression/find.py:134-199 factors the kernel of one data
 * set 2 x 99 times under a grid of (w_std, b_std, eps), train.py:178-212 takes thousands of steps at N = 245, where one
 * SPR.loss occupies two workgroups of a 256-CU chip.  These calls run nprob problems of identical shape -- same x, y, net,
 * act and depth; per-problem w_std[], b_std[], last_w_std[] and diagonal shift (host arrays of nprob doubles) -- through
 * the fused build, the factorisation and the read-out together; each problem's result is bit-identical to the serial call.
 *   smn_spr_loss_batch     nprob x smn_spr_loss: eps_abs[], df[] (NULL: Gaussian), scale[] -> logpdf_h[], quad_h[], logdet_h[],
 *                          info_h[] (any output may be NULL)
 *   smn_spr_predict_batch  nprob x smn_spr_predict: ridge_rel[], ridge_abs[] (NULL: 0) -> mean_d [nprob,t,c], cov_d
 *                          [nprob,t,ldcov] and / or var_d [nprob,t] = diag(cov) (what find.py:50-55 uses; either may be
 *                          NULL), quad_h [nprob,c], logdet_h[], info_h[]
 * Batches whose workspaces (nprob x (n_pad + (t+c)_pad)^2 elements) exceed 48 GB run in chunks (smn_debug_batch_bytes: test
 * hook that sets that budget). */
int smn_spr_loss_batch(smn_ctx* ctx, int dtype, int net, int act, int num_hiddens, int nprob,
                       const double* w_std, const double* b_std, const double* last_w_std,
                       const void* x_d, int64_t n, int64_t ldx, int64_t d, const void* y_d,
                       const double* eps_abs, const double* df, const double* scale,
                       double* logpdf_h, double* quad_h, double* logdet_h, int* info_h);
int smn_debug_batch_bytes(smn_ctx* ctx, size_t bytes);
/* Test hook: smn_spr_loss under the look-ahead (n_total >= 8192) builds the bottom-right corner of the kernel matrix as a second
 * launch on the bulk stream, beside the first super-panel's panel chain (same tiles, same bits: only the order changes).
 * on = 0 switches that off (one launch, as every other entry point builds). */
int smn_debug_split_build(smn_ctx* ctx, int on);
/* Test hook: a panel workgroup of the factorisation carries up to max_passes groups of rows when there are more groups than CUs
 * (default 4; the later groups ride through the solve alone; same bits).  1 = one group per workgroup. */
int smn_debug_panel_passes(smn_ctx* ctx, int max_passes);
int smn_spr_predict_batch(smn_ctx* ctx, int dtype, int net, int act, int num_hiddens, int nprob,
                          const double* w_std, const double* b_std, const double* last_w_std,
                          const void* x_d, int64_t n, int64_t ldx, const void* xt_d, int64_t t, int64_t ldxt,
                          int64_t d, const void* y_d, int64_t c, const double* ridge_rel, const double* ridge_abs,
                          void* mean_d, void* cov_d, int64_t ldcov, void* var_d,
                          double* quad_h, double* logdet_h, int* info_h);

/* ---- predictive NLL under a sampled scale mixture (experiments/regression/find.py:165-187) ----
 * For nprob problems with posterior mean_d / var_d [nprob,t] (normalised units; smn_spr_predict_batch), quad_h / logdet_h [nprob]
 * (smn_spr_loss_batch) and nmix proposals of nsamples sigma^2 draws each (sample_q_h [nmix,nsamples]; ratio_h = prior /
 * proposal density per draw, NULL = 1, which is what find.py:168-169 evaluates to):
 *   tnll_h[p, m] = - mean_t logsumexp_s [ log(w~_s + 1e-24) + log N(y_test_t; mean_t y_std + y_mean, sqrt(q_s var_t) y_std) ]
 * with w~ the self-normalised weights of log p(y_train | q_s) = -(n/2) log 2 pi - logdet/2 - quad/(2 q_s) - (n/2) log q_s.
 * skip_h[p] != 0 (may be NULL): NaN for that problem.  fp64 arithmetic throughout; y_test_h in original units. */
int smn_mixture_nll(smn_ctx* ctx, int dtype, int nprob, int64_t t, const void* mean_d, const void* var_d,
                    const double* quad_h, const double* logdet_h, const int* skip_h, const double* y_test_h,
                    double y_mean, double y_std, int64_t n, int nmix, int nsamples, const double* sample_q_h,
                    const double* ratio_h, double* tnll_h);

/* ---- hyper-parameter gradients of the log-marginal likelihood ----
 * What objax.GradValues(model.loss, vars) supplies to experiments/regression/train.py:61-67.
 * With K~ = K(w_std, b_std, last_w_std) + eps I, alpha = K~^-1 y and G = coef * alpha alpha^T - K~^-1:
 *     terms_h[0..3] = sum_ij G_ij dK~_ij/d{w_std, b_std, last_w_std, eps}      (so d logpdf/d theta = terms/2)
 * smn_lml_grad_terms: the contraction alone.  k0_d [n,n] = X X^T / d and q_d [n] its diagonal (smn_gram),
 *   neg_kinv_d [n,n] = -K~^-1 and alpha_d [n] as smn_predict returns them for K_td = I, K_tt = 0
 *   (covariance = -K~^-1, mean = alpha).  coef = 1 (Gaussian) or (df+n)/((df + quad/scale) scale) (Student-t).
 * smn_spr_loss_grad: everything from X and y (MLP / dense ResNet kernels): also returns quad = y^T K~^-1 y,
 *   logdet K~ and info, from which the host forms the log-pdf and the derivatives w.r.t. (a, b).
 *   On a non-PD matrix info > 0 and the terms are NaN. */
int smn_lml_grad_terms(smn_ctx* ctx, int dtype, int net, int act, int num_hiddens,
                       double w_std, double b_std, double last_w_std,
                       const void* k0_d, int64_t n, int64_t ldk0, const void* q_d,
                       const void* neg_kinv_d, int64_t ldkinv, const void* alpha_d,
                       double coef, double terms_h[4]);
int smn_spr_loss_grad(smn_ctx* ctx, int dtype, int net, int act, int num_hiddens,
                      double w_std, double b_std, double last_w_std,
                      const void* x_d, int64_t n, int64_t ldx, int64_t d, const void* y_d,
                      double eps_abs, double df, double scale,
                      double* quad_h, double* logdet_h, int* info_h, double terms_h[4]);

/* ---- multi-GPU (nothing in the reference to mirror) ----
 * One process per GPU.  Rank 0 calls smn_comm_unique_id and ships the 128 bytes to the other
 * ranks by any host channel; every rank then calls smn_comm_init.  smn_allgather is an RCCL
 * all-gather of `count` elements per rank on the context's stream; the caller states the world it
 * believes it is in: SMN_ECOMM when that is not the communicator's size (a context without a
 * communicator is a world of one, where the gather is a copy) -- a world > 1 call can never quietly
 * gather nothing. */
int smn_comm_unique_id(char id_out[128]);
int smn_comm_init(smn_ctx* ctx, int nranks, int rank, const char id[128]);
int smn_comm_destroy(smn_ctx* ctx);
int smn_allgather(smn_ctx* ctx, int nranks, int dtype, const void* send_d, void* recv_d, int64_t count);
/* Paired lower-trapezoid layout (host side: sharding.py).  The n rows are cut into 2*nranks blocks
 * of block_rows rows (a multiple of 128); rank r owns blocks r and 2*nranks-1-r and packs block b
 * densely as block_rows rows of leading dimension (b+1)*block_rows, low block first, so every rank
 * contributes block_rows^2 * (2*nranks+1) elements to ONE smn_allgather.  This call scatters the
 * gathered stage_d [nranks * that count] into the lower triangle (by 128-column tiles) of
 * k_d [n,n] ld=ldk in natural row order. */
int smn_unpack_lower_blocks(smn_ctx* ctx, int dtype, const void* stage_d, int64_t n, int nranks,
                            int64_t block_rows, void* k_d, int64_t ldk);
/* smn_lml (same outputs, same likelihood arguments) fed from the gathered staging buffer: the blocks are
 * scattered straight into the factorisation workspace, K is never assembled separately. */
int smn_lml_from_blocks(smn_ctx* ctx, int dtype, const void* stage_d, int64_t n, int nranks,
                        int64_t block_rows, const void* y_d, double eps_abs, double df, double scale,
                        double* logpdf_h, double* quad_h, double* logdet_h, int* info_h);
/* nranks / rank of the context's communicator (1 / 0 without one). */
int smn_comm_info(smn_ctx* ctx, int* nranks, int* rank);

/* ---- column-first exchange: the factorisation starts on the columns that have arrived ----
 * Cyclic layout (host side: sharding.py; nothing in the reference to mirror, run.py:17 only masks devices).  The 128-row
 * tile rows are dealt to the ranks in boustrophedon order with period 2*nranks: group j = tile rows [j P, (j+1) P), rank r
 * owns t_j(r) = j P + (j even ? r : P-1-r).  Every rank builds the same number of lower tiles and every aligned group of P
 * tile rows holds one tile row per rank, so any tile-COLUMN range [c0, c1) with c0 a multiple of P is an equal-count
 * all-gather: ceil((T - c0) / P) strips of 128 x (c1-c0)*128 elements per rank (T = ceil(n / 128)).  piece_cols[0..npieces]
 * are those boundaries (0 = piece_cols[0] < ... < piece_cols[npieces] = T, at most 16 pieces); a rank's chunk holds its
 * pieces one after the other, the staging buffer piece g of all ranks at nranks * (offset of piece g).
 *   smn_shard_begin            the factorisation workspace of smn_lml_from_shards, made ready before the first piece lands;
 *                              eps_abs is the absolute jitter (spax/models.py:96), added to the diagonal as it is scattered
 *   smn_kernel_mlp_shard_cols  the rank's whole share of the build in ONE launch, into its chunk nngp_chunk_d / ntk_chunk_d
 *   smn_shard_exchange_cols    piece `piece`: RCCL all-gather on the context's communication stream (ordered after everything
 *                              issued so far on its main stream), scatter into the workspace on a third stream; returns at
 *                              once.  nranks must equal the communicator's size (SMN_ECOMM otherwise)
 *   smn_shard_exchange_cols_to the same, scattered into k_d [n,n] ld=ldk (lower triangle by 128-column tiles): the NTK of a
 *                              joint NNGP + NTK shard (BASELINE config 5; the reference's only use: sample.ipynb cells 194-195)
 *   smn_shard_scatter_cols     the scatter half alone, from a staging buffer the caller filled on the main stream (k_d NULL:
 *                              into the workspace, as smn_shard_exchange_cols does)
 *   smn_shard_wait             the main stream waits for every piece issued so far
 *   smn_lml_from_shards        factorisation + head, same outputs as smn_lml.  The factorisation waits for the pieces one by
 *                              one as it reaches their columns: only the first piece is exposed, the rest of the exchange
 *                              rides under the first super-panel's panel chain
 *   smn_debug_delay            test hook: holds stream 0 (main) / 1 (communication) / 2 (scatter) for usec microseconds */
int smn_shard_begin(smn_ctx* ctx, int dtype, int64_t n, double eps_abs);
int smn_kernel_mlp_shard_cols(smn_ctx* ctx, int dtype, int net, int act, int num_hiddens,
                              double w_std, double b_std, double last_w_std,
                              const void* x_d, int64_t n, int64_t ldx, int64_t d,
                              int nranks, int rank, int npieces, const int64_t* piece_cols,
                              int get_mask, void* nngp_chunk_d, void* ntk_chunk_d);
int smn_shard_exchange_cols(smn_ctx* ctx, int dtype, const void* mine_d, void* stage_d, int64_t n, int nranks,
                            int npieces, const int64_t* piece_cols, int piece);
int smn_shard_exchange_cols_to(smn_ctx* ctx, int dtype, const void* mine_d, void* stage_d, int64_t n, int nranks,
                               int npieces, const int64_t* piece_cols, int piece, void* k_d, int64_t ldk);
int smn_shard_scatter_cols(smn_ctx* ctx, int dtype, const void* stage_d, int64_t n, int nranks, int npieces,
                           const int64_t* piece_cols, int piece, void* k_d, int64_t ldk);
int smn_shard_wait(smn_ctx* ctx);
int smn_lml_from_shards(smn_ctx* ctx, int dtype, int64_t n, const void* y_d, double df, double scale,
                        double* logpdf_h, double* quad_h, double* logdet_h, int* info_h);
int smn_debug_delay(smn_ctx* ctx, int stream_id, int64_t usec);

#ifdef __cplusplus
}
#endif
#endif /* SMNNGP_H */
