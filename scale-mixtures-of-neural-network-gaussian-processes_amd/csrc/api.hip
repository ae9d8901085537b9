// api.hip — context, memory and utility entry points of the C-ABI (include/smnngp.h).
#include <climits>

#include "internal.hpp"

int smn_allow_lds(smn_ctx* ctx, const void* kernel, size_t lds) {
  size_t& have = ctx->max_lds[kernel];
  if (lds <= have) return SMN_OK;
  SMN_HIP(ctx, hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  have = lds;
  return SMN_OK;
}

// Every stream of the context other than the main one has drained (before memory they may touch is freed).
static int sync_side_streams(smn_ctx* ctx) {
  if (ctx->stream_comm) SMN_HIP(ctx, hipStreamSynchronize(ctx->stream_comm));
  if (ctx->stream_scatter) SMN_HIP(ctx, hipStreamSynchronize(ctx->stream_scatter));
  if (ctx->stream_bulk) SMN_HIP(ctx, hipStreamSynchronize(ctx->stream_bulk));
  return SMN_OK;
}

int smn_workspace(smn_ctx* ctx, int slot, size_t bytes, void** out) {
  if (slot < 0 || slot >= smn_ctx::kSlots) return smn_fail(ctx, SMN_EINVAL, "bad workspace slot");
  if (ctx->ws_bytes[slot] < bytes) {
    if (slot == 2) {   // a column-first exchange in flight must not scatter into the freed buffer
      ctx->shard_a = nullptr;
      ctx->arrivals.clear();
    }
    if (ctx->ws[slot]) {
      SMN_TRY(sync_side_streams(ctx));
      SMN_HIP(ctx, hipStreamSynchronize(ctx->stream));
      SMN_HIP(ctx, hipFree(ctx->ws[slot]));
      ctx->ws[slot] = nullptr;
      ctx->ws_bytes[slot] = 0;
    }
    const size_t want = bytes + bytes / 8 + 4096;
    SMN_HIP(ctx, hipMalloc(&ctx->ws[slot], want));
    ctx->ws_bytes[slot] = want;
  }
  *out = ctx->ws[slot];
  return SMN_OK;
}

namespace {

template <typename T>
__global__ void copy_matrix_kernel(T* __restrict__ dst, int64_t ldd, const T* __restrict__ src, int64_t lds,
                                   int64_t rows, int64_t cols, int lower_only) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
    if (lower_only && c > r) continue;
    dst[r * ldd + c] = src[r * lds + c];
  }
}

template <typename T>
__global__ void transpose_kernel(T* __restrict__ dst, int64_t ldd, const T* __restrict__ src, int64_t lds,
                                 int64_t rows, int64_t cols) {
  __shared__ T tile[32][33];
  const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int64_t r = r0 + i, c = c0 + threadIdx.x;
    tile[i][threadIdx.x] = (r < rows && c < cols) ? src[r * lds + c] : T(0);
  }
  __syncthreads();
  for (int i = threadIdx.y; i < 32; i += 8) {
    const int64_t c = c0 + i, r = r0 + threadIdx.x;   // dst row = src col
    if (c < cols && r < rows) dst[c * ldd + r] = tile[threadIdx.x][i];
  }
}

// L' = J L^T J (J = exchange matrix): dst[i][j] = src[n-1-j][n-1-i], j <= i.  L' is lower triangular and
// L^T X = B  <=>  L' (J X) = J B, which turns the backward substitution into the forward one.
template <typename T>
__global__ void flip_transpose_lower_kernel(T* __restrict__ dst, int64_t ldd, const T* __restrict__ src, int64_t lds,
                                            int64_t n) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  for (int64_t i = blockIdx.y; i < n; i += gridDim.y)
    if (j <= i) dst[i * ldd + j] = src[(n - 1 - j) * lds + (n - 1 - i)];
}

// dst[c][r'] = src[r][c] with r' = flip_dst ? rows-1-r ... (plain, flag-controlled; used only by smn_trsm trans=1)
template <typename T>
__global__ void transpose_flip_kernel(T* __restrict__ dst, int64_t ldd, const T* __restrict__ src, int64_t lds,
                                      int64_t rows, int64_t cols, int flip_src_rows, int flip_dst_rows) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  for (int64_t r = blockIdx.y; r < rows; r += gridDim.y) {
    const int64_t sr = flip_src_rows ? rows - 1 - r : r;
    const int64_t dr = flip_dst_rows ? cols - 1 - c : c;
    dst[dr * ldd + r] = src[sr * lds + c];
  }
}

template <typename T>
__global__ void identity_pad_kernel(T* __restrict__ a, int64_t lda, int64_t n_pad, int64_t n_valid) {
  const int64_t i = n_valid + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_pad) a[i * lda + i] = T(1);
}

template <typename T>
__global__ void set_aug_rows_kernel(T* __restrict__ a, int64_t lda, int64_t row0, int64_t ncols,
                                    const T* __restrict__ y, int64_t n, int64_t c, int64_t ldy) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t k = blockIdx.y;
  if (i >= ncols || k >= c) return;
  a[(row0 + k) * lda + i] = i < n ? y[i * ldy + k] : T(0);
}

// The three little launches in front of a head's factorisation as one (a reference-sized SPR.loss is launch-bound):
// the appended right-hand-side rows, the absolute diagonal shift (same arithmetic as diag_shift_kernel) and the reset
// of the logdet / info scalars.
template <typename T>
__global__ void aug_prep_kernel(T* __restrict__ a, int64_t lda, int64_t row0, int64_t col0, int64_t ncols, const T* __restrict__ y,
                                int64_t n, int64_t c, int64_t ldy, int64_t n_shift, double jitter_abs,
                                double* __restrict__ logdet, int* __restrict__ info, double ridge_rel, int64_t n_trace) {
  const int64_t i = col0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // columns [col0, ncols) (a split build preps its corner later)
  const int64_t k = blockIdx.y;
  if (i >= ncols || k >= c) return;
  a[(row0 + k) * lda + i] = i < n ? y[i * ldy + k] : T(0);
  if (k == 0) {
    if (i < n_shift) {   // (the arithmetic of diag_shift_kernel; the trace sits next to logdet)
      const double sh = jitter_abs + (ridge_rel != 0.0 ? ridge_rel * logdet[1] / (double)n_trace : 0.0);
      a[i * lda + i] = (T)((double)a[i * lda + i] + sh);
    }
    if (i == 0) {
      *logdet = 0.0;
      *info = INT_MAX;
    }
  }
}

// mean[ti, k] = -a[aug0 + t + k, aug0 + ti];  cov[ti, tj] = a[aug0 + max, aug0 + min];
// quad[k] = -a[aug0 + t + k, aug0 + t + k];  with a mailbox (pinned host memory) the thread that owns quad[0] also
// publishes logdet and info there (what publish_kernel does as a launch of its own).
template <typename T>
__global__ void extract_posterior_kernel(const T* __restrict__ a, int64_t lda, int64_t aug0, int64_t t, int64_t c,
                                         T* __restrict__ mean, T* __restrict__ cov, int64_t ldcov,
                                         double* __restrict__ quad, double* __restrict__ mail,
                                         const double* __restrict__ scal, const int* __restrict__ info) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (int64_t i = blockIdx.y; i <= t; i += gridDim.y)
  if (i < t) {
    if (j < t && cov) {
      const int64_t hi = i > j ? i : j, lo = i > j ? j : i;
      cov[i * ldcov + j] = a[(aug0 + hi) * lda + aug0 + lo];
    }
    if (j < c && mean) mean[i * c + j] = -a[(aug0 + t + j) * lda + aug0 + i];
  } else if (i == t) {
    if (j < c && quad) {
      const double q = -(double)a[(aug0 + t + j) * lda + aug0 + t + j];
      quad[j] = q;
      if (mail) mail[2 + j] = q;
    }
    if (mail && j == 0) {
      mail[0] = scal[0];
      mail[1] = (double)info[0];
    }
  }
}

}  // namespace

#define DISPATCH_T(dtype, expr_f32, expr_f64) \
  do {                                        \
    if ((dtype) == SMN_F64) { expr_f64; } else { expr_f32; } \
  } while (0)

int fill_identity_pad(smn_ctx* ctx, int dtype, void* a, int64_t lda, int64_t n_pad, int64_t n_valid) {
  if (n_pad <= n_valid) return SMN_OK;
  const unsigned g = (unsigned)((n_pad - n_valid + 255) / 256);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(identity_pad_kernel<float>, dim3(g), dim3(256), 0, ctx->stream, static_cast<float*>(a), lda, n_pad, n_valid),
             hipLaunchKernelGGL(identity_pad_kernel<double>, dim3(g), dim3(256), 0, ctx->stream, static_cast<double*>(a), lda, n_pad, n_valid));
  SMN_CHECK_LAUNCH(ctx);
  return SMN_OK;
}

int copy_matrix(smn_ctx* ctx, int dtype, void* dst, int64_t ldd, const void* src, int64_t lds, int64_t rows,
                int64_t cols, int lower_only) {
  if (rows <= 0 || cols <= 0) return SMN_OK;
  dim3 g((unsigned)((cols + 255) / 256), (unsigned)(rows < 32768 ? rows : 32768));
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(copy_matrix_kernel<float>, g, dim3(256), 0, ctx->stream, static_cast<float*>(dst), ldd, static_cast<const float*>(src), lds, rows, cols, lower_only),
             hipLaunchKernelGGL(copy_matrix_kernel<double>, g, dim3(256), 0, ctx->stream, static_cast<double*>(dst), ldd, static_cast<const double*>(src), lds, rows, cols, lower_only));
  SMN_CHECK_LAUNCH(ctx);
  return SMN_OK;
}

int transpose_matrix(smn_ctx* ctx, int dtype, void* dst, int64_t ldd, const void* src, int64_t lds, int64_t rows,
                     int64_t cols) {
  if (rows <= 0 || cols <= 0) return SMN_OK;
  dim3 g((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32)), b(32, 8);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(transpose_kernel<float>, g, b, 0, ctx->stream, static_cast<float*>(dst), ldd, static_cast<const float*>(src), lds, rows, cols),
             hipLaunchKernelGGL(transpose_kernel<double>, g, b, 0, ctx->stream, static_cast<double*>(dst), ldd, static_cast<const double*>(src), lds, rows, cols));
  SMN_CHECK_LAUNCH(ctx);
  return SMN_OK;
}

int flip_transpose_lower(smn_ctx* ctx, int dtype, void* dst, int64_t ldd, const void* src, int64_t lds, int64_t n) {
  dim3 g((unsigned)((n + 255) / 256), (unsigned)(n < 32768 ? n : 32768));
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(flip_transpose_lower_kernel<float>, g, dim3(256), 0, ctx->stream, static_cast<float*>(dst), ldd, static_cast<const float*>(src), lds, n),
             hipLaunchKernelGGL(flip_transpose_lower_kernel<double>, g, dim3(256), 0, ctx->stream, static_cast<double*>(dst), ldd, static_cast<const double*>(src), lds, n));
  SMN_CHECK_LAUNCH(ctx);
  return SMN_OK;
}

int transpose_flip(smn_ctx* ctx, int dtype, void* dst, int64_t ldd, const void* src, int64_t lds, int64_t rows,
                   int64_t cols, int flip_src_rows, int flip_dst_rows) {
  if (rows <= 0 || cols <= 0) return SMN_OK;
  dim3 g((unsigned)((cols + 255) / 256), (unsigned)(rows < 32768 ? rows : 32768));
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(transpose_flip_kernel<float>, g, dim3(256), 0, ctx->stream, static_cast<float*>(dst), ldd, static_cast<const float*>(src), lds, rows, cols, flip_src_rows, flip_dst_rows),
             hipLaunchKernelGGL(transpose_flip_kernel<double>, g, dim3(256), 0, ctx->stream, static_cast<double*>(dst), ldd, static_cast<const double*>(src), lds, rows, cols, flip_src_rows, flip_dst_rows));
  SMN_CHECK_LAUNCH(ctx);
  return SMN_OK;
}

int set_aug_rows(smn_ctx* ctx, int dtype, void* a, int64_t lda, int64_t row0, int64_t ncols, const void* y, int64_t n,
                 int64_t c, int64_t ldy) {
  if (c <= 0) return SMN_OK;
  dim3 g((unsigned)((ncols + 255) / 256), (unsigned)c);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(set_aug_rows_kernel<float>, g, dim3(256), 0, ctx->stream, static_cast<float*>(a), lda, row0, ncols, static_cast<const float*>(y), n, c, ldy),
             hipLaunchKernelGGL(set_aug_rows_kernel<double>, g, dim3(256), 0, ctx->stream, static_cast<double*>(a), lda, row0, ncols, static_cast<const double*>(y), n, c, ldy));
  SMN_CHECK_LAUNCH(ctx);
  return SMN_OK;
}

int extract_posterior(smn_ctx* ctx, int dtype, const void* a, int64_t lda, int64_t aug0, int64_t t, int64_t c,
                      void* mean, void* cov, int64_t ldcov, double* quad_dev, bool publish) {
  const int64_t w = (t > c ? t : c) > 1 ? (t > c ? t : c) : 1;
  dim3 g((unsigned)((w + 255) / 256), (unsigned)(t + 1 < 32768 ? t + 1 : 32768));
  double* mail = publish ? ctx->d_mail : nullptr;
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(extract_posterior_kernel<float>, g, dim3(256), 0, ctx->stream, static_cast<const float*>(a), lda, aug0, t, c, static_cast<float*>(mean), static_cast<float*>(cov), ldcov, quad_dev, mail, ctx->d_scal, ctx->d_info),
             hipLaunchKernelGGL(extract_posterior_kernel<double>, g, dim3(256), 0, ctx->stream, static_cast<const double*>(a), lda, aug0, t, c, static_cast<double*>(mean), static_cast<double*>(cov), ldcov, quad_dev, mail, ctx->d_scal, ctx->d_info));
  SMN_CHECK_LAUNCH(ctx);
  return SMN_OK;
}

int aug_prep(smn_ctx* ctx, int dtype, void* a, int64_t lda, int64_t row0, int64_t ncols, const void* y, int64_t n,
             int64_t c, int64_t ldy, int64_t n_shift, double jitter_abs, int64_t col0, hipStream_t st, double ridge_rel, int64_t n_trace) {
  if (c <= 0 || n_shift > ncols || col0 < 0 || col0 >= ncols) return smn_fail(ctx, SMN_EINVAL, "aug_prep: bad sizes");
  if (!st) st = ctx->stream;
  dim3 g((unsigned)((ncols - col0 + 255) / 256), (unsigned)c);
  DISPATCH_T(dtype,
             hipLaunchKernelGGL(aug_prep_kernel<float>, g, dim3(256), 0, st, static_cast<float*>(a), lda, row0, col0, ncols, static_cast<const float*>(y), n, c, ldy, n_shift, jitter_abs, ctx->d_scal, ctx->d_info, ridge_rel, n_trace),
             hipLaunchKernelGGL(aug_prep_kernel<double>, g, dim3(256), 0, st, static_cast<double*>(a), lda, row0, col0, ncols, static_cast<const double*>(y), n, c, ldy, n_shift, jitter_abs, ctx->d_scal, ctx->d_info, ridge_rel, n_trace));
  SMN_CHECK_LAUNCH(ctx);
  return SMN_OK;
}

// ------------------------------------------------------------------ public API
extern "C" int smn_version(void) { return 100; }

extern "C" int smn_device_count(int* n) {
  if (!n) return SMN_EINVAL;
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) c = 0;
  *n = c;
  return SMN_OK;
}

extern "C" int smn_ctx_create(int device_id, smn_ctx** out) {
  if (!out) return SMN_EINVAL;
  *out = nullptr;
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0 || device_id < 0 || device_id >= cnt) return SMN_EHIP;
  if (hipSetDevice(device_id) != hipSuccess) return SMN_EHIP;
  smn_ctx* c = new smn_ctx();
  c->device = device_id;
  // Run-time knobs (six; everything else the earlier rounds swept is a constant now: profiles/r02_knob_sweep_final.txt):
  //   SMN_XCD_MAP, SMN_SUPER, SMN_SUPER_WIDE_ROWS, SMN_CHAIN_CUS, SMN_CHAIN_MIN_N, SMN_PANEL_LEAF
  if (const char* e = getenv("SMN_XCD_MAP")) c->xcd_map = e[0] == '1';
  if (const char* e = getenv("SMN_SUPER")) c->super_panel = atol(e);
  if (const char* e = getenv("SMN_SUPER_WIDE_ROWS")) c->super_wide_rows = atol(e);
  if (const char* e = getenv("SMN_PANEL_LEAF")) c->panel_leaf = e[0] != '0';
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0)
      c->num_cu = prop.multiProcessorCount;
  }
  int prio_lo = 0, prio_hi = 0;   // the main stream carries the critical path (the panel chain): highest priority
  (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
  // SMN_CHAIN_CUS=r gives the Cholesky a "bulk" stream that may not use the first r bits of the CU mask, so the panel
  // chain on the main stream always finds r CUs the block updates cannot occupy (CU masks do restrict kernels on this
  // stack: profiles/r01e_lookahead_ab_reserved_cus.txt).
  auto masked_stream = [&](hipStream_t* st, int lo, int hi) -> bool {
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = lo; b < hi && b < 256; ++b) mask[b >> 5] |= 1u << (b & 31);
    return hipExtStreamCreateWithCUMask(st, 8, mask) == hipSuccess;
  };
  const bool main_ok = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_hi) == hipSuccess;
  if (const char* e = getenv("SMN_CHAIN_CUS")) c->chain_cus = atoi(e);
  if (const char* e = getenv("SMN_CHAIN_MIN_N")) c->chain_min_n = atol(e);
  if (c->chain_cus > 0 && c->chain_cus < c->num_cu) {
    if (!masked_stream(&c->stream_bulk, c->chain_cus, c->num_cu)) c->stream_bulk = nullptr;   // no look-ahead then
  }
  bool ok = main_ok &&
            hipEventCreateWithFlags(&c->ev_a, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&c->ev_b, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&c->ev_c0, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&c->ev_c1, hipEventDisableTiming) == hipSuccess &&
            hipStreamCreateWithPriority(&c->stream_comm, hipStreamNonBlocking, prio_hi) == hipSuccess &&
            hipStreamCreateWithPriority(&c->stream_scatter, hipStreamNonBlocking, prio_hi) == hipSuccess &&
            hipEventCreate(&c->ev_t0) == hipSuccess && hipEventCreate(&c->ev_t1) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&c->d_scal), 64 * sizeof(double)) == hipSuccess &&
            hipMalloc(reinterpret_cast<void**>(&c->d_info), 16 * sizeof(int)) == hipSuccess &&
            hipHostMalloc(reinterpret_cast<void**>(&c->h_mail), 64 * sizeof(double), hipHostMallocMapped) == hipSuccess &&
            hipHostGetDevicePointer(reinterpret_cast<void**>(&c->d_mail), c->h_mail, 0) == hipSuccess;
  if (!ok) {
    smn_ctx_destroy(c);
    return SMN_EHIP;
  }
  *out = c;
  return SMN_OK;
}

extern "C" int smn_ctx_destroy(smn_ctx* c) {
  if (!c) return SMN_OK;
  (void)hipSetDevice(c->device);
  if (c->stream_bulk) (void)hipStreamSynchronize(c->stream_bulk);
  if (c->stream_comm) (void)hipStreamSynchronize(c->stream_comm);
  if (c->stream_scatter) (void)hipStreamSynchronize(c->stream_scatter);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) smn_comm_destroy(c);
  for (int i = 0; i < smn_ctx::kSlots; ++i)
    if (c->ws[i]) (void)hipFree(c->ws[i]);
  for (hipEvent_t e : c->prof_ev) (void)hipEventDestroy(e);
  if (c->d_scal) (void)hipFree(c->d_scal);
  if (c->d_info) (void)hipFree(c->d_info);
  if (c->h_mail) (void)hipHostFree(c->h_mail);
  if (c->ev_a) (void)hipEventDestroy(c->ev_a);
  if (c->ev_b) (void)hipEventDestroy(c->ev_b);
  if (c->ev_s0) (void)hipEventDestroy(c->ev_s0);
  if (c->ev_corner) (void)hipEventDestroy(c->ev_corner);
  if (c->tile_list) (void)hipFree(c->tile_list);
  if (c->ev_c0) (void)hipEventDestroy(c->ev_c0);
  if (c->ev_c1) (void)hipEventDestroy(c->ev_c1);
  if (c->stream_comm) (void)hipStreamDestroy(c->stream_comm);
  if (c->stream_scatter) (void)hipStreamDestroy(c->stream_scatter);
  for (hipEvent_t e : c->ev_pool) (void)hipEventDestroy(e);
  if (c->ev_t0) (void)hipEventDestroy(c->ev_t0);
  if (c->ev_t1) (void)hipEventDestroy(c->ev_t1);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->stream_bulk) (void)hipStreamDestroy(c->stream_bulk);
  delete c;
  return SMN_OK;
}

extern "C" int smn_last_error(smn_ctx* ctx, char* buf, size_t n) {
  if (!ctx || !buf || n == 0) return SMN_EINVAL;
  snprintf(buf, n, "%s", ctx->err.c_str());
  return SMN_OK;
}

extern "C" int smn_synchronize(smn_ctx* ctx) {
  if (!ctx) return SMN_EINVAL;
  SMN_ENTER(ctx);
  SMN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SMN_OK;
}

extern "C" int smn_malloc(smn_ctx* ctx, size_t bytes, void** dptr) {
  if (!ctx || !dptr) return SMN_EINVAL;
  SMN_ENTER(ctx);
  *dptr = nullptr;
  if (bytes == 0) bytes = 16;
  SMN_HIP(ctx, hipMalloc(dptr, bytes));
  return SMN_OK;
}

extern "C" int smn_free(smn_ctx* ctx, void* dptr) {
  if (!ctx) return SMN_EINVAL;
  SMN_ENTER(ctx);
  if (!dptr) return SMN_OK;
  SMN_TRY(sync_side_streams(ctx));   // a piece of it may still be in flight (exchange, scatter, a piece's build)
  SMN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  SMN_HIP(ctx, hipFree(dptr));
  return SMN_OK;
}

extern "C" int smn_memset(smn_ctx* ctx, void* dptr, int value, size_t bytes) {
  if (!ctx) return SMN_EINVAL;
  SMN_ENTER(ctx);
  SMN_HIP(ctx, hipMemsetAsync(dptr, value, bytes, ctx->stream));
  return SMN_OK;
}

extern "C" int smn_memcpy_h2d(smn_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx) return SMN_EINVAL;
  SMN_ENTER(ctx);
  SMN_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  SMN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SMN_OK;
}

extern "C" int smn_memcpy_d2h(smn_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx) return SMN_EINVAL;
  SMN_ENTER(ctx);
  SMN_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  SMN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SMN_OK;
}

extern "C" int smn_memcpy_d2d(smn_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx) return SMN_EINVAL;
  SMN_ENTER(ctx);
  SMN_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return SMN_OK;
}

extern "C" int smn_memcpy2d_h2d(smn_ctx* ctx, void* dst, size_t dpitch, const void* src, size_t spitch,
                                size_t width_bytes, size_t rows) {
  if (!ctx) return SMN_EINVAL;
  SMN_ENTER(ctx);
  SMN_HIP(ctx, hipMemcpy2DAsync(dst, dpitch, src, spitch, width_bytes, rows, hipMemcpyHostToDevice, ctx->stream));
  SMN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SMN_OK;
}

extern "C" int smn_memcpy2d_d2h(smn_ctx* ctx, void* dst, size_t dpitch, const void* src, size_t spitch,
                                size_t width_bytes, size_t rows) {
  if (!ctx) return SMN_EINVAL;
  SMN_ENTER(ctx);
  SMN_HIP(ctx, hipMemcpy2DAsync(dst, dpitch, src, spitch, width_bytes, rows, hipMemcpyDeviceToHost, ctx->stream));
  SMN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SMN_OK;
}

extern "C" int smn_timer_start(smn_ctx* ctx) {
  if (!ctx) return SMN_EINVAL;
  SMN_ENTER(ctx);
  SMN_HIP(ctx, hipEventRecord(ctx->ev_t0, ctx->stream));
  return SMN_OK;
}

extern "C" int smn_timer_stop_ms(smn_ctx* ctx, double* ms) {
  if (!ctx || !ms) return SMN_EINVAL;
  SMN_ENTER(ctx);
  SMN_HIP(ctx, hipEventRecord(ctx->ev_t1, ctx->stream));
  SMN_HIP(ctx, hipEventSynchronize(ctx->ev_t1));
  float f = 0.f;
  SMN_HIP(ctx, hipEventElapsedTime(&f, ctx->ev_t0, ctx->ev_t1));
  *ms = (double)f;
  return SMN_OK;
}

// ---- per-kernel timing hooks (bench.py's roofline numbers come from these hipEvents) ----
extern "C" int smn_profile_enable(smn_ctx* ctx, int on) {
  if (!ctx) return SMN_EINVAL;
  SMN_ENTER(ctx);
  SMN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->prof = on != 0;
  ctx->prof_mask = on == 1 ? ~0u : ((unsigned)on >> 1);   // 1: every category; 2 << c: category c only (masks add)
  ctx->prof_used = 0;
  ctx->prof_cat.clear();
  for (double& f : ctx->prof_flops) f = 0.0;
  return SMN_OK;
}

// MFMA flops the launches of a category have EXECUTED since the last smn_profile_enable (whole tiles, counted on the
// host as they are issued; categories 4 = strip update, 5 = trailing update): bench.py prices the dominant
// kernel with this instead of re-deriving the schedule.
extern "C" int smn_profile_flops(smn_ctx* ctx, int category, double* flops) {
  if (!ctx || !flops || category < 0 || category >= PROF_NCAT) return SMN_EINVAL;
  *flops = ctx->prof_flops[category];
  return SMN_OK;
}

extern "C" int smn_profile_read(smn_ctx* ctx, int category, double* total_ms, int* launches) {
  if (!ctx || category < 0 || category >= PROF_NCAT) return SMN_EINVAL;
  SMN_ENTER(ctx);
  SMN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  double tot = 0.0;
  int cnt = 0;
  for (size_t i = 0; i < ctx->prof_cat.size() && 2 * i + 1 < ctx->prof_used + 1; ++i) {
    if (ctx->prof_cat[i] != category) continue;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ctx->prof_ev[2 * i], ctx->prof_ev[2 * i + 1]) == hipSuccess) {
      tot += ms;
      ++cnt;
    }
  }
  if (total_ms) *total_ms = tot;
  if (launches) *launches = cnt;
  return SMN_OK;
}
