// comm.hip — the one collective of the path: an RCCL all-gather of kernel row blocks over xGMI
// (SURVEY.md section 8e).  librccl is loaded lazily with dlopen so that single-GPU users of
// libsmnngp.so never pay for it; the reference has no collective to mirror.
#include <dlfcn.h>

#include <string>
#include <vector>

#include "internal.hpp"

namespace {

// Minimal slice of the RCCL ABI (rccl.h): opaque comm, 128-byte unique id, result code 0 = success.
typedef struct { char internal[128]; } nccl_uid;
typedef void* nccl_comm;
enum { kNcclFloat32 = 7, kNcclFloat64 = 8 };

struct Rccl {
  void* h = nullptr;
  int (*GetUniqueId)(nccl_uid*) = nullptr;
  int (*CommInitRank)(nccl_comm*, int, nccl_uid, int) = nullptr;
  int (*CommDestroy)(nccl_comm) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, nccl_comm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};

Rccl& rccl() {
  static Rccl r;
  if (r.h) return r;
  // RCCL must sit on the SAME HIP runtime as this library (it is handed our streams and device pointers).  A
  // process that also holds PyTorch can contain two ROCm stacks -- /opt/rocm's and the one inside the wheel
  // (torch/lib/{libamdhip64,librccl}.so) -- and which libamdhip64 this library got bound to depends on the import
  // order.  So: find the HIP runtime we are bound to and take the librccl that lives beside it (already loaded: that
  // very object; otherwise a private, deep-bound copy, so its internal calls cannot land in another RCCL's global
  // symbols -- that mix made ncclCommInitRank fail with "unhandled cuda error").
  std::string dir;
  Dl_info info;
  if (dladdr(reinterpret_cast<void*>(&hipGetDeviceCount), &info) && info.dli_fname) {
    dir = info.dli_fname;
    const size_t slash = dir.rfind('/');
    dir = slash == std::string::npos ? std::string() : dir.substr(0, slash + 1);
  }
  std::vector<std::string> names;
  if (!dir.empty()) {
    names.push_back(dir + "librccl.so.1");
    names.push_back(dir + "librccl.so");
  }
  names.push_back("librccl.so.1");
  names.push_back("librccl.so");
  names.push_back("/opt/rocm/lib/librccl.so");
  for (const std::string& n : names) {
    r.h = dlopen(n.c_str(), RTLD_NOW | RTLD_NOLOAD);
    if (!r.h) r.h = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL | RTLD_DEEPBIND);
    if (r.h) break;
  }
  if (!r.h) return r;
  r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.h, "ncclGetUniqueId"));
  r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.h, "ncclCommInitRank"));
  r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.h, "ncclCommDestroy"));
  r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(r.h, "ncclAllGather"));
  r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.h, "ncclGetErrorString"));
  r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather;
  return r;
}

}  // namespace

extern "C" int smn_comm_unique_id(char id_out[128]) {
  if (!id_out) return SMN_EINVAL;
  Rccl& r = rccl();
  if (!r.ok) return SMN_ECOMM;
  nccl_uid u;
  if (r.GetUniqueId(&u) != 0) return SMN_ECOMM;
  memcpy(id_out, u.internal, 128);
  return SMN_OK;
}

extern "C" int smn_comm_init(smn_ctx* ctx, int nranks, int rank, const char id[128]) {
  if (!ctx || !id || nranks <= 0 || rank < 0 || rank >= nranks) return SMN_EINVAL;
  SMN_ENTER(ctx);
  Rccl& r = rccl();
  if (!r.ok) return smn_fail(ctx, SMN_ECOMM, "librccl.so could not be loaded");
  if (ctx->comm) return smn_fail(ctx, SMN_EINVAL, "communicator already initialised");
  nccl_uid u;
  memcpy(u.internal, id, 128);
  nccl_comm c = nullptr;
  const int rc = r.CommInitRank(&c, nranks, u, rank);
  if (rc != 0) return smn_fail(ctx, SMN_ECOMM, "ncclCommInitRank: %s", r.GetErrorString ? r.GetErrorString(rc) : "?");
  ctx->comm = c;
  ctx->nranks = nranks;
  ctx->rank = rank;
  return SMN_OK;
}

extern "C" int smn_comm_destroy(smn_ctx* ctx) {
  if (!ctx) return SMN_EINVAL;
  SMN_ENTER(ctx);
  if (ctx->comm) {
    Rccl& r = rccl();
    if (r.ok) r.CommDestroy(static_cast<nccl_comm>(ctx->comm));
    ctx->comm = nullptr;
    ctx->nranks = 1;
    ctx->rank = 0;
  }
  return SMN_OK;
}

extern "C" int smn_comm_info(smn_ctx* ctx, int* nranks, int* rank) {
  if (!ctx) return SMN_EINVAL;
  if (nranks) *nranks = ctx->comm ? ctx->nranks : 1;
  if (rank) *rank = ctx->comm ? ctx->rank : 0;
  return SMN_OK;
}

namespace {
// all-gather of `count` elements per rank on `st`; without a communicator (one rank) it is a copy
int allgather_on(smn_ctx* ctx, hipStream_t st, int dtype, const void* send_d, void* recv_d, int64_t count) {
  if (!ctx->comm) {
    if (send_d != recv_d)
      SMN_HIP(ctx, hipMemcpyAsync(recv_d, send_d, dtype_size(dtype) * (size_t)count, hipMemcpyDeviceToDevice, st));
    return SMN_OK;
  }
  Rccl& r = rccl();
  const int rc = r.AllGather(send_d, recv_d, (size_t)count, dtype == SMN_F64 ? kNcclFloat64 : kNcclFloat32,
                             static_cast<nccl_comm>(ctx->comm), st);
  if (rc != 0) return smn_fail(ctx, SMN_ECOMM, "ncclAllGather: %s", r.GetErrorString ? r.GetErrorString(rc) : "?");
  return SMN_OK;
}
}  // namespace

extern "C" int smn_allgather(smn_ctx* ctx, int nranks, int dtype, const void* send_d, void* recv_d, int64_t count) {
  if (!ctx || !send_d || !recv_d || count <= 0) return SMN_EINVAL;
  SMN_ENTER(ctx);
  if (dtype != SMN_F32 && dtype != SMN_F64) return smn_fail(ctx, SMN_EINVAL, "bad dtype");
  const int P = ctx->comm ? ctx->nranks : 1;
  if (nranks != P)
    return smn_fail(ctx, SMN_ECOMM, "smn_allgather: %d ranks asked for, the context's communicator has %d", nranks, P);
  ProfScope ps(ctx, PROF_COMM, ctx->stream);
  return allgather_on(ctx, ctx->stream, dtype, send_d, recv_d, count);
}

// ---------------------------------------------------------------- paired lower-trapezoid blocks
// The balanced symmetric shard (sharding.py: paired layout): the n rows are cut into 2P blocks of
// `h` rows; rank r builds blocks r and 2P-1-r, each only up to its own last column, packed densely
// (block b: h rows of leading dimension (b+1)h).  Every rank then holds exactly h*h*(2P+1)
// elements, so ONE equal-count all-gather moves the lower trapezoids -- about half the bytes of
// full rows.  This kernel scatters the gathered chunks into the natural row order of K; entries
// right of the diagonal's 128-column tile are not written.
namespace {

// VEC elements per lane (VEC = 4 needs 16-byte aligned rows on both sides; the host checks).  blockIdx.y strides
// over rows, blockIdx.x over 256*VEC-column spans; spans right of the row's diagonal tile leave at once.
template <typename T, int VEC>
__global__ void unpack_blocks_kernel(const T* __restrict__ stage, int64_t chunk, int64_t h, int P, int64_t n,
                                     T* __restrict__ k, int64_t ldk) {
  const int64_t c = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC;
  for (int64_t r = blockIdx.y; r < n; r += gridDim.y) {
    int64_t cend = (r / kTile + 1) * kTile;
    if (cend > n) cend = n;
    if (c >= cend) continue;
    const int64_t b = r / h;
    const int64_t owner = b < P ? b : 2 * (int64_t)P - 1 - b;
    const int64_t off = owner * chunk + (b < P ? 0 : h * (owner + 1) * h);
    const T* src = stage + off + (r - b * h) * ((b + 1) * h) + c;
    T* dst = k + r * ldk + c;
    if (VEC > 1 && c + VEC <= cend) {
      typedef T vec_t __attribute__((ext_vector_type(VEC)));
      *reinterpret_cast<vec_t*>(dst) = *reinterpret_cast<const vec_t*>(src);
    } else {
      for (int v = 0; v < VEC && c + v < cend; ++v) dst[v] = src[v];
    }
  }
}

}  // namespace

extern "C" int smn_unpack_lower_blocks(smn_ctx* ctx, int dtype, const void* stage_d, int64_t n, int nranks,
                                       int64_t block_rows, void* k_d, int64_t ldk) {
  if (!ctx || !stage_d || !k_d) return SMN_EINVAL;
  SMN_ENTER(ctx);
  if (dtype != SMN_F32 && dtype != SMN_F64) return smn_fail(ctx, SMN_EINVAL, "bad dtype");
  if (n <= 0 || nranks <= 0 || block_rows <= 0 || block_rows % kTile || ldk < n ||
      2 * (int64_t)nranks * block_rows < n)
    return smn_fail(ctx, SMN_EINVAL, "smn_unpack_lower_blocks: bad geometry (n=%lld ranks=%d block_rows=%lld ldk=%lld)",
                    (long long)n, nranks, (long long)block_rows, (long long)ldk);
  const int64_t chunk = block_rows * block_rows * (2 * (int64_t)nranks + 1);
  const size_t es = dtype_size(dtype);
  const int vec = (int)(16 / es);
  const bool aligned = (ldk % vec == 0) && (reinterpret_cast<uintptr_t>(k_d) % 16 == 0) &&
                       (reinterpret_cast<uintptr_t>(stage_d) % 16 == 0);
  const int v = aligned ? vec : 1;
  dim3 g((unsigned)((n + 256 * v - 1) / (256 * v)), (unsigned)(n < 32768 ? n : 32768));
  ProfScope ps(ctx, PROF_MISC, ctx->stream);
#define UNPACK(T, V)                                                                                              \
  hipLaunchKernelGGL((unpack_blocks_kernel<T, V>), g, dim3(256), 0, ctx->stream, static_cast<const T*>(stage_d), \
                     chunk, block_rows, nranks, n, static_cast<T*>(k_d), ldk)
  if (dtype == SMN_F64) {
    if (aligned) UNPACK(double, 2); else UNPACK(double, 1);
  } else {
    if (aligned) UNPACK(float, 4); else UNPACK(float, 1);
  }
#undef UNPACK
  SMN_CHECK_LAUNCH(ctx);
  return SMN_OK;
}

// ---------------------------------------------------------------- column-first exchange: pieces of the cyclic layout
// internal.hpp ColPieces: piece g of a rank's chunk is slots(g) strips of 128 x width(g) elements (its tile rows from tile
// column c[g] down, columns [c[g], c[g+1]) * 128); all ranks' pieces g are gathered into stage[P * off[g] ...][rank][count(g)]
// and scattered into the natural row order of K.  Strips are read and written in 16-byte vectors (a strip row and a piece
// are multiples of 128 elements); entries right of the row's diagonal tile, rows >= n and the slots of tile rows past the
// kernel are not written.
namespace {

template <typename T, int VEC>
__global__ void scatter_piece_kernel(const T* __restrict__ stage_g, int64_t count, int64_t width, int64_t c0, int P, int64_t T_rows,
                                     int64_t n, T* __restrict__ k, int64_t ldk, double diag_add) {
  const int64_t v = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * VEC;   // element inside [rank][count]
  if (v >= (int64_t)P * count) return;
  const int64_t r = v / count;                 // owner rank
  const int64_t e = v - r * count;             // element of its piece
  const int64_t srow = e / width, lc = e - srow * width;   // strip row (slot * 128 + row in tile), column inside the piece
  const int64_t j = c0 / P + srow / kTile;     // tile-row group
  const int64_t t = j * P + ((j & 1) ? P - 1 - r : r);
  if (t >= T_rows) return;
  const int64_t row = t * kTile + (srow % kTile), col = c0 * kTile + lc;
  if (row >= n) return;
  int64_t cend = (t + 1) * kTile;
  if (cend > n) cend = n;
  if (col >= cend) return;
  T* dst = k + row * ldk + col;
  const T* src = stage_g + v;
  if (VEC > 1 && col + VEC <= cend) {
    typedef T vec_t __attribute__((ext_vector_type(VEC)));
    vec_t x = *reinterpret_cast<const vec_t*>(src);
    if (diag_add != 0.0 && row >= col && row < col + VEC) x[row - col] = (T)((double)x[row - col] + diag_add);
    *reinterpret_cast<vec_t*>(dst) = x;
  } else {
    for (int i = 0; i < VEC && col + i < cend; ++i)
      dst[i] = (diag_add != 0.0 && row == col + i) ? (T)((double)src[i] + diag_add) : src[i];
  }
}

}  // namespace

int scatter_piece_on(smn_ctx* ctx, hipStream_t st, int dtype, const void* stage_d, int64_t n, const ColPieces& cp, int g,
                     void* k_d, int64_t ldk, double diag_add) {
  if (g < 0 || g >= cp.np || ldk < n) return smn_fail(ctx, SMN_EINVAL, "scatter: piece %d of %d, ldk %lld", g, cp.np, (long long)ldk);
  const size_t es = dtype_size(dtype);
  const int vec = (int)(16 / es);
  const char* sg = static_cast<const char*>(stage_d) + es * (size_t)cp.P * (size_t)cp.off[g];
  const bool aligned = (ldk % vec == 0) && (reinterpret_cast<uintptr_t>(k_d) % 16 == 0) && (reinterpret_cast<uintptr_t>(sg) % 16 == 0);
  const int64_t total = (int64_t)cp.P * cp.count(g);
  const int64_t tile_rows = cp.T;
#define SCATTERP(TY, V)                                                                                          \
  hipLaunchKernelGGL((scatter_piece_kernel<TY, V>), dim3((unsigned)((total / V + 255) / 256)), dim3(256), 0, st, \
                     reinterpret_cast<const TY*>(sg), cp.count(g), cp.width(g), cp.c[g], cp.P, tile_rows, n,    \
                     static_cast<TY*>(k_d), ldk, diag_add)
  if (dtype == SMN_F64) {
    if (aligned) SCATTERP(double, 2); else SCATTERP(double, 1);
  } else {
    if (aligned) SCATTERP(float, 4); else SCATTERP(float, 1);
  }
#undef SCATTERP
  SMN_CHECK_LAUNCH(ctx);
  return SMN_OK;
}

int allgather_piece_on(smn_ctx* ctx, hipStream_t st, int dtype, const void* mine_d, void* stage_d, const ColPieces& cp, int g) {
  if (g < 0 || g >= cp.np) return smn_fail(ctx, SMN_EINVAL, "all-gather: piece %d of %d", g, cp.np);
  const size_t es = dtype_size(dtype);
  const char* send = static_cast<const char*>(mine_d) + es * (size_t)cp.off[g];
  char* recv = static_cast<char*>(stage_d) + es * (size_t)cp.P * (size_t)cp.off[g];
  return allgather_on(ctx, st, dtype, send, recv, cp.count(g));
}
