// comm.hip — the one collective of the path: an RCCL all-gather of kernel row blocks over xGMI
// (SURVEY.md section 8e).  librccl is loaded lazily with dlopen so that single-GPU users of
// libsmnngp.so never pay for it; the reference has no collective to mirror.
#include <dlfcn.h>

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
  const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  for (const char* n : names) {
    r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
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
  Rccl& r = rccl();
  if (!r.ok) return smn_fail(ctx, SMN_ECOMM, "librccl.so could not be loaded");
  if (ctx->comm) return smn_fail(ctx, SMN_EINVAL, "communicator already initialised");
  SMN_HIP(ctx, hipSetDevice(ctx->device));
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
  if (ctx->comm) {
    Rccl& r = rccl();
    if (r.ok) r.CommDestroy(static_cast<nccl_comm>(ctx->comm));
    ctx->comm = nullptr;
    ctx->nranks = 1;
    ctx->rank = 0;
  }
  return SMN_OK;
}

extern "C" int smn_allgather(smn_ctx* ctx, int dtype, const void* send_d, void* recv_d, int64_t count) {
  if (!ctx || !send_d || !recv_d || count <= 0) return SMN_EINVAL;
  if (dtype != SMN_F32 && dtype != SMN_F64) return smn_fail(ctx, SMN_EINVAL, "bad dtype");
  if (!ctx->comm) {  // single rank: the gather is a copy
    if (send_d != recv_d)
      SMN_HIP(ctx, hipMemcpyAsync(recv_d, send_d, dtype_size(dtype) * (size_t)count, hipMemcpyDeviceToDevice, ctx->stream));
    return SMN_OK;
  }
  Rccl& r = rccl();
  const int rc = r.AllGather(send_d, recv_d, (size_t)count, dtype == SMN_F64 ? kNcclFloat64 : kNcclFloat32,
                             static_cast<nccl_comm>(ctx->comm), ctx->stream);
  if (rc != 0) return smn_fail(ctx, SMN_ECOMM, "ncclAllGather: %s", r.GetErrorString ? r.GetErrorString(rc) : "?");
  return SMN_OK;
}
