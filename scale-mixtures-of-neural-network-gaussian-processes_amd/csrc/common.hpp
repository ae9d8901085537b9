// common.hpp — context, error plumbing and small helpers shared by every translation unit of
// libsmnngp.so.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/smnngp.h"

struct smn_ctx {
  int device = 0;
  hipStream_t stream = nullptr;       // main stream (high priority): every public call is ordered on it; carries the panel chain
  hipStream_t stream_bulk = nullptr;  // CU-masked stream of the far updates F1: may not use the first chain_cus CUs
  int chain_cus = 32;                 // CUs kept free for the panel chain (env SMN_CHAIN_CUS; 0 = no look-ahead)
  int64_t chain_min_n = 8192;         // look-ahead only from this matrix size on (env SMN_CHAIN_MIN_N)
  hipStream_t stream_comm = nullptr;     // column-first exchange: the all-gathers of the pieces, one after the other (high priority)
  hipStream_t stream_scatter = nullptr;  // the scatter of a gathered piece into the workspace: beside the NEXT piece's all-gather (link-bound), not in front of it
  hipEvent_t ev_a = nullptr, ev_b = nullptr, ev_t0 = nullptr, ev_t1 = nullptr;
  hipEvent_t ev_c0 = nullptr, ev_c1 = nullptr;   // main -> comm (the rank's share is built, the workspace is up); scatter -> main (every piece issued so far has landed)
  // Column-first exchange (smn_shard_begin ... smn_lml_from_shards): the factorisation workspace (slot 2) the pieces are
  // scattered into, the absolute jitter the scatter adds to the diagonal entries it writes, and one Arrival per piece issued:
  // columns [col_begin, col_end) of the workspace are complete once `ev` has fired.  The factorisation waits piece by piece
  // (cholesky.hip need_columns): the panel chain of a super-panel for its own columns only, the far update for the rest.
  void* shard_a = nullptr; int64_t shard_lda = 0, shard_n = 0; int shard_dtype = -1; double shard_eps = 0.0;
  struct Arrival { int64_t col_begin, col_end; hipEvent_t ev; };
  std::vector<Arrival> arrivals;          // in issue order; cleared by smn_shard_begin and by the factorisation that consumed them
  bool consume_arrivals = false;          // set by smn_lml_from_shards around its factorisation
  std::vector<hipEvent_t> ev_pool;        // events of the pieces (gathered / arrived), reused from pipeline to pipeline
  size_t ev_pool_used = 0;
  std::string err;
  // cached workspace arenas (grown on demand, freed with the context)
  static constexpr int kSlots = 10;
  void* ws[kSlots] = {};
  size_t ws_bytes[kSlots] = {};
  // small device scalar block: [0..15] doubles scratch, ints after
  double* d_scal = nullptr;   // 64 doubles
  int* d_info = nullptr;      // 16 ints
  // pinned, device-mapped host mailbox: one tiny kernel publishes logdet / info / quadratic forms into it and the
  // host reads them after ONE stream synchronisation (three pageable device-to-host copies cost ~0.1 ms per call)
  double* h_mail = nullptr;   // 64 doubles, host pointer
  double* d_mail = nullptr;   // the same memory as the device sees it
  void* comm = nullptr;       // ncclComm_t when smn_comm_init was called
  int nranks = 1, rank = 0;
  // per-kernel timing (smn_profile_*): hipEvent pairs around launches, resolved on read
  bool prof = false;
  unsigned prof_mask = ~0u;          // categories that get event pairs (smn_profile_enable)
  std::vector<hipEvent_t> prof_ev;   // pool, used pairwise
  std::vector<int> prof_cat;         // category of pair i
  size_t prof_used = 0;              // events handed out
  double prof_flops[12] = {};         // MFMA flops EXECUTED per category since smn_profile_enable (whole tiles; host-side count)
  int num_cu = 256;                  // hipDeviceProp_t::multiProcessorCount
  int64_t super_panel = 1024;        // columns per super-panel of the two-level Cholesky (env SMN_SUPER)
  int64_t super_wide_rows = 18432;   // super-panels are kSuperWide columns while at least this many rows are left (env SMN_SUPER_WIDE_ROWS;
                                     // measured: wide pays from ~18k rows left on, profiles/r02_wide_super_panel_sweep.txt; 0 rows = always wide)
  // structural-zero hint for the factorisation in flight: appended rows [id0, id1) hold an identity block
  // (row id0 + i is zero left of column i), set by cholesky_padded, -1 = none
  int64_t chol_id0 = -1, chol_id1 = -1;
  // batched factorisation (heads.hip spr_batch): `batch` problems of identical shape, problem g at a + g * batch_stride
  // elements, its logdet / info in batch_logdet[g] / batch_info[g]; every panel / update launch gets grid.y = batch
  size_t batch_bytes = (size_t)48 << 30;   // workspace budget of one batched pass (smn_debug_batch_bytes)
  int batch = 1; int64_t batch_stride = 0, batch_ldiag_stride = 0; double* batch_logdet = nullptr; int* batch_info = nullptr;
  // Split build of the headline path (heads.hip aug_build, smn_spr_loss with the look-ahead): the bottom-right corner of the
  // kernel matrix (tile rows and columns >= corner_tile) is built on the bulk stream BESIDE the first super-panel's panel
  // chain, which needs none of it; the factorisation takes it as one Arrival.  The two tile orders live in tile_list.
  int* tile_list = nullptr; int64_t tile_list_cap = 0; int tile_list_tm = 0, tile_list_tb = 0, tile_list_na = 0, tile_list_nb = 0;
  hipEvent_t ev_s0 = nullptr, ev_corner = nullptr;   // main -> bulk (the first launch is done); bulk -> main (the corner has landed)
  bool trace_ready = false;          // the build in front left the trace of the kernel's diagonal in d_scal[1] (BuildCall::want_trace)
  bool split_build = true;           // smn_debug_split_build
  int panel_max_passes = 4;          // row groups one panel workgroup may carry (smn_debug_panel_passes; 1: one group each, as before round 4)
  int64_t corner_col = 0;            // first column of the corner of the split build in flight (0: none)
  bool chol_noschur = false;         // the factorisation in flight leaves the appended rows' trailing block alone (cholesky.hip)
  bool chol_prepped = false;         // the caller has shifted the diagonal and reset logdet / info already (aug_prep)
  std::unordered_map<const void*, size_t> max_lds;   // largest dynamic-LDS size already allowed per kernel (smn_allow_lds)
  bool lds_attrs_done[2] = {false, false};   // hipFuncSetAttribute(MaxDynamicSharedMemorySize) issued for f32 / f64 kernels
  bool panel_leaf = true;            // panelr_kernel (register-resident 16x16 leaf) in the factorisation; env SMN_PANEL_LEAF=0: panel_kernel
  bool xcd_map = true;               // XCD-aware patch tile order of launches of 512 tiles and more (env SMN_XCD_MAP=0: linear order)
};

enum { PROF_PREP = 0, PROF_BUILD = 1, PROF_RECURSION = 2, PROF_PANEL = 3, PROF_STRIP = 4, PROF_TRAIL = 5,
       PROF_MISC = 6, PROF_COMM = 7, PROF_EXPOSED = 8, PROF_STALL = 9, PROF_NCAT = 10 };

// Brackets the launches issued during its lifetime with an event pair when profiling is on.
struct ProfScope {
  smn_ctx* c; hipStream_t st; bool on;
  ProfScope(smn_ctx* ctx, int cat, hipStream_t stream)
      : c(ctx), st(stream), on(ctx->prof && ((ctx->prof_mask >> cat) & 1u)) {
    if (!on) return;
    if (c->prof_used + 2 > c->prof_ev.size()) {
      for (int i = 0; i < 256; ++i) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) { on = false; return; }
        c->prof_ev.push_back(e);
      }
    }
    c->prof_cat.push_back(cat);
    (void)hipEventRecord(c->prof_ev[c->prof_used], st);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(c->prof_ev[c->prof_used + 1], st);
    c->prof_used += 2;
  }
};

inline int smn_fail(smn_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  return code;
}

#define SMN_HIP(ctx, call)                                                                   \
  do {                                                                                       \
    hipError_t e_ = (call);                                                                  \
    if (e_ != hipSuccess)                                                                    \
      return smn_fail((ctx), e_ == hipErrorOutOfMemory ? SMN_ENOMEM : SMN_EHIP, "%s: %s (%s:%d)", \
                      #call, hipGetErrorString(e_), __FILE__, __LINE__);                     \
  } while (0)

#define SMN_TRY(expr)         \
  do {                        \
    int rc_ = (expr);         \
    if (rc_ != SMN_OK) return rc_; \
  } while (0)

#define SMN_CHECK_LAUNCH(ctx) SMN_HIP(ctx, hipGetLastError())

// The current HIP device is a property of the calling host THREAD (default 0), not of the context: allocations,
// function attributes, events and plain launches all go to it.  Every C-ABI entry that takes a context therefore opens
// with SMN_ENTER(ctx): it makes ctx->device current for the duration of the call and puts the caller's device back on
// exit (worker threads of sweeps.py, two contexts on two devices in one thread, torch.cuda.set_device() after the
// context was made).
struct SmnDeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit SmnDeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
  }
  ~SmnDeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
  SmnDeviceGuard(const SmnDeviceGuard&) = delete;
  SmnDeviceGuard& operator=(const SmnDeviceGuard&) = delete;
};
#define SMN_ENTER(ctx) SmnDeviceGuard smn_device_guard_((ctx)->device)

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

// workspace slot `slot` of at least `bytes` bytes (contents undefined)
int smn_workspace(smn_ctx* ctx, int slot, size_t bytes, void** out);
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per kernel and context (it costs ~20 us of host time per call: a
// reference-sized SPR.loss is 13 launches and 130 us in all).
int smn_allow_lds(smn_ctx* ctx, const void* kernel, size_t lds);

inline size_t dtype_size(int dtype) { return dtype == SMN_F64 ? 8 : 4; }
