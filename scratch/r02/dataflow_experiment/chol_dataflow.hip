// chol_dataflow.hip — the blocked Cholesky of cholesky.hip as ONE persistent launch with tile-granular dependencies.
//
// Why: the stream schedule of cholesky.hip loses ~3 ms of its 16 ms at N = 16384 to things no kernel can fix from the
// inside (profiles/r02_notes.md): 32 CUs stay reserved for the panel chain for the whole factorisation, every far-update
// launch ends on a tail of half-empty CUs, and a chain step waits for whole launches (F0) where it needs one tile.  Here
// every 128x128 tile operation is a TASK; workgroups of one resident grid take tasks from one list and wait, per task, for
// exactly the tiles that task reads.
//
// Tasks (same arithmetic, same K grouping per tile as the two-level schedule of cholesky.hip):
//   POTRF(j)         factor the diagonal tile, invert it in place (potrf_tile.hpp), publish L_jj and X_j = L_jj^-1
//   TRSM(i, j)       tile (i,j) <- tile (i,j) X_j^T   (a K = 128 GEMM on the tile engine)
//   UPDATE(i, c, K)  tile (i,c) -= L[i, K] L[c, K]^T   K = 128 (strip), 256 (near), the super-panel width (far)
// State, in device memory, zeroed per call:  done[i][j] = 1 once tile (i,j) is final;  ver[i][c] = updates applied so far.
// A task lists what it needs: UPDATE: ver[i][c] == its index in the tile's update sequence, done[i][k] and done[c][k] over its
// K range;  TRSM: done[j][j] and ver[i][j] == the tile's update count;  POTRF: ver[j][j] == the tile's update count.
//
// Order.  ONE program order (the sequential two-level algorithm, tile by tile: a topological order of the task graph) is
// cut into four queues that each keep that order:
//   Q0  every task whose tile lies on the diagonal or right below it (i - c <= 1): the critical path
//   Q1  the other panel-chain tasks (TRSM, strip and near updates)
//   Q2  far updates whose tile column belongs to the NEXT super-panel (what the next chain waits for)
//   Q3  the other far updates (the bulk), row-major inside a super-panel, the order their operand rows become final in
// A free workgroup looks at the heads of Q0..Q3 in this order and CLAIMS (compare-and-swap on the queue's head) the first
// head whose dependencies are all met; it never holds a task it cannot run, so no slot is ever spent waiting while other
// work is ready.  No deadlock: the unclaimed task that comes first in program order has every dependency claimed (hence
// running, hence finishing) and is the head of its queue.  (The first version took tickets from ONE list in the start order
// of a list schedule simulated on the host: wherever the real run lagged the model, workgroups sat on tasks that were
// not ready -- 31 % of all slot time at N = 16384 -- and the critical path waited for its turn: profiles/r02_notes.md.)
// Spins are bounded; a timeout raises `err` and every workgroup leaves.
//
// Hand-offs (cdna_hip_programming.md Guideline 16; checked on this chip with scratch/r02/coherence/xcd_handoff.hip: a
// consumer whose L1 / L2 hold the OLD version of a line reads the new one after this sequence, 0 stale words in 1.9e9):
//   producer: write-through (sc1) stores of the tile -> every wave s_waitcnt vmcnt(0) -> barrier -> one lane: relaxed
//             agent-scope atomic on the flag;
//   consumer: one lane polls (relaxed, agent) -> ONE acquire fence (agent) -> s_waitcnt vmcnt(0) -> barrier -> plain loads.
#include <algorithm>
#include <vector>

#include "internal.hpp"
#include "potrf_tile.hpp"

namespace {

using potrf_detail::DiagCfg;
using potrf_detail::PB;
using potrf_detail::store_wt;

enum : uint16_t { DF_POTRF = 0, DF_TRSM = 1, DF_UPDATE = 2 };

struct DfTask {            // 12 bytes
  uint16_t type, i, c;     // tile row, tile column (POTRF: i == c)
  uint16_t k0, nk;         // UPDATE: first K tile, number of K tiles
  uint16_t need;           // ver[i][c] this task waits for
};

constexpr int kQueues = 4;

struct DfState {           // device words polled / bumped inside the launch
  int head[kQueues];       // next unclaimed task of each queue
  int err;                 // 1: a spin timed out
  int pad[11];
};

template <typename T>
struct DfArgs {
  T* a; int64_t lda;
  const DfTask* tasks; int ntasks;     // queue q = tasks[qoff[q] .. qoff[q+1])
  int qoff[kQueues + 1];
  int* done; int* ver; int tl;         // [tl][tl] each
  DfState* st;
  T* linv;                             // [n_factor/128][128*128]
  double* logdet; int* info;
  long long* trace;                    // debugging (SMN_DF_TRACE): per ticket {claimed, ready, done} on the 100 MHz wall clock, {workgroup, XCC}
  int dbg;                             // debugging (SMN_DF_DBG): bit 0 / 1 / 2 skips the body of POTRF / TRSM / UPDATE tasks
};

constexpr int kSpinLimit = 1 << 19;    // scans with up to ~3.5 us of sleep between them: about two seconds, not forever

__device__ __forceinline__ int ld_relaxed(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The three task bodies are separate (non-inlined) functions: inlined into one loop they share one register allocation and
// the update's pipelined K loop (236 VGPRs on its own) spills 185 registers.
// ONE compare-and-swap per wave: executed with every lane active, performed by lane 0 only (exec is narrowed around the
// instruction; a 64-lane atomicCAS is 64 memory-side atomics, and 512 workgroups claiming that way saturate the fabric).
// Returns the old value in lane 0 (the other lanes' result is undefined: read it with readfirstlane).
__device__ __forceinline__ int cas_lane0(int* addr, int cmp, int val) {
  int old;
  unsigned long long saved;
  const unsigned long long data = ((unsigned long long)(unsigned)cmp << 32) | (unsigned)val;   // {swap, compare}
  asm volatile(
      "s_mov_b64 %0, exec\n\t"
      "s_mov_b64 exec, 1\n\t"
      "global_atomic_cmpswap %1, %2, %3, off sc0\n\t"
      "s_waitcnt vmcnt(0)\n\t"
      "s_mov_b64 exec, %0"
      : "=&s"(saved), "=&v"(old)
      : "v"(addr), "v"(data)
      : "memory");
  return old;
}

template <typename T>
__device__ __attribute__((noinline)) void run_update(T* __restrict__ a, int64_t lda, int64_t row0, int64_t col0, int64_t kk, int K,
                                                     char* smem) {
  using Tile = TileNT<T, kTile, kTile, 2>;
  using M = typename Tile::M;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  Tile tile;
  // acc starts at -C: the C read is in flight with the first operand loads, C_new = -acc
#pragma unroll
  for (int m = 0; m < Tile::MT; ++m)
#pragma unroll
    for (int n = 0; n < Tile::NT; ++n)
#pragma unroll
      for (int i = 0; i < M::ACC; ++i) {
        const int64_t gr = row0 + wr * Tile::WM + m * M::TM + M::acc_row(lane, i);
        const int64_t gc = col0 + wc * Tile::WN + n * M::TN + M::acc_col(lane);
        tile.acc[m][n][i] = -a[gr * lda + gc];
      }
  tile.template mainloop<1>(a + row0 * lda + kk, lda, a + col0 * lda + kk, lda, K, smem);
#pragma unroll
  for (int m = 0; m < Tile::MT; ++m)
#pragma unroll
    for (int n = 0; n < Tile::NT; ++n)
#pragma unroll
      for (int i = 0; i < M::ACC; ++i) {
        const int64_t gr = row0 + wr * Tile::WM + m * M::TM + M::acc_row(lane, i);
        const int64_t gc = col0 + wc * Tile::WN + n * M::TN + M::acc_col(lane);
        store_wt(&a[gr * lda + gc], -tile.acc[m][n][i]);
      }
}

template <typename T>
__device__ __attribute__((noinline)) void run_trsm(T* __restrict__ a, int64_t lda, int64_t row0, int64_t col0,
                                                   const T* __restrict__ linv, char* smem) {
  using Tile = TileNT<T, kTile, kTile, 2>;
  using M = typename Tile::M;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  Tile tile;
  tile.zero();
  tile.template mainloop<0>(a + row0 * lda + col0, lda, linv, PB, PB, smem);
#pragma unroll
  for (int m = 0; m < Tile::MT; ++m)
#pragma unroll
    for (int n = 0; n < Tile::NT; ++n)
#pragma unroll
      for (int i = 0; i < M::ACC; ++i) {
        const int64_t gr = row0 + wr * Tile::WM + m * M::TM + M::acc_row(lane, i);
        const int64_t gc = col0 + wc * Tile::WN + n * M::TN + M::acc_col(lane);
        store_wt(&a[gr * lda + gc], tile.acc[m][n][i]);
      }
}

template <typename T>
__device__ __attribute__((noinline)) void run_potrf(T* __restrict__ a, int64_t lda, int64_t col0, double* logdet, int* info,
                                                    T* __restrict__ linv, char* smem) {
  potrf_detail::potrf_tile<T>(a, lda, col0, 0, logdet, info, linv, smem);
}

template <typename T>
__global__ void __launch_bounds__(256, 2) chol_dataflow_kernel(DfArgs<T> g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // the last 16 bytes of the dynamic region: ticket and bail-out word (all LDS in one object: Guideline 17)
  int* ctl = reinterpret_cast<int*>(smem + DiagCfg<T>::LDS);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // NO lane-divergent branch in this loop.  Everything "one lane" would do is done by ALL lanes of wave 0 in a form that
  // is idempotent (same value to the same word) or that the compiler folds into one request (an atomic add of 1 from
  // lane 0 and 0 from the others), under the wave-uniform branch `wave == 0`.  The first version used `if (tid == 0)`
  // for the ticket and the publish: the compiler restructured the loop so that lane 0 left an inner loop that lanes 1..63
  // of its own wave kept running -- they met the next barrier without it, read the OLD ticket and ran the same task for
  // ever (found in the ISA; readfirstlane on the loop exits alone did not change that structure).
  for (;;) {
    const long long t_look = g.trace ? wall_clock64() : 0;
    if (wave == 0) {
      int got = -1;                               // index of the claimed task; -1: every queue is exhausted; -2: gave up
      int nap = 1;                                // back-off between scans that found nothing ready
      for (int spin = 0;; ++spin) {
        bool any_left = false, raced = false;
        // the four heads: one coalesced load (lane & 3 picks the word)
        const int hv = ld_relaxed(&g.st->head[lane & 3]);
        for (int q = 0; q < kQueues && got < 0 && !raced; ++q) {
          const int qn = g.qoff[q + 1] - g.qoff[q];
          const int h = __builtin_amdgcn_readlane(hv, q);
          if (h >= qn) continue;
          any_left = true;
          const DfTask t = g.tasks[g.qoff[q] + h];
          // everything the task reads, one word per lane (one L2 round trip for all of them): lane 0 the tile's update
          // count; UPDATE: lanes 1..nk the row operand's tiles, nk+1..2nk the column operand's; TRSM: lane 1 the
          // diagonal tile of its column.  Idle lanes read lane 0's word.
          size_t off = (size_t)t.i * g.tl + t.c;
          bool use_done = false;
          if (t.type == DF_TRSM) {
            use_done = lane == 1;
            off = use_done ? (size_t)t.c * g.tl + t.c : off;
          } else if (t.type == DF_UPDATE) {
            const bool rowop = lane >= 1 && lane <= t.nk, colop = lane > t.nk && lane <= 2 * t.nk;
            use_done = rowop || colop;
            off = rowop ? (size_t)t.i * g.tl + t.k0 + lane - 1 : (colop ? (size_t)t.c * g.tl + t.k0 + lane - 1 - t.nk : off);
          }
          const int want = use_done ? 1 : (int)t.need;
          if (!__all(ld_relaxed((use_done ? g.done : g.ver) + off) == want)) continue;
          // ready: claim it (ONE atomic, h -> h + 1)
          const int old = __builtin_amdgcn_readfirstlane(cas_lane0(&g.st->head[q], h, h + 1));
          if (old == h) got = g.qoff[q] + h;
          else raced = true;                      // somebody else was faster: look again from Q0
        }
        if (got >= 0 || !any_left) break;
        if (raced) continue;
        if (spin > kSpinLimit || ((spin & 63) == 63 && __builtin_amdgcn_readfirstlane(ld_relaxed(&g.st->err)))) {
          got = -2;
          break;
        }
        // nothing ready: sleep (64 clocks per unit), longer each time up to ~3.5 us, so that idle workgroups do not load
        // the fabric the running ones stream their operands through
        for (int z = 0; z < nap; ++z) __builtin_amdgcn_s_sleep(16);
        if (nap < 8) ++nap;
      }
      if (got == -2) __hip_atomic_store(&g.st->err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ctl[0] = got;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // after the polls that showed the task's inputs final
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    // readfirstlane: the claimed index is wave-uniform and the compiler has to KNOW it (uniform loop exit)
    const int ticket = __builtin_amdgcn_readfirstlane(ctl[0]);
    if (ticket < 0) break;
    const DfTask t = g.tasks[ticket];
    const int64_t row0 = (int64_t)t.i * kTile, col0 = (int64_t)t.c * kTile;
    if (g.trace && tid == 64) g.trace[4 * (size_t)ticket + 0] = t_look;

    const int type = __builtin_amdgcn_readfirstlane((int)t.type);
    if (g.trace && tid == 64) g.trace[4 * (size_t)ticket + 1] = wall_clock64();
    if ((g.dbg >> type) & 1) {
    } else if (type == DF_POTRF) run_potrf<T>(g.a, g.lda, col0, g.logdet, g.info, g.linv + (size_t)t.c * (PB * PB), smem);
    else if (type == DF_TRSM) run_trsm<T>(g.a, g.lda, row0, col0, g.linv + (size_t)t.c * (PB * PB), smem);
    else run_update<T>(g.a, g.lda, row0, col0, (int64_t)t.k0 * kTile, (int)t.nk * kTile, smem);
    // publish: every storing wave has drained its write-through stores, then wave 0 moves the tile's word
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (g.trace && tid == 64) {
      g.trace[4 * (size_t)ticket + 2] = wall_clock64();
      g.trace[4 * (size_t)ticket + 3] = (long long)blockIdx.x;
    }
    if (wave == 0) {
      int* w = (type == DF_UPDATE ? g.ver : g.done) + (size_t)t.i * g.tl + t.c;
      __hip_atomic_store(w, type == DF_UPDATE ? (int)t.need + 1 : 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // (the next iteration's first barrier separates this task's last LDS reads from the next task's staging writes)
  }
}

// ------------------------------------------------------------------------------------------------ host: the plan
struct DfPlan {
  int64_t n_total = 0, n_factor = 0, S = 0;
  int tl = 0, ntasks = 0;
  int qoff[kQueues + 1] = {};
  DfTask* tasks_dev = nullptr;
  int* state_dev = nullptr;      // DfState, then done[tl*tl], then ver[tl*tl]
  size_t state_bytes = 0;
};

// Tasks in program order (the two-level algorithm of cholesky.hip, tile by tile), cut into the four queues.
void build_queues(int tl, int tf, int spt, std::vector<DfTask>* out, int qoff[kQueues + 1]) {
  std::vector<DfTask> q[kQueues];
  std::vector<uint16_t> nver((size_t)tl * tl, 0);
  auto at = [&](int i, int c) { return (size_t)i * tl + c; };
  auto put = [&](const DfTask& t, int next_sp_begin, int next_sp_end) {
    int k;
    if (t.i - t.c <= 1) k = 0;                                          // diagonal and first sub-diagonal: the critical path
    else if (t.type != DF_UPDATE || t.nk <= 2) k = 1;                   // the rest of the panel chain
    else k = (t.c >= next_sp_begin && t.c < next_sp_end) ? 2 : 3;       // far updates: next super-panel's columns first
    q[k].push_back(t);
  };
  auto update = [&](int i, int c, int k0, int nk, int nb, int ne) {
    put(DfTask{DF_UPDATE, (uint16_t)i, (uint16_t)c, (uint16_t)k0, (uint16_t)nk, nver[at(i, c)]}, nb, ne);
    ++nver[at(i, c)];
  };
  for (int s0 = 0; s0 < tf; s0 += spt) {
    const int s1 = std::min(tf, s0 + spt), s2 = std::min(tf, s1 + spt);
    for (int j0 = s0; j0 < s1; j0 += 2) {
      const int w = std::min(2, s1 - j0);
      for (int js = j0; js < j0 + w; ++js) {
        if (js > j0)                                         // strip: column js by column j0
          for (int i = js; i < tl; ++i) update(i, js, j0, 1, s1, s2);
        put(DfTask{DF_POTRF, (uint16_t)js, (uint16_t)js, 0, 0, nver[at(js, js)]}, s1, s2);
        for (int i = js + 1; i < tl; ++i) put(DfTask{DF_TRSM, (uint16_t)i, (uint16_t)js, 0, 0, nver[at(i, js)]}, s1, s2);
      }
      for (int i = j0 + w; i < tl; ++i)                      // near: the rest of the super-panel by this pair, row by row
        for (int c = j0 + w; c < s1 && c <= i; ++c) update(i, c, j0, w, s1, s2);
    }
    for (int i = s1; i < tl; ++i)                            // far: everything right of the super-panel, row by row (the
      for (int c = s1; c <= i; ++c) update(i, c, s0, s1 - s0, s1, s2);   // order the operand rows become final in)
  }
  out->clear();
  for (int k = 0; k < kQueues; ++k) {
    qoff[k] = (int)out->size();
    out->insert(out->end(), q[k].begin(), q[k].end());
  }
  qoff[kQueues] = (int)out->size();
}

// a spin that timed out leaves err = 1: the factorisation is invalid, info = -1 tells the caller (the facade maps any info != 0 to NaN)
__global__ void df_check_kernel(const DfState* st, int* info) {
  if (st->err) *info = -1;
}

}  // namespace

// One plan per context, rebuilt when the shape changes.
struct smn_df_cache {
  DfPlan plan;
};

void dataflow_release(smn_ctx* ctx) {
  smn_df_cache* c = static_cast<smn_df_cache*>(ctx->df_cache);
  if (!c) return;
  if (c->plan.tasks_dev) (void)hipFree(c->plan.tasks_dev);
  if (c->plan.state_dev) (void)hipFree(c->plan.state_dev);
  delete c;
  ctx->df_cache = nullptr;
}

// The factorisation of cholesky_padded as one persistent launch.  Returns SMN_ENOTSUP when this shape / dtype stays on the
// stream schedule (the caller then runs that); the diagonal shift has been applied by the caller.
int cholesky_dataflow(smn_ctx* ctx, int dtype, void* a, int64_t n_total, int64_t n_factor, int64_t lda) {
  if (dtype != SMN_F32) return SMN_ENOTSUP;             // the f64 diagonal block alone takes 136 KB of LDS: one workgroup per CU
  if (ctx->chol_id0 >= 0) return SMN_ENOTSUP;           // the identity-row hint of the gradient path is a stream-schedule feature
  const int tl = (int)(n_total / kTile), tf = (int)(n_factor / kTile);
  if (tl > ctx->dataflow_max_tiles || tl < 2) return SMN_ENOTSUP;
  int64_t S = ctx->super_panel / (2 * kTile) * (2 * kTile);
  if (S < 2 * kTile) S = 2 * kTile;
  smn_df_cache* c = static_cast<smn_df_cache*>(ctx->df_cache);
  if (!c) {
    c = new smn_df_cache();
    ctx->df_cache = c;
  }
  DfPlan& p = c->plan;
  if (p.n_total != n_total || p.n_factor != n_factor || p.S != S) {
    SMN_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (p.tasks_dev) (void)hipFree(p.tasks_dev);
    if (p.state_dev) (void)hipFree(p.state_dev);
    p = DfPlan();
    std::vector<DfTask> order;
    build_queues(tl, tf, (int)(S / kTile), &order, p.qoff);
    p.ntasks = (int)order.size();
    p.tl = tl;
    p.state_bytes = (sizeof(DfState) + 2 * sizeof(int) * (size_t)tl * tl + 15) / 16 * 16;
    SMN_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&p.tasks_dev), sizeof(DfTask) * order.size()));
    SMN_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&p.state_dev), p.state_bytes));
    SMN_HIP(ctx, hipMemcpy(p.tasks_dev, order.data(), sizeof(DfTask) * order.size(), hipMemcpyHostToDevice));
    p.n_total = n_total; p.n_factor = n_factor; p.S = S;
    if (getenv("SMN_DF_DEBUG"))
      fprintf(stderr, "dataflow plan: n_total %lld n_factor %lld S %lld: %d tasks in queues of %d / %d / %d / %d\n",
              (long long)n_total, (long long)n_factor, (long long)S, p.ntasks, p.qoff[1] - p.qoff[0], p.qoff[2] - p.qoff[1],
              p.qoff[3] - p.qoff[2], p.qoff[4] - p.qoff[3]);
  }
  void* side = nullptr;   // inverses of the diagonal blocks
  SMN_TRY(smn_workspace(ctx, 3, sizeof(float) * (size_t)n_factor * PB, &side));
  hipStream_t st = ctx->stream;
  SMN_HIP(ctx, hipMemsetAsync(p.state_dev, 0, p.state_bytes, st));
  DfArgs<float> g;
  g.a = static_cast<float*>(a); g.lda = lda;
  g.tasks = p.tasks_dev; g.ntasks = p.ntasks;
  for (int k = 0; k <= kQueues; ++k) g.qoff[k] = p.qoff[k];
  g.st = reinterpret_cast<DfState*>(p.state_dev);
  g.done = p.state_dev + sizeof(DfState) / sizeof(int);
  g.ver = g.done + (size_t)tl * tl;
  g.tl = tl;
  g.linv = static_cast<float*>(side);
  g.logdet = ctx->d_scal; g.info = ctx->d_info;
  g.dbg = getenv("SMN_DF_DBG") ? atoi(getenv("SMN_DF_DBG")) : 0;
  g.trace = nullptr;
  const char* trace_path = getenv("SMN_DF_TRACE");
  if (trace_path) {   // debugging: one record per ticket, written to the file with the task list in front
    SMN_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&g.trace), 32 * (size_t)p.ntasks));
    SMN_HIP(ctx, hipMemsetAsync(g.trace, 0, 32 * (size_t)p.ntasks, st));
  }
  const size_t lds = DiagCfg<float>::LDS + 16;
  static_assert(DiagCfg<float>::LDS % 16 == 0, "control words start 16-byte aligned");
  static_assert(DiagCfg<float>::LDS >= TileNT<float, kTile, kTile, 2>::LDS_BYTES, "the tile engine's stages fit the diagonal image");
  auto kern = chol_dataflow_kernel<float>;
  if (!ctx->df_attr_done) {
    SMN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ctx->df_attr_done = true;
  }
  int per_cu = 0;
  SMN_HIP(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, lds));
  if (per_cu < 1) return SMN_ENOTSUP;
  if (per_cu > 2) per_cu = 2;
  {
    ProfScope ps(ctx, PROF_TRAIL, st);
    hipLaunchKernelGGL(kern, dim3((unsigned)(per_cu * ctx->num_cu)), dim3(256), lds, st, g);
  }
  hipLaunchKernelGGL(df_check_kernel, dim3(1), dim3(1), 0, st, reinterpret_cast<const DfState*>(p.state_dev), ctx->d_info);
  SMN_CHECK_LAUNCH(ctx);
  if (trace_path) {
    SMN_HIP(ctx, hipStreamSynchronize(st));
    std::vector<long long> tr(4 * (size_t)p.ntasks);
    std::vector<DfTask> tk(p.ntasks);
    SMN_HIP(ctx, hipMemcpy(tr.data(), g.trace, 32 * (size_t)p.ntasks, hipMemcpyDeviceToHost));
    SMN_HIP(ctx, hipMemcpy(tk.data(), p.tasks_dev, sizeof(DfTask) * (size_t)p.ntasks, hipMemcpyDeviceToHost));
    (void)hipFree(g.trace);
    if (FILE* f = fopen(trace_path, "wb")) {
      const long long n = p.ntasks;
      fwrite(&n, 8, 1, f);
      fwrite(tk.data(), sizeof(DfTask), tk.size(), f);
      fwrite(tr.data(), 8, tr.size(), f);
      fclose(f);
    }
  }
  return SMN_OK;
}
