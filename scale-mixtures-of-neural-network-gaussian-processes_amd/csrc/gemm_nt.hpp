// gemm_nt.hpp — the one MFMA tile engine every dense kernel of the path is built on.
//
// Computes a BM x BN block of  A[M,K] * B[N,K]^T  (both operands row-major, K contiguous: the
// "NT" form, which is what X X^T, P P^T and B L^-T all are) with the exact-f32 MFMA
// v_mfma_f32_16x16x4_f32 or the f64 MFMA v_mfma_f64_16x16x4_f64.  256 threads = 4 waves in a
// 2x2 arrangement; each wave owns a (BM/2) x (BN/2) sub-block in accumulator registers.
//
// LDS image: one 128-byte row per matrix row per K-step (32 f32 / 16 f64), 16-byte chunks
// XOR-swizzled by ((row >> 1) & 7) so the ds_read_b128 fragment reads of 16 different rows hit 16
// different 16-byte slots of the 256-byte bank row (cdna_hip_programming.md section 2 / T2).
// Fragments are read 16 bytes per lane; the K index inside a K-step is permuted identically for
// both operands (a dot product does not care), which lets one b128 read feed 4 (f32) / 2 (f64)
// MFMAs instead of one b32/b64 read per MFMA.
//
// Staging is register double-buffering (global_load_dwordx4 for step k+1 issued before the MFMAs
// of step k, ds_write_b128 after them, one barrier per K-step): the f32 MFMA is so slow (64 flop per
// cycle and SIMD) that a K-step of one wave is ~4k cycles, an order of magnitude above HBM latency.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <typename T>
struct Mfma;

// v_mfma_f32_16x16x4_f32: the same 64 flop/cycle/SIMD as the 32x32x2 form, but an accumulator register is rewritten every
// 4 k instead of every 2: half the accumulator traffic on the SIMD's VGPR ports, which the 32x32x2 form saturates by itself
// (every LDS / global-load return then steals MFMA cycles: scratch/r02/tile_probe, profiles/r02_tile_probe.txt).
template <>
struct Mfma<float> {
  static constexpr int TM = 16, TN = 16;
  static constexpr int ACC = 4;
  static constexpr int BK = 32;
  static constexpr int VEC = 4;
  static constexpr int KG = 2;
  using acc_t = f32x4;
  using vec_t = f32x4;
  static __device__ __forceinline__ int frag_row(int lane) { return lane & 15; }
  static __device__ __forceinline__ int frag_chunk(int lane, int g) { return 4 * g + (lane >> 4); }
  static __device__ __forceinline__ void mma(acc_t& c, const vec_t& a, const vec_t& b) {
#pragma unroll
    for (int i = 0; i < 4; ++i) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[i], c, 0, 0, 0);
  }
  static __device__ __forceinline__ int acc_row(int lane, int i) { return 4 * (lane >> 4) + i; }
  static __device__ __forceinline__ int acc_col(int lane) { return lane & 15; }
};

template <>
struct Mfma<double> {
  static constexpr int TM = 16, TN = 16;
  static constexpr int ACC = 4;
  static constexpr int BK = 16;
  static constexpr int VEC = 2;
  static constexpr int KG = 2;
  using acc_t = f64x4;
  using vec_t = f64x2;
  static __device__ __forceinline__ int frag_row(int lane) { return lane & 15; }
  static __device__ __forceinline__ int frag_chunk(int lane, int g) { return 4 * g + (lane >> 4); }
  static __device__ __forceinline__ void mma(acc_t& c, const vec_t& a, const vec_t& b) {
#pragma unroll
    for (int i = 0; i < 2; ++i) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[i], c, 0, 0, 0);
  }
  // f64 MFMA has its own C/D map: col = lane & 15, row = (lane >> 4) + 4 * reg
  static __device__ __forceinline__ int acc_row(int lane, int i) { return (lane >> 4) + 4 * i; }
  static __device__ __forceinline__ int acc_col(int lane) { return lane & 15; }
};

template <typename T, int BM_, int BN_, int STAGES_ = 2>
struct TileNT {
  using M = Mfma<T>;
  using acc_t = typename M::acc_t;
  using vec_t = typename M::vec_t;
  static constexpr int BM = BM_, BN = BN_;
  static constexpr int THREADS = 256;
  static constexpr int WM = BM / 2, WN = BN / 2;              // per-wave sub-block
  static constexpr int MT = WM / M::TM, NT = WN / M::TN;      // MFMA tiles per wave
  static constexpr int ROWB = 128;                            // LDS bytes per row per K-step
  static constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
  static constexpr int STAGE = A_BYTES + B_BYTES;
  static constexpr int STAGES = STAGES_;                      // 2: double-buffered LDS (1 barrier per K-step); 1: single buffer
  static constexpr int LDS_BYTES = STAGES * STAGE;           //    (2 barriers per K-step, half the LDS -> twice the workgroups per CU)
  static constexpr int PA = BM / 32, PB = BN / 32;            // 16-byte loads per thread per K-step
  static_assert(BM % 32 == 0 && BN % 32 == 0 && WM % M::TM == 0 && WN % M::TN == 0, "tile shape");
  static_assert(WM % 16 == 0 && WN % 16 == 0, "swizzle assumes 16-row aligned wave blocks");

  acc_t acc[MT][NT];

  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int i = 0; i < M::ACC; ++i) acc[m][n][i] = T(0);
  }

  __device__ __forceinline__ void compute(const char* stage, int lane, int wr, int wc) {
    const char* sa = stage + (wr * WM) * ROWB;
    const char* sb = stage + A_BYTES + (wc * WN) * ROWB;
    const int fr = M::frag_row(lane);
    const int swz = (fr >> 1) & 7;
#pragma unroll
    for (int g = 0; g < M::KG; ++g) {
      const int off = fr * ROWB + ((M::frag_chunk(lane, g) ^ swz) << 4);
      vec_t a[MT], b[NT];
#pragma unroll
      for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const vec_t*>(sa + m * M::TM * ROWB + off);
#pragma unroll
      for (int n = 0; n < NT; ++n) b[n] = *reinterpret_cast<const vec_t*>(sb + n * M::TN * ROWB + off);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) M::mma(acc[m][n], a[m], b[n]);
    }
  }

  // acc += A[0:BM, 0:K] * B[0:BN, 0:K]^T.  A/B point at the first row of the block; K % BK == 0,
  // lda/ldb % VEC == 0 and 16-byte aligned bases (the callers pad).  Ends with a barrier, so the
  // caller may reuse smem right away.
  //
  // Schedule of one K-step (KG fragment groups; double-buffered LDS stages, double-buffered fragments), written so that
  // the MFMA stream of a wave never waits on the LDS (rocprofv3 PMC on the first version: MFMA issue slots busy 76-79 %,
  // the rest per-K-step bubbles: fragment reads issued right in front of the MFMAs that need them, the staging writes
  // and the first reads of the next step all behind the last MFMA):
  //   group g          fragments of g+1 are read BEFORE the MFMAs of g (a whole group, 1k cycles, ahead of their use);
  //   group WG         the next stage's ds_writes are spread between this group's MFMAs (their global loads were issued
  //                    a K-step earlier);
  //   group KG-1       the barrier sits IN FRONT of the last group's MFMAs: every wave has written the next stage and has
  //                    issued all its reads of this one, the last group's fragments are already in registers, so the
  //                    MFMAs start at once and the first fragments of the NEXT step + the global loads of the step after
  //                    it ride under them.
  // Same arithmetic in the same order as the plain loop: identical bits.  (Round-2 A/B switches -- the 32x32x2 MFMA form,
  // the un-pipelined loop, later global loads, a later staging-write group, the f64 128x128 tile on the pipelined loop
  // (it spills) and the timing probes that each dropped one ingredient -- are gone from the source; their measurements are
  // profiles/r02_tile_probe.txt, r02_mfma16_ab.txt, r02_gload_placement_ab.txt, r01f_pipelined_kloop_ab.txt.)
  __device__ __forceinline__ void frag_load(const char* stage, int g, int lane, int wr, int wc, vec_t (&a)[MT], vec_t (&b)[NT]) {
    const char* sa = stage + (wr * WM) * ROWB;
    const char* sb = stage + A_BYTES + (wc * WN) * ROWB;
    const int fr = M::frag_row(lane);
    const int off = fr * ROWB + ((M::frag_chunk(lane, g) ^ ((fr >> 1) & 7)) << 4);
#pragma unroll
    for (int m = 0; m < MT; ++m) a[m] = *reinterpret_cast<const vec_t*>(sa + m * M::TM * ROWB + off);
#pragma unroll
    for (int n = 0; n < NT; ++n) b[n] = *reinterpret_cast<const vec_t*>(sb + n * M::TN * ROWB + off);
  }

  // MODE 1: pipelined loop only, 0: plain loop only, 2: chosen at run time (pipelined from 8 K-steps on).  Kernels that
  // know their K range pick 0 or 1: one loop in the kernel instead of two keeps the f32 128x128 tile off the register limit.
  template <int MODE = 2>
  __device__ __forceinline__ void mainloop(const T* __restrict__ A, int64_t lda,
                                           const T* __restrict__ B, int64_t ldb, int K, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int lrow = tid >> 3, lchunk = tid & 7;
    const int wpos = lrow * ROWB + ((lchunk ^ ((lrow >> 1) & 7)) << 4);
    const T* ga = A + (int64_t)lrow * lda + lchunk * M::VEC;
    const T* gb = B + (int64_t)lrow * ldb + lchunk * M::VEC;
    vec_t ra[PA], rb[PB];
    const int nk = K / M::BK;
    auto gload = [&](int kt) {
      const int ko = kt * M::BK;
#pragma unroll
      for (int p = 0; p < PA; ++p) ra[p] = *reinterpret_cast<const vec_t*>(ga + (int64_t)(32 * p) * lda + ko);
#pragma unroll
      for (int p = 0; p < PB; ++p) rb[p] = *reinterpret_cast<const vec_t*>(gb + (int64_t)(32 * p) * ldb + ko);
    };

    gload(0);
#pragma unroll
    for (int p = 0; p < PA; ++p) *reinterpret_cast<vec_t*>(smem + wpos + 32 * p * ROWB) = ra[p];
#pragma unroll
    for (int p = 0; p < PB; ++p) *reinterpret_cast<vec_t*>(smem + A_BYTES + wpos + 32 * p * ROWB) = rb[p];
    __syncthreads();

    // (the f64 128x128 tile has no registers left for a second fragment set: it keeps the plain loop)
    constexpr bool can_pipe = MODE != 0 && STAGES == 2 && M::KG >= 2 && !(sizeof(T) == 8 && MT * NT >= 16);
    if (can_pipe && (MODE == 1 ? nk >= 1 : nk >= 8)) {
      constexpr int KG = M::KG;
      constexpr int WG = 0;                           // group whose MFMAs the staging writes are spread between
      constexpr int NW = PA + PB;                     // staging writes per thread per K-step
      constexpr int NMM = MT * NT;                    // MFMA tiles per group
      // The loop body is branch-free: the last K-step also runs its barrier, its "next" fragment reads, its staging writes
      // (into the stage nobody reads any more) and global loads (clamped to the last valid K-step).  Those redundant loads
      // are waited for by the next staging writes, which costs short loops more than the pipelining buys them: K-loops
      // of fewer than 8 steps take the plain loop below.
      const int last = nk - 1;
      gload(1 < last ? 1 : last);
      vec_t fa[2][MT], fb[2][NT];
      frag_load(smem, 0, lane, wr, wc, fa[0], fb[0]);
      for (int kt = 0; kt < nk; ++kt) {
        const char* stage = smem + (kt & 1) * STAGE;
        char* nstage = smem + ((kt + 1) & 1) * STAGE;
#pragma unroll
        for (int g = 0; g < KG; ++g) {
          const int cur = g & 1, nxt = cur ^ 1;
          if (g + 1 < KG) {
            frag_load(stage, g + 1, lane, wr, wc, fa[nxt], fb[nxt]);
          } else {
            // Every read of THIS stage must be issued before the barrier: a faster wave overwrites it one group into the
            // next K-step.  The last group's fragments were read a group ago; the empty asm consumes them here so that no
            // compiler pass can sink those reads past the barrier.
#pragma unroll
            for (int m = 0; m < MT; ++m) asm volatile("" : "+v"(fa[cur][m]));
#pragma unroll
            for (int n = 0; n < NT; ++n) asm volatile("" : "+v"(fb[cur][n]));
            __syncthreads();   // in front of the last group's MFMAs (see above)
            frag_load(nstage, 0, lane, wr, wc, fa[nxt], fb[nxt]);
          }
          // sched_barrier(0): the compiler's scheduler otherwise sinks the reads back down in front of their first use
          // and gathers the staging writes behind the last MFMA -- the very bubbles this loop is written to remove
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < NMM; ++j) {
            M::mma(acc[j / NT][j % NT], fa[cur][j / NT], fb[cur][j % NT]);
            if (g == WG) {
              // the NW staging writes of the next stage, spread over the NMM tile products of this group
#pragma unroll
              for (int w = 0; w < NW; ++w) {
                if (w >= (j * NW) / NMM && w < ((j + 1) * NW) / NMM) {
                  if (w < PA) *reinterpret_cast<vec_t*>(nstage + wpos + 32 * w * ROWB) = ra[w < PA ? w : 0];
                  else *reinterpret_cast<vec_t*>(nstage + A_BYTES + wpos + 32 * (w - PA) * ROWB) = rb[w >= PA ? w - PA : 0];
                }
              }
              __builtin_amdgcn_sched_barrier(0);
              // the staging registers are free as soon as their writes have issued: the loads of the step after the next
              // go out here, a whole K-step (not half of one) ahead of the writes that wait for them (clamped, not
              // branched around: with a branch the compiler sank the previous group's fragment reads below the barrier)
              if (j == NMM - 1) {
                gload(kt + 2 < last ? kt + 2 : last);
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __syncthreads();
      return;
    }

    if (can_pipe && MODE == 1) return;   // (nk == 0: nothing to add; the prologue barrier has run)
    for (int kt = 0; kt < nk; ++kt) {
      const bool more = kt + 1 < nk;
      if (more) gload(kt + 1);
      compute(smem + (STAGES == 2 ? (kt & 1) * STAGE : 0), lane, wr, wc);
      if (more) {
        if (STAGES == 1) __syncthreads();   // every wave is done reading the only buffer
        char* st = smem + (STAGES == 2 ? ((kt + 1) & 1) * STAGE : 0);
#pragma unroll
        for (int p = 0; p < PA; ++p) *reinterpret_cast<vec_t*>(st + wpos + 32 * p * ROWB) = ra[p];
#pragma unroll
        for (int p = 0; p < PB; ++p) *reinterpret_cast<vec_t*>(st + A_BYTES + wpos + 32 * p * ROWB) = rb[p];
      }
      __syncthreads();
    }
  }

  // Visit every accumulator element of this lane: f(m, n, i, local_row, local_col) with
  // local_* relative to the block's top-left corner.
  template <typename F>
  __device__ __forceinline__ void for_each(F&& f) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int i = 0; i < M::ACC; ++i)
          f(m, n, i, wr * WM + m * M::TM + M::acc_row(lane, i), wc * WN + n * M::TN + M::acc_col(lane));
  }
};

// Lower-triangular tile enumeration: linear index t -> (tr, tc) with tc <= tr, row-major.
__host__ __device__ __forceinline__ void tri_decode(int64_t t, int& tr, int& tc) {
  int r = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
  while ((int64_t)(r + 1) * (r + 2) / 2 <= t) ++r;
  while ((int64_t)r * (r + 1) / 2 > t) --r;
  tr = r;
  tc = (int)(t - (int64_t)r * (r + 1) / 2);
}

// XCD-aware blockIdx -> tile map (cdna_hip_programming.md T1).  Workgroups are dealt round-robin over
// the 8 XCDs, each with its own 4 MiB L2, so block b and block b+8 share an L2.  Tiles are enumerated
// patch-major (patches of PS x PS tiles, row-major inside a patch; of a diagonal patch of a lower-triangular grid only
// its lower half) and every XCD gets one contiguous, EQUAL share of that order: the ~64 workgroups resident on an XCD
// then work on one patch and share 2*PS operand panels out of L2 instead of fetching 2 per tile from the Infinity Cache
// (fabric traffic of the Cholesky's trailing updates 377 -> 261 MB per launch, profiles/r02_pmc_traffic.json).  The first
// version of this map enumerated whole 64-slot patches, so XCDs whose share held many half-empty diagonal patches ran
// out of work early and the order measured 9 % SLOWER than the linear one; balanced it is 0.15 ms per step faster.
// Placement only changes speed: any dispatch order computes the same tiles.
struct TileMap {
  int tm, tn;        // tile grid
  int lower;         // 1: only tiles with tc <= tr (tm == tn)
  int pm, pn;        // patch grid
  int npatch;        // rectangular grids: patches enumerated
  int ntiles;        // lower grids: tm (tm + 1) / 2, enumerated WITHOUT the empty upper halves of diagonal patches
  int grid;          // launch size, a multiple of 8: every XCD gets the same number of consecutive tiles of the order
  static constexpr int PS = 8;
  static TileMap make(int64_t tm_, int64_t tn_, int lower_) {
    TileMap t;
    t.tm = (int)tm_; t.tn = (int)tn_; t.lower = lower_;
    t.pm = (t.tm + PS - 1) / PS; t.pn = (t.tn + PS - 1) / PS;
    t.npatch = lower_ ? t.pm * (t.pm + 1) / 2 : t.pm * t.pn;
    t.ntiles = lower_ ? t.tm * (t.tm + 1) / 2 : t.npatch * PS * PS;
    t.grid = (t.ntiles + 7) / 8 * 8;
    return t;
  }
  __device__ __forceinline__ bool decode(unsigned b, int& tr, int& tc) const {
    const int chunk = grid >> 3;
    const int l = (int)(b & 7) * chunk + (int)(b >> 3);
    return decode_linear(l, tr, tc);
  }
  // tile l of the patch-major order (host too: the tile lists of a split build are written from it)
  __host__ __device__ __forceinline__ bool decode_linear(int l, int& tr, int& tc) const {
    if (l >= ntiles) return false;
    if (lower) {
      // patch row q (PS tile rows) holds q full patches and the lower half of a diagonal one: 64 q + 36 tiles, so
      // 32 q^2 + 4 q tiles lie in front of it; the last patch row may have fewer tile rows.
      int q = (int)((sqrt(16.0 + 128.0 * (double)l) - 4.0) * (1.0 / 64.0));
      while (32 * (q + 1) * (q + 1) + 4 * (q + 1) <= l) ++q;
      while (32 * q * q + 4 * q > l) --q;
      if (q > pm - 1) q = pm - 1;
      const int off = l - (32 * q * q + 4 * q);
      const int rows = (q == pm - 1) ? tm - PS * q : PS;      // tile rows of this patch row
      const int per = rows * PS;                               // tiles of one of its full-width patches
      if (off < per * q) {
        const int in = off % per;
        tr = q * PS + in / PS;
        tc = (off / per) * PS + in % PS;
      } else {
        int a, c;
        tri_decode(off - per * q, a, c);
        tr = q * PS + a;
        tc = q * PS + c;
      }
      return true;
    }
    const int patch = l / (PS * PS), slot = l % (PS * PS);
    if (patch >= npatch) return false;
    tr = (patch / pn) * PS + slot / PS;
    tc = (patch % pn) * PS + slot % PS;
    return tr < tm && tc < tn;
  }
};

// The 128 x 128 tile every dense kernel of the path uses.  SMN_STAGES (build-time) selects LDS
// double-buffering (2, default) or a single buffer (1).
#ifndef SMN_STAGES
#define SMN_STAGES 2
#endif
template <typename T>
using MainTile = TileNT<T, 128, 128, SMN_STAGES>;
