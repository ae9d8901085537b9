// gemm_nt.hpp — the one MFMA tile engine every dense kernel of the path is built on.
//
// Computes a BM x BN block of  A[M,K] * B[N,K]^T  (both operands row-major, K contiguous: the
// "NT" form, which is what X X^T, P P^T and B L^-T all are) with the exact-f32 MFMA
// v_mfma_f32_32x32x2_f32 or the f64 MFMA v_mfma_f64_16x16x4_f64.  256 threads = 4 waves in a
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
// of step k, ds_write_b128 after them, one barrier per K-step): the f32 MFMA is so slow (64 cycles
// per 32x32x2) that a K-step of one wave is ~4k cycles, an order of magnitude above HBM latency.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <typename T>
struct Mfma;

template <>
struct Mfma<float> {
  static constexpr int TM = 32, TN = 32;  // MFMA output tile
  static constexpr int ACC = 16;          // accumulator elements per lane per tile
  static constexpr int BK = 32;           // elements per 128-byte LDS row (one K-step)
  static constexpr int VEC = 4;           // elements per 16-byte chunk
  static constexpr int KG = 4;            // fragment reads per K-step
  using acc_t = f32x16;
  using vec_t = f32x4;
  static __device__ __forceinline__ int frag_row(int lane) { return lane & 31; }
  static __device__ __forceinline__ int frag_chunk(int lane, int g) { return 2 * g + (lane >> 5); }
  static __device__ __forceinline__ void mma(acc_t& c, const vec_t& a, const vec_t& b) {
#pragma unroll
    for (int i = 0; i < 4; ++i) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[i], c, 0, 0, 0);
  }
  // C/D map of the 32x32 f32 accumulator (cdna_hip_programming.md section 3)
  static __device__ __forceinline__ int acc_row(int lane, int i) {
    return (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
  }
  static __device__ __forceinline__ int acc_col(int lane) { return lane & 31; }
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

#pragma unroll
    for (int p = 0; p < PA; ++p) ra[p] = *reinterpret_cast<const vec_t*>(ga + (int64_t)(32 * p) * lda);
#pragma unroll
    for (int p = 0; p < PB; ++p) rb[p] = *reinterpret_cast<const vec_t*>(gb + (int64_t)(32 * p) * ldb);
#pragma unroll
    for (int p = 0; p < PA; ++p) *reinterpret_cast<vec_t*>(smem + wpos + 32 * p * ROWB) = ra[p];
#pragma unroll
    for (int p = 0; p < PB; ++p) *reinterpret_cast<vec_t*>(smem + A_BYTES + wpos + 32 * p * ROWB) = rb[p];
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
      const bool more = kt + 1 < nk;
      if (more) {
        const int ko = (kt + 1) * M::BK;
#pragma unroll
        for (int p = 0; p < PA; ++p)
          ra[p] = *reinterpret_cast<const vec_t*>(ga + (int64_t)(32 * p) * lda + ko);
#pragma unroll
        for (int p = 0; p < PB; ++p)
          rb[p] = *reinterpret_cast<const vec_t*>(gb + (int64_t)(32 * p) * ldb + ko);
      }
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
// patch-major (patches of PS x PS tiles, row-major inside a patch) and every XCD gets one contiguous
// range of that order: the ~64 workgroups resident on an XCD then work on one patch and share
// 2*PS operand panels out of L2 instead of fetching 2 per tile from the Infinity Cache.
// Placement only changes speed: any dispatch order computes the same tiles.
struct TileMap {
  int tm, tn;        // tile grid
  int lower;         // 1: only tiles with tc <= tr (tm == tn)
  int pm, pn;        // patch grid
  int npatch;        // patches enumerated (lower: pm*(pm+1)/2)
  int grid;          // launch size: npatch * PS*PS rounded up to a multiple of 8
  static constexpr int PS = 8;
  static TileMap make(int64_t tm_, int64_t tn_, int lower_) {
    TileMap t;
    t.tm = (int)tm_; t.tn = (int)tn_; t.lower = lower_;
    t.pm = (t.tm + PS - 1) / PS; t.pn = (t.tn + PS - 1) / PS;
    t.npatch = lower_ ? t.pm * (t.pm + 1) / 2 : t.pm * t.pn;
    t.grid = (t.npatch * PS * PS + 7) / 8 * 8;
    return t;
  }
  __device__ __forceinline__ bool decode(unsigned b, int& tr, int& tc) const {
    const int chunk = grid >> 3;
    const int l = (int)(b & 7) * chunk + (int)(b >> 3);
    const int patch = l / (PS * PS), slot = l % (PS * PS);
    if (patch >= npatch) return false;
    int sr, sc;
    if (lower) {
      tri_decode(patch, sr, sc);
    } else {
      sr = patch / pn;
      sc = patch % pn;
    }
    tr = sr * PS + slot / PS;
    tc = sc * PS + slot % PS;
    return tr < tm && tc < tn && (!lower || tc <= tr);
  }
};

// The 128 x 128 tile every dense kernel of the path uses.  SMN_STAGES (build-time) selects LDS
// double-buffering (2, default) or a single buffer (1).
#ifndef SMN_STAGES
#define SMN_STAGES 2
#endif
template <typename T>
using MainTile = TileNT<T, 128, 128, SMN_STAGES>;
