// cholesky.hip — jittered blocked Cholesky carried through appended rows (partial factorisation).
//
// Replaces lax.linalg.cholesky + triangular_solve (spax/utils.py:179-180), the Cholesky inside
// jax.scipy.stats.multivariate_normal.logpdf (spax/likelihoods.py:27) and the cho_factor /
// cho_solve pair of neural_tangents' gradient_descent_mse_ensemble (spax/kernels.py:29-32).
//
// One engine serves all of them: factor the first n_factor columns of a symmetric matrix whose
// trailing rows hold right-hand sides (y^T, K_td).  On exit those rows hold B L^-T, and the
// trailing block holds the Schur complement C - B A^-1 B^T: predictive covariance, -mean and
// -y^T K^-1 y drop out of the same trailing update that the factorisation needs anyway, so the
// path has no separate triangular-solve kernels.
//
// Right-looking.  Columns are cut three ways: super-panels of S = 1024 columns, outer panels of 256 (512 under the look-ahead) inside them,
// 128-column sub-panels inside those.
//   panel_kernel   one workgroup per block of rows below the diagonal block; every workgroup re-factors the
//                  128x128 diagonal block in LDS and carries its own rows through the same column operations
//                  (a true TRSM, no explicit inverse): 16-column blocks are brought up to date with MFMAs
//                  straight from the LDS image, 8-column micro-panels are factored in registers.
//                  Workgroup 0 stores L_kk (to a side buffer), sum(log pivots) and the info flag.
//   panelr_kernel  the form the factorisation launches (round 3): the micro-panels are replaced by a register-resident
//                  16x16 leaf (panel_leaf.hpp: factor + TRSM of the lane's own row through DPP broadcasts), the block
//                  updates run on all waves behind each leaf, the image streams in beside the first leaves and the
//                  solved columns leave as they finish.  panel_kernel stays for the solve-only sweep of smn_trsm and
//                  for A/B (SMN_PANEL_LEAF=0).
//   update_kernel  C -= A B^T on the f32/f64 MFMA (gemm_nt.hpp), four uses:
//     strip   the 128 columns in front of the second sub-panel (K = 128);
//     near    after an outer panel, the columns of ITS super-panel only (K = 256, a lower trapezoid);
//     far     after a super-panel, everything to its right, once, with K = S (where most of the N^3/3 flops run,
//             at the long K the tile engine likes).  From n_total = 8192 on the far update is split by tile column:
//     F0      the next super-panel's columns, on the caller's (high-priority) stream, and
//     F1      the rest, on a stream whose CU mask leaves 32 CUs alone: the next super-panel's panel chain (135 KB
//             of LDS per workgroup, so it needs whole CUs) runs beside F1 on the CUs F1 cannot occupy.
// (Round 2 tried the alternative -- ONE diagonal workgroup per sub-panel that also inverts L_kk, the rows below as an
// MFMA GEMM against the inverse, and a windowed left-looking order of the block updates -- and measured it slower at
// every setting; and the whole factorisation as ONE persistent launch with tile-granular dependencies (tasks POTRF /
// TRSM / UPDATE, release / acquire hand-offs between workgroups): correct, 17.2 ms against 16.1 ms at N = 16384, because
// its critical path -- POTRF + inverse, TRSM tile, diagonal update, three hand-offs per column -- is twice as long as the
// fused panel's.  profiles/r02_notes.md; their sources are in the git history of round 2.)
//   trail_kernel   persistent form of the near update (one stream of K-steps per workgroup).
// Rows [id0, id1) may be declared an identity block (analytic gradients: cholesky_padded's hint): panel and update
// workgroups whose rows are still structurally zero in the columns at hand leave at once.
#include <climits>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "gemm_nt.hpp"
#include "internal.hpp"
#include "panel_leaf.hpp"

namespace {

constexpr int PB = 128;  // sub-panel width == diagonal block edge == GEMM tile edge
constexpr int MP = 8;    // micro-panel width of the in-LDS factorisation
// Launch-shape thresholds (round-1/2 run-time knobs; the final sweeps were flat around these values: profiles/r02_knob_sweep_final.txt)
constexpr int kQuarterTileMax = 256;   // update launches of at most this many 128x128 tiles use 64x64 tiles
constexpr int kHalfTileMax = 384;      // ... and of at most this many, 64-row tiles
constexpr int kPersistMaxK = 512;      // largest K the persistent trailing kernel takes
constexpr int kPanelSmallRows = 4096;  // f32 panels with at most this many rows below use 16-row workgroups (kPanelSmallXR): the chain-bound
constexpr int kPanelSmallXR = 16;      // tail and small problems.  At most 256 of them, one per CU, each with the least block-update work
                                       // behind its (redundant) factorisation of the diagonal block; batches that fill the chip keep 64-row
                                       // ones (launch_panel).  Against 64-row ones and other thresholds: profiles/r04_panel_rows_ab.txt
constexpr int64_t kSuperWide = 2048;   // super-panel width while at least ctx->super_wide_rows rows are left (profiles/r02_wide_super_panel_sweep.txt)
constexpr int64_t kOuterWide = 512;     // outer-panel width under the look-ahead schedule (256 without)
constexpr int64_t kF0FirstTiles = 2000;   // F1 launches of at most this many tiles start behind F0, not beside it (profiles/r02_f0_first_ab.txt)

template <typename T>
struct PanelCfg;
#ifndef SMN_PANEL_XR
#define SMN_PANEL_XR 128   // build-time: 32 shrinks the panel workgroup to 85 KB of LDS (look-ahead co-residency)
#endif
template <>
struct PanelCfg<float> {
  static constexpr int XR = SMN_PANEL_XR;                     // appended rows per workgroup (multiple of 32)
  static constexpr int LD = PB + 4;    // row stride (elements): 16-byte aligned rows, b128 reads conflict-free
};
template <>
struct PanelCfg<double> {
  static constexpr int XR = 16;        // the 128 x 130 f64 diagonal block already takes 133 KB of LDS
  static constexpr int LD = PB + 2;
};
// 16x16 MFMA tiles for the in-LDS block updates of the panel (operands read straight from the row-major LDS
// image with one 16-byte read per lane; the K index is permuted identically for both operands).
template <typename T>
struct PanelMma;
template <>
struct PanelMma<float> {   // v_mfma_f32_16x16x4_f32: lane = (row | col) + 16 * k-group
  static constexpr int TM = 16, ACC = 4, KSTEP = 16, NK = 4;
  using acc_t = f32x4;
  using vec_t = f32x4;
  static __device__ __forceinline__ int frag_row(int lane) { return lane & 15; }
  static __device__ __forceinline__ int frag_k(int lane) { return (lane >> 4) * 4; }
  static __device__ __forceinline__ void mma1(acc_t& c, float a, float b) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int acc_row(int lane, int i) { return 4 * (lane >> 4) + i; }
  static __device__ __forceinline__ int acc_col(int lane) { return lane & 15; }
};
template <>
struct PanelMma<double> {  // v_mfma_f64_16x16x4_f64
  static constexpr int TM = 16, ACC = 4, KSTEP = 8, NK = 2;
  using acc_t = f64x4;
  using vec_t = f64x2;
  static __device__ __forceinline__ int frag_row(int lane) { return lane & 15; }
  static __device__ __forceinline__ int frag_k(int lane) { return (lane >> 4) * 2; }
  static __device__ __forceinline__ void mma1(acc_t& c, double a, double b) {
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int acc_row(int lane, int i) { return (lane >> 4) + 4 * i; }
  static __device__ __forceinline__ int acc_col(int lane) { return lane & 15; }
};

template <typename T>
constexpr size_t panel_lds_bytes(int xr = PanelCfg<T>::XR) {
#ifdef SMN_PANEL_TIMING
  return sizeof(T) * ((size_t)(PB + xr) * PanelCfg<T>::LD + MP * MP + PB) + 2048;   // room for panelr_kernel's timeline
#else
  return sizeof(T) * ((size_t)(PB + xr) * PanelCfg<T>::LD + MP * MP + PB);
#endif
}
constexpr int panel_threads(int xr) { return (PB + xr + 63) / 64 * 64; }   // one thread per LDS row, whole waves
// panelr_kernel: the image + one private copy of the first diagonal tile per row wave (+ the timeline of the timing build)
constexpr int kLeafTileLd(size_t elem) { return 16 + 16 / (int)elem; }
template <typename T>
constexpr size_t panelr_lds_bytes(int xr = PanelCfg<T>::XR) {
  return sizeof(T) * ((size_t)(PB + xr) * PanelCfg<T>::LD + (size_t)(panel_threads(xr) / 64) * 16 * kLeafTileLd(sizeof(T)) + PB)   // (+ the PB reciprocal pivots)
#ifdef SMN_PANEL_TIMING
         + 4096
#endif
      ;
}

#ifdef SMN_PANEL_TIMING   // debug build only: phase times of workgroup 0 of the first panel, printed by the kernel
#define PT_DECL long long pt_t = wall_clock64(), pt_acc[6] = {0, 0, 0, 0, 0, 0}; (void)pt_acc
#define PT_MARK(i) do { const long long n_ = wall_clock64(); pt_acc[i] += n_ - pt_t; pt_t = n_; } while (0)
#else
#define PT_DECL
#define PT_MARK(i)
#endif

__device__ __forceinline__ float rsqrt_t(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ double rsqrt_t(double x) { return 1.0 / sqrt(x); }

// Fused POTRF + TRSM of one 128-column sub-panel, entirely in LDS / registers.
//   a: matrix base, j0: first column of the sub-panel, rows [j0, j0+128) are the diagonal block,
//   row blocks of XR rows follow from `rbeg` up to n_total; one workgroup per row block, and every
//   workgroup re-factors the diagonal block (redundant, but it removes the potrf -> trsm launch
//   dependency and needs no explicit inverse).
// Algorithm: left-looking over micro-panels of MP = 8 columns, one thread per row.
//   0. at every 16-column boundary the block's columns are brought up to date with all finished columns by
//      MFMAs that read both operands from the LDS image (16x16 tiles, 4 row tiles per wave);
//   1. each thread pulls its 8 entries into registers and subtracts the contribution of the (at most 8)
//      finished columns of the current 16-column block (16-byte LDS reads, pivot rows broadcast);
//   2. the 8 pivot rows publish their updated 8x8 diagonal micro-block; barrier;
//   3. every thread factors that 8x8 block redundantly in registers and runs the 8-step
//      triangular solve on its own 8 values (for a pivot row this reproduces its row of L,
//      diagonal included: d * rsqrt(d) = sqrt(d)); writes them back; barrier.
// 2 barriers per micro-panel (32 per sub-panel, + 7 for the MFMA blocks) instead of 2 per column.
// (Round 2's panelh_kernel -- these row threads with the MFMA block updates on a second set of waves one block ahead,
// 27 us per sub-panel against 29 -- is superseded by panelr_kernel below and gone from the source: profiles/r02_panel_helpers_ab.txt.)
// prefactored != 0: the diagonal block already holds L (solve only; used by smn_trsm).
// XRV = rows below the diagonal block carried per workgroup.  The default (128 in f32) minimises the number of
// workgroups that each redo the diagonal factorisation; the f32 16-row form (kPanelSmallXR) is launched when the panel has few
// rows anyway (late, chain-bound super-panels; small problems): the MFMA block updates of a workgroup shrink by half.
template <typename T, int XRV = PanelCfg<T>::XR>
__global__ void __launch_bounds__(panel_threads(XRV)) panel_kernel(T* __restrict__ a, int64_t lda, int64_t j0,
                                                                     int64_t rbeg, int64_t n_total, int prefactored,
                                                                     double* __restrict__ logdet,
                                                                     int* __restrict__ info, T* __restrict__ ldiag_out,
                                                                     int64_t id0, int64_t id1, int64_t bstride, int64_t lstride) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int XR = XRV, NT = panel_threads(XRV), LD = PanelCfg<T>::LD;
  if (bstride) {   // batched factorisation: problem blockIdx.y has its own matrix, scalars and side buffer
    a += (int64_t)blockIdx.y * bstride; logdet += blockIdx.y; info += blockIdx.y;
    if (ldiag_out) ldiag_out += (int64_t)blockIdx.y * lstride;
  }
  constexpr int VEC = 16 / sizeof(T);
  using vec_t = typename Mfma<T>::vec_t;
  T* S = reinterpret_cast<T*>(smem);        // [PB + XR][LD]
  T* blk = S + (PB + XR) * LD;              // [MP][MP] staging of the diagonal micro-block
  T* piv = blk + MP * MP;                   // [PB] pivots d_j = L_jj^2 (for logdet / info)
  const int tid = threadIdx.x;
  const int64_t rb = rbeg + (int64_t)blockIdx.x * XR;  // first appended row of this workgroup
  // identity rows that start right of this sub-panel are zero in its columns and stay zero
  if (id0 >= 0 && rb >= id0 && rb + XR <= id1 && rb - id0 >= j0 + PB) return;
  const int nx = (int)max((int64_t)0, min((int64_t)XR, n_total - rb));
  // global -> LDS in 16-byte pieces (rows are 16-byte aligned on both sides).  SI loads per thread are in flight
  // at once, and the first round of the appended rows is issued before the diagonal block's LDS writes.
  constexpr int RV = PB / VEC;  // vectors per row
  constexpr int SI = 16;
  PT_DECL;
  auto gload = [&](vec_t (&tmp)[SI], int base, int nvec, int64_t grow0) {
#pragma unroll
    for (int u = 0; u < SI; ++u) {
      const int idx = base + u * NT + tid;
      if (idx < nvec) tmp[u] = *reinterpret_cast<const vec_t*>(&a[(grow0 + idx / RV) * lda + j0 + (idx % RV) * VEC]);
    }
  };
  auto lstore = [&](const vec_t (&tmp)[SI], int base, int nvec, int lrow0) {
#pragma unroll
    for (int u = 0; u < SI; ++u) {
      const int idx = base + u * NT + tid;
      if (idx < nvec) *reinterpret_cast<vec_t*>(&S[(lrow0 + idx / RV) * LD + (idx % RV) * VEC]) = tmp[u];
    }
  };
  {
    const int nd = PB * RV, nxv = nx * RV;
    vec_t ta[SI], tb[SI];
    gload(tb, 0, nxv, rb);
    for (int base = 0; base < nd; base += NT * SI) {
      gload(ta, base, nd, j0);
      lstore(ta, base, nd, 0);
    }
    lstore(tb, 0, nxv, PB);
    for (int base = NT * SI; base < nxv; base += NT * SI) {
      gload(tb, base, nxv, rb);
      lstore(tb, base, nxv, PB);
    }
  }
  __syncthreads();
  PT_MARK(0);

  const int row = tid;
  const bool active = row < PB + nx && !(prefactored && row < PB);
  using M = PanelMma<T>;
  constexpr int CB = 16;                               // column block brought up to date on the MFMA
  constexpr int NW = NT / 64;                          // waves
  constexpr int RT = (PB + XR) / M::TM;                // 16-row tiles of the LDS image
  constexpr int TPW = (RT + NW - 1) / NW;              // row tiles per wave: independent accumulators, one B fragment
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = M::frag_row(lane), fk = M::frag_k(lane);
  for (int c0 = 0; c0 < PB; c0 += MP) {
    const int cb = c0 & ~(CB - 1);                     // first column of the current 16-column block
    if (c0 == cb && cb > 0) {
      // S[rows >= rmin, cb:cb+16] -= S[rows, 0:cb] * S[cb:cb+16, 0:cb]^T   (MFMA, operands straight from S)
      const int rmin = prefactored ? PB : cb;
      // Tiles wholly above rmin are finished rows: wave w skips its first u0 tiles.  The skip count is wave-uniform
      // and fixed for the whole K loop, so each count gets its own straight-line instantiation (no branches
      // between the MFMAs, accumulators stay in place).
      static_assert(RT % NW == 0, "every wave owns the same number of 16-row tiles");
      const int first = rmin / M::TM;                                   // first tile that still needs the update
      const int u0 = first <= wave ? 0 : (first - wave + NW - 1) / NW;  // tiles of this wave to skip
      auto block_update = [&](auto u0c) {
        constexpr int U0 = decltype(u0c)::value;
        if constexpr (U0 < TPW) {
          typename M::acc_t acc[TPW];
#pragma unroll
          for (int u = U0; u < TPW; ++u) {
            const int rt = (wave + u * NW) * M::TM;
#pragma unroll
            for (int i = 0; i < M::ACC; ++i) acc[u][i] = -S[(rt + M::acc_row(lane, i)) * LD + cb + M::acc_col(lane)];
          }
          const T* pb = &S[(cb + fr) * LD + fk];
          const T* pa = &S[(wave * M::TM + fr) * LD + fk];
          for (int kb = 0; kb < cb; kb += M::KSTEP) {
            const typename M::vec_t bv = *reinterpret_cast<const typename M::vec_t*>(pb + kb);
            typename M::vec_t av[TPW];
#pragma unroll
            for (int u = U0; u < TPW; ++u)
              av[u] = *reinterpret_cast<const typename M::vec_t*>(pa + u * NW * M::TM * LD + kb);
            // k-slice outer, tile inner: consecutive MFMAs hit different accumulators
#pragma unroll
            for (int i = 0; i < M::NK; ++i)
#pragma unroll
              for (int u = U0; u < TPW; ++u) M::mma1(acc[u], av[u][i], bv[i]);
          }
#pragma unroll
          for (int u = U0; u < TPW; ++u) {
            const int rt = (wave + u * NW) * M::TM;
#pragma unroll
            for (int i = 0; i < M::ACC; ++i) S[(rt + M::acc_row(lane, i)) * LD + cb + M::acc_col(lane)] = -acc[u][i];
          }
        }
      };
      static_assert(TPW <= 4, "dispatch below covers up to 4 tiles per wave");
      switch (u0) {
        case 0: block_update(std::integral_constant<int, 0>{}); break;
        case 1: block_update(std::integral_constant<int, 1>{}); break;
        case 2: block_update(std::integral_constant<int, 2>{}); break;
        case 3: block_update(std::integral_constant<int, 3>{}); break;
        default: break;
      }
      __syncthreads();
      PT_MARK(1);
    }
    const bool work = active && row >= c0;
    T v[MP];
    if (work) {
#pragma unroll
      for (int q = 0; q < MP; q += VEC) {
        const vec_t t = *reinterpret_cast<const vec_t*>(&S[row * LD + c0 + q]);
#pragma unroll
        for (int e = 0; e < VEC; ++e) v[q + e] = t[e];
      }
      for (int k = cb; k < c0; k += VEC) {   // columns left of cb were folded in by the MFMA block update
        const vec_t av = *reinterpret_cast<const vec_t*>(&S[row * LD + k]);
#pragma unroll
        for (int q = 0; q < MP; ++q) {
          const vec_t bv = *reinterpret_cast<const vec_t*>(&S[(c0 + q) * LD + k]);
#pragma unroll
          for (int e = 0; e < VEC; ++e) v[q] = fma(-av[e], bv[e], v[q]);
        }
      }
      if (!prefactored && row < c0 + MP) {
#pragma unroll
        for (int q = 0; q < MP; ++q) blk[(row - c0) * MP + q] = v[q];
      }
    }
    __syncthreads();
    PT_MARK(2);
    if (work) {
      T lm[MP][MP], rinv[MP];
#pragma unroll
      for (int i = 0; i < MP; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) lm[i][j] = prefactored ? S[(c0 + i) * LD + c0 + j] : blk[i * MP + j];
      if (prefactored) {
#pragma unroll
        for (int j = 0; j < MP; ++j) rinv[j] = T(1) / lm[j][j];
      } else {
#pragma unroll
        for (int j = 0; j < MP; ++j) {
          const T d = lm[j][j];
          if (row == PB - 1) piv[c0 + j] = d;   // the last diagonal row takes part in every micro-panel
          rinv[j] = rsqrt_t(d);
#pragma unroll
          for (int i = j + 1; i < MP; ++i) lm[i][j] *= rinv[j];
#pragma unroll
          for (int i = j + 1; i < MP; ++i)
#pragma unroll
            for (int jj = j + 1; jj <= i; ++jj) lm[i][jj] = fma(-lm[i][j], lm[jj][j], lm[i][jj]);
        }
      }
#pragma unroll
      for (int j = 0; j < MP; ++j) {
        T x = v[j];
#pragma unroll
        for (int jj = 0; jj < j; ++jj) x = fma(-v[jj], lm[j][jj], x);
        v[j] = x * rinv[j];
      }
#pragma unroll
      for (int q = 0; q < MP; q += VEC) {
        vec_t t;
#pragma unroll
        for (int e = 0; e < VEC; ++e) t[e] = v[q + e];
        *reinterpret_cast<vec_t*>(&S[row * LD + c0 + q]) = t;
      }
    }
    __syncthreads();
    PT_MARK(3);
  }

  if (blockIdx.x == 0 && !prefactored) {
    // logdet += sum_j log d_j, info = first non-positive pivot (one log per thread, not per pivot per thread)
    if (tid < 64) {   // one wave, two pivots per lane: a single deterministic atomic per sub-panel
      const T d0 = piv[tid], d1 = piv[tid + 64];
      double lg = log((double)d0) + log((double)d1);
      int bad = !(d0 > T(0)) ? tid : (!(d1 > T(0)) ? tid + 64 : INT_MAX);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        lg += __shfl_xor(lg, o);
        bad = min(bad, __shfl_xor(bad, o));
      }
      if (tid == 0) {
        atomicAdd(logdet, lg);
        if (bad != INT_MAX) atomicMin(info, (int)(j0 + bad + 1));
      }
    }
    // L_kk goes to a side buffer, NOT in place: the other workgroups of this launch still read the
    // un-factored A_kk, and they may start after this one has finished (grid larger than the chip, or
    // CUs shared with the trailing update on the other stream).  copy_diag_kernel moves it home at the end.
    for (int idx = tid; idx < PB * RV; idx += NT) {
      const int r = idx / RV, c = (idx % RV) * VEC;
      *reinterpret_cast<vec_t*>(&ldiag_out[r * PB + c]) = *reinterpret_cast<const vec_t*>(&S[r * LD + c]);
    }
  }
  for (int idx = tid; idx < nx * RV; idx += NT) {
    const int r = idx / RV, c = (idx % RV) * VEC;
    *reinterpret_cast<vec_t*>(&a[(rb + r) * lda + j0 + c]) = *reinterpret_cast<const vec_t*>(&S[(PB + r) * LD + c]);
  }
#ifdef SMN_PANEL_TIMING
  __syncthreads();
  PT_MARK(4);
  if (tid == 0 && j0 == 0 && (blockIdx.x == 0 || blockIdx.x == 5))
    printf("panel wg%d (100 MHz ticks): stage_in %lld  mfma_blocks %lld  dots %lld  factor+solve %lld  store %lld\n",
           (int)blockIdx.x, pt_acc[0], pt_acc[1], pt_acc[2], pt_acc[3], pt_acc[4]);
#endif
}

// panelr_kernel — the form the factorisation launches from round 3 on: the 8-column in-LDS micro-panels are replaced by
// a register-resident 16x16 leaf (panel_leaf.hpp).  Per 16-column block b (columns cb .. cb+15, cn = cb + 16):
//   Row threads (waves [0, NW), one LDS row each) load the lane's own 16 entries and a copy of diagonal-tile row
//   (lane & 15), run the leaf (factor + TRSM of the own row through DPP row_newbcast: no LDS traffic, no barrier inside
//   the block) and write the solved entries back to the LDS image.                                     -- barrier A --
//   They then send the same registers straight to global memory (the appended rows' columns of this block are final:
//   nothing is left to store when the last block is done) while ALL waves bring block b+1's columns up to date with
//   every finished column (at most two 16x16 tiles per wave, K = cn).                                  -- barrier B --
//   Helper waves ([NW, 2 NW)) stream the raw columns of the blocks to come from global memory into LDS beside the leaf
//   (three chunks in flight; the row threads take block 0 straight into registers, so the first leaf starts after ONE
//   global round trip instead of after the whole 128 KB image) and are the other half of the MFMA phase.
// Two workgroup barriers per block instead of panelh_kernel's five.  One wave issues a vector instruction every ~6.5
// cycles (profiles/r03_valu_issue_rate.txt), so the leaf's 288 are ~0.85 us per block and half of the sub-panel.
template <typename T, int XRV = PanelCfg<T>::XR>
__global__ void __launch_bounds__(2 * panel_threads(XRV)) panelr_kernel(T* __restrict__ a, int64_t lda, int64_t j0,
                                                                          int64_t rbeg, int64_t n_total,
                                                                          double* __restrict__ logdet,
                                                                          int* __restrict__ info, T* __restrict__ ldiag_out,
                                                                          int64_t id0, int64_t id1, int64_t bstride, int64_t lstride, int passes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int XR = XRV, NTV = panel_threads(XRV), LD = PanelCfg<T>::LD;
  if (bstride) {   // batched factorisation: problem blockIdx.y has its own matrix, scalars and side buffer
    a += (int64_t)blockIdx.y * bstride; logdet += blockIdx.y; info += blockIdx.y;
    if (ldiag_out) ldiag_out += (int64_t)blockIdx.y * lstride;
  }
  constexpr int VEC = 16 / sizeof(T);
  constexpr int CB = 16, NB = PB / CB, VPC = CB / VEC, ROWS = PB + XR;
  using vec_t = typename Mfma<T>::vec_t;
  using M = PanelMma<T>;
  constexpr int NW = NTV / 64;
  constexpr int RT = ROWS / M::TM;
  static_assert((ROWS * VPC) % NTV == 0, "chunk vectors split evenly over the helper threads");
  static_assert(RT - 1 <= 2 * 2 * NW, "update() gives every wave at most two tiles");
  T* S = reinterpret_cast<T*>(smem);        // [PB + XR][LD]
  const int tid = threadIdx.x;
  // `passes` > 1 (launch_panel_x: more row groups than CUs): this workgroup carries `passes` consecutive groups of XR rows,
  // the first one through the loop below, the others through the v-steps alone (further down)
  const int64_t rb = rbeg + (int64_t)blockIdx.x * passes * XR;
  if (id0 >= 0 && rb >= id0 && rb + XR <= id1 && rb - id0 >= j0 + PB) return;   // (identity hints: passes == 1)
  const int nx = (int)max((int64_t)0, min((int64_t)XR, n_total - rb));
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = M::frag_row(lane), fk = M::frag_k(lane);
  // Block b+1's columns brought up to date with every finished column: S[t, cn:cn+16] -= S[t, 0:cn] S[cn:cn+16, 0:cn]^T
  // for the tiles t from `first` down, dealt round-robin to all 2 NW waves (at most two per wave: two independent
  // accumulators, one B fragment).  It runs BEHIND the leaf, not beside it: f32 MFMAs and vector instructions of two
  // waves on one SIMD do not overlap (each MFMA of a partner wave costs the leaf its full 32 cycles,
  // profiles/r03_leaf_probe.txt), so a helper pass beside the leaf only stretches the critical path.
  auto update = [&](int cn, int first) {
    int rt[2];
    bool on[2];
    // Second tiles go to the waves that are alone on their SIMD first (waves w and w + 4 share one: of six waves, 2 and 3
    // are alone), so that no SIMD carries three tiles while another carries one.
    const int slot2 = 2 * NW == 6 ? (wave == 2 ? 0 : wave == 3 ? 1 : wave < 2 ? wave + 2 : wave) : wave;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int t = first + (j == 0 ? wave : 2 * NW + slot2);
      on[j] = t < RT;
      rt[j] = (on[j] ? t : first + wave) * M::TM;
    }
    if (!on[0]) return;   // wave-uniform: nothing for this wave (on[1] implies on[0])
    T cv[2][M::ACC];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < M::ACC; ++i) cv[j][i] = S[(rt[j] + M::acc_row(lane, i)) * LD + cn + M::acc_col(lane)];
    typename M::acc_t acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < M::ACC; ++i) acc[j][i] = T(0);
    const T* pb = &S[(cn + fr) * LD + fk];
    const T* pa0 = &S[(rt[0] + fr) * LD + fk];
    const T* pa1 = &S[(rt[1] + fr) * LD + fk];
    // Two fragment sets, ping-pong: the reads of K-step s+1 are issued, THEN the MFMAs of step s (sched_barrier keeps the
    // order; written as a plain prefetch loop hipcc folds it back into read -> wait -> MFMA, 420 cycles per step for 256 of
    // MFMAs).  Reads past the end are clamped to the last step, so no branch sits between the reads and the MFMAs.
    using cvp = const typename M::vec_t*;
    const int klast = cn - M::KSTEP;
    auto kloop = [&](auto twoc) {
      constexpr bool TWO = decltype(twoc)::value;
      typename M::vec_t b0, x0, y0, b1, x1, y1;
      auto rd = [&](typename M::vec_t& bq, typename M::vec_t& xq, typename M::vec_t& yq, int kb) {
        bq = *reinterpret_cast<cvp>(pb + kb);
        xq = *reinterpret_cast<cvp>(pa0 + kb);
        if (TWO) yq = *reinterpret_cast<cvp>(pa1 + kb);
      };
      auto mm = [&](const typename M::vec_t& bq, const typename M::vec_t& xq, const typename M::vec_t& yq) {
#pragma unroll
        for (int i = 0; i < M::NK; ++i) {
          M::mma1(acc[0], xq[i], bq[i]);
          if (TWO) M::mma1(acc[1], yq[i], bq[i]);
        }
      };
      rd(b0, x0, y0, 0);
      for (int kb = 0;;) {
        rd(b1, x1, y1, min(kb + M::KSTEP, klast));
        __builtin_amdgcn_sched_barrier(0);
        mm(b0, x0, y0);
        kb += M::KSTEP;
        if (kb >= cn) break;
        rd(b0, x0, y0, min(kb + M::KSTEP, klast));
        __builtin_amdgcn_sched_barrier(0);
        mm(b1, x1, y1);
        kb += M::KSTEP;
        if (kb >= cn) break;
      }
    };
    if (on[1]) kloop(std::true_type{}); else kloop(std::false_type{});
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (on[j]) {
#pragma unroll
        for (int i = 0; i < M::ACC; ++i) S[(rt[j] + M::acc_row(lane, i)) * LD + cn + M::acc_col(lane)] = cv[j][i] - acc[j][i];
      }
  };
  // Block c's columns are final once its leaf is done: they leave for global memory (the appended rows' entries of L^-T
  // products in place, the factored diagonal block to the side buffer) in the chunk layout -- VPC lanes per row -- by
  // `nthr` threads that have nothing else to do (the helper waves beside the next leaf; everybody after the last one).
  auto store_out = [&](int c, int t0, auto nthrc) {
    // every LDS read first, then every store (a read -> store loop pays one LDS round trip per vector: 1.2 us per block
    // in f64, more than the leaf it is meant to hide behind)
    constexpr int NTHR = decltype(nthrc)::value, IT = (ROWS * VPC + NTHR - 1) / NTHR;
    vec_t buf[IT];
    T* dst[IT];
#pragma unroll
    for (int u = 0; u < IT; ++u) {
      const int idx = u * NTHR + t0, r = idx / VPC, v = idx % VPC;
      dst[u] = nullptr;
      if (idx < ROWS * VPC && r >= c * CB) {
        if (r < PB) {
          if (blockIdx.x == 0 && ldiag_out) dst[u] = ldiag_out + (int64_t)r * PB + c * CB + v * VEC;
        } else if (r - PB < nx) {
          dst[u] = a + (rb + (r - PB)) * lda + j0 + c * CB + v * VEC;
        }
      }
      if (dst[u]) buf[u] = *reinterpret_cast<const vec_t*>(&S[r * LD + c * CB + v * VEC]);
    }
#pragma unroll
    for (int u = 0; u < IT; ++u)
      if (dst[u]) *reinterpret_cast<vec_t*>(dst[u]) = buf[u];
  };
#ifdef SMN_PANEL_TIMING   // timeline of wg 0 of the first panel: ticks (100 MHz) since kernel start, per block, per wave, 6 marks
  int* const tlog = reinterpret_cast<int*>(S + ROWS * LD + NW * CB * kLeafTileLd(sizeof(T)) + PB);
  const long long tl0 = wall_clock64();
#define TL(b, i) do { if (lane == 0 && blockIdx.x == 0 && j0 == 0) tlog[(wave * NB + (b)) * 6 + (i)] = (int)(wall_clock64() - tl0); } while (0)
#else
#define TL(b, i)
#endif
  T* const rinv = S + ROWS * LD + NW * CB * kLeafTileLd(sizeof(T));   // [PB] reciprocal pivots, kept for the later passes
  if (wave >= NW) {
    // ------------------------------------------------------------ helpers: stage-in of blocks 1..
    const int hid = tid - NTV;
    constexpr int CPT = ROWS * VPC / NTV;   // 16-byte vectors of one 16-column chunk per helper thread
    vec_t cbuf[3][CPT];
    auto chunk_load = [&](vec_t (&buf)[CPT], int c) {   // columns [16 c, 16 c + 16) of the rows from 16 c down
#pragma unroll
      for (int u = 0; u < CPT; ++u) {
        const int idx = u * NTV + hid, r = idx / VPC, v = idx % VPC;
        const int64_t grow = r < PB ? j0 + r : rb + (r - PB);
        vec_t t;
#pragma unroll
        for (int e = 0; e < VEC; ++e) t[e] = T(0);
        if (r >= c * CB && (r < PB || r - PB < nx)) t = *reinterpret_cast<const vec_t*>(&a[grow * lda + j0 + c * CB + v * VEC]);
        buf[u] = t;
      }
    };
    auto chunk_store = [&](const vec_t (&buf)[CPT], int c) {
#pragma unroll
      for (int u = 0; u < CPT; ++u) {
        const int idx = u * NTV + hid, r = idx / VPC, v = idx % VPC;
        if (r >= c * CB) *reinterpret_cast<vec_t*>(&S[r * LD + c * CB + v * VEC]) = buf[u];
      }
    };
    // Block b: chunk b+1 (requested three chunks ago) goes to LDS -- its first reader is update() behind barrier A --
    // and chunk b+4 is requested into the same registers.
    auto hblock = [&](int b, auto slotc) {
      constexpr int SLOT = decltype(slotc)::value;
      if (b >= NB) return;
      TL(b, 0);
      if (b + 1 < NB) {
        chunk_store(cbuf[SLOT], b + 1);
        if (b + 4 < NB) chunk_load(cbuf[SLOT], b + 4);
      }
      TL(b, 1);
      if (b > 0) store_out(b - 1, hid, std::integral_constant<int, NTV>{});        // beside leaf b
      TL(b, 2);
      if (b + 1 >= NB) return;
      __syncthreads();                              // A: block b is solved in every row
      TL(b, 3);
      update((b + 1) * CB, b + 1);
      TL(b, 4);
      __syncthreads();                              // B: block b + 1 is up to date
      TL(b, 5);
    };
    chunk_load(cbuf[1], 1);
    chunk_load(cbuf[2], 2);
    chunk_load(cbuf[0], 3);
    for (int b0 = 0; b0 < NB; b0 += 3) {   // chunk b + 1 lives in slot (b + 1) % 3
      hblock(b0, std::integral_constant<int, 1>{});
      hblock(b0 + 1, std::integral_constant<int, 2>{});
      hblock(b0 + 2, std::integral_constant<int, 0>{});
    }
  } else {
    // ------------------------------------------------------------ row threads: one leaf per 16-column block
    const int row = tid, r16 = lane & 15;
    const bool in_s = row < ROWS;
    const int srow = in_s ? row : ROWS - 1;
    T D[CB], V[CB];
    constexpr int TLD = kLeafTileLd(sizeof(T));
    T* const tile0 = S + ROWS * LD + wave * CB * TLD;
    {
      // Block 0 of this wave's own 64 rows and of the diagonal tile: VPC lanes per row (whole 64-byte row pieces per
      // request instead of one 16-byte piece per row and lane: a quarter of the cache-line look-ups, and the tile is
      // fetched once per wave, not once per 16 lanes), through LDS: the rows into the image, the tile into a private
      // copy per wave (another wave may already have written its solved rows over the image's tile).  LDS operations of
      // one wave execute in order, so the wave reads back what it wrote: no barrier.
      vec_t own[VPC], dg[(CB * VPC + 63) / 64];
#pragma unroll
      for (int u = 0; u < VPC; ++u) {
        const int idx = u * 64 + lane, r = wave * 64 + idx / VPC, v = idx % VPC;
        const int64_t gr = r < PB ? j0 + r : rb + (r - PB);
        vec_t t;
#pragma unroll
        for (int e = 0; e < VEC; ++e) t[e] = T(0);
        if (r < ROWS && (r < PB || r - PB < nx)) t = *reinterpret_cast<const vec_t*>(&a[gr * lda + j0 + v * VEC]);
        own[u] = t;
      }
#pragma unroll
      for (int u = 0; u < (CB * VPC + 63) / 64; ++u) {
        const int idx = u * 64 + lane, r = idx / VPC, v = idx % VPC;
        if (idx < CB * VPC) dg[u] = *reinterpret_cast<const vec_t*>(&a[(j0 + r) * lda + j0 + v * VEC]);
      }
#pragma unroll
      for (int u = 0; u < VPC; ++u) {
        const int idx = u * 64 + lane, r = wave * 64 + idx / VPC, v = idx % VPC;
        if (r < ROWS) *reinterpret_cast<vec_t*>(&S[r * LD + v * VEC]) = own[u];
      }
#pragma unroll
      for (int u = 0; u < (CB * VPC + 63) / 64; ++u) {
        const int idx = u * 64 + lane, r = idx / VPC, v = idx % VPC;
        if (idx < CB * VPC) *reinterpret_cast<vec_t*>(&tile0[r * TLD + v * VEC]) = dg[u];
      }
    }
#pragma unroll 1
    for (int b = 0; b < NB; ++b) {
      const int cb = b * CB;
      const bool live = 64 * (wave + 1) > cb;   // wave-uniform: some row of this wave is not finished yet
      if (live) {
#pragma unroll
        for (int q = 0; q < VPC; ++q) {
          const vec_t v = *reinterpret_cast<const vec_t*>(&S[srow * LD + cb + q * VEC]);
          const vec_t d = b == 0 ? *reinterpret_cast<const vec_t*>(&tile0[r16 * TLD + q * VEC])
                                 : *reinterpret_cast<const vec_t*>(&S[(cb + r16) * LD + cb + q * VEC]);
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            V[q * VEC + e] = v[e];
            D[q * VEC + e] = d[e];
          }
        }
        TL(b, 0);
        T R[CB];
        leaf::run(D, V, R);
        TL(b, 1);
        if (passes > 1 && wave == NW - 1 && lane == 0) {   // (the last row wave takes part in every block)
#pragma unroll
          for (int q = 0; q < VPC; ++q) {
            vec_t t;
#pragma unroll
            for (int e = 0; e < VEC; ++e) t[e] = R[q * VEC + e];
            *reinterpret_cast<vec_t*>(&rinv[cb + q * VEC]) = t;
          }
        }
        if (in_s && row >= cb) {
#pragma unroll
          for (int q = 0; q < VPC; ++q) {
            vec_t t;
#pragma unroll
            for (int e = 0; e < VEC; ++e) t[e] = V[q * VEC + e];
            *reinterpret_cast<vec_t*>(&S[row * LD + cb + q * VEC]) = t;
          }
        }
        TL(b, 2);
      }
      if (b + 1 < NB) __syncthreads();   // A
      TL(b, 3);
      if (b + 1 < NB) {
        update(cb + CB, b + 1);
        TL(b, 4);
        __syncthreads();   // B
        TL(b, 5);
      }
    }
  }
  __syncthreads();
#ifdef SMN_PANEL_TIMING
  if (tid == 0 && blockIdx.x == 0 && j0 == 0)
    for (int w = 0; w < 2 * NW; ++w)
      for (int b = 0; b < NB; ++b)
        printf("panelr wave %d block %d: %s %5d  %s %5d  %s %5d  A %5d  upd %5d  B %5d\n", w, b, w < NW ? "leaf0" : "chnk0", tlog[(w * NB + b) * 6 + 0],
               w < NW ? "leaf1" : "  -  ", tlog[(w * NB + b) * 6 + 1], w < NW ? "stor" : "chnk", tlog[(w * NB + b) * 6 + 2], tlog[(w * NB + b) * 6 + 3],
               tlog[(w * NB + b) * 6 + 4], tlog[(w * NB + b) * 6 + 5]);
#endif
#undef TL
  store_out(NB - 1, tid, std::integral_constant<int, 2 * NTV>{});
  // Later passes: the image holds L and rinv the reciprocal pivots, so XR more rows at a time ride through the same column
  // operations without the factorisation -- per block the MFMA update of their tiles (K = the finished columns, the same
  // K loop) and the leaf's v-steps alone.  The same instructions on the same operands as a first pass would run: the same
  // bits, whatever `passes` is.
  constexpr int RV = PB / VEC, PRE = (XR * RV + 2 * NTV - 1) / (2 * NTV);
  vec_t pre[PRE];                                                // the NEXT pass's raw rows, in flight while this pass runs
  auto prefetch = [&](int p) {
    const int64_t rbp = rb + (int64_t)p * XR;
#pragma unroll
    for (int u = 0; u < PRE; ++u) {
      const int idx = u * 2 * NTV + tid, r = idx / RV, c = (idx % RV) * VEC;
#pragma unroll
      for (int e = 0; e < VEC; ++e) pre[u][e] = T(0);
      if (idx < XR * RV && rbp + r < n_total) pre[u] = *reinterpret_cast<const vec_t*>(&a[(rbp + r) * lda + j0 + c]);
    }
  };
  if (passes > 1) prefetch(1);
  for (int p = 1; p < passes; ++p) {
    const int64_t rbp = rb + (int64_t)p * XR;
    if (rbp >= n_total) break;                                   // (uniform)
    const int nxp = (int)min((int64_t)XR, n_total - rbp);
    __syncthreads();                                             // the previous pass's rows have left the image
#pragma unroll
    for (int u = 0; u < PRE; ++u) {
      const int idx = u * 2 * NTV + tid, r = idx / RV, c = (idx % RV) * VEC;
      if (idx < XR * RV) *reinterpret_cast<vec_t*>(&S[(PB + r) * LD + c]) = pre[u];
    }
    if (p + 1 < passes) prefetch(p + 1);
    __syncthreads();
#pragma unroll 1
    for (int b = 0; b < NB; ++b) {
      const int cb = b * CB;
      if (b > 0) {
        update(cb, PB / M::TM);                                  // the appended tiles only
        __syncthreads();
      }
      const int row = tid, r16 = lane & 15;
      if (wave < NW && 64 * (wave + 1) > PB) {                   // row waves that hold appended rows
        const int srow = row < ROWS ? row : ROWS - 1;
        T D[CB], V[CB], R[CB];
#pragma unroll
        for (int q = 0; q < VPC; ++q) {
          const vec_t v = *reinterpret_cast<const vec_t*>(&S[srow * LD + cb + q * VEC]);
          const vec_t d = *reinterpret_cast<const vec_t*>(&S[(cb + r16) * LD + cb + q * VEC]);
          const vec_t r = *reinterpret_cast<const vec_t*>(&rinv[cb + q * VEC]);
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            V[q * VEC + e] = v[e];
            D[q * VEC + e] = d[e];
            R[q * VEC + e] = r[e];
          }
        }
        leaf::solve(D, R, V);
        if (row >= PB && row < ROWS) {
#pragma unroll
          for (int q = 0; q < VPC; ++q) {
            vec_t t;
#pragma unroll
            for (int e = 0; e < VEC; ++e) t[e] = V[q * VEC + e];
            *reinterpret_cast<vec_t*>(&S[row * LD + cb + q * VEC]) = t;
          }
        }
      }
      __syncthreads();
    }
    for (int idx = tid; idx < XR * RV; idx += 2 * NTV) {
      const int r = idx / RV, c = (idx % RV) * VEC;
      if (r < nxp) *reinterpret_cast<vec_t*>(&a[(rbp + r) * lda + j0 + c]) = *reinterpret_cast<const vec_t*>(&S[(PB + r) * LD + c]);
    }
  }
  if (blockIdx.x == 0 && tid < 64) {
    // logdet += 2 sum_j log L_jj; info = first pivot that is not a positive number (d <= 0 came out of the leaf as NaN)
    const T d0 = S[tid * LD + tid], d1 = S[(tid + 64) * LD + tid + 64];
    double lg = 2.0 * (log((double)d0) + log((double)d1));
    int bad = !(d0 > T(0)) ? tid : (!(d1 > T(0)) ? tid + 64 : INT_MAX);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      lg += __shfl_xor(lg, o);
      bad = min(bad, __shfl_xor(bad, o));
    }
    if (tid == 0) {
      atomicAdd(logdet, lg);
      if (bad != INT_MAX) atomicMin(info, (int)(j0 + bad + 1));
    }
  }
}

// Lower triangles of ALL factored diagonal blocks: side buffer -> matrix, once, after the last panel
// (nothing inside the factorisation reads L_kk again; the LML / predictive heads never need it).
template <typename T>
__global__ void copy_diag_kernel(T* __restrict__ a, int64_t lda, const T* __restrict__ ldiag) {
  const int64_t j0 = (int64_t)blockIdx.x * PB;
  const T* src = ldiag + (int64_t)blockIdx.x * PB * PB;
  for (int idx = threadIdx.x; idx < PB * PB; idx += blockDim.x) {
    const int r = idx / PB, c = idx % PB;
    if (c <= r) a[(j0 + r) * lda + j0 + c] = src[idx];
  }
}

// C[r0 + ., c0 + .] -= A_rows * B_rows^T over K columns starting at column k0 of the same matrix.
//   A_rows = a[r0 + tile_r*128 ..., k0 : k0+K],  B_rows = a[c0 + tile_c*128 ..., k0 : k0+K]
// lower: 0 = rectangle tiles_m x tiles_n; 1 = lower triangle (tc <= tr, tiles_m == tiles_n, r0 == c0);
//        2 = lower trapezoid: tiles_n tile columns, all rows from the diagonal down (triangle first, then
//            the rectangle under it), r0 == c0.
template <typename T>
struct UpdArgs {
  T* a; int64_t lda; int64_t r0, c0, k0; int K; int tiles_n; int lower;
  int use_map; TileMap map;   // XCD-aware patch order (gemm_nt.hpp) instead of the linear one
  int64_t bstride;            // batched factorisation: problem blockIdx.y works on a + blockIdx.y * bstride (0: one problem)
};

template <typename T>
__device__ __forceinline__ void upd_decode(const UpdArgs<T>& u, int tl, int& tr, int& tc) {
  if (u.lower == 1) {
    tri_decode(tl, tr, tc);
  } else if (u.lower == 2) {
    const int ntri = u.tiles_n * (u.tiles_n + 1) / 2;
    if (tl < ntri) {
      tri_decode(tl, tr, tc);
    } else {
      tr = u.tiles_n + (tl - ntri) / u.tiles_n;
      tc = (tl - ntri) % u.tiles_n;
    }
  } else {
    tr = tl / u.tiles_n;
    tc = tl % u.tiles_n;
  }
}

// The same enumerations with 64-row tiles (128 columns): tile row r64 of the lower triangle owns the column tiles
// 0 .. r64/2, so tile-row PAIR t holds 2 (t + 1) tiles and starts at linear index t (t + 1).
template <typename T>
__device__ __forceinline__ void upd_decode_half(const UpdArgs<T>& u, int tl, int& r64, int& tc) {
  auto tri = [](int l, int& r, int& c) {
    int t = (int)((sqrt(4.0 * (double)l + 1.0) - 1.0) * 0.5);
    while ((t + 1) * (t + 2) <= l) ++t;
    while (t * (t + 1) > l) --t;
    const int rem = l - t * (t + 1);
    r = 2 * t + rem / (t + 1);
    c = rem % (t + 1);
  };
  if (u.lower == 1) {
    tri(tl, r64, tc);
  } else if (u.lower == 2) {
    const int ntri = u.tiles_n * (u.tiles_n + 1);
    if (tl < ntri) {
      tri(tl, r64, tc);
    } else {
      r64 = 2 * u.tiles_n + (tl - ntri) / u.tiles_n;
      tc = (tl - ntri) % u.tiles_n;
    }
  } else {
    r64 = tl / u.tiles_n;
    tc = tl % u.tiles_n;
  }
}

// TAG only separates the two uses into two symbols (0: strip update, 1: trailing update) so that
// rocprofv3 --stats reports them on separate lines.
// BM = 128: the 128x128 tile of every large launch.  BM = 64: half-height tiles for launches too small to fill the
// chip with 128x128 ones (the near / F0 updates of the late, chain-bound super-panels): twice the workgroups, each
// done in half the time.
// BN = 64 (with BM = 64): quarter tiles for the smallest launches (strips, and the near / F0 updates once fewer
// than ~a CU's worth of 128x128 tiles per CU is left): four workgroups per 128x128 tile, a quarter of the time each.
// The accumulation order over K of an element does not depend on the tile shape, so every form gives the same bits.
template <typename T, int TAG, int BM = kTile, int BN = kTile>
__global__ void __launch_bounds__(256, BN == 64 ? 4 : (BM == 64 ? (sizeof(T) == 8 ? 2 : 3) : 2)) update_kernel(UpdArgs<T> u) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u.a += (int64_t)blockIdx.y * u.bstride;
  using Tile = TileNT<T, BM, BN, SMN_STAGES>;
  using M = typename Tile::M;
  static_assert(BN == kTile || (BM == 64 && BN == 64), "tile shapes: 128x128, 64x128, 64x64");
  int tr, tc;
  int64_t qr = 0, qc = 0;   // quarter-tile offset inside the 128x128 tile
  if (BN == 64) {
    upd_decode(u, (int)(blockIdx.x >> 2), tr, tc);
    qr = (blockIdx.x & 2) ? 64 : 0;
    qc = (blockIdx.x & 1) ? 64 : 0;
    tr *= 2;                // row0 below multiplies by BM = 64
  } else if (BM == 64) {
    upd_decode_half(u, (int)blockIdx.x, tr, tc);
  } else if (u.use_map) {
    if (!u.map.decode(blockIdx.x, tr, tc)) return;   // padding slot of a patch (uniform per workgroup)
  } else {
    upd_decode(u, (int)blockIdx.x, tr, tc);
  }
  const int64_t row0 = u.r0 + (int64_t)tr * BM + qr, col0 = u.c0 + (int64_t)tc * kTile + qc;
  Tile t;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  // acc starts at -C so the C read is in flight together with the first operand loads and the
  // epilogue is a pure store: acc = -C + A B^T, C_new = -acc.
#pragma unroll
  for (int m = 0; m < Tile::MT; ++m)
#pragma unroll
    for (int n = 0; n < Tile::NT; ++n)
#pragma unroll
      for (int i = 0; i < M::ACC; ++i) {
        const int64_t gr = row0 + wr * Tile::WM + m * M::TM + M::acc_row(lane, i);
        const int64_t gc = col0 + wc * Tile::WN + n * M::TN + M::acc_col(lane);
        t.acc[m][n][i] = -u.a[gr * u.lda + gc];
      }
  // trailing updates (TAG 1) run K = 256 ... 1024: the pipelined K loop; strips (K = 128) the plain one
  t.template mainloop<TAG == 1 ? 1 : 0>(u.a + row0 * u.lda + u.k0, u.lda, u.a + col0 * u.lda + u.k0, u.lda, u.K, smem);
#pragma unroll
  for (int m = 0; m < Tile::MT; ++m)
#pragma unroll
    for (int n = 0; n < Tile::NT; ++n)
#pragma unroll
      for (int i = 0; i < M::ACC; ++i) {
        const int64_t gr = row0 + wr * Tile::WM + m * M::TM + M::acc_row(lane, i);
        const int64_t gc = col0 + wc * Tile::WN + n * M::TN + M::acc_col(lane);
        u.a[gr * u.lda + gc] = -t.acc[m][n][i];
      }
}

// Persistent form of the trailing update (lower tiles, f32).  With K = 256 a tile is only 8 K-steps,
// so in the one-tile-per-workgroup kernel above the C read, the first operand fetch, the store and the
// workgroup turn-around are ~30 % of a tile's life.  Here a workgroup walks tiles blockIdx.x,
// blockIdx.x + gridDim.x, ... as ONE continuous stream of K-steps: the operands of the next tile's
// first step and its C block (into a second accumulator-shaped register set) are fetched during the
// current tile's last step, so the MFMA stream never waits at a tile boundary.
template <typename T>
__global__ void __launch_bounds__(256, 2) trail_kernel(UpdArgs<T> u, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using Tile = TileNT<T, kTile, kTile, 2>;
  using M = typename Tile::M;
  using vec_t = typename Tile::vec_t;
  using acc_t = typename Tile::acc_t;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int lrow = tid >> 3, lchunk = tid & 7;
  const int wpos = lrow * Tile::ROWB + ((lchunk ^ ((lrow >> 1) & 7)) << 4);
  const int nk = u.K / M::BK;
  const int G = gridDim.x;
  const int64_t lda = u.lda;
  T* const a = u.a;
  Tile t;
  vec_t ra[Tile::PA], rb[Tile::PB];

  auto origin = [&](int tl, int64_t& row0, int64_t& col0) {
    int tr, tc;
    upd_decode(u, tl, tr, tc);
    row0 = u.r0 + (int64_t)tr * kTile;
    col0 = u.c0 + (int64_t)tc * kTile;
  };
  auto gload = [&](int64_t row0, int64_t col0, int kt) {
    const T* ga = a + (row0 + lrow) * lda + u.k0 + kt * M::BK + lchunk * M::VEC;
    const T* gb = a + (col0 + lrow) * lda + u.k0 + kt * M::BK + lchunk * M::VEC;
#pragma unroll
    for (int p = 0; p < Tile::PA; ++p) ra[p] = *reinterpret_cast<const vec_t*>(ga + (int64_t)(32 * p) * lda);
#pragma unroll
    for (int p = 0; p < Tile::PB; ++p) rb[p] = *reinterpret_cast<const vec_t*>(gb + (int64_t)(32 * p) * lda);
  };
  auto swrite = [&](int buf) {
    char* st = smem + buf * Tile::STAGE;
#pragma unroll
    for (int p = 0; p < Tile::PA; ++p) *reinterpret_cast<vec_t*>(st + wpos + 32 * p * Tile::ROWB) = ra[p];
#pragma unroll
    for (int p = 0; p < Tile::PB; ++p) *reinterpret_cast<vec_t*>(st + Tile::A_BYTES + wpos + 32 * p * Tile::ROWB) = rb[p];
  };
  auto cload = [&](int64_t row0, int64_t col0, acc_t (&dst)[Tile::MT][Tile::NT]) {   // dst = -C
#pragma unroll
    for (int m = 0; m < Tile::MT; ++m)
#pragma unroll
      for (int n = 0; n < Tile::NT; ++n)
#pragma unroll
        for (int i = 0; i < M::ACC; ++i)
          dst[m][n][i] = -a[(row0 + wr * Tile::WM + m * M::TM + M::acc_row(lane, i)) * lda + col0 + wc * Tile::WN +
                            n * M::TN + M::acc_col(lane)];
  };

  int tile = blockIdx.x;
  if (tile >= ntiles) return;
  int64_t row0, col0;
  origin(tile, row0, col0);
  t.zero();
  gload(row0, col0, 0);
  swrite(0);
  __syncthreads();
  int cur = 0;
  while (true) {
    const int nxt = tile + G;
    const bool has_next = nxt < ntiles;
    int64_t nrow0 = 0, ncol0 = 0;
    if (has_next) origin(nxt, nrow0, ncol0);
    for (int kt = 0; kt + 1 < nk; ++kt) {
      gload(row0, col0, kt + 1);
      t.compute(smem + cur * Tile::STAGE, lane, wr, wc);
      swrite(cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
    // last K-step (peeled so that `cn` is live only from here to the store below)
    if (has_next) gload(nrow0, ncol0, 0);     // the next tile's first K-step rides under this tile's last one
    t.compute(smem + cur * Tile::STAGE, lane, wr, wc);
    if (has_next) swrite(cur ^ 1);
    __builtin_amdgcn_sched_barrier(0);        // keep the C loads BELOW the MFMAs and the staging writes:
    acc_t cn[Tile::MT][Tile::NT];             //   staging registers are free again, C (negated) goes in flight
    cload(row0, col0, cn);                    //   across the barrier; the co-resident workgroup covers the wait
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    cur ^= 1;
    // C_new = C - acc = -(cn + acc)
#pragma unroll
    for (int m = 0; m < Tile::MT; ++m)
#pragma unroll
      for (int n = 0; n < Tile::NT; ++n)
#pragma unroll
        for (int i = 0; i < M::ACC; ++i) {
          a[(row0 + wr * Tile::WM + m * M::TM + M::acc_row(lane, i)) * lda + col0 + wc * Tile::WN + n * M::TN +
            M::acc_col(lane)] = -(cn[m][n][i] + t.acc[m][n][i]);
          t.acc[m][n][i] = T(0);
        }
    if (!has_next) break;
    tile = nxt;
    row0 = nrow0;
    col0 = ncol0;
  }
}

// out[i, j] = - sum_k x[i, k] x[j, k] for j <= i, where row i of x is zero left of column (i / 128) * 128 (x = L^-T as the
// factorisation leaves it in the appended rows of an identity block): one workgroup per lower 128x128 tile, K from the tile
// row's first column to kcols.  Tile rows are walked top down, so the longest K loops start first.
template <typename T>
__global__ void __launch_bounds__(256, 2) syrk_rows_kernel(const T* __restrict__ x, int64_t ldx, int64_t kcols, T* __restrict__ out,
                                                          int64_t ldo, int64_t n) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using Tile = MainTile<T>;
  using M = typename Tile::M;
  int tr, tc;
  tri_decode(blockIdx.x, tr, tc);
  const int64_t row0 = (int64_t)tr * kTile, col0 = (int64_t)tc * kTile, k0 = row0;
  Tile t;
  t.zero();
  t.mainloop(x + row0 * ldx + k0, ldx, x + col0 * ldx + k0, ldx, (int)(kcols - k0), smem);   // (run-time choice of the K loop: K = 128 ... kcols)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
#pragma unroll
  for (int m = 0; m < Tile::MT; ++m)
#pragma unroll
    for (int nn = 0; nn < Tile::NT; ++nn)
#pragma unroll
      for (int i = 0; i < M::ACC; ++i) {
        const int64_t gr = row0 + wr * Tile::WM + m * M::TM + M::acc_row(lane, i);
        const int64_t gc = col0 + wc * Tile::WN + nn * M::TN + M::acc_col(lane);
        if (gr < n && gc < n) out[gr * ldo + gc] = -t.acc[m][nn][i];
      }
}

// alpha[i] = sum_k x[i, k] z[k] over k >= (i / 128) * 128 (one wave per row, fp64 accumulation); block 0 also leaves
// quad = sum_k z[k]^2 (fixed order: reproducible).
template <typename T>
__global__ void __launch_bounds__(256) rows_dot_kernel(const T* __restrict__ x, int64_t ldx, const T* __restrict__ z, int64_t kcols,
                                                       int64_t n, T* __restrict__ alpha, double* __restrict__ quad) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row < n) {
    double s = 0.0;
    for (int64_t k = row / kTile * kTile + lane; k < kcols; k += 64) s += (double)x[row * ldx + k] * (double)z[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) alpha[row] = (T)s;
  }
  if (blockIdx.x == 0) {
    __shared__ double red[256];
    double s = 0.0;
    for (int64_t k = threadIdx.x; k < kcols; k += 256) s += (double)z[k] * (double)z[k];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
      __syncthreads();
    }
    if (threadIdx.x == 0) *quad = red[0];
  }
}

template <typename T>
__global__ void diag_trace_kernel(const T* __restrict__ a, int64_t lda, int64_t n, double* __restrict__ out) {
  // single block; deterministic tree
  __shared__ double red[256];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += (double)a[i * lda + i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = red[0];
}

template <typename T>
__global__ void diag_shift_kernel(T* __restrict__ a, int64_t lda, int64_t n, double jitter_abs, double ridge_rel,
                                  const double* __restrict__ trace) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double sh = jitter_abs + (ridge_rel != 0.0 ? ridge_rel * (*trace) / (double)n : 0.0);
  a[i * lda + i] = (T)((double)a[i * lda + i] + sh);
}

__global__ void init_scalars_kernel(double* logdet, int* info) {
  *logdet = 0.0;
  *info = INT_MAX;
}

template <typename T>
int launch_update(smn_ctx* ctx, hipStream_t st, T* a, int64_t lda, int64_t r0, int64_t c0, int64_t k0, int64_t K,
                  int64_t tiles_m, int64_t tiles_n, int lower, int tag = -1) {
  if (tiles_m <= 0 || tiles_n <= 0 || K <= 0) return SMN_OK;
  if (tag < 0) tag = lower ? 1 : 0;   // 0: strip update, 1: trailing update (separate symbols / profile categories)
  if (ctx->chol_id0 >= 0) {
    // Identity rows [id0, id1) (analytic gradients): row id0 + i of the panel is zero left of column i, so the rows from
    // id0 + k0 + K on multiply zeros in this update -- as row operand AND as column operand.  What is left is the
    // contiguous rows [r0, act) in the caller's shape plus the rows below the identity block (the right-hand sides)
    // against the live columns; launched as such (the same tiles, the same arithmetic), not as one grid whose dead
    // workgroups leave at once: the XCD-aware tile order hands every XCD one contiguous eighth of the grid, and the dead
    // tiles are the last 70 % of it (profiles/r03_grad_identity_skip.txt: 3 of 8 XCDs were doing all the work).
    const int64_t id0 = ctx->chol_id0, id1 = ctx->chol_id1, r_end = r0 + tiles_m * kTile;
    const int64_t act = std::min(id1, id0 + (k0 + K + kTile - 1) / kTile * kTile);
    if (act < id1 && r_end > act) {
      ctx->chol_id0 = -1;
      int rc = SMN_OK;
      if (r0 < act) rc = launch_update<T>(ctx, st, a, lda, r0, c0, k0, K, (act - r0) / kTile, tiles_n, lower, tag);
      if (rc == SMN_OK && r_end > id1) {
        const int64_t tb = (r_end - id1) / kTile;
        // live columns of those rows: the caller's own (strip, trapezoid: all left of id0) or, for a triangle, [c0, act)
        const int64_t nb = lower == 1 ? (act - c0) / kTile : tiles_n;
        rc = launch_update<T>(ctx, st, a, lda, id1, c0, k0, K, tb, nb, 0, tag);
        if (rc == SMN_OK && lower == 1) rc = launch_update<T>(ctx, st, a, lda, id1, id1, k0, K, tb, tb, 1, tag);
      }
      ctx->chol_id0 = id0;
      return rc;
    }
  }
  if (lower == 2 && tiles_n >= tiles_m) {   // a trapezoid as wide as it is tall is the triangle
    lower = 1;
    tiles_n = tiles_m;
  }
  UpdArgs<T> u{a, lda, r0, c0, k0, (int)K, (int)tiles_n, lower, 0, TileMap::make(tiles_m, tiles_n, lower == 1),
               ctx->batch_logdet ? ctx->batch_stride : 0};
  const unsigned gy = (unsigned)(ctx->batch_logdet ? ctx->batch : 1);
  int64_t nt = lower == 1   ? tiles_m * (tiles_m + 1) / 2
               : lower == 2 ? tiles_n * (tiles_n + 1) / 2 + (tiles_m - tiles_n) * tiles_n
                            : tiles_m * tiles_n;
  ctx->prof_flops[tag ? PROF_TRAIL : PROF_STRIP] += 2.0 * kTile * kTile * (double)K * (double)nt * gy;   // executed: whole tiles
  if (ctx->xcd_map && nt >= 512 && lower != 2) {   // small launches do not fill the XCDs anyway
    u.use_map = 1;
    nt = u.map.grid;
  }
  const size_t lds = MainTile<T>::LDS_BYTES;
  if constexpr (sizeof(T) == 4) {
    // CUs this stream may use: the bulk stream of the look-ahead is masked off the chain's CUs
    const int cus = (st == ctx->stream_bulk && st != nullptr) ? ctx->num_cu - ctx->chain_cus : ctx->num_cu;
    if (tag == 1 && lower && !u.use_map && gy == 1 && nt > 2 * cus && K <= kPersistMaxK) {
      // persistent walk over the lower tiles, two workgroups per CU
      const size_t plds = TileNT<T, kTile, kTile, 2>::LDS_BYTES;
      ProfScope ps(ctx, PROF_TRAIL, st);
      hipLaunchKernelGGL(trail_kernel<T>, dim3((unsigned)(2 * cus)), dim3(256), plds, st, u, (int)nt);
      SMN_CHECK_LAUNCH(ctx);
      return SMN_OK;
    }
  }
  if (!u.use_map && nt <= kQuarterTileMax) {
    // very few tiles (strips; near / F0 updates of the chain-bound tail): 64x64 tiles, four workgroups per tile
    ProfScope ps(ctx, tag ? PROF_TRAIL : PROF_STRIP, st);
    constexpr size_t qlds = TileNT<T, 64, 64, SMN_STAGES>::LDS_BYTES;
    if (tag) {
      auto kern = update_kernel<T, 1, 64, 64>;
      hipLaunchKernelGGL(kern, dim3((unsigned)(4 * nt), gy), dim3(256), qlds, st, u);
    } else {
      auto kern = update_kernel<T, 0, 64, 64>;
      hipLaunchKernelGGL(kern, dim3((unsigned)(4 * nt), gy), dim3(256), qlds, st, u);
    }
    SMN_CHECK_LAUNCH(ctx);
    return SMN_OK;
  }
  if (!u.use_map && nt <= kHalfTileMax) {
    // too few 128x128 tiles to fill the chip: 64-row tiles, twice as many workgroups
    const int64_t nh = lower == 1   ? tiles_m * (tiles_m + 1)
                       : lower == 2 ? tiles_n * (tiles_n + 1) + 2 * (tiles_m - tiles_n) * tiles_n
                                    : 2 * tiles_m * tiles_n;
    ProfScope ps(ctx, tag ? PROF_TRAIL : PROF_STRIP, st);
    constexpr size_t hlds = TileNT<T, 64, kTile, SMN_STAGES>::LDS_BYTES;
    if (tag) {
      auto kern = update_kernel<T, 1, 64>;
      hipLaunchKernelGGL(kern, dim3((unsigned)nh, gy), dim3(256), hlds, st, u);
    } else {
      auto kern = update_kernel<T, 0, 64>;
      hipLaunchKernelGGL(kern, dim3((unsigned)nh, gy), dim3(256), hlds, st, u);
    }
    SMN_CHECK_LAUNCH(ctx);
    return SMN_OK;
  }
  {
    ProfScope ps(ctx, tag ? PROF_TRAIL : PROF_STRIP, st);
    if (tag) {
      auto kern = update_kernel<T, 1>;
      hipLaunchKernelGGL(kern, dim3((unsigned)nt, gy), dim3(256), lds, st, u);
    } else {
      auto kern = update_kernel<T, 0>;
      hipLaunchKernelGGL(kern, dim3((unsigned)nt, gy), dim3(256), lds, st, u);
    }
  }
  SMN_CHECK_LAUNCH(ctx);
  return SMN_OK;
}

template <typename T, int XRV>
int launch_panel_x(smn_ctx* ctx, hipStream_t st, T* a, int64_t lda, int64_t j0, int64_t n_total, int prefactored) {
  const int64_t rbeg = j0 + PB;
  const int64_t below = n_total - rbeg;
  const unsigned grid = below > 0 ? (unsigned)((below + XRV - 1) / XRV) : 1u;
  const bool batched = ctx->batch_logdet != nullptr;   // (a batch of ONE problem still keeps its scalars in the batch arrays)
  const unsigned gy = (unsigned)(batched ? ctx->batch : 1);
  const int64_t bs = batched ? ctx->batch_stride : 0, ls = batched ? ctx->batch_ldiag_stride : 0;
  double* logdet = batched ? ctx->batch_logdet : ctx->d_scal;
  int* info = batched ? ctx->batch_info : ctx->d_info;
  if (ctx->panel_leaf && !prefactored) {
    ProfScope ps(ctx, PROF_PANEL, st);
    T* ldiag = reinterpret_cast<T*>(ctx->ws[3]) + (j0 / PB) * (int64_t)(PB * PB);
    // f64, more row groups than CUs (16 rows per workgroup: from 4096 rows on): every workgroup re-factors the diagonal block,
    // and a second round of them would do it all over again.  Instead each carries `passes` groups, the later ones through
    // the solve alone (panelr_kernel).  Same bits.  Measured (profiles/r04_multipass_panel.txt): f64 N = 8192 6.24 -> 5.92 ms,
    // N = 16384 32.6 -> 31.1 ms; more passes than groups / CUs lose (even under the look-ahead's contention), and the f32
    // 128-row form gains nothing at N = 36864 (its workgroups already carry 128 rows): f64 only.
    int passes = 1;
    if (sizeof(T) == 8 && !batched && ctx->chol_id0 < 0 && (int64_t)grid > (int64_t)ctx->num_cu)
      passes = (int)std::min<int64_t>(((int64_t)grid + ctx->num_cu - 1) / ctx->num_cu, (int64_t)ctx->panel_max_passes);
    // a batch that fills the chip is throughput-bound: fewer workgroups per problem, each factoring the diagonal block once
    // for up to four groups of rows, is less work in all (the grid search's two batches 13.3 -> 12.9 ms, 256 problems of
    // N = 245 1.43 -> 1.34 us each)
    if (batched && ctx->chol_id0 < 0 && (int64_t)gy * grid > 2 * (int64_t)ctx->num_cu && grid > 1)
      passes = (int)std::min<int64_t>((int64_t)grid, (int64_t)ctx->panel_max_passes);
    const unsigned gridp = (grid + (unsigned)passes - 1) / (unsigned)passes;
    auto kernr = panelr_kernel<T, XRV>;
    hipLaunchKernelGGL(kernr, dim3(gridp, gy), dim3(2 * panel_threads(XRV)), panelr_lds_bytes<T>(XRV), st, a, lda, j0, rbeg, n_total,
                       logdet, info, ldiag, ctx->chol_id0, ctx->chol_id1, bs, ls, passes);
    SMN_CHECK_LAUNCH(ctx);
    return SMN_OK;
  }
  auto kern = panel_kernel<T, XRV>;
  {
    ProfScope ps(ctx, PROF_PANEL, st);
    T* ldiag = reinterpret_cast<T*>(ctx->ws[3]) + (j0 / PB) * (int64_t)(PB * PB);
    hipLaunchKernelGGL(kern, dim3(grid, gy), dim3(panel_threads(XRV)), panel_lds_bytes<T>(XRV), st, a, lda, j0, rbeg, n_total,
                       prefactored, logdet, info, ldiag, ctx->chol_id0, ctx->chol_id1, bs, ls);
  }
  SMN_CHECK_LAUNCH(ctx);
  return SMN_OK;
}

template <typename T>
int launch_panel(smn_ctx* ctx, hipStream_t st, T* a, int64_t lda, int64_t j0, int64_t n_total, int prefactored) {
  if constexpr (sizeof(T) == 4 && PanelCfg<T>::XR == 128) {
    // few rows left: 16-row workgroups (eight times as many, each with half the block-update work)
    const int64_t below = n_total - (j0 + PB);
    if (below <= (int64_t)kPanelSmallRows) {
      // latency-bound (one problem, or a batch that does not fill the chip anyway): 16-row workgroups; a batch that does fill
      // it is throughput-bound, and every workgroup redoes the diagonal block: 64-row ones, a quarter as many
      const int64_t gy = ctx->batch_logdet ? ctx->batch : 1;
      if (gy * ((below + kPanelSmallXR - 1) / kPanelSmallXR) <= 2 * (int64_t)ctx->num_cu)
        return launch_panel_x<T, kPanelSmallXR>(ctx, st, a, lda, j0, n_total, prefactored);
      return launch_panel_x<T, 64>(ctx, st, a, lda, j0, n_total, prefactored);
    }
  }
  return launch_panel_x<T, PanelCfg<T>::XR>(ctx, st, a, lda, j0, n_total, prefactored);
}

template <typename T>
int set_lds_attrs(smn_ctx* ctx) {
  bool& done = ctx->lds_attrs_done[sizeof(T) == 8 ? 1 : 0];   // per context (= per device) and dtype
  if (done) return SMN_OK;
  SMN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(panel_kernel<T>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)panel_lds_bytes<T>()));
  if constexpr (sizeof(T) == 4 && PanelCfg<T>::XR == 128) {
    SMN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(panel_kernel<T, kPanelSmallXR>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)panel_lds_bytes<T>(kPanelSmallXR)));
    SMN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(panel_kernel<T, 64>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)panel_lds_bytes<T>(64)));
  }
  SMN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(panelr_kernel<T>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)panelr_lds_bytes<T>()));
  if constexpr (sizeof(T) == 4 && PanelCfg<T>::XR == 128) {
    SMN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(panelr_kernel<T, kPanelSmallXR>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)panelr_lds_bytes<T>(kPanelSmallXR)));
    SMN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(panelr_kernel<T, 64>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)panelr_lds_bytes<T>(64)));
  }

  SMN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(update_kernel<T, 0>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)MainTile<T>::LDS_BYTES));
  SMN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(update_kernel<T, 1>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)MainTile<T>::LDS_BYTES));
  SMN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(update_kernel<T, 1, 64>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(TileNT<T, 64, kTile, SMN_STAGES>::LDS_BYTES)));
  SMN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(update_kernel<T, 0, 64>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(TileNT<T, 64, kTile, SMN_STAGES>::LDS_BYTES)));
  if constexpr (sizeof(T) == 4)
    SMN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(trail_kernel<T>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)TileNT<T, kTile, kTile, 2>::LDS_BYTES));
  done = true;
  return SMN_OK;
}

template <typename T>
int cholesky_t(smn_ctx* ctx, T* a, int64_t n_total, int64_t n_factor, int64_t lda, int64_t n_shift, double jitter_abs,
               double ridge_rel, bool keep_factor) {
  SMN_TRY(set_lds_attrs<T>(ctx));
  void* side = nullptr;   // factored diagonal blocks, [n_factor/128][128*128]
  SMN_TRY(smn_workspace(ctx, 3, sizeof(T) * (size_t)n_factor * PB * (size_t)(ctx->batch_logdet ? ctx->batch : 1), &side));
  ctx->batch_ldiag_stride = n_factor * PB;
  hipStream_t st = ctx->stream;
  if (!ctx->chol_prepped) hipLaunchKernelGGL(init_scalars_kernel, dim3(1), dim3(1), 0, st, ctx->d_scal, ctx->d_info);
  if (!ctx->chol_prepped && n_shift > 0 && (jitter_abs != 0.0 || ridge_rel != 0.0)) {
    if (ridge_rel != 0.0)
      hipLaunchKernelGGL(diag_trace_kernel<T>, dim3(1), dim3(256), 0, st, a, lda, n_shift, ctx->d_scal + 1);
    hipLaunchKernelGGL(diag_shift_kernel<T>, dim3((unsigned)((n_shift + 255) / 256)), dim3(256), 0, st, a, lda,
                       n_shift, jitter_abs, ridge_rel, ctx->d_scal + 1);
  }
  SMN_CHECK_LAUNCH(ctx);
  // Two-level, right-looking.  Inside a super-panel of S columns the K = 256 updates touch only the super-panel's
  // own columns (a lower trapezoid); everything to the right of it is brought up to date ONCE, when the super-panel is
  // finished, with K = S.  Most flops run at the long K the tile engine likes (profiles/r01b_gemm_probe.txt) and the C
  // block is read and written once per S columns instead of once per 256.
  // Look-ahead across super-panels (n_total >= chain_min_n): that far update is split by tile column into
  //   F0  the NEXT super-panel's columns: stays on the caller's (high-priority) stream, the next chain needs it;
  //   F1  everything beyond: goes to the CU-masked bulk stream and runs beside the next super-panel's panel chain,
  //       which always finds the reserved CUs free.
  // F1 launches are serial on the bulk stream; F0(s) writes tiles F1(s-1) also writes, so it waits for it (ev_b);
  // F1(s) and the chain of s+1 touch disjoint columns.  Two events, two streams; results are bit-identical to the
  // one-stream order (same launches, same tiles).
  hipStream_t sb = (n_total >= ctx->chain_min_n && ctx->stream_bulk) ? ctx->stream_bulk : nullptr;
  // Outer panels: 256 columns (one strip of K = 128, then a near update of K = 256); with the look-ahead 512 (strips of K = 128,
  // 256, 384 through the plain K loop -- the pipelined one measured slower there --, near updates of K = 512: half as many
  // near launches on the chain; C4 -0.7 %, C5 -0.5 %, C2 and f64 N = 8192 flat: profiles/r04_outer_panel_sweep.txt)
  const int64_t W = sb ? kOuterWide : 2 * PB;
  int64_t S = ctx->super_panel / W * W;
  if (S < W) S = W;
  int64_t Swide = kSuperWide;
  if (Swide < S) Swide = S;
  bool bulk_busy = false;
  // ctx->chol_noschur: the trailing block of the appended rows (rows and columns >= n_factor) is neither read nor written
  // -- the caller wants B L^-T only (grad.hip: X = L^-T from B = I, then K~^-1 = X X^T as one full-rate launch), and the
  // matrix need not even have those columns (lda >= n_factor suffices)
  const bool noschur = ctx->chol_noschur && n_factor < n_total;
  int rc = SMN_OK;
  auto hip_ok = [&](hipError_t e) {
    if (e != hipSuccess && rc == SMN_OK)
      rc = smn_fail(ctx, SMN_EHIP, "cholesky: %s", hipGetErrorString(e));
  };
  // Column-first multi-GPU exchange (heads.hip smn_lml_from_shards): the kernel's columns land in this workspace piece by
  // piece while the factorisation is already being issued.  A stream about to touch columns [.., col_hi) waits for the
  // pieces that hold them, once each: a sub-panel for its own 128 columns, the near updates for their super-panel's, F0 for the next super-panel's,
  // the bulk update F1 for everything.  The first wait (super-panel 0) is what of the exchange is exposed; later ones are
  // stalls the first panel chain did not cover (separate profile categories).
  const bool arriving = ctx->consume_arrivals && !ctx->arrivals.empty();
  std::vector<char> seen_main, seen_bulk;
  if (arriving) {
    seen_main.assign(ctx->arrivals.size(), 0);
    seen_bulk.assign(ctx->arrivals.size(), 0);
  }
  bool first_need = true;
  auto need_columns = [&](hipStream_t s, int64_t col_hi) -> int {
    if (!arriving) return SMN_OK;
    std::vector<char>& seen = (s == st) ? seen_main : seen_bulk;
    bool any = false;
    for (size_t i = 0; i < ctx->arrivals.size(); ++i)
      if (!seen[i] && ctx->arrivals[i].col_begin < col_hi) any = true;
    if (!any) return SMN_OK;
    ProfScope ps(ctx, first_need ? PROF_EXPOSED : PROF_STALL, s);
    first_need = false;
    for (size_t i = 0; i < ctx->arrivals.size(); ++i)
      if (!seen[i] && ctx->arrivals[i].col_begin < col_hi) {
        SMN_HIP(ctx, hipStreamWaitEvent(s, ctx->arrivals[i].ev, 0));
        seen[i] = 1;
      }
    return SMN_OK;
  };
  auto body = [&]() -> int {
    // Super-panels are wider while many rows are left (the update-bound phase: a K = 2048 far update runs closer to the
    // tile engine's rate and halves the launches and their tails) and S wide in the chain-bound rest.
    auto width = [&](int64_t c0) { return (sb && n_total - c0 >= ctx->super_wide_rows) ? Swide : S; };
    for (int64_t s0 = 0, s_stop = 0; s0 < n_factor; s0 = s_stop) {
      const int64_t Sc = width(s0);
      const int64_t s_end = (n_factor - s0 < Sc) ? n_factor : s0 + Sc;
      s_stop = s_end;
      for (int64_t j0 = s0; j0 < s_end; j0 += W) {
        const int64_t w = (s_end - j0 < W) ? s_end - j0 : W;
        for (int64_t js = j0; js < j0 + w; js += PB) {
          SMN_TRY(need_columns(st, js + PB));   // (arrivals: a sub-panel and its strip touch their own 128 columns only)
          if (js > j0)   // strip: this sub-panel's 128 columns, K = the outer panel's finished columns
            SMN_TRY(launch_update<T>(ctx, st, a, lda, js, js, j0, js - j0, (n_total - js) / kTile, 1, 0));
          SMN_TRY(launch_panel<T>(ctx, st, a, lda, js, n_total, 0));
        }
        const int64_t j1 = j0 + w;
        if (j1 < s_end) SMN_TRY(need_columns(st, s_end));
        if (j1 < s_end)   // near update: columns [j1, s_end), all rows from the diagonal down
          SMN_TRY(launch_update<T>(ctx, st, a, lda, j1, j1, j0, w, (n_total - j1) / kTile, (s_end - j1) / kTile, 2));
      }
      if (s_end >= n_total) break;
      const int64_t K = s_end - s0;
      if (!sb) {   // far update, one launch
        SMN_TRY(need_columns(st, n_total));
        const int64_t tm = (n_total - s_end) / kTile;
        if (!noschur) SMN_TRY(launch_update<T>(ctx, st, a, lda, s_end, s_end, s0, K, tm, tm, 1));
        else if (n_factor > s_end) SMN_TRY(launch_update<T>(ctx, st, a, lda, s_end, s_end, s0, K, tm, (n_factor - s_end) / kTile, 2));
        continue;
      }
      const int64_t Sn = width(s_end);   // the next super-panel's width decides where F0 ends and F1 begins
      const int64_t s_next = s_end >= n_factor ? s_end : ((n_factor - s_end < Sn) ? n_factor : s_end + Sn);
      // Once F1 is small (the chain-bound tail) it starts BEHIND F0 instead of beside it: F0 is on the chain's critical path
      // and, sharing the chip with an F1 that nobody waits for, takes three times as long (profiles/r02_tail_chain_timeline.txt).
      const int64_t tm1 = n_total > s_next ? (n_total - s_next) / kTile : 0;
      const int64_t tn1 = noschur ? (n_factor > s_next ? (n_factor - s_next) / kTile : 0) : tm1;   // F1's tile columns
      const bool f0_first = tn1 * (tn1 + 1) / 2 + (tm1 - tn1) * tn1 <= kF0FirstTiles;
      if (!f0_first) {
        SMN_HIP(ctx, hipEventRecord(ctx->ev_a, st));
        SMN_HIP(ctx, hipStreamWaitEvent(sb, ctx->ev_a, 0));
      }
      if (bulk_busy) SMN_HIP(ctx, hipStreamWaitEvent(st, ctx->ev_b, 0));
      if (s_next > s_end) SMN_TRY(need_columns(st, s_next));
      if (s_next > s_end)   // F0
        SMN_TRY(launch_update<T>(ctx, st, a, lda, s_end, s_end, s0, K, (n_total - s_end) / kTile, (s_next - s_end) / kTile, 2));
      if (f0_first) {
        SMN_HIP(ctx, hipEventRecord(ctx->ev_a, st));
        SMN_HIP(ctx, hipStreamWaitEvent(sb, ctx->ev_a, 0));
      }
      if (n_total > s_next && tn1 > 0) {   // F1
        const int64_t tm = (n_total - s_next) / kTile;
        bulk_busy = true;       // set first: an error below must still join the bulk stream
        SMN_TRY(need_columns(sb, n_total));
        if (!noschur) {
          SMN_TRY(launch_update<T>(ctx, sb, a, lda, s_next, s_next, s0, K, tm, tm, 1));
        } else {
          // no Schur block: the triangle of the columns left, then the rectangle of the appended rows under it (two
          // launches, so that each takes the XCD patch order a trapezoid does not have)
          SMN_TRY(launch_update<T>(ctx, sb, a, lda, s_next, s_next, s0, K, tn1, tn1, 1));
          if (tm > tn1) SMN_TRY(launch_update<T>(ctx, sb, a, lda, s_next + tn1 * kTile, s_next, s0, K, tm - tn1, tn1, 0, 1));
        }
        SMN_HIP(ctx, hipEventRecord(ctx->ev_b, sb));
      }
    }
    return SMN_OK;
  };
  rc = body();
  if (arriving) {   // the caller's stream ends up behind every piece, whatever the column structure above consumed
    const int rc2 = need_columns(st, INT64_MAX);
    if (rc == SMN_OK) rc = rc2;
    if (rc != SMN_OK) (void)hipStreamSynchronize(ctx->stream_scatter);
  }
  // The caller's stream continues after the bulk stream whatever happened above: an error exit must not leave F1 work
  // running behind a workspace the caller is about to free or reuse.
  if (bulk_busy) {
    if (rc != SMN_OK) {
      (void)hipStreamSynchronize(sb);
    } else {
      hip_ok(hipEventRecord(ctx->ev_b, sb));
      hip_ok(hipStreamWaitEvent(st, ctx->ev_b, 0));
    }
  }
  if (rc == SMN_OK && keep_factor && ctx->batch_logdet) rc = smn_fail(ctx, SMN_ENOTSUP, "cholesky: keep_factor in a batched factorisation");
  if (rc == SMN_OK && keep_factor) {
    hipLaunchKernelGGL(copy_diag_kernel<T>, dim3((unsigned)(n_factor / PB)), dim3(1024), 0, st, a, lda,
                       static_cast<const T*>(side));
    hip_ok(hipGetLastError());
  }
  return rc;
}

}  // namespace

// K~^-1 and alpha from the appended rows of a no-Schur factorisation of [[K~], [I], [y^T]] (internal.hpp).
int inverse_from_rows(smn_ctx* ctx, int dtype, const void* x, int64_t ldx, const void* z, int64_t kcols, int64_t n,
                      void* neg_inv, int64_t ldo, void* alpha, double* quad_dev) {
  const int64_t t = (n + kTile - 1) / kTile, ntiles = t * (t + 1) / 2;
  if (dtype == SMN_F64) SMN_TRY(set_lds_attrs<double>(ctx)); else SMN_TRY(set_lds_attrs<float>(ctx));
  {
    ProfScope ps(ctx, PROF_TRAIL, ctx->stream);
    ctx->prof_flops[PROF_TRAIL] += 2.0 * kTile * kTile * (double)kTile * (double)(t * (t + 1) * (t + 2) / 6);
    if (dtype == SMN_F64) {
      SMN_TRY(smn_allow_lds(ctx, reinterpret_cast<const void*>(syrk_rows_kernel<double>), MainTile<double>::LDS_BYTES));
      hipLaunchKernelGGL(syrk_rows_kernel<double>, dim3((unsigned)ntiles), dim3(256), MainTile<double>::LDS_BYTES, ctx->stream,
                         static_cast<const double*>(x), ldx, kcols, static_cast<double*>(neg_inv), ldo, n);
    } else {
      SMN_TRY(smn_allow_lds(ctx, reinterpret_cast<const void*>(syrk_rows_kernel<float>), MainTile<float>::LDS_BYTES));
      hipLaunchKernelGGL(syrk_rows_kernel<float>, dim3((unsigned)ntiles), dim3(256), MainTile<float>::LDS_BYTES, ctx->stream,
                         static_cast<const float*>(x), ldx, kcols, static_cast<float*>(neg_inv), ldo, n);
    }
  }
  SMN_CHECK_LAUNCH(ctx);
  const unsigned gb = (unsigned)((n + 3) / 4);
  if (dtype == SMN_F64)
    hipLaunchKernelGGL(rows_dot_kernel<double>, dim3(gb), dim3(256), 0, ctx->stream, static_cast<const double*>(x), ldx,
                       static_cast<const double*>(z), kcols, n, static_cast<double*>(alpha), quad_dev);
  else
    hipLaunchKernelGGL(rows_dot_kernel<float>, dim3(gb), dim3(256), 0, ctx->stream, static_cast<const float*>(x), ldx,
                       static_cast<const float*>(z), kcols, n, static_cast<float*>(alpha), quad_dev);
  SMN_CHECK_LAUNCH(ctx);
  return SMN_OK;
}

int cholesky_padded(smn_ctx* ctx, int dtype, void* a, int64_t n_total, int64_t n_factor, int64_t lda, int64_t n_shift,
                    double jitter_abs, double ridge_rel, bool keep_factor, int64_t id0, int64_t id1) {
  if (n_total % kTile || n_factor % kTile || n_factor > n_total || n_factor <= 0)
    return smn_fail(ctx, SMN_EINVAL, "cholesky_padded: n_total=%lld n_factor=%lld must be multiples of %d",
                    (long long)n_total, (long long)n_factor, kTile);
  if (lda % (16 / (int)dtype_size(dtype)) || (reinterpret_cast<uintptr_t>(a) & 15))
    return smn_fail(ctx, SMN_EINVAL, "cholesky_padded: matrix must be 16-byte aligned");
  if (id0 >= 0 && (id0 < n_factor || id0 % kTile || id1 < id0 || id1 > n_total))
    return smn_fail(ctx, SMN_EINVAL, "cholesky_padded: bad identity-row hint");
  ctx->chol_id0 = id0;
  ctx->chol_id1 = id0 >= 0 ? id1 : -1;
  const int rc = dtype == SMN_F64
                     ? cholesky_t<double>(ctx, static_cast<double*>(a), n_total, n_factor, lda, n_shift, jitter_abs,
                                          ridge_rel, keep_factor)
                     : cholesky_t<float>(ctx, static_cast<float*>(a), n_total, n_factor, lda, n_shift, jitter_abs, ridge_rel,
                                         keep_factor);
  ctx->chol_id0 = ctx->chol_id1 = -1;
  return rc;
}

// Solve-only sweep: rows [n_factor, n_total) of `a` <- rows * L^-T with L = the (already factored)
// leading block.  Left-looking over block columns; used by smn_trsm.
int solve_rows_padded(smn_ctx* ctx, int dtype, void* a, int64_t n_total, int64_t n_factor, int64_t lda) {
  if (n_total % kTile || n_factor % kTile) return smn_fail(ctx, SMN_EINVAL, "solve_rows_padded: padding");
  hipStream_t st = ctx->stream;
  const int64_t tm = (n_total - n_factor) / kTile;
  // Two-level like the factorisation: left-looking inside a super-panel of S columns (the update of a 128-column block
  // reaches back to the super-panel's first column only), and when the super-panel is solved every later column of the
  // appended rows takes its contribution at once (K = S, tm x (columns left / 128) tiles: a launch that fills the chip,
  // where the purely left-looking sweep issued n_factor / 128 launches of tm tiles with K up to n_factor).
  int64_t S = ctx->super_panel / PB * PB;
  if (S < PB) S = PB;
  auto upd = [&](int64_t c0, int64_t k0, int64_t K, int64_t tiles_n) -> int {
    const int tag = K >= 256 ? 1 : 0;   // the pipelined K loop from K = 256 on (profile category: trailing update)
    if (dtype == SMN_F64) return launch_update<double>(ctx, st, static_cast<double*>(a), lda, n_factor, c0, k0, K, tm, tiles_n, 0, tag);
    return launch_update<float>(ctx, st, static_cast<float*>(a), lda, n_factor, c0, k0, K, tm, tiles_n, 0, tag);
  };
  if (dtype == SMN_F64) SMN_TRY(set_lds_attrs<double>(ctx));
  else SMN_TRY(set_lds_attrs<float>(ctx));
  for (int64_t s0 = 0; s0 < n_factor; s0 += S) {
    const int64_t s_end = (n_factor - s0 < S) ? n_factor : s0 + S;
    for (int64_t js = s0; js < s_end; js += PB) {
      // appended rows only: C = a[n_factor:, js:js+128] -= a[n_factor:, s0:js] * a[js:js+128, s0:js]^T
      if (js > s0) SMN_TRY(upd(js, s0, js - s0, 1));
      // panel solve against the prefactored diagonal block; row blocks start at n_factor
      if (dtype == SMN_F64) {
        constexpr int XR = PanelCfg<double>::XR;
        const unsigned grid = (unsigned)((n_total - n_factor + XR - 1) / XR);
        hipLaunchKernelGGL(panel_kernel<double>, dim3(grid), dim3(panel_threads(XR)),
                           panel_lds_bytes<double>(), st, static_cast<double*>(a), lda, js, n_factor,
                           n_total, 1, ctx->d_scal, ctx->d_info, static_cast<double*>(nullptr), (int64_t)-1, (int64_t)-1, (int64_t)0, (int64_t)0);
      } else {
        constexpr int XR = PanelCfg<float>::XR;
        const unsigned grid = (unsigned)((n_total - n_factor + XR - 1) / XR);
        hipLaunchKernelGGL(panel_kernel<float>, dim3(grid), dim3(panel_threads(XR)),
                           panel_lds_bytes<float>(), st, static_cast<float*>(a), lda, js, n_factor,
                           n_total, 1, ctx->d_scal, ctx->d_info, static_cast<float*>(nullptr), (int64_t)-1, (int64_t)-1, (int64_t)0, (int64_t)0);
      }
      SMN_CHECK_LAUNCH(ctx);
    }
    if (s_end < n_factor) SMN_TRY(upd(s_end, s0, s_end - s0, (n_factor - s_end) / kTile));
  }
  return SMN_OK;
}

namespace {
// mail[0] = logdet, mail[1] = info, mail[2 .. 2+nq) = the quadratic forms: written straight into pinned host memory
__global__ void publish_kernel(const double* __restrict__ scal, const int* __restrict__ info,
                               const double* __restrict__ quad, int nq, double* __restrict__ mail) {
  const int t = threadIdx.x;
  if (t == 0) mail[0] = scal[0];
  if (t == 1) mail[1] = (double)info[0];
  if (t < nq) mail[2 + t] = quad[t];
}
}  // namespace

int fetch_results(smn_ctx* ctx, const double* quad_dev, int nq, double* quad_h, double* logdet, int* info) {
  if (nq < 0 || nq > 62) return smn_fail(ctx, SMN_EINVAL, "fetch_results: %d values", nq);
  hipLaunchKernelGGL(publish_kernel, dim3(1), dim3(64), 0, ctx->stream, ctx->d_scal, ctx->d_info, quad_dev, nq,
                     ctx->d_mail);
  SMN_CHECK_LAUNCH(ctx);
  SMN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const volatile double* m = ctx->h_mail;
  int inf = (int)m[1];
  if (inf == INT_MAX) inf = 0;
  if (logdet) *logdet = m[0];
  if (info) *info = inf;
  for (int i = 0; i < nq && quad_h; ++i) quad_h[i] = m[2 + i];
  return SMN_OK;
}

int fetch_mail(smn_ctx* ctx, int nq, double* quad_h, double* logdet, int* info) {
  if (nq < 0 || nq > 62) return smn_fail(ctx, SMN_EINVAL, "fetch_mail: %d values", nq);
  SMN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const volatile double* m = ctx->h_mail;
  int inf = (int)m[1];
  if (inf == INT_MAX) inf = 0;
  if (logdet) *logdet = m[0];
  if (info) *info = inf;
  for (int i = 0; i < nq && quad_h; ++i) quad_h[i] = m[2 + i];
  return SMN_OK;
}

int fetch_logdet_info(smn_ctx* ctx, double* logdet, int* info) {
  return fetch_results(ctx, nullptr, 0, nullptr, logdet, info);
}
