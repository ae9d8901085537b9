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
// Columns are cut three ways: blocks of S columns (SMN_SUPER, default 1024), outer panels of 256 inside them,
// 128-column sub-panels inside those.
//   potrf_kernel   ONE workgroup per sub-panel: POTRF of the 128x128 diagonal block in LDS (16-column blocks brought up
//                  to date with MFMAs straight from the LDS image, 8-column micro-panels factored in registers), then
//                  its inverse, in place (recursive doubling on the MFMA); stores L_kk, sum(log pivots), info, L_kk^-1.
//   trsm_kernel    the rows below as a GEMM against L_kk^-1 on the tile engine (64-row tiles, K = 128).
//   update_kernel  C -= A B^T on the f32/f64 MFMA (gemm_nt.hpp):
//     strip   the 128 columns in front of the second sub-panel of a pair (K = 128);
//     near    after an outer panel, the rest of ITS block only (K = 256, a lower trapezoid);
//     B1..B3  after a block, the blocks to its right (cholesky_t below: right-looking inside a window of D blocks,
//             one long-K left-looking update for the block that enters the window).  From n_total = chain_min_n on
//             they run on a stream whose CU mask leaves chain_cus CUs alone, beside the next block's chain.
//   trail_kernel   persistent form of the K <= 512 updates (one stream of K-steps per workgroup).
// Rows [id0, id1) may be declared an identity block (analytic gradients: cholesky_padded's hint): TRSM and update
// workgroups whose rows are still structurally zero in the columns at hand leave at once.
#include <climits>
#include <type_traits>
#include <vector>

#include "gemm_nt.hpp"
#include "internal.hpp"

namespace {

constexpr int PB = 128;  // sub-panel width == diagonal block edge == GEMM tile edge
constexpr int MP = 8;    // micro-panel width of the in-LDS factorisation

// 16x16 MFMA tiles for the in-LDS block operations of the diagonal kernel.
template <typename T>
struct PanelMma;
template <>
struct PanelMma<float> {   // v_mfma_f32_16x16x4_f32: A[m = lane & 15][k = lane >> 4], B[k = lane >> 4][n = lane & 15]
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
struct DiagCfg {
  static constexpr int LD = PB + (sizeof(T) == 4 ? 4 : 2);   // row stride (elements): 16-byte aligned rows, b128 reads conflict-free
  static constexpr int THREADS = 256;                        // 4 waves: one thread per row in the column phases, 2 MFMA row tiles per wave
  static constexpr size_t LDS = sizeof(T) * ((size_t)PB * LD + MP * MP + 2 * PB);
};

#ifdef SMN_PANEL_TIMING   // debug build only: phase times of the first diagonal block, printed by the kernel
#define PT_DECL long long pt_t = wall_clock64(), pt_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define PT_MARK(i) do { const long long n_ = wall_clock64(); pt_acc[i] += n_ - pt_t; pt_t = n_; } while (0)
#else
#define PT_DECL
#define PT_MARK(i)
#endif

__device__ __forceinline__ float rsqrt_t(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ double rsqrt_t(double x) { return 1.0 / sqrt(x); }

// The diagonal step of one 128-column sub-panel, ONE workgroup:
//   POTRF   L L^T = A_jj (lower triangle read, lower triangle written back in place), sum(log pivots), info;
//   TRTRI   X = L^-1, in place in LDS, written as a dense 128x128 block (zeros above the diagonal) to `linv_out`.
// The rows below the diagonal block are then solved by trsm_kernel as a GEMM against X (B <- B X^T), so the panel
// needs neither 135 KB of LDS per 128 rows nor a re-factorisation of A_jj in every workgroup (round 1's panel_kernel
// did both: 125 workgroups x 29 us on whole CUs per sub-panel at N = 16384).
// POTRF, left-looking over micro-panels of MP = 8 columns, one thread per row:
//   0. at every 16-column boundary the block's columns are brought up to date with all finished columns by
//      MFMAs that read both operands from the LDS image (16x16 tiles, 2 row tiles per wave);
//   1. each thread pulls its 8 entries into registers and subtracts the contribution of the (at most 8)
//      finished columns of the current 16-column block (16-byte LDS reads, pivot rows broadcast);
//   2. the 8 pivot rows publish their updated 8x8 diagonal micro-block; barrier;
//   3. every thread factors that 8x8 block redundantly in registers and runs the 8-step triangular solve on its own
//      8 values (for a pivot row this reproduces its row of L: d * rsqrt(d) = sqrt(d)); writes them back; barrier.
// TRTRI, recursive doubling in place: the eight 16x16 diagonal blocks are inverted by forward substitution (one
// thread per column), then for h = 16, 32, 64 every pair of finished h x h diagonal inverses X11, X22 turns the block
// L21 between them into X21 = -X22 (L21 X11): two MFMA passes per level, results held in accumulators across a
// barrier so that the block can be overwritten where it stands.
// prefactored != 0: the block already holds L (smn_trsm): only the inverse is formed.
template <typename T>
__global__ void __launch_bounds__(DiagCfg<T>::THREADS) potrf_kernel(T* __restrict__ a, int64_t lda, int64_t j0, int prefactored,
                                                                    double* __restrict__ logdet, int* __restrict__ info,
                                                                    T* __restrict__ linv_out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NT = DiagCfg<T>::THREADS, LD = DiagCfg<T>::LD;
  constexpr int VEC = 16 / sizeof(T);
  using vec_t = typename Mfma<T>::vec_t;
  using M = PanelMma<T>;
  T* S = reinterpret_cast<T*>(smem);        // [PB][LD]
  T* blk = S + PB * LD;                     // [MP][MP] staging of the diagonal micro-block
  T* piv = blk + MP * MP;                   // [PB] pivots d_j = L_jj^2 (for logdet / info)
  T* rdiag = piv + PB;                      // [PB] 1 / L_jj
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int RV = PB / VEC;              // 16-byte vectors per row
  PT_DECL;
  {
    constexpr int PER = PB * RV / NT;       // vectors per thread (16 in f32, 32 in f64), all in flight at once
    vec_t tmp[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int idx = u * NT + tid;
      tmp[u] = *reinterpret_cast<const vec_t*>(&a[(j0 + idx / RV) * lda + j0 + (idx % RV) * VEC]);
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int idx = u * NT + tid;
      *reinterpret_cast<vec_t*>(&S[(idx / RV) * LD + (idx % RV) * VEC]) = tmp[u];
    }
  }
  __syncthreads();
  PT_MARK(0);

  constexpr int CB = 16;                    // column block brought up to date on the MFMA
  constexpr int NW = NT / 64;               // waves
  constexpr int RT = PB / M::TM;            // 16-row tiles of the LDS image
  constexpr int TPW = RT / NW;              // row tiles per wave
  static_assert(RT % NW == 0 && TPW == 2, "two 16-row tiles per wave");
  const int fr = M::frag_row(lane), fk = M::frag_k(lane);
  const int row = tid;
  if (!prefactored) {
    for (int c0 = 0; c0 < PB; c0 += MP) {
      const int cb = c0 & ~(CB - 1);        // first column of the current 16-column block
      if (c0 == cb && cb > 0) {
        // S[rows >= cb, cb:cb+16] -= S[rows, 0:cb] * S[cb:cb+16, 0:cb]^T.  Tiles wholly above cb are finished rows; wave w
        // owns tiles w and w + NW, the skip count is wave-uniform, so each count gets a straight-line instantiation.
        const int first = cb / M::TM;
        const int u0 = first <= wave ? 0 : (first - wave + NW - 1) / NW;
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
        switch (u0) {
          case 0: block_update(std::integral_constant<int, 0>{}); break;
          case 1: block_update(std::integral_constant<int, 1>{}); break;
          default: break;
        }
        __syncthreads();
        PT_MARK(1);
      }
      const bool work = row < PB && row >= c0;
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
        if (row < c0 + MP) {
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
          for (int j = 0; j <= i; ++j) lm[i][j] = blk[i * MP + j];
#pragma unroll
        for (int j = 0; j < MP; ++j) {
          const T d = lm[j][j];
          rinv[j] = rsqrt_t(d);
          if (row == PB - 1) {                 // the last row takes part in every micro-panel
            piv[c0 + j] = d;
            rdiag[c0 + j] = rinv[j];
          }
#pragma unroll
          for (int i = j + 1; i < MP; ++i) lm[i][j] *= rinv[j];
#pragma unroll
          for (int i = j + 1; i < MP; ++i)
#pragma unroll
            for (int jj = j + 1; jj <= i; ++jj) lm[i][jj] = fma(-lm[i][j], lm[jj][j], lm[i][jj]);
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
    // logdet += sum_j log d_j, info = first non-positive pivot: two pivots per lane, ONE atomic per sub-panel.  Wave 3 owns
    // no rows, so its double-precision logs run beside the other waves' stores.
    if (wave == NW - 1) {
      const T d0 = piv[lane], d1 = piv[lane + 64];
      double lg = log((double)d0) + log((double)d1);
      int bad = !(d0 > T(0)) ? lane : (!(d1 > T(0)) ? lane + 64 : INT_MAX);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        lg += __shfl_xor(lg, o);
        bad = min(bad, __shfl_xor(bad, o));
      }
      if (lane == 0) {
        atomicAdd(logdet, lg);
        if (bad != INT_MAX) atomicMin(info, (int)(j0 + bad + 1));
      }
    }
    // L home, lower triangle only: whole 16-byte vectors left of the diagonal, predicated elements across it
#pragma unroll 4
    for (int idx = tid; idx < PB * RV; idx += NT) {
      const int r = idx / RV, c = (idx % RV) * VEC;
      if (c > r) continue;
      const vec_t t = *reinterpret_cast<const vec_t*>(&S[r * LD + c]);
      T* dst = &a[(j0 + r) * lda + j0 + c];
      if (c + VEC - 1 <= r) {
        *reinterpret_cast<vec_t*>(dst) = t;
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          if (c + e <= r) dst[e] = t[e];
      }
    }
    PT_MARK(4);
  } else {
    if (tid < PB) rdiag[tid] = T(1) / S[tid * LD + tid];
    __syncthreads();
  }

  // ---- X = L^-1 in place.  16x16 diagonal blocks: thread (b, c) solves L_bb x = e_c for column c of block b; the whole
  // column is written, zeros above the diagonal included (the doubling passes read these blocks as dense tiles).
  {
    T x[CB];
    const int b = tid >> 4, c = tid & 15;
    if (tid < PB) {
      const T* lb = &S[(CB * b) * LD + CB * b];
#pragma unroll
      for (int i = 0; i < CB; ++i) {
        T acc = (i == c) ? T(1) : T(0);
#pragma unroll
        for (int k = 0; k < i; ++k) acc = fma(-lb[i * LD + k], x[k], acc);
        x[i] = acc * rdiag[CB * b + i];
      }
    }
    __syncthreads();
    if (tid < PB) {
#pragma unroll
      for (int i = 0; i < CB; ++i) S[(CB * b + i) * LD + CB * b + c] = x[i];
    }
    // the 16x16 tiles right of the diagonal tiles still hold factorisation scratch (possibly NaN): clear the ones the
    // doubling passes read as parts of X11 / X22 (tile (bi, bj), bi < bj, inside one 64x64 diagonal block)
    for (int idx = tid; idx < PB * (PB / VEC); idx += NT) {
      const int r = idx / (PB / VEC), cv = (idx % (PB / VEC)) * VEC;
      if ((cv / CB) > (r / CB) && (cv / 64) == (r / 64)) {
        vec_t z;
#pragma unroll
        for (int e = 0; e < VEC; ++e) z[e] = T(0);
        *reinterpret_cast<vec_t*>(&S[r * LD + cv]) = z;
      }
    }
    __syncthreads();
  }
  PT_MARK(5);
  // Doubling passes.  Level h: pairs p, r0 = 2hp; block B = S[r0+h : r0+2h, r0 : r0+h].
  //   pass A:  B <- B X11     (X11 = S[r0 : r0+h, r0 : r0+h])
  //   pass B:  B <- -X22 B    (X22 = S[r0+h : r0+2h, r0+h : r0+2h])
  // X11 / X22 are lower triangular; their 16x16 tiles right of the diagonal tile (factorisation scratch until now) were
  // cleared above, so both passes run the FULL k range: uniform trip counts, every load of a k-step issued for all of a
  // wave's tiles before its MFMAs (the triangular k ranges differ per tile and left one dependent LDS round trip in front
  // of every MFMA: 7.5 us for the three levels; this form: half the MFMAs are multiplications by zero and it is 2x faster).
  // Output tiles (16x16) are dealt round-robin to the waves, at most 4 per wave (h = 64: 16 tiles), kept in accumulators
  // until every wave has finished reading the block.
  auto doubling_level = [&](auto hc) {
    constexpr int h = decltype(hc)::value;
    constexpr int th = h / CB;                       // 16-tiles per block edge
    constexpr int ntile = (PB / (2 * h)) * th * th;  // output tiles of this level: 4, 8, 16
    constexpr int NU = (ntile + NW - 1) / NW;        // tiles per wave: 1, 2, 4
    const int g = lane >> 4, l15 = lane & 15;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      typename M::acc_t acc[NU];
      const T* pa[NU];
      const T* pb[NU];
      int orow[NU], ocol[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const int tl = wave + u * NW;                // ntile is a multiple of NW: every wave owns exactly NU tiles
        const int p = tl / (th * th), mt = (tl % (th * th)) / th, nt = tl % th;
        const int r0 = 2 * h * p;
        orow[u] = r0 + h + CB * mt;
        ocol[u] = r0 + CB * nt;
        if (pass == 0) {   // out[m][n] = sum_k B[m][k] X11[k][n]
          pa[u] = &S[(orow[u] + l15) * LD + r0 + g];
          pb[u] = &S[(r0 + g) * LD + ocol[u] + l15];
        } else {           // out[m][n] = -sum_k X22[m][k] B[k][n]
          pa[u] = &S[(orow[u] + l15) * LD + r0 + h + g];
          pb[u] = &S[(r0 + h + g) * LD + ocol[u] + l15];
        }
#pragma unroll
        for (int i = 0; i < M::ACC; ++i) acc[u][i] = T(0);
      }
#pragma unroll
      for (int k = 0; k < h; k += 4) {
        T av[NU], bv[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          av[u] = pa[u][k];
          bv[u] = pb[u][k * LD];
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) M::mma1(acc[u], av[u], bv[u]);
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int i = 0; i < M::ACC; ++i)
          S[(orow[u] + M::acc_row(lane, i)) * LD + ocol[u] + M::acc_col(lane)] = pass == 0 ? acc[u][i] : -acc[u][i];
      __syncthreads();
    }
  };
  static_assert(NW == 4, "tile dealing of the doubling passes assumes four waves");
  doubling_level(std::integral_constant<int, 16>{});
  doubling_level(std::integral_constant<int, 32>{});
  doubling_level(std::integral_constant<int, 64>{});
  PT_MARK(6);
  // X out: dense 128 x 128, zeros above the diagonal (the LDS image still holds factorisation scratch there)
  for (int idx = tid; idx < PB * RV; idx += NT) {
    const int r = idx / RV, c = (idx % RV) * VEC;
    vec_t t = *reinterpret_cast<const vec_t*>(&S[r * LD + c]);
#pragma unroll
    for (int e = 0; e < VEC; ++e)
      if (c + e > r) t[e] = T(0);
    *reinterpret_cast<vec_t*>(&linv_out[r * PB + c]) = t;
  }
#ifdef SMN_PANEL_TIMING
  PT_MARK(7);
  if (tid == 0 && j0 == 0)
    printf("potrf (100 MHz ticks): stage_in %lld  mfma_blocks %lld  dots %lld  factor+solve %lld  store_L %lld  inv16 %lld  doubling %lld  store_X %lld\n",
           pt_acc[0], pt_acc[1], pt_acc[2], pt_acc[3], pt_acc[4], pt_acc[5], pt_acc[6], pt_acc[7]);
#endif
}

// TRSM of one sub-panel as a GEMM: rows [rbeg + BM * blockIdx.x, +BM) of columns [j0, j0+128) <- rows * X^T with
// X = L_jj^-1 (dense, from potrf_kernel).  In place: a workgroup owns its rows over ALL 128 columns and the tile
// engine has read every operand byte before its epilogue stores.  64-row tiles: the K = 128 product of a tile is 8k
// MFMA cycles (3.7 us) and a sub-panel at N = 16384 is 256 workgroups of 48 KB of LDS.
template <typename T, int BM>
__global__ void __launch_bounds__(256, 2) trsm_kernel(T* __restrict__ a, int64_t lda, int64_t j0, int64_t rbeg,
                                                       const T* __restrict__ linv, int64_t id0, int64_t id1) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using Tile = TileNT<T, BM, PB, 2>;
  using M = typename Tile::M;
  const int64_t row0 = rbeg + (int64_t)blockIdx.x * BM;
  // identity rows that start right of this sub-panel are zero in its columns and stay zero
  if (id0 >= 0 && row0 >= id0 && row0 + BM <= id1 && row0 - id0 >= j0 + PB) return;
  Tile t;
  t.zero();
  t.template mainloop<0>(a + row0 * lda + j0, lda, linv, PB, PB, smem);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wr = wave >> 1, wc = wave & 1;
#pragma unroll
  for (int m = 0; m < Tile::MT; ++m)
#pragma unroll
    for (int n = 0; n < Tile::NT; ++n)
#pragma unroll
      for (int i = 0; i < M::ACC; ++i) {
        const int64_t gr = row0 + wr * Tile::WM + m * M::TM + M::acc_row(lane, i);
        const int64_t gc = j0 + wc * Tile::WN + n * M::TN + M::acc_col(lane);
        a[gr * lda + gc] = t.acc[m][n][i];
      }
}

// ---------------------------------------------------------------- fused panel (POTRF + TRSM in one launch)
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
template <typename T>
constexpr size_t panel_lds_bytes(int xr = PanelCfg<T>::XR) {
  return sizeof(T) * ((size_t)(PB + xr) * PanelCfg<T>::LD + MP * MP + PB);
}
constexpr int panel_threads(int xr) { return (PB + xr + 63) / 64 * 64; }   // one thread per LDS row, whole waves

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
// prefactored != 0: the diagonal block already holds L (solve only; used by smn_trsm).
// XRV = rows below the diagonal block carried per workgroup.  The default (128 in f32) minimises the number of
// workgroups that each redo the diagonal factorisation; the f32 64-row form is launched when the panel has few
// row blocks anyway (late, chain-bound super-panels): the MFMA block updates of a workgroup shrink by a quarter.
template <typename T, int XRV = PanelCfg<T>::XR>
__global__ void __launch_bounds__(panel_threads(XRV)) panel_kernel(T* __restrict__ a, int64_t lda, int64_t j0,
                                                                     int64_t rbeg, int64_t n_total, int prefactored,
                                                                     double* __restrict__ logdet,
                                                                     int* __restrict__ info, T* __restrict__ ldiag_out,
                                                                     int64_t id0, int64_t id1) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int XR = XRV, NT = panel_threads(XRV), LD = PanelCfg<T>::LD;
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
  // identity rows [id0, id1) (id0 < 0: none): a tile whose rows lie inside them and start right of the K range
  // multiplies zeros -- the workgroup leaves at once
  int64_t id0, id1;
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
  if (u.id0 >= 0 && row0 >= u.id0 && row0 + BM <= u.id1 && row0 - u.id0 >= u.k0 + u.K) return;
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
  if (lower == 2 && tiles_n >= tiles_m) {   // a trapezoid as wide as it is tall is the triangle
    lower = 1;
    tiles_n = tiles_m;
  }
  UpdArgs<T> u{a, lda, r0, c0, k0, (int)K, (int)tiles_n, lower, 0, TileMap::make(tiles_m, tiles_n, lower == 1),
               ctx->chol_id0, ctx->chol_id1};
  int64_t nt = lower == 1   ? tiles_m * (tiles_m + 1) / 2
               : lower == 2 ? tiles_n * (tiles_n + 1) / 2 + (tiles_m - tiles_n) * tiles_n
                            : tiles_m * tiles_n;
  ctx->prof_flops[tag ? PROF_TRAIL : PROF_STRIP] += 2.0 * kTile * kTile * (double)K * (double)nt;   // executed: whole tiles
  if (ctx->xcd_map && nt >= 512 && lower != 2) {   // small launches do not fill the XCDs anyway
    u.use_map = 1;
    nt = u.map.grid;
  }
  const size_t lds = MainTile<T>::LDS_BYTES;
  if constexpr (sizeof(T) == 4) {
    // CUs this stream may use: the bulk stream of the look-ahead is masked off the chain's CUs
    const int cus = (st == ctx->stream_bulk && st != nullptr) ? ctx->num_cu - ctx->chain_cus : ctx->num_cu;
    if (tag == 1 && lower && !u.use_map && ctx->persistent_trail && nt > 2 * cus && K <= ctx->persist_max_k &&
        ctx->chol_id0 < 0) {   // the persistent walk has no tile skipping
      // persistent walk over the lower tiles, two workgroups per CU
      const size_t plds = TileNT<T, kTile, kTile, 2>::LDS_BYTES;
      ProfScope ps(ctx, PROF_TRAIL, st);
      hipLaunchKernelGGL(trail_kernel<T>, dim3((unsigned)(2 * cus)), dim3(256), plds, st, u, (int)nt);
      SMN_CHECK_LAUNCH(ctx);
      return SMN_OK;
    }
  }
  if (!u.use_map && nt <= ctx->quarter_tile_max) {
    // very few tiles (strips; the updates of the late, chain-bound blocks): 64x64 tiles, four workgroups per tile
    ProfScope ps(ctx, tag ? PROF_TRAIL : PROF_STRIP, st);
    constexpr size_t qlds = TileNT<T, 64, 64, SMN_STAGES>::LDS_BYTES;
    if (tag) {
      auto kern = update_kernel<T, 1, 64, 64>;
      hipLaunchKernelGGL(kern, dim3((unsigned)(4 * nt)), dim3(256), qlds, st, u);
    } else {
      auto kern = update_kernel<T, 0, 64, 64>;
      hipLaunchKernelGGL(kern, dim3((unsigned)(4 * nt)), dim3(256), qlds, st, u);
    }
    SMN_CHECK_LAUNCH(ctx);
    return SMN_OK;
  }
  if (!u.use_map && nt <= ctx->half_tile_max) {
    // too few 128x128 tiles to fill the chip: 64-row tiles, twice as many workgroups
    const int64_t nh = lower == 1   ? tiles_m * (tiles_m + 1)
                       : lower == 2 ? tiles_n * (tiles_n + 1) + 2 * (tiles_m - tiles_n) * tiles_n
                                    : 2 * tiles_m * tiles_n;
    ProfScope ps(ctx, tag ? PROF_TRAIL : PROF_STRIP, st);
    constexpr size_t hlds = TileNT<T, 64, kTile, SMN_STAGES>::LDS_BYTES;
    if (tag) {
      auto kern = update_kernel<T, 1, 64>;
      hipLaunchKernelGGL(kern, dim3((unsigned)nh), dim3(256), hlds, st, u);
    } else {
      auto kern = update_kernel<T, 0, 64>;
      hipLaunchKernelGGL(kern, dim3((unsigned)nh), dim3(256), hlds, st, u);
    }
    SMN_CHECK_LAUNCH(ctx);
    return SMN_OK;
  }
  {
    ProfScope ps(ctx, tag ? PROF_TRAIL : PROF_STRIP, st);
    if (tag) {
      auto kern = update_kernel<T, 1>;
      hipLaunchKernelGGL(kern, dim3((unsigned)nt), dim3(256), lds, st, u);
    } else {
      auto kern = update_kernel<T, 0>;
      hipLaunchKernelGGL(kern, dim3((unsigned)nt), dim3(256), lds, st, u);
    }
  }
  SMN_CHECK_LAUNCH(ctx);
  return SMN_OK;
}

constexpr int kTrsmBM = 64;   // rows per TRSM workgroup

// One sub-panel: the diagonal workgroup (POTRF + inverse), then the rows below as a GEMM against the inverse.
// The inverses live in workspace slot 3, one 128x128 block per sub-panel (nothing reads them after the TRSM of their
// sub-panel, but a ring would have to be fenced against the look-ahead; n_factor x 128 elements is small).
template <typename T, int XRV>
int launch_fused_x(smn_ctx* ctx, hipStream_t st, T* a, int64_t lda, int64_t j0, int64_t n_total) {
  const int64_t rbeg = j0 + PB;
  const int64_t below = n_total - rbeg;
  const unsigned grid = below > 0 ? (unsigned)((below + XRV - 1) / XRV) : 1u;
  T* ldiag = reinterpret_cast<T*>(ctx->ws[3]) + (j0 / PB) * (int64_t)(PB * PB);   // L_kk goes home after the last panel
  hipLaunchKernelGGL((panel_kernel<T, XRV>), dim3(grid), dim3(panel_threads(XRV)), panel_lds_bytes<T>(XRV), st, a, lda, j0,
                     rbeg, n_total, 0, ctx->d_scal, ctx->d_info, ldiag, ctx->chol_id0, ctx->chol_id1);
  return SMN_OK;
}

template <typename T>
int launch_panel(smn_ctx* ctx, hipStream_t st, T* a, int64_t lda, int64_t j0, int64_t rbeg, int64_t n_total, int prefactored) {
  T* linv = reinterpret_cast<T*>(ctx->ws[3]) + (j0 / PB) * (int64_t)(PB * PB);
  const int64_t below = n_total - rbeg;
  ProfScope ps(ctx, PROF_PANEL, st);
  if (!prefactored && rbeg == j0 + PB && below <= ctx->fused_rows) {
    // few rows left: the fused panel (every workgroup re-factors the diagonal block and carries its own rows through the
    // same column operations) is one launch and needs no inverse; its 135 KB workgroups fit the reserved CUs by now
    if constexpr (sizeof(T) == 4) {
      if (below <= 2048) SMN_TRY((launch_fused_x<T, 64>(ctx, st, a, lda, j0, n_total)));
      else SMN_TRY((launch_fused_x<T, 128>(ctx, st, a, lda, j0, n_total)));
    } else {
      SMN_TRY((launch_fused_x<T, 16>(ctx, st, a, lda, j0, n_total)));
    }
    if (ctx->fused_from < 0 || j0 < ctx->fused_from) ctx->fused_from = j0;
    SMN_CHECK_LAUNCH(ctx);
    return SMN_OK;
  }
  hipLaunchKernelGGL(potrf_kernel<T>, dim3(1), dim3(DiagCfg<T>::THREADS), DiagCfg<T>::LDS, st, a, lda, j0, prefactored,
                     ctx->d_scal, ctx->d_info, linv);
  if (below > 0) {
    constexpr size_t tlds = TileNT<T, kTrsmBM, PB, 2>::LDS_BYTES;
    hipLaunchKernelGGL((trsm_kernel<T, kTrsmBM>), dim3((unsigned)(below / kTrsmBM)), dim3(256), tlds, st, a, lda, j0, rbeg,
                       static_cast<const T*>(linv), ctx->chol_id0, ctx->chol_id1);
    ctx->prof_flops[PROF_PANEL] += 2.0 * PB * PB * (double)below;
  }
  SMN_CHECK_LAUNCH(ctx);
  return SMN_OK;
}

template <typename T>
int set_lds_attrs(smn_ctx* ctx) {
  bool& done = ctx->lds_attrs_done[sizeof(T) == 8 ? 1 : 0];   // per context (= per device) and dtype
  if (done) return SMN_OK;
  SMN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(potrf_kernel<T>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)DiagCfg<T>::LDS));
  if constexpr (sizeof(T) == 4) {
    SMN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(panel_kernel<T, 128>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)panel_lds_bytes<T>(128)));
    SMN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(panel_kernel<T, 64>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)panel_lds_bytes<T>(64)));
  } else {
    SMN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(panel_kernel<T, 16>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)panel_lds_bytes<T>(16)));
  }
  SMN_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(trsm_kernel<T, kTrsmBM>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(TileNT<T, kTrsmBM, PB, 2>::LDS_BYTES)));
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

// The chain of one column block [c0, c1): sub-panels of 128 columns in pairs (outer panels of 256).  Inside the
// block the updates stay inside it: the strip brings the second sub-panel of a pair up to date (K = 128), the near
// update brings the rest of the block up to date with the finished pair (K = 256, a lower trapezoid).
template <typename T>
int chain_block(smn_ctx* ctx, hipStream_t st, T* a, int64_t lda, int64_t c0, int64_t c1, int64_t n_total) {
  constexpr int64_t W = 2 * PB;
  for (int64_t j0 = c0; j0 < c1; j0 += W) {
    const int64_t w = (c1 - j0 < W) ? c1 - j0 : W;
    for (int64_t js = j0; js < j0 + w; js += PB) {
      if (js > j0)
        SMN_TRY(launch_update<T>(ctx, st, a, lda, js, js, j0, js - j0, (n_total - js) / kTile, 1, 0));
      SMN_TRY(launch_panel<T>(ctx, st, a, lda, js, js + PB, n_total, 0));
    }
    const int64_t j1 = j0 + w;
    if (j1 < c1)
      SMN_TRY(launch_update<T>(ctx, st, a, lda, j1, j1, j0, w, (n_total - j1) / kTile, (c1 - j1) / kTile, 2));
  }
  return SMN_OK;
}

// Right-looking inside a window, left-looking beyond it.  The columns are cut into blocks of S (the appended
// rows' own columns [n_factor, n_total) continue the same grid; they are only ever updated).  After the chain of
// block s three updates bring the blocks to its right up to date:
//   B1  block s+1                 with block s (K = S): what the next chain waits for;
//   B2  blocks s+2 .. s+D         with block s (K = S);
//   B3  block s+D+1               with ALL finished columns [0, end of s) at once (K = (s+1) S): the first time this
//                                 block is touched -- one long-K product instead of s+1 short ones, and each of its
//                                 tiles is read and written once for them.
// With the look-ahead (n_total >= chain_min_n) B1..B3 go to the CU-masked bulk stream and chain(s+1) starts on the
// caller's (high-priority) stream as soon as B1 is done: the chain always finds the reserved CUs free and runs beside
// B2, B3.  Pure right-looking (D = infinity) leaves the bulk stream without work while the last third of the chains
// run (round 1: the GPU idled for 4 of 16 ms at N = 16384); pure left-looking (D = 0) leaves it without work during
// the first third.  The window keeps it busy at both ends.
template <typename T>
int cholesky_t(smn_ctx* ctx, T* a, int64_t n_total, int64_t n_factor, int64_t lda, int64_t n_shift, double jitter_abs,
               double ridge_rel) {
  SMN_TRY(set_lds_attrs<T>(ctx));
  void* side = nullptr;   // inverses of the diagonal blocks, [n_factor/128][128*128]
  SMN_TRY(smn_workspace(ctx, 3, sizeof(T) * (size_t)n_factor * PB, &side));
  hipStream_t st = ctx->stream;
  hipLaunchKernelGGL(init_scalars_kernel, dim3(1), dim3(1), 0, st, ctx->d_scal, ctx->d_info);
  if (n_shift > 0 && (jitter_abs != 0.0 || ridge_rel != 0.0)) {
    if (ridge_rel != 0.0)
      hipLaunchKernelGGL(diag_trace_kernel<T>, dim3(1), dim3(256), 0, st, a, lda, n_shift, ctx->d_scal + 1);
    hipLaunchKernelGGL(diag_shift_kernel<T>, dim3((unsigned)((n_shift + 255) / 256)), dim3(256), 0, st, a, lda,
                       n_shift, jitter_abs, ridge_rel, ctx->d_scal + 1);
  }
  SMN_CHECK_LAUNCH(ctx);
  constexpr int64_t W = 2 * PB;
  int64_t S = ctx->super_panel / W * W;
  if (S < W) S = W;
  const int64_t D = ctx->window < 1 ? 1 : ctx->window;
  // block boundaries: 0, S, 2S, .. n_factor, then n_factor + S, .. n_total
  std::vector<int64_t> cb;
  for (int64_t x = 0; x < n_factor; x += S) cb.push_back(x);
  cb.push_back(n_factor);
  const int nb = (int)cb.size() - 1;            // factored blocks
  for (int64_t x = n_factor + S; x < n_total; x += S) cb.push_back(x);
  if (n_total > n_factor) cb.push_back(n_total);
  const int nc = (int)cb.size() - 1;            // all blocks
  hipStream_t sb = (n_total >= ctx->chain_min_n && ctx->stream_bulk) ? ctx->stream_bulk : nullptr;
  // rows >= cb[lo] of blocks lo..hi (clipped) -= A[rows, k0:k1] A[cols, k0:k1]^T
  auto upd = [&](hipStream_t s_, int lo, int hi, int64_t k0, int64_t k1) -> int {
    if (lo >= nc || hi < lo) return SMN_OK;
    if (hi >= nc) hi = nc - 1;
    const int64_t r0 = cb[lo];
    const int64_t mk = ctx->max_k < S ? S : ctx->max_k / S * S;
    for (int64_t k = k0; k < k1; k += mk)
      SMN_TRY(launch_update<T>(ctx, s_, a, lda, r0, r0, k, (k1 - k < mk ? k1 - k : mk), (n_total - r0) / kTile,
                               (cb[hi + 1] - r0) / kTile, 2));
    return SMN_OK;
  };
  bool bulk_used = false;
  int rc = SMN_OK;
  ctx->fused_from = -1;
  for (int s = 0; s < nb && rc == SMN_OK; ++s) {
    if (sb && s > 0) rc = hipStreamWaitEvent(st, ctx->ev_b, 0) == hipSuccess ? SMN_OK : SMN_EHIP;   // B1(s-1)
    if (rc == SMN_OK) rc = chain_block<T>(ctx, st, a, lda, cb[s], cb[s + 1], n_total);
    if (rc != SMN_OK || s + 1 >= nc) break;
    hipStream_t bs = sb ? sb : st;
    if (sb) {
      if (hipEventRecord(ctx->ev_a, st) != hipSuccess || hipStreamWaitEvent(sb, ctx->ev_a, 0) != hipSuccess) rc = SMN_EHIP;
      bulk_used = true;
    }
    if (rc == SMN_OK) rc = upd(bs, s + 1, s + 1, cb[s], cb[s + 1]);                          // B1
    if (rc == SMN_OK && sb && hipEventRecord(ctx->ev_b, sb) != hipSuccess) rc = SMN_EHIP;
    if (rc == SMN_OK) rc = upd(bs, s + 2, s + (int)D, cb[s], cb[s + 1]);                     // B2
    if (rc == SMN_OK) rc = upd(bs, s + (int)D + 1, s == nb - 1 ? nc - 1 : s + (int)D + 1, 0, cb[s + 1]);   // B3
  }
  // The caller's stream continues after the bulk stream whatever happened above: an error exit must not leave bulk
  // work running behind a workspace the caller is about to free or reuse.
  if (bulk_used) {
    if (hipEventRecord(ctx->ev_a, sb) != hipSuccess || hipStreamWaitEvent(st, ctx->ev_a, 0) != hipSuccess) {
      (void)hipStreamSynchronize(sb);
      if (rc == SMN_OK) rc = SMN_EHIP;
    }
  }
  if (rc == SMN_OK && ctx->fused_from >= 0) {   // the fused panels left their L_kk in the side buffer
    const int64_t f = ctx->fused_from;
    hipLaunchKernelGGL(copy_diag_kernel<T>, dim3((unsigned)((n_factor - f) / PB)), dim3(1024), 0, st, a + f * lda + f, lda,
                       static_cast<const T*>(side) + (f / PB) * (int64_t)(PB * PB));
    if (hipGetLastError() != hipSuccess) rc = SMN_EHIP;
  }
  if (rc == SMN_EHIP && ctx->err.empty()) ctx->err = "cholesky: stream / event call failed";
  return rc;
}

}  // namespace

int cholesky_padded(smn_ctx* ctx, int dtype, void* a, int64_t n_total, int64_t n_factor, int64_t lda, int64_t n_shift,
                    double jitter_abs, double ridge_rel, int64_t id0, int64_t id1) {
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
                                          ridge_rel)
                     : cholesky_t<float>(ctx, static_cast<float*>(a), n_total, n_factor, lda, n_shift, jitter_abs, ridge_rel);
  ctx->chol_id0 = ctx->chol_id1 = -1;
  return rc;
}

// Solve-only sweep: rows [n_factor, n_total) of `a` <- rows * L^-T with L = the (already factored)
// leading block.  Left-looking over block columns; used by smn_trsm.
namespace {
template <typename T>
int solve_rows_t(smn_ctx* ctx, T* a, int64_t n_total, int64_t n_factor, int64_t lda) {
  SMN_TRY(set_lds_attrs<T>(ctx));
  void* side = nullptr;
  SMN_TRY(smn_workspace(ctx, 3, sizeof(T) * (size_t)n_factor * PB, &side));
  hipStream_t st = ctx->stream;
  for (int64_t js = 0; js < n_factor; js += PB) {
    // appended rows only: C = a[n_factor:, js:js+128] -= a[n_factor:, 0:js] * a[js:js+128, 0:js]^T
    SMN_TRY(launch_update<T>(ctx, st, a, lda, n_factor, js, 0, js, (n_total - n_factor) / kTile, 1, 0));
    // inverse of the prefactored diagonal block, then the appended rows against it
    SMN_TRY(launch_panel<T>(ctx, st, a, lda, js, n_factor, n_total, 1));
  }
  return SMN_OK;
}
}  // namespace

int solve_rows_padded(smn_ctx* ctx, int dtype, void* a, int64_t n_total, int64_t n_factor, int64_t lda) {
  if (n_total % kTile || n_factor % kTile) return smn_fail(ctx, SMN_EINVAL, "solve_rows_padded: padding");
  return dtype == SMN_F64 ? solve_rows_t<double>(ctx, static_cast<double*>(a), n_total, n_factor, lda)
                          : solve_rows_t<float>(ctx, static_cast<float*>(a), n_total, n_factor, lda);
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

int fetch_logdet_info(smn_ctx* ctx, double* logdet, int* info) {
  return fetch_results(ctx, nullptr, 0, nullptr, logdet, info);
}
