// potrf_tile.hpp — the diagonal step of one 128-column sub-panel as a device function of ONE 256-thread workgroup:
// POTRF of the 128x128 block in LDS, then its inverse in place (recursive doubling on the 16x16x4 MFMA).  Used by the
// dataflow Cholesky (chol_dataflow.hip), where the rows below are solved as GEMM tiles against the inverse.  Every global
// store is a write-through (sc1) store: the tile is handed to other workgroups inside the same launch
// (cdna_hip_programming.md Guideline 16, R1).
#pragma once
#include <climits>
#include <type_traits>

#include "gemm_nt.hpp"

namespace potrf_detail {

constexpr int PB = 128;  // diagonal block edge == GEMM tile edge
constexpr int MP = 8;    // micro-panel width of the in-LDS factorisation

// write-through stores (global_store ... sc1): visible to other XCDs once the storing wave has drained them
__device__ __forceinline__ void store_wt(float* p, float v) {
  __hip_atomic_store(reinterpret_cast<unsigned*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void store_wt(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}

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

#ifdef SMN_PANEL_TIMING_DF
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
__device__ __forceinline__ void potrf_tile(T* __restrict__ a, int64_t lda, int64_t j0, int prefactored,
                                           double* __restrict__ logdet, int* __restrict__ info,
                                           T* __restrict__ linv_out, char* smem) {
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
#pragma unroll
      for (int e = 0; e < VEC; ++e)
        if (c + e <= r) store_wt(dst + e, t[e]);
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
#pragma unroll
    for (int e = 0; e < VEC; ++e) store_wt(&linv_out[r * PB + c + e], t[e]);
  }

}

}  // namespace potrf_detail
