// cnn.hip — conv-NNGP kernel of experiments/nt_kernels.py:34-45:
//     L x [Conv(1 ch, 3x3, stride 1, SAME, W_std=w, b_std=b); act];  Flatten;  Dense(last_w, b=0)
// (neural_tangents stax.Conv / Flatten).  Because there is no pooling before
// Flatten only same-pixel covariances are needed: for an image pair (n, m) the state is one H x W map
//     K0[h,w] = sum_c x1[n,h,w,c] x2[m,h,w,c] / C
//     K <- w^2 * (3x3 zero-padded box SUM of K) / 9 + b^2 ;  K <- act(K; q1[n,h,w], q2[m,h,w])   (per pixel)
//     out[n,m] = last_w^2 * mean_hw K
// The N^2 * H * W per-pixel kernel entries can never be materialised (C3: 819 GB in fp64), so each
// wave carries one pair's map through all layers on chip: the map lives in LDS with a zero halo
// (ping-pong between layers), lanes own pixels, the 3x3 stencil is 9 LDS reads.  The per-image
// pre-activation variance maps q~_l[n,h,w] (the same stencil recursion on the diagonal) are computed
// once per image by conv_q_kernel and streamed from L2.  VALU-bound by construction (a 9-tap sum and
// an asin per pixel, pair and layer); no MFMA: there is no GEMM here to find.
#include <type_traits>

#include "internal.hpp"
#include "nngp_math.hpp"

namespace {

template <typename T>
__device__ __forceinline__ T rsqrt_any(T x);
template <>
__device__ __forceinline__ float rsqrt_any<float>(float x) { return __builtin_amdgcn_rsqf(x); }
template <>
__device__ __forceinline__ double rsqrt_any<double>(double x) { return 1.0 / sqrt(x); }
template <typename T>
__device__ __forceinline__ T rcp_any(T x);
template <>
__device__ __forceinline__ float rcp_any<float>(float x) { return __builtin_amdgcn_rcpf(x); }
template <>
__device__ __forceinline__ double rcp_any<double>(double x) {   // v_rcp_f64 + two Newton steps: full double precision
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  return fma(fma(-x, r, 1.0), r, r);
}

struct ConvProg {
  int act, layers, H, W, C;
  double w2, b2, lw2;
};

// One workgroup per image: Q[img][l][p] = pre-activation variance of layer l at pixel p,
// diag[img] = last_w^2 * mean_p q_L (the exact K(img, img)).
// R is what the pair kernel multiplies (like the r table of the MLP build); ONE table, because the pair kernel is
// bound by streaming these tables out of L2, not by its arithmetic:
//   ReLU: r = 1/sqrt(q~) (0 where q~ <= 0); the second factor s_i s_j = sqrt(q_i q_j)/(2 pi) = 1/(2 pi r_i r_j)
//   erf:  r = 1/sqrt(1 + 2 q~)
// (A second table s = sqrt(q~ / 2 pi), so that s_i s_j is one multiplication instead of 1 / (2 pi r_i r_j) with a reciprocal, a
// Newton step and a select, was tried in round 3: 6 vector instructions fewer per pixel and layer, but 64 more registers for the
// table vectors of a layer -- 2 waves per SIMD instead of 3 -- and the kernel build went from 572 to 1003 ms.  Not kept.)
// patch44 != 0 (32x32 images, conv_pair44_kernel): R is written in the order that kernel's lanes read it -- for layer l the
// vectors [j = 4-pixel patch row r x vector v][lane = patch (py, px)][16 bytes], so one load instruction of a wave is 1 KB of
// consecutive bytes -- and xperm receives the image in the same order ([vector of a patch row][lane][16 bytes], a patch
// row being 4 pixels x C channels).  Gathered patch rows straight from the [pixel][channel] image cost the pair kernel 44 %
// of its time (6x the cache-line look-ups: profiles/r03_cnn_phase_timing.txt).
template <typename T>
__device__ __forceinline__ int64_t patch44_index(int px, int vec_elems) {   // pixel -> element index inside one layer's table
  const int row = px >> 5, col = px & 31;
  const int lane = (row >> 2) * 8 + (col >> 2), r = row & 3, c = col & 3;
  const int rv = 4 / vec_elems;                                              // vectors per patch row
  return ((int64_t)(r * rv + c / vec_elems) * 64 + lane) * vec_elems + c % vec_elems;
}
template <typename T>
__global__ void __launch_bounds__(256) conv_q_kernel(const T* __restrict__ x, int64_t n, ConvProg p,
                                                     T* __restrict__ R, T* __restrict__ diag, int patch44, T* __restrict__ xperm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int H = p.H, W = p.W, HW = H * W, PW = W + 2, PSZ = (H + 2) * PW;
  double* m0 = reinterpret_cast<double*>(smem);   // padded map, double for the diagonal
  double* m1 = m0 + PSZ;
  const int64_t img = blockIdx.x;
  for (int i = threadIdx.x; i < 2 * PSZ; i += blockDim.x) m0[i] = 0.0;
  __syncthreads();
  for (int px = threadIdx.x; px < HW; px += blockDim.x) {
    const T* xp = x + (img * HW + px) * p.C;
    double s = 0.0;
    for (int c = 0; c < p.C; ++c) s += (double)xp[c] * (double)xp[c];
    m0[(px / W + 1) * PW + px % W + 1] = s / p.C;
  }
  __syncthreads();
  double* cur = m0;
  double* nxt = m1;
  for (int l = 0; l < p.layers; ++l) {
    for (int px = threadIdx.x; px < HW; px += blockDim.x) {
      const int h = px / W, w = px % W;
      const double* c = cur + h * PW + w;   // top-left of the 3x3 window in the padded map
      const double bs = c[0] + c[1] + c[2] + c[PW] + c[PW + 1] + c[PW + 2] + c[2 * PW] + c[2 * PW + 1] + c[2 * PW + 2];
      const double qt = p.w2 * bs / 9.0 + p.b2;
      const int64_t ti = (img * p.layers + l) * HW + (patch44 ? patch44_index<T>(px, 16 / (int)sizeof(T)) : (int64_t)px);
      if (p.act == 0) R[ti] = qt > 0.0 ? (T)(1.0 / sqrt(qt)) : T(0);
      else R[ti] = (T)(1.0 / sqrt(1.0 + 2.0 * qt));
      const double qa = p.act == 0 ? 0.5 * qt : (2.0 / nngp::kPi) * asin(2.0 * qt / (1.0 + 2.0 * qt));
      nxt[(h + 1) * PW + w + 1] = qa;
    }
    __syncthreads();
    double* t = cur; cur = nxt; nxt = t;
  }
  if (patch44 && xperm) {
    // element e' = c4 * C + ch of patch row r of lane `lane` lives in vector (r * XV + e' / VEC), slot e' % VEC
    constexpr int VEC = 16 / (int)sizeof(T);
    const int C = p.C, XV = 4 * C / VEC;
    for (int i = threadIdx.x; i < HW * C; i += blockDim.x) {
      const int px = i / C, ch = i % C;
      const int row = px >> 5, col = px & 31;
      const int lane = (row >> 2) * 8 + (col >> 2), r = row & 3, e = (col & 3) * C + ch;
      xperm[img * HW * C + ((int64_t)(r * XV + e / VEC) * 64 + lane) * VEC + e % VEC] = x[img * HW * C + i];
    }
  }
  // mean over pixels (block tree reduction in the free map)
  double s = 0.0;
  for (int px = threadIdx.x; px < HW; px += blockDim.x) s += cur[(px / W + 1) * PW + px % W + 1];
  double* red = m0 + 2 * PSZ;   // 256 doubles of scratch behind the two maps
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) diag[img] = (T)(p.lw2 * red[0] / HW);
}

template <typename T>
struct PairArgs {
  const T* x1; const T* x2; const T* R1; const T* R2; const T* diag;
  int64_t n1, n2; int symmetric, mirror;
  ConvProg prog;
  T* out; int64_t ldo; int64_t npairs;
  int tile_bn;   // > 0: XCD-tiled pair order (below), tiles of tile_bn x 32 image pairs; 0: plain strided order
};

constexpr int kMaxPix = 64;   // pixels per lane (H*W <= 4096)

// 4 waves per workgroup, each wave walks its own list of image pairs: the map of a wave is private to it,
// LDS operations of one wave execute in order, so the layers need no workgroup barrier (only a compiler fence).
// ONE padded map per wave: a lane keeps the values of its NP pixels in registers, publishes them to the map,
// reads the 9 taps of each of its pixels and overwrites the registers; the next layer's publish is issued after
// every tap read of this one (same wave, in-order LDS), so no second map is needed -- half the LDS of a ping-pong
// pair, twice the waves per CU to cover the latency of the table loads and of the f64 chains.
// NP = pixels per lane (compile-time bound): padded-map offsets are computed once per kernel, not per layer.
// Lanes whose pixel index runs past H*W are not branched around: they load from a clamped (valid) pixel, publish to
// a dummy slot behind the map whose 3x3 neighbourhood is also behind the map, and are dropped from the final sum
// (EXACT: H*W == 64*NP, there are none).
#ifndef SMN_CNN_K0_BATCH
#define SMN_CNN_K0_BATCH 16   // pixels whose K0 channel loads are issued together: the phase is pure load latency
#endif                        // (4 -> 16: fp64 +12 %, fp32 +5 %, profiles/r01f_cnn_k0_batch_ab.txt)
constexpr int KB0 = SMN_CNN_K0_BATCH;
#ifndef SMN_CNN_OCC_F32
#define SMN_CNN_OCC_F32 2   // f32 wants the registers (ILP over its pixels) more than the waves: 4 spills and loses 27 %
#endif
#ifndef SMN_CNN_OCC_F64
#define SMN_CNN_OCC_F64 4   // workgroups per CU the f64 form is compiled for (128 VGPRs, 9 spilled doubles; 2 / 3: -6 %)
#endif
// The pair list of one wave: plain strided order, or the XCD-tiled order described below.  next() is wave-uniform.
template <typename T>
struct PairWalk {
  const PairArgs<T>& a;
  bool tiled; int xcd, tidx; int64_t tiles_m, tiles_n, tn, tm, pr, step;
  __device__ __forceinline__ PairWalk(const PairArgs<T>& a_, int wave) : a(a_) {
    tiled = a.tile_bn > 0;
    xcd = (int)(blockIdx.x & 7);
    tidx = (int)(blockIdx.x >> 3) * 4 + wave;   // this wave's pair inside every tile of its XCD
    tiles_m = (a.n2 + 31) / 32;
    tiles_n = tiled ? (a.n1 + a.tile_bn - 1) / a.tile_bn : 0;
    tn = 0;
    tm = xcd - 8;
    step = (int64_t)gridDim.x * 4;
    pr = (int64_t)blockIdx.x * 4 + wave - step;
  }
  __device__ __forceinline__ int64_t row_tiles(int64_t r) const {   // tiles of tile row r that hold a wanted pair
    if (!a.symmetric) return tiles_m;
    const int64_t c = (r * a.tile_bn + a.tile_bn - 1) / 32 + 1;
    return c < tiles_m ? c : tiles_m;
  }
  __device__ __forceinline__ bool next(int64_t& n, int64_t& m) {
    if (tiled) {
      // XCD x walks the lower (or all) tiles with (tn + tm) % 8 == x, row by row: dealt round-robin inside a tile row
      // with the offset rotating from row to row, so every XCD gets the same share of the triangle
      for (;;) {
        tm += 8;
        while (tn < tiles_n && tm >= row_tiles(tn)) {
          ++tn;
          tm = (xcd - tn) & 7;
        }
        if (tn >= tiles_n) return false;
        n = tn * a.tile_bn + (tidx >> 5);
        m = tm * 32 + (tidx & 31);
        if (n < a.n1 && m < a.n2 && !(a.symmetric && m > n)) return true;
      }
    }
    pr += step;
    if (pr >= a.npairs) return false;
    if (a.symmetric) {
      int64_t r = (int64_t)((sqrt(8.0 * (double)pr + 1.0) - 1.0) * 0.5);
      while ((r + 1) * (r + 2) / 2 <= pr) ++r;
      while (r * (r + 1) / 2 > pr) --r;
      n = r;
      m = pr - r * (r + 1) / 2;
    } else {
      n = pr / a.n2;
      m = pr % a.n2;
    }
    return true;
  }
};

template <typename T>
constexpr int pair_occ(int np, bool exact) {   // workgroups per CU a form is compiled for (and launched at)
  return np > 16 ? 1 : np <= 4 ? 4 : !exact ? 2 : sizeof(T) == 8 ? SMN_CNN_OCC_F64 : SMN_CNN_OCC_F32;
}
// Pair order.  Every pair streams two images' inputs and factor tables (56 KB in f64 for 32x32x3, 4 layers); in the plain
// order the waves in flight touch ~8k different images, far beyond the 4 MB L2 of an XCD, and the f64 kernel spent 75 % of
// its wave cycles waiting on those loads (VALU busy 49 %, rocprofv3 PMC).  Tiled order: the grid is exactly the resident
// set, workgroup b runs on XCD b % 8 (round-robin dispatch), and the workgroups of one XCD walk tiles of tile_bn x 32
// pairs together -- one pair per wave per tile -- so an XCD's L2 holds the tile_bn + 32 images its waves are reading.
template <typename T, int ACT, int NP, bool EXACT>
__global__ void __launch_bounds__(256, pair_occ<T>(NP, EXACT)) conv_pair_kernel(PairArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const ConvProg& p = a.prog;
  const int H = p.H, W = p.W, HW = H * W, PW = W + 2, PSZ = (H + 2) * PW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int MSZ = PSZ + 2 * PW + 3;                     // map + the dummy slot's neighbourhood
  T* map = reinterpret_cast<T*>(smem) + (size_t)wave * MSZ;
  for (int i = lane; i < MSZ; i += 64) map[i] = T(0);   // halo stays zero for the whole kernel
  // centre of pixel lane + 64 i in the padded map.  EXACT forms (W divides 64 as well): pixel i sits 64 / W rows below
  // pixel i - 1, so the offsets are off0 + i * rstep and no per-pixel table is kept in registers.
  int off_tab[EXACT ? 1 : NP];
  const int off0 = (lane / W + 1) * PW + lane % W + 1, rstep = (64 / W) * PW;
  if (!EXACT) {
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int px = lane + 64 * i;
      off_tab[i] = px < HW ? (px / W + 1) * PW + px % W + 1 : PSZ + PW + 1;
    }
  }
  auto off = [&](int i) { return EXACT ? off0 + i * rstep : off_tab[EXACT ? 0 : i]; };
  auto pix = [&](int i) { return EXACT ? lane + 64 * i : min(lane + 64 * i, HW - 1); };   // pixel a lane loads from
  const T w2_9 = (T)(p.w2 / 9.0), b2 = (T)p.b2;
  const T inv_c = (T)(1.0 / p.C);
  // (the fp64 16-pixel exact form lives on 128 VGPRs: a batch of 16 spills there and costs 20 %)
  constexpr int KB = (sizeof(T) == 8 && EXACT && NP == 16) ? 4 : (KB0 < NP ? KB0 : NP);
  PairWalk<T> walk(a, wave);
  int64_t n, m;
  while (walk.next(n, m)) {
    // K0 map: channel loop outside, pixel loop inside, so the 2 NP loads of one channel are in flight together
    const T* xa = a.x1 + n * HW * p.C;
    const T* xb = a.x2 + m * HW * p.C;
    T val[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) val[i] = T(0);
    for (int c = 0; c < p.C; ++c) {
#pragma unroll
      for (int i0 = 0; i0 < NP; i0 += KB) {   // 2 KB loads in flight per batch
        T va[KB], vb[KB];
#pragma unroll
        for (int j = 0; j < KB; ++j) {
          va[j] = xa[pix(i0 + j) * p.C + c];
          vb[j] = xb[pix(i0 + j) * p.C + c];
        }
#pragma unroll
        for (int j = 0; j < KB; ++j) val[i0 + j] = fma(va[j], vb[j], val[i0 + j]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) val[i] *= inv_c;
    for (int l = 0; l < p.layers; ++l) {
      const T* r1 = a.R1 + (n * p.layers + l) * HW;
      const T* r2 = a.R2 + (m * p.layers + l) * HW;
      // publish this layer's input; the previous layer's tap reads were issued before (in-order LDS of one wave)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < NP; ++i) map[off(i)] = val[i];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const T rr = r1[pix(i)] * r2[pix(i)];
        const T* c = map + off(i);
        const T bs = ((c[-PW - 1] + c[-PW]) + (c[-PW + 1] + c[-1])) + ((c[0] + c[1]) + (c[PW - 1] + c[PW])) + c[PW + 1];
        const T kt = fma(w2_9, bs, b2);
        if (ACT == 0) {
          const T ss = rr > T(0) ? T(1.0 / (2.0 * nngp::kPi)) * rcp_any<T>(rr) : T(0);
          val[i] = nngp::relu_map<T, false>(kt, rr, ss).k;
        } else {
          val[i] = nngp::erf_map<T, false>(kt, rr, T(0)).k;
        }
      }
    }
    // Flatten (mean over pixels) + last Dense
    T s = T(0);
#pragma unroll
    for (int i = 0; i < NP; ++i) s += (EXACT || lane + 64 * i < HW) ? val[i] : T(0);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) {
      T v = (T)p.lw2 * s / (T)HW;
      if (a.symmetric && n == m) v = a.diag[n];
      a.out[n * a.ldo + m] = v;
      if (a.symmetric && a.mirror && n != m) a.out[m * a.ldo + n] = v;
    }
  }
}

// ---------------------------------------------------------------- 32 x 32 images (CIFAR: BASELINE.json's C3)
// The 3x3 box sum without LDS.  Lane l holds column l & 31 of rows 2i + (l >> 5), i = 0..15, so
//   horizontal:  the neighbours of a pixel are the same register of lanes l -+ 1 (DPP wave_shr / wave_shl; the image
//                border columns are masked by a per-lane 0/1 factor folded into the add),
//   vertical:    v_permlane32_swap turns the register of row pair i into E_i = row 2i and O_i = row 2i+1, each in ALL
//                lanes; a lower lane (row 2i) adds O_{i-1} + E_i + O_i, an upper lane (row 2i+1) E_i + O_i + E_{i+1}.
// No map, no halo, no LDS instruction in the layer loop: what is left to wait for are the factor-table loads.
template <typename T>
__device__ __forceinline__ T dpp_lane(T v, bool left);
template <>
__device__ __forceinline__ float dpp_lane<float>(float v, bool left) {
  const int x = __float_as_int(v);
  return __int_as_float(left ? __builtin_amdgcn_update_dpp(0, x, 0x138, 0xf, 0xf, false)     // wave_shr:1: lane l-1
                             : __builtin_amdgcn_update_dpp(0, x, 0x130, 0xf, 0xf, false));   // wave_shl:1: lane l+1
}
template <>
__device__ __forceinline__ double dpp_lane<double>(double v, bool left) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  if (left)
    return __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0x138, 0xf, 0xf, false),
                            __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xf, 0xf, false));
  return __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0x130, 0xf, 0xf, false),
                          __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xf, 0xf, false));
}
// e = the lower half-wave's values in all lanes, o = the upper half-wave's (scratch/dpp_probe: permlane32_swap(x, x))
__device__ __forceinline__ void rows32(float h, float& e, float& o) {
  const int x = __float_as_int(h);
  const auto p = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  e = __int_as_float((int)p[0]);
  o = __int_as_float((int)p[1]);
}
__device__ __forceinline__ void rows32(double h, double& e, double& o) {
  const int lo = __double2loint(h), hi = __double2hiint(h);
  const auto pl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto ph = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  e = __hiloint2double((int)ph[0], (int)pl[0]);
  o = __hiloint2double((int)ph[1], (int)pl[1]);
}

#ifndef SMN_CNN32_OCC
#define SMN_CNN32_OCC 2
#endif
template <typename T, int ACT>
__global__ void __launch_bounds__(256, SMN_CNN32_OCC) conv_pair32_kernel(PairArgs<T> a) {
  constexpr int NP = 16, HW = 1024, KB = KB0 < NP ? KB0 : NP;
  const ConvProg& p = a.prog;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 31;
  const bool upper = lane >= 32;
  const T ml = col > 0 ? T(1) : T(0), mr = col < 31 ? T(1) : T(0);
  const T w2_9 = (T)(p.w2 / 9.0), b2 = (T)p.b2;
  const T inv_c = (T)(1.0 / p.C);
  PairWalk<T> walk(a, wave);
  int64_t n, m;
  while (walk.next(n, m)) {
    const T* xa = a.x1 + n * HW * p.C;
    const T* xb = a.x2 + m * HW * p.C;
    T val[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) val[i] = T(0);
    for (int c = 0; c < p.C; ++c) {
#pragma unroll
      for (int i0 = 0; i0 < NP; i0 += KB) {   // 2 KB loads in flight per batch
        T va[KB], vb[KB];
#pragma unroll
        for (int j = 0; j < KB; ++j) {
          va[j] = xa[(lane + 64 * (i0 + j)) * p.C + c];
          vb[j] = xb[(lane + 64 * (i0 + j)) * p.C + c];
        }
#pragma unroll
        for (int j = 0; j < KB; ++j) val[i0 + j] = fma(va[j], vb[j], val[i0 + j]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) val[i] *= inv_c;
    for (int l = 0; l < p.layers; ++l) {
      const T* r1 = a.R1 + (n * p.layers + l) * HW + lane;
      const T* r2 = a.R2 + (m * p.layers + l) * HW + lane;
      T e[NP], o[NP];
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const T h = fma(ml, dpp_lane<T>(val[i], true), fma(mr, dpp_lane<T>(val[i], false), val[i]));
        rows32(h, e[i], o[i]);
      }
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        const T above = i > 0 ? o[i - 1] : T(0);        // row 2i - 1, for the lower lanes
        const T below = i + 1 < NP ? e[i + 1] : T(0);   // row 2i + 2, for the upper lanes
        const T bs = (e[i] + o[i]) + (upper ? below : above);
        const T rr = r1[64 * i] * r2[64 * i];   // (all 32 loads of the layer issued up front: 5 % slower, registers)
        const T kt = fma(w2_9, bs, b2);
        if (ACT == 0) {
          const T ss = rr > T(0) ? T(1.0 / (2.0 * nngp::kPi)) * rcp_any<T>(rr) : T(0);
          val[i] = nngp::relu_map<T, false>(kt, rr, ss).k;
        } else {
          val[i] = nngp::erf_map<T, false>(kt, rr, T(0)).k;
        }
      }
    }
    T s = T(0);
#pragma unroll
    for (int i = 0; i < NP; ++i) s += val[i];
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) s += __shfl_xor(s, o2);
    if (lane == 0) {
      T v = (T)p.lw2 * s / (T)HW;
      if (a.symmetric && n == m) v = a.diag[n];
      a.out[n * a.ldo + m] = v;
      if (a.symmetric && a.mirror && n != m) a.out[m * a.ldo + n] = v;
    }
  }
}

// conv_pair44_kernel — 32x32 images, one wave per image pair, a 4x4 PATCH of pixels per lane (round 3).
// conv_pair32_kernel keeps one pixel row pair per half-wave and moves every pixel's value to its neighbours (two DPP moves
// and a v_permlane32_swap per pixel and layer, 18 of its 65 vector instructions per pixel and layer, f64:
// profiles/r03_cnn_instruction_budget.txt).  With lane (py, px) = (lane >> 3, lane & 7) holding rows 4 py .. 4 py + 3 x columns
// 4 px .. 4 px + 3, the separable 3 + 3 box sum runs inside the lane's registers: per layer and lane 48 additions for 16 pixels and
// 16 halo values from the four neighbouring lanes, fetched by ds_bpermute_b32 (the LDS crossbar: no LDS memory, no vector-ALU
// instruction).  Border patches multiply their missing halo by a 0 / 1 factor folded into the addition (an fma).  The loads
// are whole 16-byte vectors as well: a patch row is 4 consecutive pixels, so its C channels and its factor-table entries
// are contiguous -- and fully coalesced: conv_q_kernel writes the factor tables and a copy of the images in exactly the order the
// lanes read them ([vector of a patch row][lane][16 bytes]).  One reciprocal step less for 1 / (r_i r_j) (v_rcp_f64 + one Newton
// step: 2^-50, against 2 steps).  (A degree-12 J(c), 1.1e-13 and 5 fused multiply-adds fewer, measured no gain: 579 against 572 ms.)
typedef float cnn_f32x4 __attribute__((ext_vector_type(4)));
typedef double cnn_f64x2 __attribute__((ext_vector_type(2)));
template <typename T>
__device__ __forceinline__ T lane_fetch(T v, int src_lane_bytes);
template <>
__device__ __forceinline__ float lane_fetch<float>(float v, int sb) {
  return __int_as_float(__builtin_amdgcn_ds_bpermute(sb, __float_as_int(v)));
}
template <>
__device__ __forceinline__ double lane_fetch<double>(double v, int sb) {
  return __hiloint2double(__builtin_amdgcn_ds_bpermute(sb, __double2hiint(v)), __builtin_amdgcn_ds_bpermute(sb, __double2loint(v)));
}

__device__ __forceinline__ float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double rcp_fast(double x) {
  const double r = __builtin_amdgcn_rcp(x);
  return fma(fma(-x, r, 1.0), r, r);
}

template <typename T, int ACT, int C>
__global__ void __launch_bounds__(256, SMN_CNN32_OCC) conv_pair44_kernel(PairArgs<T> a) {
  constexpr int HW = 1024, VEC = 16 / sizeof(T);
  constexpr int XV = 4 * C / VEC;          // 16-byte vectors of one patch row of an input image (4 pixels x C channels)
  constexpr int RV = 4 / VEC;              // ... of one patch row of a factor table (4 pixels); 1 (f32: 4 per vector) or 2
  static_assert((4 * C) % VEC == 0 && 4 % VEC == 0, "patch rows are whole 16-byte vectors");
  using vec_t = typename std::conditional<sizeof(T) == 8, cnn_f64x2, cnn_f32x4>::type;
  const ConvProg& p = a.prog;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int py = lane >> 3, px = lane & 7;
  const T ml = px > 0 ? T(1) : T(0), mr = px < 7 ? T(1) : T(0), mt = py > 0 ? T(1) : T(0), mb = py < 7 ? T(1) : T(0);
  const int sl = 4 * (px > 0 ? lane - 1 : lane), sr = 4 * (px < 7 ? lane + 1 : lane);     // ds_bpermute source lanes (x 4 bytes)
  const int st = 4 * (py > 0 ? lane - 8 : lane), sb = 4 * (py < 7 ? lane + 8 : lane);
  const T w2_9 = (T)(p.w2 / 9.0), b2 = (T)p.b2;
  const T inv_c = (T)(1.0 / C);
  PairWalk<T> walk(a, wave);
  int64_t n, m;
#ifdef SMN_CNN_TIMING   // wave 0 of workgroup 0: s_memtime cycles per phase, summed over its pairs
  long long tc[4] = {0, 0, 0, 0}, tp = __builtin_readcyclecounter();
  int npair = 0;
#define CT(i) do { const long long n_ = __builtin_readcyclecounter(); tc[i] += n_ - tp; tp = n_; } while (0)
#else
#define CT(i)
#endif
  while (walk.next(n, m)) {
    CT(3);
    const T* xa = a.x1 + n * HW * C + lane * VEC;    // a.x1 / a.x2: the patch-order copies made by conv_q_kernel
    const T* xb = a.x2 + m * HW * C + lane * VEC;
    T val[4][4];
    {
      vec_t va[4][XV], vb[4][XV];            // all loads of the pair's inputs in flight at once
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int v = 0; v < XV; ++v) {
          va[r][v] = *reinterpret_cast<const vec_t*>(xa + (r * XV + v) * 64 * VEC);
          vb[r][v] = *reinterpret_cast<const vec_t*>(xb + (r * XV + v) * 64 * VEC);
        }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          T s = T(0);
#pragma unroll
          for (int ch = 0; ch < C; ++ch) {
            const int e = c * C + ch;
            s = fma(va[r][e / VEC][e % VEC], vb[r][e / VEC][e % VEC], s);
          }
          val[r][c] = s * inv_c;
        }
    }
    asm volatile("" : "+v"(val[0][0]));
    CT(0);
    for (int l = 0; l < p.layers; ++l) {
      const T* r1 = a.R1 + (n * p.layers + l) * HW + lane * VEC;
      const T* r2 = a.R2 + (m * p.layers + l) * HW + lane * VEC;
      vec_t t1[4][RV], t2[4][RV];            // factor tables: requested first, used last
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int v = 0; v < RV; ++v) {
          t1[r][v] = *reinterpret_cast<const vec_t*>(r1 + (r * RV + v) * 64 * VEC);
          t2[r][v] = *reinterpret_cast<const vec_t*>(r2 + (r * RV + v) * 64 * VEC);
        }
      // horizontal 3-sums; the halo columns come from the lanes left and right
      T hl[4], hr[4], h[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        hl[r] = lane_fetch<T>(val[r][3], sl);
        hr[r] = lane_fetch<T>(val[r][0], sr);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const T s01 = val[r][0] + val[r][1], s23 = val[r][2] + val[r][3];
        h[r][1] = s01 + val[r][2];
        h[r][2] = val[r][1] + s23;
        h[r][0] = fma(ml, hl[r], s01);
        h[r][3] = fma(mr, hr[r], s23);
      }
      // the table products of the 16 pixels, formed once the loads have landed: 8 vectors per table die here
      T rr[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) rr[r][c] = t1[r][c / VEC][c % VEC] * t2[r][c / VEC][c % VEC];
      // vertical 3-sums; the halo rows come from the lanes above and below
      T ht[4], hb[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        ht[c] = lane_fetch<T>(h[3][c], st);
        hb[c] = lane_fetch<T>(h[0][c], sb);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const T s01 = h[0][c] + h[1][c], s23 = h[2][c] + h[3][c];
        const T b1 = s01 + h[2][c], b2_ = h[1][c] + s23;
        const T b0 = fma(mt, ht[c], s01), b3 = fma(mb, hb[c], s23);
        T bs[4] = {b0, b1, b2_, b3};
#ifdef SMN_CNN_TIMING
        if (c == 0) { asm volatile("" : "+v"(bs[0])); CT(1); }
#endif
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const T kt = fma(w2_9, bs[r], b2);
          if (ACT == 0) {
            const T ss = rr[r][c] > T(0) ? T(1.0 / (2.0 * nngp::kPi)) * rcp_fast(rr[r][c]) : T(0);
            val[r][c] = nngp::relu_map<T, false>(kt, rr[r][c], ss).k;
          } else {
            val[r][c] = nngp::erf_map<T, false>(kt, rr[r][c], T(0)).k;
          }
        }
      }
    }
    asm volatile("" : "+v"(val[0][0]));
    CT(2);
#ifdef SMN_CNN_TIMING
    ++npair;
#endif
    T s = T(0);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) s += val[r][c];
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) s += __shfl_xor(s, o2);
    if (lane == 0) {
      T v = (T)p.lw2 * s / (T)HW;
      if (a.symmetric && n == m) v = a.diag[n];
      a.out[n * a.ldo + m] = v;
      if (a.symmetric && a.mirror && n != m) a.out[m * a.ldo + n] = v;
    }
  }
#ifdef SMN_CNN_TIMING
  if (threadIdx.x == 0 && blockIdx.x == 0 && npair > 0)
    printf("conv_pair44 wg0 wave0: %d pairs; cycles per pair: K0 (x loads + products) %lld, box sums of %d layers (incl. waiting for the halo) %lld, activation maps %lld, reduce/store/next %lld\n",
           npair, tc[0] / npair, p.layers, tc[1] / npair, tc[2] / npair, tc[3] / npair);
#endif
#undef CT
}

// Launch one form of the pair kernel.  Tiled pair order once there are >= 64 tiles per XCD: the grid is then exactly the
// resident set (occupancy API; a multiple of 64 workgroups, so a tile is a whole number of 32-pair rows).
template <typename T, typename K>
int launch_pair_form(smn_ctx* ctx, K kern, PairArgs<T> a, int64_t blocks, size_t lds) {
  if (lds > 0)
    SMN_TRY(smn_allow_lds(ctx, reinterpret_cast<const void*>(kern), lds));
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, lds) == hipSuccess &&
      per_cu > 0) {
    const int64_t g = (int64_t)ctx->num_cu * per_cu / 64 * 64;
    if (g >= 64 && a.npairs >= 64 * 8 * (g / 8) * 4) {
      blocks = g;
      a.tile_bn = (int)(g / 64);
    }
  }
  ProfScope ps(ctx, PROF_BUILD, ctx->stream);
  hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, ctx->stream, a);
  return SMN_OK;
}

template <typename T, int ACT>
int launch_pairs(smn_ctx* ctx, const PairArgs<T>& a, int64_t blocks, size_t lds, int64_t hw) {
  // register-only stencil, no LDS: fp64 by default (+2 ... +11 % with growing N; the fp32 LDS-map kernel is VALU-bound at
  // 72 % busy and 5 % FASTER than its register form: profiles/r01f_cnn_fast32_ab.txt)
  if (a.prog.H == 32 && a.prog.W == 32 && sizeof(T) == 8) {
    // a 4x4 patch per lane where the patch rows are whole vectors (1 or 3 channels: MNIST- / CIFAR-shaped inputs)
    // (cnn_t made the same choice: the tables and a.x1 / a.x2 are in patch order then)
    if (a.prog.C == 3) return launch_pair_form<T>(ctx, conv_pair44_kernel<T, ACT, 3>, a, blocks, 0);
    if (a.prog.C == 1) return launch_pair_form<T>(ctx, conv_pair44_kernel<T, ACT, 1>, a, blocks, 0);
    return launch_pair_form<T>(ctx, conv_pair32_kernel<T, ACT>, a, blocks, 0);
  }
#define PAIR_CASE(NP)                                                                                         \
  if (hw <= 64 * NP) {                                                                                        \
    if (hw == 64 * NP && 64 % a.prog.W == 0)                                                                  \
      return launch_pair_form<T>(ctx, conv_pair_kernel<T, ACT, NP, true>, a, blocks, lds);                    \
    return launch_pair_form<T>(ctx, conv_pair_kernel<T, ACT, NP, false>, a, blocks, lds);                     \
  }
  PAIR_CASE(4)
  PAIR_CASE(16)
  PAIR_CASE(kMaxPix)
#undef PAIR_CASE
  return smn_fail(ctx, SMN_ENOTSUP, "smn_kernel_cnn: H*W > %d", 64 * kMaxPix);
}

template <typename T>
int cnn_t(smn_ctx* ctx, int act, int layers, double w, double b, double lw, const void* x1, int64_t n1,
          const void* x2, int64_t n2, int64_t H, int64_t W, int64_t C, int fill, void* out, int64_t ldk) {
  const bool sym = x2 == nullptr;
  if (sym) n2 = n1;
  ConvProg p{act, layers, (int)H, (int)W, (int)C, w * w, b * b, lw * lw};
  const int64_t HW = H * W;
  const size_t psz = (size_t)(H + 2) * (W + 2);
  const size_t lds_q = (2 * psz + 256) * sizeof(double);
  const size_t lds_p = 4 * (psz + 2 * (W + 2) + 3) * sizeof(T);   // one padded map (+ dummy slot) per wave
  if (lds_q > 160 * 1024 || lds_p > 160 * 1024)
    return smn_fail(ctx, SMN_ENOTSUP, "smn_kernel_cnn: image %lldx%lld too large for the on-chip pair map", (long long)H, (long long)W);
  // tables: R1 [n1][L][HW], diag1 [n1] (+ R2, diag2); patch order + patch-order copies of the inputs for conv_pair44_kernel
  const bool patch44 = H == 32 && W == 32 && (C == 1 || C == 3) && sizeof(T) == 8;
  const size_t qn1 = (size_t)n1 * (size_t)(layers > 0 ? layers : 1) * HW, qn2 = sym ? 0 : (size_t)n2 * (size_t)(layers > 0 ? layers : 1) * HW;
  const size_t xn1 = patch44 ? (size_t)n1 * HW * C : 0, xn2 = (patch44 && !sym) ? (size_t)n2 * HW * C : 0;
  void* tv = nullptr;
  SMN_TRY(smn_workspace(ctx, 1, sizeof(T) * (qn1 + qn2 + (size_t)n1 + (size_t)n2 + xn1 + xn2) + 64, &tv));
  T* R1 = static_cast<T*>(tv);
  T* R2 = sym ? R1 : R1 + qn1;
  T* d1 = R1 + qn1 + qn2;
  T* d2 = d1 + n1;
  T* xp1 = reinterpret_cast<T*>((reinterpret_cast<uintptr_t>(d2 + n2) + 15) & ~(uintptr_t)15);   // 16-byte aligned
  T* xp2 = sym ? xp1 : xp1 + xn1;
  {
    ProfScope ps(ctx, PROF_PREP, ctx->stream);
    SMN_TRY(smn_allow_lds(ctx, reinterpret_cast<const void*>(conv_q_kernel<T>), lds_q));
    hipLaunchKernelGGL(conv_q_kernel<T>, dim3((unsigned)n1), dim3(256), lds_q, ctx->stream,
                       static_cast<const T*>(x1), n1, p, R1, d1, patch44 ? 1 : 0, patch44 ? xp1 : nullptr);
    if (!sym)
      hipLaunchKernelGGL(conv_q_kernel<T>, dim3((unsigned)n2), dim3(256), lds_q, ctx->stream,
                         static_cast<const T*>(x2), n2, p, R2, d2, patch44 ? 1 : 0, patch44 ? xp2 : nullptr);
  }
  SMN_CHECK_LAUNCH(ctx);
  PairArgs<T> a;
  a.x1 = static_cast<const T*>(x1); a.x2 = sym ? a.x1 : static_cast<const T*>(x2);
  if (patch44) { a.x1 = xp1; a.x2 = xp2; }
  a.R1 = R1; a.R2 = R2; a.diag = d1; a.n1 = n1; a.n2 = n2;
  a.symmetric = sym ? 1 : 0; a.mirror = (sym && fill == SMN_FILL_FULL) ? 1 : 0;
  a.prog = p; a.out = static_cast<T*>(out); a.ldo = ldk;
  a.npairs = sym ? n1 * (n1 + 1) / 2 : n1 * n2;
  int64_t blocks = (a.npairs + 3) / 4;
  if (blocks > 256 * 8) blocks = 256 * 8;   // persistent-ish: waves stride over the pair list
  a.tile_bn = 0;
  SMN_TRY(act == SMN_ACT_RELU ? (launch_pairs<T, 0>(ctx, a, blocks, lds_p, HW)) : (launch_pairs<T, 1>(ctx, a, blocks, lds_p, HW)));
  SMN_CHECK_LAUNCH(ctx);
  return SMN_OK;
}

}  // namespace

extern "C" int smn_kernel_cnn(smn_ctx* ctx, int dtype, int act, int num_hiddens, double w_std, double b_std,
                              double last_w_std, const void* x1_d, int64_t n1, const void* x2_d, int64_t n2, int64_t H,
                              int64_t W, int64_t C, int fill, void* nngp_d, int64_t ldk) {
  if (!ctx || !x1_d || !nngp_d) return SMN_EINVAL;
  SMN_ENTER(ctx);
  if (dtype != SMN_F32 && dtype != SMN_F64) return smn_fail(ctx, SMN_EINVAL, "bad dtype %d", dtype);
  if (act != SMN_ACT_RELU && act != SMN_ACT_ERF) return smn_fail(ctx, SMN_EINVAL, "Unsupported act %d", act);
  if (n1 <= 0 || (x2_d && n2 <= 0) || H <= 0 || W <= 0 || C <= 0 || num_hiddens < 0)
    return smn_fail(ctx, SMN_EINVAL, "smn_kernel_cnn: bad sizes");
  if (H * W > 64 * kMaxPix) return smn_fail(ctx, SMN_ENOTSUP, "smn_kernel_cnn: H*W > %d", 64 * kMaxPix);
  if (dtype == SMN_F64)
    return cnn_t<double>(ctx, act, num_hiddens, w_std, b_std, last_w_std, x1_d, n1, x2_d, n2, H, W, C, fill, nngp_d, ldk);
  return cnn_t<float>(ctx, act, num_hiddens, w_std, b_std, last_w_std, x1_d, n1, x2_d, n2, H, W, C, fill, nngp_d, ldk);
}
