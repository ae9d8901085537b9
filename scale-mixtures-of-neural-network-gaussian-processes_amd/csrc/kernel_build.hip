// kernel_build.hip — NNGP / NTK kernel-matrix construction.
//
// Replaces kernel_fn(x1, x2, get) of experiments/nt_kernels.py:21-31,83-103 (neural_tangents
// stax.serial of Dense / Relu / Erf; called at spax/kernels.py:23-27, find.py:64-70):
//   1. pad_rows:       zero-padded operand copy + q_i = ||x_i||^2 / d          (HBM streaming)
//   2. diag_tables:    per-row, per-layer factors r, s and the closed-form diagonal (O(N L))
//   3. build_kernel:   K0 = X1 X2^T / d on the f32/f64 MFMA with the WHOLE layer recursion fused
//                      into the epilogue, so K0 never exists in HBM
//   4. recursion_kernel: the same per-element program as an HBM-streaming pass over a stored K0
//                      (hyper-parameter sweeps that reuse K0; the roofline measurement of a3)
#include <cmath>
#include <vector>

#include "gemm_nt.hpp"
#include "internal.hpp"
#include "layer_prog.hpp"

namespace {

constexpr int64_t kHalfTileBuildMax = 600;   // f32 build launches of at most this many 128x128 tiles use 64-row half tiles

// ------------------------------------------------------------------ prep kernels
template <typename T>
__global__ void pad_rows_kernel(const T* __restrict__ src, int64_t n, int64_t lds, int64_t d,
                                T* __restrict__ dst, int64_t rows_pad, int64_t kp, double inv_d,
                                double* __restrict__ q, int64_t rows_a = 0, const T* __restrict__ src2 = nullptr,
                                int64_t n2 = 0, int64_t lds2 = 0) {
  // one wave per padded row; rows_a > 0: the rows from rows_a on come from a second matrix (the appended block of an
  // augmented operand: one launch instead of two)
  const int lane = threadIdx.x & 63;
  int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows_pad) return;
  dst += row * kp;
  if (q) q += row;
  if (rows_a > 0 && row >= rows_a) {
    row -= rows_a; src = src2; n = n2; lds = lds2;
  }
  double s = 0.0;
  for (int64_t c = lane; c < kp; c += 64) {
    T v = (row < n && c < d) ? src[row * lds + c] : T(0);
    dst[c] = v;
    s += (double)v * (double)v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0 && q) *q = s * inv_d;
}

// Per-row tables.  tab[(set*2+0)*ldt + i] = r, tab[(set*2+1)*ldt + i] = s; dg[i] / dgt[i] = the
// closed-form NNGP / NTK diagonal (c = 1: ReLU Kdot = 1/2, erf Kdot = 4 / (pi sqrt(1 + 4q))).
template <typename T, typename Q = double>
__global__ void diag_tables_kernel(const Q* __restrict__ q0, int64_t n, LayerProg p,
                                   T* __restrict__ tab, int64_t ldt, T* __restrict__ dg, T* __restrict__ dgt,
                                   const LayerProg* __restrict__ progs = nullptr, int64_t tab_bs = 0) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (progs) {   // batched build: problem blockIdx.y has its own (w, b, last_w) and its own table block
    p = progs[blockIdx.y];
    tab += (int64_t)blockIdx.y * tab_bs; dg += (int64_t)blockIdx.y * tab_bs; dgt += (int64_t)blockIdx.y * tab_bs;
  }
  double q = (double)q0[i], th = 0.0;   // (Q = T: smn_recursion takes q in the compute type)
  if (p.net == NET_RESNET) {
    q = p.w2 * q + p.b2;
    th = q;
  }
  const double w = sqrt(p.w2), b = sqrt(p.b2);
  double s_prev = 1.0 / w;   // FAST: u^0 = w r^0 (s_prev * w == 1)
  for (int s = 0; s < p.nsets; ++s) {
    const double qt = (p.net == NET_MLP) ? p.w2 * q + p.b2 : q;   // pre-activation variance
    const double tht = (p.net == NET_MLP) ? qt + p.w2 * th : th;
    double r, sv, qa, kdot;
    if (p.act == ACT_RELU) {
      r = qt > 0.0 ? 1.0 / sqrt(qt) : 0.0;
      sv = sqrt(qt / (2.0 * nngp::kPi));
      qa = 0.5 * qt;
      kdot = 0.5;
      if (p.fast) sv = sqrt(0.5 * qt);   // the fast map returns J / pi: s_i s_j pi (J / pi) = (sqrt(pi) s_i)(sqrt(pi) s_j) (J / pi)
    } else {
      r = 1.0 / sqrt(1.0 + 2.0 * qt);
      sv = 0.0;
      qa = (2.0 / nngp::kPi) * asin(2.0 * qt / (1.0 + 2.0 * qt));
      kdot = 4.0 / (nngp::kPi * sqrt(1.0 + 4.0 * qt));
      if (p.fast) {   // correlation space: c = 2 K~ r_i r_j = K~ r'_i r'_j, x = asin c, K_act = (2/pi) x = s^2 x
        r *= sqrt(2.0);
        sv = sqrt(2.0 / nngp::kPi);
      }
    }
    if (p.fast) {   // correlation-space tables (layer_prog.hpp): u = w s_prev r, v = b r
      tab[(int64_t)(2 * s) * ldt + i] = (T)(s == 0 ? w * r : w * s_prev * r);
      tab[(int64_t)(2 * s + 1) * ldt + i] = (T)(b * r);
      s_prev = sv;
    } else {
      tab[(int64_t)(2 * s) * ldt + i] = (T)r;
      tab[(int64_t)(2 * s + 1) * ldt + i] = (T)sv;
    }
    if (p.net == NET_MLP || s == p.nsets - 1) {
      q = qa;
      th = tht * kdot;
    } else {
      const double ka = p.w2 * qa + p.b2;
      th += ka + p.w2 * (tht * kdot);
      q += ka;
    }
  }
  if (p.net != NET_NONE) {
    q *= p.lw2;
    th = q + p.lw2 * th;
  }
  dg[i] = (T)q;
  // FAST: sigma = last_w * s^{L-1} (no hidden layer: K = last_w^2 K0, sigma = last_w)
  dgt[i] = p.fast ? (T)(sqrt(p.lw2) * (p.nsets > 0 ? s_prev : 1.0)) : (T)th;
}

// sum of the exact diagonal dg[0, n) -- what diag_trace_kernel (cholesky.hip) reads off the built matrix, from the table the build
// takes it from: the same values summed by the same tree (the relative ridge of the predictive path needs the trace BEFORE
// the matrix exists when the prep launch is to carry the shift, and the matrix's diagonal is a 4-byte read per cache line)
template <typename T>
__global__ void table_trace_kernel(const T* __restrict__ dg, int64_t n, double* __restrict__ out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += (double)dg[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) *out = red[0];
}

// ------------------------------------------------------------------ fused Gram + recursion
template <typename T>
struct BuildArgs {
  const T* x1; const T* x2; int64_t ld1, ld2; int kp;
  int tiles_n; int tiles_m; int symmetric; int mirror;
  int lower_skip;             // rectangular grid: drop tiles lying wholly above the global diagonal
  const T* tab1; const T* tab2; int64_t ldt1, ldt2; const T* dg; const T* dgt;
  T inv_d; LayerProg prog;
  int64_t row_off, col_off; int exact_diag;
  int store_mode; int64_t out_rows, out_cols; int64_t nv0, aug0, nv1;
  T* out_k; T* out_t; int64_t ldo;
  int use_map; TileMap map;   // XCD-aware patch order (gemm_nt.hpp)
  // paired lower-block shard (smn_kernel_mlp_shard): two trapezoids of row tiles, each packed into its own
  // output with its own leading dimension; operands and tables are the symmetric ones
  int shard; int sh_b0[2]; int sh_cnt0; T* sh_k[2]; T* sh_t[2]; int64_t sh_ld[2]; int64_t sh_cols[2];
  // cyclic column-first shard (smn_kernel_mlp_shard_cols, shard == 2): the rank's tile rows are t_j = j P + (j even ? rank :
  // P-1-rank), every lower tile of them, stored into the rank's chunk piece by piece (a piece = the tile columns
  // [cy_c[g], cy_c[g+1]) of every tile row from cy_c[g] down, cy_c[g] a multiple of P): slot (j - cy_c[g]/P) of piece g is a
  // 128 x (width of the piece) strip.  sh_k[0] / sh_t[0] are the chunk bases.
  int cy_P, cy_rank, cy_T, cy_np; int cy_c[kMaxColPieces + 1]; int64_t cy_off[kMaxColPieces];
  // batched build (smn_spr_loss_batch / smn_spr_predict_batch): problem blockIdx.y = the same operands under its own layer
  // program progs[y] (same net, act and depth; its own w, b, last_w), its own table block and its own output matrix
  const LayerProg* progs; int64_t tab_bs, out_bs; int nbatch;
  // explicit tile order (split build): tile l = tlist[l] = tr | tc << 16, dealt to the XCDs in equal contiguous shares like the map
  const int* tlist; int tlist_n;
};

// BM = 64 (f32 launches of few tiles: a rank's share of a sharded build, small kernels): two workgroups per 128x128 tile, 64
// rows each -- at about one tile per CU a launch ends in a tail as long as a tile; halves end in half of it.
template <typename T, int NET, int ACT, bool NTK, int BM = kTile>
__global__ void __launch_bounds__(256, (sizeof(T) == 8 && NTK) ? 1 : (BM == 64 && !NTK ? 3 : 2)) build_kernel(BuildArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using Tile = TileNT<T, BM, kTile, SMN_STAGES>;
  using M = typename Tile::M;
  static_assert(BM == kTile || BM == 64, "row tile: 128 or 64");
  int tr, tc, half = 0;
  T* out_k = a.out_k; T* out_t = a.out_t;
  int64_t ldo = a.ldo, out_cols = a.out_cols;
  if (a.shard == 2) {
    int idx = blockIdx.x;
    if (BM == 64) {
      half = idx & 1;
      idx >>= 1;
    }
    int j = 0, t = a.cy_rank;                  // tile row t_j holds t_j + 1 lower tiles (none when it lies past the kernel)
    while (true) {
      const int cnt = t < a.cy_T ? t + 1 : 0;
      if (idx < cnt) break;
      idx -= cnt;
      ++j;
      t = j * a.cy_P + ((j & 1) ? a.cy_P - 1 - a.cy_rank : a.cy_rank);
      if (j * a.cy_P >= a.cy_T) return;        // (grid padding; uniform per workgroup)
    }
    tr = t;
    tc = idx;
    int g = 0;
    while (g + 1 < a.cy_np && tc >= a.cy_c[g + 1]) ++g;
    ldo = (int64_t)(a.cy_c[g + 1] - a.cy_c[g]) * kTile;
    out_cols = a.out_cols;
    // strip of slot j - c_g / P: local row (gr - t*128), local column (gc - c_g*128); shifts folded into the base pointers
    const int64_t shift = a.cy_off[g] + ((int64_t)(j - a.cy_c[g] / a.cy_P) - t) * kTile * ldo - (int64_t)a.cy_c[g] * kTile;
    out_k = a.sh_k[0] ? a.sh_k[0] + shift : nullptr;
    out_t = a.sh_t[0] ? a.sh_t[0] + shift : nullptr;
  } else if (a.shard) {
    int idx = blockIdx.x;
    if (BM == 64) {
      half = idx & 1;
      idx >>= 1;
    }
    const int w = idx >= a.sh_cnt0 ? 1 : 0;
    if (w) idx -= a.sh_cnt0;
    const int b0 = w ? a.sh_b0[1] : a.sh_b0[0];
    int t = 0;                                 // row tile b0+t of the block holds b0+t+1 lower tiles
    while (idx >= b0 + t + 1) { idx -= b0 + t + 1; ++t; }
    tr = b0 + t;
    tc = idx;
    ldo = w ? a.sh_ld[1] : a.sh_ld[0];
    out_cols = w ? a.sh_cols[1] : a.sh_cols[0];
    // packed block: row (gr - b0*128) of leading dimension ldo; fold the row shift into the base pointers
    out_k = w ? a.sh_k[1] : a.sh_k[0];
    out_t = w ? a.sh_t[1] : a.sh_t[0];
    if (out_k) out_k -= (int64_t)b0 * kTile * ldo;
    if (out_t) out_t -= (int64_t)b0 * kTile * ldo;
  } else {
    unsigned bid = blockIdx.x;
    if (BM == 64) {   // (grids padded to a multiple of 8 tiles) both halves of a tile on the same XCD: workgroups b and b + 8 share one
      half = (bid >> 3) & 1;
      bid = ((bid >> 4) << 3) | (bid & 7);
    }
    if (a.tlist) {
      const int l = (int)(bid & 7) * (int)(gridDim.x >> 3) + (int)(bid >> 3);
      if (l >= a.tlist_n) return;
      const int v = a.tlist[l];
      tr = v & 0xffff;
      tc = v >> 16;
    } else if (a.use_map) {
      if (!a.map.decode(bid, tr, tc)) return;
    } else if (a.symmetric) {
      if (bid >= (unsigned)(a.tiles_n * (a.tiles_n + 1) / 2)) return;
      tri_decode(bid, tr, tc);
    } else {
      tr = bid / a.tiles_n;
      tc = bid % a.tiles_n;
      if (tr >= a.tiles_m) return;
      if (a.lower_skip && (int64_t)tc * kTile + a.col_off > (int64_t)tr * kTile + a.row_off + kTile - 1) return;
    }
  }
  const int64_t row0 = (int64_t)tr * kTile + half * BM, col0 = (int64_t)tc * kTile;
  Tile t;
  t.zero();
  // a tile of the augmented matrix with no valid row or no valid column is pure identity padding (the y-row tile
  // row of SPR.loss): no Gram to compute, the store below writes the padding
  auto no_valid = [&](int64_t o) {
    return o >= a.nv0 && !(a.nv1 > 0 && o < a.aug0 + a.nv1 && o + kTile > a.aug0);
  };
  const bool dead = a.store_mode == STORE_PAD_IDENTITY && (no_valid(row0) || no_valid(col0));
  t.mainloop(a.x1 + row0 * a.ld1, a.ld1, a.x2 + col0 * a.ld2, a.ld2, dead ? 0 : a.kp, smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  ElemProg<T, NET, ACT, NTK> prog(a.prog);
  const int nsets = a.prog.nsets;
  // (locals, not writes into `a`: a kernel argument that is written gets a private copy in scratch)
  const T* tab1 = a.tab1; const T* tab2 = a.tab2; const T* dgp = a.dg; const T* dgtp = a.dgt;
  if (a.progs) {
    prog = ElemProg<T, NET, ACT, NTK>(a.progs[blockIdx.y]);
    const int64_t tb = (int64_t)blockIdx.y * a.tab_bs, ob = (int64_t)blockIdx.y * a.out_bs;
    tab1 += tb; tab2 += tb; dgp += tb; dgtp += tb;
    if (out_k) out_k += ob;
    if (out_t) out_t += ob;
  }

  // stage the per-row / per-column layer tables of this tile in LDS (mainloop ended on a barrier)
  constexpr bool FAST = ElemProg<T, NET, ACT, NTK>::FAST;
  const int trows = nsets * 2 + (FAST ? 1 : 0);   // FAST: one more row, sigma (kept in the NTK-diagonal slot)
  T* srow = reinterpret_cast<T*>(smem);
  T* scol = srow + trows * kTile;
  for (int idx = tid; idx < trows * kTile; idx += 256) {
    const int s2 = idx / kTile, r = idx % kTile;
    const int g2 = s2 < nsets * 2 ? s2 : nsets * 2 + 1;
    if (r < BM) srow[idx] = tab1[(int64_t)g2 * a.ldt1 + row0 + r];
    scol[idx] = tab2[(int64_t)g2 * a.ldt2 + col0 + r];
  }
  __syncthreads();

  typename Tile::acc_t th[Tile::MT][Tile::NT];
#pragma unroll
  for (int m = 0; m < Tile::MT; ++m)
#pragma unroll
    for (int n = 0; n < Tile::NT; ++n)
#pragma unroll
      for (int i = 0; i < M::ACC; ++i) {
        T k = t.acc[m][n][i] * a.inv_d;
        T h = T(0);
        prog.pre(k, h);
        t.acc[m][n][i] = k;
        th[m][n][i] = h;
      }

  for (int s = 0; s < nsets; ++s) {
    const T* sr = srow + s * 2 * kTile;
    const T* sc = scol + s * 2 * kTile;
    T cr[Tile::NT], cs[Tile::NT];
#pragma unroll
    for (int n = 0; n < Tile::NT; ++n) {
      const int lc = wc * Tile::WN + n * M::TN + M::acc_col(lane);
      cr[n] = sc[lc];
      cs[n] = sc[kTile + lc];
    }
#pragma unroll
    for (int m = 0; m < Tile::MT; ++m)
#pragma unroll
      for (int i = 0; i < M::ACC; ++i) {
        const int lr = wr * Tile::WM + m * M::TM + M::acc_row(lane, i);
        const T ri = sr[lr], si = sr[kTile + lr];
#pragma unroll
        for (int n = 0; n < Tile::NT; ++n) {
          T k = t.acc[m][n][i], h = th[m][n][i];
          prog.step(s, k, h, ri * cr[n], si * cs[n]);
          t.acc[m][n][i] = k;
          th[m][n][i] = h;
        }
      }
  }

  const bool do_mirror = a.symmetric && a.mirror && tr != tc;
#pragma unroll
  for (int m = 0; m < Tile::MT; ++m)
#pragma unroll
    for (int n = 0; n < Tile::NT; ++n)
#pragma unroll
      for (int i = 0; i < M::ACC; ++i) {
        const int64_t gr = row0 + wr * Tile::WM + m * M::TM + M::acc_row(lane, i);
        const int64_t gc = col0 + wc * Tile::WN + n * M::TN + M::acc_col(lane);
        T k = t.acc[m][n][i], h = th[m][n][i];
        T sig = T(0);
        if (FAST)
          sig = srow[nsets * 2 * kTile + (int)(gr - row0)] * scol[nsets * 2 * kTile + (int)(gc - col0)];
        prog.post(k, h, sig);
        if (a.exact_diag && gr + a.row_off == gc + a.col_off) {
          k = dgp[gr];
          if (NTK) h = dgtp[gr];
        }
        bool wr_ok;
        if (a.store_mode == STORE_PAD_IDENTITY) {
          const bool vr = gr < a.nv0 || (gr >= a.aug0 && gr < a.aug0 + a.nv1);
          const bool vc = gc < a.nv0 || (gc >= a.aug0 && gc < a.aug0 + a.nv1);
          if (!(vr && vc)) {
            k = (gr == gc) ? T(1) : T(0);
            h = k;
          }
          wr_ok = true;
        } else {
          wr_ok = gr < a.out_rows && gc < out_cols;
        }
        if (wr_ok) {
          if (out_k) out_k[gr * ldo + gc] = k;
          if (NTK && out_t) out_t[gr * ldo + gc] = h;
        }
        if (do_mirror && gc < a.out_rows && gr < out_cols) {
          if (out_k) out_k[gc * ldo + gr] = k;
          if (NTK && out_t) out_t[gc * ldo + gr] = h;
        }
      }
}

// ------------------------------------------------------------------ standalone recursion (HBM streaming)
template <typename T>
struct RecArgs {
  const T* k0; int64_t ldk0; int64_t n1, n2;
  const T* tab1; const T* tab2; int64_t ldt1, ldt2; const T* dg; const T* dgt;
  LayerProg prog; int exact_diag;
  T* out_k; T* out_t; int64_t ldo; int rows_per_block;
  int sym_tiles;   // recursion_sym_kernel: 1 one workgroup per LOWER 64x64 tile (+ mirror), 2 one per tile of the n1 x n2 rectangle
};

// Stand-alone layer recursion over a stored K0, one workgroup per 64x64 tile, column and row tables in LDS.
// Symmetric form (x2 == x1, full output): LOWER tiles only.  The element chains run once per unordered pair; the tile is
// written straight (16-byte stores) and, through an LDS transpose, mirrored into the upper triangle.  Halves the VALU work
// that bounds the 4-layer map and the K0 bytes read.  Cross form (sym_tiles == 2): every tile of the rectangle, no mirror.
template <typename T, int NET, int ACT, bool NTK>
__global__ void __launch_bounds__(256) recursion_sym_kernel(RecArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int TS = 64, VEC = 16 / sizeof(T);
  constexpr int LPR = TS / VEC;        // lanes per tile row
  constexpr int RPP = 256 / LPR;       // tile rows covered per pass of the workgroup
  constexpr int NP = TS / RPP;         // passes
  constexpr int LDT = TS + 1;          // transpose buffer stride: scalar LDS accesses, conflict-free both ways
  constexpr bool FAST = ElemProg<T, NET, ACT, NTK>::FAST;
  using vec_t = typename Mfma<T>::vec_t;
  const int tid = threadIdx.x;
  const int nsets = a.prog.nsets;
  const int trows = nsets * 2 + 1;
  T* scol = reinterpret_cast<T*>(smem);          // [trows][TS] factors of the tile's columns
  T* srow = scol + trows * TS;                   // [trows][TS] factors of the tile's rows
  T* tbk = srow + trows * TS;                    // [TS][LDT] tile of K for the mirror
  T* tbt = tbk + TS * LDT;                       // [TS][LDT] tile of Theta (NTK only)
  // Cross kernels (x1 != x2) take the same tile shape over the whole n1 x n2 rectangle: every 16-byte load of a 64x64 tile in
  // flight before the tables are staged, 8 workgroups per CU (the r01 strip kernel had two loads per lane in flight and read
  // at 51 % of the HBM rate where the map itself is cheap: profiles/r03_recursion_table.txt).
  const bool rect = a.sym_tiles == 2;
  int tr, tc;
  if (rect) {
    const int tn = (int)((a.n2 + TS - 1) / TS);
    tr = (int)(blockIdx.x / tn);
    tc = (int)(blockIdx.x % tn);
  } else {
    tri_decode(blockIdx.x, tr, tc);
  }
  const int64_t row0 = (int64_t)tr * TS, col0 = (int64_t)tc * TS, n = a.n1, nc = rect ? a.n2 : a.n1;
  const T* tabc = rect ? a.tab2 : a.tab1;
  const int64_t ldtc = rect ? a.ldt2 : a.ldt1;
  const int lr0 = tid / LPR, lc = (tid % LPR) * VEC;
  const int64_t gc = col0 + lc;
  const bool cfull = gc + VEC <= nc;
  vec_t kv[NP], hv[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {                 // every load of the tile in flight before the tables are staged
    const int64_t gr = row0 + p * RPP + lr0;
    if (gr < n && cfull) {
      kv[p] = *reinterpret_cast<const vec_t*>(a.k0 + gr * a.ldk0 + gc);
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e) kv[p][e] = (gr < n && gc + e < nc) ? a.k0[gr * a.ldk0 + gc + e] : T(0);
    }
  }
  for (int idx = tid; idx < trows * TS; idx += 256) {
    const int s2 = idx / TS, r = idx % TS;
    const int g2 = s2 < nsets * 2 ? s2 : nsets * 2 + 1;
    scol[idx] = col0 + r < nc ? tabc[(int64_t)g2 * ldtc + col0 + r] : T(0);
    srow[idx] = row0 + r < n ? a.tab1[(int64_t)g2 * a.ldt1 + row0 + r] : T(0);
  }
  __syncthreads();
  const ElemProg<T, NET, ACT, NTK> prog(a.prog);
  const bool mirror = !rect && tr != tc;
  const bool diag_exact = rect ? a.exact_diag != 0 : true;
#pragma unroll
  for (int pp = 0; pp < NP; pp += 2) {           // two passes at a time: 2 * VEC independent chains per lane
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        T k = kv[pp + r][e], h = T(0);
        prog.pre(k, h);
        kv[pp + r][e] = k;
        hv[pp + r][e] = h;
      }
    for (int s = 0; s < nsets; ++s) {
      const vec_t cr = *reinterpret_cast<const vec_t*>(scol + (2 * s) * TS + lc);
      const vec_t cs = *reinterpret_cast<const vec_t*>(scol + (2 * s + 1) * TS + lc);
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int lr = (pp + r) * RPP + lr0;
        const T ri = srow[(2 * s) * TS + lr], si = srow[(2 * s + 1) * TS + lr];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          T k = kv[pp + r][e], h = hv[pp + r][e];
          prog.step(s, k, h, ri * cr[e], si * cs[e]);
          kv[pp + r][e] = k;
          hv[pp + r][e] = h;
        }
      }
    }
    vec_t sgc = vec_t{};
    if (FAST) sgc = *reinterpret_cast<const vec_t*>(scol + (2 * nsets) * TS + lc);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int lr = (pp + r) * RPP + lr0;
      const int64_t gr = row0 + lr;
      const T sgr = FAST ? srow[(2 * nsets) * TS + lr] : T(0);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        T k = kv[pp + r][e], h = hv[pp + r][e];
        prog.post(k, h, sgr * sgc[e]);
        if (diag_exact && gr == gc + e) {
          k = a.dg[gr < n ? gr : 0];
          if (NTK) h = a.dgt[gr < n ? gr : 0];
        }
        kv[pp + r][e] = k;
        hv[pp + r][e] = h;
        if (mirror) {
          tbk[lr * LDT + lc + e] = k;
          if (NTK) tbt[lr * LDT + lc + e] = h;
        }
      }
      if (gr < n) {
        if (cfull) {
          if (a.out_k) *reinterpret_cast<vec_t*>(a.out_k + gr * a.ldo + gc) = kv[pp + r];
          if (NTK && a.out_t) *reinterpret_cast<vec_t*>(a.out_t + gr * a.ldo + gc) = hv[pp + r];
        } else {
#pragma unroll
          for (int e = 0; e < VEC; ++e)
            if (gc + e < nc) {
              if (a.out_k) a.out_k[gr * a.ldo + gc + e] = kv[pp + r][e];
              if (NTK && a.out_t) a.out_t[gr * a.ldo + gc + e] = hv[pp + r][e];
            }
        }
      }
    }
  }
  if (!mirror) return;                           // uniform per workgroup
  __syncthreads();
  // mirrored store: output row col0 + cc takes the tile's column cc; lanes again own VEC consecutive output columns
  const int64_t oc = row0 + lc;
  const bool ofull = oc + VEC <= n;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int cc = p * RPP + lr0;
    const int64_t orow = col0 + cc;
    if (orow >= n) continue;
    vec_t vk, vt = vec_t{};
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      vk[e] = tbk[(lc + e) * LDT + cc];
      if (NTK) vt[e] = tbt[(lc + e) * LDT + cc];
    }
    if (ofull) {
      if (a.out_k) *reinterpret_cast<vec_t*>(a.out_k + orow * a.ldo + oc) = vk;
      if (NTK && a.out_t) *reinterpret_cast<vec_t*>(a.out_t + orow * a.ldo + oc) = vt;
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e)
        if (oc + e < n) {
          if (a.out_k) a.out_k[orow * a.ldo + oc + e] = vk[e];
          if (NTK && a.out_t) a.out_t[orow * a.ldo + oc + e] = vt[e];
        }
    }
  }
}

// ------------------------------------------------------------------ host-side dispatch
int make_prog(smn_ctx* ctx, const BuildSpec& s, LayerProg* p) {
  if (s.net != SMN_NET_MLP && s.net != SMN_NET_DENSE_RESNET && s.net != NET_NONE)
    return smn_fail(ctx, SMN_EINVAL, "unknown net %d", s.net);
  if (s.act != SMN_ACT_RELU && s.act != SMN_ACT_ERF)
    return smn_fail(ctx, SMN_EINVAL, "Unsupported act %d", s.act);   // nt_kernels.py:18 KeyError
  if (s.num_hiddens < 0) return smn_fail(ctx, SMN_EINVAL, "num_hiddens < 0");
  p->net = s.net;
  p->act = s.act;
  p->nsets = s.net == NET_NONE ? 0 : (s.net == SMN_NET_MLP ? s.num_hiddens : s.num_hiddens + 1);
  if (p->nsets > kMaxSets) return smn_fail(ctx, SMN_ENOTSUP, "num_hiddens too large (max %d activation layers)", kMaxSets);
  p->w2 = s.w_std * s.w_std;
  p->b2 = s.b_std * s.b_std;
  p->lw2 = s.last_w_std * s.last_w_std;
  p->fast = 0;
  return SMN_OK;
}

// The launched kernel template decides FAST at compile time (layer_prog.hpp); the tables must agree.
template <typename T>
void set_fast(LayerProg* p, bool ntk) {
  p->fast = (sizeof(T) == 4 && p->net == NET_MLP && !ntk) ? 1 : 0;
}

template <typename T, int NET, int ACT, bool NTK, int BM = kTile>
int launch_build_t(smn_ctx* ctx, const BuildArgs<T>& a, int64_t ntiles, size_t lds, hipStream_t st) {
  auto kern = build_kernel<T, NET, ACT, NTK, BM>;
  if (BM == 64) {
    ntiles *= 2;
    lds = std::max<size_t>(TileNT<T, 64, kTile, SMN_STAGES>::LDS_BYTES, (size_t)(a.prog.nsets * 2 + 1) * 2 * kTile * sizeof(T));
  }
  SMN_TRY(smn_allow_lds(ctx, reinterpret_cast<const void*>(kern), lds));
  {
    ProfScope ps(ctx, PROF_BUILD, st);
    hipLaunchKernelGGL(kern, dim3((unsigned)ntiles, (unsigned)(a.progs ? a.nbatch : 1)), dim3(256), lds, st, a);
  }
  SMN_CHECK_LAUNCH(ctx);
  return SMN_OK;
}

template <typename T, int NET, int ACT>
int launch_build_n(smn_ctx* ctx, const BuildArgs<T>& a, int64_t ntiles, size_t lds, bool ntk, hipStream_t st) {
  if constexpr (sizeof(T) == 4) {
    // f32 launches of few tiles (a rank's share of a sharded build, small kernels): 64-row half tiles.  Measured per rank
    // (profiles/r03_shard_pieces_probe.json): 11-17 % faster for launches of 260-520 tiles, a few % either way around 1000 tiles;
    // the un-sharded 8392-tile build of one GPU is 5 % SLOWER with them (6.97 against 6.6 ms): launches above kHalfTileBuildMax
    // keep the 128-row tile.  (Un-sharded launches too, from 64 tiles on: C2's 528-tile build 0.144 -> 0.118 ms.)
    if (ntiles <= kHalfTileBuildMax && (a.shard || ntiles >= 64) && !a.tlist) {
      if (!a.shard) ntiles = (ntiles + 7) / 8 * 8;   // the un-sharded decode pairs the halves inside groups of 16 workgroups
      return ntk ? launch_build_t<T, NET, ACT, true, 64>(ctx, a, ntiles, lds, st)
                 : launch_build_t<T, NET, ACT, false, 64>(ctx, a, ntiles, lds, st);
    }
  }
  return ntk ? launch_build_t<T, NET, ACT, true>(ctx, a, ntiles, lds, st)
             : launch_build_t<T, NET, ACT, false>(ctx, a, ntiles, lds, st);
}

template <typename T>
int launch_build(smn_ctx* ctx, const BuildArgs<T>& a, int64_t ntiles, size_t lds, bool ntk, hipStream_t st = nullptr) {
  const int net = a.prog.net, act = a.prog.act;
  if (!st) st = ctx->stream;
  if (net == NET_NONE) return launch_build_t<T, NET_NONE, ACT_RELU, false>(ctx, a, ntiles, lds, st);
  if (net == NET_MLP && act == ACT_RELU) return launch_build_n<T, NET_MLP, ACT_RELU>(ctx, a, ntiles, lds, ntk, st);
  if (net == NET_MLP && act == ACT_ERF) return launch_build_n<T, NET_MLP, ACT_ERF>(ctx, a, ntiles, lds, ntk, st);
  if (net == NET_RESNET && act == ACT_RELU) return launch_build_n<T, NET_RESNET, ACT_RELU>(ctx, a, ntiles, lds, ntk, st);
  return launch_build_n<T, NET_RESNET, ACT_ERF>(ctx, a, ntiles, lds, ntk, st);
}

// Tile orders of a split build of T tile rows with a corner of TB: the XCD patch order of the whole triangle, filtered into
// "everything but the corner" (na tiles, at tile_list) and "the corner" (nb tiles, behind them).  Cached per (T, TB).
int split_tile_lists(smn_ctx* ctx, int T, int TB) {
  if (ctx->tile_list && ctx->tile_list_tm == T && ctx->tile_list_tb == TB) return SMN_OK;
  const TileMap map = TileMap::make(T, T, 1);
  std::vector<int> la, lb;
  la.reserve((size_t)map.ntiles);
  for (int l = 0; l < map.ntiles; ++l) {
    int tr = 0, tc = 0;
    if (!map.decode_linear(l, tr, tc)) return smn_fail(ctx, SMN_EINVAL, "split_tile_lists: tile order");
    ((tr >= T - TB && tc >= T - TB) ? lb : la).push_back(tr | (tc << 16));
  }
  if ((int64_t)lb.size() != (int64_t)TB * (TB + 1) / 2) return smn_fail(ctx, SMN_EINVAL, "split_tile_lists: corner count");
  // (a kernel of an earlier call may still read the old lists only if that call failed half-way: drain before rewriting)
  SMN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->stream_bulk) SMN_HIP(ctx, hipStreamSynchronize(ctx->stream_bulk));
  if (ctx->tile_list_cap < map.ntiles) {
    if (ctx->tile_list) (void)hipFree(ctx->tile_list);
    ctx->tile_list = nullptr; ctx->tile_list_cap = 0; ctx->tile_list_tm = 0;
    SMN_HIP(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->tile_list), sizeof(int) * (size_t)map.ntiles));
    ctx->tile_list_cap = map.ntiles;
  }
  ctx->tile_list_tm = 0;
  SMN_HIP(ctx, hipMemcpy(ctx->tile_list, la.data(), sizeof(int) * la.size(), hipMemcpyHostToDevice));
  SMN_HIP(ctx, hipMemcpy(ctx->tile_list + la.size(), lb.data(), sizeof(int) * lb.size(), hipMemcpyHostToDevice));
  ctx->tile_list_tm = T; ctx->tile_list_tb = TB; ctx->tile_list_na = (int)la.size(); ctx->tile_list_nb = (int)lb.size();
  return SMN_OK;
}

template <typename T>
int run_build_t(smn_ctx* ctx, const BuildCall& c) {
  LayerProg prog;
  SMN_TRY(make_prog(ctx, c.spec, &prog));
  if (c.rows1 % kTile || c.rows2 % kTile || c.kp % Mfma<T>::BK)
    return smn_fail(ctx, SMN_EINVAL, "run_build: operands not padded");
  const bool ntk = (c.get_mask & SMN_GET_NTK) != 0;
  if (ntk && prog.net == NET_NONE) return smn_fail(ctx, SMN_EINVAL, "NTK of a bare Gram");
  set_fast<T>(&prog, ntk);
  // tables: [2*nsets + 2] rows of length rows1 (+ rows2 when not symmetric); one such block per problem of a batched build
  const int trows = 2 * prog.nsets + 2;
  const int64_t tlen = c.symmetric ? c.rows1 : c.rows1 + c.rows2;
  const int nb = c.nbatch > 0 ? c.nbatch : 1;
  const int64_t tab_bs = (int64_t)trows * tlen;
  void* tabv = nullptr;
  SMN_TRY(smn_workspace(ctx, 1, sizeof(T) * (size_t)tab_bs * (size_t)nb, &tabv));
  T* tab1 = static_cast<T*>(tabv);
  T* dg1 = tab1 + (int64_t)(2 * prog.nsets) * tlen;
  T* dgt1 = dg1 + tlen;
  LayerProg* progs_d = nullptr;
  if (c.nbatch > 0) {
    if (!c.symmetric || c.shard) return smn_fail(ctx, SMN_EINVAL, "run_build: a batched build is symmetric and unsharded");
    std::vector<LayerProg> hp((size_t)nb);
    for (int g = 0; g < nb; ++g) {
      BuildSpec sg = c.spec;
      sg.w_std = c.bw[g]; sg.b_std = c.bb[g]; sg.last_w_std = c.blw[g];
      SMN_TRY(make_prog(ctx, sg, &hp[(size_t)g]));
      set_fast<T>(&hp[(size_t)g], ntk);
    }
    void* pv = nullptr;
    SMN_TRY(smn_workspace(ctx, 6, sizeof(LayerProg) * (size_t)nb, &pv));
    progs_d = static_cast<LayerProg*>(pv);
    // (pageable source: the copy is staged before the call returns, so the vector may go)
    SMN_HIP(ctx, hipMemcpyAsync(progs_d, hp.data(), sizeof(LayerProg) * (size_t)nb, hipMemcpyHostToDevice, ctx->stream));
    SMN_HIP(ctx, hipStreamSynchronize(ctx->stream));
  }
  hipLaunchKernelGGL(diag_tables_kernel<T>, dim3((unsigned)((c.rows1 + 255) / 256), (unsigned)nb), dim3(256), 0, ctx->stream,
                     c.q1, c.rows1, prog, tab1, tlen, dg1, dgt1, progs_d, tab_bs);
  SMN_CHECK_LAUNCH(ctx);
  ctx->trace_ready = false;
  if (c.want_trace && c.symmetric && c.exact_diag && c.nbatch == 0 && !c.shard && c.nv0 > 0) {
    hipLaunchKernelGGL(table_trace_kernel<T>, dim3(1), dim3(256), 0, ctx->stream, dg1, c.nv0, ctx->d_scal + 1);
    SMN_CHECK_LAUNCH(ctx);
    ctx->trace_ready = true;
  }
  T* tab2 = tab1;
  if (!c.symmetric) {
    tab2 = tab1 + c.rows1;
    hipLaunchKernelGGL(diag_tables_kernel<T>, dim3((unsigned)((c.rows2 + 255) / 256)), dim3(256), 0, ctx->stream,
                       c.q2, c.rows2, prog, tab2, tlen, dg1 + c.rows1, dgt1 + c.rows1);
    SMN_CHECK_LAUNCH(ctx);
  }
  BuildArgs<T> a;
  a.x1 = static_cast<const T*>(c.x1p); a.x2 = static_cast<const T*>(c.x2p);
  a.ld1 = c.ld1; a.ld2 = c.ld2; a.kp = c.kp;
  const int64_t tm = c.rows1 / kTile, tn = c.rows2 / kTile;
  a.tiles_n = (int)tn; a.tiles_m = (int)tm; a.symmetric = c.symmetric; a.mirror = c.mirror;
  a.lower_skip = (!c.symmetric && c.lower_skip) ? 1 : 0;
  a.tab1 = tab1; a.tab2 = tab2; a.ldt1 = tlen; a.ldt2 = tlen; a.dg = dg1; a.dgt = dgt1;
  a.inv_d = (T)(1.0 / (double)c.d); a.prog = prog;
  a.progs = progs_d; a.tab_bs = tab_bs; a.out_bs = c.out_bs; a.nbatch = nb;
  a.row_off = c.row_off; a.col_off = c.col_off; a.exact_diag = c.exact_diag;
  a.store_mode = c.store_mode; a.out_rows = c.out_rows; a.out_cols = c.out_cols;
  a.nv0 = c.nv0; a.aug0 = c.aug0; a.nv1 = c.nv1;
  a.out_k = (c.get_mask & SMN_GET_NNGP) ? static_cast<T*>(c.out_k) : nullptr;
  a.out_t = ntk ? static_cast<T*>(c.out_t) : nullptr;
  a.ldo = c.ldo;
  int64_t ntiles = c.symmetric ? tm * (tm + 1) / 2 : tm * tn;
  a.use_map = 0;
  a.map = TileMap::make(tm, tn, c.symmetric);
  a.shard = c.shard;
  a.tlist = nullptr; a.tlist_n = 0;
  if (c.shard == 2) {
    if (!c.symmetric) return smn_fail(ctx, SMN_EINVAL, "run_build: shard mode needs the symmetric operands");
    a.cy_P = c.cy_P; a.cy_rank = c.cy_rank; a.cy_T = (int)tm; a.cy_np = c.cy_np;
    for (int g = 0; g <= c.cy_np; ++g) a.cy_c[g] = (int)c.cy_c[g];
    for (int g = 0; g < c.cy_np; ++g) a.cy_off[g] = c.cy_off[g];
    a.sh_k[0] = (c.get_mask & SMN_GET_NNGP) ? static_cast<T*>(c.shard_k[0]) : nullptr;
    a.sh_t[0] = ntk ? static_cast<T*>(c.shard_t[0]) : nullptr;
    ntiles = 0;
    for (int64_t j = 0; j * c.cy_P < tm; ++j) {
      const int64_t t = j * c.cy_P + ((j & 1) ? c.cy_P - 1 - c.cy_rank : c.cy_rank);
      if (t < tm) ntiles += t + 1;
    }
    if (ntiles == 0) return SMN_OK;
  } else if (c.shard) {
    if (!c.symmetric) return smn_fail(ctx, SMN_EINVAL, "run_build: shard mode needs the symmetric operands");
    int64_t cnt[2] = {0, 0};
    for (int w = 0; w < 2; ++w) {
      const int64_t rb = c.shard_rb[w], re = c.shard_re[w];
      if (re > rb && rb % kTile) return smn_fail(ctx, SMN_EINVAL, "run_build: shard block not tile aligned");
      const int64_t nt = re > rb ? (re - rb + kTile - 1) / kTile : 0, b0 = nt ? rb / kTile : 0;
      cnt[w] = nt * b0 + nt * (nt + 1) / 2;
      a.sh_b0[w] = (int)b0;
      a.sh_k[w] = (c.get_mask & SMN_GET_NNGP) ? static_cast<T*>(c.shard_k[w]) : nullptr;
      a.sh_t[w] = ntk ? static_cast<T*>(c.shard_t[w]) : nullptr;
      a.sh_ld[w] = c.shard_ld[w];
      a.sh_cols[w] = re;
    }
    a.sh_cnt0 = (int)cnt[0];
    ntiles = cnt[0] + cnt[1];
    if (ntiles == 0) return SMN_OK;
  }
  if (ctx->xcd_map && ntiles >= 512 && !a.lower_skip && !a.shard) {
    a.use_map = 1;
    ntiles = a.map.grid;
  }
  size_t lds = MainTile<T>::LDS_BYTES;
  const size_t tab_lds = (size_t)(prog.nsets * 2 + 1) * 2 * kTile * sizeof(T);
  if (tab_lds > lds) lds = tab_lds;
  ctx->corner_col = 0;
  if (c.split_corner > 0 && c.symmetric && !c.shard && c.nbatch == 0 && !c.mirror && ctx->stream_bulk && c.split_corner < tm) {
    // Two launches: everything but the bottom-right corner on the caller's stream, the corner on the bulk stream (CU-masked,
    // like the far updates it will be followed by) -- beside whatever the caller issues next on its own stream.
    const int TB = c.split_corner;
    SMN_TRY(split_tile_lists(ctx, (int)tm, TB));
    if (!ctx->ev_s0) SMN_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_s0, hipEventDisableTiming));
    if (!ctx->ev_corner) SMN_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_corner, hipEventDisableTiming));
    a.use_map = 0;
    a.tlist = ctx->tile_list; a.tlist_n = ctx->tile_list_na;
    SMN_TRY(launch_build<T>(ctx, a, (a.tlist_n + 7) / 8 * 8, lds, ntk, ctx->stream));
    // the corner starts BEHIND the first launch (it is meant to share the chip with the panel chain, not with the build)
    SMN_HIP(ctx, hipEventRecord(ctx->ev_s0, ctx->stream));
    SMN_HIP(ctx, hipStreamWaitEvent(ctx->stream_bulk, ctx->ev_s0, 0));
    a.tlist = ctx->tile_list + ctx->tile_list_na; a.tlist_n = ctx->tile_list_nb;
    const int rc = launch_build<T>(ctx, a, (a.tlist_n + 7) / 8 * 8, lds, ntk, ctx->stream_bulk);
    if (rc != SMN_OK) {   // the corner did not go out: nothing on the bulk stream to wait for, but the matrix is incomplete
      (void)hipStreamSynchronize(ctx->stream);
      return rc;
    }
    ctx->corner_col = (tm - TB) * kTile;
    return SMN_OK;
  }
  return launch_build<T>(ctx, a, ntiles, lds, ntk);
}


template <typename T, int NET, int ACT, bool NTK>
int launch_rec_t(smn_ctx* ctx, const RecArgs<T>& a, dim3 grid, size_t lds) {
  auto kern = recursion_sym_kernel<T, NET, ACT, NTK>;
  SMN_TRY(smn_allow_lds(ctx, reinterpret_cast<const void*>(kern), lds));
  {
    ProfScope ps(ctx, PROF_RECURSION, ctx->stream);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, ctx->stream, a);
  }
  SMN_CHECK_LAUNCH(ctx);
  return SMN_OK;
}

template <typename T>
int launch_rec(smn_ctx* ctx, const RecArgs<T>& a, dim3 grid, size_t lds, bool ntk) {
  const int net = a.prog.net, act = a.prog.act;
#define REC_CASE(N, A)                                                        \
  if (net == N && act == A)                                                   \
    return ntk ? launch_rec_t<T, N, A, true>(ctx, a, grid, lds) : launch_rec_t<T, N, A, false>(ctx, a, grid, lds);
  REC_CASE(NET_MLP, ACT_RELU)
  REC_CASE(NET_MLP, ACT_ERF)
  REC_CASE(NET_RESNET, ACT_RELU)
  REC_CASE(NET_RESNET, ACT_ERF)
#undef REC_CASE
  return smn_fail(ctx, SMN_EINVAL, "recursion: bad net/act");
}

template <typename T>
__global__ void cast_from_double_kernel(const double* __restrict__ s, T* __restrict__ d, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) d[i] = (T)s[i];
}
template <typename T>
int recursion_t(smn_ctx* ctx, const BuildSpec& spec, const void* k0, int64_t n1, int64_t n2, int64_t ldk0,
                const void* q1, const void* q2, int symmetric, int get_mask, void* nngp, void* ntk, int64_t ldk) {
  LayerProg prog;
  SMN_TRY(make_prog(ctx, spec, &prog));
  if (prog.net == NET_NONE) return smn_fail(ctx, SMN_EINVAL, "recursion needs a net");
  const bool want_ntk = (get_mask & SMN_GET_NTK) != 0;
  set_fast<T>(&prog, want_ntk);
  const int trows = 2 * prog.nsets + 2;
  const int64_t tlen = n1 + n2;
  void* tabv = nullptr;
  SMN_TRY(smn_workspace(ctx, 1, sizeof(T) * (size_t)trows * tlen, &tabv));
  T* tab1 = static_cast<T*>(tabv);
  T* dg1 = tab1 + (int64_t)(2 * prog.nsets) * tlen;
  T* dgt1 = dg1 + tlen;
  // one table launch per operand, straight from q in the compute type; the same operand on both sides (the gradient's
  // symmetric call) shares one set of tables
  const bool same = q1 == q2 && n1 == n2;
  hipLaunchKernelGGL((diag_tables_kernel<T, T>), dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, ctx->stream,
                     static_cast<const T*>(q1), n1, prog, tab1, tlen, dg1, dgt1, static_cast<const LayerProg*>(nullptr), (int64_t)0);
  if (!same)
    hipLaunchKernelGGL((diag_tables_kernel<T, T>), dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, ctx->stream,
                       static_cast<const T*>(q2), n2, prog, tab1 + n1, tlen, dg1 + n1, dgt1 + n1, static_cast<const LayerProg*>(nullptr), (int64_t)0);
  SMN_CHECK_LAUNCH(ctx);
  RecArgs<T> a;
  a.k0 = static_cast<const T*>(k0); a.ldk0 = ldk0; a.n1 = n1; a.n2 = n2;
  a.tab1 = tab1; a.tab2 = same ? tab1 : tab1 + n1; a.ldt1 = tlen; a.ldt2 = tlen; a.dg = dg1; a.dgt = dgt1;
  a.prog = prog; a.exact_diag = symmetric;
  a.out_k = (get_mask & SMN_GET_NNGP) ? static_cast<T*>(nngp) : nullptr;
  a.out_t = want_ntk ? static_cast<T*>(ntk) : nullptr;
  a.ldo = ldk; a.rows_per_block = 0;
  // 16-byte vector path needs aligned rows
  if (ldk0 % (16 / sizeof(T)) || ldk % (16 / sizeof(T)) || (reinterpret_cast<uintptr_t>(k0) & 15) ||
      (a.out_k && (reinterpret_cast<uintptr_t>(a.out_k) & 15)) || (a.out_t && (reinterpret_cast<uintptr_t>(a.out_t) & 15)))
    return smn_fail(ctx, SMN_EINVAL, "smn_recursion: k0/out must be 16-byte aligned with ld %% %d == 0", (int)(16 / sizeof(T)));
  constexpr int TS = 64;
  const int64_t t1 = (n1 + TS - 1) / TS, t2 = (n2 + TS - 1) / TS;
  const bool lower = symmetric && n1 == n2;
  const int64_t ntiles = lower ? t1 * (t1 + 1) / 2 : t1 * t2;
  if (ntiles >= (int64_t)INT32_MAX) return smn_fail(ctx, SMN_ENOTSUP, "smn_recursion: too many tiles");
  a.sym_tiles = lower ? 1 : 2;
  const size_t slds = sizeof(T) * ((size_t)(prog.nsets * 2 + 1) * 2 * TS + (lower ? (size_t)(want_ntk ? 2 : 1) * TS * (TS + 1) : 0));
  return launch_rec<T>(ctx, a, dim3((unsigned)ntiles), slds, want_ntk);
}

}  // namespace

int pad_rows(smn_ctx* ctx, int dtype, const void* src, int64_t n, int64_t lds, int64_t d,
             void* dst, int64_t rows_pad, int64_t kp, double* q, int64_t rows_a, const void* src2, int64_t n2, int64_t lds2) {
  const unsigned blocks = (unsigned)((rows_pad + 3) / 4);
  ProfScope ps(ctx, PROF_PREP, ctx->stream);
  if (dtype == SMN_F64)
    hipLaunchKernelGGL(pad_rows_kernel<double>, dim3(blocks), dim3(256), 0, ctx->stream, static_cast<const double*>(src),
                       n, lds, d, static_cast<double*>(dst), rows_pad, kp, 1.0 / (double)d, q, rows_a,
                       static_cast<const double*>(src2), n2, lds2);
  else
    hipLaunchKernelGGL(pad_rows_kernel<float>, dim3(blocks), dim3(256), 0, ctx->stream, static_cast<const float*>(src),
                       n, lds, d, static_cast<float*>(dst), rows_pad, kp, 1.0 / (double)d, q, rows_a,
                       static_cast<const float*>(src2), n2, lds2);
  SMN_CHECK_LAUNCH(ctx);
  return SMN_OK;
}

int run_build(smn_ctx* ctx, const BuildCall& c) {
  return c.spec.dtype == SMN_F64 ? run_build_t<double>(ctx, c) : run_build_t<float>(ctx, c);
}

// Size of the corner of a split build: the corner's tiles take about as long on the bulk stream's CUs as the first
// super-panel's panel chain (1.2 ms at N = 16384: 8 sub-panels at 71 us on the reserved CUs, strips, two near updates), and
// what is left fills a whole number of rounds of two workgroups per CU as nearly as possible.  Measured
// (profiles/r04_split_build.txt): N = 16384 20.90 -> 20.68 ms, N = 32768 105.40 -> 105.34; N = 8000 and 12345 lose 0.5-1 %
// (their first chain is shorter than any corner worth a launch): only from 112 tile rows on.
int split_corner_tiles(const smn_ctx* ctx, int64_t tiles) {
  if (!ctx->stream_bulk || tiles * kTile < ctx->chain_min_n || tiles < 112 || tiles > 0x7fff) return 0;
  const int64_t total = tiles * (tiles + 1) / 2, round = 2 * (int64_t)ctx->num_cu;
  int best = 0;
  double best_fill = -1.0;
  for (int tb = 32; tb <= 56; ++tb) {
    const int64_t na = total - (int64_t)tb * (tb + 1) / 2;
    const double r = (double)na / (double)round;
    const double fill = r - std::floor(r);            // 0.98 = the last round almost full; 0.0 = exactly full
    const double score = fill < 1e-9 ? 1.0 : fill;
    if (score > best_fill) { best_fill = score; best = tb; }
  }
  return best;
}

// ------------------------------------------------------------------ public entry points
static int check_common(smn_ctx* ctx, int dtype, int64_t n1, int64_t n2, int64_t d) {
  if (!ctx) return SMN_EINVAL;
  if (dtype != SMN_F32 && dtype != SMN_F64) return smn_fail(ctx, SMN_EINVAL, "bad dtype %d", dtype);
  if (n1 <= 0 || n2 <= 0 || d <= 0) return smn_fail(ctx, SMN_EINVAL, "empty operand (n1=%lld n2=%lld d=%lld)",
                                                    (long long)n1, (long long)n2, (long long)d);
  return SMN_OK;
}

// Shared by smn_kernel_mlp / smn_kernel_mlp_rows / smn_gram.
static int build_public(smn_ctx* ctx, const BuildSpec& spec, const void* x1, int64_t n1, int64_t ldx1,
                        const void* x2, int64_t n2, int64_t ldx2, int64_t d, int get_mask, int fill,
                        int64_t row_begin, int64_t row_end, void* out_k, void* out_t, int64_t ldk,
                        void* q1_out, void* q2_out, bool lower_rows = false) {
  const int dtype = spec.dtype;
  const size_t es = dtype_size(dtype);
  const bool sym = (x2 == nullptr);
  const int64_t kp = k_pad(dtype, d);
  const int64_t r1 = round_up(n1, kTile), r2 = sym ? r1 : round_up(n2, kTile);
  void* xs = nullptr;
  const size_t xbytes = es * (size_t)kp * (size_t)(r1 + (sym ? 0 : r2)) + sizeof(double) * (size_t)(r1 + r2);
  SMN_TRY(smn_workspace(ctx, 0, xbytes, &xs));
  double* q1 = static_cast<double*>(xs);
  double* q2 = q1 + r1;
  char* x1p = reinterpret_cast<char*>(q2 + r2);
  char* x2p = sym ? x1p : x1p + es * (size_t)kp * (size_t)r1;
  SMN_TRY(pad_rows(ctx, dtype, x1, n1, ldx1, d, x1p, r1, kp, q1));
  if (!sym) SMN_TRY(pad_rows(ctx, dtype, x2, n2, ldx2, d, x2p, r2, kp, q2));
  BuildCall c{};
  c.spec = spec;
  c.kp = (int)kp; c.d = d; c.get_mask = get_mask;
  c.out_k = out_k; c.out_t = out_t; c.ldo = ldk;
  c.store_mode = STORE_BOUNDS;
  if (row_end > row_begin) {            // row shard of the symmetric kernel: rows [rb,re) x all columns
    const int64_t rb = row_begin, nr = row_end - row_begin;
    const int64_t rr = round_up(nr, kTile);
    if (rb % kTile == 0) {              // tile-aligned shard: its rows are a slice of the padded copy
      c.x1p = x1p + es * (size_t)kp * (size_t)rb; c.ld1 = kp; c.rows1 = rr; c.q1 = q1 + rb;
    } else {                            // a shard's last tile may reach past the padded copy -> re-pad from the source
      void* xr = nullptr;
      SMN_TRY(smn_workspace(ctx, 3, es * (size_t)kp * (size_t)rr + sizeof(double) * (size_t)rr, &xr));
      double* qr = static_cast<double*>(xr);
      char* xrp = reinterpret_cast<char*>(qr + rr);
      SMN_TRY(pad_rows(ctx, dtype, static_cast<const char*>(x1) + es * (size_t)rb * (size_t)ldx1, nr, ldx1, d, xrp, rr, kp, qr));
      c.x1p = xrp; c.ld1 = kp; c.rows1 = rr; c.q1 = qr;
    }
    c.x2p = x1p; c.ld2 = kp; c.rows2 = r1; c.q2 = q1;
    c.symmetric = 0; c.mirror = 0; c.row_off = rb; c.col_off = 0; c.exact_diag = 1;
    c.out_rows = nr; c.out_cols = n1;
    if (lower_rows) {                   // lower trapezoid: columns [0, row_end) only, tiles above the diagonal dropped
      c.rows2 = round_up(row_end, kTile);
      c.out_cols = row_end;
      c.lower_skip = 1;
    }
    return run_build(ctx, c);
  }
  c.x1p = x1p; c.ld1 = kp; c.rows1 = r1; c.q1 = q1;
  c.x2p = x2p; c.ld2 = kp; c.rows2 = r2; c.q2 = sym ? q1 : q2;
  c.symmetric = sym ? 1 : 0; c.mirror = (sym && fill == SMN_FILL_FULL) ? 1 : 0;
  c.exact_diag = sym ? 1 : 0;
  c.out_rows = n1; c.out_cols = sym ? n1 : n2;
  SMN_TRY(run_build(ctx, c));
  if (q1_out || q2_out) {
    const int64_t m1 = n1, m2 = sym ? n1 : n2;
    if (dtype == SMN_F64) {
      if (q1_out) SMN_HIP(ctx, hipMemcpyAsync(q1_out, q1, 8 * (size_t)m1, hipMemcpyDeviceToDevice, ctx->stream));
      if (q2_out) SMN_HIP(ctx, hipMemcpyAsync(q2_out, sym ? q1 : q2, 8 * (size_t)m2, hipMemcpyDeviceToDevice, ctx->stream));
    } else {
      if (q1_out) hipLaunchKernelGGL(cast_from_double_kernel<float>, dim3((unsigned)((m1 + 255) / 256)), dim3(256), 0,
                                     ctx->stream, q1, static_cast<float*>(q1_out), m1);
      if (q2_out) hipLaunchKernelGGL(cast_from_double_kernel<float>, dim3((unsigned)((m2 + 255) / 256)), dim3(256), 0,
                                     ctx->stream, sym ? q1 : q2, static_cast<float*>(q2_out), m2);
      SMN_CHECK_LAUNCH(ctx);
    }
  }
  return SMN_OK;
}

extern "C" int smn_kernel_mlp(smn_ctx* ctx, int dtype, int net, int act, int num_hiddens, double w_std,
                              double b_std, double last_w_std, const void* x1_d, int64_t n1, int64_t ldx1,
                              const void* x2_d, int64_t n2, int64_t ldx2, int64_t d, int get_mask, int fill,
                              void* nngp_d, void* ntk_d, int64_t ldk) {
  SMN_TRY(check_common(ctx, dtype, n1, x2_d ? n2 : 1, d));
  SMN_ENTER(ctx);
  if (!(get_mask & (SMN_GET_NNGP | SMN_GET_NTK))) return smn_fail(ctx, SMN_EINVAL, "empty get mask");
  if (((get_mask & SMN_GET_NNGP) && !nngp_d) || ((get_mask & SMN_GET_NTK) && !ntk_d))
    return smn_fail(ctx, SMN_EINVAL, "requested output pointer is NULL");
  BuildSpec s{dtype, net, act, num_hiddens, w_std, b_std, last_w_std};
  return build_public(ctx, s, x1_d, n1, ldx1, x2_d, n2, ldx2, d, get_mask, fill, 0, 0, nngp_d, ntk_d, ldk, nullptr, nullptr);
}

extern "C" int smn_kernel_mlp_rows(smn_ctx* ctx, int dtype, int net, int act, int num_hiddens, double w_std,
                                   double b_std, double last_w_std, const void* x_d, int64_t n, int64_t ldx,
                                   int64_t d, int64_t row_begin, int64_t row_end, int get_mask,
                                   void* nngp_rows_d, void* ntk_rows_d, int64_t ldk) {
  SMN_TRY(check_common(ctx, dtype, n, 1, d));
  SMN_ENTER(ctx);
  if (row_begin < 0 || row_end > n || row_end <= row_begin)
    return smn_fail(ctx, SMN_EINVAL, "bad row range [%lld,%lld)", (long long)row_begin, (long long)row_end);
  BuildSpec s{dtype, net, act, num_hiddens, w_std, b_std, last_w_std};
  return build_public(ctx, s, x_d, n, ldx, nullptr, 0, 0, d, get_mask, SMN_FILL_FULL, row_begin, row_end,
                      nngp_rows_d, ntk_rows_d, ldk, nullptr, nullptr);
}

extern "C" int smn_kernel_mlp_lower_rows(smn_ctx* ctx, int dtype, int net, int act, int num_hiddens, double w_std,
                                         double b_std, double last_w_std, const void* x_d, int64_t n, int64_t ldx,
                                         int64_t d, int64_t row_begin, int64_t row_end, int get_mask,
                                         void* nngp_rows_d, void* ntk_rows_d, int64_t ldk) {
  SMN_TRY(check_common(ctx, dtype, n, 1, d));
  SMN_ENTER(ctx);
  if (row_begin < 0 || row_end > n || row_end <= row_begin)
    return smn_fail(ctx, SMN_EINVAL, "bad row range [%lld,%lld)", (long long)row_begin, (long long)row_end);
  if (ldk < row_end) return smn_fail(ctx, SMN_EINVAL, "ldk %lld < row_end %lld", (long long)ldk, (long long)row_end);
  BuildSpec s{dtype, net, act, num_hiddens, w_std, b_std, last_w_std};
  return build_public(ctx, s, x_d, n, ldx, nullptr, 0, 0, d, get_mask, SMN_FILL_FULL, row_begin, row_end,
                      nngp_rows_d, ntk_rows_d, ldk, nullptr, nullptr, true);
}

// One rank's whole share of the paired layout: the lower trapezoids of row blocks `rank` and 2 nranks - 1 - rank, packed
// into the rank's chunk (low block first).
extern "C" int smn_kernel_mlp_shard(smn_ctx* ctx, int dtype, int net, int act, int num_hiddens, double w_std,
                                    double b_std, double last_w_std, const void* x_d, int64_t n, int64_t ldx,
                                    int64_t d, int nranks, int rank, int64_t block_rows, int get_mask,
                                    void* nngp_chunk_d, void* ntk_chunk_d) {
  SMN_TRY(check_common(ctx, dtype, n, 1, d));
  SMN_ENTER(ctx);
  if (nranks <= 0 || rank < 0 || rank >= nranks || block_rows <= 0 || block_rows % kTile ||
      2 * (int64_t)nranks * block_rows < n)
    return smn_fail(ctx, SMN_EINVAL, "smn_kernel_mlp_shard: bad geometry (n=%lld ranks=%d rank=%d block_rows=%lld)",
                    (long long)n, nranks, rank, (long long)block_rows);
  if (!x_d || ((get_mask & SMN_GET_NNGP) && !nngp_chunk_d) || ((get_mask & SMN_GET_NTK) && !ntk_chunk_d))
    return smn_fail(ctx, SMN_EINVAL, "smn_kernel_mlp_shard: null pointer");
  const size_t es = dtype_size(dtype);
  const int64_t kp = k_pad(dtype, d), r1 = round_up(n, kTile), h = block_rows;
  void* xs = nullptr;
  SMN_TRY(smn_workspace(ctx, 0, es * (size_t)kp * (size_t)r1 + sizeof(double) * (size_t)r1, &xs));
  double* q1 = static_cast<double*>(xs);
  char* x1p = reinterpret_cast<char*>(q1 + r1);
  SMN_TRY(pad_rows(ctx, dtype, x_d, n, ldx, d, x1p, r1, kp, q1));
  BuildCall c{};
  c.spec = BuildSpec{dtype, net, act, num_hiddens, w_std, b_std, last_w_std};
  c.kp = (int)kp; c.d = d; c.get_mask = get_mask;
  c.x1p = x1p; c.ld1 = kp; c.rows1 = r1; c.q1 = q1;
  c.x2p = x1p; c.ld2 = kp; c.rows2 = r1; c.q2 = q1;
  c.symmetric = 1; c.mirror = 0; c.exact_diag = 1;
  c.store_mode = STORE_BOUNDS; c.out_rows = n; c.out_cols = n;
  c.shard = 1;
  const int64_t blk[2] = {rank, 2 * (int64_t)nranks - 1 - rank};
  for (int w = 0; w < 2; ++w) {
    const int64_t ld = (blk[w] + 1) * h;
    int64_t rb = blk[w] * h, re = blk[w] * h + h;
    if (rb > n) rb = n;
    if (re > n) re = n;
    c.shard_rb[w] = rb;
    c.shard_re[w] = re;
    c.shard_ld[w] = ld;
    const size_t off = es * (size_t)(w == 0 ? 0 : h * (rank + 1) * h);   // low block first, then the high one
    c.shard_k[w] = nngp_chunk_d ? static_cast<char*>(nngp_chunk_d) + off : nullptr;
    c.shard_t[w] = ntk_chunk_d ? static_cast<char*>(ntk_chunk_d) + off : nullptr;
  }
  return run_build(ctx, c);
}

int col_pieces_make(smn_ctx* ctx, int64_t n, int nranks, int npieces, const int64_t* piece_cols, ColPieces* out) {
  if (n <= 0 || nranks <= 0 || npieces <= 0 || npieces > kMaxColPieces || !piece_cols)
    return smn_fail(ctx, SMN_EINVAL, "column pieces: n=%lld ranks=%d pieces=%d (at most %d)", (long long)n, nranks, npieces, kMaxColPieces);
  ColPieces cp;
  cp.P = nranks; cp.np = npieces; cp.T = (n + kTile - 1) / kTile;
  for (int g = 0; g <= npieces; ++g) cp.c[g] = piece_cols[g];
  if (cp.c[0] != 0 || cp.c[npieces] != cp.T)
    return smn_fail(ctx, SMN_EINVAL, "column pieces must span the tile columns [0, %lld)", (long long)cp.T);
  for (int g = 0; g < npieces; ++g) {
    if (cp.c[g + 1] <= cp.c[g])
      return smn_fail(ctx, SMN_EINVAL, "column piece %d = [%lld, %lld): boundaries must ascend", g, (long long)cp.c[g],
                      (long long)cp.c[g + 1]);
    cp.off[g + 1] = cp.off[g] + cp.count(g);
  }
  *out = cp;
  return SMN_OK;
}

// One rank's whole share of the cyclic column-first layout (internal.hpp ColPieces) in ONE launch on every CU: all lower
// tiles of its tile rows, stored piece by piece into its chunk, ready for smn_shard_exchange_cols.
extern "C" int smn_kernel_mlp_shard_cols(smn_ctx* ctx, int dtype, int net, int act, int num_hiddens, double w_std,
                                         double b_std, double last_w_std, const void* x_d, int64_t n, int64_t ldx,
                                         int64_t d, int nranks, int rank, int npieces, const int64_t* piece_cols,
                                         int get_mask, void* nngp_chunk_d, void* ntk_chunk_d) {
  SMN_TRY(check_common(ctx, dtype, n, 1, d));
  SMN_ENTER(ctx);
  if (rank < 0 || rank >= nranks) return smn_fail(ctx, SMN_EINVAL, "smn_kernel_mlp_shard_cols: rank %d of %d", rank, nranks);
  if (!x_d || !(get_mask & (SMN_GET_NNGP | SMN_GET_NTK)) || ((get_mask & SMN_GET_NNGP) && !nngp_chunk_d) ||
      ((get_mask & SMN_GET_NTK) && !ntk_chunk_d))
    return smn_fail(ctx, SMN_EINVAL, "smn_kernel_mlp_shard_cols: null pointer or empty get mask");
  ColPieces cp;
  SMN_TRY(col_pieces_make(ctx, n, nranks, npieces, piece_cols, &cp));
  const size_t es = dtype_size(dtype);
  const int64_t kp = k_pad(dtype, d), r1 = round_up(n, kTile);
  void* xs = nullptr;
  SMN_TRY(smn_workspace(ctx, 0, es * (size_t)kp * (size_t)r1 + sizeof(double) * (size_t)r1, &xs));
  double* q1 = static_cast<double*>(xs);
  char* x1p = reinterpret_cast<char*>(q1 + r1);
  SMN_TRY(pad_rows(ctx, dtype, x_d, n, ldx, d, x1p, r1, kp, q1));
  BuildCall c{};
  c.spec = BuildSpec{dtype, net, act, num_hiddens, w_std, b_std, last_w_std};
  c.kp = (int)kp; c.d = d; c.get_mask = get_mask;
  c.x1p = x1p; c.ld1 = kp; c.rows1 = r1; c.q1 = q1;
  c.x2p = x1p; c.ld2 = kp; c.rows2 = r1; c.q2 = q1;
  c.symmetric = 1; c.mirror = 0; c.exact_diag = 1;
  c.store_mode = STORE_BOUNDS; c.out_rows = n; c.out_cols = n;
  c.shard = 2;
  c.cy_P = nranks; c.cy_rank = rank; c.cy_np = npieces;
  for (int g = 0; g <= npieces; ++g) c.cy_c[g] = cp.c[g];
  for (int g = 0; g < npieces; ++g) c.cy_off[g] = cp.off[g];
  c.shard_k[0] = nngp_chunk_d;
  c.shard_t[0] = ntk_chunk_d;
  return run_build(ctx, c);
}

// smn_gram of x with itself, lower 128x128 tiles only (diagonal tiles whole): what the symmetric recursion and the gradient
// contraction read; saves the mirrored store of the full form.
int gram_lower(smn_ctx* ctx, int dtype, const void* x_d, int64_t n, int64_t ldx, int64_t d, void* k0_d, int64_t ldk, void* q_d) {
  BuildSpec s{dtype, NET_NONE, SMN_ACT_RELU, 0, 1.0, 0.0, 1.0};
  return build_public(ctx, s, x_d, n, ldx, nullptr, 0, 0, d, SMN_GET_NNGP, SMN_FILL_LOWER, 0, 0, k0_d, nullptr, ldk, q_d, nullptr);
}

extern "C" int smn_gram(smn_ctx* ctx, int dtype, const void* x1_d, int64_t n1, int64_t ldx1, const void* x2_d,
                        int64_t n2, int64_t ldx2, int64_t d, void* k0_d, int64_t ldk, void* q1_d, void* q2_d) {
  SMN_TRY(check_common(ctx, dtype, n1, x2_d ? n2 : 1, d));
  SMN_ENTER(ctx);
  BuildSpec s{dtype, NET_NONE, SMN_ACT_RELU, 0, 1.0, 0.0, 1.0};
  return build_public(ctx, s, x1_d, n1, ldx1, x2_d, n2, ldx2, d, SMN_GET_NNGP, SMN_FILL_FULL, 0, 0, k0_d, nullptr,
                      ldk, q1_d, q2_d);
}

extern "C" int smn_recursion(smn_ctx* ctx, int dtype, int net, int act, int num_hiddens, double w_std, double b_std,
                             double last_w_std, const void* k0_d, int64_t n1, int64_t n2, int64_t ldk0,
                             const void* q1_d, const void* q2_d, int symmetric, int get_mask, void* nngp_d,
                             void* ntk_d, int64_t ldk) {
  SMN_TRY(check_common(ctx, dtype, n1, n2, 1));
  SMN_ENTER(ctx);
  BuildSpec s{dtype, net, act, num_hiddens, w_std, b_std, last_w_std};
  if (dtype == SMN_F64)
    return recursion_t<double>(ctx, s, k0_d, n1, n2, ldk0, q1_d, q2_d, symmetric, get_mask, nngp_d, ntk_d, ldk);
  return recursion_t<float>(ctx, s, k0_d, n1, n2, ldk0, q1_d, q2_d, symmetric, get_mask, nngp_d, ntk_d, ldk);
}
