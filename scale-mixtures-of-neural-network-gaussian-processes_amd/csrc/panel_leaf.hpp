// panel_leaf.hpp — the register-resident 16x16 leaf of the Cholesky panel (panelr_kernel, cholesky.hip).
//
// Every 16-lane row of a wave holds a copy of the 16x16 diagonal tile, one tile row per lane (lane & 15), in
// D[16]; every lane also holds 16 entries of ITS OWN matrix row (the tile's columns) in V[16].  Sixteen elimination
// steps factor the tile (redundantly in every 16-lane row) and carry the lane's own row through the same column
// operations -- x L^T = b, a true TRSM without an inverse -- with the multipliers L_ck read from lane c of the
// 16-lane row by DPP row_newbcast inside the fused multiply-add itself: no LDS, no barrier, no cross-wave traffic.
//   step k:  rinv = rsqrt(d_kk);  D_k *= rinv;  V_k *= rinv;  for c > k:  D_c -= bcast_c(D_k) D_k,  V_c -= bcast_c(D_k) V_k
// On exit V = b L^-T (for a lane whose own row IS tile row r: row r of L, with garbage right of the diagonal that
// nothing reads), D = L rows (unused).  288 vector instructions per 16 columns (f32); a non-positive pivot turns
// into NaN (rsqrt of a negative number, 0 * inf) and propagates, which is how the caller detects it.
// Replaces the in-LDS micro-panels of panel_kernel on the path of lax.linalg.cholesky (spax/utils.py:179).
#pragma once
#include <hip/hip_runtime.h>

namespace leaf {

template <int K>
__device__ __forceinline__ float bcast(float x) {   // lane K of this lane's 16-lane row
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x150 + K, 0xf, 0xf, false));
}
template <int K>
__device__ __forceinline__ double bcast(double x) {
  const long long b = __builtin_bit_cast(long long, x);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)b, 0x150 + K, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), 0x150 + K, 0xf, 0xf, false);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

__device__ __forceinline__ float rsqrt_pivot(float d) { return __builtin_amdgcn_rsqf(d); }
// v_rsq_f64 (~2^-26 relative) + one third-order correction: y (1 + e/2 + 3 e^2/8), e = 1 - d y^2 -> below 2^-70 before
// rounding; 6 instructions against ~60 for 1.0 / sqrt(d)
__device__ __forceinline__ double rsqrt_pivot(double d) {
  const double y = __builtin_amdgcn_rsq(d);
  const double e = __builtin_fma(-(d * y), y, 1.0);
  return __builtin_fma(y * e, __builtin_fma(0.375, e, 0.5), y);
}

#define LEAF_RINV(K) R[K] = rsqrt_pivot(bcast<K>(D[K]));

// R[k] = 1/sqrt(d_kk) of every step stays with the caller (a multi-pass panel workgroup stores them for its later passes).
__device__ __forceinline__ void run(float (&D)[16], float (&V)[16], float (&R)[16]) {
#define LEAF_STEPS_F32
#include "panel_leaf_steps.inc"
#undef LEAF_STEPS_F32
}
__device__ __forceinline__ void run(double (&D)[16], double (&V)[16], double (&R)[16]) {
#define LEAF_STEPS_F64
#include "panel_leaf_steps.inc"
#undef LEAF_STEPS_F64
}
#undef LEAF_RINV

// The v-steps alone: the lane's own row carried through a tile that is already factored (D = rows of L, R = the reciprocal
// pivots the factoring pass kept).  The instructions of the fused leaf's v-steps on the same operands: the same bits.
__device__ __forceinline__ void solve(const float (&D)[16], const float (&R)[16], float (&V)[16]) {
#define LEAF_SOLVE_F32
#include "panel_leaf_steps.inc"
#undef LEAF_SOLVE_F32
}
__device__ __forceinline__ void solve(const double (&D)[16], const double (&R)[16], double (&V)[16]) {
#define LEAF_SOLVE_F64
#include "panel_leaf_steps.inc"
#undef LEAF_SOLVE_F64
}

}  // namespace leaf
