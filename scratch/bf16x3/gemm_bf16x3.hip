// R&D probe, NOT part of the product: C = A B^T / d with f32 operands split on the fly into three bf16 planes
// (x = hi + mid + lo exactly, 8 + 8 + 8 significand bits) and SIX bf16 MFMA products per k-slice
//   hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid        (dropped: mid*lo, lo*mid, lo*lo <= 2^-23 |a||b|)
// accumulated in f32 by v_mfma_f32_32x32x16_bf16.  Question it answers: how fast, and how close to an exact-f32
// GEMM, would the Gram / trailing-update engine be on the 16x faster bf16 pipe.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 128, BK = 32;          // 128 x 128 tile, 32 f32 k-values per step
constexpr int ROWB = 80;                  // bytes per LDS row of one plane: 32 bf16 + 16 pad (conflict-free b128 reads)
constexpr int PLANE = BM * ROWB;          // one plane of one operand
constexpr int LDS_BYTES = 2 * 3 * PLANE;  // A and B, three planes each: 61,440 B -> two workgroups per CU

__device__ __forceinline__ void split3(float x, uint16_t& hi, uint16_t& mid, uint16_t& lo) {
  const uint32_t b = __float_as_uint(x);
  hi = (uint16_t)(b >> 16);
  const float r1 = x - __uint_as_float(b & 0xFFFF0000u);           // exact
  const uint32_t b1 = __float_as_uint(r1);
  mid = (uint16_t)(b1 >> 16);
  const float r2 = r1 - __uint_as_float(b1 & 0xFFFF0000u);         // exact, <= 8 significant bits left
  lo = (uint16_t)(__float_as_uint(r2) >> 16);
}

__global__ void __launch_bounds__(256, 2) gemm_bf16x3(const float* __restrict__ A, const float* __restrict__ B,
                                                      float* __restrict__ C, int n, int m, int k, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tiles_n = m / BM;
  const int tr = blockIdx.x / tiles_n, tc = blockIdx.x % tiles_n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int lrow = tid >> 3, lchunk = tid & 7;                      // global -> LDS: rows lrow + 32 p, 4 floats at 4*lchunk
  const float* ga = A + (int64_t)(tr * BM + lrow) * k + lchunk * 4;
  const float* gb = B + (int64_t)(tc * BM + lrow) * k + lchunk * 4;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  f32x4 ra[4], rb[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    ra[p] = *reinterpret_cast<const f32x4*>(ga + (int64_t)(32 * p) * k);
    rb[p] = *reinterpret_cast<const f32x4*>(gb + (int64_t)(32 * p) * k);
  }
  const int nk = k / BK;
  for (int kt = 0; kt < nk; ++kt) {
    // registers -> three bf16 planes in LDS
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      uint16_t h[4], mi[4], l[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) split3(ra[p][e], h[e], mi[e], l[e]);
      char* dst = smem + (lrow + 32 * p) * ROWB + lchunk * 8;
      *reinterpret_cast<uint2*>(dst) = make_uint2(h[0] | (uint32_t)h[1] << 16, h[2] | (uint32_t)h[3] << 16);
      *reinterpret_cast<uint2*>(dst + PLANE) = make_uint2(mi[0] | (uint32_t)mi[1] << 16, mi[2] | (uint32_t)mi[3] << 16);
      *reinterpret_cast<uint2*>(dst + 2 * PLANE) = make_uint2(l[0] | (uint32_t)l[1] << 16, l[2] | (uint32_t)l[3] << 16);
#pragma unroll
      for (int e = 0; e < 4; ++e) split3(rb[p][e], h[e], mi[e], l[e]);
      dst += 3 * PLANE;
      *reinterpret_cast<uint2*>(dst) = make_uint2(h[0] | (uint32_t)h[1] << 16, h[2] | (uint32_t)h[3] << 16);
      *reinterpret_cast<uint2*>(dst + PLANE) = make_uint2(mi[0] | (uint32_t)mi[1] << 16, mi[2] | (uint32_t)mi[3] << 16);
      *reinterpret_cast<uint2*>(dst + 2 * PLANE) = make_uint2(l[0] | (uint32_t)l[1] << 16, l[2] | (uint32_t)l[3] << 16);
    }
    __syncthreads();
    if (kt + 1 < nk) {                                              // next step's loads fly under the MFMAs
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        ra[p] = *reinterpret_cast<const f32x4*>(ga + (int64_t)(32 * p) * k + (kt + 1) * BK);
        rb[p] = *reinterpret_cast<const f32x4*>(gb + (int64_t)(32 * p) * k + (kt + 1) * BK);
      }
    }
    const int fr = lane & 31, fh = lane >> 5;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 a[2][3], b[2][3];
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          a[t][pl] = *reinterpret_cast<const bf16x8*>(smem + pl * PLANE + (wr * 64 + t * 32 + fr) * ROWB + 32 * kk + 16 * fh);
          b[t][pl] = *reinterpret_cast<const bf16x8*>(smem + (3 + pl) * PLANE + (wc * 64 + t * 32 + fr) * ROWB + 32 * kk + 16 * fh);
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          f32x16 c = acc[i][j];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);   // hi*lo
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);   // lo*hi
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);   // mid*mid
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);   // hi*mid
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);   // mid*hi
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);   // hi*hi
          acc[i][j] = c;
        }
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = tr * BM + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        const int col = tc * BM + wc * 64 + j * 32 + (lane & 31);
        C[(int64_t)row * m + col] = acc[i][j][e] * scale;
      }
}

// A [n,k], B [m,k], C [n,m] device pointers; n, m multiples of 128, k of 32.  Returns the average ms over `reps`.
extern "C" double bf16x3_gemm(const float* A, const float* B, float* C, int n, int m, int k, int reps) {
  if (n % BM || m % BM || k % BK) return -1.0;
  hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_bf16x3), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  const dim3 grid((unsigned)((n / BM) * (m / BM)));
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(gemm_bf16x3, grid, dim3(256), LDS_BYTES, 0, A, B, C, n, m, k, 1.0f / (float)k);   // warm-up
  hipEventRecord(e0, 0);
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL(gemm_bf16x3, grid, dim3(256), LDS_BYTES, 0, A, B, C, n, m, k, 1.0f / (float)k);
  hipEventRecord(e1, 0);
  if (hipEventSynchronize(e1) != hipSuccess) return -2.0;
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return ms / reps;
}
