// R&D probe, NOT part of the product (see gemm_bf16x3.hip): same six-product emulation, but the f32 operand is split
// into its three bf16 planes ONCE (split_planes), and the GEMM streams the planes (6 bytes per element) with no
// conversion work in its loop.
#include <hip/hip_runtime.h>
#include <cstdint>

typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BK = 32;
constexpr int ROWB = 80;                  // 32 bf16 + 16 B pad per LDS row of one plane
constexpr int PLANE = BM * ROWB;
constexpr int STAGE = 2 * 3 * PLANE;      // A and B, three planes each
constexpr int LDS_BYTES = STAGE;          // single stage, two workgroups per CU

__global__ void split_planes(const float* __restrict__ x, uint16_t* __restrict__ p, int64_t count) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const float v = x[i];
  const uint32_t b = __float_as_uint(v);
  const float r1 = v - __uint_as_float(b & 0xFFFF0000u);
  const uint32_t b1 = __float_as_uint(r1);
  const float r2 = r1 - __uint_as_float(b1 & 0xFFFF0000u);
  p[i] = (uint16_t)(b >> 16);
  p[count + i] = (uint16_t)(b1 >> 16);
  p[2 * count + i] = (uint16_t)(__float_as_uint(r2) >> 16);
}

// planes: [3][rows][k] bf16.  C [n,m] = A B^T * scale.
__global__ void __launch_bounds__(256) gemm_planes(const uint16_t* __restrict__ A, const uint16_t* __restrict__ B,
                                                      float* __restrict__ C, int n, int m, int k, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tiles_n = m / BM;
  const int tr = blockIdx.x / tiles_n, tc = blockIdx.x % tiles_n;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  // per plane and operand: 128 rows x 32 bf16 = 128 x 64 B = 512 x 16 B -> two 16-byte pieces per thread
  const int lrow = tid >> 2, lpiece = tid & 3;                     // rows lrow and lrow + 64, 8 bf16 at 8 * lpiece
  const int64_t sa = (int64_t)n * k, sb = (int64_t)m * k;
  const uint16_t* ga = A + (int64_t)(tr * BM + lrow) * k + lpiece * 8;
  const uint16_t* gb = B + (int64_t)(tc * BM + lrow) * k + lpiece * 8;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  uint4 ra0, ra1, ra2, ra3, ra4, ra5, rb0, rb1, rb2, rb3, rb4, rb5;   // plane-major, two row halves each (named: no arrays)
#define GLOAD(kt)                                                                           \
  ra0 = *reinterpret_cast<const uint4*>(ga + (kt) * BK);                                    \
  ra1 = *reinterpret_cast<const uint4*>(ga + (int64_t)64 * k + (kt) * BK);                  \
  ra2 = *reinterpret_cast<const uint4*>(ga + sa + (kt) * BK);                               \
  ra3 = *reinterpret_cast<const uint4*>(ga + sa + (int64_t)64 * k + (kt) * BK);             \
  ra4 = *reinterpret_cast<const uint4*>(ga + 2 * sa + (kt) * BK);                           \
  ra5 = *reinterpret_cast<const uint4*>(ga + 2 * sa + (int64_t)64 * k + (kt) * BK);         \
  rb0 = *reinterpret_cast<const uint4*>(gb + (kt) * BK);                                    \
  rb1 = *reinterpret_cast<const uint4*>(gb + (int64_t)64 * k + (kt) * BK);                  \
  rb2 = *reinterpret_cast<const uint4*>(gb + sb + (kt) * BK);                               \
  rb3 = *reinterpret_cast<const uint4*>(gb + sb + (int64_t)64 * k + (kt) * BK);             \
  rb4 = *reinterpret_cast<const uint4*>(gb + 2 * sb + (kt) * BK);                           \
  rb5 = *reinterpret_cast<const uint4*>(gb + 2 * sb + (int64_t)64 * k + (kt) * BK);
  GLOAD(0)
  const int nk = k / BK;
  const int fr = lane & 31, fh = lane >> 5;
  for (int kt = 0; kt < nk; ++kt) {
    {
      char* w0 = smem + lrow * ROWB + lpiece * 16;
      char* w1 = w0 + 64 * ROWB;
      *reinterpret_cast<uint4*>(w0) = ra0;              *reinterpret_cast<uint4*>(w1) = ra1;
      *reinterpret_cast<uint4*>(w0 + PLANE) = ra2;      *reinterpret_cast<uint4*>(w1 + PLANE) = ra3;
      *reinterpret_cast<uint4*>(w0 + 2 * PLANE) = ra4;  *reinterpret_cast<uint4*>(w1 + 2 * PLANE) = ra5;
      *reinterpret_cast<uint4*>(w0 + 3 * PLANE) = rb0;  *reinterpret_cast<uint4*>(w1 + 3 * PLANE) = rb1;
      *reinterpret_cast<uint4*>(w0 + 4 * PLANE) = rb2;  *reinterpret_cast<uint4*>(w1 + 4 * PLANE) = rb3;
      *reinterpret_cast<uint4*>(w0 + 5 * PLANE) = rb4;  *reinterpret_cast<uint4*>(w1 + 5 * PLANE) = rb5;
    }
    __syncthreads();
    if (kt + 1 < nk) { GLOAD(kt + 1) }
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
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], c, 0, 0, 0);
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

// X [n,k] f32 device pointer, planes: 3*n*k uint16 device scratch, C [n,n].  Returns avg ms of the GEMM alone;
// *split_ms receives the one-off split time.
extern "C" double bf16x3_gemm_presplit(const float* X, uint16_t* planes, float* C, int n, int k, int reps, double* split_ms) {
  if (n % BM || k % BK) return -1.0;
  hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_planes), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  hipEvent_t e0, e1, e2;
  hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
  const int64_t count = (int64_t)n * k;
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(split_planes, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, 0, X, planes, count);
  hipEventRecord(e1, 0);
  const dim3 grid((unsigned)((n / BM) * (n / BM)));
  hipLaunchKernelGGL(gemm_planes, grid, dim3(256), LDS_BYTES, 0, planes, planes, C, n, n, k, 1.0f / (float)k);   // warm-up
  hipEventRecord(e1, 0);
  for (int r = 0; r < reps; ++r)
    hipLaunchKernelGGL(gemm_planes, grid, dim3(256), LDS_BYTES, 0, planes, planes, C, n, n, k, 1.0f / (float)k);
  hipEventRecord(e2, 0);
  if (hipEventSynchronize(e2) != hipSuccess) return -2.0;
  float ms = 0.f, sm = 0.f;
  hipEventElapsedTime(&ms, e1, e2);
  hipEventElapsedTime(&sm, e0, e1);
  if (split_ms) *split_ms = sm;
  hipEventDestroy(e0); hipEventDestroy(e1); hipEventDestroy(e2);
  return ms / reps;
}
