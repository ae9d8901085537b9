// tile_probe.hip — where do the MFMA bubbles of the tile engine's K loop come from?  The library's own loop
// (gemm_nt.hpp, TileNT<float,128,128,2>::mainloop<1>) in a bare kernel: every workgroup runs R tiles of K columns from a
// 16384 x 16384 matrix, nothing else.  Built several times with one ingredient dropped each (-DSMN_PROBE_NO_*), and run
// at 2 and at 1 workgroup per CU.  Prints TFLOP/s against the 157.3 peak.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../../scale-mixtures-of-neural-network-gaussian-processes_amd/csrc/gemm_nt.hpp"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

using Tile = TileNT<float, 256, 128, 2>;

__global__ void __launch_bounds__(256, 1) probe_kernel(const float* __restrict__ a, int64_t lda, int K, int R, int same, float* out) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Tile t;
  t.zero();
  for (int r = 0; r < R; ++r) {
    const int idx = same ? (blockIdx.x & 7) : (blockIdx.x * R + r);
    const int64_t row0 = (int64_t)(idx % 60) * 256, col0 = (int64_t)((idx / 60) % 120) * 128;
    t.mainloop<1>(a + row0 * lda, lda, a + col0 * lda, lda, K, smem);
  }
  float s = 0.f;
  t.for_each([&](int, int, int i, int, int) { (void)i; });
#pragma unroll
  for (int m = 0; m < Tile::MT; ++m)
#pragma unroll
    for (int n = 0; n < Tile::NT; ++n)
#pragma unroll
      for (int i = 0; i < Tile::M::ACC; ++i) s += t.acc[m][n][i];
  if (s == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = s;   // keeps the accumulators alive
}

int main(int argc, char** argv) {
  const int64_t n = 16384;
  float* a; float* out;
  CK(hipMalloc(&a, sizeof(float) * n * n));
  CK(hipMalloc(&out, sizeof(float) * 1024 * 256));
  std::vector<float> h(1 << 20);
  for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
  for (int64_t off = 0; off < n * n; off += (1 << 20)) CK(hipMemcpy(a + off, h.data(), sizeof(float) << 20, hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int K = 1024, R = 12;
  struct Cfg { const char* name; int grid; size_t lds; int same; };
  const Cfg cfgs[] = {{"1 WG/CU (256x128 tile), streaming", 256, (size_t)Tile::LDS_BYTES, 0},
                      {"1 WG/CU (256x128 tile), 8 operand panels", 256, (size_t)Tile::LDS_BYTES, 1},
                      {"1 WG/CU (256x128) again, streaming", 256, (size_t)Tile::LDS_BYTES, 0},
                      {"1 WG/CU (256x128) again, 8 panels", 256, (size_t)Tile::LDS_BYTES, 1}};
  for (const Cfg& c : cfgs) {
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(probe_kernel, dim3(c.grid), dim3(256), c.lds, 0, a, n, K, R, c.same, out);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep >= 2 && ms < best) best = ms;
    }
    const double fl = 2.0 * 256 * 128 * (double)K * R * c.grid;
    printf("  %-44s %8.3f ms  %7.1f TFLOP/s  (%.1f %% of 157.3)\n", c.name, best, fl / best * 1e-9, fl / best * 1e-9 / 157.3 * 100);
  }
  return 0;
}
