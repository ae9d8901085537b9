// xcc_probe.hip — which XCD does workgroup b of a launch land on, on an unmasked stream and on a stream whose CU mask
// leaves out the first R bits (the Cholesky's bulk stream: R = 32)?  The XCD tile map (gemm_nt.hpp) assumes b & 7.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void where_kernel(int* xcc, int* cu, int spin) {
  if (threadIdx.x == 0) {
    unsigned x, h;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(h));
    xcc[blockIdx.x] = (int)(x & 0xf);
    cu[blockIdx.x] = (int)h;
  }
  // keep the workgroup resident for a while so that the launch really spreads over the chip
  float v = (float)threadIdx.x;
  for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
  if (v == 12345.f) xcc[0] = -1;
}

int main() {
  const int nb = 2048;
  int *dx, *dc;
  CK(hipMalloc(&dx, nb * sizeof(int))); CK(hipMalloc(&dc, nb * sizeof(int)));
  std::vector<int> hx(nb), hc(nb);
  for (int reserve : {0, 8, 32, 40, 64}) {
    hipStream_t st;
    uint32_t mask[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = reserve; b < 256; ++b) mask[b >> 5] |= 1u << (b & 31);
    CK(hipExtStreamCreateWithCUMask(&st, 8, mask));
    hipLaunchKernelGGL(where_kernel, dim3(nb), dim3(256), 64 * 1024, st, dx, dc, 20000);
    CK(hipStreamSynchronize(st));
    CK(hipMemcpy(hx.data(), dx, nb * sizeof(int), hipMemcpyDeviceToHost));
    CK(hipMemcpy(hc.data(), dc, nb * sizeof(int), hipMemcpyDeviceToHost));
    int cnt[16] = {0}, match8 = 0, first[16];
    for (int i = 0; i < 16; ++i) first[i] = -1;
    for (int b = 0; b < nb; ++b) { cnt[hx[b]]++; if (hx[b] == hx[b & 7]) match8++; }
    printf("mask leaves out the first %2d bits: workgroups per XCD:", reserve);
    for (int x = 0; x < 8; ++x) printf(" %4d", cnt[x]);
    printf("   xcc(b) == xcc(b & 7) for %d of %d;  first 24 blocks:", match8, nb);
    for (int b = 0; b < 24; ++b) printf(" %d", hx[b]);
    printf("\n");
    CK(hipStreamDestroy(st));
  }
  return 0;
}
