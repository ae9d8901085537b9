// Which lane does each cross-lane primitive read from?  (semantics check for the 32-wide stencil path of cnn.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* o) {
  int v = threadIdx.x;
  o[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);        // wave_shr:1
  o[64 + threadIdx.x] = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false);   // wave_shl:1
  auto p = __builtin_amdgcn_permlane32_swap(v, v + 100, false, false);
  o[128 + threadIdx.x] = p[0];
  o[192 + threadIdx.x] = p[1];
}
int main() {
  int* d; hipMalloc(&d, 256 * 4);
  k<<<1, 64>>>(d);
  int h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* names[4] = {"wave_shr:1", "wave_shl:1", "permlane32_swap(v, v+100)[0]", "permlane32_swap(v, v+100)[1]"};
  for (int s = 0; s < 4; ++s) {
    printf("%-30s lanes 0,1,2,30,31,32,33,62,63:", names[s]);
    int ls[9] = {0, 1, 2, 30, 31, 32, 33, 62, 63};
    for (int l : ls) printf(" %d", h[64 * s + l]);
    printf("\n");
  }
  return 0;
}
