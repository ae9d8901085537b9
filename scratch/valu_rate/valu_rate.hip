// v_fma_f64 / v_fma_f32 issue rate on gfx950: N independent chains per lane, 4 or 8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T, int CH>
__global__ void __launch_bounds__(256) fma_kernel(T* out, T a, T b, int iters) {
  T x[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) x[i] = (T)threadIdx.x + (T)i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < CH; ++i) x[i] = x[i] * a + b;
  }
  T s = 0;
#pragma unroll
  for (int i = 0; i < CH; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename T, int CH>
void run(const char* name, int wg_per_cu) {
  const int blocks = 256 * wg_per_cu, iters = 4096;
  T* out; hipMalloc(&out, sizeof(T) * blocks * 256);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  fma_kernel<T, CH><<<blocks, 256>>>(out, (T)0.999, (T)0.001, iters);
  hipEventRecord(e0);
  fma_kernel<T, CH><<<blocks, 256>>>(out, (T)0.999, (T)0.001, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double fmas = (double)blocks * 256 * iters * CH;
  // wave-instructions per SIMD: each WG = 4 waves, one per SIMD; per SIMD: wg_per_cu waves x iters x CH instr
  const double cyc_per_instr = (ms * 1e-3 * 2.4e9) / ((double)wg_per_cu * iters * CH);
  printf("%s chains=%d wg/CU=%d: %.3f ms, %.2f TFLOP/s, %.2f cycles(at 2.4 GHz) per wave-instruction per SIMD\n", name, CH, wg_per_cu,
         ms, 2 * fmas / (ms * 1e-3) / 1e12, cyc_per_instr);
  hipFree(out);
}
int main() {
  run<double, 8>("f64", 1); run<double, 8>("f64", 2); run<double, 8>("f64", 4);
  run<float, 8>("f32", 1); run<float, 8>("f32", 2); run<float, 8>("f32", 4);
  return 0;
}
