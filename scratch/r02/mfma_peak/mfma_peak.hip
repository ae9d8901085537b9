// MFMA issue-rate microbench ("re-measure the f64 matrix peak with a v_mfma_f64_16x16x4_f64 microbench and
// state it").  Every wave runs a long loop of independent MFMAs (8 accumulators, no memory traffic); 1, 2 or 4 waves per
// SIMD; the whole chip.  Prints TFLOP/s for v_mfma_f64_16x16x4_f64 and, as a cross-check of the method against the guide's
// 157.3 TF, for v_mfma_f32_32x32x2_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(256) k_f64(double* out, int iters, double a0, double b0) {
  f64x4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = {0.0, 0.0, 0.0, 0.0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678) out[0] = s;   // keep the result alive
}
__global__ void __launch_bounds__(256) k_f32(float* out, int iters, float a0, float b0) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 12345.678f) out[0] = s;
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  double* d; hipMalloc(&d, 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("device: %s, %d CUs, clockRate %d kHz\n", p.gcnArchName, cus, p.clockRate);
  for (int wps = 1; wps <= 4; wps *= 2) {          // waves per SIMD = workgroups (256 threads = 4 waves) per CU
    const int grid = cus * wps, iters = 200000 / wps;
    for (int kind = 0; kind < 2; ++kind) {
      float best = 1e30f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL(k_f64, dim3(grid), dim3(256), 0, 0, d, iters, 1.0001, 0.9999);
        else hipLaunchKernelGGL(k_f32, dim3(grid), dim3(256), 0, 0, (float*)d, iters, 1.0001f, 0.9999f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;        // first repetition warms the clocks
      }
      const double per = kind == 0 ? 8.0 * 2 * 16 * 16 * 4 : 4.0 * 2 * 32 * 32 * 2;   // flops per wave per iteration
      const double fl = per * iters * 4.0 * grid;
      printf("%s  %d wave(s)/SIMD  grid %5d  %8.3f ms  %7.2f TFLOP/s\n", kind == 0 ? "v_mfma_f64_16x16x4_f64" : "v_mfma_f32_32x32x2_f32",
             wps, grid, best, fl / (best * 1e-3) / 1e12);
    }
  }
  return 0;
}
