// graph_gap.hip — how long does a chain of dependent small kernels take on one stream, launched one by one, against the
// same chain captured once into a hipGraph and replayed?  (Is the ~5-10 us gap between the Cholesky's chain launches a
// host / packet cost a graph removes, or a device-side end-of-kernel cost it cannot?)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void step_kernel(float* p, int work) {
  float v = p[threadIdx.x + blockIdx.x * blockDim.x];
  for (int i = 0; i < work; ++i) v = v * 1.0001f + 0.5f;
  p[threadIdx.x + blockIdx.x * blockDim.x] = v;
}

int main() {
  float* d;
  CK(hipMalloc(&d, 1 << 24));
  CK(hipMemset(d, 0, 1 << 24));
  hipStream_t st;
  int lo, hi;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithPriority(&st, hipStreamNonBlocking, hi));
  const int chain = 256;
  for (int grid : {1, 128}) for (int work : {16, 4000}) {
    auto run_stream = [&]() { for (int i = 0; i < chain; ++i) hipLaunchKernelGGL(step_kernel, dim3(grid), dim3(256), 0, st, d, work); };
    run_stream(); CK(hipStreamSynchronize(st));
    // one kernel alone
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st)); hipLaunchKernelGGL(step_kernel, dim3(grid), dim3(256), 0, st, d, work); CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st)); float one_ms; CK(hipEventElapsedTime(&one_ms, e0, e1));
    double best_s = 1e9, best_g = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      auto t0 = std::chrono::steady_clock::now();
      run_stream(); CK(hipStreamSynchronize(st));
      best_s = std::min(best_s, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    run_stream();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    for (int rep = 0; rep < 5; ++rep) {
      auto t0 = std::chrono::steady_clock::now();
      CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
      best_g = std::min(best_g, std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count());
    }
    printf("grid %4d work %5d: one kernel (events) %.1f us; chain of %d: stream %.1f us/kernel, graph %.1f us/kernel\n", grid, work,
           one_ms * 1e3, chain, best_s / chain, best_g / chain);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
