// r03: what v_mfma_f64_16x16x4_f64 can sustain on this chip, and at which clock.
// 16 independent accumulator chains per wave (operands pinned in VGPRs, no memory traffic), 1 / 2 / 4 waves per SIMD, the whole
// chip; burst length swept from ~0.3 ms to ~80 ms with an idle gap in front of every burst, then back-to-back.  The shader
// clock during the burst is read in-kernel: delta s_memtime / delta s_memrealtime x 100 MHz (median over workgroups).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
#include <type_traits>
typedef double f64x4 __attribute__((ext_vector_type(4)));
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int CH, int VAR = 0>
__global__ void __launch_bounds__(256) k_f64(double* out, long long* clk, int iters, double a0, double b0) {
  f64x4 acc[CH];
  double av[8], bv[8], fill = a0;
#pragma unroll
  for (int i = 0; i < 8; ++i) { av[i] = a0 + i * 1e-3 + (threadIdx.x % 7) * 1e-3; bv[i] = b0 - i * 1e-3; asm volatile("" : "+v"(av[i]), "+v"(bv[i])); }
#pragma unroll
  for (int i = 0; i < CH; ++i) acc[i] = {0.0, 0.0, 0.0, 0.0};
  double a = a0 + (threadIdx.x % 7) * 1e-3, b = b0 - (threadIdx.x % 5) * 1e-3;
  asm volatile("" : "+v"(a), "+v"(b));
  const long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      if (VAR & 1) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i & 7], bv[(i + 3) & 7], acc[i], 0, 0, 0);
      else acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      if (VAR & 2) asm volatile("v_add_f64 %0, %0, %1" : "+v"(fill) : "v"(b));
    }
  }
  if (fill == 77.0) out[1] = fill;
  const long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  double s = 0;
#pragma unroll
  for (int i = 0; i < CH; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678) out[0] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
  hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  double* d; long long* clk;
  CHK(hipMalloc(&d, 64)); CHK(hipMalloc(&clk, sizeof(long long) * 2 * cus * 8));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  printf("device: %s, %d CUs; v_mfma_f64_16x16x4_f64 = 2048 flop; datasheet 78.6 TF = one per 64 cycles per SIMD at 2.4 GHz\n", p.gcnArchName, cus);
  auto run_t = [&](auto chc, int wps, int iters, bool gap, const char* tag, auto varc) -> int {
    constexpr int CH = decltype(chc)::value;
    constexpr int VAR = decltype(varc)::value;
    const int grid = cus * wps;
    if (gap) std::this_thread::sleep_for(std::chrono::milliseconds(60));
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_f64<CH, VAR>), dim3(grid), dim3(256), 0, 0, d, clk, iters, 1.0001, 0.9999);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> h(2 * grid);
    CHK(hipMemcpy(h.data(), clk, sizeof(long long) * 2 * grid, hipMemcpyDeviceToHost));
    std::vector<double> mhz(grid);
    for (int i = 0; i < grid; ++i) mhz[i] = h[2 * i + 1] > 0 ? 100.0 * (double)h[2 * i] / (double)h[2 * i + 1] : 0.0;
    std::sort(mhz.begin(), mhz.end());
    const double fl = (double)CH * 2048.0 * iters * 4.0 * grid;
    const double tf = fl / (ms * 1e-3) / 1e12;
    const double clkm = mhz[grid / 2];
    printf("%-14s %2d chains  %d wave(s)/SIMD  %7.3f ms  %6.2f TFLOP/s  shader clock %4.0f MHz  -> %5.1f cycles per MFMA per SIMD\n", tag, CH, wps, ms, tf,
           clkm, 2048.0 / (tf * 1e12 / (cus * 4.0) / (clkm * 1e6)));
    return 0;
  };
  using V0 = std::integral_constant<int, 0>;
  auto run = [&](int wps, int iters, bool gap, const char* tag) { return run_t(std::integral_constant<int, 8>{}, wps, iters, gap, tag, V0{}); };
  run(2, 2000, false, "warm-up");
  printf("chains per wave x waves per SIMD, ~4 ms bursts with an idle gap in front:\n");
  for (int wps : {1, 2, 4}) {
    run_t(std::integral_constant<int, 2>{}, wps, 64000 / wps, true, "burst", V0{});
    run_t(std::integral_constant<int, 4>{}, wps, 32000 / wps, true, "burst", V0{});
    run_t(std::integral_constant<int, 8>{}, wps, 16000 / wps, true, "burst", V0{});
    run_t(std::integral_constant<int, 12>{}, wps, 10000 / wps, true, "burst", V0{});
    run_t(std::integral_constant<int, 16>{}, wps, 8000 / wps, true, "burst", V0{});
  }
  printf("operand / filler variants, 8 chains, ~6 ms bursts: same operand registers | 8 operand pairs | + a v_add_f64 per MFMA | both\n");
  for (int wps : {2, 4}) {
    run_t(std::integral_constant<int, 8>{}, wps, 16000 / wps, true, "same-operands", std::integral_constant<int, 0>{});
    run_t(std::integral_constant<int, 8>{}, wps, 16000 / wps, true, "8-operand-pairs", std::integral_constant<int, 1>{});
    run_t(std::integral_constant<int, 8>{}, wps, 16000 / wps, true, "+v_add_f64", std::integral_constant<int, 2>{});
    run_t(std::integral_constant<int, 8>{}, wps, 16000 / wps, true, "pairs+v_add", std::integral_constant<int, 3>{});
  }
  printf("burst length, 8 chains, 2 waves per SIMD:\n");
  for (int iters : {500, 2000, 8000, 32000, 128000}) if (run(2, iters / 2, true, "burst")) return 1;
  printf("back to back (no gap), 8 chains, 2 waves/SIMD:\n");
  for (int rep = 0; rep < 6; ++rep) if (run(2, 16000, false, "sustained")) return 1;
  return 0;
}
