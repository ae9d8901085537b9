// r04: the SAME hand-written instruction stream with the accumulators (dst / srcC) in ARCHITECTURAL VGPRs and in AGPRs.
// 8 independent chains per wave, operands pinned, no memory traffic; f64 16x16x4 (2048 flop) and f32 16x16x4 (2048 flop).
// (The compiler's own choices for __builtin_amdgcn_mfma_* in a bare loop are not a clean test: without a min-blocks launch
// bound it selects the AGPR form AND copies all accumulators VGPR <-> AGPR every iteration.)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
#define CLOB_V "v128","v129","v130","v131","v132","v133","v134","v135","v136","v137","v138","v139","v140","v141","v142","v143","v144","v145","v146","v147","v148","v149","v150","v151","v152","v153","v154","v155","v156","v157","v158","v159","v160","v161","v162","v163","v164","v165","v166","v167","v168","v169","v170","v171","v172","v173","v174","v175","v176","v177","v178","v179","v180","v181","v182","v183","v184","v185","v186","v187","v188","v189","v190","v191"
#define CLOB_A "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63"

template <int VAR>   // 0: f64 VGPR, 1: f64 AGPR, 2: f32 VGPR, 3: f32 AGPR
__global__ void __launch_bounds__(256) k(double* out, long long* clk, int iters, double a0, double b0) {
  double a = a0 + (threadIdx.x % 7) * 1e-3, b = b0 - (threadIdx.x % 5) * 1e-3;
  float af = (float)a, bf = (float)b;
  asm volatile("" : "+v"(a), "+v"(b), "+v"(af), "+v"(bf));
  // zero the accumulators
  if (VAR == 0 || VAR == 2) {
    for (int i = 0; i < 1; ++i)
      asm volatile(
          "v_mov_b32 v128, 0\n v_mov_b32 v129, 0\n v_mov_b32 v130, 0\n v_mov_b32 v131, 0\n v_mov_b32 v132, 0\n v_mov_b32 v133, 0\n v_mov_b32 v134, 0\n v_mov_b32 v135, 0\n"
          "v_mov_b32 v136, 0\n v_mov_b32 v137, 0\n v_mov_b32 v138, 0\n v_mov_b32 v139, 0\n v_mov_b32 v140, 0\n v_mov_b32 v141, 0\n v_mov_b32 v142, 0\n v_mov_b32 v143, 0\n"
          "v_mov_b32 v144, 0\n v_mov_b32 v145, 0\n v_mov_b32 v146, 0\n v_mov_b32 v147, 0\n v_mov_b32 v148, 0\n v_mov_b32 v149, 0\n v_mov_b32 v150, 0\n v_mov_b32 v151, 0\n"
          "v_mov_b32 v152, 0\n v_mov_b32 v153, 0\n v_mov_b32 v154, 0\n v_mov_b32 v155, 0\n v_mov_b32 v156, 0\n v_mov_b32 v157, 0\n v_mov_b32 v158, 0\n v_mov_b32 v159, 0\n"
          "v_mov_b32 v160, 0\n v_mov_b32 v161, 0\n v_mov_b32 v162, 0\n v_mov_b32 v163, 0\n v_mov_b32 v164, 0\n v_mov_b32 v165, 0\n v_mov_b32 v166, 0\n v_mov_b32 v167, 0\n"
          "v_mov_b32 v168, 0\n v_mov_b32 v169, 0\n v_mov_b32 v170, 0\n v_mov_b32 v171, 0\n v_mov_b32 v172, 0\n v_mov_b32 v173, 0\n v_mov_b32 v174, 0\n v_mov_b32 v175, 0\n"
          "v_mov_b32 v176, 0\n v_mov_b32 v177, 0\n v_mov_b32 v178, 0\n v_mov_b32 v179, 0\n v_mov_b32 v180, 0\n v_mov_b32 v181, 0\n v_mov_b32 v182, 0\n v_mov_b32 v183, 0\n"
          "v_mov_b32 v184, 0\n v_mov_b32 v185, 0\n v_mov_b32 v186, 0\n v_mov_b32 v187, 0\n v_mov_b32 v188, 0\n v_mov_b32 v189, 0\n v_mov_b32 v190, 0\n v_mov_b32 v191, 0\n s_nop 7\n" ::: CLOB_V);
  } else {
    asm volatile(
        "v_accvgpr_write_b32 a0, 0\n v_accvgpr_write_b32 a1, 0\n v_accvgpr_write_b32 a2, 0\n v_accvgpr_write_b32 a3, 0\n v_accvgpr_write_b32 a4, 0\n v_accvgpr_write_b32 a5, 0\n v_accvgpr_write_b32 a6, 0\n v_accvgpr_write_b32 a7, 0\n"
        "v_accvgpr_write_b32 a8, 0\n v_accvgpr_write_b32 a9, 0\n v_accvgpr_write_b32 a10, 0\n v_accvgpr_write_b32 a11, 0\n v_accvgpr_write_b32 a12, 0\n v_accvgpr_write_b32 a13, 0\n v_accvgpr_write_b32 a14, 0\n v_accvgpr_write_b32 a15, 0\n"
        "v_accvgpr_write_b32 a16, 0\n v_accvgpr_write_b32 a17, 0\n v_accvgpr_write_b32 a18, 0\n v_accvgpr_write_b32 a19, 0\n v_accvgpr_write_b32 a20, 0\n v_accvgpr_write_b32 a21, 0\n v_accvgpr_write_b32 a22, 0\n v_accvgpr_write_b32 a23, 0\n"
        "v_accvgpr_write_b32 a24, 0\n v_accvgpr_write_b32 a25, 0\n v_accvgpr_write_b32 a26, 0\n v_accvgpr_write_b32 a27, 0\n v_accvgpr_write_b32 a28, 0\n v_accvgpr_write_b32 a29, 0\n v_accvgpr_write_b32 a30, 0\n v_accvgpr_write_b32 a31, 0\n"
        "v_accvgpr_write_b32 a32, 0\n v_accvgpr_write_b32 a33, 0\n v_accvgpr_write_b32 a34, 0\n v_accvgpr_write_b32 a35, 0\n v_accvgpr_write_b32 a36, 0\n v_accvgpr_write_b32 a37, 0\n v_accvgpr_write_b32 a38, 0\n v_accvgpr_write_b32 a39, 0\n"
        "v_accvgpr_write_b32 a40, 0\n v_accvgpr_write_b32 a41, 0\n v_accvgpr_write_b32 a42, 0\n v_accvgpr_write_b32 a43, 0\n v_accvgpr_write_b32 a44, 0\n v_accvgpr_write_b32 a45, 0\n v_accvgpr_write_b32 a46, 0\n v_accvgpr_write_b32 a47, 0\n"
        "v_accvgpr_write_b32 a48, 0\n v_accvgpr_write_b32 a49, 0\n v_accvgpr_write_b32 a50, 0\n v_accvgpr_write_b32 a51, 0\n v_accvgpr_write_b32 a52, 0\n v_accvgpr_write_b32 a53, 0\n v_accvgpr_write_b32 a54, 0\n v_accvgpr_write_b32 a55, 0\n"
        "v_accvgpr_write_b32 a56, 0\n v_accvgpr_write_b32 a57, 0\n v_accvgpr_write_b32 a58, 0\n v_accvgpr_write_b32 a59, 0\n v_accvgpr_write_b32 a60, 0\n v_accvgpr_write_b32 a61, 0\n v_accvgpr_write_b32 a62, 0\n v_accvgpr_write_b32 a63, 0\n s_nop 7\n" ::: CLOB_A);
  }
  const long long c0 = __builtin_readcyclecounter(), r0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
    if (VAR == 0)
      asm volatile(
          "v_mfma_f64_16x16x4_f64 v[128:135], %0, %1, v[128:135]\n v_mfma_f64_16x16x4_f64 v[136:143], %0, %1, v[136:143]\n"
          "v_mfma_f64_16x16x4_f64 v[144:151], %0, %1, v[144:151]\n v_mfma_f64_16x16x4_f64 v[152:159], %0, %1, v[152:159]\n"
          "v_mfma_f64_16x16x4_f64 v[160:167], %0, %1, v[160:167]\n v_mfma_f64_16x16x4_f64 v[168:175], %0, %1, v[168:175]\n"
          "v_mfma_f64_16x16x4_f64 v[176:183], %0, %1, v[176:183]\n v_mfma_f64_16x16x4_f64 v[184:191], %0, %1, v[184:191]\n" :: "v"(a), "v"(b) : CLOB_V);
    else if (VAR == 1)
      asm volatile(
          "v_mfma_f64_16x16x4_f64 a[0:7], %0, %1, a[0:7]\n v_mfma_f64_16x16x4_f64 a[8:15], %0, %1, a[8:15]\n"
          "v_mfma_f64_16x16x4_f64 a[16:23], %0, %1, a[16:23]\n v_mfma_f64_16x16x4_f64 a[24:31], %0, %1, a[24:31]\n"
          "v_mfma_f64_16x16x4_f64 a[32:39], %0, %1, a[32:39]\n v_mfma_f64_16x16x4_f64 a[40:47], %0, %1, a[40:47]\n"
          "v_mfma_f64_16x16x4_f64 a[48:55], %0, %1, a[48:55]\n v_mfma_f64_16x16x4_f64 a[56:63], %0, %1, a[56:63]\n" :: "v"(a), "v"(b) : CLOB_A);
    else if (VAR == 2)
      asm volatile(
          "v_mfma_f32_16x16x4_f32 v[128:131], %0, %1, v[128:131]\n v_mfma_f32_16x16x4_f32 v[132:135], %0, %1, v[132:135]\n"
          "v_mfma_f32_16x16x4_f32 v[136:139], %0, %1, v[136:139]\n v_mfma_f32_16x16x4_f32 v[140:143], %0, %1, v[140:143]\n"
          "v_mfma_f32_16x16x4_f32 v[144:147], %0, %1, v[144:147]\n v_mfma_f32_16x16x4_f32 v[148:151], %0, %1, v[148:151]\n"
          "v_mfma_f32_16x16x4_f32 v[152:155], %0, %1, v[152:155]\n v_mfma_f32_16x16x4_f32 v[156:159], %0, %1, v[156:159]\n" :: "v"(af), "v"(bf) : CLOB_V);
    else
      asm volatile(
          "v_mfma_f32_16x16x4_f32 a[0:3], %0, %1, a[0:3]\n v_mfma_f32_16x16x4_f32 a[4:7], %0, %1, a[4:7]\n"
          "v_mfma_f32_16x16x4_f32 a[8:11], %0, %1, a[8:11]\n v_mfma_f32_16x16x4_f32 a[12:15], %0, %1, a[12:15]\n"
          "v_mfma_f32_16x16x4_f32 a[16:19], %0, %1, a[16:19]\n v_mfma_f32_16x16x4_f32 a[20:23], %0, %1, a[20:23]\n"
          "v_mfma_f32_16x16x4_f32 a[24:27], %0, %1, a[24:27]\n v_mfma_f32_16x16x4_f32 a[28:31], %0, %1, a[28:31]\n" :: "v"(af), "v"(bf) : CLOB_A);
  }
  asm volatile("s_nop 15\n s_nop 15" ::: "memory");
  const long long c1 = __builtin_readcyclecounter(), r1 = wall_clock64();
  if (a == 12345.678) out[0] = a;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

int main() {
  hipDeviceProp_t p; CHK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  double* d; long long* clk;
  CHK(hipMalloc(&d, 64)); CHK(hipMalloc(&clk, sizeof(long long) * 2 * cus * 8));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  printf("device: %s, %d CUs; datasheet: f64 78.6 TF (64 cycles per v_mfma_f64_16x16x4_f64 per SIMD at 2.4 GHz), f32 157.3 TF (32 cycles per v_mfma_f32_16x16x4_f32)\n", p.gcnArchName, cus);
  auto run = [&](auto var, int wps, int iters, const char* tag) -> int {
    constexpr int VAR = decltype(var)::value;
    const int grid = cus * wps;
    for (int rep = 0; rep < 3; ++rep) {
      CHK(hipEventRecord(e0));
      hipLaunchKernelGGL((k<VAR>), dim3(grid), dim3(256), 0, 0, d, clk, iters, 1.0001, 0.9999);
      CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
      float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
      if (rep < 2) continue;
      std::vector<long long> h(2 * grid);
      CHK(hipMemcpy(h.data(), clk, sizeof(long long) * 2 * grid, hipMemcpyDeviceToHost));
      std::vector<double> mhz(grid);
      for (int i = 0; i < grid; ++i) mhz[i] = h[2 * i + 1] > 0 ? 100.0 * (double)h[2 * i] / (double)h[2 * i + 1] : 0.0;
      std::sort(mhz.begin(), mhz.end());
      const double fl = 8.0 * 2048.0 * iters * 4.0 * grid;
      const double tf = fl / (ms * 1e-3) / 1e12, clkm = mhz[grid / 2];
      printf("%-40s %d wave(s)/SIMD  %7.3f ms  %6.2f TFLOP/s  clock %4.0f MHz  -> %5.1f cycles per MFMA per SIMD\n", tag, wps, ms, tf, clkm,
             2048.0 / (tf * 1e12 / (cus * 4.0) / (clkm * 1e6)));
    }
    return 0;
  };
  for (int wps : {1, 2, 4}) {
    if (run(std::integral_constant<int, 0>{}, wps, 32000 / wps, "f64 16x16x4, accumulators in VGPRs")) return 1;
    if (run(std::integral_constant<int, 1>{}, wps, 32000 / wps, "f64 16x16x4, accumulators in AGPRs")) return 1;
    if (run(std::integral_constant<int, 2>{}, wps, 64000 / wps, "f32 16x16x4, accumulators in VGPRs")) return 1;
    if (run(std::integral_constant<int, 3>{}, wps, 64000 / wps, "f32 16x16x4, accumulators in AGPRs")) return 1;
  }
  return 0;
}
