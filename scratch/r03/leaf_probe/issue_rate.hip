// r03 probe: issue interval of one wave's VALU stream on gfx950 (one wave per SIMD, 256 threads per workgroup, one workgroup per CU)
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s\n", hipGetErrorString(e_)); return 1; } } while (0)
#define BODY16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)
#define S_(x) #x
#define S(x) S_(x)
#define FMA(i)   "v_fma_f32 %" S(i) ", %16, %17, %" S(i) "\n\t"
#define FMAC(i)  "v_fmac_f32 %" S(i) ", %16, %17\n\t"
#define FMACD(i) "v_fmac_f32_dpp %" S(i) ", %16, %17 row_newbcast:" S(i) " row_mask:0xf bank_mask:0xf\n\t"
#define MOVD(i)  "v_mov_b32_dpp %" S(i) ", %16 row_newbcast:" S(i) " row_mask:0xf bank_mask:0xf\n\t"
#define FMACS(i) "v_fmac_f32 %" S(i) ", s20, %17\n\t"
#define RDL(i)   "v_readlane_b32 s20, %16, " S(i) "\n\t"
#define RDLFMA(i) "v_readlane_b32 s20, %16, " S(i) "\n\tv_fmac_f32 %" S(i) ", s20, %17\n\t"
#define RDLFMA2(i) "v_readlane_b32 s2" S(i) ", %16, " S(i) "\n\t"
#define MUL(i)   "v_mul_f32 %" S(i) ", %16, %" S(i) "\n\t"
#define RSQ(i)   "v_rsq_f32 %" S(i) ", %" S(i) "\n\t"
#define FMA64(i) "v_fma_f64 %" S(i) ", %16, %17, %" S(i) "\n\t"
#define FMAC64D(i) "v_fmac_f64_dpp %" S(i) ", %16, %17 row_newbcast:" S(i) " row_mask:0xf bank_mask:0xf\n\t"
#define DEP(i)   "v_fmac_f32 %0, %16, %0\n\t"
#define DEPD(i)  "v_fmac_f32_dpp %0, %0, %17 row_newbcast:" S(i) " row_mask:0xf bank_mask:0xf\n\t"
#define OPS32 "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]), "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15])
template <int VAR, typename T>
__global__ void __launch_bounds__(256) k(T* out, long long* cyc, int iters) {
  __shared__ char pad[100 * 1024];
  if (threadIdx.x == 0) pad[blockIdx.x & 1023] = 1;
  T x[16], a = (T)0.999, b = (T)(1e-3 * threadIdx.x);
  for (int i = 0; i < 16; ++i) x[i] = (T)(threadIdx.x + i);
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if constexpr (VAR == 0) asm volatile(BODY16(FMA) : OPS32 : "v"(a), "v"(b));
    if constexpr (VAR == 1) asm volatile(BODY16(FMAC) : OPS32 : "v"(a), "v"(b));
    if constexpr (VAR == 2) asm volatile(BODY16(FMACD) : OPS32 : "v"(a), "v"(b));
    if constexpr (VAR == 3) asm volatile(BODY16(MOVD) : OPS32 : "v"(a), "v"(b));
    if constexpr (VAR == 4) asm volatile(BODY16(RDLFMA) : OPS32 : "v"(a), "v"(b) : "s20");
    if constexpr (VAR == 5) asm volatile(BODY16(MUL) : OPS32 : "v"(a), "v"(b));
    if constexpr (VAR == 6) asm volatile(BODY16(RSQ) : OPS32 : "v"(a), "v"(b));
    if constexpr (VAR == 7) asm volatile(BODY16(FMA64) : OPS32 : "v"(a), "v"(b));
    if constexpr (VAR == 8) asm volatile(BODY16(FMAC64D) : OPS32 : "v"(a), "v"(b));
    if constexpr (VAR == 9) asm volatile(BODY16(DEP) : OPS32 : "v"(a), "v"(b));
    if constexpr (VAR == 10) asm volatile(BODY16(DEPD) : OPS32 : "v"(a), "v"(b));
    if constexpr (VAR == 11) asm volatile(BODY16(RDL) : OPS32 : "v"(a), "v"(b) : "s20");
  }
  const long long t1 = __builtin_readcyclecounter();
  T s = 0;
  for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int VAR, typename T>
int run(const char* name, int wg_per_cu, int ops_per_body = 16) {
  T* out; long long* cyc;
  const int blocks = 256 * wg_per_cu, iters = 4000;
  CHK(hipMalloc(&out, sizeof(T) * blocks * 256)); CHK(hipMalloc(&cyc, 8 * blocks * 4));
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<VAR, T>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
  CHK(hipDeviceSynchronize());
  long long* h = new long long[blocks * 4];
  CHK(hipMemcpy(h, cyc, 8 * blocks * 4, hipMemcpyDeviceToHost));
  double s = 0; for (int i = 0; i < blocks * 4; ++i) s += (double)h[i];
  printf("%-44s %5.2f cycles per instruction (one wave's view)\n", name, s / (blocks * 4) / iters / ops_per_body);
  delete[] h; CHK(hipFree(out)); CHK(hipFree(cyc));
  return 0;
}
int main() {
  run<0, float>("v_fma_f32 indep", 1);
  run<1, float>("v_fmac_f32 indep", 1);
  run<2, float>("v_fmac_f32_dpp row_newbcast indep", 1);
  run<3, float>("v_mov_b32_dpp row_newbcast", 1);
  run<4, float>("v_readlane + v_fmac(sgpr) pair", 1, 32);
  run<11, float>("v_readlane", 1);
  run<5, float>("v_mul_f32", 1);
  run<6, float>("v_rsq_f32 (dependent on itself per reg)", 1);
  run<7, double>("v_fma_f64 indep", 1);
  run<8, double>("v_fmac_f64_dpp row_newbcast indep", 1);
  run<9, float>("v_fmac_f32 dependent chain", 1);
  run<10, float>("v_fmac_f32_dpp dependent chain (dpp src)", 1);
  return 0;
}
