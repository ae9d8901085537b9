// Does a consumer workgroup see a STALE line that its own XCD's L2 (and its CU's L1) cached from an EARLIER read, after
// another workgroup on another XCD has rewritten it and handed it off with the guide's release / acquire recipe?
// (The Cholesky as one persistent launch would do exactly this: a tile is read as a C block by one XCD, later rewritten by a
// panel task on another XCD, then read again as an operand.)
//   pair p: consumer = block 2p, producer = block 2p+1 (round-robin dispatch puts them on different XCDs; XCC id is recorded).
//   round r: consumer reads all of D_p (warms L1/L2 with version r), raises READY;
//            producer waits READY, writes version r+1 (plain stores), waits vmcnt(0), barrier, lane 0: release fence (agent),
//            vmcnt(0), relaxed agent store DONE = r+1;
//            consumer polls DONE (relaxed agent load), acquire fence (agent), vmcnt(0), barrier, reads D_p with PLAIN loads and
//            counts words that are not version r+1.
// MODE 1: the consumer's re-read uses sc1 loads (atomic relaxed agent 8-byte) instead of plain loads behind the acquire.
// Every spin is bounded; a timeout sets err and everybody leaves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define SPIN_LIMIT (1 << 22)

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

__global__ void __launch_bounds__(256) handoff(unsigned long long* data, int words, unsigned* ready, unsigned* done, int rounds,
                                               unsigned long long* stale, unsigned* err, unsigned* xcc, int mode) {
  const int pair = blockIdx.x >> 1, role = blockIdx.x & 1, tid = threadIdx.x;
  unsigned long long* d = data + (size_t)pair * words;
  if (tid == 0) xcc[blockIdx.x] = xcc_id();
  __shared__ int bail;
  if (tid == 0) bail = 0;
  __syncthreads();
  unsigned long long nstale = 0;
  for (int r = 0; r < rounds; ++r) {
    if (role == 0) {   // ---------------- consumer
      unsigned long long s = 0;
      for (int i = tid; i < words; i += 256) s += d[i];          // warm L1 / L2 with version r
      if (s == 0xdeadbeefULL) stale[0] = s;                      // keep the loads
      __syncthreads();
      if (tid == 0) {
        __hip_atomic_store(&ready[pair], (unsigned)(r + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int n = 0;
        while (__hip_atomic_load(&done[pair], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)(r + 1)) {
          __builtin_amdgcn_s_sleep(2);
          if (++n > SPIN_LIMIT || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { bail = 1; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
      if (bail) { if (tid == 0) atomicExch(err, 1u); return; }
      const unsigned long long want = (unsigned long long)(r + 1);
      for (int i = tid; i < words; i += 256) {
        unsigned long long v = mode == 1 ? __hip_atomic_load(&d[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : d[i];
        if (v != want) ++nstale;
      }
      __syncthreads();
    } else {           // ---------------- producer
      if (tid == 0) {
        int n = 0;
        while (__hip_atomic_load(&ready[pair], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)(r + 1)) {
          __builtin_amdgcn_s_sleep(2);
          if (++n > SPIN_LIMIT || __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { bail = 1; break; }
        }
      }
      __syncthreads();
      if (bail) { if (tid == 0) atomicExch(err, 1u); return; }
      for (int i = tid; i < words; i += 256) d[i] = (unsigned long long)(r + 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&done[pair], (unsigned)(r + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
  if (role == 0 && nstale) atomicAdd(stale + 1 + pair % 8, nstale);
}

int main(int argc, char** argv) {
  const int pairs = 128, rounds = 200;
  for (int mode = 0; mode < 2; ++mode)
    for (int kb : {4, 64, 512}) {
      const int words = kb * 1024 / 8;
      unsigned long long *data, *stale; unsigned *ready, *done, *err, *xcc;
      hipMalloc(&data, (size_t)pairs * words * 8); hipMemset(data, 0, (size_t)pairs * words * 8);
      hipMalloc(&stale, 16 * 8); hipMemset(stale, 0, 16 * 8);
      hipMalloc(&ready, pairs * 4); hipMemset(ready, 0, pairs * 4);
      hipMalloc(&done, pairs * 4); hipMemset(done, 0, pairs * 4);
      hipMalloc(&err, 4); hipMemset(err, 0, 4);
      hipMalloc(&xcc, 2 * pairs * 4);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0);
      hipLaunchKernelGGL(handoff, dim3(2 * pairs), dim3(256), 0, 0, data, words, ready, done, rounds, stale, err, xcc, mode);
      hipEventRecord(e1);
      hipError_t rc = hipEventSynchronize(e1);
      float ms = 0; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long hs[16]; unsigned herr, hx[2 * pairs];
      hipMemcpy(hs, stale, sizeof hs, hipMemcpyDeviceToHost); hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
      hipMemcpy(hx, xcc, sizeof hx, hipMemcpyDeviceToHost);
      unsigned long long tot = 0; for (int i = 1; i < 9; ++i) tot += hs[i];
      int cross = 0; for (int p = 0; p < pairs; ++p) cross += hx[2 * p] != hx[2 * p + 1];
      printf("mode %d (%s re-read)  %4d KB per pair  %d pairs (%d cross-XCD)  %d rounds: stale words %llu of %llu  timeout %u  rc %d  %.2f ms  (%.2f us per round)\n",
             mode, mode ? "sc1" : "plain+acquire", kb, pairs, cross, rounds, tot, (unsigned long long)pairs * rounds * words, herr, (int)rc, ms, ms * 1e3 / rounds);
      hipFree(data); hipFree(stale); hipFree(ready); hipFree(done); hipFree(err); hipFree(xcc);
    }
  return 0;
}
