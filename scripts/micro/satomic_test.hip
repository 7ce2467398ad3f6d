// Does gfx950 execute scalar-memory atomics (s_atomic_add), and are they coherent across the waves of one XCD?  (block-count instrumentation, scripts/instr_blocks.py)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_scalar(unsigned* counters, int reps) {
  unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11)) & 7u;
  unsigned* base = counters + xcc * 64;
  unsigned one = 1;
  for (int i = 0; i < reps; ++i) {
    asm volatile("s_atomic_add %0, %1, 0x10" :: "s"(one), "s"(base) : "memory");
    asm volatile("s_atomic_add %0, %1, 0x14" :: "s"(one), "s"(base) : "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
__global__ void k_vector(unsigned* counters, int reps) {
  for (int i = 0; i < reps; ++i) {
    unsigned long long ex;
    asm volatile("s_mov_b64 %0, exec\n s_mov_b64 exec, 1" : "=s"(ex));
    unsigned one = 1, zero = 0;
    asm volatile("global_atomic_add %0, %1, %2 offset:32 sc1" :: "v"(zero), "v"(one), "s"(counters) : "memory");
    asm volatile("s_mov_b64 exec, %0" :: "s"(ex));
  }
}
int main() {
  unsigned* d; hipMalloc(&d, 4096); hipMemset(d, 0, 4096);
  const int blocks = 2048, threads = 256, reps = 1000;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a); k_scalar<<<blocks, threads>>>(d, reps); hipEventRecord(b);
  hipError_t e = hipDeviceSynchronize();
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  unsigned h[1024]; hipMemcpy(h, d, 4096, hipMemcpyDeviceToHost);
  unsigned long long s0 = 0, s1 = 0; for (int x = 0; x < 8; ++x) { s0 += h[x * 64 + 4]; s1 += h[x * 64 + 5]; }
  printf("scalar: err=%d sum0=%llu sum1=%llu expected=%llu  %.2f ms (%.1f M atomics/s)\n", (int)e, s0, s1, (unsigned long long)blocks * (threads / 64) * reps, ms, 2.0 * blocks * (threads / 64) * reps / ms / 1e3);
  for (int x = 0; x < 8; ++x) printf("  xcc %d: %u %u\n", x, h[x * 64 + 4], h[x * 64 + 5]);
  hipMemset(d, 0, 4096);
  hipEventRecord(a); k_vector<<<blocks, threads>>>(d, reps); hipEventRecord(b);
  e = hipDeviceSynchronize(); hipEventElapsedTime(&ms, a, b);
  hipMemcpy(h, d, 4096, hipMemcpyDeviceToHost);
  printf("vector sc1, one lane: err=%d sum=%u expected=%llu  %.2f ms (%.1f M atomics/s)\n", (int)e, h[8], (unsigned long long)blocks * (threads / 64) * reps, ms, 1.0 * blocks * (threads / 64) * reps / ms / 1e3);
  return 0;
}
