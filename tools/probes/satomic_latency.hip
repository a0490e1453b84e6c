// Round-trip latency of a returning atomic from one wave, three ways: s_atomic_add ... glc (scalar), global_atomic_add_u32 ... sc0 (vector, one lane),
// and a plain s_load_dword ... glc of the same word; s_memrealtime ticks (100 MHz) over 2000 dependent operations each.
//   hipcc --offload-arch=gfx950 -O2 -o satomic_latency satomic_latency.hip && ./satomic_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void lat(unsigned* counter, unsigned long long* out) {
  const int N = 2000;
  unsigned long long t0, t1, t2, t3;
  unsigned v = 1;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0));
  for (int i = 0; i < N; ++i) {
    v = 1;
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(counter) : "memory");
  }
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1));
  unsigned acc = 0;
  for (int i = 0; i < N; ++i) {
    if (threadIdx.x == 0) acc += __hip_atomic_fetch_add(counter + 64, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2));
  unsigned w = 0, wacc = 0;
  for (int i = 0; i < N; ++i) {
    asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(w) : "s"(counter) : "memory");
    wacc += w;
  }
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t3));
  if (threadIdx.x == 0) {
    out[0] = t1 - t0; out[1] = t2 - t1; out[2] = t3 - t2; out[3] = acc + wacc + v;
  }
}
int main() {
  unsigned* c; unsigned long long *o, h[4];
  hipMalloc(&c, 1024); hipMalloc(&o, 32);
  hipMemset(c, 0, 1024);
  for (int r = 0; r < 2; ++r) {
    hipLaunchKernelGGL(lat, dim3(1), dim3(64), 0, 0, c, o);
    if (hipDeviceSynchronize() != hipSuccess) { printf("FAILED\n"); return 1; }
  }
  hipMemcpy(h, o, 32, hipMemcpyDeviceToHost);
  printf("per operation: s_atomic_add glc %.0f ns, global atomic (returning, one lane) %.0f ns, s_load_dword glc %.0f ns\n", h[0] * 10.0 / 2000, h[1] * 10.0 / 2000, h[2] * 10.0 / 2000);
  return 0;
}
