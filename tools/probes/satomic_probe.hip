// Does gfx950 execute the scalar atomics the assembler accepts (s_atomic_add ... glc: returned value in an SGPR, tracked by lgkmcnt)?
// A ticket drawn this way would not touch vmcnt, i.e. not disturb the counted waits of the LDS-DMA stream.
//   hipcc --offload-arch=gfx950 -O2 -o satomic_probe satomic_probe.hip && ./satomic_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(unsigned* counter, unsigned* out) {
  unsigned got = 1;  // the increment; replaced by the value before the add
  asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(got) : "s"(counter) : "memory");
  if (threadIdx.x == 0) out[blockIdx.x] = got;
}
int main() {
  unsigned *c, *o, h[64];
  hipMalloc(&c, 4); hipMalloc(&o, 64 * 4);
  hipMemset(c, 0, 4);
  hipLaunchKernelGGL(probe, dim3(64), dim3(64), 0, 0, c, o);
  if (hipDeviceSynchronize() != hipSuccess) { printf("FAILED: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
  unsigned total = 0;
  hipMemcpy(&total, c, 4, hipMemcpyDeviceToHost);
  hipMemcpy(h, o, 64 * 4, hipMemcpyDeviceToHost);
  unsigned long long seen = 0;
  for (int i = 0; i < 64; ++i) if (h[i] < 64) seen |= 1ull << h[i];
  printf("counter %u (want 64); distinct returned values %d (want 64); first few %u %u %u %u\n", total, __builtin_popcountll(seen), h[0], h[1], h[2], h[3]);
  return 0;
}
