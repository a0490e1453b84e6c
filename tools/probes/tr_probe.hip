// Probe: what does ds_read_b64_tr_b16 return?  LDS holds u16 value = its own element index.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(uint16_t* out, int mode) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  int l = threadIdx.x;
  uint32_t addr;
  if (mode == 0) addr = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)lds + l * 8;          // lane-linear 8 B
  else           addr = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)lds + ((l >> 2) * 256 + (l & 3) * 8) * 1;  // row (l>>2) of 128 elems (256 B), cols 4*(l&3)
  uint2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
  out[l * 4 + 0] = v.x & 0xffff; out[l * 4 + 1] = v.x >> 16; out[l * 4 + 2] = v.y & 0xffff; out[l * 4 + 3] = v.y >> 16;
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  uint16_t h[256];
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    printf("mode %d (lane: 4 returned element indices)\n", mode);
    for (int l = 0; l < 64; ++l) { printf("L%02d:%5d %5d %5d %5d   ", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]); if (l % 4 == 3) printf("\n"); }
  }
  return 0;
}
