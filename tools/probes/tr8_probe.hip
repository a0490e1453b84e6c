// Probe: what does ds_read_b64_tr_b8 return?  Two passes: LDS byte i holds (i & 255), then (i >> 8); combined = the byte's index.
// Lane l supplies the address l * 8 (lane-linear 8-byte chunks).  Output: for every lane, the 8 source byte indices it received.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(uint8_t* out, int pass, int mode) {
  __shared__ __attribute__((aligned(16))) uint8_t lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (uint8_t)(pass ? (i >> 8) : (i & 255));
  __syncthreads();
  int l = threadIdx.x;
  uint32_t base = (uint32_t)(size_t)(__attribute__((address_space(3))) void*)lds;
  uint32_t addr = mode == 0 ? base + l * 8 : base + (l >> 1) * 256 + (l & 1) * 8;  // mode 1: lanes (2j, 2j+1) = the two halves of a 16-byte segment of row j (rows 256 B apart)
  uint2 v;
  asm volatile("ds_read_b64_tr_b8 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
  for (int j = 0; j < 4; ++j) { out[l * 8 + j] = (v.x >> (8 * j)) & 255; out[l * 8 + 4 + j] = (v.y >> (8 * j)) & 255; }
}
int main() {
  uint8_t* d; hipMalloc(&d, 64 * 8);
  uint8_t lo[512], hi[512];
  for (int mode = 0; mode < 2; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 0, mode); hipMemcpy(lo, d, 512, hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 1, mode); hipMemcpy(hi, d, 512, hipMemcpyDeviceToHost);
    printf("mode %d (lane: 8 source byte indices)\n", mode);
    for (int l = 0; l < 64; ++l) {
      printf("L%02d:", l);
      for (int j = 0; j < 8; ++j) printf(" %4d", lo[l * 8 + j] | (hi[l * 8 + j] << 8));
      printf(l % 2 == 1 ? "\n" : "   ");
    }
  }
  return 0;
}
