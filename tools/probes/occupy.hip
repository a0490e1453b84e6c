// Probe only (not part of the library): a kernel that holds `wgs` compute units busy for `usec` microseconds, standing in for
// a collective's kernel that shares the GPU with the backward.  96 KiB of LDS per workgroup keeps a GEMM workgroup off the CU.
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void occupy_kernel(long long ticks, unsigned* sink) {
  extern __shared__ unsigned pad[];
  const long long t0 = wall_clock64();
  unsigned v = threadIdx.x;
  while (wall_clock64() - t0 < ticks) v = v * 1664525u + 1013904223u;
  pad[threadIdx.x] = v;
  if (v == 0x12345678u) sink[0] = pad[(threadIdx.x + 1) & 255];
}

extern "C" int occupy(int wgs, int usec, int lds_kib, void* stream) {
  static unsigned* sink = nullptr;
  if (!sink && hipMalloc(&sink, 64) != hipSuccess) return 1;
  const size_t lds = (size_t)lds_kib * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(occupy_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(occupy_kernel, dim3(wgs), dim3(256), lds, (hipStream_t)stream, (long long)usec * 100, sink);  // 100 MHz clock
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
