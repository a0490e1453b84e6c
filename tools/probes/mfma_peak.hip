// Probe: what does the matrix pipe sustain on THIS chip when nothing else runs -- v_mfma_f32_32x32x16_bf16 back to back from registers,
// 2 waves per SIMD (512 threads, one workgroup per CU), 8 independent accumulators per wave -- with all-zero operands and with
// full-entropy random bf16 operands?  The difference is the power management's doing (MI355X_MICROARCH.md "DVFS give-back"), and the
// random-data figure is the ceiling any bf16 GEMM main loop can reach.   Also the block-scaled fp8 K = 64 form.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_peak.hip -o tools/probes/mfma_peak && tools/probes/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int FP8>
__global__ __launch_bounds__(512) void k(const u32x4* __restrict__ src, float* __restrict__ out, int iters) {
  const int tid = blockIdx.x * 512 + threadIdx.x;
  u32x4 a[4], b[2];
  for (int i = 0; i < 4; ++i) a[i] = src[(tid * 8 + i) & 0xffff];
  for (int j = 0; j < 2; ++j) b[j] = src[(tid * 8 + 4 + j) & 0xffff];
  f32x16 acc[4][2];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (FP8) {
          i32x8 av, bv;
          for (int q = 0; q < 4; ++q) { av[q] = a[i][q]; av[4 + q] = a[(i + 1) & 3][q]; bv[q] = b[j][q]; bv[4 + q] = b[j ^ 1][q]; }
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        } else {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
        }
      }
    // rotate the operands a little so that consecutive MFMAs do not see identical inputs (cheap VALU, hidden under the MFMAs)
    a[it & 3][it & 3] ^= 0x00010001u * (uint32_t)(it & 7);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[tid] = s;
}

typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ __launch_bounds__(512) void k16(const u32x4* __restrict__ src, float* __restrict__ out, int iters) {
  const int tid = blockIdx.x * 512 + threadIdx.x;
  u32x4 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = src[(tid * 8 + i) & 0xffff]; b[i] = src[(tid * 8 + 4 + i) & 0xffff]; }
  f32x4 acc[4][4];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a[i]), __builtin_bit_cast(bf16x8, b[j]), acc[i][j], 0, 0, 0);
    a[it & 3][it & 3] ^= 0x00010001u * (uint32_t)(it & 7);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) s += acc[i][j][r];
  out[tid] = s;
}

int main() {
  const int blocks = 256, iters = 4000;
  u32x4* d; float* o;
  hipMalloc(&d, 65536 * 16); hipMalloc(&o, blocks * 512 * 4);
  uint32_t* h = (uint32_t*)malloc(65536 * 16);
  for (int mode = 0; mode < 2; ++mode) {
    for (int i = 0; i < 65536 * 4; ++i) {
      if (mode == 0) h[i] = 0;
      else {  // random bf16 pairs in [-2, 2): random sign, exponent 120..127, random mantissa
        uint32_t r = (uint32_t)rand() ^ ((uint32_t)rand() << 16);
        uint32_t lo = (r & 0x807f) | ((120 + ((r >> 8) & 7)) << 7), hi = ((r >> 16) & 0x807f) | ((120 + ((r >> 24) & 7)) << 7);
        h[i] = lo | (hi << 16);
      }
    }
    hipMemcpy(d, h, 65536 * 16, hipMemcpyHostToDevice);
    {  // 16x16x32 bf16: 16 independent accumulators, same FLOPs per iteration (16 x 16384 = 8 x 32768)
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int l = 0; l < 20; ++l) hipLaunchKernelGGL(k16, dim3(blocks), dim3(512), 0, 0, d, o, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = 20.0 * blocks * 8 * (double)iters * 16 * 16384.0;
        if (rep == 2) printf("%s operands, bf16 16x16x32: %.1f ms for 20 launches -> %.0f TFLOP/s\n", mode ? "random" : "zero  ", ms, flops / ms / 1e9);
      }
    }
    for (int fp8 = 0; fp8 < 2; ++fp8) {
      if (fp8 && mode == 1) for (int i = 0; i < 65536 * 4; ++i) { uint32_t r = (uint32_t)rand() ^ ((uint32_t)rand() << 16); h[i] = r & 0xf7f7f7f7u & 0xbfbfbfbfu; }  // e4m3, |v| < 2^1, no NaN
      if (fp8 && mode == 1) hipMemcpy(d, h, 65536 * 16, hipMemcpyHostToDevice);
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int l = 0; l < 20; ++l) { if (fp8) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, d, o, iters); else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, d, o, iters); }
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flops = 20.0 * blocks * 8 /*waves*/ * (double)iters * 8 /*mfma*/ * (fp8 ? 131072.0 : 32768.0);
        if (rep == 2) printf("%s operands, %s: %.1f ms for 20 launches -> %.0f TFLOP/s\n", mode ? "random" : "zero  ", fp8 ? "fp8 32x32x64 (scaled)" : "bf16 32x32x16", ms, flops / ms / 1e9);
      }
    }
  }
  return 0;
}
