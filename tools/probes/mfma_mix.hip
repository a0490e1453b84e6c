// Probe: the matrix-pipe time of one 32 x 32 x (32 k) block product in two operand forms, back to back from registers on random
// data (2 waves per SIMD, 8 accumulators per wave, nothing else running):
//   x3 : hi*hi + lo*hi + hi*lo on v_mfma_f32_32x32x16_bf16      = 6 MFMAs (48 passes)
//   h8 : fp16 hi*hi (2 x v_mfma_f32_32x32x16_f16) + ONE block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 over the K-concatenated e4m3
//        planes [a_lo | a_hi] . [w_hi | w_lo]                    = 3 MFMAs (32 passes)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_mix.hip -o tools/probes/mfma_mix && tools/probes/mfma_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int MODE>
__global__ __launch_bounds__(512) void k(const u32x4* __restrict__ src, const u32x4* __restrict__ src8, float* __restrict__ out, int iters) {
  const int tid = blockIdx.x * 512 + threadIdx.x;
  // per k16 step: 4 row fragments + 2 column fragments, hi and lo planes; two k16 steps = one 32-k slab row
  u32x4 ah[2][4], al[2][4], bh[2][2], bl[2][2];
  i32x8 a8[4], b8[2];
  for (int s = 0; s < 2; ++s) {
    for (int i = 0; i < 4; ++i) { ah[s][i] = src[(tid * 32 + s * 12 + i) & 0xffff]; al[s][i] = src[(tid * 32 + s * 12 + 4 + i) & 0xffff]; }
    for (int j = 0; j < 2; ++j) { bh[s][j] = src[(tid * 32 + s * 12 + 8 + j) & 0xffff]; bl[s][j] = src[(tid * 32 + s * 12 + 10 + j) & 0xffff]; }
  }
  for (int i = 0; i < 4; ++i) for (int q = 0; q < 8; ++q) a8[i][q] = src8[(tid * 16 + 2 * i + (q >> 2)) & 0xffff][q & 3];
  for (int j = 0; j < 2; ++j) for (int q = 0; q < 8; ++q) b8[j][q] = src8[(tid * 16 + 8 + 2 * j + (q >> 2)) & 0xffff][q & 3];
  f32x16 acc[4][2];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[s][i]), __builtin_bit_cast(bf16x8, bh[s][j]), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[s][i]), __builtin_bit_cast(bf16x8, bh[s][j]), acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[s][i]), __builtin_bit_cast(bf16x8, bl[s][j]), acc[i][j], 0, 0, 0);
          }
    } else {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[s][i]), __builtin_bit_cast(f16x8, bh[s][j]), acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i], b8[j], acc[i][j], 0, 0, 0, 0x73737373, 0, 0x7f7f7f7f);
    }
    ah[0][0][0] ^= 0x00010001u * (uint32_t)(it & 7);  // (static indices: dynamic ones would put the arrays into scratch)
    ah[1][2][3] ^= 0x00010001u * (uint32_t)(it & 5);
    a8[1][5] ^= 0x01010101 * (it & 3);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[tid] = s;
}

int main() {
  const int blocks = 256, iters = 2000;
  u32x4 *d, *d8; float* o;
  hipMalloc(&d, 65536 * 16); hipMalloc(&d8, 65536 * 16); hipMalloc(&o, blocks * 512 * 4);
  uint32_t* h = (uint32_t*)malloc(65536 * 16);
  for (int form = 0; form < 2; ++form) {  // 0: bf16 bit patterns, 1: fp16 bit patterns
    for (int i = 0; i < 65536 * 4; ++i) {
      uint32_t r = (uint32_t)rand() ^ ((uint32_t)rand() << 16);
      uint32_t lo, hi;
      if (form == 0) { lo = (r & 0x807f) | ((120 + ((r >> 8) & 7)) << 7); hi = ((r >> 16) & 0x807f) | ((120 + ((r >> 24) & 7)) << 7); }
      else { lo = (r & 0x83ff) | ((10 + ((r >> 10) & 7)) << 10); hi = ((r >> 16) & 0x83ff) | ((10 + ((r >> 26) & 7)) << 10); }
      h[i] = lo | (hi << 16);
    }
    hipMemcpy(d, h, 65536 * 16, hipMemcpyHostToDevice);
    for (int i = 0; i < 65536 * 4; ++i) { uint32_t r = (uint32_t)rand() ^ ((uint32_t)rand() << 16); h[i] = r & 0xf7f7f7f7u & 0xbfbfbfbfu; }
    hipMemcpy(d8, h, 65536 * 16, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      for (int l = 0; l < 20; ++l) { if (form) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(512), 0, 0, d, d8, o, iters); else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(512), 0, 0, d, d8, o, iters); }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double prods = 20.0 * blocks * 8 /*waves*/ * (double)iters * 8 /*blocks of 32x32x32 per iteration*/;
      if (rep == 2) printf("%s: %.1f ms for 20 launches -> %.2f ns per 32x32x32 block product and SIMD pair-of-waves slot, %.0f algorithmic TFLOP/s\n",
                           form ? "h8 (2 fp16 + 1 scaled fp8 MFMA)" : "x3 (6 bf16 MFMAs)          ", ms, ms * 1e6 / (20.0 * iters * 8 * 2), prods * 65536.0 / ms / 1e9);
    }
  }
  return 0;
}
