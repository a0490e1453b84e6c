// Probe: v_mfma_scale_f32_32x32x64_f8f6f4 with fp6 (e2m3) operands on gfx950 -- (1) operand semantics: element i of a lane's 32 k-slots
// sits at bits [6 i, 6 i + 6) of the lane's first 6 operand registers, lanes 0..31 feed k 0..31 of row (lane % 32), lanes 32..63 k 32..63,
// the block scale is one E8M0 byte per lane taken from byte `opsel` of the scale register -- checked against a host evaluation of the
// decoded values; (2) matrix-pipe time of the three block-product forms of the sampling tiers from registers (2 waves per SIMD,
// 8 accumulators per wave):  h8: 2 fp16 MFMAs + one e4m3 K=64 MFMA per 32 k;  h6: 2 fp16 MFMAs + one fp6 K=64 MFMA per 32 k.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f6.hip -o tools/probes/mfma_f6 && tools/probes/mfma_f6
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

__global__ void one(const uint32_t* a, const uint32_t* b, float* out) {  // a, b: [64 lanes][8 dwords]: 6 data dwords, dword 6 = scale byte
  const int lane = threadIdx.x;
  i32x8 av, bv;
  for (int q = 0; q < 8; ++q) {
    av[q] = (int)a[lane * 8 + q];
    bv[q] = (int)b[lane * 8 + q];
  }
  f32x16 acc;
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, acc, 2, 2, 0, av[6], 0, bv[6]);
  for (int r = 0; r < 16; ++r) out[lane * 16 + r] = acc[r];
}
static float dec_e2m3(int c) {
  const int s = (c >> 5) & 1, e = (c >> 3) & 3, m = c & 7;
  const float v = e == 0 ? m * 0.125f : (1.0f + m * 0.125f) * (float)(1 << (e - 1));
  return s ? -v : v;
}

template <int MODE>
__global__ __launch_bounds__(512) void k(const u32x4* __restrict__ src, float* __restrict__ out, int iters) {
  const int tid = blockIdx.x * 512 + threadIdx.x;
  u32x4 ah[2][4], bh[2][2];
  i32x8 a8[4], b8[2];
  for (int s = 0; s < 2; ++s) {
    for (int i = 0; i < 4; ++i) ah[s][i] = src[(tid * 32 + s * 12 + i) & 0xffff];
    for (int j = 0; j < 2; ++j) bh[s][j] = src[(tid * 32 + s * 12 + 8 + j) & 0xffff];
  }
  for (int i = 0; i < 4; ++i) for (int q = 0; q < 8; ++q) a8[i][q] = src[(tid * 16 + 2 * i + (q >> 2)) & 0xffff][q & 3] & 0x1b1b1b1b;
  for (int j = 0; j < 2; ++j) for (int q = 0; q < 8; ++q) b8[j][q] = src[(tid * 16 + 8 + 2 * j + (q >> 2)) & 0xffff][q & 3] & 0x1b1b1b1b;
  f32x16 acc[4][2];
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
    if (MODE != 2) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[s][i]), __builtin_bit_cast(f16x8, bh[s][j]), acc[i][j], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (MODE == 0) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i], b8[j], acc[i][j], 0, 0, 0, 0x73737373, 0, 0x7f7f7f7f);
        else acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i], b8[j], acc[i][j], 2, 2, 0, a8[i][6], 0, b8[j][6]);
      }
    ah[0][0][0] ^= 0x00010001u * (uint32_t)(it & 7);
    a8[1][5] ^= 0x01010101 * (it & 3);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[tid] = s;
}

int main() {
  // ---- (1) semantics
  uint32_t ha[64 * 8], hb[64 * 8];
  int ca[32][64], cb[32][64], sa[64], sb[64];
  srand(7);
  for (int l = 0; l < 64; ++l) {
    sa[l] = 120 + rand() % 12;
    sb[l] = 122 + rand() % 10;
    uint64_t bits_a[3] = {0, 0, 0}, bits_b[3] = {0, 0, 0};
    for (int i = 0; i < 32; ++i) {
      const int x = rand() & 63, y = rand() & 63;
      ca[l % 32][32 * (l / 32) + i] = x;
      cb[l % 32][32 * (l / 32) + i] = y;
      for (int bit = 0; bit < 6; ++bit) {
        const int pos = 6 * i + bit;
        if ((x >> bit) & 1) bits_a[pos / 64] |= 1ull << (pos % 64);
        if ((y >> bit) & 1) bits_b[pos / 64] |= 1ull << (pos % 64);
      }
    }
    for (int q = 0; q < 6; ++q) {
      ha[l * 8 + q] = (uint32_t)(bits_a[q / 2] >> (32 * (q % 2)));
      hb[l * 8 + q] = (uint32_t)(bits_b[q / 2] >> (32 * (q % 2)));
    }
    ha[l * 8 + 6] = (uint32_t)sa[l] | 0xabcdef00u;  // (upper bytes: junk -- only byte 0 is the scale)
    hb[l * 8 + 6] = (uint32_t)sb[l] | 0x12345600u;
    ha[l * 8 + 7] = 0xdeadbeefu;
    hb[l * 8 + 7] = 0xfeedfaceu;
  }
  uint32_t *da, *db;
  float* dout;
  hipMalloc(&da, sizeof ha);
  hipMalloc(&db, sizeof hb);
  hipMalloc(&dout, 64 * 16 * 4);
  hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice);
  hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
  one<<<1, 64>>>(da, db, dout);
  float ho[64 * 16];
  hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
  double worst = 0.0, big = 0.0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 16; ++r) {
      const int col = l & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);  // C[row][col] = sum_k A[row][k] B[col][k]
      double ref = 0.0;
      for (int kb = 0; kb < 2; ++kb) {
        double part = 0.0;
        for (int i = 0; i < 32; ++i) part += (double)dec_e2m3(ca[row][32 * kb + i]) * (double)dec_e2m3(cb[col][32 * kb + i]);
        ref += part * ldexp(1.0, sa[row + 32 * kb] - 127) * ldexp(1.0, sb[col + 32 * kb] - 127);
      }
      worst = fmax(worst, fabs(ref - (double)ho[l * 16 + r]));
      big = fmax(big, fabs(ref));
    }
  printf("fp6 e2m3 MFMA vs host decode: max |d| = %.3e (max |ref| %.3e) -> %s\n", worst, big, worst <= 1e-6 * big ? "layout and scales as assumed" : "MISMATCH");
  // ---- (2) time
  const int n = 1 << 16;
  u32x4* src;
  float* out;
  hipMalloc(&src, n * sizeof(u32x4));
  hipMalloc(&out, 256 * 2 * 512 * 4);
  uint32_t* h = (uint32_t*)malloc(n * 16);
  for (int i = 0; i < n * 4; ++i) h[i] = ((rand() & 0x3ff) | 0x3800) * 0x00010001u;  // fp16 values near 0.5 .. 1
  hipMemcpy(src, h, n * 16, hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 4000;
  for (int mode = 0; mode < 3; ++mode) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) k<0><<<512, 512>>>(src, out, iters);
      else if (mode == 1) k<1><<<512, 512>>>(src, out, iters);
      else k<2><<<512, 512>>>(src, out, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      best = ms < best ? ms : best;
    }
    const double blocks = 512.0 * 8 * 8 * iters;  // 32 x 32 x 32-k block products
    printf("%s: %.3f ms  -> %.2f ns per 1e3 block products  (%.1f bf16-equivalent TFLOP/s of 32x32x32 blocks)\n",
           mode == 0 ? "h8 (2 f16 + 1 e4m3)" : (mode == 1 ? "h6 (2 f16 + 1 fp6) " : "fp6 MFMA alone     "), best, best * 1e6 / blocks * 1e3,
           blocks * 2.0 * 32 * 32 * 32 / (best * 1e-3) / 1e12);
  }
  return 0;
}
