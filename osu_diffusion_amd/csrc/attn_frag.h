// MFMA operand fragments shared by the attention forward and backward kernels (bf16 tier, head_dim 64):
// LDS tiles are row-major [rows][padded head width]; see AttnTile for the two geometries.
#pragma once
#include "common.h"

namespace osud {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// Tile geometry by padded head width HDP (head_dim rounded up to a multiple of 32; the pad columns hold zeros):
//   HDP = 64 : 128-byte rows, 16-byte chunk index XOR-swizzled with (row>>1)&7 (conflict-free ds_read_b128)
//   HDP = 96 : 192-byte rows padded to 208 (52 dwords: 16 consecutive rows start in 16 different 4-bank groups), no swizzle
//              -- DiT-XL's head_dim 72
template <int HDP> struct AttnTile {
  static_assert(HDP == 64 || HDP == 96, "attention tiles are built for padded head widths 64 and 96");
  static constexpr int RS = HDP == 64 ? 128 : 208;  // row stride in bytes
  static constexpr int CPR = HDP / 8;               // 16-byte chunks per row
  static __device__ __forceinline__ int off(int row, int chunk) {
    return row * RS + (HDP == 64 ? ((chunk ^ ((row >> 1) & 7)) << 4) : (chunk << 4));
  }
};

template <int HDP> __device__ __forceinline__ u32x4 rowfrag(const char* tile, int row, int chunk) {
  return *reinterpret_cast<const u32x4*>(tile + AttnTile<HDP>::off(row, chunk));
}
// A operand whose contraction index is the ROW of a row-major [rows][HDP] tile: lane (d = d0 + (lane&31), half) gets, for
// column d, the 8 rows
//   r0 + 4*half + {0..3}  and  r0 + 8 + 4*half + {0..3}
// — the same permuted order in which P / dS leave the S-layout registers (see pack8) — via two transposing reads
// (each 16-lane group: 4 rows x 16 columns; semantics pinned by tools/probes/tr_probe.hip).
template <int HDP> __device__ __forceinline__ u32x4 trfrag(const char* tile, int r0, int d0, int lane) {
  const int row = r0 + 4 * (lane >> 5) + ((lane & 15) >> 2);
  const int colb = (d0 + 16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;  // byte offset of this lane's 4 columns
  const int chunk = colb >> 4, within = colb & 15;
  const uint32_t base = (uint32_t)(size_t)(const __attribute__((address_space(3))) void*)tile;
  const uint32_t a0 = base + AttnTile<HDP>::off(row, chunk) + within;
  const uint32_t a1 = base + AttnTile<HDP>::off(row + 8, chunk) + within;
  // reads and their wait are ONE asm statement: a separate s_waitcnt statement does not stop the scheduler from
  // moving the consumers (register moves, MFMA) above it, because the asm outputs look ready to the compiler
  u32x2 lo, hi;
  asm volatile(
      "ds_read_b64_tr_b16 %0, %2\n\t"
      "ds_read_b64_tr_b16 %1, %3\n\t"
      "s_waitcnt lgkmcnt(0)"
      : "=&v"(lo), "=&v"(hi)
      : "v"(a0), "v"(a1)
      : "memory");
  u32x4 v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
  return v;
}
__device__ __forceinline__ f32x16 mfma_bf16(const u32x4& a, const u32x4& b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ u32x4 pack8(const f32x16& v, int base) {
  u32x4 r;
  r[0] = pack_bf2(v[base + 0], v[base + 1]);
  r[1] = pack_bf2(v[base + 2], v[base + 3]);
  r[2] = pack_bf2(v[base + 4], v[base + 5]);
  r[3] = pack_bf2(v[base + 6], v[base + 7]);
  return r;
}

}  // namespace osud
