// MFMA operand fragments shared by the attention forward and backward kernels (bf16 tier, head_dim 64):
// LDS tiles are row-major [rows][padded head width]; see AttnTile for the two geometries.
#pragma once
#include "common.h"

namespace osud {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(3))) uint32_t lds_u32;  // (a word read through an LDS pointer is a ds_read; through a generic
                                                             //  pointer it is a flat load, which also counts as vector memory)

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
  // (the compiler's own builtin: it tracks the LDS counter itself, so these reads are hoisted above the MFMAs of the previous
  // k-step and waited for only where they are used; the first version issued them from inline asm with the wait in the same
  // statement -- every fragment cost a full LDS round trip with the matrix pipe idle)
  typedef short s16x4 __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const lds_s16x4* base = (const lds_s16x4*)(const __attribute__((address_space(3))) char*)tile;
  const lds_s16x4* p0 = (const lds_s16x4*)((const __attribute__((address_space(3))) char*)base + AttnTile<HDP>::off(row, chunk) + within);
  const lds_s16x4* p1 = (const lds_s16x4*)((const __attribute__((address_space(3))) char*)base + AttnTile<HDP>::off(row + 8, chunk) + within);
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(const_cast<lds_s16x4*>(p0));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(const_cast<lds_s16x4*>(p1));
  const u32x2 l2 = __builtin_bit_cast(u32x2, lo), h2 = __builtin_bit_cast(u32x2, hi);
  u32x4 v;
  v[0] = l2[0]; v[1] = l2[1]; v[2] = h2[0]; v[3] = h2[1];
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
// F16 (the fp16 tier, forward kernels only): the same 16-bit containers hold IEEE half values -- the loads, LDS tiles and transposing
// reads do not care; the MFMA and the f32 -> 16-bit packs do
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_attn;
template <bool F16> __device__ __forceinline__ uint32_t pack2_h(float a, float b) { return F16 ? pack_f16x2(a, b) : pack_bf2(a, b); }
template <bool F16> __device__ __forceinline__ f32x16 mfma_h(const u32x4& a, const u32x4& b, f32x16 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_attn, a), __builtin_bit_cast(f16x8_attn, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <bool F16> __device__ __forceinline__ u32x4 pack8_h(const f32x16& v, int base) {
  u32x4 r;
  r[0] = pack2_h<F16>(v[base + 0], v[base + 1]);
  r[1] = pack2_h<F16>(v[base + 2], v[base + 3]);
  r[2] = pack2_h<F16>(v[base + 4], v[base + 5]);
  r[3] = pack2_h<F16>(v[base + 6], v[base + 7]);
  return r;
}

// a * b ROUNDED to fp32 before anything consumes it: an output split into hi + lo (or fp16 + residual) parts subtracts the hi part from the
// product, and the compiler contracts that into one fma in some kernels and not in others (it depends on how many uses the product
// has after inlining) -- two kernels that must produce the same bits pin the product with this
__device__ __forceinline__ float mul_rn(float a, float b) {
  float v = a * b;
  asm("" : "+v"(v));
  return v;
}

// split-bf16 tier: the lo halves of the same 8 values, given their packed hi halves: bf16(v - float(hi))
__device__ __forceinline__ u32x4 pack8_lo(const f32x16& v, int base, const u32x4& hi) {
  u32x4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i)
    r[i] = pack_bf2(v[base + 2 * i] - __uint_as_float(hi[i] << 16), v[base + 2 * i + 1] - __uint_as_float(hi[i] & 0xffff0000u));
  return r;
}

// 32 accumulator rows of a wave (lane = row, 64 columns) -> bf16 rows in global memory through a 2 KiB LDS patch of the wave:
// a lane owns a ROW of the accumulators, so a direct store instruction touches 32 rows with 16 bytes each -- 24 such
// instructions per head and wave were 56 of the streamed backward kernel's 132 us (tools/attn_bench.py, stores compiled out).  Sixteen rows at a time go to the patch (16-byte chunks XOR-swizzled with
// the row) and come back as 16 bytes per lane, eight lanes per 128-byte row: every store instruction writes eight full lines.
// COLSUM: the column sums of the 32 rows ride along -- the 16 rows that sit in the patch (the AttnTile<64> layout, so trfrag reads
// it) are contracted with a ones operand on the matrix pipe: D[d][*] += sum_rows bf16(row)[d].  Every lane ends with 16 of the
// 64 column sums per accumulator (row index of the MFMA result = column d); lanes 0 and 32 together hold all of them.
template <bool COLSUM = false, bool F16 = false>
__device__ __forceinline__ void store_rows_patch(char* patch, bf16_t* rows, size_t ld, const f32x16 (&acc)[2], int lane,
                                                 f32x16* cacc = nullptr) {
  static_assert(!(COLSUM && F16), "the column sums ride with the bf16 backward pass only");
  if constexpr (COLSUM) asm volatile("" : "+v"(lane));  // (opaque: per-lane offsets recomputed here, not held in registers by the caller's loop)
  const int frow = lane & 31, fhalf = lane >> 5;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if ((frow >> 4) == half) {
      const int r = frow & 15;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          u32x2 v;
          v[0] = pack2_h<F16>(acc[dt][4 * g], acc[dt][4 * g + 1]);
          v[1] = pack2_h<F16>(acc[dt][4 * g + 2], acc[dt][4 * g + 3]);
          *reinterpret_cast<u32x2*>(patch + AttnTile<64>::off(r, dt * 4 + g) + fhalf * 8) = v;
        }
    }
    asm volatile("" ::: "memory");  // (LDS operations of a wave execute in order; this only pins the compiler's order)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int rr = (lane >> 3) + 8 * j, c = lane & 7;
      const u32x4 v = *reinterpret_cast<const u32x4*>(patch + AttnTile<64>::off(rr, c));
      *reinterpret_cast<u32x4*>(rows + (size_t)(16 * half + rr) * ld + c * 8) = v;
    }
    if constexpr (COLSUM) {
      const u32x4 ones = {0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};  // bf16 1.0 x 8
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) cacc[dt] = mfma_bf16(trfrag<64>(patch, 0, dt * 32, lane), ones, cacc[dt]);
    }
    asm volatile("" ::: "memory");
  }
}

// ---- pieces of the persistent streamed kernels for head_dim 72 (tiles of 208-byte rows, 96 padded columns)
static __device__ uint4 g_attn_zero16 = {0u, 0u, 0u, 0u};  // source of the zero pad columns 72..95 of a tile row

__device__ __forceinline__ void gload16(u32x4& dst, const void* p) {
  asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(dst) : "v"(p) : "memory");
}
__device__ __forceinline__ void gload4(float& dst, const void* p) {
  asm volatile("global_load_dword %0, %1, off" : "=&v"(dst) : "v"(p) : "memory");
}
#define OSUD_VM_WAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// 32 accumulator rows (lane = row, 72 real of 96 columns) -> bf16 rows of 144 bytes through a 16-row LDS patch of the wave:
// 6 store instructions per tile (2 x 16 rows x 9 chunks = 288 chunks of 16 bytes), each writing whole 144-byte rows
__device__ __forceinline__ void store_rows_patch72(char* patch, bf16_t* rows, size_t ld, const f32x16 (&acc)[3], int lane) {
  asm volatile("" : "+v"(lane));  // (opaque: per-lane offsets are recomputed per call instead of living in registers across the passes)
  const int frow = lane & 31, fhalf = lane >> 5;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if ((frow >> 4) == half) {
      const int r = frow & 15;
#pragma unroll
      for (int dt = 0; dt < 3; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = dt * 32 + 8 * g + 4 * fhalf;
          if (d < 72) {
            u32x2 v;
            v[0] = pack_bf2(acc[dt][4 * g], acc[dt][4 * g + 1]);
            v[1] = pack_bf2(acc[dt][4 * g + 2], acc[dt][4 * g + 3]);
            *reinterpret_cast<u32x2*>(patch + r * 208 + d * 2) = v;
          }
        }
    }
    asm volatile("" ::: "memory");  // (LDS operations of a wave execute in order; this only pins the compiler's order)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int idx = j * 64 + lane;
      if (idx < 144) {
        const int rr = idx / 9, c = idx - 9 * rr;
        const u32x4 v = *reinterpret_cast<const u32x4*>(patch + rr * 208 + c * 16);
        *reinterpret_cast<u32x4*>(rows + (size_t)(16 * half + rr) * ld + c * 8) = v;
      }
    }
    asm volatile("" ::: "memory");
  }
}

// ---- host side of the kernels that want more than 64 KiB of dynamic LDS: the attribute is per device, so remember it per device
//      (one process normally drives one GPU; tests and tools may touch several)
inline int device_index() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
  return dev;
}
inline int device_cus() {
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_index()) != hipSuccess || cus <= 0) cus = 256;
  return cus;
}
#define OSUD_BIG_LDS_ONCE(kernel)                                                                                                  \
  do {                                                                                                                             \
    static bool done_[64] = {};                                                                                                    \
    const int d_ = device_index() & 63;                                                                                            \
    if (!done_[d_]) {                                                                                                              \
      OSUD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
      done_[d_] = true;                                                                                                            \
    }                                                                                                                              \
  } while (0)

}  // namespace osud
