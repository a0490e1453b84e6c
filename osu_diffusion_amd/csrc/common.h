// Shared declarations for libosud.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/osud.h"

namespace osud {

// ---- error plumbing: no C++ exception crosses the ABI -------------------------------
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what, const char* file, int line);

#define OSUD_HIP(call)                                                      \
  do {                                                                      \
    hipError_t e__ = (call);                                                \
    if (e__ != hipSuccess) return ::osud::hip_fail(e__, #call, __FILE__, __LINE__); \
  } while (0)

#define OSUD_CHECK_ARG(cond, ...)        \
  do {                                   \
    if (!(cond)) {                       \
      ::osud::set_error(__VA_ARGS__);    \
      return OSUD_ERR_ARG;               \
    }                                    \
  } while (0)

#define OSUD_TRY(expr)            \
  do {                            \
    int rc__ = (expr);            \
    if (rc__ != OSUD_OK) return rc__; \
  } while (0)

// ---- element types of the two arithmetic tiers ---------------------------------------
typedef uint16_t bf16_t;  // storage only
struct fp8_t { uint8_t v; };  // OCP e4m3 storage (experimental GEMM operand type: osud_op_gemm precision 2)
// Split-bf16 tier (OSUD_PREC_BF16X3): an fp32 value v travels as the pair hi = bf16(v), lo = bf16(v - hi), i.e. 16 significand bits
// (2^-17 relative, finer than the TF32 the reference's matmuls use); a product is three bf16 MFMAs: hi*hi + lo*hi + hi*lo.
// LAYOUT: a matrix with logical leading dimension ld is a bf16 array whose rows are 2 * ld long -- the hi plane in columns
// [0, ld), the lo plane in [ld, 2 ld).  x3_t is the element type of such a matrix in templates (one plane element = one bf16).
struct x3_t { uint16_t v; };
// fp16 + e4m3 residual tier (OSUD_PREC_F16F8, the four big GEMMs of a block inside the split-bf16 tier): an fp32 value v travels as
//   hi = fp16(v) (11 significand bits), lo8 = e4m3((v - hi) * 2^12) (4 more), hi8 = e4m3(v) (the partner of the OTHER operand's lo8)
// and a product over 32 k is   hi.hi  (two v_mfma_f32_32x32x16_f16)  +  2^-12 (lo8_a.hi8_w + hi8_a.lo8_w)  (ONE block-scaled
// v_mfma_scale_f32_32x32x64_f8f6f4 over the K-concatenated e4m3 planes): 32 MFMA passes where the split-bf16 form needs 48, at
// 15 significand bits per operand (split-bf16: 16; TF32, the reference's sampling arithmetic: 11).
// LAYOUT (h8_t: 4 bytes per logical element, K-blocked): a row is a sequence of 128-byte groups of 32 logical k --
//   [ 64 B: 32 x fp16 hi | 32 B: plane P | 32 B: plane Q ],   activations: P = lo8, Q = hi8;  weights: P = hi8, Q = lo8
// -- exactly one stage row of the GEMM's LDS image, so the operand travels by the plain 128-byte LDS-DMA; P meets P and Q meets Q
// in the fp8 MFMA, which pairs lo8_a with hi8_w and hi8_a with lo8_w.  Leading dimensions and K are multiples of 32.
struct h8_t { uint8_t b[4]; };
// fp16 activations x (fp16 + e4m3 residual) weights (OSUD_PREC_F16W8; the operand form of GEMMs whose ACTIVATION may be rounded to
// fp16's 11 bits while the WEIGHT keeps 15): rounding an activation is a fresh random error per token and step, rounding a weight
// is the same error in every product of every step -- measured on the 1000-step loop: both in fp16 5.5e-3 from the fp32 tier, only
// the activations 6.6e-4, neither (h8_t) 1.3e-4.  A product over 128 k is
//   a_hi . w_hi  (eight v_mfma_f32_32x32x16_f16)  +  2^-12 a_hi8 . w_lo8  (two v_mfma_scale_f32_32x32x64_f8f6f4, weight-side scale)
// = 96 matrix-pipe passes per 32 x 32 block where the h8_t form issues 128 and the split-bf16 form 192.
// LAYOUT (w8_t: 3 bytes per logical element, K-blocked): a row is a sequence of 384-byte super-groups of 128 logical k --
//   [ 128 B: fp16 of k 0..63 | 128 B: fp16 of k 64..127 | 128 B: one e4m3 per k 0..127 ],
//   activations: e4m3 plane = e4m3(v) (the partner of the weight's residual);  weights: e4m3 plane = e4m3((w - fp16(w)) 2^12)
// -- three stage rows of the GEMM's LDS image (two fp16 slabs, one e4m3 slab).  Leading dimensions and K are multiples of 128.
struct w8_t { uint8_t b[3]; };
constexpr float kH8LoScale = 4096.0f;  // 2^12: lo8 = e4m3((v - hi) * 2^12); the fp8 MFMA's A-side block scale is 2^-12
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// fp32 -> bf16, round-to-nearest-even: gfx950 has the conversion in hardware (v_cvt_pk_bf16_f32)
typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
typedef float f32x2_hw __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
  f32x2_hw v;
  v[0] = lo;
  v[1] = hi;
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_hw));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((uint32_t)h) << 16); }

// planes per row of a TE matrix (row stride = kPlanes * ld elements)
template <typename TE> struct Planes { static constexpr int k = 1; };
template <> struct Planes<x3_t> { static constexpr int k = 2; };

template <typename TE> struct ElemTraits;
template <> struct ElemTraits<bf16_t> {
  static constexpr int kPerChunk = 8;  // elements per 16-byte chunk
  static constexpr int kPrec = OSUD_PREC_BF16;
};
template <> struct ElemTraits<float> {
  static constexpr int kPerChunk = 4;
  static constexpr int kPrec = OSUD_PREC_F32;
};

__device__ __forceinline__ void store_elem(bf16_t* p, float v) { *p = f2bf(v); }
__device__ __forceinline__ void store_elem(float* p, float v) { *p = v; }
__device__ __forceinline__ float load_elem(const bf16_t* p) { return bf2f(*p); }
__device__ __forceinline__ float load_elem(const float* p) { return *p; }

// 4 consecutive elements (8 B bf16 / 16 B f32), pointer suitably aligned
__device__ __forceinline__ void store4(bf16_t* p, float a, float b, float c, float d) {
  uint2 v;
  v.x = pack_bf2(a, b);
  v.y = pack_bf2(c, d);
  *reinterpret_cast<uint2*>(p) = v;
}
__device__ __forceinline__ void store4(float* p, float a, float b, float c, float d) {
  *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
}

// 8 consecutive elements (16 B bf16 / 32 B f32), pointer 16-byte aligned
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
  uint4 u;
  u.x = pack_bf2(v[0], v[1]);
  u.y = pack_bf2(v[2], v[3]);
  u.z = pack_bf2(v[4], v[5]);
  u.w = pack_bf2(v[6], v[7]);
  *reinterpret_cast<uint4*>(p) = u;
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
  const uint4 u = *reinterpret_cast<const uint4*>(p);
  v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
  v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
  v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
  v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
}
// ---- fp16 tier (OSUD_PREC_F16, inference only): the bf16 tier's kernels on IEEE half operands -- 11 significand bits, which is what
// the TF32 matmuls of the reference's own sampling path carry (sample.py:25-26), at the bf16 tier's MFMA rate (v_mfma_f32_32x32x16_f16).
// The activations of a forward pass sit far inside half's range (|v| < 65504); the tier is not built for gradients.
struct f16_t { uint16_t v; };
typedef _Float16 f16x2_rn __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_f16x2(float lo, float hi) {  // round-to-nearest-even
  f32x2_hw v;
  v[0] = lo;
  v[1] = hi;
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2_rn));
}
__device__ __forceinline__ float h2f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ void unpack_f16x2(uint32_t u, float& lo, float& hi) {
  const f16x2_rn h = __builtin_bit_cast(f16x2_rn, u);
  lo = (float)h[0];
  hi = (float)h[1];
}
__device__ __forceinline__ void store_elem(f16_t* p, float v) { p->v = (uint16_t)(pack_f16x2(v, 0.f) & 0xffffu); }
__device__ __forceinline__ float load_elem(const f16_t* p) { return h2f(p->v); }
__device__ __forceinline__ void store2(f16_t* p, float a, float b) { *reinterpret_cast<uint32_t*>(p) = pack_f16x2(a, b); }
__device__ __forceinline__ void store4(f16_t* p, float a, float b, float c, float d) {
  uint2 v;
  v.x = pack_f16x2(a, b);
  v.y = pack_f16x2(c, d);
  *reinterpret_cast<uint2*>(p) = v;
}
__device__ __forceinline__ void store8(f16_t* p, const float (&v)[8]) {
  uint4 u;
  u.x = pack_f16x2(v[0], v[1]);
  u.y = pack_f16x2(v[2], v[3]);
  u.z = pack_f16x2(v[4], v[5]);
  u.w = pack_f16x2(v[6], v[7]);
  *reinterpret_cast<uint4*>(p) = u;
}
__device__ __forceinline__ void load2(const f16_t* p, float& a, float& b) { unpack_f16x2(*reinterpret_cast<const uint32_t*>(p), a, b); }
__device__ __forceinline__ void load8(const f16_t* p, float (&v)[8]) {
  const uint4 u = *reinterpret_cast<const uint4*>(p);
  unpack_f16x2(u.x, v[0], v[1]);
  unpack_f16x2(u.y, v[2], v[3]);
  unpack_f16x2(u.z, v[4], v[5]);
  unpack_f16x2(u.w, v[6], v[7]);
}
// fp32 -> OCP e4m3 (hardware v_cvt_pk_fp8_f32, round-to-nearest-even), saturating at the format's +-448
__device__ __forceinline__ uint32_t pack_fp8x4(float a, float b, float c, float d) {
  a = fminf(fmaxf(a, -448.f), 448.f); b = fminf(fmaxf(b, -448.f), 448.f);
  c = fminf(fmaxf(c, -448.f), 448.f); d = fminf(fmaxf(d, -448.f), 448.f);
  int r = 0;
  r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, r, false);
  r = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
  return (uint32_t)r;
}
__device__ __forceinline__ void store8(fp8_t* p, const float (&v)[8]) {
  uint2 u;
  u.x = pack_fp8x4(v[0], v[1], v[2], v[3]);
  u.y = pack_fp8x4(v[4], v[5], v[6], v[7]);
  *reinterpret_cast<uint2*>(p) = u;
}
__device__ __forceinline__ void store4(fp8_t* p, float a, float b, float c, float d) {
  *reinterpret_cast<uint32_t*>(p) = pack_fp8x4(a, b, c, d);
}
__device__ __forceinline__ void store2(fp8_t* p, float a, float b) {
  *reinterpret_cast<uint16_t*>(p) = (uint16_t)(pack_fp8x4(a, b, 0.f, 0.f) & 0xffffu);
}
__device__ __forceinline__ void load2(const fp8_t*, float& a, float& b) { a = b = 0.f; }  // never read back
__device__ __forceinline__ void load8(const fp8_t*, float (&v)[8]) {  // fp8 tensors are never epilogue operands
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = 0.f;
}
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// ---- split-bf16 rows: W consecutive logical elements at `p` (hi plane), their lo halves `ld` elements further on
__device__ __forceinline__ void split_hi_lo(float v0, float v1, uint32_t& hi, uint32_t& lo) {
  hi = pack_bf2(v0, v1);
  lo = pack_bf2(v0 - __uint_as_float(hi << 16), v1 - __uint_as_float(hi & 0xffff0000u));  // (both differences are exact in fp32)
}
__device__ __forceinline__ void store2_x3(bf16_t* p, size_t ld, float a, float b) {
  uint32_t hi, lo;
  split_hi_lo(a, b, hi, lo);
  *reinterpret_cast<uint32_t*>(p) = hi;
  *reinterpret_cast<uint32_t*>(p + ld) = lo;
}
__device__ __forceinline__ void pack4_x3(float a, float b, float c, float d, uint2& hi, uint2& lo) {
  split_hi_lo(a, b, hi.x, lo.x);
  split_hi_lo(c, d, hi.y, lo.y);
}
__device__ __forceinline__ void store4_x3(bf16_t* p, size_t ld, float a, float b, float c, float d) {
  uint2 hi, lo;
  pack4_x3(a, b, c, d, hi, lo);
  *reinterpret_cast<uint2*>(p) = hi;
  *reinterpret_cast<uint2*>(p + ld) = lo;
}
__device__ __forceinline__ void store8_x3(bf16_t* p, size_t ld, const float (&v)[8]) {
  uint4 hi, lo;
  split_hi_lo(v[0], v[1], hi.x, lo.x);
  split_hi_lo(v[2], v[3], hi.y, lo.y);
  split_hi_lo(v[4], v[5], hi.z, lo.z);
  split_hi_lo(v[6], v[7], hi.w, lo.w);
  *reinterpret_cast<uint4*>(p) = hi;
  *reinterpret_cast<uint4*>(p + ld) = lo;
}
__device__ __forceinline__ void store_elem_x3(bf16_t* p, size_t ld, float v) {
  const bf16_t hi = f2bf(v);
  p[0] = hi;
  p[ld] = f2bf(v - bf2f(hi));
}
// ---- fp16 + e4m3 rows (h8_t): 8 (4) consecutive logical elements at column x (x % 8 == 0 (4)) of the row starting at `row`
typedef _Float16 f16x2_hw __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_h2(float a, float b, float& ra, float& rb) {  // fp16 pair (RNE) and the residuals
  f32x2_hw v;
  v[0] = a;
  v[1] = b;
  const f16x2_hw h = __builtin_convertvector(v, f16x2_hw);
  ra = a - (float)h[0];
  rb = b - (float)h[1];
  return __builtin_bit_cast(uint32_t, h);
}
// (the bytes of 4 consecutive elements: `hi` for byte 2 i of the group, `p64` / `p96` for bytes 64 + i / 96 + i)
template <bool WEIGHT> __device__ __forceinline__ void pack4_h8(float a, float b, float c, float d, uint2& hi, uint32_t& p64, uint32_t& p96) {
  float r0, r1, r2, r3;
  hi.x = pack_h2(a, b, r0, r1);
  hi.y = pack_h2(c, d, r2, r3);
  const uint32_t lo8 = pack_fp8x4(r0 * kH8LoScale, r1 * kH8LoScale, r2 * kH8LoScale, r3 * kH8LoScale), hi8 = pack_fp8x4(a, b, c, d);
  p64 = WEIGHT ? hi8 : lo8;
  p96 = WEIGHT ? lo8 : hi8;
}
template <bool WEIGHT> __device__ __forceinline__ void store4_h8(h8_t* row, int x, float a, float b, float c, float d) {
  char* g = reinterpret_cast<char*>(row) + (size_t)(x >> 5) * 128;
  const int i = x & 31;
  uint2 hi;
  uint32_t p64, p96;
  pack4_h8<WEIGHT>(a, b, c, d, hi, p64, p96);
  *reinterpret_cast<uint2*>(g + 2 * i) = hi;
  *reinterpret_cast<uint32_t*>(g + 64 + i) = p64;
  *reinterpret_cast<uint32_t*>(g + 96 + i) = p96;
}
template <bool WEIGHT> __device__ __forceinline__ void store8_h8(h8_t* row, int x, const float (&v)[8]) {
  char* g = reinterpret_cast<char*>(row) + (size_t)(x >> 5) * 128;
  const int i = x & 31;
  float r[8];
  uint4 hi;
  hi.x = pack_h2(v[0], v[1], r[0], r[1]);
  hi.y = pack_h2(v[2], v[3], r[2], r[3]);
  hi.z = pack_h2(v[4], v[5], r[4], r[5]);
  hi.w = pack_h2(v[6], v[7], r[6], r[7]);
  *reinterpret_cast<uint4*>(g + 2 * i) = hi;
  uint2 lo8, hi8;
  lo8.x = pack_fp8x4(r[0] * kH8LoScale, r[1] * kH8LoScale, r[2] * kH8LoScale, r[3] * kH8LoScale);
  lo8.y = pack_fp8x4(r[4] * kH8LoScale, r[5] * kH8LoScale, r[6] * kH8LoScale, r[7] * kH8LoScale);
  hi8.x = pack_fp8x4(v[0], v[1], v[2], v[3]);
  hi8.y = pack_fp8x4(v[4], v[5], v[6], v[7]);
  *reinterpret_cast<uint2*>(g + 64 + i) = WEIGHT ? hi8 : lo8;
  *reinterpret_cast<uint2*>(g + 96 + i) = WEIGHT ? lo8 : hi8;
}
template <bool WEIGHT> __device__ __forceinline__ void store2_h8(h8_t* row, int x, float a, float b) {
  char* g = reinterpret_cast<char*>(row) + (size_t)(x >> 5) * 128;
  const int i = x & 31;
  float r0, r1;
  *reinterpret_cast<uint32_t*>(g + 2 * i) = pack_h2(a, b, r0, r1);
  const uint16_t lo8 = (uint16_t)(pack_fp8x4(r0 * kH8LoScale, r1 * kH8LoScale, 0.f, 0.f) & 0xffffu), hi8 = (uint16_t)(pack_fp8x4(a, b, 0.f, 0.f) & 0xffffu);
  *reinterpret_cast<uint16_t*>(g + 64 + i) = WEIGHT ? hi8 : lo8;
  *reinterpret_cast<uint16_t*>(g + 96 + i) = WEIGHT ? lo8 : hi8;
}
// ---- fp16 + one e4m3 plane rows (w8_t): 8 / 4 / 2 consecutive logical elements at column x of the row starting at `row`
template <bool WEIGHT> __device__ __forceinline__ void pack4_w8(float a, float b, float c, float d, uint2& hi, uint32_t& p8) {
  float r0, r1, r2, r3;
  hi.x = pack_h2(a, b, r0, r1);
  hi.y = pack_h2(c, d, r2, r3);
  p8 = WEIGHT ? pack_fp8x4(r0 * kH8LoScale, r1 * kH8LoScale, r2 * kH8LoScale, r3 * kH8LoScale) : pack_fp8x4(a, b, c, d);
}
template <bool WEIGHT> __device__ __forceinline__ void store4_w8(w8_t* row, int x, float a, float b, float c, float d) {
  char* g = reinterpret_cast<char*>(row) + (size_t)(x >> 7) * 384;
  const int i = x & 127;
  uint2 hi;
  uint32_t p8;
  pack4_w8<WEIGHT>(a, b, c, d, hi, p8);
  *reinterpret_cast<uint2*>(g + 2 * i) = hi;
  *reinterpret_cast<uint32_t*>(g + 256 + i) = p8;
}
template <bool WEIGHT> __device__ __forceinline__ void store8_w8(w8_t* row, int x, const float (&v)[8]) {
  char* g = reinterpret_cast<char*>(row) + (size_t)(x >> 7) * 384;
  const int i = x & 127;
  float r[8];
  uint4 hi;
  hi.x = pack_h2(v[0], v[1], r[0], r[1]);
  hi.y = pack_h2(v[2], v[3], r[2], r[3]);
  hi.z = pack_h2(v[4], v[5], r[4], r[5]);
  hi.w = pack_h2(v[6], v[7], r[6], r[7]);
  *reinterpret_cast<uint4*>(g + 2 * i) = hi;
  uint2 p8;
  if constexpr (WEIGHT) {
    p8.x = pack_fp8x4(r[0] * kH8LoScale, r[1] * kH8LoScale, r[2] * kH8LoScale, r[3] * kH8LoScale);
    p8.y = pack_fp8x4(r[4] * kH8LoScale, r[5] * kH8LoScale, r[6] * kH8LoScale, r[7] * kH8LoScale);
  } else {
    p8.x = pack_fp8x4(v[0], v[1], v[2], v[3]);
    p8.y = pack_fp8x4(v[4], v[5], v[6], v[7]);
  }
  *reinterpret_cast<uint2*>(g + 256 + i) = p8;
}
template <bool WEIGHT> __device__ __forceinline__ void store2_w8(w8_t* row, int x, float a, float b) {
  char* g = reinterpret_cast<char*>(row) + (size_t)(x >> 7) * 384;
  const int i = x & 127;
  float r0, r1;
  *reinterpret_cast<uint32_t*>(g + 2 * i) = pack_h2(a, b, r0, r1);
  *reinterpret_cast<uint16_t*>(g + 256 + i) =
      (uint16_t)((WEIGHT ? pack_fp8x4(r0 * kH8LoScale, r1 * kH8LoScale, 0.f, 0.f) : pack_fp8x4(a, b, 0.f, 0.f)) & 0xffffu);
}
template <int W> __device__ __forceinline__ void storew_w8(w8_t* row, int x, const float* v) {  // (an activation row)
  if constexpr (W == 4) store4_w8<false>(row, x, v[0], v[1], v[2], v[3]);
  else store2_w8<false>(row, x, v[0], v[1]);
}
__device__ __forceinline__ void store8(w8_t*, const float (&)[8]) { __builtin_trap(); }
__device__ __forceinline__ void store4(w8_t*, float, float, float, float) { __builtin_trap(); }
__device__ __forceinline__ void store2(w8_t*, float, float) { __builtin_trap(); }
__device__ __forceinline__ void store_elem(w8_t*, float) { __builtin_trap(); }
__device__ __forceinline__ void load8(const w8_t*, float (&v)[8]) {
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
  __builtin_trap();
}
__device__ __forceinline__ void load2(const w8_t*, float& a, float& b) { a = b = 0.f; __builtin_trap(); }
__device__ __forceinline__ float load_elem(const w8_t*) { __builtin_trap(); return 0.f; }
template <int W> __device__ __forceinline__ void storew_h8(h8_t* row, int x, const float* v) {  // (an activation row)
  if constexpr (W == 4) store4_h8<false>(row, x, v[0], v[1], v[2], v[3]);
  else store2_h8<false>(row, x, v[0], v[1]);
}
// (like x3_t: written through the helpers above, never read back element-wise; these overloads only let shared templates compile)
__device__ __forceinline__ void store8(h8_t*, const float (&)[8]) { __builtin_trap(); }
__device__ __forceinline__ void store4(h8_t*, float, float, float, float) { __builtin_trap(); }
__device__ __forceinline__ void store2(h8_t*, float, float) { __builtin_trap(); }
__device__ __forceinline__ void store_elem(h8_t*, float) { __builtin_trap(); }
__device__ __forceinline__ void load8(const h8_t*, float (&v)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = 0.f;
  __builtin_trap();
}
__device__ __forceinline__ void load2(const h8_t*, float& a, float& b) { a = b = 0.f; __builtin_trap(); }
__device__ __forceinline__ float load_elem(const h8_t*) { __builtin_trap(); return 0.f; }

// x3_t tensors are written through the *_x3 helpers above (they need the plane distance) and never read back element-wise by the
// row-wise kernels: these overloads only let the shared templates compile -- reaching one is a bug
__device__ __forceinline__ void store8(x3_t*, const float (&)[8]) { __builtin_trap(); }
__device__ __forceinline__ void store4(x3_t*, float, float, float, float) { __builtin_trap(); }
__device__ __forceinline__ void store2(x3_t*, float, float) { __builtin_trap(); }
__device__ __forceinline__ void store_elem(x3_t*, float) { __builtin_trap(); }
__device__ __forceinline__ void load8(const x3_t*, float (&v)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = 0.f;
  __builtin_trap();
}
__device__ __forceinline__ void load2(const x3_t*, float& a, float& b) { a = b = 0.f; __builtin_trap(); }
__device__ __forceinline__ float load_elem(const x3_t*) { __builtin_trap(); return 0.f; }

// 2 consecutive elements (4 B bf16 / 8 B f32)
__device__ __forceinline__ void store2(bf16_t* p, float a, float b) { *reinterpret_cast<uint32_t*>(p) = pack_bf2(a, b); }
__device__ __forceinline__ void store2(float* p, float a, float b) { *reinterpret_cast<float2*>(p) = make_float2(a, b); }

__device__ __forceinline__ void load2(const bf16_t* p, float& a, float& b) {
  const uint32_t u = *reinterpret_cast<const uint32_t*>(p);
  a = __uint_as_float(u << 16);
  b = __uint_as_float(u & 0xffff0000u);
}
__device__ __forceinline__ void load2(const float* p, float& a, float& b) {
  const float2 v = *reinterpret_cast<const float2*>(p);
  a = v.x;
  b = v.y;
}

// W consecutive elements (W = 2 or 4): the row-wise kernels give every lane W columns per group so that hidden sizes whose
// per-lane count is a multiple of 4 (768, 1024) move 16 bytes per fp32 access, the others (1152 = 18 per lane) 8
template <int W> __device__ __forceinline__ void loadw(const float* p, float* v) {
  if constexpr (W == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
    const float2 t = *reinterpret_cast<const float2*>(p);
    v[0] = t.x; v[1] = t.y;
  }
}
template <int W> __device__ __forceinline__ void loadw(const bf16_t* p, float* v) {
  if constexpr (W == 4) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
  } else {
    load2(p, v[0], v[1]);
  }
}
template <int W> __device__ __forceinline__ void loadw(const f16_t* p, float* v) {
  if constexpr (W == 4) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    unpack_f16x2(u.x, v[0], v[1]);
    unpack_f16x2(u.y, v[2], v[3]);
  } else {
    load2(p, v[0], v[1]);
  }
}
template <int W> __device__ __forceinline__ void storew(f16_t* p, const float* v) {
  if constexpr (W == 4) store4(p, v[0], v[1], v[2], v[3]);
  else store2(p, v[0], v[1]);
}
template <int W> __device__ __forceinline__ void loadw(const fp8_t*, float* v) {  // fp8 tensors are never read back by these kernels
#pragma unroll
  for (int e = 0; e < W; ++e) v[e] = 0.f;
}
template <int W> __device__ __forceinline__ void loadw(const x3_t*, float* v) {  // (see the note at store8(x3_t*))
#pragma unroll
  for (int e = 0; e < W; ++e) v[e] = 0.f;
  __builtin_trap();
}
template <int W> __device__ __forceinline__ void loadw(const h8_t*, float* v) {
#pragma unroll
  for (int e = 0; e < W; ++e) v[e] = 0.f;
  __builtin_trap();
}
template <int W> __device__ __forceinline__ void storew(h8_t*, const float*) { __builtin_trap(); }
template <int W> __device__ __forceinline__ void loadw(const w8_t*, float* v) {
#pragma unroll
  for (int e = 0; e < W; ++e) v[e] = 0.f;
  __builtin_trap();
}
template <int W> __device__ __forceinline__ void storew(w8_t*, const float*) { __builtin_trap(); }
template <int W> __device__ __forceinline__ void storew(x3_t*, const float*) { __builtin_trap(); }
template <int W> __device__ __forceinline__ void storew_x3(bf16_t* p, size_t ld, const float* v) {
  if constexpr (W == 4) store4_x3(p, ld, v[0], v[1], v[2], v[3]);
  else store2_x3(p, ld, v[0], v[1]);
}
template <int W> __device__ __forceinline__ void storew(float* p, const float* v) {
  if constexpr (W == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  else *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
}
template <int W> __device__ __forceinline__ void storew(bf16_t* p, const float* v) {
  if constexpr (W == 4) store4(p, v[0], v[1], v[2], v[3]);
  else store2(p, v[0], v[1]);
}
template <int W> __device__ __forceinline__ void storew(fp8_t* p, const float* v) {
  if constexpr (W == 4) store4(p, v[0], v[1], v[2], v[3]);
  else store2(p, v[0], v[1]);
}

// FAST = bf16 tier (hardware v_exp_f32 based), !FAST = parity tier (accurate expf / division)
template <bool FAST> __device__ __forceinline__ float exp_t(float v) { return FAST ? __expf(v) : expf(v); }
// The fast tier spends exactly two transcendental issues per sigmoid (v_exp_f32 + v_rcp_f32): `1.0f / x` and
// __frcp_rn expand to the ~10-instruction IEEE division sequence, which made the GELU epilogue VALU-bound.
template <bool FAST> __device__ __forceinline__ float sigmoid_t(float v) {
  if (FAST) return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v));
  return 1.0f / (1.0f + expf(-v));
}
template <bool FAST> __device__ __forceinline__ float silu_t(float v) { return v * sigmoid_t<FAST>(v); }
// nn.GELU(approximate="tanh"): 0.5 z (1 + tanh(u)), u = sqrt(2/pi) (z + 0.044715 z^3)
//   == z * sigmoid(2u)  (one exp instead of a tanh)
//   == z / (1 + 2^(z (A z^2 + B))),  B = -log2(e) 2 sqrt(2/pi),  A = 0.044715 B      (fast tier: 5 VALU + 2 transcendental)
constexpr float kGeluK2 = 2.0f * 0.7978845608028654f;
constexpr float kGeluB = -1.4426950408889634f * kGeluK2;
constexpr float kGeluA = kGeluB * 0.044715f;
template <bool FAST> __device__ __forceinline__ float gelu_tanh_t(float z) {
  if (FAST) return z * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * fmaf(z * z, kGeluA, kGeluB)));
  float u2 = kGeluK2 * (z + 0.044715f * z * z * z);
  return z * sigmoid_t<FAST>(u2);
}
// d/dz of the above: s + z s (1 - s) du2/dz
template <bool FAST> __device__ __forceinline__ float gelu_tanh_grad_t(float z) {
  if (FAST) {
    const float z2 = z * z;
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * fmaf(z2, kGeluA, kGeluB)));
    const float t = z * fmaf(z2, 3.0f * 0.044715f * kGeluK2, kGeluK2);
    return fmaf(t, fmaf(-s, s, s), s);
  }
  float u2 = kGeluK2 * (z + 0.044715f * z * z * z);
  float s = sigmoid_t<FAST>(u2);
  float du2 = kGeluK2 * (1.0f + 3.0f * 0.044715f * z * z);
  return s + z * s * (1.0f - s) * du2;
}

// value and derivative together (one exp + one rcp for both)
template <bool FAST> __device__ __forceinline__ void gelu_tanh_both_t(float z, float& g, float& dg) {
  if (FAST) {
    const float z2 = z * z;
    const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * fmaf(z2, kGeluA, kGeluB)));
    const float t = z * fmaf(z2, 3.0f * 0.044715f * kGeluK2, kGeluK2);
    g = z * s;
    dg = fmaf(t, fmaf(-s, s, s), s);
  } else {
    g = gelu_tanh_t<false>(z);
    dg = gelu_tanh_grad_t<false>(z);
  }
}

// The saved GELU derivative as an 8-bit code (option gelu_code): tanh-GELU's derivative lies in [-0.1289, 1.1289]; code c = rne(200 dg) + 26,
// dg' = (c - 26) / 200: absolute error <= 2.5e-3 (bf16 near 1: 2e-3), 0 and 1 -- the saturated ends -- exact.  A tensor of codes is a grid of
// 32 x 32 blocks of 1 KiB, block (y / 32, x / 32) at ((y / 32) * (ld / 32) + x / 32) * 1024; inside a block the GEMM epilogue's lane l = 4 * (row & 15)
// + (col >> 3) holds 16 bytes at 16 l: columns (col & ~7) .. + 7 of row (row & 15), then of row 16 + (row & 15) -- one 1 KiB store / load per wave.
constexpr float kGeluCodeScale = 200.0f, kGeluCodeZero = 26.0f;
__device__ __forceinline__ uint32_t gelu_code4(float a, float b, float c, float d) {  // four derivatives -> four code bytes (a lowest)
  // (v + 2^23: the integer nearest to v, ties to even, sits in the low mantissa bits; 0 <= v <= 252 for every finite z)
  const uint32_t ua = __float_as_uint(fmaf(a, kGeluCodeScale, kGeluCodeZero + 8388608.0f)), ub = __float_as_uint(fmaf(b, kGeluCodeScale, kGeluCodeZero + 8388608.0f));
  const uint32_t uc = __float_as_uint(fmaf(c, kGeluCodeScale, kGeluCodeZero + 8388608.0f)), ud = __float_as_uint(fmaf(d, kGeluCodeScale, kGeluCodeZero + 8388608.0f));
  const uint32_t lo = __builtin_amdgcn_perm(ub, ua, 0x0c0c0400u), hi = __builtin_amdgcn_perm(ud, uc, 0x0c0c0400u);
  return lo | (hi << 16);
}
__device__ __forceinline__ void gelu_decode4(uint32_t w, float* o) {
  constexpr float s = 1.0f / kGeluCodeScale, z = -kGeluCodeZero / kGeluCodeScale;
  o[0] = fmaf((float)(w & 0xffu), s, z);
  o[1] = fmaf((float)((w >> 8) & 0xffu), s, z);
  o[2] = fmaf((float)((w >> 16) & 0xffu), s, z);
  o[3] = fmaf((float)(w >> 24), s, z);
}

static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
// bytes per LOGICAL element (the split-bf16 tier stores two bf16 planes)
static inline size_t elem_size(int prec) {
  return (prec == OSUD_PREC_BF16 || prec == OSUD_PREC_F16) ? 2 : (prec == 2 ? 1 : (prec == OSUD_PREC_F16W8 ? 3 : 4));
}

// wave-level reductions (wave = 64 lanes)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

}  // namespace osud
