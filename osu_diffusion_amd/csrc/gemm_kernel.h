// The GEMM kernel template and its launch-side tile choice (see gemm.h for the design).  Included by one translation unit per
// operand element type (gemm_bf16.hip, gemm_f32.hip, gemm_fp8.hip, gemm_x3.hip) so that the instantiations build in parallel;
// the process-wide host state (CU count, tile-queue counter pool, switches) lives in gemm.hip.  gfx950 only.
#pragma once
#include <stdlib.h>

#include <string>
#include <type_traits>

#include "gemm.h"

namespace osud {

// host state shared by the translation units (defined in gemm.hip)
int gemm_num_cus();
unsigned* gemm_sched_slot();       // next counter set of the dynamic tile queue, nullptr before gemm_sched_init()
bool gemm_dynamic_tiles_wanted();  // osud_set_gemm_dynamic_tiles

namespace {

constexpr int SLAB = 128;  // bytes of K per pipeline stage row

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Fragment reads are inline asm on purpose: hipcc cannot prove that a compiler-visible
// ds_read does not alias the in-flight LDS-DMA of later slabs and would put
// `s_waitcnt vmcnt(0)` in front of every read, serialising load and MFMA.  The asm reads are
// ordered by the counted vmcnt + barrier at the top of each slab and by counted lgkmcnt.
template <int OFF> __device__ __forceinline__ u32x4 ds_read16(uint32_t addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
  return v;
}

template <int OFF> __device__ __forceinline__ f32x4 ds_read16f(uint32_t addr) {
  f32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
  return v;
}
__device__ __forceinline__ void ds_write16(uint32_t addr, const f32x4& v) {
  asm volatile("ds_write_b128 %0, %1" : : "v"(addr), "v"(v) : "memory");
}

template <typename TE> __device__ __forceinline__ void mma(f32x16& acc, const u32x4& a, const u32x4& b);
template <> __device__ __forceinline__ void mma<bf16_t>(f32x16& acc, const u32x4& a, const u32x4& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0,
                                                0, 0);
}
template <> __device__ __forceinline__ void mma<x3_t>(f32x16& acc, const u32x4& a, const u32x4& b) {  // a plane pair: one bf16 MFMA
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0,
                                                0, 0);
}
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_mma;
template <> __device__ __forceinline__ void mma<f16_t>(f32x16& acc, const u32x4& a, const u32x4& b) {  // fp16 tier: the same loop, half operands
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_mma, a), __builtin_bit_cast(f16x8_mma, b), acc, 0, 0, 0);
}
template <> __device__ __forceinline__ void mma<float>(f32x16& acc, const u32x4& a, const u32x4& b) {
  const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
  for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j], bf[j], acc, 0, 0, 0);
}

// fp8 (e4m3) operands: one v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales (E8M0 127) consumes TWO 16-byte chunks per
// lane and operand -- K = 64 per instruction at twice the bf16 rate.  Which 32 of the 64 k-slots a lane half feeds does not matter
// as long as A and B agree, so the fragments of two consecutive sub-steps (chunks 2s+h and 2s+2+h) are simply concatenated.
typedef int i32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void mma_f8(f32x16& acc, const u32x4& a0, const u32x4& a1, const u32x4& b0, const u32x4& b1) {
  i32x8 a, b;
  a[0] = a0[0]; a[1] = a0[1]; a[2] = a0[2]; a[3] = a0[3]; a[4] = a1[0]; a[5] = a1[1]; a[6] = a1[2]; a[7] = a1[3];
  b[0] = b0[0]; b[1] = b0[1]; b[2] = b0[2]; b[3] = b0[3]; b[4] = b1[0]; b[5] = b1[1]; b[6] = b1[2]; b[7] = b1[3];
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}
template <> __device__ __forceinline__ void mma<fp8_t>(f32x16&, const u32x4&, const u32x4&) {}  // fp8 goes through mma_f8
template <> __device__ __forceinline__ void mma<h8_t>(f32x16&, const u32x4&, const u32x4&) {}   // (compute_slab_h8)
template <> __device__ __forceinline__ void mma<w8_t>(f32x16&, const u32x4&, const u32x4&) {}   // (compute_slab_w8)
// fp16 + e4m3 operands (h8_t, common.h): the fp16 hi product, and the two cross terms in ONE e4m3 instruction whose A-side block
// scale 2^-12 (E8M0 115) undoes the 2^12 the lo8 planes carry
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
__device__ __forceinline__ void mma_f16(f32x16& acc, const u32x4& a, const u32x4& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), acc, 0, 0, 0);
}
__device__ __forceinline__ void mma_f8_lo(f32x16& acc, const u32x4& a0, const u32x4& a1, const u32x4& b0, const u32x4& b1) {
  i32x8 a, b;
  a[0] = a0[0]; a[1] = a0[1]; a[2] = a0[2]; a[3] = a0[3]; a[4] = a1[0]; a[5] = a1[1]; a[6] = a1[2]; a[7] = a1[3];
  b[0] = b0[0]; b[1] = b0[1]; b[2] = b0[2]; b[3] = b0[3]; b[4] = b1[0]; b[5] = b1[1]; b[6] = b1[2]; b[7] = b1[3];
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, 0x73737373, 0, 0x7f7f7f7f);
}

#define OSUD_LGKM_WAIT(n)                                  \
  asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory"); \
  __builtin_amdgcn_sched_barrier(0)

// Per-wave register block: (RY x 32) rows of Y by (RX x 32) rows of X -> RY x RX MFMA 32x32 accumulators.
template <int RY, int RX> struct FragSet {
  u32x4 y[RY], x[RX];
};
// ya/xa: this lane's LDS byte address of (first Y / X row of the wave, k-substep s) in the current stage
template <int RY, int RX> __device__ __forceinline__ void read_set(FragSet<RY, RX>& f, uint32_t ya, uint32_t xa) {
  constexpr int SB = SLAB;
  f.y[0] = ds_read16<0>(ya);
  if constexpr (RY >= 2) f.y[1] = ds_read16<32 * SB>(ya);
  if constexpr (RY >= 3) f.y[2] = ds_read16<64 * SB>(ya);
  if constexpr (RY == 4) f.y[3] = ds_read16<96 * SB>(ya);
  f.x[0] = ds_read16<0>(xa);
  f.x[1] = ds_read16<32 * SB>(xa);
  if constexpr (RX >= 3) f.x[2] = ds_read16<64 * SB>(xa);
  if constexpr (RX == 4) f.x[3] = ds_read16<96 * SB>(xa);
}
template <typename TE, int RY, int RX>
__device__ __forceinline__ void mma_set(f32x16 (&acc)[RY][RX], const FragSet<RY, RX>& f) {
#pragma unroll
  for (int i = 0; i < RY; ++i)
#pragma unroll
    for (int j = 0; j < RX; ++j) mma<TE>(acc[i][j], f.x[j], f.y[i]);
}
template <int N> __device__ __forceinline__ void wait_lgkm() {
  if constexpr (N == 3) { OSUD_LGKM_WAIT(3); }
  else if constexpr (N == 4) { OSUD_LGKM_WAIT(4); }
  else if constexpr (N == 5) { OSUD_LGKM_WAIT(5); }
  else if constexpr (N == 6) { OSUD_LGKM_WAIT(6); }
  else if constexpr (N == 8) { OSUD_LGKM_WAIT(8); }
  else { OSUD_LGKM_WAIT(0); }
}
// One 128-byte K slab from the LDS stage at byte offset `so`: 4 sub-steps, reads of sub-step s+1 in
// flight under the MFMAs of sub-step s (LDS returns in order, so lgkmcnt(R) == "all but the newest R").
// experiments (-DOSUD_EXP_MODE=<flags>, one build per variant: run-time flags here cost registers and spill): 64 = fragment reads without the MFMAs, 128 = no
// LDS-DMA (stale stages), 256 = MFMAs on stale fragments (no reads), 512 = neither reads nor MFMAs (LDS-DMA only)
template <typename TE, int RY, int RX>
__device__ __forceinline__ void compute_slab(f32x16 (&acc)[RY][RX], const uint32_t (&ya)[4], const uint32_t (&xa)[4],
                                             uint32_t so) {
  FragSet<RY, RX> f0, f1;
#ifdef OSUD_EXP_MODE
  constexpr int g_exp_flags = OSUD_EXP_MODE;
  if (sizeof(TE) == 2 && (g_exp_flags & 512)) return;  // LDS-DMA only
  if (sizeof(TE) == 2 && (g_exp_flags & (64 | 256))) {
    if (g_exp_flags & 64) {  // reads only
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        read_set<RY, RX>(f0, ya[s2] + so, xa[s2] + so);
        wait_lgkm<0>();
#pragma unroll
        for (int i = 0; i < RY; ++i) asm volatile("" ::"v"(f0.y[i]));
#pragma unroll
        for (int j = 0; j < RX; ++j) asm volatile("" ::"v"(f0.x[j]));
      }
    } else {  // MFMAs only, on whatever the registers hold
#pragma unroll
      for (int j = 0; j < RX; ++j) asm volatile("" : "=v"(f0.x[j]));
#pragma unroll
      for (int i = 0; i < RY; ++i) asm volatile("" : "=v"(f0.y[i]));
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) mma_set<TE, RY, RX>(acc, f0);
    }
    return;
  }
#endif
  if constexpr (sizeof(TE) == 1) {  // fp8: two K = 64 instructions per accumulator block and slab (128 bytes = 128 k)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      read_set<RY, RX>(f0, ya[2 * half] + so, xa[2 * half] + so);
      read_set<RY, RX>(f1, ya[2 * half + 1] + so, xa[2 * half + 1] + so);
      wait_lgkm<0>();
#pragma unroll
      for (int i = 0; i < RY; ++i)
#pragma unroll
        for (int j = 0; j < RX; ++j) mma_f8(acc[i][j], f0.x[j], f1.x[j], f0.y[i], f1.y[i]);
    }
    return;
  }
  read_set<RY, RX>(f0, ya[0] + so, xa[0] + so);
  read_set<RY, RX>(f1, ya[1] + so, xa[1] + so);
  wait_lgkm<RY + RX>();
  mma_set<TE, RY, RX>(acc, f0);
  read_set<RY, RX>(f0, ya[2] + so, xa[2] + so);
  wait_lgkm<RY + RX>();
  mma_set<TE, RY, RX>(acc, f1);
  read_set<RY, RX>(f1, ya[3] + so, xa[3] + so);
  wait_lgkm<RY + RX>();
  mma_set<TE, RY, RX>(acc, f0);
  wait_lgkm<0>();
  mma_set<TE, RY, RX>(acc, f1);
}

// 8 consecutive output elements of row `row` at column x of a TO matrix with logical leading dimension ldo
template <typename TO> __device__ __forceinline__ void store8_out(void* base, size_t row, int ldo, int x, const float (&v)[8]) {
  if constexpr (std::is_same<TO, x3_t>::value) store8_x3(reinterpret_cast<bf16_t*>(base) + row * (size_t)(2 * ldo) + x, (size_t)ldo, v);
  else if constexpr (std::is_same<TO, h8_t>::value) store8_h8<false>(reinterpret_cast<h8_t*>(base) + row * (size_t)ldo, x, v);  // (an activation)
  else if constexpr (std::is_same<TO, w8_t>::value) store8_w8<false>(reinterpret_cast<w8_t*>(base) + row * (size_t)ldo, x, v);  // (an activation)
  else store8(reinterpret_cast<TO*>(base) + row * (size_t)ldo + x, v);
}

// Split-bf16 operands: a stage row holds BOTH planes of 32 logical k -- chunks 0..3 = hi, 4..7 = lo (the LDS-DMA gathers the two
// 64-byte halves of a row from the two planes) -- so a fragment pair is read once and used by all of its products:
//   k-step j (16 k):  acc += X_hi.Y_hi + X_lo.Y_hi + X_hi.Y_lo        (sub-steps: hi_j = j, lo_j = 2 + j)
// 48 MFMAs and 4 fragment sets per slab and wave, against 32 and 4 of the plain bf16 form on the same 64 KiB of operands: the
// slab's MFMA time (2 waves x 48 x 32 cycles per SIMD) now exceeds its LDS-DMA fill (~2800 cycles), so the matrix pipe, not
// operand delivery, bounds the loop.  Three fragment sets live at a time: the lo set of k-step 0 is dead before the lo set of
// k-step 1 is read into its registers (the sched_barrier pins that order).
template <int RY, int RX> __device__ __forceinline__ void mma_cross(f32x16 (&acc)[RY][RX], const FragSet<RY, RX>& fx, const FragSet<RY, RX>& fy) {
#pragma unroll
  for (int i = 0; i < RY; ++i)
#pragma unroll
    for (int j = 0; j < RX; ++j) mma<x3_t>(acc[i][j], fx.x[j], fy.y[i]);
}
template <int RY, int RX>
__device__ __forceinline__ void compute_slab_x3(f32x16 (&acc)[RY][RX], const uint32_t (&ya)[4], const uint32_t (&xa)[4], uint32_t so) {
  FragSet<RY, RX> h0, lo, h1;
  read_set<RY, RX>(h0, ya[0] + so, xa[0] + so);
  read_set<RY, RX>(lo, ya[2] + so, xa[2] + so);
  wait_lgkm<RY + RX>();  // h0 (lo still in flight)
  mma_cross<RY, RX>(acc, h0, h0);
  read_set<RY, RX>(h1, ya[1] + so, xa[1] + so);
  wait_lgkm<RY + RX>();  // lo of k-step 0
  mma_cross<RY, RX>(acc, lo, h0);
  mma_cross<RY, RX>(acc, h0, lo);
  __builtin_amdgcn_sched_barrier(0);  // the lo registers are re-used: their last readers are issued
  read_set<RY, RX>(lo, ya[3] + so, xa[3] + so);
  wait_lgkm<RY + RX>();  // h1
  mma_cross<RY, RX>(acc, h1, h1);
  wait_lgkm<0>();
  mma_cross<RY, RX>(acc, lo, h1);
  mma_cross<RY, RX>(acc, h1, lo);
}

// fp16 + e4m3 operands (h8_t): a 128-byte stage row is one K-blocked group of 32 logical k -- chunks 0..3 the fp16 hi values (sub-steps
// 0, 1), chunks 4, 5 plane P and 6, 7 plane Q (sub-steps 2, 3 hand a lane P | Q of its 16 k) -- 16 fp16 MFMAs and 8 K = 64 e4m3 MFMAs per
// slab and wave: 32 matrix-pipe passes per 32 x 32 x 32 block where the split-bf16 form issues 48.  Three fragment sets live at a time.
template <int RY, int RX>
__device__ __forceinline__ void compute_slab_h8(f32x16 (&acc)[RY][RX], const uint32_t (&ya)[4], const uint32_t (&xa)[4], uint32_t so) {
  FragSet<RY, RX> a, b, c;
  read_set<RY, RX>(a, ya[0] + so, xa[0] + so);
  read_set<RY, RX>(b, ya[1] + so, xa[1] + so);
  wait_lgkm<RY + RX>();  // a (b still in flight)
#pragma unroll
  for (int i = 0; i < RY; ++i)
#pragma unroll
    for (int j = 0; j < RX; ++j) mma_f16(acc[i][j], a.x[j], a.y[i]);
  read_set<RY, RX>(c, ya[2] + so, xa[2] + so);
  wait_lgkm<RY + RX>();  // b
#pragma unroll
  for (int i = 0; i < RY; ++i)
#pragma unroll
    for (int j = 0; j < RX; ++j) mma_f16(acc[i][j], b.x[j], b.y[i]);
  __builtin_amdgcn_sched_barrier(0);  // a's registers are re-used: its readers are issued
  read_set<RY, RX>(a, ya[3] + so, xa[3] + so);
  wait_lgkm<0>();
#pragma unroll
  for (int i = 0; i < RY; ++i)
#pragma unroll
    for (int j = 0; j < RX; ++j) mma_f8_lo(acc[i][j], c.x[j], a.x[j], c.y[i], a.y[i]);
}

// fp16 activations x (fp16 + e4m3 residual) weights (w8_t): the slabs of a row come in threes -- two fp16 slabs (64 k each: the plain
// half-precision loop) and one e4m3 slab (128 k: two K = 64 block-scaled MFMAs per accumulator, activation e4m3(v) against the weight's
// residual, whose 2^12 the X-side block scale 2^-12 undoes).  96 matrix-pipe passes per 32 x 32 block and 128 k (h8_t: 128).
template <int RY, int RX, bool LO>
__device__ __forceinline__ void compute_slab_w8(f32x16 (&acc)[RY][RX], const uint32_t (&ya)[4], const uint32_t (&xa)[4], uint32_t so) {
  FragSet<RY, RX> f0, f1;
  if constexpr (LO) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      read_set<RY, RX>(f0, ya[2 * half] + so, xa[2 * half] + so);
      read_set<RY, RX>(f1, ya[2 * half + 1] + so, xa[2 * half + 1] + so);
      wait_lgkm<0>();
#pragma unroll
      for (int i = 0; i < RY; ++i)
#pragma unroll
        for (int j = 0; j < RX; ++j) mma_f8_lo(acc[i][j], f0.x[j], f1.x[j], f0.y[i], f1.y[i]);
    }
  } else {
    read_set<RY, RX>(f0, ya[0] + so, xa[0] + so);
    read_set<RY, RX>(f1, ya[1] + so, xa[1] + so);
    wait_lgkm<RY + RX>();
#pragma unroll
    for (int i = 0; i < RY; ++i)
#pragma unroll
      for (int j = 0; j < RX; ++j) mma_f16(acc[i][j], f0.x[j], f0.y[i]);
    read_set<RY, RX>(f0, ya[2] + so, xa[2] + so);
    wait_lgkm<RY + RX>();
#pragma unroll
    for (int i = 0; i < RY; ++i)
#pragma unroll
      for (int j = 0; j < RX; ++j) mma_f16(acc[i][j], f1.x[j], f1.y[i]);
    read_set<RY, RX>(f1, ya[3] + so, xa[3] + so);
    wait_lgkm<RY + RX>();
#pragma unroll
    for (int i = 0; i < RY; ++i)
#pragma unroll
      for (int j = 0; j < RX; ++j) mma_f16(acc[i][j], f0.x[j], f0.y[i]);
    wait_lgkm<0>();
#pragma unroll
    for (int i = 0; i < RY; ++i)
#pragma unroll
      for (int j = 0; j < RX; ++j) mma_f16(acc[i][j], f1.x[j], f1.y[i]);
  }
}

// Tile geometry: WY x WX waves, each wave (RY*32) x (RX*32) outputs: BM = WY*RY*32 rows of Y, BN = WX*RX*32 rows of X.
// A stage holds one K slab of both operands as ONE (BM+BN)-row x 128-byte image (two 64 KiB stages for the 256-wide tiles: one slab
// in flight while one is consumed).  (Built, measured slower and removed: 64-byte half slabs in a ring of four with the epilogue
// patches behind the ring -- fc1 183 vs 168 us, the doubled barrier count costs more than the extra lead buys; a role-split main
// loop in which the two waves of a SIMD alternate between MFMAs and LDS-DMA issue -- equal time at a proportionally lower clock.
// HISTORY.md section 4 has both write-ups.)
template <int WY, int WX, int RY, int RX> struct Geo {
  static constexpr int SB = SLAB;
  static constexpr int BM = WY * RY * 32, BN = WX * RX * 32, NW = WY * WX, NT = 64 * NW;
  static constexpr int STAGE = (BM + BN) * SB;
  static constexpr int RPP = 1024 / SB;        // rows per 1 KiB LDS-DMA piece
  // One persistent workgroup per CU owns all 160 KiB.  (Tried and dropped: two 4-wave workgroups per CU on 128x192 tiles so that
  // one's epilogue runs under the other's main loop, 25-45 % slower; four waves of 128x128 with one wave per SIMD and 512
  // registers, 20-60 % slower under hipcc's scheduling.)
  static constexpr int WGS = 1;
  static constexpr int LDS_MAX = 160 * 1024 / WGS;
  static constexpr int RING_MAX = LDS_MAX;
  static constexpr int NSTAGE = STAGE * 5 <= RING_MAX ? 5 : (STAGE * 4 <= RING_MAX ? 4 : (STAGE * 3 <= RING_MAX ? 3 : 2));
  static constexpr int PIECES = STAGE / 1024, PPW = PIECES / NW;  // 1 KiB LDS-DMA pieces per slab, per wave
  static_assert(PIECES % NW == 0 && BM % RPP == 0, "pieces must divide evenly over the waves and not straddle the operands");
  static_assert(NSTAGE * STAGE <= RING_MAX && NW * 4096 <= STAGE, "stage ring must fit the LDS; the epilogue patches live in the free stage");
};

// HBM -> LDS: this wave's share of one K slab (PPW pieces of 8 rows x 128 bytes).  The 16-byte chunk index
// is XOR-swizzled with (row>>1)&7 on the SOURCE side (the LDS side of an LDS-DMA is lane-linear).  The address is split as
// SGPR base (the tile's Y or X panel at this slab) + a per-lane 32-bit offset computed once per kernel (`dma_off`), so the
// loop carries no per-piece 64-bit VALU address arithmetic (-24...-32 VGPRs, -1.5 % per sampling step vs the builtin with
// per-lane 64-bit pointers).  M0 (the LDS destination) is written in the same asm statement that uses it.
template <typename G>
__device__ __forceinline__ void stage_slab(const char* gy, const char* gx, uint32_t stage_lds, const uint32_t (&voff)[G::PPW],
                                           int wave) {
#ifdef OSUD_EXP_MODE
  if (OSUD_EXP_MODE & 128) return;
#endif
#pragma unroll
  for (int q = 0; q < G::PPW; ++q) {
    const int piece = wave * G::PPW + q;  // wave-uniform
    const char* sbase = (piece * G::RPP < G::BM) ? gy : gx;
    const uint32_t dst = stage_lds + (uint32_t)__builtin_amdgcn_readfirstlane(piece * 1024);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff[q]), "s"(sbase), "s"(dst) : "memory");
  }
}

// Linear tile index -> (ty, tx).  Workgroup b runs on XCD b % 8 and walks tiles first, first + G, ...; `first`
// is chosen so that in every round each XCD holds a run of G/8 consecutive indices.  Plain order (x fastest)
// makes every XCD sweep ALL weight panels each round (the PMC pass showed 5x the operand bytes being fetched).
// Banded order (tile_order = 2): XCD k owns the row band [k*nty/8, (k+1)*nty/8) and walks it column-block by column-block
// (CBW tile columns at a time, rows inside), so CBW weight panels stay L2-resident across consecutive rounds
// while the activation panels stream.  Measured on fc1 (M=32768): fetched bytes -22 %, time +2 % (the re-reads are
// Infinity-Cache hits, not the limiter) -> the plain order stays the default.
struct TileMap {
  int ntx, nty, G, banded, rb, cbw;
  __device__ __forceinline__ void coords(int t, int& ty, int& tx) const {
    if (!banded) {
      ty = t / ntx;
      tx = t % ntx;
      return;
    }
    const int per = G >> 3;                       // tiles per XCD per round
    const int round = t / G, in_round = t % G;
    const int xcd = in_round / per, li = round * per + in_round % per;  // index inside the XCD's band
    const int blk = rb * cbw;                     // tiles per column block
    const int cb = li / blk, rem = li % blk;
    ty = xcd * rb + rem / cbw;
    tx = cb * cbw + rem % cbw;
  }
};

template <int N> __device__ __forceinline__ void wait_vm() {
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
  else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory");
  else if constexpr (N == 14) asm volatile("s_waitcnt vmcnt(14) lgkmcnt(0)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)" ::: "memory");
  else if constexpr (N == 20) asm volatile("s_waitcnt vmcnt(20) lgkmcnt(0)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
  else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)" ::: "memory");
  else static_assert(N < 0, "add the literal");
}

// ---- the epilogue of one output tile, shared by the slab loop (gemm_kernel) and the phased loop (gemm_phased.h) -----------------
template <typename TE, int EPI> struct EpiTraits {
  static constexpr bool FAST = !std::is_same<TE, float>::value;
  static constexpr bool kF8 = sizeof(TE) == 1;  // fp8 operands: outputs are bf16 (EPI_BIAS_TE), fp8 (EPI_BIAS_GELU_TE) or fp32
  // fp16 + e4m3 operands (the trunk GEMMs of the tolerance tier): the bias epilogue (in_proj) feeds the split-bf16 attention kernel and
  // writes hi | lo planes; the GELU epilogue (fc1) writes the next GEMM's operand form
  static constexpr bool kW8 = std::is_same<TE, w8_t>::value;  // (fp16 x (fp16 + e4m3) operands: outputs as for h8_t)
  static constexpr bool kH8 = std::is_same<TE, h8_t>::value || kW8;
  using TOalt = typename std::conditional<kW8, h8_t, w8_t>::type;  // (EPI_BIAS_GELU_ALT: the other K-blocked activation form)
  using TO = typename std::conditional<kF8, typename std::conditional<EPI == EPI_BIAS_GELU_TE, fp8_t, bf16_t>::type,
                                       typename std::conditional<kH8 && EPI == EPI_BIAS_TE, x3_t,
                                                                 typename std::conditional<kH8 && EPI == EPI_BIAS_GELU_ALT, TOalt, TE>::type>::type>::type;
  static constexpr bool kGelu = EPI == EPI_BIAS_GELU_TE || EPI == EPI_BIAS_GELU_BF || EPI == EPI_BIAS_GELU_ALT;  // _BF: fp8 operands with bf16 outputs (training)
};

// acc[i][j]: the wave's 32 x 32 block (Y block i, X block j) as the MFMA left it; (ty, tx) the tile, (wy, wx) the wave inside it;
// pw / pr0 / pr1: this lane's write / read-back addresses in the wave's 4 KiB LDS patch, pso the patch area's byte offset;
// q_amax: fp8 training, running max |value| of this lane's share of the e4m3 output
template <typename TE, int EPI, int WY, int WX, int RY, int RX>
__device__ __forceinline__ void tile_epilogue(const GemmP& p, f32x16 (&acc)[RY][RX], const int ty, const int tx, const int wy, const int wx,
                                              const int lane, const uint32_t (&pw)[4], const uint32_t pr0, const uint32_t pr1,
                                              const uint32_t pso, float& q_amax) {
  using ET = EpiTraits<TE, EPI>;
  using TO = typename ET::TO;
  constexpr bool FAST = ET::FAST, kF8 = ET::kF8, kGelu = ET::kGelu;
  constexpr int BM = WY * RY * 32, BN = WX * RX * 32;
  // ---- epilogue -------------------------------------------------------------------------------
  // The MFMA leaves lane (frow, fhalf) with y = frow and x = 8g + 4*fhalf + {0..3}: stored as is, one store
  // instruction touches 32 rows with 8..32 bytes each and the epilogue is bound by the L2 REQUEST rate
  // (measured: ~9 us per 256x256 tile, a third of the kernel).  So every 32x32 block takes a round trip
  // through a wave-private 4 KiB LDS patch (16-byte slot index XOR-swizzled with row&7, conflict free both
  // ways) and comes back row-major: lane l holds rows (l>>2) and 16 + (l>>2) and x = 8*(l&3) + {0..7} -- 4 lanes
  // cover one 128-byte (f32) / 64-byte (bf16) row segment with 16-byte (bf16) / 2 x 16-byte (f32) stores per lane, which
  // halves the number of store instructions of a bf16 output (the TA spends ~16 cycles per vector-memory instruction
  // whatever its width); the operand loads (residual, saved pre-activation) are coalesced the same way.
  // All loads of a block are issued BEFORE its stores: vmcnt counts stores too, so a load waited for
  // between stores would drain every earlier store.
  constexpr bool kBias = EPI == EPI_BIAS_F32 || EPI == EPI_BIAS_TE || EPI == EPI_BIAS_SILU_TE ||
                         kGelu || EPI == EPI_GATE_RES;
  const int lrow = lane >> 2, lcol = 8 * (lane & 3);
  const int xw = tx * BN + wx * RX * 32 + lcol;  // + j*32
  float csv[RX][8];
  if (kF8) {
#pragma unroll
    for (int j = 0; j < RX; ++j) {
#pragma unroll
      for (int e = 0; e < 8; ++e) csv[j][e] = 1.0f;
      if (p.colscale != nullptr) load8(p.colscale + xw + j * 32, csv[j]);
      if (p.act_inv != nullptr || p.act_inv_host != 0.f) {  // the activation's de-quantisation factor: static (host) or dynamic
        const float ai = (p.act_inv != nullptr ? *p.act_inv : 1.0f) * (p.act_inv_host != 0.f ? p.act_inv_host : 1.0f);  // (device: fp8 training)
#pragma unroll
        for (int e = 0; e < 8; ++e) csv[j][e] *= ai;
      }
    }
  }
  float bv[RX][8];
  if (kBias) {
#pragma unroll
    for (int j = 0; j < RX; ++j) {
      const float4 b0 = *reinterpret_cast<const float4*>(p.bias + xw + j * 32);
      const float4 b1 = *reinterpret_cast<const float4*>(p.bias + xw + j * 32 + 4);
      bv[j][0] = b0.x; bv[j][1] = b0.y; bv[j][2] = b0.z; bv[j][3] = b0.w;
      bv[j][4] = b1.x; bv[j][5] = b1.y; bv[j][6] = b1.z; bv[j][7] = b1.w;
    }
  }
  // The blocks are software-pipelined: block b+1 enters the patch (4 writes + 4 reads) before block b is finished, so
  // the LDS round trip hides under block b's arithmetic and stores.  One wave's LDS operations execute in order: the
  // reads of b are done once at most the 8 newer operations are outstanding, and the writes of b+1 cannot overtake them.
  constexpr int NB = RY * RX;
  auto patch_trip = [&](int b, f32x4 (&t)[4]) {
    const int i = b / RX, j = b % RX;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      f32x4 v;
      v[0] = acc[i][j][4 * g + 0]; v[1] = acc[i][j][4 * g + 1]; v[2] = acc[i][j][4 * g + 2]; v[3] = acc[i][j][4 * g + 3];
      ds_write16(pw[g] + pso, v);
    }
    t[0] = ds_read16f<0>(pr0 + pso);      // rows 0..15 : x 0..3 | 4..7 of this lane's 8
    t[1] = ds_read16f<0>(pr1 + pso);
    t[2] = ds_read16f<2048>(pr0 + pso);   // rows 16..31
    t[3] = ds_read16f<2048>(pr1 + pso);
  };
  // fp8 training: the e4m3 twin of a bf16 output (the operand of the NEXT fp8 GEMM) written here instead of by a separate pass
  constexpr bool kOut8 = kF8 && (EPI == EPI_BIAS_GELU_BF || EPI == EPI_GELUGRAD_TE);
  const bool out8_on = kOut8 && p.out8 != nullptr;
  const float q_scale = out8_on ? p.out8_slot[0] : 1.0f;
  constexpr bool kColsum = EPI == EPI_GELUGRAD_TE;
  float cs[kColsum ? RX : 1][8];
  if (kColsum) {
#pragma unroll
    for (int j = 0; j < RX; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) cs[j][e] = 0.f;
  }
  // The saved GELU derivative (EPI_GELUGRAD_TE: 16 bytes per lane and half block) is fetched PF blocks ahead of its use into a small ring
  // of raw registers (the fragments are dead by now) instead of at the top of its own block.  Measured: 24.43 -> 24.39 (PF 2) / 24.36 ms (PF 3,
  // which spills in the slab kernel's 256 x 256 form) per training step -- the launch is 35 us longer than its plain twin because all
  // compute units run their epilogues at the same time and ask HBM for 7 TB/s while they do, not because of the loads' latency.
  constexpr bool kPfAux = EPI == EPI_GELUGRAD_TE && sizeof(TO) == 2;
  constexpr int PF = NB < 2 ? NB : 2;
  uint4 praw[kPfAux ? PF : 1][2];
  // (8-bit block code of the derivative, p.aux_code: block (y / 32, x / 32) is 1 KiB, this lane's 16 bytes at 16 * lane: common.h)
  auto code_block = [&](int b) -> size_t {
    const int yb32 = (ty * BM + wy * RY * 32 + (b / RX) * 32) >> 5, xb32 = (tx * BN + wx * RX * 32 + (b % RX) * 32) >> 5;
    return ((size_t)yb32 * (size_t)(p.ldo >> 5) + (size_t)xb32) * 1024 + (size_t)lane * 16;
  };
  auto aux_issue = [&](int b) {
    if (p.aux_code) {
      praw[b % PF][0] = *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(p.aux) + code_block(b));
      return;
    }
    const size_t row = (size_t)(ty * BM + wy * RY * 32 + (b / RX) * 32 + lrow);
    const int x = xw + (b % RX) * 32;
#pragma unroll
    for (int q = 0; q < 2; ++q)
      praw[b % PF][q] = *reinterpret_cast<const uint4*>(reinterpret_cast<const TO*>(p.aux) + (row + 16 * q) * p.ldo + x);
  };
  if constexpr (kPfAux) {
#pragma unroll
    for (int b = 0; b < PF; ++b) aux_issue(b);
  }
  f32x4 tq[2][4];
  patch_trip(0, tq[0]);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int i = b / RX, j = b % RX;
    const int yb = ty * BM + wy * RY * 32 + i * 32;  // wave-uniform first row of the block
    const int y0 = yb + lrow;                           // + 16q
    int sample = 0;
    if (EPI == EPI_GATE_RES) {  // rows_per_sample % 32 == 0: one sample per 32-row block
      sample = __builtin_amdgcn_readfirstlane(yb) / p.rows_per_sample;
      if (sample >= p.n_samples) sample = p.n_samples - 1;  // padding rows
    }
    f32x4 (&t)[4] = tq[b & 1];
    const int x = xw + j * 32;
    uint32_t dcode[4];  // (8-bit derivative code: the lane's two rows of this block)
    float gv[8], rv[2][8];
    float rb[2];
    if (EPI == EPI_GATE_RES) {
      const float* rsrc = p.res ? p.res : reinterpret_cast<const float*>(p.out);
      load8(p.gate + (size_t)sample * p.ld_gate + x, gv);
#pragma unroll
      for (int q = 0; q < 2; ++q) load8(rsrc + (size_t)(y0 + 16 * q) * p.ldo + x, rv[q]);
    }
    if (EPI == EPI_ACCUM_F32) {
#pragma unroll
      for (int q = 0; q < 2; ++q) load8(reinterpret_cast<const float*>(p.out) + (size_t)(y0 + 16 * q) * p.ldo + x, rv[q]);
    }
    if (EPI == EPI_GELUGRAD_TE && !kPfAux) {
#pragma unroll
      for (int q = 0; q < 2; ++q) load8(reinterpret_cast<const TO*>(p.aux) + (size_t)(y0 + 16 * q) * p.ldo + x, rv[q]);
    }
    if (EPI == EPI_ROWBIAS_TE) {
#pragma unroll
      for (int q = 0; q < 2; ++q) rb[q] = p.bias[y0 + 16 * q];
    }
    if (b + 1 < NB) {
      patch_trip(b + 1, tq[(b + 1) & 1]);
      OSUD_LGKM_WAIT(8);
    } else {
      OSUD_LGKM_WAIT(0);
    }
    if constexpr (kPfAux) {  // this block's rows out of the ring (bf16 pairs / code bytes -> floats), its slot refilled for block b + PF
      if (p.aux_code) {
        const uint4 u = praw[b % PF][0];
        gelu_decode4(u.x, rv[0]); gelu_decode4(u.y, rv[0] + 4);
        gelu_decode4(u.z, rv[1]); gelu_decode4(u.w, rv[1] + 4);
      } else
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const uint4 u = praw[b % PF][q];
        rv[q][0] = __uint_as_float(u.x << 16); rv[q][1] = __uint_as_float(u.x & 0xffff0000u);
        rv[q][2] = __uint_as_float(u.y << 16); rv[q][3] = __uint_as_float(u.y & 0xffff0000u);
        rv[q][4] = __uint_as_float(u.z << 16); rv[q][5] = __uint_as_float(u.z & 0xffff0000u);
        rv[q][6] = __uint_as_float(u.w << 16); rv[q][7] = __uint_as_float(u.w & 0xffff0000u);
      }
      if (b + PF < NB) aux_issue(b + PF);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      float v[8], w[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = t[2 * q][e];
        v[4 + e] = t[2 * q + 1][e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (kF8) v[e] *= csv[j][e];
        if (kBias) v[e] += bv[j][e];
        if (EPI == EPI_ROWBIAS_TE) v[e] += rb[q];
      }
      const size_t orow = (size_t)(y0 + 16 * q);
      const size_t o = orow * p.ldo + x;  // (fp32 outputs and the single-plane TE forms)
      if (EPI == EPI_NONE_F32 && p.seg_rows > 0) {  // (a 32-row block never straddles two segments: seg_rows % 32 == 0)
        const int sg = __builtin_amdgcn_readfirstlane(yb / p.seg_rows);
        store8(p.seg_out[sg] + (size_t)(y0 + 16 * q - sg * p.seg_rows) * p.ldo + x, v);
      } else if (EPI == EPI_BIAS_F32 || EPI == EPI_NONE_F32) {
        store8(reinterpret_cast<float*>(p.out) + o, v);
      } else if (EPI == EPI_ACCUM_F32) {
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = rv[q][e] + v[e];
        store8(reinterpret_cast<float*>(p.out) + o, w);
      } else if (EPI == EPI_GATE_RES) {
        if (p.out2) store8_out<TO>(p.out2, orow, p.ldo, x, v);  // branch output (training)
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = rv[q][e] + gv[e] * v[e];
        store8(reinterpret_cast<float*>(p.out) + o, w);
      } else if (EPI == EPI_BIAS_SILU_TE) {
        if (p.out2) store8_out<TO>(p.out2, orow, p.ldo, x, v);  // pre-activation (training)
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = silu_t<FAST>(v[e]);
        store8_out<TO>(p.out, orow, p.ldo, x, w);
      } else if (kGelu) {
        if (p.out2) {  // training: the DERIVATIVE goes out (same exp/rcp as the value), so that the backward epilogue
                       // is a plain multiply instead of two more quarter-rate transcendentals per element
          float dg[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) gelu_tanh_both_t<FAST>(v[e], w[e], dg[e]);
          // The derivative is read again a whole forward pass later: a streaming (non-temporal) store keeps its 201 MB from displacing
          // the GELU output next to it, which the fc2 forward GEMM reads right away.  Same box, in-step: this launch 184 -> 173 us, the
          // GEMM after it 103 -> 97 us, the training step -0.3 ms.  (Measured and not taken: the same hint on the GELU output itself
          // 180 us; on the backward's loads of the derivative +5 us; on that launch's output +20 us -- profiles/r05_ab_runs.md.)
          if constexpr (std::is_same<TO, bf16_t>::value) {
            typedef uint32_t u4v __attribute__((ext_vector_type(4)));
            if (p.aux_code) {  // 8-bit code: this row's eight bytes wait for the other row of the lane, one 16-byte store per block (below)
              dcode[2 * q] = gelu_code4(dg[0], dg[1], dg[2], dg[3]);
              dcode[2 * q + 1] = gelu_code4(dg[4], dg[5], dg[6], dg[7]);
              if (q == 1) {
                const u4v u = {dcode[0], dcode[1], dcode[2], dcode[3]};
                __builtin_nontemporal_store(u, reinterpret_cast<u4v*>(reinterpret_cast<char*>(p.out2) + code_block(b)));
              }
            } else {
            const u4v u = {pack_bf2(dg[0], dg[1]), pack_bf2(dg[2], dg[3]), pack_bf2(dg[4], dg[5]), pack_bf2(dg[6], dg[7])};
            __builtin_nontemporal_store(u, reinterpret_cast<u4v*>(reinterpret_cast<bf16_t*>(p.out2) + orow * (size_t)p.ldo + x));
            }
          } else {
            store8_out<TO>(p.out2, orow, p.ldo, x, dg);
          }
          if constexpr (kOut8) {
            if (out8_on) {
              float q8[8];
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                q_amax = fmaxf(q_amax, fabsf(w[e]));
                q8[e] = w[e] * q_scale;
              }
              store8(reinterpret_cast<fp8_t*>(p.out8) + o, q8);
            }
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) w[e] = gelu_tanh_t<FAST>(v[e]) * ((kF8 && EPI == EPI_BIAS_GELU_TE) ? p.out_scale : 1.0f);
        }
        if (!kOut8 || p.out != nullptr) store8_out<TO>(p.out, orow, p.ldo, x, w);  // (null: only the e4m3 twin is consumed)
      } else if (EPI == EPI_GELUGRAD_TE) {
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = v[e] * rv[q][e];
        if constexpr (kOut8) {
          if (out8_on) {
            float q8[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              q_amax = fmaxf(q_amax, fabsf(w[e]));
              q8[e] = w[e] * q_scale;
            }
            store8(reinterpret_cast<fp8_t*>(p.out8) + o, q8);
          }
        }
        if (kColsum) {
#pragma unroll
          for (int e = 0; e < 8; ++e) cs[j][e] += w[e];
        }
        if (!kOut8 || p.out != nullptr) store8(reinterpret_cast<TO*>(p.out) + o, w);
      } else {  // EPI_BIAS_TE, EPI_ROWBIAS_TE, EPI_NONE_TE
        store8_out<TO>(p.out, orow, p.ldo, x, v);
      }
    }
  }
  if (kColsum && p.colpart != nullptr) {
    // bias gradient riding along: this wave's RY*32 rows are summed per column -- over the lane's own rows above, over
    // the 16 lane-rows here (fixed butterfly: deterministic) -- and stored as one row of partial sums
#pragma unroll
    for (int j = 0; j < RX; ++j) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = cs[j][e];
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        cs[j][e] = v;
      }
      if (lane < 4) store8(p.colpart + (size_t)(ty * WY + wy) * p.Nx + xw + j * 32, cs[j]);
    }
  }
}

template <typename TE, int EPI, int WY, int WX, int RY, int RX>
__global__ __launch_bounds__((Geo<WY, WX, RY, RX>::NT)) void gemm_kernel(GemmP p) {
  using G = Geo<WY, WX, RY, RX>;
  constexpr int BN = G::BN, SB = SLAB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using ET = EpiTraits<TE, EPI>;
  constexpr bool kW8 = ET::kW8, kH8 = ET::kH8;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wy = wave / WX, wx = wave % WX;
  const int frow = lane & 31, fhalf = lane >> 5;

  const int ntx = p.Nx / BN, ntiles = (p.My / G::BM) * ntx;
  // split-bf16 operands (x3_t): global rows are [hi plane | lo plane], each ld long; a slab is 64 bytes of BOTH planes (32 logical
  // k), gathered into one 128-byte stage row by the LDS-DMA's per-lane source addresses (see compute_slab_x3)
  constexpr bool kX3 = std::is_same<TE, x3_t>::value;
  constexpr int kPlanes = Planes<TE>::k;
  constexpr int kSlabGlobal = kX3 ? SB / 2 : SB;  // bytes a slab advances along a (plane's) row
  int nk = (int)((size_t)p.K * sizeof(TE) / kSlabGlobal);
  const size_t ldy_b = (size_t)p.ldy * sizeof(TE) * kPlanes, ldx_b = (size_t)p.ldx * sizeof(TE) * kPlanes;
  const char* gy0 = reinterpret_cast<const char*>(p.Y);
  const char* gx0 = reinterpret_cast<const char*>(p.X);
  if (p.split_k > 1) {  // this workgroup's share of the contraction (ranges differ by at most one slab)
    const int k0 = (int)((long)blockIdx.y * nk / p.split_k), k1 = (int)((long)(blockIdx.y + 1) * nk / p.split_k);
    nk = k1 - k0;
    gy0 += (size_t)k0 * kSlabGlobal;
    gx0 += (size_t)k0 * kSlabGlobal;
    p.out = reinterpret_cast<char*>(p.out) + (size_t)blockIdx.y * p.split_stride * (EPI == EPI_NONE_F32 ? 4 : sizeof(TE));
  }
  // Persistent workgroups: gridDim.x <= #CUs.  Blocks are dispatched round-robin over the 8 XCDs (b % 8);
  // in every round give each XCD a contiguous run of tiles (x fastest) so its private L2 sees whole panels.
  const int G8 = gridDim.x;
  int first;
  {
    const int b = blockIdx.x, q = G8 >> 3, r = G8 & 7, xcd = b & 7, idx = b >> 3;
    first = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  TileMap tm;
  tm.ntx = ntx; tm.nty = p.My / G::BM; tm.G = G8;
  tm.rb = tm.nty >> 3;
  tm.cbw = ntx % 6 == 0 ? 6 : (ntx % 4 == 0 ? 4 : (ntx % 3 == 0 ? 3 : ntx));
  tm.banded = (p.tile_order == 2) && tm.nty % 8 == 0 && G8 % 8 == 0 && ntiles % G8 == 0 && tm.rb >= 2 && ntx >= 2;
  auto tile_ptrs = [&](int tile, int kt, const char*& gy, const char*& gx) {
    int ty, tx;
    tm.coords(tile, ty, tx);
#ifdef OSUD_GEMM_TIMING
    if (p.tile_order == 4) ty = tx = 0;           // experiment: every workgroup streams the SAME panels (all L2 hits)
    if (p.tile_order == 5) { ty = ty % 8; tx = 0; }
#endif
    gy = gy0 + (size_t)ty * G::BM * ldy_b + (size_t)kt * kSlabGlobal;
    gx = gx0 + (size_t)tx * BN * ldx_b + (size_t)kt * kSlabGlobal;
  };

  // per-lane LDS byte addresses of the wave's first Y/X row for the 4 k-substeps (stage 0)
  const uint32_t lds0 = (uint32_t)(size_t)(lds_void*)smem;
  uint32_t ya[4], xa[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const uint32_t sw = (uint32_t)(((2 * s + fhalf) ^ ((frow >> 1) & 7)) << 4);
    ya[s] = lds0 + (wy * RY * 32 + frow) * SB + sw;
    xa[s] = lds0 + (G::BM + wx * RX * 32 + frow) * SB + sw;
  }

  // epilogue patch (4 KiB per wave, inside whichever stage is free when the epilogue runs): write address per
  // register group g and read address, relative to that stage
  const uint32_t patch = lds0 + wave * 4096;
  uint32_t pw[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) pw[g] = patch + frow * 128 + (((2 * g + fhalf) ^ (frow & 7)) << 4);
  // read-back: lane l takes rows (l>>2) and 16 + (l>>2), 8 consecutive x = two 16-byte slots 2*(l&3), 2*(l&3)+1
  const uint32_t pr0 = patch + (lane >> 2) * 128 + (((2 * (lane & 3)) ^ ((lane >> 2) & 7)) << 4);
  const uint32_t pr1 = patch + (lane >> 2) * 128 + (((2 * (lane & 3) + 1) ^ ((lane >> 2) & 7)) << 4);

  // per-lane byte offsets of this wave's LDS-DMA pieces inside a (Y panel | X panel) slab; < 2^32 is checked by the launcher
  uint32_t dma_off[G::PPW];
#pragma unroll
  for (int q = 0; q < G::PPW; ++q) {
    const int piece = wave * G::PPW + q;
    const int R = piece * G::RPP + (lane >> 3);
    const int c = (lane & 7) ^ ((R >> 1) & 7);
    const bool isy = piece * G::RPP < G::BM;
    // (split-bf16: source chunks 0..3 come from the hi plane, 4..7 from the same columns of the lo plane, ld elements further on)
    const size_t cb = kX3 ? (size_t)(c & 3) * 16 + (c >= 4 ? (size_t)(isy ? p.ldy : p.ldx) * sizeof(TE) : 0) : (size_t)c * 16;
    dma_off[q] = (uint32_t)((isy ? (size_t)R * ldy_b : (size_t)(R - G::BM) * ldx_b) + cb);
  }
  // ---- tile sequence.  The first tile of a workgroup is static (`first`: XCD-contiguous runs).  Launches with more tiles
  // than workgroups draw every later tile from a ticket counter (p.sched), two tiles ahead of the one being computed, so that a
  // workgroup that starts late or runs slowly -- a collective's or an optimizer's kernel holding its compute unit -- takes fewer
  // tiles instead of stretching the launch (measured: 8 held CUs cost a static launch 1.5x).  The ticket for tile j+2 is
  // requested (wave 0, lane 0) at the top of tile j, has returned by the vmcnt(0) before tile j's epilogue, and is published
  // through LDS across the epilogue barrier; the ticket for the second tile is requested before the prologue, is older than
  // every LDS-DMA piece and so has returned after the first slab's wait.
  constexpr bool kDynFits = G::NSTAGE * G::STAGE + 64 <= G::LDS_MAX;
  const bool dyn = kDynFits && p.sched != nullptr;
  volatile __attribute__((address_space(3))) uint32_t* sched_lds =
      reinterpret_cast<volatile __attribute__((address_space(3))) uint32_t*>((lds_void*)smem) + (G::NSTAGE * G::STAGE) / 4;
  const bool ticket_lane = dyn && wave == 0 && lane == 0;
  // One queue per XCD (16-bit fields of p.sched[0..3], two per word; [8] counts finished workgroups): ticket k of XCD x is the
  // k-th tile of the runs a static schedule would give that XCD in rounds 1, 2, ... -- undisturbed, every XCD's L2 keeps seeing
  // the same contiguous panels (one global queue scattered them: the K = 3072 launches got 25 % slower).  Queues are not
  // stolen from: tickets run two tiles ahead of the arithmetic, so a workgroup that finds its queue empty cannot tell a
  // neighbour in trouble from one about to finish (stealing on a snapshot of the counters made undisturbed launches 15-30 %
  // slower); kernels that share the GPU spread over the XCDs round-robin like these workgroups do.
  const int xcd = blockIdx.x & 7, per = G8 >> 3;
  auto id_of = [&](uint32_t word) -> int {
    const uint32_t k = (word >> (16 * (xcd & 1))) & 0xffffu;
    return (int)((uint32_t)G8 * (1u + k / (uint32_t)per) + (uint32_t)(xcd * per) + k % (uint32_t)per);  // >= ntiles: queue empty
  };
  uint32_t tk_start = 0, tk = 0;
  // (compiler-visible atomics, not inline asm: an asm result may be copied to another register before it has returned.  The
  // compiler knows nothing of the asm-issued LDS-DMA pieces, so where it needs the ticket it waits for vmcnt(0) -- both
  // places of use sit right behind a drain of this wave's queue anyway.  gemm.hip is built with LLVM's atomic optimizer off:
  // it rewrites a uniform-address atomic into "one lane + readfirstlane" and reads the result at once)
  auto take_ticket = [&](uint32_t& dst) {
    if (ticket_lane)
      dst = __hip_atomic_fetch_add(p.sched + (xcd >> 1), 1u << (16 * (xcd & 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto resolve = [&](uint32_t word) -> int {
    const int id = id_of(word);
    return id < ntiles ? id : 0x7fffffff;
  };
  take_ticket(tk_start);
  int t_cur = first, t_nxt = dyn ? 0x7fffffff : first + G8;
  int ic_rel = 0, ic_kt = 0;  // issue cursor: which of (t_cur, t_nxt) it is in, and the slab
  int issued = 0, consumed = 0;
  auto issue_next = [&]() {
    const int ic_tile = ic_rel == 0 ? t_cur : t_nxt;
    if (ic_rel < 2 && ic_tile < ntiles) {
      const char *gy, *gx;
      tile_ptrs(ic_tile, ic_kt, gy, gx);
      stage_slab<G>(gy, gx, lds0 + (uint32_t)((issued % G::NSTAGE) * G::STAGE), dma_off, wave);
      ++issued;
      if (++ic_kt == nk) {
        ic_kt = 0;
        ++ic_rel;
      }
    }
  };
#pragma unroll
  for (int i = 0; i < G::NSTAGE - 1; ++i) issue_next();
  // (Measured and not kept, round 4: `s_setprio 1` around every MFMA cluster -- fc1 168.8 vs 163.7 us, step 26.20 / 26.31 vs 26.17 / 26.27 ms
  //  -- and a static priority for the later-dispatched half of the waves -- 26.29 / 26.43: with one barrier per slab the waves run
  //  in lock step and there is nothing for the arbiter to prefer, as the guide says of single-phase loops.)
  int landed = 0;  // slabs known to have landed already (waited for before the previous epilogue)

#ifdef OSUD_GEMM_TIMING
  uint64_t tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const uint64_t tk0 = __builtin_readcyclecounter();
#endif
  float q_amax = 0.f;  // fp8 training: running max |value| of this lane's share of the e4m3 output
  bool first_tile = true;
#ifdef OSUD_PH_TIMING   // (tuning builds, tools/gemm_phase_stamps.py: the whole kernel's shader-clock cycles next to the phased loop's)
  uint64_t ph_t0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ph_t0));
#endif
  while (t_cur < ntiles) {
    int ty, tx;
    tm.coords(t_cur, ty, tx);
    take_ticket(tk);  // for the tile after next
    f32x16 acc[RY][RX];
#pragma unroll
    for (int i = 0; i < RY; ++i)
#pragma unroll
      for (int j = 0; j < RX; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    uint32_t pso;
    int t_nxt2 = t_nxt + G8;
#ifdef OSUD_GEMM_TIMING
    uint64_t te0 = __builtin_readcyclecounter();
#endif
    // one slab: wait for it, barrier, issue a later one, compute.  LO (w8_t operands only): the e4m3 slab of a super-group -- a compile-time
    // property of the call site, so that the two MFMA bodies never share a function (with a run-time switch between them the register
    // allocation of the 256 x 256 geometry spilled 500 bytes per lane and the kernel ran 5x slower)
    auto slab = [&](const int kt, auto LO) {
#ifdef OSUD_GEMM_TIMING
      const uint64_t tt0 = __builtin_readcyclecounter();
#endif
      // slab `consumed` must have landed; up to NSTAGE-2 later slabs may stay in flight (loads return in order)
      // (stores of an epilogue may sit in the queue too; they only make the counted wait conservative)
      const int ahead = issued - consumed - 1;
      if (landed > 0) --landed;
      else if (ahead <= 0 || G::NSTAGE == 2) wait_vm<0>();
      else if (ahead == 1) wait_vm<G::PPW>();
      else wait_vm<(G::NSTAGE > 3 ? 2 * G::PPW : 0)>();
#ifdef OSUD_GEMM_TIMING
      const uint64_t tt1 = __builtin_readcyclecounter();
#endif
      if (dyn && first_tile && kt == 0) {  // the second tile's ticket (older than every piece waited for above)
        if (ticket_lane) sched_lds[1] = (uint32_t)resolve(tk_start);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();  // every wave's share landed; the stage consumed last round is free again
      if (dyn && first_tile && kt == 0) t_nxt = __builtin_amdgcn_readfirstlane((int)sched_lds[1]);
#ifdef OSUD_GEMM_TIMING
      const uint64_t tt2 = __builtin_readcyclecounter();
#endif
      // (Measured with -DOSUD_GEMM_TIMING, ticks calibrated at 1.67 per ns = the clock the chip holds under this load: a slab
      // costs ~2900 cycles against 2 x 1024 cycles of MFMA issue per SIMD (71 % pipe utilisation); the rest is the barrier,
      // the LDS-DMA issue (~90-160 cycles per 1 KiB piece) and the first fragment reads.  Not L2/HBM -- all workgroups
      // streaming the SAME panels run no faster -- and not LDS bandwidth either: with the X operand neither staged nor read
      // (half the DMA, 2/3 of the reads; timing experiment) the slab only drops to ~2700.  Issuing half the waves' pieces mid-slab instead of here changes nothing; touching the
      // lines of the slab 2-4 ahead with one plain load per wave (L2 warm-up) costs 4-7 % end to end.)
      issue_next();
#ifdef OSUD_GEMM_TIMING
      const uint64_t tt3 = __builtin_readcyclecounter();
#endif
      if constexpr (kX3) compute_slab_x3<RY, RX>(acc, ya, xa, (uint32_t)((consumed % G::NSTAGE) * G::STAGE));
      else if constexpr (kW8) compute_slab_w8<RY, RX, decltype(LO)::value>(acc, ya, xa, (uint32_t)((consumed % G::NSTAGE) * G::STAGE));
      else if constexpr (kH8) compute_slab_h8<RY, RX>(acc, ya, xa, (uint32_t)((consumed % G::NSTAGE) * G::STAGE));
      else compute_slab<TE, RY, RX>(acc, ya, xa, (uint32_t)((consumed % G::NSTAGE) * G::STAGE));
#ifdef OSUD_GEMM_TIMING
      {
        const uint64_t tt4 = __builtin_readcyclecounter();
        tsum[0] += tt1 - tt0; tsum[1] += tt2 - tt1; tsum[2] += tt3 - tt2; tsum[3] += tt4 - tt3; tsum[4] += 1;
      }
#endif
      ++consumed;
    };
    if constexpr (kW8) {  // super-groups of three slabs: fp16, fp16, e4m3 (K % 128 == 0 is the launcher's check)
      for (int kt = 0; kt < nk; kt += 3) {
        slab(kt, std::false_type{});
        slab(kt + 1, std::false_type{});
        slab(kt + 2, std::true_type{});
      }
    } else {
      for (int kt = 0; kt < nk; ++kt) slab(kt, std::false_type{});
    }
    // The next tile's first slabs are already in flight: make sure they landed NOW, while no store is
    // queued behind them, so the next tile can start right after the epilogue without draining its stores.
#ifdef OSUD_GEMM_TIMING
    te0 = __builtin_readcyclecounter();
#endif
    wait_vm<0>();
    landed = issued - consumed;
    if (dyn) {  // the ticket requested at the top of this tile has returned: publish it across the epilogue barrier
      if (ticket_lane) sched_lds[0] = (uint32_t)resolve(tk);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // The stage consumed last holds nothing the next tile needs (its prefetch sits in the other stages): it becomes
    // the epilogue patch area, once every wave has finished reading the last slab from it.  The barrier at the top
    // of the next tile's first slab orders the patch reads before the LDS-DMA that refills the stage.
#ifdef OSUD_GEMM_TIMING
    const uint64_t te1 = __builtin_readcyclecounter();
#endif
    __builtin_amdgcn_s_barrier();
    if (dyn) t_nxt2 = __builtin_amdgcn_readfirstlane((int)sched_lds[0]);
#ifdef OSUD_GEMM_TIMING
    const uint64_t te2 = __builtin_readcyclecounter();
    tsum[6] += te1 - te0;  // drain wait before the epilogue
    tsum[7] += te2 - te1;  // epilogue barrier
#endif
    pso = (uint32_t)(((consumed + G::NSTAGE - 1) % G::NSTAGE) * G::STAGE);


    tile_epilogue<TE, EPI, WY, WX, RY, RX>(p, acc, ty, tx, wy, wx, lane, pw, pr0, pr1, pso, q_amax);
#ifdef OSUD_GEMM_TIMING
    tsum[5] += __builtin_readcyclecounter() - te0;  // epilogue (incl. the drain wait and barrier)
#endif
    t_cur = t_nxt;
    t_nxt = t_nxt2;
    if (ic_rel > 0) --ic_rel;
    first_tile = false;
  }  // tile loop
  if (p.out8 != nullptr && p.out8_slot != nullptr) {  // one amax atomic per workgroup (through LDS: the ring is idle by now)
    q_amax = wave_max(q_amax);
    __syncthreads();
    volatile float* red = reinterpret_cast<volatile float*>(smem);
    if (lane == 0) red[wave] = q_amax;
    __syncthreads();
    if (tid == 0) {
      float mx = 0.f;
      for (int w2 = 0; w2 < G::NW; ++w2) mx = fmaxf(mx, red[w2]);
      if (mx > 0.f) atomicMax(reinterpret_cast<unsigned*>(p.out8_slot) + 2, __float_as_uint(mx));
    }
  }
  if (ticket_lane) {  // the last workgroup out re-arms the counters for the next launch that borrows this slot
    const unsigned done = __hip_atomic_fetch_add(p.sched + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done == gridDim.x - 1) {
#pragma unroll
      for (int y = 0; y < 9; ++y) __hip_atomic_store(p.sched + y, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
#ifdef OSUD_PH_TIMING
  if (EPI != EPI_GATE_RES && p.gate != nullptr && lane == 0 && blockIdx.x < 32) {
    uint64_t ph_t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ph_t1));
    float* dbg = const_cast<float*>(p.gate) + (blockIdx.x * 8 + wave) * 8;
    dbg[6] = (float)(ph_t1 - ph_t0);
    dbg[4] = 0.f;
  }
#endif
#ifdef OSUD_GEMM_TIMING
  if (p.gate != nullptr && lane == 0 && blockIdx.x < 16) {
    float* dbg = const_cast<float*>(p.gate) + (blockIdx.x * G::NW + wave) * 8;
    for (int i = 0; i < 6; ++i) dbg[i] = (float)tsum[i];
    dbg[6] = (float)tsum[6];
    dbg[7] = (float)(__builtin_readcyclecounter() - tk0);  // whole kernel, in the same ticks (calibrates ticks per ns)
  }
#endif
}

template <typename TE, int EPI, int WY, int WX, int RY, int RX> int launch_w(const GemmP& p_in, hipStream_t st) {
  using G = Geo<WY, WX, RY, RX>;
  GemmP p = p_in;
  const size_t ring = (size_t)G::NSTAGE * G::STAGE;  // stage ring (the epilogue patches borrow the free stage)
  constexpr bool kDynFits = G::NSTAGE * G::STAGE + 64 <= G::LDS_MAX;
  const size_t lds = ring + (kDynFits ? 64 : 0);
  static bool attr_set = false;
  if (!attr_set) {
    OSUD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<TE, EPI, WY, WX, RY, RX>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  if (p.colpart_rows != nullptr) *p.colpart_rows = p.My / (RY * 32);
  const int ntiles = (p.My / G::BM) * (p.Nx / G::BN), splits = p.split_k > 1 ? p.split_k : 1;
  int grid = gemm_num_cus() * G::WGS / splits;  // persistent workgroups: WGS per CU (LDS-limited), shared by the K splits
  if (grid < 1) grid = 1;
  if (grid > ntiles) grid = ntiles;
  {
    const bool dyn_on = gemm_dynamic_tiles_wanted();
    const int nk = (int)((size_t)p.K * sizeof(TE) / (std::is_same<TE, x3_t>::value ? SLAB / 2 : SLAB));
    p.sched = (dyn_on && kDynFits && splits == 1 && ntiles > grid && grid % 8 == 0 && nk >= G::NSTAGE && ntiles / 8 + 2 * grid < 60000) ? gemm_sched_slot() : nullptr;
  }
  hipLaunchKernelGGL((gemm_kernel<TE, EPI, WY, WX, RY, RX>), dim3(grid, splits), dim3(G::NT), lds, st, p);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

// Tile choice.  Geometries (all 8 waves = 2 per SIMD, except the 128x128 fallback):
//   256x256: 2x4 waves of 128x64   least LDS traffic per MFMA; needs Nx % 256 == 0
//   256x192: 4x2 waves of 64x96    for Nx = 768-like widths (3 x 256 would leave 1.5 rounds of tiles)
//   192x256: 2x4 waves of 96x64    the same for My = 768-like heights (V^T projection, weight gradients)
//   128x128: 2x2 waves of 64x64    small problems
//    64x128: 2x2 waves of 32x64    problems with fewer 128x128 tiles than 3/4 of the CUs
// Pick by the fraction of CU-rounds doing useful work, preferring the larger tile on ties.
template <typename TE, int EPI> int launch_phased_or(const GemmP& p, int pick, hipStream_t st, bool& taken);  // gemm_phased.h
template <typename TE, int EPI> int launch_t(const GemmP& p, hipStream_t st) {
  const int cus = gemm_num_cus(), splits = p.split_k > 1 ? p.split_k : 1;
  auto eff = [&](int bm, int bn) -> double {
    if (p.My % bm || p.Nx % bn) return 0.0;
    const long tiles = (long)(p.My / bm) * (p.Nx / bn) * splits;
    const long rounds = (tiles + cus - 1) / cus;
    return (double)tiles / (double)(rounds * cus);
  };
  const double e[4] = {eff(128, 128), eff(256, 192), eff(256, 256), eff(192, 256)};
  int pick = 0;
  if (e[1] > 0 && e[1] + 0.05 >= e[0]) pick = 1;
  if (e[3] > 0 && e[3] + 0.05 >= e[pick] && pick == 0) pick = 3;
  if (e[2] > 0 && e[2] + 0.05 >= e[pick]) pick = 2;
  // few tiles (one beatmap, few variants: M ~ 2-4 K tokens): halve the tile height to double the workgroups in flight
  if (pick == 0 && (long)(p.My / 128) * (p.Nx / 128) * splits * 100 < (long)cus * 75) pick = 6;
  // one row of 128-row tiles and a wide output (the adaLN product of a sampling step: 128 x 56 832 x 768 streams 87 MB of weights):
  // 128x256 tiles make it one round instead of 1.7 (28.1 -> 23.9 us)
  if (pick == 0 && p.My == 128 && p.Nx % 256 == 0 && splits == 1 && (long)(p.Nx / 128) > cus) pick = 8;
  if (const int force = opt(OPT_GEMM_TILE)) {  // osud_set_option("gemm_tile", ..): tests of every geometry on one shape, tuning runs
    if (force == 128) pick = 0;
    else if (force == 64) pick = 6;
    else if (force == 192 && e[1] > 0) pick = 1;
    else if (force == 256 && e[2] > 0) pick = 2;
    else if (force == 1192 && e[3] > 0) pick = 3;
    else if (force == 1256 && p.Nx % 256 == 0) pick = 8;
  }
  {  // the phased main loop (gemm_phased.h) where it exists; option gemm_loop = 0 keeps every launch on the slab loop
    bool taken = false;
    const int rc = launch_phased_or<TE, EPI>(p, pick, st, taken);
    if (taken) return rc;
  }
  if (pick == 2) return launch_w<TE, EPI, 2, 4, 4, 2>(p, st);
  if (pick == 1) return launch_w<TE, EPI, 4, 2, 2, 3>(p, st);
  if (pick == 3) return launch_w<TE, EPI, 2, 4, 3, 2>(p, st);
  if (pick == 6) return launch_w<TE, EPI, 2, 2, 1, 2>(p, st);
  if (pick == 8) return launch_w<TE, EPI, 2, 4, 2, 2>(p, st);
  return launch_w<TE, EPI, 2, 2, 2, 2>(p, st);
}

}  // namespace
}  // namespace osud

#include "gemm_phased.h"
