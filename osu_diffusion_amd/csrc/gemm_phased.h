// The phased main loop for the two 256-row tile geometries: the same product, the same LDS-DMA staging, the same fragment reads,
// MFMAs and epilogue as gemm_kernel (gemm_kernel.h) -- every accumulator sees its k in the same order, so the results are
// bit-identical -- in a different SCHEDULE.  Operands: bf16 / fp16 (training, the bf16 and fp16 sampling tiers), fp16 + e4m3 rows
// (h8_t: the sampling tier inside the tolerance) and e4m3 (fp8 training).  gfx950 only.
//
// gemm_kernel's loop is one barrier per 128-byte K slab: all eight waves wait for the slab, all issue their eight LDS-DMA pieces
// of the next one (the vector-memory path takes a piece per ~16 cycles and CU; the issuing wave issues nothing else meanwhile),
// then all run the slab's MFMAs with the fragment reads interleaved.  Both waves of a SIMD are in the same segment at the same
// time, so the matrix pipe idles through every DMA burst: ~750 cycles per 8 MFMAs and wave pair where the pipe needs 512.
//
// Here a K slab is cut into NPH phases.  A phase of one wave is
//     [fragment reads of the phase's cluster | this phase's share of LDS-DMA pieces | counted vmcnt]  s_barrier
//     [lgkmcnt(0) | s_setprio 1 | a cluster of MFMAs worth ~256 pipe cycles | s_setprio 0]            s_barrier
// and the two waves of every SIMD (wave w and w + 4: group 0 and group 1) run ONE barrier apart: while a group-0 wave issues its
// cluster, its group-1 partner on the same SIMD reads fragments and issues DMA, and the other way round in the next interval.
// The matrix pipe always has one wave feeding it, and no wave ever issues an MFMA behind a DMA burst of its own.
// Measured (profiles/r05_gemm_phase_stamps.md: s_memtime stamps, kernel totals): 590-690 cycles per phase = 295-345 per interval
// against 256 of MFMA issue, 20 % fewer cycles per launch than the slab loop on every training shape -- and a shader clock that
// falls by 8-18 % in exchange (the chip runs these kernels at its power limit: 1.3-1.4 GHz on random operands, 1.9-2.0 on zeros
// with the SAME cycle counts), so 7-13 % less time stand-alone and 3-16 % per kernel inside a training step.
//
// Staging is a stream of 1 KiB pieces in the order the phases need them (the LDS image of a slab is laid out in that order:
// PhGeo::src_row), two slab buffers, AHEAD pieces per wave in flight in front of the consumer, cnt[p] more issued per phase.
// Loads return in order, so `vmcnt(W[p])` before the first barrier of phase p retires exactly what phase p + 1 reads; with plain
// operands it is never 0 in steady state (8 pieces = a whole slab stay in flight).  The rules a table is checked against at
// compile time (phased.h: sched_ok):
//   RAW  a piece is read one phase AFTER the phase whose counted wait (+ barrier, which every wave passes after its own wait)
//        retires it -- one barrier more than lock step needs, because group 1 waits one barrier later than group 0;
//   WAR  a region of a buffer is restaged at the earliest two phases after the phase that read it.
// Plain operands: a cluster is one quadrant of the wave's block grid over the slab's four k sub-steps, so each phase needs only the
// rows of its quadrant and a region is free again two phases later: a slab and a half in flight.  K-blocked rows (h8_t: fp16 | e4m3
// planes of 32 k per 128-byte row; e4m3: 128 k per row) put every k sub-step of a row into the same piece, every phase touches most
// rows, and a buffer is refilled only while the other one is consumed: all of a slab's pieces are issued in its predecessor's first
// two phases and waited for (vmcnt(0)) in its last -- the overlap of the two wave groups is the same, the lead is half.
// Tiles: persistent workgroups, static XCD-contiguous order as in gemm_kernel; the stream runs across tile boundaries, everything in
// flight is waited for once before the epilogue's stores join the queue, and waits of a tile's first slab that only cover pieces
// issued before that drain are skipped.  The stagger is per tile: group 1 enters a tile with one extra barrier, group 0 pays it
// back after its epilogue; the epilogues of both groups run side by side (patches live behind the ring: no barrier inside).
#pragma once
// (included at the end of gemm_kernel.h)
#include "phased.h"

namespace osud {
namespace {

// ---- geometry: wave grid, LDS image of a slab in need order, fragment offsets ----------------------------------------------------
template <int GEO> struct PhGeo;
// 256 x 256: 2 x 4 waves of 128 x 64 (4 x 2 accumulator blocks).  LDS rows of a slab: Ya | X0 | X1 | Yb, 128 rows each (Ya / Yb =
// Y blocks 0, 1 / 2, 3 of both wave rows; Xj = X block j of the four wave columns): slots 0-1 | 2-3 | 4-5 | 6-7.
template <> struct PhGeo<0> {
  static constexpr int WY = 2, WX = 4, RY = 4, RX = 2, BN = 256, PPW = 8;
  static constexpr bool is_y(int m) { return m < 2 || m >= 6; }
  static constexpr int Y_OFF[RY] = {0, 32 * SLAB, 384 * SLAB, 416 * SLAB};  // byte offsets of the wave's Y blocks from block 0
  static constexpr int X_OFF[RX] = {0, 128 * SLAB};
  __device__ static int y_row0(int wy) { return wy * 64; }
  __device__ static int x_row0(int wx) { return 128 + wx * 32; }
  __device__ static int src_row(int r, bool& isy) {  // LDS row r of a slab -> row inside the tile's Y (isy) or X panel
    const int q = r >> 7, t = r & 127;
    isy = q == 0 || q == 3;
    if (q == 0) return (t >> 6) * 128 + (t & 63);
    if (q == 3) return (t >> 6) * 128 + 64 + (t & 63);
    return (t >> 5) * 64 + (q == 2 ? 32 : 0) + (t & 31);
  }
};
// 256 x 192: 4 x 2 waves of 64 x 96 (2 x 3 blocks).  LDS rows: Y (256: slots 0-3) | X0 | X1 | X2 (64 rows each: block j of wave
// column 0, then of wave column 1: slots 4 | 5 | 6).
template <> struct PhGeo<1> {
  static constexpr int WY = 4, WX = 2, RY = 2, RX = 3, BN = 192, PPW = 7;
  static constexpr bool is_y(int m) { return m < 4; }
  static constexpr int Y_OFF[RY] = {0, 32 * SLAB};
  static constexpr int X_OFF[RX] = {0, 64 * SLAB, 128 * SLAB};
  __device__ static int y_row0(int wy) { return wy * 64; }
  __device__ static int x_row0(int wx) { return 256 + wx * 32; }
  __device__ static int src_row(int r, bool& isy) {
    isy = r < 256;
    if (isy) return r;
    const int t = r - 256;
    return ((t >> 5) & 1) * 96 + (t >> 6) * 32 + (t & 31);
  }
};

// ---- phase tables (phased.h has the field meanings) -----------------------------------------------------------------------------
template <int GEO, bool BLK> struct PhSched;
// plain operands, 256 x 256: phases (Ya, X0) (Ya, X1) (Yb, X1) (Yb, X0) over the four k sub-steps; the Yb fragments take Ya's registers.
// (Tuning builds measured {0, 2, 2, 4} / AHEAD 14 and {1, 2, 2, 3} / 13 -- nothing beside the 12 reads of phase 0 -- within +-1 %.)
template <> struct PhSched<0, false> : PhGeo<0> {
  static constexpr int NPH = 4, AHEAD = 12;
  static constexpr int cnt[NPH] = {2, 2, 2, 2};
  static constexpr int need[NPH] = {3, 5, 7, -1};
  static constexpr int read_phase[PPW] = {0, 0, 0, 0, 1, 1, 2, 2};
};
// plain operands, 256 x 192: phases (Y, X0) (Y, X1) (Y, X2); all of Y is read in phase 0, so its five slots can only be restaged in phase 2
template <> struct PhSched<1, false> : PhGeo<1> {
  static constexpr int NPH = 3, AHEAD = 12;
  static constexpr int cnt[NPH] = {1, 1, 5};
  static constexpr int need[NPH] = {4, 5, 6};
  static constexpr int read_phase[PPW] = {0, 0, 0, 0, 0, 1, 2};
};
// K-blocked rows, 256 x 256.  h8_t: phases (fp16 k 0-15, all blocks) (fp16 k 16-31, all blocks) (e4m3 planes, Ya x X) (e4m3, Yb x X);
// e4m3: (k 0-63, Ya x X) (k 0-63, Yb x X) (k 64-127, Ya x X) (k 64-127, Yb x X).  X fragments of the last cluster pair stay in registers.
template <> struct PhSched<0, true> : PhGeo<0> {
  static constexpr int NPH = 4, AHEAD = 8;
  static constexpr int cnt[NPH] = {6, 2, 0, 0};
  static constexpr int need[NPH] = {7, 7, 5, 7};
  static constexpr int read_phase[PPW] = {2, 2, 2, 2, 2, 2, 3, 3};
};
// K-blocked rows, 256 x 192.  h8_t: (fp16 k 0-15) (fp16 k 16-31) (e4m3, Y x X0, X1) (e4m3, Y x X2); e4m3: (k 0-63, Y x X0, X1) (k 0-63, Y x X2)
// (k 64-127, Y x X0, X1) (k 64-127, Y x X2): the Y fragments of a half stay in registers for its second cluster
template <> struct PhSched<1, true> : PhGeo<1> {
  static constexpr int NPH = 4, AHEAD = 7;
  static constexpr int cnt[NPH] = {6, 1, 0, 0};
  static constexpr int need[NPH] = {6, 6, 6, 6};
  static constexpr int read_phase[PPW] = {2, 2, 2, 2, 2, 2, 3};
};

// timing builds (tools/build_gemm_variants.sh, tools/gemm_phase_stamps.py): -DOSUD_PH_TIMING=1 stamps the kernel and its epilogues
// (the main loop runs unperturbed), =2 every segment of every phase.  -DOSUD_PH_EXP=<bits> leaves parts out (results meaningless):
// 1 no LDS-DMA after the prologue, 2 no counted waits, 16 no fragment reads, 32 no MFMAs.
#ifndef OSUD_PH_EXP
#define OSUD_PH_EXP 0
#endif
#if defined(OSUD_PH_TIMING) && OSUD_PH_TIMING >= 2
#define OSUD_PH_STAMP(x) asm volatile("s_memtime %0" : "=s"(x))
#define OSUD_PH_SEGMENTS 1
#else
#define OSUD_PH_STAMP(x)
#endif

template <typename TE, int EPI, int GEO> __global__ __launch_bounds__(512) void gemm_phased_kernel(GemmP p) {
  constexpr bool kPlain = sizeof(TE) == 2 && Planes<TE>::k == 1;
  constexpr bool kH8 = std::is_same<TE, h8_t>::value, kF8 = sizeof(TE) == 1;
  static_assert(kPlain || kH8 || kF8, "operand forms of the phased loop: bf16, fp16, fp16 + e4m3 rows, e4m3");
  using S = PhSched<GEO, !kPlain>;
  static_assert(sched_ok<S>(), "phase table breaks a staging rule");
  constexpr int WY = S::WY, WX = S::WX, RY = S::RY, RX = S::RX, BM = 256, BN = S::BN, NPH = S::NPH, PPW = S::PPW;
  constexpr int STAGE = (BM + BN) * SLAB, RING = 2 * STAGE;
  static_assert(WY * RY * 32 == BM && WX * RX * 32 == BN && STAGE == 8 * PPW * 1024 && RING + 8 * 4096 <= 160 * 1024, "geometry");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wy = wave / WX, wx = wave % WX;
  const int grp = wave >> 2;  // waves w and w + 4 share a SIMD: one of each group
  const int frow = lane & 31, fhalf = lane >> 5;

  const int ntx = p.Nx / BN, ntiles = (p.My / BM) * ntx;
  const int nk = (int)((size_t)p.K * sizeof(TE) / SLAB);
  const size_t ldy_b = (size_t)p.ldy * sizeof(TE), ldx_b = (size_t)p.ldx * sizeof(TE);
  const char* const gy0 = reinterpret_cast<const char*>(p.Y);
  const char* const gx0 = reinterpret_cast<const char*>(p.X);
  const int G8 = gridDim.x;
  int first;  // workgroup b runs on XCD b % 8: every XCD walks a contiguous run of tiles per round (as gemm_kernel)
  {
    const int b = blockIdx.x, q = G8 >> 3, r = G8 & 7, xcd = b & 7, idx = b >> 3;
    first = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const uint32_t lds0 = (uint32_t)(size_t)(lds_void*)smem;

  // fragment read addresses (buffer 0): the wave's first Y / X block, k sub-step s; 16-byte chunks swizzled by (row >> 1) & 7
  uint32_t ya[4], xa[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const uint32_t sw = (uint32_t)(((2 * s + fhalf) ^ ((frow >> 1) & 7)) << 4);
    ya[s] = lds0 + (uint32_t)((S::y_row0(wy) + frow) * SLAB) + sw;
    xa[s] = lds0 + (uint32_t)((S::x_row0(wx) + frow) * SLAB) + sw;
  }
  // epilogue patches: 4 KiB per wave behind the ring (same addressing as gemm_kernel)
  const uint32_t patch = lds0 + RING + wave * 4096;
  uint32_t pw[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) pw[g] = patch + frow * 128 + (((2 * g + fhalf) ^ (frow & 7)) << 4);
  const uint32_t pr0 = patch + (lane >> 2) * 128 + (((2 * (lane & 3)) ^ ((lane >> 2) & 7)) << 4);
  const uint32_t pr1 = patch + (lane >> 2) * 128 + (((2 * (lane & 3) + 1) ^ ((lane >> 2) & 7)) << 4);

  // LDS-DMA: wave w owns pieces 8 m + w of a slab (m = 0 .. PPW - 1: its slot m), 8 LDS rows of 128 bytes each; per-lane source
  // offsets inside the tile's (Y | X) panel, the chunk swizzle on the source side
  uint32_t voff[PPW];
#pragma unroll
  for (int m = 0; m < PPW; ++m) {
    const int r = (8 * m + wave) * 8 + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);
    bool isy;
    const int gr = S::src_row(r, isy);
    voff[m] = (uint32_t)((size_t)gr * (isy ? ldy_b : ldx_b) + (size_t)c * 16);
  }

  // ---- the staging stream: a cursor over (tile, slab, slot) in issue order ----------------------------------------------------
  int c_tile = first, c_kt = 0;
  uint32_t c_buf = 0;  // byte offset of the cursor's slab buffer
  bool c_live = true;  // (grid <= ntiles: every workgroup has a first tile)
  const char *c_gy, *c_gx;
  auto tile_base = [&](int t, const char*& gy, const char*& gx) {
    const int ty = t / ntx, tx = t - ty * ntx;
    gy = gy0 + (size_t)ty * BM * ldy_b;
    gx = gx0 + (size_t)tx * BN * ldx_b;
  };
  tile_base(c_tile, c_gy, c_gx);
  auto stage_slot = [&](auto M) {  // the cursor's slab, this wave's slot M
    constexpr int m = decltype(M)::value;
    const char* sb = S::is_y(m) ? c_gy : c_gx;
    const uint32_t dst = lds0 + c_buf + (uint32_t)((8 * m + wave) * 1024);
    const uint32_t vo = voff[m];  // (named outside the asm statement: an asm operand alone does not capture in a generic lambda)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(vo), "s"(sb), "s"(dst) : "memory");
  };
  auto advance = [&]() {
    c_buf = STAGE - c_buf;
    c_gy += SLAB;
    c_gx += SLAB;
    if (++c_kt == nk) {
      c_kt = 0;
      c_tile += G8;
      c_live = c_tile < ntiles;
      if (c_live) tile_base(c_tile, c_gy, c_gx);
    }
  };
  // stream positions [P0, P0 + N): slot = position % PPW, the cursor moves on behind a slab's last slot
  auto stage_run = [&](auto P0, auto N) {
    static_for<decltype(N)::value>([&](auto I) {
      constexpr int m = (decltype(P0)::value + decltype(I)::value) % PPW;
      if (c_live) stage_slot(std::integral_constant<int, m>{});
      if constexpr (m == PPW - 1) {
        if (c_live) advance();
      }
    });
  };

  // prologue: AHEAD pieces, all landed, everywhere
  stage_run(std::integral_constant<int, 0>{}, std::integral_constant<int, S::AHEAD>{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

#ifdef OSUD_PH_TIMING
  // cycle stamps (shader clock, s_memtime) of this wave: [0] read + DMA issue + counted wait, [1] first barrier, [2] lgkmcnt + cluster issue,
  // [3] second barrier, [4] phases, [5] epilogue (with its drain), [6] whole kernel.  A stamp is an SMEM load: it is consumed only behind
  // the phase's own lgkmcnt(0), so the stamps add no wait of their own.  Written to p.gate (floats) for the first 32 workgroups.
  uint64_t tsum[7] = {0, 0, 0, 0, 0, 0, 0};
  uint64_t ts_k0, ts_d = 0, ts_e = 0, ts_c0 = 0;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_k0));
#endif
  uint32_t r_buf = 0;  // the consumer's slab buffer
  float q_amax = 0.f;  // fp8 training: running max |value| of this lane's share of the e4m3 output
  for (int t_cur = first; t_cur < ntiles; t_cur += G8) {
    const int ty = t_cur / ntx, tx = t_cur - ty * ntx;
    f32x16 acc[RY][RX];
#pragma unroll
    for (int i = 0; i < RY; ++i)
#pragma unroll
      for (int j = 0; j < RX; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // fragments.  Plain: fy[i][s] two Y blocks x 4 sub-steps, fx[j][s] X block(s) x 4 sub-steps.  K-blocked: fy[i][h] up to four Y blocks x
    // two chunk sets, fx[j][h] up to three X blocks x two chunk sets.
    u32x4 fy[4][4], fx[3][4];
    if constexpr ((OSUD_PH_EXP & 16) != 0) {
#pragma unroll
      for (int s = 0; s < 4; ++s) asm volatile("" : "=v"(fy[0][s]), "=v"(fy[1][s]), "=v"(fx[0][s]), "=v"(fx[1][s]));
    }
    if (grp == 1) __builtin_amdgcn_s_barrier();  // the stagger

    for (int k = 0; k < nk; ++k) {
      const bool first_slab = k == 0, last = k == nk - 1;
      static_for<NPH>([&](auto PH) {
        constexpr int P = decltype(PH)::value;
#ifdef OSUD_PH_SEGMENTS
        uint64_t ts_a, ts_b, ts_c;
        OSUD_PH_STAMP(ts_a);
#endif
        // ---- the fragment reads of this phase's cluster
        if constexpr ((OSUD_PH_EXP & 16) != 0) {
        } else if constexpr (kPlain && GEO == 0) {
          if constexpr (P == 0 || P == 2) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              fy[0][s] = ds_read16<S::Y_OFF[P]>(ya[s] + r_buf);
              fy[1][s] = ds_read16<S::Y_OFF[P + 1]>(ya[s] + r_buf);
            }
          }
          if constexpr (P == 0 || P == 1) {
#pragma unroll
            for (int s = 0; s < 4; ++s) fx[P][s] = ds_read16<S::X_OFF[P]>(xa[s] + r_buf);
          }
        } else if constexpr (kPlain) {
          if constexpr (P == 0) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              fy[0][s] = ds_read16<S::Y_OFF[0]>(ya[s] + r_buf);
              fy[1][s] = ds_read16<S::Y_OFF[1]>(ya[s] + r_buf);
            }
          }
#pragma unroll
          for (int s = 0; s < 4; ++s) fx[0][s] = ds_read16<S::X_OFF[P]>(xa[s] + r_buf);
        } else if constexpr (kH8 && P < 2) {  // fp16 values of k sub-step P: every block
          static_for<RY>([&](auto I) { fy[decltype(I)::value][0] = ds_read16<S::Y_OFF[decltype(I)::value]>(ya[P] + r_buf); });
          static_for<RX>([&](auto J) { fx[decltype(J)::value][0] = ds_read16<S::X_OFF[decltype(J)::value]>(xa[P] + r_buf); });
        } else {  // a K = 64 block-scaled cluster: chunk sets (c0, c0 + 1) = sub-steps (2, 3) for h8_t, (0, 1) / (2, 3) for the e4m3 halves
          constexpr int c0 = kH8 ? 2 : (P < 2 ? 0 : 2);
          constexpr bool second = (P & 1) != 0;  // the second cluster of a pair: GEO 0 the other two Y blocks, GEO 1 the third X block
          if constexpr (GEO == 0) {
            constexpr int ib = second ? 2 : 0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              fy[0][h] = ds_read16<S::Y_OFF[ib]>(ya[c0 + h] + r_buf);
              fy[1][h] = ds_read16<S::Y_OFF[ib + 1]>(ya[c0 + h] + r_buf);
            }
            if constexpr (!second) {
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                fx[0][h] = ds_read16<S::X_OFF[0]>(xa[c0 + h] + r_buf);
                fx[1][h] = ds_read16<S::X_OFF[1]>(xa[c0 + h] + r_buf);
              }
            }
          } else {
            if constexpr (!second) {
#pragma unroll
              for (int h = 0; h < 2; ++h) {
                fy[0][h] = ds_read16<S::Y_OFF[0]>(ya[c0 + h] + r_buf);
                fy[1][h] = ds_read16<S::Y_OFF[1]>(ya[c0 + h] + r_buf);
                fx[0][h] = ds_read16<S::X_OFF[0]>(xa[c0 + h] + r_buf);
                fx[1][h] = ds_read16<S::X_OFF[1]>(xa[c0 + h] + r_buf);
              }
            } else {
#pragma unroll
              for (int h = 0; h < 2; ++h) fx[2][h] = ds_read16<S::X_OFF[2]>(xa[c0 + h] + r_buf);
            }
          }
        }
        // ---- this phase's share of the stream, and the counted wait for what the NEXT phase reads
        if constexpr ((OSUD_PH_EXP & 1) == 0)
          stage_run(std::integral_constant<int, S::AHEAD + ph_issued_before<S>(P)>{}, std::integral_constant<int, S::cnt[P]>{});
        constexpr int W = ph_wait<S>(P);
        if constexpr (W >= 0) {
          if (!c_live) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stream has ended: nothing younger to count on
          else if ((!first_slab || !ph_wait_predrained<S>(P)) && (OSUD_PH_EXP & 2) == 0) wait_vmcnt<W>();
        }
        __builtin_amdgcn_sched_barrier(0);
        OSUD_PH_STAMP(ts_b);
        __builtin_amdgcn_s_barrier();
        OSUD_PH_STAMP(ts_c);
        OSUD_LGKM_WAIT(0);
#ifdef OSUD_PH_SEGMENTS
        tsum[0] += ts_b - ts_a; tsum[1] += ts_c - ts_b;
        if (ts_d != 0) { tsum[2] += ts_d - ts_c0; tsum[3] += ts_e - ts_d; }
        ts_c0 = ts_c;
        __builtin_amdgcn_sched_barrier(0);
#endif
        // ---- the cluster
        __builtin_amdgcn_s_setprio(1);
        if constexpr ((OSUD_PH_EXP & 32) != 0) {
        } else if constexpr (kPlain && GEO == 0) {
          constexpr int jb = (P == 0 || P == 3) ? 0 : 1, ib = P < 2 ? 0 : 2;
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            mma<TE>(acc[ib][jb], fx[jb][s], fy[0][s]);
            mma<TE>(acc[ib + 1][jb], fx[jb][s], fy[1][s]);
          }
        } else if constexpr (kPlain) {
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            mma<TE>(acc[0][P], fx[0][s], fy[0][s]);
            mma<TE>(acc[1][P], fx[0][s], fy[1][s]);
          }
        } else if constexpr (kH8 && P < 2) {
#pragma unroll
          for (int i = 0; i < RY; ++i)
#pragma unroll
            for (int j = 0; j < RX; ++j) mma_f16(acc[i][j], fx[j][0], fy[i][0]);
        } else {
          constexpr bool second = (P & 1) != 0;
          auto mm = [&](f32x16& a, const u32x4& x0, const u32x4& x1, const u32x4& y0, const u32x4& y1) {
            if constexpr (kH8) mma_f8_lo(a, x0, x1, y0, y1);
            else mma_f8(a, x0, x1, y0, y1);
          };
          if constexpr (GEO == 0) {
            constexpr int ib = second ? 2 : 0;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j) mm(acc[ib + i][j], fx[j][0], fx[j][1], fy[i][0], fy[i][1]);
          } else if constexpr (!second) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j) mm(acc[i][j], fx[j][0], fx[j][1], fy[i][0], fy[i][1]);
          } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) mm(acc[i][2], fx[2][0], fx[2][1], fy[i][0], fy[i][1]);
          }
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        OSUD_PH_STAMP(ts_d);
        if (P != NPH - 1 || !last || grp == 0) __builtin_amdgcn_s_barrier();
        OSUD_PH_STAMP(ts_e);
      });
      r_buf = STAGE - r_buf;
    }
    // everything in flight is the next tile's first slab (and a half): landed before the epilogue's stores queue behind it
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef OSUD_PH_TIMING
    uint64_t ts_f, ts_g;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_f));
    tsum[4] += (uint64_t)(NPH * nk);
#ifdef OSUD_PH_SEGMENTS
    tsum[2] += ts_d - ts_c0; tsum[3] += ts_e - ts_d; ts_d = 0;
#endif
#endif
    tile_epilogue<TE, EPI, WY, WX, RY, RX>(p, acc, ty, tx, wy, wx, lane, pw, pr0, pr1, 0u, q_amax);
    __builtin_amdgcn_sched_barrier(0);
#ifdef OSUD_PH_TIMING
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_g));
    tsum[5] += ts_g - ts_f;
#endif
    __builtin_amdgcn_s_barrier();  // group 0 pays the stagger back; both groups: the next tile's first slab has landed everywhere
  }
  if constexpr (kF8) {
    if (p.out8 != nullptr && p.out8_slot != nullptr) {  // one amax atomic per workgroup (through LDS: the ring is idle by now)
      q_amax = wave_max(q_amax);
      __syncthreads();
      volatile float* red = reinterpret_cast<volatile float*>(smem);
      if (lane == 0) red[wave] = q_amax;
      __syncthreads();
      if (tid == 0) {
        float mx = 0.f;
        for (int w2 = 0; w2 < 8; ++w2) mx = fmaxf(mx, red[w2]);
        if (mx > 0.f) atomicMax(reinterpret_cast<unsigned*>(p.out8_slot) + 2, __float_as_uint(mx));
      }
    }
  }
#ifdef OSUD_PH_TIMING
  {
    uint64_t ts_end;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts_end));
    tsum[6] = ts_end - ts_k0;
    if (EPI != EPI_GATE_RES && p.gate != nullptr && lane == 0 && blockIdx.x < 32) {
      float* dbg = const_cast<float*>(p.gate) + (blockIdx.x * 8 + wave) * 8;
      for (int i = 0; i < 7; ++i) dbg[i] = (float)tsum[i];
    }
  }
#endif
}

template <typename TE, int EPI, int GEO> int launch_phased(const GemmP& p_in, hipStream_t st) {
  using G = PhGeo<GEO>;
  GemmP p = p_in;
  constexpr int STAGE = (256 + G::BN) * SLAB;
  const size_t lds = 2 * (size_t)STAGE + 8 * 4096;
  static bool attr_set = false;
  if (!attr_set) {
    OSUD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_phased_kernel<TE, EPI, GEO>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                 (int)lds));
    attr_set = true;
  }
  if (p.colpart_rows != nullptr) *p.colpart_rows = p.My / (G::RY * 32);
  const int ntiles = (p.My / 256) * (p.Nx / G::BN);
  int grid = gemm_num_cus();
  if (grid > ntiles) grid = ntiles;
  p.sched = nullptr;
  hipLaunchKernelGGL((gemm_phased_kernel<TE, EPI, GEO>), dim3(grid), dim3(512), lds, st, p);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

// launch_t's hook (pick: its geometry choice; 2 = 256 x 256, 1 = 256 x 192): takes the launch where the phased loop is built for it
template <typename TE, int EPI> int launch_phased_or(const GemmP& p, int pick, hipStream_t st, bool& taken) {
  taken = false;
  if constexpr ((sizeof(TE) == 2 && Planes<TE>::k == 1) || std::is_same<TE, h8_t>::value || sizeof(TE) == 1) {
    // option gemm_loop: 1 (default) = plain operands only; 2 = the K-blocked forms too.  Measured on those (profiles/r05_gemm_phase_stamps.md):
    // the half-length lead costs what the overlap gains -- fp16f8 sampling step 7.53 -> 7.87 ms, DiT-XL fp8 step 85.9 -> 91.4 ms -- so they
    // stay on the slab loop unless asked for (tests/test_gpu_phased.py holds both schedules to the same bits in every form).
    constexpr bool kPlain = sizeof(TE) == 2 && Planes<TE>::k == 1;
    if (opt(OPT_GEMM_LOOP) >= (kPlain ? 1 : 2) && (pick == 1 || pick == 2) && p.split_k <= 1 && !gemm_dynamic_tiles_wanted() &&
        (size_t)p.K * sizeof(TE) / SLAB >= 2) {
      taken = true;
      return pick == 2 ? launch_phased<TE, EPI, 0>(p, st) : launch_phased<TE, EPI, 1>(p, st);
    }
  }
  return OSUD_OK;
}

}  // namespace
}  // namespace osud
