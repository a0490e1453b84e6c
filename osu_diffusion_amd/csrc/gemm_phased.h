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
// Loads return in order, so `vmcnt(W[p])` before the first barrier of phase p retires exactly what phase p + 1 reads; it is never 0 in steady state (7-8 pieces = a whole slab stay in flight).  The rules a table is checked against at
// compile time (phased.h: sched_ok):
//   RAW  a piece is read one phase AFTER the phase whose counted wait (+ barrier, which every wave passes after its own wait)
//        retires it -- one barrier more than lock step needs, because group 1 waits one barrier later than group 0;
//   WAR  a region of a buffer is restaged at the earliest two phases after the phase that read it.
// A cluster is one quadrant of the wave's block grid -- two Y blocks against one X block -- over the WHOLE slab (its four 32-byte chunk
// pairs), so each phase needs only the rows of its quadrant and a region is free again two phases later: a slab and a half in flight.
// That holds for every operand form, because a form only decides what the chunk pairs of a row mean: bf16 / fp16: four k sub-steps
// of 16 (4 MFMAs per block); h8_t (fp16 | e4m3 planes of 32 k per row): two fp16 MFMAs + one block-scaled K = 64 MFMA per block; e4m3
// (128 k per row): two K = 64 MFMAs per block -- 128 pipe cycles per block and slab in all three, 256 per cluster.  (A first version cut
// the K-blocked forms by k sub-step instead: every phase then touches every row, a buffer can only be refilled while the other is
// consumed, the lead halves -- and it measured SLOWER than the slab loop: fp16f8 sampling 7.53 -> 7.87 ms, DiT-XL fp8 85.9 -> 91.4 ms.
// By quadrant: 7.60 -> 7.28 ms and 86.0 -> 86.2: profiles/r05_ab_runs.md.)
// Tiles: persistent workgroups, static XCD-contiguous order as in gemm_kernel or -- shared-GPU mode -- its per-XCD ticket queues (see
// "tile sequence" below); the stream runs across tile boundaries, everything in
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
template <int GEO> struct PhSched;
// 256 x 256: phases (Ya, X0) (Ya, X1) (Yb, X1) (Yb, X0), each over the slab's four 32-byte chunk pairs; the Yb fragments take Ya's registers.
// (Tuning builds measured {0, 2, 2, 4} / AHEAD 14 and {1, 2, 2, 3} / 13 -- nothing beside the 12 reads of phase 0 -- within +-1 %.)
template <> struct PhSched<0> : PhGeo<0> {
  static constexpr int NPH = 4, AHEAD = 12;
  static constexpr int cnt[NPH] = {2, 2, 2, 2};
  static constexpr int need[NPH] = {3, 5, 7, -1};
  static constexpr int read_phase[PPW] = {0, 0, 0, 0, 1, 1, 2, 2};
};
// 256 x 192: phases (Y, X0) (Y, X1) (Y, X2); all of Y is read in phase 0, so its five slots can only be restaged in phase 2
template <> struct PhSched<1> : PhGeo<1> {
  static constexpr int NPH = 3, AHEAD = 12;
  static constexpr int cnt[NPH] = {1, 1, 5};
  static constexpr int need[NPH] = {4, 5, 6};
  static constexpr int read_phase[PPW] = {0, 0, 0, 0, 0, 1, 2};
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
  using S = PhSched<GEO>;
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

  // ---- tile sequence.  Static: first, first + G8, ...  Queued (p.sched != nullptr: osud_set_gemm_dynamic_tiles, multi-round launches): every
  // later tile is drawn from the ticket counter of this workgroup's XCD, two tiles ahead of the arithmetic, exactly as in gemm_kernel
  // (same counters, same ticket -> tile map: an undisturbed launch walks the static order; every tile is computed the same way whoever
  // takes it, so the results are bit-identical).  The ticket for tile j + 2 is requested at the top of tile j, has returned by the drain
  // in front of tile j's epilogue -- the only vmcnt(0) of the loop: the compiler's own wait for the returned value costs nothing there
  // -- and is published behind the epilogue through the first word of wave 0's epilogue patch (free again by then; the 256 x 256
  // geometry fills the LDS to the last byte), across the tile's closing barrier.  The staging cursor crosses into the next tile once per
  // tile, 1.5 slabs ahead of the consumer, and takes the tile id the consumer already holds (t_nxt).
  const bool dyn = p.sched != nullptr;
  volatile __attribute__((address_space(3))) uint32_t* const sched_word =
      reinterpret_cast<volatile __attribute__((address_space(3))) uint32_t*>((lds_void*)smem) + RING / 4;
  const bool ticket_lane = dyn && wave == 0 && lane == 0;
  const int xcd = blockIdx.x & 7, per = G8 >> 3;
  auto resolve = [&](uint32_t word) -> int {  // one queue per XCD: 16-bit fields of p.sched[0..3] (gemm_kernel.h has the design notes)
    const uint32_t k = (word >> (16 * (xcd & 1))) & 0xffffu;
    const int id = (int)((uint32_t)G8 * (1u + k / (uint32_t)per) + (uint32_t)(xcd * per) + k % (uint32_t)per);
    return id < ntiles ? id : 0x7fffffff;
  };
  auto take_ticket = [&](uint32_t& dst) {
    if (ticket_lane)
      dst = __hip_atomic_fetch_add(p.sched + (xcd >> 1), 1u << (16 * (xcd & 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  uint32_t tk_start = 0, tk = 0;
  take_ticket(tk_start);  // the second tile's (older than every LDS-DMA piece: back by the prologue's wait)
  int t_nxt = dyn ? 0x7fffffff : first + G8;

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
    // (M0 is written and consumed inside ONE asm statement.  It cannot be named as a clobber: hipcc treats M0 as a reserved register and rejects it
    //  from clobber lists with a warning ("may not be preserved across the asm statement") -- the compiler never keeps a value of its own in M0
    //  across an asm statement on this target: it re-materialises M0 in front of each of its own uses (s_movrel, LDS-DMA builtins, sendmsg).)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(vo), "s"(sb), "s"(dst) : "memory");
  };
  auto advance = [&]() {
    c_buf = STAGE - c_buf;
    c_gy += SLAB;
    c_gx += SLAB;
    if (++c_kt == nk) {
      c_kt = 0;
      c_tile = t_nxt;  // (the consumer is in the tile the cursor leaves: its next tile is the cursor's)
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
  if (dyn) {
    if (ticket_lane) sched_word[0] = (uint32_t)resolve(tk_start);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (dyn) t_nxt = __builtin_amdgcn_readfirstlane((int)sched_word[0]);

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
  for (int t_cur = first; t_cur < ntiles;) {
    const int ty = t_cur / ntx, tx = t_cur - ty * ntx;
    take_ticket(tk);  // for the tile after next
    f32x16 acc[RY][RX];
#pragma unroll
    for (int i = 0; i < RY; ++i)
#pragma unroll
      for (int j = 0; j < RX; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // fragments: fy[i][s] two Y blocks x the slab's four 32-byte chunk pairs, fx[j][s] one or two X blocks (GEO 1 uses fx[0] only)
    u32x4 fy[2][4], fx[2][4];
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
        } else if constexpr (GEO == 0) {
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
        } else {
          if constexpr (P == 0) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              fy[0][s] = ds_read16<S::Y_OFF[0]>(ya[s] + r_buf);
              fy[1][s] = ds_read16<S::Y_OFF[1]>(ya[s] + r_buf);
            }
          }
#pragma unroll
          for (int s = 0; s < 4; ++s) fx[0][s] = ds_read16<S::X_OFF[P]>(xa[s] + r_buf);
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
        // two accumulator blocks (Y blocks y0, y1 of the wave against one X block) over the slab's four chunk pairs, the two accumulators
        // alternating; per accumulator the products come in the slab loop's order (compute_slab / _h8 / the e4m3 halves): same bits
        auto pair = [&](f32x16& a0, f32x16& a1, const u32x4 (&x)[4], const u32x4 (&y0)[4], const u32x4 (&y1)[4]) {
          if constexpr (kPlain) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              mma<TE>(a0, x[s], y0[s]);
              mma<TE>(a1, x[s], y1[s]);
            }
          } else if constexpr (kH8) {  // chunk pairs 0, 1: the fp16 values of k 0-15 / 16-31; 2 | 3: the e4m3 planes P | Q of all 32 k
            mma_f16(a0, x[0], y0[0]);
            mma_f16(a1, x[0], y1[0]);
            mma_f16(a0, x[1], y0[1]);
            mma_f16(a1, x[1], y1[1]);
            mma_f8_lo(a0, x[2], x[3], y0[2], y0[3]);
            mma_f8_lo(a1, x[2], x[3], y1[2], y1[3]);
          } else {  // e4m3: chunk pairs 0 | 1 = k 0-63, 2 | 3 = k 64-127
            mma_f8(a0, x[0], x[1], y0[0], y0[1]);
            mma_f8(a1, x[0], x[1], y1[0], y1[1]);
            mma_f8(a0, x[2], x[3], y0[2], y0[3]);
            mma_f8(a1, x[2], x[3], y1[2], y1[3]);
          }
        };
        if constexpr ((OSUD_PH_EXP & 32) != 0) {
        } else if constexpr (GEO == 0) {
          constexpr int jb = (P == 0 || P == 3) ? 0 : 1, ib = P < 2 ? 0 : 2;
          pair(acc[ib][jb], acc[ib + 1][jb], fx[jb], fy[0], fy[1]);
        } else {
          pair(acc[0][P], acc[1][P], fx[0], fy[0], fy[1]);
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
    int t_nxt2 = t_nxt + G8, t_pub = 0;  // (t_pub: the ticket lane's own value -- per-lane, so that the tile ids themselves stay wave-uniform)
    if (ticket_lane) {  // the ticket has returned with the drain above: resolve it HERE (pinned: behind the epilogue the compiler's wait for it would drain the stores)
      t_pub = resolve(tk);
      asm volatile("" : "+v"(t_pub)::"memory");
    }
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
    if (dyn) {  // wave 0's patch is idle again: the id of the tile after next, for every wave behind the closing barrier
      if (ticket_lane) sched_word[0] = (uint32_t)t_pub;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();  // group 0 pays the stagger back; both groups: the next tile's first slab has landed everywhere
    if (dyn) t_nxt2 = __builtin_amdgcn_readfirstlane((int)sched_word[0]);
    t_cur = t_nxt;
    t_nxt = t_nxt2;
  }
  if (ticket_lane) {  // the last workgroup out re-arms the counters for the next launch that borrows this slot
    const unsigned done = __hip_atomic_fetch_add(p.sched + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done == gridDim.x - 1) {
#pragma unroll
      for (int y = 0; y < 9; ++y) __hip_atomic_store(p.sched + y, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
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
  // shared-GPU mode (collectives on the same compute units): multi-round launches draw their tiles from the per-XCD ticket queues
  p.sched = (gemm_dynamic_tiles_wanted() && ntiles > grid && grid % 8 == 0 && ntiles / 8 + 2 * grid < 60000) ? gemm_sched_slot() : nullptr;
  hipLaunchKernelGGL((gemm_phased_kernel<TE, EPI, GEO>), dim3(grid), dim3(512), lds, st, p);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

// launch_t's hook (pick: its geometry choice; 2 = 256 x 256, 1 = 256 x 192): takes the launch where the phased loop is built for it
template <typename TE, int EPI> int launch_phased_or(const GemmP& p, int pick, hipStream_t st, bool& taken) {
  taken = false;
  if constexpr ((sizeof(TE) == 2 && Planes<TE>::k == 1) || std::is_same<TE, h8_t>::value || sizeof(TE) == 1) {
    if (opt(OPT_GEMM_LOOP) != 0 && (pick == 1 || pick == 2) && p.split_k <= 1 && (size_t)p.K * sizeof(TE) / SLAB >= 2) {
      taken = true;
      return pick == 2 ? launch_phased<TE, EPI, 0>(p, st) : launch_phased<TE, EPI, 1>(p, st);
    }
  }
  return OSUD_OK;
}

}  // namespace
}  // namespace osud
