// Weight-gradient product without operand transposes (bf16 tier):
//     out[y][x] = sum_m P[m][y0 + y] * Q[m][x0 + x]          (m = token index)
// P = gradient of a Linear's output [M][ldp], Q = its input activations [M][ldq], both row-major as
// the forward/backward kernels left them, i.e. the contraction index m is the ROW index.  The MFMA
// wants 8 consecutive m per lane for a fixed feature; gfx950's ds_read_b64_tr_b16 delivers exactly that
// from a token-major LDS tile: each 16-lane group reads a 4-token x 16-feature block and every lane
// receives one feature's 4 tokens (semantics pinned on hardware by tools/probes/tr_probe.hip).
//
// Structure = gemm.hip's: persistent workgroups, stages of 64 tokens HBM->LDS via global_load_lds
// (1 KiB pieces), counted waits, one barrier per stage, 8 waves of (RY*32) x (RX*32) outputs, split over
// the token axis (blockIdx.y) into fp32 partial slabs combined deterministically by splitk_reduce.
// LDS image per stage: [64 tokens][BM features] | [64 tokens][BN features]; the 16-byte chunk index of a
// token row is XOR-swizzled with (token&3)<<2 (source side) so the 4 token rows of one transposing read
// fall into 4 different 64-byte bank segments.
#include <stdlib.h>

#include <atomic>

#include "gemm.h"
#include "kernels.h"
#include "phased.h"

namespace osud {

namespace {

constexpr int BKT = 64;  // tokens per stage
#ifndef OSUD_WG192_MARGIN
// how much better a 192-wide geometry has to fill the chip (useful tile area x occupied compute units) than the padded 256 x 256 one to be
// taken.  At DiT-XL's shapes the ratio is 1.27-1.34 and the two phased kernels are a wash: a block's four weight gradients 1127 us
// (192-wide) vs 1172 us (padded 256 x 256) stand-alone, but 111.2 vs 110.7 ms per XL bf16 training step -- the 256 x 256 geometry stays the
// choice there (profiles/r06_wgrad_xl_geometry.txt); the slab loop's 192-wide kernels were 5 % behind both
#define OSUD_WG192_MARGIN 1.5
#endif
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));

template <int OFF> __device__ __forceinline__ u32x2 ds_read_tr(uint32_t addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
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

template <int WY, int WX, int RY, int RX> struct WGeo {
  static constexpr int BM = WY * RY * 32, BN = WX * RX * 32, NW = WY * WX, NT = 64 * NW;
  static constexpr int ROWY = BM * 2, ROWX = BN * 2;               // bytes per token row
  static constexpr int YB = BKT * ROWY, XB = BKT * ROWX, STAGE = YB + XB;
  static constexpr int NSTAGE = STAGE * 3 <= 160 * 1024 ? 3 : 2;
  static constexpr int PIECES = STAGE / 1024, PPW = PIECES / NW;   // 1 KiB LDS-DMA pieces per stage / per wave
  // token rows of 256 / 512 bytes (128 / 256 features) divide a 1 KiB piece; rows of 384 bytes (192 features: DiT-XL's 1152 =
  // 6 x 192) do not -- the per-lane source offsets below are computed per 16-byte chunk, so a piece may straddle token rows
  static_assert(PIECES % NW == 0 && YB % 1024 == 0 && XB % 1024 == 0 && (ROWY == 256 || ROWY == 384 || ROWY == 512) &&
                    (ROWX == 256 || ROWX == 384 || ROWX == 512),
                "geometry");
  // swizzle of the 16-byte chunk index of a token row (applied on the source side of the LDS-DMA): power-of-two rows XOR the
  // chunk bits 2..3 with token & 3 (the 4 token rows of one transposing read fall into 4 different 64-byte bank segments);
  // 384-byte rows start 32 banks apart, so rows t and t + 2 collide: XOR chunk bit 2 with (token >> 1) & 1
  static constexpr int swz(int rowb, int tok) { return rowb == 384 ? ((tok >> 1) & 1) : (tok & 3); }
  static_assert(NSTAGE * STAGE + NW * 4096 <= 160 * 1024, "stage ring + epilogue patches must fit the LDS");
};

// this wave's share of one 64-token stage: SGPR panel base + per-lane 32-bit offset computed once (see gemm.hip)
template <typename G>
__device__ __forceinline__ void stage_tokens(const char* gp, const char* gq, uint32_t stage_lds, const uint32_t (&voff)[G::PPW],
                                             int wave) {
#pragma unroll
  for (int q = 0; q < G::PPW; ++q) {
    const int piece = wave * G::PPW + q;  // wave-uniform
    const char* sbase = (piece * 1024 < G::YB) ? gp : gq;
    const uint32_t dst = stage_lds + (uint32_t)__builtin_amdgcn_readfirstlane(piece * 1024);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff[q]), "s"(sbase), "s"(dst) : "memory");
  }
}

template <int RY, int RX> struct TFrag {
  u32x2 y[RY][2], x[RX][2];
};
template <int RY, int RX, int S, int ROWY, int ROWX>
__device__ __forceinline__ void read_frags(TFrag<RY, RX>& f, const uint32_t (&ya)[RY], const uint32_t (&xa)[RX], uint32_t so) {
#pragma unroll
  for (int i = 0; i < RY; ++i) {
    f.y[i][0] = ds_read_tr<S * 16 * ROWY>(ya[i] + so);
    f.y[i][1] = ds_read_tr<S * 16 * ROWY + 4 * ROWY>(ya[i] + so);
  }
#pragma unroll
  for (int j = 0; j < RX; ++j) {
    f.x[j][0] = ds_read_tr<S * 16 * ROWX>(xa[j] + so);
    f.x[j][1] = ds_read_tr<S * 16 * ROWX + 4 * ROWX>(xa[j] + so);
  }
}
template <int RY, int RX> __device__ __forceinline__ void mma_frags(f32x16 (&acc)[RY][RX], const TFrag<RY, RX>& f) {
#pragma unroll
  for (int i = 0; i < RY; ++i) {
    u32x4 yv;
    yv[0] = f.y[i][0][0]; yv[1] = f.y[i][0][1]; yv[2] = f.y[i][1][0]; yv[3] = f.y[i][1][1];
#pragma unroll
    for (int j = 0; j < RX; ++j) {
      u32x4 xv;
      xv[0] = f.x[j][0][0]; xv[1] = f.x[j][0][1]; xv[2] = f.x[j][1][0]; xv[3] = f.x[j][1][1];
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, xv), __builtin_bit_cast(bf16x8, yv),
                                                          acc[i][j], 0, 0, 0);
    }
  }
}
#define OSUD_WG_WAIT(n)                                     \
  asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory");  \
  __builtin_amdgcn_sched_barrier(0)

struct WgradP {
  const bf16_t* P;
  const bf16_t* Q;
  int ldp, ldq;   // elements
  int Ny, Nx, M;  // output rows (features of P), output cols (features of Q), tokens
  float* out;     // [splits][Ny][Nx] fp32
  int split_k;
  size_t split_stride;
  unsigned* queue;  // shared-GPU mode (256x256 geometry, splits > 1): [tile] chunk tickets, [63] finished workgroups; else null
  int chunk;        // chunks per tile (a multiple of split_k)
  // (built, the same bits, measured slower and removed: the split-K combine inside this launch -- the workgroups of a tile meeting
  //  on an arrival counter -- 29.99 vs 28.73 ms per training step: HISTORY.md section 4 "round 2")
  // one (tile, split) unit per workgroup, plain mode: units in split-major order, one contiguous run per XCD (workgroup L runs on
  // XCD L % 8), so that the ~32 workgroups of an XCD walk the SAME token rows -- one or two splits, all of their tiles -- and share
  // every stage's operand rows through that XCD's L2.  With blockIdx = (tile, split) an XCD held 4-5 tiles of each of the splits
  // and read the Q operand 8x and the P operand 1.5x from memory: 579 MB of L2 fills per launch for 251 MB of operands
  // (profiles/r02_pmc_hbm.json), at 5.9 TB/s the bound of the kernel.
  int xcd_units;
};

template <int WY, int WX, int RY, int RX>
__global__ __launch_bounds__((WGeo<WY, WX, RY, RX>::NT)) void wgrad_kernel(WgradP p) {
  using G = WGeo<WY, WX, RY, RX>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wy = wave / WX, wx = wave % WX;
  const int frow = lane & 31, fhalf = lane >> 5;

  // (edge tiles: Ny / Nx may be odd multiples of 128 under the 256-wide geometry; the out-of-range half is computed on
  // whatever the staging reads -- see dev_alloc's slack -- and never stored)
  const int ntx = (p.Nx + G::BN - 1) / G::BN, ntiles = ((p.Ny + G::BM - 1) / G::BM) * ntx;
  // this split's share of the token axis (any split count: ranges differ by at most one stage)
  const int st_total = p.M / BKT;
  int sidx = blockIdx.y, bx = blockIdx.x;  // split, tile-block index
  if (p.xcd_units) {
    const int total = gridDim.x * gridDim.y, L = blockIdx.x + blockIdx.y * gridDim.x;
    const int q = total >> 3, r = total & 7, xcd = L & 7, idx = L >> 3;
    const int u = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    sidx = u / (int)gridDim.x;
    bx = u - sidx * (int)gridDim.x;
  }
  const int st_begin = (int)((long)sidx * st_total / p.split_k);
  const int nst = (int)((long)(sidx + 1) * st_total / p.split_k) - st_begin;
  const size_t ldp_b = (size_t)p.ldp * 2, ldq_b = (size_t)p.ldq * 2;
  const char* gp0 = reinterpret_cast<const char*>(p.P) + (size_t)st_begin * BKT * ldp_b;
  const char* gq0 = reinterpret_cast<const char*>(p.Q) + (size_t)st_begin * BKT * ldq_b;
  float* outp = p.out + (size_t)sidx * p.split_stride;

  // per-lane LDS byte addresses (stage 0, k-substep 0, first of the two transposing reads):
  //   token row = 8*fhalf + ((lane&15)>>2), feature = block + 32*i + 16*((lane>>4)&1) + 4*(lane&3)
  const uint32_t lds0 = (uint32_t)(size_t)(lds_void*)smem;
  const int tok0 = 8 * fhalf + ((lane & 15) >> 2);
  const int swy = G::swz(G::ROWY, tok0), swx = G::swz(G::ROWX, tok0);  // (+4 and +16 token rows keep them)
  const int fbyte = 32 * ((lane >> 4) & 1) + 8 * (lane & 3);   // byte offset inside a 64-byte (32-feature) group
  uint32_t ya[RY], xa[RX];
#pragma unroll
  for (int i = 0; i < RY; ++i) {
    const int hi = (wy * RY + i) ^ swy;  // 64-byte group index, swizzled (chunk bits 2..3 <-> group bits 0..1)
    ya[i] = lds0 + tok0 * G::ROWY + hi * 64 + fbyte;
  }
#pragma unroll
  for (int j = 0; j < RX; ++j) {
    const int hi = (wx * RX + j) ^ swx;
    xa[j] = lds0 + G::YB + tok0 * G::ROWX + hi * 64 + fbyte;
  }

  // per-lane byte offsets of this wave's LDS-DMA pieces inside a (P panel | Q panel) stage of 64 tokens
  uint32_t dma_off[G::PPW];
#pragma unroll
  for (int q = 0; q < G::PPW; ++q) {
    const int piece = wave * G::PPW + q;
    const bool isY = piece * 1024 < G::YB;
    const int rowb = isY ? G::ROWY : G::ROWX;                  // bytes per token row in this part
    const int pb = isY ? piece * 1024 : piece * 1024 - G::YB;  // byte offset inside the part
    const int lpr = rowb / 16;                                  // 16-byte chunks per token row
    const int cidx = pb / 16 + lane;                            // this lane's chunk inside the part
    const int tok = cidx / lpr, pos = cidx - tok * lpr;
    const int c = pos ^ (G::swz(rowb, tok) << 2);               // source chunk for LDS position `pos`
    dma_off[q] = (uint32_t)((size_t)tok * (isY ? ldp_b : ldq_b) + c * 16);
  }
  // epilogue patch (4 KiB per wave, behind the stage ring)
  const uint32_t patch = lds0 + G::NSTAGE * G::STAGE + wave * 4096;
  uint32_t pw[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) pw[g] = patch + frow * 128 + (((2 * g + fhalf) ^ (frow & 7)) << 4);
  const uint32_t pr = patch + (lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) << 4);

  f32x16 acc[RY][RX];
  auto clear_acc = [&]() {
#pragma unroll
    for (int i = 0; i < RY; ++i)
#pragma unroll
      for (int j = 0; j < RX; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  auto compute_stage = [&](uint32_t so) {
    TFrag<RY, RX> f0, f1;
    read_frags<RY, RX, 0, G::ROWY, G::ROWX>(f0, ya, xa, so);
    read_frags<RY, RX, 1, G::ROWY, G::ROWX>(f1, ya, xa, so);
    if constexpr (RY + RX == 6) { OSUD_WG_WAIT(12); } else if constexpr (RY + RX == 5) { OSUD_WG_WAIT(10); } else { OSUD_WG_WAIT(8); }
    mma_frags<RY, RX>(acc, f0);
    read_frags<RY, RX, 2, G::ROWY, G::ROWX>(f0, ya, xa, so);
    if constexpr (RY + RX == 6) { OSUD_WG_WAIT(12); } else if constexpr (RY + RX == 5) { OSUD_WG_WAIT(10); } else { OSUD_WG_WAIT(8); }
    mma_frags<RY, RX>(acc, f1);
    read_frags<RY, RX, 3, G::ROWY, G::ROWX>(f1, ya, xa, so);
    if constexpr (RY + RX == 6) { OSUD_WG_WAIT(12); } else if constexpr (RY + RX == 5) { OSUD_WG_WAIT(10); } else { OSUD_WG_WAIT(8); }
    mma_frags<RY, RX>(acc, f0);
    OSUD_WG_WAIT(0);
    mma_frags<RY, RX>(acc, f1);
  };
  // The MFMA leaves lane (frow, fhalf) with y = frow, x = 8g + 4*fhalf + {0..3}; every 32x32 block goes through the
  // wave's 4 KiB LDS patch and comes back row-major (rows 8q + (lane>>3), x = 4*(lane&7)..+3) so that 8 lanes
  // store one full 128-byte row segment (see gemm.hip's epilogue).
  auto store_tile = [&](int ty, int tx) {
#pragma unroll
    for (int i = 0; i < RY; ++i) {
      const int y0 = ty * G::BM + wy * RY * 32 + i * 32 + (lane >> 3);
#pragma unroll
      for (int j = 0; j < RX; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v;
          v[0] = acc[i][j][4 * g + 0]; v[1] = acc[i][j][4 * g + 1]; v[2] = acc[i][j][4 * g + 2]; v[3] = acc[i][j][4 * g + 3];
          ds_write16(pw[g], v);
        }
        f32x4 t[4];
        t[0] = ds_read16f<0>(pr);
        t[1] = ds_read16f<1024>(pr);
        t[2] = ds_read16f<2048>(pr);
        t[3] = ds_read16f<3072>(pr);
        OSUD_WG_WAIT(0);
        const int x = tx * G::BN + wx * RX * 32 + j * 32 + 4 * (lane & 7);
        if (x < p.Nx) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (y0 + 8 * q < p.Ny) store4(outp + (size_t)(y0 + 8 * q) * p.Nx + x, t[q][0], t[q][1], t[q][2], t[q][3]);
        }
      }
    }
  };

  if constexpr (G::NSTAGE == 2) {
    if (p.queue != nullptr) {
      // ---- shared-GPU mode: the token axis of a tile is cut into p.chunk equal chunks; the tile's split_k workgroups start
      // on chunks 0..split_k-1 and draw every later chunk from the tile's ticket counter, accumulating in registers, so that a
      // workgroup whose compute unit is held by another kernel leaves its share to the others instead of doubling the launch.
      // With two stages the loop drains this wave's queue at every stage: the ticket requested behind the barrier of a chunk's
      // first stage has returned by the wait of its second and is published through LDS (a word of the idle epilogue patch).
      const int tile = blockIdx.x, ty = tile / ntx, tx = tile % ntx;
      const int nch = p.chunk;  // chunks per tile: a multiple of split_k, so an undisturbed launch gives every workgroup the same number
      auto chunk_begin = [&](int c) { return (int)((long)c * st_total / nch); };  // lengths differ by at most one stage
      auto chunk_len = [&](int c) { return chunk_begin(c + 1) - chunk_begin(c); };
      volatile __attribute__((address_space(3))) uint32_t* word =
          reinterpret_cast<volatile __attribute__((address_space(3))) uint32_t*>((lds_void*)smem) + (G::NSTAGE * G::STAGE) / 4;
      const bool ticket_lane = wave == 0 && lane == 0;
      const char* gpt = reinterpret_cast<const char*>(p.P) + (size_t)ty * G::BM * 2;
      const char* gqt = reinterpret_cast<const char*>(p.Q) + (size_t)tx * G::BN * 2;
      int c_cur = blockIdx.y, c_nxt = 0x7fffffff;
      int ic_rel = 0, ic_st = 0, issued = 0, consumed = 0;
      auto issue_next = [&]() {
        const int c = ic_rel == 0 ? c_cur : c_nxt;
        if (ic_rel < 2 && c < nch) {
          const size_t stage = (size_t)chunk_begin(c) + ic_st;
          stage_tokens<G>(gpt + stage * BKT * ldp_b, gqt + stage * BKT * ldq_b, lds0 + (uint32_t)((issued % G::NSTAGE) * G::STAGE),
                          dma_off, wave);
          ++issued;
          if (++ic_st == chunk_len(c)) {
            ic_st = 0;
            ++ic_rel;
          }
        }
      };
      issue_next();
      clear_acc();
      uint32_t tk = 0;
      while (c_cur < nch) {
        const int len = chunk_len(c_cur);
        for (int st = 0; st < len; ++st) {
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
          if (st == 1) {
            if (ticket_lane) word[0] = (uint32_t)p.split_k + tk;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          }
          __builtin_amdgcn_s_barrier();
          if (st == 1) c_nxt = __builtin_amdgcn_readfirstlane((int)word[0]);
          if (st == 0 && ticket_lane) tk = __hip_atomic_fetch_add(p.queue + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          issue_next();
          compute_stage((uint32_t)((consumed % G::NSTAGE) * G::STAGE));
          ++consumed;
        }
        c_cur = c_nxt;
        c_nxt = 0x7fffffff;
        if (ic_rel > 0) --ic_rel;
      }
      store_tile(ty, tx);
      if (ticket_lane) {  // the last workgroup out re-arms the counters
        const unsigned done = __hip_atomic_fetch_add(p.queue + 63, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (done == gridDim.x * gridDim.y - 1)
          for (int i = 0; i < 64; ++i) __hip_atomic_store(p.queue + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      return;
    }
  }

  const int G8 = gridDim.x;
  int first;
  {
    const int b = blockIdx.x, q = G8 >> 3, r = G8 & 7, xcd = b & 7, idx = b >> 3;
    first = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    if (p.xcd_units) first = bx;  // (already placed by the unit remap above)
  }
  int ic_tile = first, ic_st = 0, issued = 0, consumed = 0;
  auto issue_next = [&]() {
    if (ic_tile < ntiles) {
      const int ty = ic_tile / ntx, tx = ic_tile % ntx;
      const char* gp = gp0 + (size_t)ic_st * BKT * ldp_b + (size_t)ty * G::BM * 2;
      const char* gq = gq0 + (size_t)ic_st * BKT * ldq_b + (size_t)tx * G::BN * 2;
      stage_tokens<G>(gp, gq, lds0 + (uint32_t)((issued % G::NSTAGE) * G::STAGE), dma_off, wave);
      ++issued;
      if (++ic_st == nst) {
        ic_st = 0;
        ic_tile += G8;
      }
    }
  };
#pragma unroll
  for (int i = 0; i < G::NSTAGE - 1; ++i) issue_next();

  for (int tile = first; tile < ntiles; tile += G8) {
    const int ty = tile / ntx, tx = tile % ntx;
    clear_acc();
    for (int st = 0; st < nst; ++st) {
      const int ahead = issued - consumed - 1;
      if (ahead <= 0 || G::NSTAGE == 2) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else if (G::PPW == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      issue_next();
      compute_stage((uint32_t)((consumed % G::NSTAGE) * G::STAGE));
      ++consumed;
    }
    store_tile(ty, tx);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The 256 x 256 geometry in the phased schedule (gemm_phased.h has the design; phased.h the table rules): the same staging, the
// same transposing reads and MFMAs, every accumulator sees its tokens in the same order -- bit-identical partial slabs -- but
// the two waves of a SIMD run one barrier apart.  The cut is natural here: phase p of a stage is k sub-step p, i.e. tokens
// 16 p .. 16 p + 15 of BOTH operands (12 transposing reads, 8 MFMAs over all eight accumulators), and the LDS image is token-major
// already, so a phase needs exactly one quarter of the stage (16 KiB = two pieces per wave: slot 2 p the P rows, slot 2 p + 1 the
// Q rows) and the stream is in need order as it stands.  vmcnt(10) per phase in steady state: five quarters in flight.
struct WgSched {
  static constexpr int NPH = 4, PPW = 8, AHEAD = 12;
  static constexpr int cnt[NPH] = {2, 2, 2, 2};
  static constexpr int need[NPH] = {1, 3, 5, 7};
  static constexpr int read_phase[PPW] = {0, 0, 1, 1, 2, 2, 3, 3};
};

// (DYN: the queue mode is its own instantiation -- with the claim protocol compiled into the one kernel the plain mode's launches ran 13 % longer,
//  109.5 -> 123.8 us in the training step on the same box: profiles/r06_ab_runs.md)
template <bool DYN> __global__ __launch_bounds__(512) void wgrad_phased_kernel(WgradP p) {
  using G = WGeo<2, 4, 4, 2>;
  using S = WgSched;
  static_assert(sched_ok<S>(), "phase table breaks a staging rule");
  constexpr int WX = 4, RY = 4, RX = 2, NPH = S::NPH, PPW = S::PPW, STAGE = G::STAGE, RING = 2 * STAGE;
  static_assert(G::NSTAGE == 2 && G::PPW == PPW && G::ROWY == 512 && G::ROWX == 512, "geometry");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wy = wave / WX, wx = wave % WX, grp = wave >> 2;
  const int frow = lane & 31, fhalf = lane >> 5;
  const int ntx = (p.Nx + G::BN - 1) / G::BN, ntiles = ((p.Ny + G::BM - 1) / G::BM) * ntx;
  const int st_total = p.M / BKT;
  // Shared-GPU mode (p.queue != nullptr; one (tile, split) unit per workgroup, placed as in the plain mode): the split's share of the token
  // axis is cut into per_wg = p.chunk / split_k chunks.  The workgroup OWNS its split's chunks and walks them front to back -- in an undisturbed
  // launch exactly the plain mode's stages in the plain mode's order: the same L2 locality (WgradP::xcd_units) and the same bits -- but it
  // has to CLAIM every chunk after the first from the split's counter word, because a workgroup that has run out of chunks (its own compute
  // unit was free while another's was held by a collective) steals from the BACK of the other splits of its tile:
  //     word = {low 16 bits: chunks the owner has claimed beyond chunk 0 | high 16 bits: chunks stolen from the back}
  //     owner: old = add(word, 1)        -> chunk 1 + old.lo, valid while 1 + old.lo <= per_wg - 1 - old.hi
  //     thief: old = add(word, 1 << 16)  -> chunk per_wg - 1 - old.hi, valid under the same condition
  // (an invalid claim only pushes a counter further past the other: the range stays empty).  A stolen chunk is accumulated in the thief's
  // registers and leaves in the thief's partial slab: the combine adds the slabs in a fixed order, so only WHICH slab holds a chunk's sum
  // -- rounding -- depends on who was held.  A run of stages ("segment") is a tile's share of the token axis in the plain mode and one
  // chunk here; the staging stream runs from segment to segment, the epilogue follows a TILE.
  // The claims are SCALAR atomics (s_atomic_add ... glc: the value before the add comes back in an SGPR and is tracked by lgkmcnt -- gfx950
  // executes it, tools/probes/satomic_probe.hip), so they never enter the vector-memory queue whose counted waits the stream lives on.
  // Wave 0 runs the protocol in the phases of a chunk's first two stages, one scalar memory operation per phase, issued at the phase's top and
  // found back behind the phase's own barrier + lgkmcnt(0) (~300 ns on an idle chip, tools/probes/satomic_latency.hip: the price of a chunk): the
  // claim on the own word; once the own share is used up, a LOOK at eight words of the tile's row (s_load_dwordx8 ... glc) for the split with the
  // most unclaimed chunks, then the claim on that split's word.  An undisturbed launch pays one claim per chunk and, at its end, one failed
  // claim and one look that finds nothing.  The chunk goes out through the first word of wave 0's (idle) epilogue patch (six operations at most:
  // up to phase 1 of the second stage); all waves read it in phase 3 of the second stage -- the cursor of a four-stage chunk, the shortest
  // the launcher allows, moves on in phase 1 of the third.
  constexpr bool dyn = DYN;
  const int nch = dyn ? p.chunk : 1, per_wg = dyn ? p.chunk / p.split_k : 1;
  auto chunk_begin = [&](int c) { return (int)((long)c * st_total / nch); };  // (lengths differ by at most one stage)
  int sidx = blockIdx.y, bx = blockIdx.x;
  if (p.xcd_units) {  // (tile, split) units in split-major order, one contiguous run per XCD: see WgradP::xcd_units
    const int total = gridDim.x * gridDim.y, L = blockIdx.x + blockIdx.y * gridDim.x;
    const int q = total >> 3, r = total & 7, xcd = L & 7, idx = L >> 3;
    const int u = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    sidx = u / (int)gridDim.x;
    bx = u - sidx * (int)gridDim.x;
  }
  const int st_begin = dyn ? 0 : (int)((long)sidx * st_total / p.split_k);
  const int nst = dyn ? 0 : (int)((long)(sidx + 1) * st_total / p.split_k) - st_begin;
  const size_t ldp_b = (size_t)p.ldp * 2, ldq_b = (size_t)p.ldq * 2;
  const char* const gp0 = reinterpret_cast<const char*>(p.P) + (size_t)st_begin * BKT * ldp_b;
  const char* const gq0 = reinterpret_cast<const char*>(p.Q) + (size_t)st_begin * BKT * ldq_b;
  float* const outp = p.out + (size_t)sidx * p.split_stride;
  const int G8 = gridDim.x;
  int first;
  {
    const int b = blockIdx.x, q = G8 >> 3, r = G8 & 7, xcd = b & 7, idx = b >> 3;
    first = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    if (p.xcd_units) first = bx;
    if (dyn) first = bx;
  }

  // fragment addresses (stage 0, sub-step 0), exactly wgrad_kernel's
  const uint32_t lds0 = (uint32_t)(size_t)(lds_void*)smem;
  const int tok0 = 8 * fhalf + ((lane & 15) >> 2);
  const int swz = tok0 & 3;
  const int fbyte = 32 * ((lane >> 4) & 1) + 8 * (lane & 3);
  uint32_t ya[RY], xa[RX];
#pragma unroll
  for (int i = 0; i < RY; ++i) ya[i] = lds0 + tok0 * G::ROWY + (((wy * RY + i) ^ swz) * 64) + fbyte;
#pragma unroll
  for (int j = 0; j < RX; ++j) xa[j] = lds0 + G::YB + tok0 * G::ROWX + (((wx * RX + j) ^ swz) * 64) + fbyte;
  // LDS-DMA: slot m = 2 q + h of wave w is piece 8 q + w of the P part (h = 0) or of the Q part (h = 1): token rows 16 q + 2 w, + 1
  uint32_t voff[PPW];
#pragma unroll
  for (int m = 0; m < PPW; ++m) {
    const bool isY = (m & 1) == 0;
    const int pb = (8 * (m >> 1) + wave) * 1024;   // byte offset inside the part
    const int cidx = pb / 16 + lane;                // this lane's 16-byte chunk inside the part (32 chunks per token row)
    const int tok = cidx >> 5, pos = cidx & 31;
    const int c = pos ^ ((tok & 3) << 2);
    voff[m] = (uint32_t)((size_t)tok * (isY ? ldp_b : ldq_b) + (size_t)c * 16);
  }
  const uint32_t patch = lds0 + RING + wave * 4096;
  uint32_t pw[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) pw[g] = patch + frow * 128 + (((2 * g + fhalf) ^ (frow & 7)) << 4);
  const uint32_t pr = patch + (lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) << 4);
  volatile __attribute__((address_space(3))) uint32_t* const q_word =
      reinterpret_cast<volatile __attribute__((address_space(3))) uint32_t*>((lds_void*)smem) + RING / 4;  // (wave 0's patch: idle until the tile's store)

  // ---- segments: (first stage, stages).  Plain: one per tile, the same range for every tile.  Queue: the chunks this workgroup draws.
  int s_cur = dyn ? sidx * per_wg : 0, s_nxt = 0x7fffffff;  // queue mode: the consumer's chunk and the one after it (known from phase 3 of s_cur's second stage on)
  auto seg_first = [&](int c) { return dyn ? chunk_begin(c) : 0; };
  auto seg_len = [&](int c) { return dyn ? chunk_begin(c + 1) - chunk_begin(c) : nst; };

  // ---- the staging stream
  int c_tile = first, c_st = 0, c_len = seg_len(s_cur);
  uint32_t c_buf = 0;
  bool c_live = c_tile < ntiles && c_len > 0;
  const char *c_gp, *c_gq;
  auto seg_base = [&](int t, int stage0, const char*& gp, const char*& gq) {
    const int ty = t / ntx, tx = t - ty * ntx;
    gp = gp0 + (size_t)ty * G::BM * 2 + (size_t)stage0 * BKT * ldp_b;
    gq = gq0 + (size_t)tx * G::BN * 2 + (size_t)stage0 * BKT * ldq_b;
  };
  seg_base(c_tile, seg_first(s_cur), c_gp, c_gq);
  auto stage_slot = [&](auto M) {
    constexpr int m = decltype(M)::value;
    const char* sb = (m & 1) == 0 ? c_gp : c_gq;
    const uint32_t dst = lds0 + c_buf + (uint32_t)(((m & 1) ? G::YB : 0) + (8 * (m >> 1) + wave) * 1024);
    const uint32_t vo = voff[m];
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(vo), "s"(sb), "s"(dst) : "memory");
  };
  auto advance = [&]() {
    c_buf = STAGE - c_buf;
    c_gp += (size_t)BKT * ldp_b;
    c_gq += (size_t)BKT * ldq_b;
    if (++c_st == c_len) {
      c_st = 0;
      if (dyn) {  // the consumer is in the chunk the cursor leaves: its next chunk is the cursor's
        c_live = s_nxt < nch;
        if (c_live) {
          c_len = seg_len(s_nxt);
          seg_base(c_tile, seg_first(s_nxt), c_gp, c_gq);
        }
      } else {
        c_tile += G8;
        c_live = c_tile < ntiles;
        if (c_live) seg_base(c_tile, 0, c_gp, c_gq);
      }
    }
  };
  auto stage_run = [&](auto P0, auto N) {
    static_for<decltype(N)::value>([&](auto I) {
      constexpr int m = (decltype(P0)::value + decltype(I)::value) % PPW;
      if (c_live) stage_slot(std::integral_constant<int, m>{});
      if constexpr (m == PPW - 1) {
        if (c_live) advance();
      }
    });
  };
  stage_run(std::integral_constant<int, 0>{}, std::integral_constant<int, S::AHEAD>{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  uint32_t r_buf = 0;
  // ---- queue mode, wave 0: the claim protocol (see the top of the kernel).  The SGPR an atomic returns into is defined at the top of a phase and
  // consumed behind that SAME phase's barrier + lgkmcnt(0): one variable per phase body, no join of two claims, no loop back-edge in between.
  // The compiler believes an asm result is there at once and is free to copy the register before the value has landed (a version that let a
  // claim fly into the next phase got exactly such copies): tools/check_satomic.py holds the listing to it -- no path from an s_atomic_add
  // reads or writes its destination in front of an s_waitcnt lgkmcnt(0); run it after touching this kernel or the toolchain.
  const unsigned* const q_row = dyn ? p.queue + 1 + bx * p.split_k : nullptr;  // this tile's words, one per split ([0] of the slot counts finished workgroups)
  bool q_need = false, own_done = false;  // this chunk's successor is still to be found; the own share is used up (persistent)
  int q_pick = -1, q_win = 0;             // the split to steal from next (-1: look first); the window of eight words the next look covers (persistent)
  auto q_publish = [&](int chunk) {
    q_word[0] = (uint32_t)chunk;  // (every lane of wave 0 stores the same word: no per-lane branch inside the scalar protocol)
    q_need = false;
  };
  // what this phase does for the search: 0 claim on the own word, 1 claim on q_pick's, 2 look at window q_win, -1 nothing left anywhere
  auto q_action = [&]() -> int {
    if (!own_done) return 0;
#ifdef OSUD_WGQ_NOSTEAL
    return -1;
#endif
    if (q_pick >= 0) return 1;
    return q_win * 8 < p.split_k ? 2 : -1;
  };
  auto q_land_claim = [&](int on, uint32_t word) {  // the word before the add
    const int lo = (int)(word & 0xffffu), hi = (int)(word >> 16);
    if (1 + lo <= per_wg - 1 - hi) q_publish(on * per_wg + (on == sidx ? 1 + lo : per_wg - 1 - hi));
    else if (on == sidx) own_done = true;
    else q_pick = -1;  // emptied meanwhile: look again
  };
  auto q_land_look = [&](const u32x8& w) {  // eight words of the tile's row from split 8 * q_win: the split with the most unclaimed chunks
    int best = -1, most = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int split = 8 * q_win + j;
      const int left = per_wg - 1 - (int)(w[j] >> 16) - (int)(w[j] & 0xffffu);
      if (split < p.split_k && split != sidx && left > most) {
        most = left;
        best = split;
      }
    }
    if (best >= 0) q_pick = best;
    else ++q_win;  // nothing in these eight, for good (counters only grow)
  };
  for (int tile = first; tile < ntiles; tile += G8) {
    const int ty = tile / ntx, tx = tile - ty * ntx;
    f32x16 acc[RY][RX];
#pragma unroll
    for (int i = 0; i < RY; ++i)
#pragma unroll
      for (int j = 0; j < RX; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    TFrag<RY, RX> f;
    if (grp == 1) __builtin_amdgcn_s_barrier();  // the stagger
    bool tile_first = true;
    do {  // the tile's segments (plain mode: one)
      const int len = seg_len(s_cur);
      for (int st = 0; st < len; ++st) {
        const bool counted = st > 0 || !tile_first;
        const bool draw = dyn && st < 2 && wave == 0;  // wave 0, this chunk's first two stages: the claim for the chunk after it
        if (draw && st == 0) q_need = true;
        static_for<NPH>([&](auto PH) {
          constexpr int P = decltype(PH)::value;
          // queue mode, wave 0: this phase's claim, issued here and found back behind this phase's own barrier + lgkmcnt(0)
          int q_on = -1;
          uint32_t q_ret = 0;
          int q_act = -1;
          u32x8 q_look;
#ifndef OSUD_WGQ_NOCLAIM
          if (draw && q_need) {
            q_act = q_action();
            if (q_act < 0) q_publish(0x7fffffff);
            else if (q_act == 2) {
              const unsigned* a = q_row + 8 * q_win;
              asm volatile("s_load_dwordx8 %0, %1, 0x0 glc" : "=s"(q_look) : "s"(a) : "memory");
            } else {
              q_on = q_act == 0 ? sidx : q_pick;
              q_ret = q_act == 0 ? 1u : 0x10000u;
              const unsigned* a = q_row + q_on;
              asm volatile("s_atomic_add %0, %1, 0x0 glc" : "+s"(q_ret) : "s"(a) : "memory");
            }
          }
#else  // (tuning builds: every workgroup walks its own chunks, no claims, no steals -- what the chunking alone costs)
          if (draw && q_need) q_publish((s_cur + 1) % per_wg != 0 ? s_cur + 1 : 0x7fffffff);
#endif
          read_frags<RY, RX, P, G::ROWY, G::ROWX>(f, ya, xa, r_buf);
          stage_run(std::integral_constant<int, S::AHEAD + ph_issued_before<S>(P)>{}, std::integral_constant<int, S::cnt[P]>{});
          constexpr int W = ph_wait<S>(P);
          if (!c_live) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          else if (counted) wait_vmcnt<W>();
          __builtin_amdgcn_sched_barrier(0);
          __builtin_amdgcn_s_barrier();
          OSUD_WG_WAIT(0);
          if (q_act >= 0) {  // (wave 0, a claim or a look in flight: back with the wait above)
            if (q_act == 2) {
              asm volatile("" : "+s"(q_look));
              q_land_look(q_look);
            } else {
              asm volatile("" : "+s"(q_ret));
              q_land_claim(q_on, q_ret);
            }
            if (q_need && ((P >= 1 && st == 1) || q_action() < 0)) q_publish(0x7fffffff);  // out of time (six operations), or nothing left anywhere
          }
          if constexpr (P == 3) {  // (wave 0 published behind the wait of the second stage's phase 1 at the latest: two phases = four barriers ago)
            if (dyn && st == 1) {
              const int c = __builtin_amdgcn_readfirstlane((int)q_word[0]);
              s_nxt = c < nch ? c : 0x7fffffff;
            }
          }
          __builtin_amdgcn_s_setprio(1);
          mma_frags<RY, RX>(acc, f);
          __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
          const bool last = st == len - 1 && (!dyn || s_nxt >= nch);  // (the tile's last stage)
          if (P != NPH - 1 || !last || grp == 0) __builtin_amdgcn_s_barrier();
        });
        r_buf = STAGE - r_buf;
      }
      tile_first = false;
      s_cur = dyn ? s_nxt : 0x7fffffff;
      s_nxt = 0x7fffffff;
    } while (dyn && s_cur < nch);
    s_cur = 0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- the tile's partial slab (as wgrad_kernel's store_tile)
#pragma unroll
    for (int i = 0; i < RY; ++i) {
      const int y0 = ty * G::BM + wy * RY * 32 + i * 32 + (lane >> 3);
#pragma unroll
      for (int j = 0; j < RX; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v;
          v[0] = acc[i][j][4 * g + 0]; v[1] = acc[i][j][4 * g + 1]; v[2] = acc[i][j][4 * g + 2]; v[3] = acc[i][j][4 * g + 3];
          ds_write16(pw[g], v);
        }
        f32x4 t[4];
        t[0] = ds_read16f<0>(pr);
        t[1] = ds_read16f<1024>(pr);
        t[2] = ds_read16f<2048>(pr);
        t[3] = ds_read16f<3072>(pr);
        OSUD_WG_WAIT(0);
        const int x = tx * G::BN + wx * RX * 32 + j * 32 + 4 * (lane & 7);
        if (x < p.Nx) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (y0 + 8 * q < p.Ny) store4(outp + (size_t)(y0 + 8 * q) * p.Nx + x, t[q][0], t[q][1], t[q][2], t[q][3]);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if (dyn) break;  // (one tile per workgroup)
  }
  if (dyn && wave == 0 && lane == 0) {  // the last workgroup out re-arms the counters
    const unsigned done = __hip_atomic_fetch_add(p.queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done == gridDim.x * gridDim.y - 1)
      for (int i = 0; i < 1 + (int)(gridDim.x * gridDim.y); ++i) __hip_atomic_store(p.queue + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}


// ---------------------------------------------------------------------------------------------------------------------------
// The 256 x 192 (NARROW_Y = false) and 192 x 256 (NARROW_Y = true) geometries in the phased schedule: DiT-XL's widths (1152 = 6 x 192,
// 3456 = 18 x 192, 4608 = 24 x 192) tile exactly with a 192-wide side, where the 256 x 256 kernel pads 10 % of its tiles' area and fills
// two thirds of the chip.  Same staging image, transposing reads and MFMAs as wgrad_kernel<4, 2, 2, 3> / <2, 4, 3, 2>, every accumulator
// sees its tokens in the same order: bit-identical partial slabs.  A phase is one k sub-step (16 tokens of both operands: 10 transposing
// reads, 6 MFMAs = 192 pipe cycles per wave).  The wide part (512-byte token rows) splits into one 8-piece slot per sub-step; the narrow part
// (384-byte rows: 24 pieces per stage, 6 per sub-step) into three 8-piece slots Na Nb Nc, whose pieces straddle token rows -- the per-lane
// source offsets are computed per 16-byte chunk -- so sub-step s reads Na | Na Nb | Nb Nc | Nc.  Stream order W0 Na W1 Nb W2 Nc W3: need
// {1, 3, 5, 6}; seven pieces per wave and stage, ten in flight, vmcnt(8) per phase.
struct Wg192Sched {
  static constexpr int NPH = 4, PPW = 7, AHEAD = 10;
  static constexpr int cnt[NPH] = {2, 2, 1, 2};
  static constexpr int need[NPH] = {1, 3, 5, 6};
  static constexpr int read_phase[PPW] = {0, 1, 1, 2, 2, 3, 3};
};

template <bool NARROW_Y> __global__ __launch_bounds__(512) void wgrad_phased192_kernel(WgradP p) {
  using G = typename std::conditional<NARROW_Y, WGeo<2, 4, 3, 2>, WGeo<4, 2, 2, 3>>::type;
  using S = Wg192Sched;
  static_assert(sched_ok<S>(), "phase table breaks a staging rule");
  constexpr int WX = NARROW_Y ? 4 : 2, RY = NARROW_Y ? 3 : 2, RX = NARROW_Y ? 2 : 3, NPH = S::NPH, PPW = S::PPW, STAGE = G::STAGE, RING = 2 * STAGE;
  static_assert(G::NSTAGE == 2 && G::PPW == PPW && G::ROWY == (NARROW_Y ? 384 : 512) && G::ROWX == (NARROW_Y ? 512 : 384) && RING + 8 * 4096 <= 160 * 1024, "geometry");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wy = wave / WX, wx = wave % WX, grp = wave >> 2;
  const int frow = lane & 31, fhalf = lane >> 5;
  const int ntx = (p.Nx + G::BN - 1) / G::BN, ntiles = ((p.Ny + G::BM - 1) / G::BM) * ntx;
  const int st_total = p.M / BKT;
  int sidx = blockIdx.y, bx = blockIdx.x;
  if (p.xcd_units) {  // (tile, split) units in split-major order, one contiguous run per XCD: see WgradP::xcd_units
    const int total = gridDim.x * gridDim.y, L = blockIdx.x + blockIdx.y * gridDim.x;
    const int q = total >> 3, r = total & 7, xcd = L & 7, idx = L >> 3;
    const int u = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    sidx = u / (int)gridDim.x;
    bx = u - sidx * (int)gridDim.x;
  }
  const int st_begin = (int)((long)sidx * st_total / p.split_k);
  const int nst = (int)((long)(sidx + 1) * st_total / p.split_k) - st_begin;
  const size_t ldp_b = (size_t)p.ldp * 2, ldq_b = (size_t)p.ldq * 2;
  const char* const gp0 = reinterpret_cast<const char*>(p.P) + (size_t)st_begin * BKT * ldp_b;
  const char* const gq0 = reinterpret_cast<const char*>(p.Q) + (size_t)st_begin * BKT * ldq_b;
  float* const outp = p.out + (size_t)sidx * p.split_stride;
  const int G8 = gridDim.x;
  int first;
  {
    const int b = blockIdx.x, q = G8 >> 3, r = G8 & 7, xcd = b & 7, idx = b >> 3;
    first = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    if (p.xcd_units) first = bx;
  }

  // fragment addresses (stage 0, sub-step 0), exactly wgrad_kernel's for this geometry
  const uint32_t lds0 = (uint32_t)(size_t)(lds_void*)smem;
  const int tok0 = 8 * fhalf + ((lane & 15) >> 2);
  const int swy = G::swz(G::ROWY, tok0), swx = G::swz(G::ROWX, tok0);
  const int fbyte = 32 * ((lane >> 4) & 1) + 8 * (lane & 3);
  uint32_t ya[RY], xa[RX];
#pragma unroll
  for (int i = 0; i < RY; ++i) ya[i] = lds0 + tok0 * G::ROWY + (((wy * RY + i) ^ swy) * 64) + fbyte;
#pragma unroll
  for (int j = 0; j < RX; ++j) xa[j] = lds0 + G::YB + tok0 * G::ROWX + (((wx * RX + j) ^ swx) * 64) + fbyte;
  // LDS-DMA: stream order W0 Na W1 Nb W2 Nc W3.  Even slot m = 2 s: piece 8 s + w of the WIDE part (token rows 16 s + 2 w, + 1); odd slot
  // m = 2 j + 1: piece 8 j + w of the NARROW part (1 KiB of 384-byte rows: per-chunk token / position).  P is the wide part unless NARROW_Y.
  uint32_t voff[PPW];
#pragma unroll
  for (int m = 0; m < PPW; ++m) {
    const bool wide = (m & 1) == 0;
    const bool isY = wide != NARROW_Y;                      // this slot belongs to the P part
    const int rowb = wide ? 512 : 384, lpr = rowb / 16;
    const int pb = (8 * (m >> 1) + wave) * 1024;            // byte offset inside the part
    const int cidx = pb / 16 + lane;
    const int tok = cidx / lpr, pos = cidx - tok * lpr;
    const int c = pos ^ (G::swz(rowb, tok) << 2);
    voff[m] = (uint32_t)((size_t)tok * (isY ? ldp_b : ldq_b) + (size_t)c * 16);
  }
  const uint32_t patch = lds0 + RING + wave * 4096;
  uint32_t pw[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) pw[g] = patch + frow * 128 + (((2 * g + fhalf) ^ (frow & 7)) << 4);
  const uint32_t pr = patch + (lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) << 4);

  // ---- the staging stream
  int c_tile = first, c_st = 0;
  uint32_t c_buf = 0;
  bool c_live = c_tile < ntiles && nst > 0;
  const char *c_gp, *c_gq;
  auto tile_base = [&](int t, const char*& gp, const char*& gq) {
    const int ty = t / ntx, tx = t - ty * ntx;
    gp = gp0 + (size_t)ty * G::BM * 2;
    gq = gq0 + (size_t)tx * G::BN * 2;
  };
  tile_base(c_tile, c_gp, c_gq);
  auto stage_slot = [&](auto M) {
    constexpr int m = decltype(M)::value;
    constexpr bool wide = (m & 1) == 0, isY = wide != NARROW_Y;
    const char* sb = isY ? c_gp : c_gq;
    const uint32_t dst = lds0 + c_buf + (uint32_t)((isY ? 0 : G::YB) + (8 * (m >> 1) + wave) * 1024);
    const uint32_t vo = voff[m];
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(vo), "s"(sb), "s"(dst) : "memory");
  };
  auto advance = [&]() {
    c_buf = STAGE - c_buf;
    c_gp += (size_t)BKT * ldp_b;
    c_gq += (size_t)BKT * ldq_b;
    if (++c_st == nst) {
      c_st = 0;
      c_tile += G8;
      c_live = c_tile < ntiles;
      if (c_live) tile_base(c_tile, c_gp, c_gq);
    }
  };
  auto stage_run = [&](auto P0, auto N) {
    static_for<decltype(N)::value>([&](auto I) {
      constexpr int m = (decltype(P0)::value + decltype(I)::value) % PPW;
      if (c_live) stage_slot(std::integral_constant<int, m>{});
      if constexpr (m == PPW - 1) {
        if (c_live) advance();
      }
    });
  };
  stage_run(std::integral_constant<int, 0>{}, std::integral_constant<int, S::AHEAD>{});
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  uint32_t r_buf = 0;
  for (int tile = first; tile < ntiles; tile += G8) {
    const int ty = tile / ntx, tx = tile - ty * ntx;
    f32x16 acc[RY][RX];
#pragma unroll
    for (int i = 0; i < RY; ++i)
#pragma unroll
      for (int j = 0; j < RX; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    TFrag<RY, RX> f;
    if (grp == 1) __builtin_amdgcn_s_barrier();  // the stagger
    for (int st = 0; st < nst; ++st) {
      const bool last = st == nst - 1;
      static_for<NPH>([&](auto PH) {
        constexpr int P = decltype(PH)::value;
        read_frags<RY, RX, P, G::ROWY, G::ROWX>(f, ya, xa, r_buf);
        stage_run(std::integral_constant<int, S::AHEAD + ph_issued_before<S>(P)>{}, std::integral_constant<int, S::cnt[P]>{});
        constexpr int W = ph_wait<S>(P);
        if (!c_live) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (st > 0 || !ph_wait_predrained<S>(P)) wait_vmcnt<W>();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        OSUD_WG_WAIT(0);
        __builtin_amdgcn_s_setprio(1);
        mma_frags<RY, RX>(acc, f);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        if (P != NPH - 1 || !last || grp == 0) __builtin_amdgcn_s_barrier();
      });
      r_buf = STAGE - r_buf;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- the tile's partial slab (as wgrad_kernel's store_tile)
#pragma unroll
    for (int i = 0; i < RY; ++i) {
      const int y0 = ty * G::BM + wy * RY * 32 + i * 32 + (lane >> 3);
#pragma unroll
      for (int j = 0; j < RX; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 v;
          v[0] = acc[i][j][4 * g + 0]; v[1] = acc[i][j][4 * g + 1]; v[2] = acc[i][j][4 * g + 2]; v[3] = acc[i][j][4 * g + 3];
          ds_write16(pw[g], v);
        }
        f32x4 t[4];
        t[0] = ds_read16f<0>(pr);
        t[1] = ds_read16f<1024>(pr);
        t[2] = ds_read16f<2048>(pr);
        t[3] = ds_read16f<3072>(pr);
        OSUD_WG_WAIT(0);
        const int x = tx * G::BN + wx * RX * 32 + j * 32 + 4 * (lane & 7);
        if (x < p.Nx) {
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (y0 + 8 * q < p.Ny) store4(outp + (size_t)(y0 + 8 * q) * p.Nx + x, t[q][0], t[q][1], t[q][2], t[q][3]);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// Weight gradients on e4m3 operands (fp8 training, BASELINE config 5):  out[y][x] = inv_p * inv_q * sum_m P8[m][y0 + y] * Q8[m][x0 + x]
// P8 / Q8 are the e4m3 twins of the gradient / activation tensors (token-major, one byte per element, each quantised with its
// slot's delayed scale; inv_p / inv_q = the slots' 1 / scale, device scalars).  Same structure as wgrad_kernel -- stages by
// LDS-DMA, counted waits, one barrier per stage, 8 waves of 128 x 64 outputs, split over the token axis into fp32 partial slabs --
// with 128 tokens per 64 KiB stage and the block-scaled K = 64 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, unit scales: twice the
// bf16 rate).  The MFMA wants 32 consecutive tokens of one feature per lane; gfx950's ds_read_b64_tr_b8 delivers 8 of them from
// the token-major tile: a 16-lane group reads an 8-token x 16-feature block (lanes 2j, 2j + 1 point at the two 8-byte halves of
// token j's 16 bytes) and lane i receives feature i's 8 tokens (semantics pinned on hardware by tools/probes/tr8_probe.hip); four
// such reads make a lane's operand.  Token rows are 256 bytes (256 features); the 16-byte chunk index of a row is XOR-swizzled
// with (token & 7) << 1 on the source side of the LDS-DMA, so the 8 token rows of a transposing read -- and the neighbouring
// feature block the other 16-lane group of the same cycle reads -- fall on 16 different chunks: conflict free.
constexpr int BKT8 = 128;  // tokens per stage
template <int OFF> __device__ __forceinline__ u32x2 ds_read_tr8(uint32_t addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b8 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
  return v;
}
typedef int i32x8w __attribute__((ext_vector_type(8)));

struct Wgrad8P {
  const uint8_t* P;
  const uint8_t* Q;
  int ldp, ldq;   // elements = bytes
  int Ny, Nx, M;
  float* out;     // [splits][Ny][Nx] fp32 (already multiplied by inv_p * inv_q)
  int split_k;
  size_t split_stride;
  const float* inv_p;
  const float* inv_q;
};

__global__ __launch_bounds__(512) void wgrad8_kernel(Wgrad8P p) {
  constexpr int WX = 4, RY = 4, RX = 2, BM = 256, BN = 256, NW = 8, ROW = 256;
  constexpr int YB = BKT8 * ROW, STAGE = 2 * YB, NSTAGE = 2, PPW = STAGE / 1024 / NW;  // 32 KiB + 32 KiB per stage, 8 pieces per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wy = wave / WX, wx = wave % WX;
  const int frow = lane & 31, fhalf = lane >> 5;
  const int ntx = (p.Nx + BN - 1) / BN, ntiles = ((p.Ny + BM - 1) / BM) * ntx;
  const int st_total = p.M / BKT8;
  // units in split-major order, one contiguous run per XCD (see WgradP::xcd_units)
  int sidx, bx;
  {
    const int total = gridDim.x * gridDim.y, L = blockIdx.x + blockIdx.y * gridDim.x;
    const int q = total >> 3, r = total & 7, xcd = L & 7, idx = L >> 3;
    const int u = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    sidx = u / (int)gridDim.x;
    bx = u - sidx * (int)gridDim.x;
  }
  const int st_begin = (int)((long)sidx * st_total / p.split_k);
  const int nst = (int)((long)(sidx + 1) * st_total / p.split_k) - st_begin;
  const int ty = bx / ntx, tx = bx - ty * ntx;
  const size_t ldp_b = (size_t)p.ldp, ldq_b = (size_t)p.ldq;
  const char* gp0 = reinterpret_cast<const char*>(p.P) + (size_t)st_begin * BKT8 * ldp_b + (size_t)ty * BM;
  const char* gq0 = reinterpret_cast<const char*>(p.Q) + (size_t)st_begin * BKT8 * ldq_b + (size_t)tx * BN;
  float* outp = p.out + (size_t)sidx * p.split_stride;

  // fragment addresses: lane (idx = lane & 15, fblock = (lane >> 4) & 1, khalf = lane >> 5) reads, for k-step ks and quarter q,
  // token ks * 64 + khalf * 32 + 8 q + (idx >> 1), bytes (idx & 1) * 8 .. + 7 of 16-byte chunk (block * 2 + fblock) ^ ((idx >> 1) << 1)
  const uint32_t lds0 = (uint32_t)(size_t)(lds_void*)smem;
  const int idx16 = lane & 15, fblock = (lane >> 4) & 1, trow = idx16 >> 1;
  const uint32_t lane_base = (uint32_t)((fhalf * 32 + trow) * ROW + (idx16 & 1) * 8);
  uint32_t ya[RY], xa[RX];
#pragma unroll
  for (int i = 0; i < RY; ++i) ya[i] = lds0 + lane_base + ((((wy * RY + i) * 2 + fblock) ^ (trow << 1)) << 4);
#pragma unroll
  for (int j = 0; j < RX; ++j) xa[j] = lds0 + YB + lane_base + ((((wx * RX + j) * 2 + fblock) ^ (trow << 1)) << 4);

  // LDS-DMA: piece = 1 KiB = 4 token rows; lane l -> row (l >> 4) of the piece, LDS position l & 15, source chunk = position ^ ((token & 7) << 1)
  uint32_t dma_off[PPW];
#pragma unroll
  for (int q = 0; q < PPW; ++q) {
    const int piece = wave * PPW + q;
    const bool isY = piece * 1024 < YB;
    const int pb = isY ? piece * 1024 : piece * 1024 - YB;
    const int tok = pb / ROW + (lane >> 4), pos = lane & 15;
    const int c = pos ^ ((tok & 7) << 1);
    dma_off[q] = (uint32_t)((size_t)tok * (isY ? ldp_b : ldq_b) + c * 16);
  }
  const uint32_t patch = lds0 + NSTAGE * STAGE + wave * 4096;
  uint32_t pw[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) pw[g] = patch + frow * 128 + (((2 * g + fhalf) ^ (frow & 7)) << 4);
  const uint32_t pr = patch + (lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) << 4);

  int issued = 0, consumed = 0, ic_st = 0;
  auto issue_next = [&]() {
    if (ic_st >= nst) return;
    const char* gp = gp0 + (size_t)ic_st * BKT8 * ldp_b;
    const char* gq = gq0 + (size_t)ic_st * BKT8 * ldq_b;
    const uint32_t stage_lds = lds0 + (uint32_t)((issued % NSTAGE) * STAGE);
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
      const int piece = wave * PPW + q;
      const char* sbase = (piece * 1024 < YB) ? gp : gq;
      const uint32_t dst = stage_lds + (uint32_t)__builtin_amdgcn_readfirstlane(piece * 1024);
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(dma_off[q]), "s"(sbase), "s"(dst) : "memory");
    }
    ++issued;
    ++ic_st;
  };
  issue_next();

  f32x16 acc[RY][RX];
#pragma unroll
  for (int i = 0; i < RY; ++i)
#pragma unroll
    for (int j = 0; j < RX; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  for (int st = 0; st < nst; ++st) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    issue_next();
    const uint32_t so = (uint32_t)((consumed % NSTAGE) * STAGE);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {  // two K = 64 steps per 128-token stage
      u32x2 yf[RY][4], xf[RX][4];
#pragma unroll
      for (int i = 0; i < RY; ++i) {
        if (ks == 0) { yf[i][0] = ds_read_tr8<0 * ROW>(ya[i] + so); yf[i][1] = ds_read_tr8<8 * ROW>(ya[i] + so); yf[i][2] = ds_read_tr8<16 * ROW>(ya[i] + so); yf[i][3] = ds_read_tr8<24 * ROW>(ya[i] + so); }
        else { yf[i][0] = ds_read_tr8<64 * ROW>(ya[i] + so); yf[i][1] = ds_read_tr8<72 * ROW>(ya[i] + so); yf[i][2] = ds_read_tr8<80 * ROW>(ya[i] + so); yf[i][3] = ds_read_tr8<88 * ROW>(ya[i] + so); }
      }
#pragma unroll
      for (int j = 0; j < RX; ++j) {
        if (ks == 0) { xf[j][0] = ds_read_tr8<0 * ROW>(xa[j] + so); xf[j][1] = ds_read_tr8<8 * ROW>(xa[j] + so); xf[j][2] = ds_read_tr8<16 * ROW>(xa[j] + so); xf[j][3] = ds_read_tr8<24 * ROW>(xa[j] + so); }
        else { xf[j][0] = ds_read_tr8<64 * ROW>(xa[j] + so); xf[j][1] = ds_read_tr8<72 * ROW>(xa[j] + so); xf[j][2] = ds_read_tr8<80 * ROW>(xa[j] + so); xf[j][3] = ds_read_tr8<88 * ROW>(xa[j] + so); }
      }
      OSUD_WG_WAIT(0);
#pragma unroll
      for (int i = 0; i < RY; ++i) {
        i32x8w yv;
#pragma unroll
        for (int q = 0; q < 4; ++q) { yv[2 * q] = (int)yf[i][q][0]; yv[2 * q + 1] = (int)yf[i][q][1]; }
#pragma unroll
        for (int j = 0; j < RX; ++j) {
          i32x8w xv;
#pragma unroll
          for (int q = 0; q < 4; ++q) { xv[2 * q] = (int)xf[j][q][0]; xv[2 * q + 1] = (int)xf[j][q][1]; }
          acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(xv, yv, acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
      }
    }
    ++consumed;
  }
  const float dq = p.inv_p[0] * p.inv_q[0];
#pragma unroll
  for (int i = 0; i < RY; ++i) {
    const int y0 = ty * BM + wy * RY * 32 + i * 32 + (lane >> 3);
#pragma unroll
    for (int j = 0; j < RX; ++j) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 v;
        v[0] = acc[i][j][4 * g + 0] * dq; v[1] = acc[i][j][4 * g + 1] * dq; v[2] = acc[i][j][4 * g + 2] * dq; v[3] = acc[i][j][4 * g + 3] * dq;
        ds_write16(pw[g], v);
      }
      f32x4 t[4];
      t[0] = ds_read16f<0>(pr);
      t[1] = ds_read16f<1024>(pr);
      t[2] = ds_read16f<2048>(pr);
      t[3] = ds_read16f<3072>(pr);
      OSUD_WG_WAIT(0);
      const int x = tx * BN + wx * RX * 32 + j * 32 + 4 * (lane & 7);
      if (x < p.Nx) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (y0 + 8 * q < p.Ny) store4(outp + (size_t)(y0 + 8 * q) * p.Nx + x, t[q][0], t[q][1], t[q][2], t[q][3]);
      }
    }
  }
}

// column sums of a bf16 [M][N] matrix: out[c] = sum_m a[m][c]   (bias gradients).  HBM-bound: a workgroup sweeps
// 256 rows x 256 columns with 16-byte loads (32 lanes = one 512-byte row segment, 8 rows per pass, 4 passes in
// flight), combines its 8 row groups through LDS and writes its share to out[row block][N]: the launcher's fixed-order
// column pass adds the row blocks (float atomics made the sum depend on the order of arrival).
// QUANT (fp8 training): the same pass also writes the matrix's e4m3 twin (value x slot[0], the slot's delayed scale) and records this
// step's amax in slot[2] (one atomic per workgroup) -- dqkv is read once for its bias gradient AND its quantisation.
constexpr int CS_ROWS = 256;
template <bool QUANT>
__global__ __launch_bounds__(256) void colsum_bf16_kernel(const bf16_t* __restrict__ a, int ld, int M, int N,
                                                          float* __restrict__ out, fp8_t* __restrict__ q8 = nullptr,
                                                          float* __restrict__ slot = nullptr) {
  __shared__ float part[8][256 + 4];
  __shared__ float red[4];
  const float q_scale = QUANT ? slot[0] : 1.0f;
  float amax = 0.f;
  const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
  const int c = blockIdx.x * 256 + cl * 8;
  const int r0 = blockIdx.y * CS_ROWS, r1 = r0 + CS_ROWS < M ? r0 + CS_ROWS : M;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (c < N) {
    for (int r = r0 + rg; r < r1; r += 32) {
      uint4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int rr = r + 8 * u;
        v[u] = rr < r1 ? *reinterpret_cast<const uint4*>(a + (size_t)rr * ld + c) : make_uint4(0, 0, 0, 0);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float e[8];
        e[0] = __uint_as_float(v[u].x << 16); e[1] = __uint_as_float(v[u].x & 0xffff0000u);
        e[2] = __uint_as_float(v[u].y << 16); e[3] = __uint_as_float(v[u].y & 0xffff0000u);
        e[4] = __uint_as_float(v[u].z << 16); e[5] = __uint_as_float(v[u].z & 0xffff0000u);
        e[6] = __uint_as_float(v[u].w << 16); e[7] = __uint_as_float(v[u].w & 0xffff0000u);
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] += e[j];
        if constexpr (QUANT) {
          const int rr = r + 8 * u;
          if (rr < r1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              amax = fmaxf(amax, fabsf(e[j]));
              e[j] *= q_scale;
            }
            if (q8 != nullptr) store8(q8 + (size_t)rr * ld + c, e);
          }
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) part[rg][cl * 8 + j] = s[j];
  __syncthreads();
  const int t = threadIdx.x, col = blockIdx.x * 256 + t;
  if (col < N) {
    float sum = 0.f;
#pragma unroll
    for (int g = 0; g < 8; ++g) sum += part[g][t];
    out[(size_t)blockIdx.y * N + col] = sum;
  }
  if constexpr (QUANT) {
    amax = wave_max(amax);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (threadIdx.x == 0) {
      amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
      if (amax > 0.f) atomicMax(reinterpret_cast<unsigned*>(slot) + 2, __float_as_uint(amax));
    }
  }
}

// per-device state, as in gemm.hip
constexpr int kMaxDevices = 16;
int cur_device_w() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
  return dev;
}
int num_cus_w() {
  static int n[kMaxDevices] = {};
  const int dev = cur_device_w();
  if (n[dev] == 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) n[dev] = prop.multiProcessorCount;
    if (n[dev] <= 0) n[dev] = 256;
  }
  return n[dev];
}

// counter sets of the chunk queues, re-armed by the last workgroup out.  wgrad_kernel's queue mode: [tile] tickets, [63] finished workgroups;
// wgrad_phased_kernel's: [0] finished workgroups, [1 + tile * splits + split] the split's claim word (tiles x splits <= compute units)
constexpr int kQueueSlots = 64, kQueueWords = 576;
unsigned* g_queue_pool[kMaxDevices] = {};
unsigned* queue_slot() {
  static std::atomic<unsigned> seq{0};
  const int dev = cur_device_w();
  if (!g_queue_pool[dev]) {
    unsigned* pool = nullptr;
    if (hipMalloc(&pool, kQueueSlots * kQueueWords * sizeof(unsigned)) != hipSuccess) return nullptr;
    if (hipMemset(pool, 0, kQueueSlots * kQueueWords * sizeof(unsigned)) != hipSuccess) return nullptr;
    g_queue_pool[dev] = pool;
  }
  return g_queue_pool[dev] + kQueueWords * (seq.fetch_add(1) % kQueueSlots);
}

template <int WY, int WX, int RY, int RX> int launch_wg(const WgradP& p, hipStream_t st) {
  using G = WGeo<WY, WX, RY, RX>;
  const size_t lds = (size_t)G::NSTAGE * G::STAGE + (size_t)G::NW * 4096;
  static bool attr_set = false;
  if (!attr_set) {
    OSUD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_kernel<WY, WX, RY, RX>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  const int ntiles = ((p.Ny + G::BM - 1) / G::BM) * ((p.Nx + G::BN - 1) / G::BN);
  int grid = num_cus_w() / p.split_k;
  if (grid < 1) grid = 1;
  if (grid > ntiles || p.split_k > 1) grid = ntiles;  // with splits: one tile per workgroup
  hipLaunchKernelGGL((wgrad_kernel<WY, WX, RY, RX>), dim3(grid, p.split_k), dim3(G::NT), lds, st, p);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

}  // namespace

// out[Ny][Nx] (fp32, ld = Nx) = P[:, 0:Ny]^T . Q[:, 0:Nx] over M tokens; `ws` holds the split-K partial slabs.
int launch_wgrad_tr(const void* P, int ldp, const void* Q, int ldq, int Ny, int Nx, int M, float* out, float* ws,
                    size_t ws_elems, hipStream_t st) {
  OSUD_CHECK_ARG(Ny % 128 == 0 && Nx % 128 == 0 && M % BKT == 0 && ldp % 8 == 0 && ldq % 8 == 0,
                 "wgrad: Ny=%d Nx=%d must be multiples of 128, M=%d of 64", Ny, Nx, M);
  // 256x256 tiles (8 waves) also for odd multiples of 128 -- DiT-XL's 1152 and 3456 -- with half-empty edge tiles, as long as
  // the padding stays below 25 % of the work: the 128x128 geometry (4 waves) runs at ~0.55x the 256-wide kernel's rate
  // (profiles/r02_xl_*: 582 vs 1033 TFLOP/s), which made the weight gradients 30 % of a DiT-XL training step
  const int cus = num_cus_w();
  const int t256 = ((Ny + 255) / 256) * ((Nx + 255) / 256);
  const bool big = Ny >= 256 && Nx >= 256 && (double)t256 * 65536.0 <= 1.25 * (double)Ny * (double)Nx;
  // ... and 256x192 / 192x256 tiles where a side is a multiple of 192 (1152 = 6 x 192, 3456 = 18 x 192, 4608 = 24 x 192) and
  // that fills the chip better: useful fraction of the tiles x fraction of the CUs that one round of tiles x splits occupies
  // (fc1 of DiT-XL, 4608 x 1152: 90 padded 256-wide tiles x 2 splits = 180 of 256 CUs at 90 % useful; 108 exact 256x192 tiles
  // x 2 = 216 CUs).  The narrower tile reads 11 % more LDS bytes per MFMA, so it has to win by more than that.
  auto score = [&](int bm, int bn, int* tiles_out) {
    const int t = ((Ny + bm - 1) / bm) * ((Nx + bn - 1) / bn);
    int s = cus / t;
    if (s > 32) s = 32;
    while (s > 1 && (M / BKT) / s < 8) --s;
    if (s < 1) s = 1;
    const int rounds = (t * s + cus - 1) / cus;
    *tiles_out = t;
    return (double)Ny * Nx / ((double)t * bm * bn) * ((double)t * s / ((double)rounds * cus));
  };
  int geo = big ? 0 : 3, tiles = big ? t256 : (Ny / 128) * (Nx / 128);  // 0: 256x256, 1: 256x192, 2: 192x256, 3: 128x128
  if (big && !gemm_dynamic_tiles_on()) {
    int t0 = 0, t1 = 0, t2 = 0;
    const double s0 = score(256, 256, &t0);
    const double s1 = Nx % 192 == 0 ? score(256, 192, &t1) : 0.0, s2 = Ny % 192 == 0 ? score(192, 256, &t2) : 0.0;
    const double margin = OSUD_WG192_MARGIN;
    if (s1 >= s2 && s1 > margin * s0) { geo = 1; tiles = t1; }
    else if (s2 > s1 && s2 > margin * s0) { geo = 2; tiles = t2; }
  }
  const int stages = M / BKT;
  // fill the chip in ONE round: splits = CUs / tiles, each split at least 8 stages (512 tokens)
  int S = cus / tiles;
  if (S > 32) S = 32;
  while (S > 1 && (stages / S < 8 || (size_t)S * Ny * Nx > ws_elems)) --S;
  if (S < 1) S = 1;
  WgradP p{};
  p.P = (const bf16_t*)P; p.Q = (const bf16_t*)Q; p.ldp = ldp; p.ldq = ldq; p.Ny = Ny; p.Nx = Nx; p.M = M;
  p.split_k = S; p.split_stride = (size_t)Ny * Nx; p.out = S > 1 ? ws : out;
  // the phased schedule (same bits) where it exists; its queue mode needs chunks of at least four stages (a chunk's claim is settled in its
  // first two stages and the stream needs the next chunk a stage and a half before the current one ends)
  bool phased = geo == 0 && opt(OPT_GEMM_LOOP) != 0 && stages / S >= 2;
  if (geo == 0 && S > 1 && gemm_dynamic_tiles_on()) {  // the GPU is shared with collectives: queue the K-chunks
#ifndef OSUD_WGQ_PERWG
#define OSUD_WGQ_PERWG 8
#endif
    const int share = stages / S, per_wg = share / 4 < OSUD_WGQ_PERWG ? (share / 4 < 1 ? 1 : share / 4) : OSUD_WGQ_PERWG;  // chunks of >= 4 stages, at most 8 per workgroup (a claim costs a scalar-memory round trip)
    const bool ph_q = phased && per_wg >= 2 && stages / (S * per_wg) >= 4 && 1 + tiles * S <= kQueueWords;
    if (ph_q || tiles <= 62) {  // (wgrad_kernel's per-tile ticket queue has room for 62 tiles)
      p.chunk = S * per_wg;
      p.queue = queue_slot();
      phased = ph_q;
    }
  }
  // one unit per workgroup: XCD-contiguous unit order (WgradP::xcd_units) in the plain mode and in the phased kernel's queue mode
  p.xcd_units = (S > 1 && (p.queue == nullptr || phased)) ? 1 : 0;
  if (phased) {
    constexpr size_t lds = 2 * (size_t)WGeo<2, 4, 4, 2>::STAGE + 8 * 4096;
    static bool attr_set = false;
    if (!attr_set) {
      OSUD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_phased_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      OSUD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_phased_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr_set = true;
    }
    int grid = cus / p.split_k;
    if (grid < 1) grid = 1;
    if (grid > tiles || p.split_k > 1) grid = tiles;
    if (p.queue != nullptr) hipLaunchKernelGGL(wgrad_phased_kernel<true>, dim3(grid, p.split_k), dim3(512), lds, st, p);
    else hipLaunchKernelGGL(wgrad_phased_kernel<false>, dim3(grid, p.split_k), dim3(512), lds, st, p);
    OSUD_HIP(hipGetLastError());
  } else if (geo == 0) OSUD_TRY((launch_wg<2, 4, 4, 2>(p, st)));
  else if ((geo == 1 || geo == 2) && opt(OPT_GEMM_LOOP) != 0 && stages / S >= 2) {  // the 192-wide geometries in the phased schedule (same bits)
    constexpr size_t lds = 2 * (size_t)WGeo<4, 2, 2, 3>::STAGE + 8 * 4096;
    static_assert(WGeo<4, 2, 2, 3>::STAGE == WGeo<2, 4, 3, 2>::STAGE, "one LDS size for both");
    static bool attr_set = false;
    if (!attr_set) {
      OSUD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_phased192_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      OSUD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_phased192_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      attr_set = true;
    }
    int grid = cus / p.split_k;
    if (grid < 1) grid = 1;
    if (grid > tiles || p.split_k > 1) grid = tiles;
    if (geo == 1) hipLaunchKernelGGL(wgrad_phased192_kernel<false>, dim3(grid, p.split_k), dim3(512), lds, st, p);
    else hipLaunchKernelGGL(wgrad_phased192_kernel<true>, dim3(grid, p.split_k), dim3(512), lds, st, p);
    OSUD_HIP(hipGetLastError());
  } else if (geo == 1) OSUD_TRY((launch_wg<4, 2, 2, 3>(p, st)));
  else if (geo == 2) OSUD_TRY((launch_wg<2, 4, 3, 2>(p, st)));
  else OSUD_TRY((launch_wg<2, 2, 2, 2>(p, st)));
  if (S > 1) OSUD_TRY(launch_splitk_reduce(ws, S, (size_t)Ny * Nx, out, (size_t)Ny * Nx, st));
  return OSUD_OK;
}



// out[Ny][Nx] (fp32) = inv_p * inv_q * P8[:, 0:Ny]^T . Q8[:, 0:Nx] over M tokens, e4m3 operands (see wgrad8_kernel); `ws`: split-K slabs
int launch_wgrad8_tr(const void* P8, int ldp, const void* Q8, int ldq, int Ny, int Nx, int M, float* out, float* ws, size_t ws_elems,
                     const float* inv_p, const float* inv_q, hipStream_t st) {
  OSUD_CHECK_ARG(Ny % 128 == 0 && Nx % 128 == 0 && M % BKT8 == 0 && ldp % 16 == 0 && ldq % 16 == 0 && inv_p && inv_q,
                 "wgrad8: Ny=%d Nx=%d must be multiples of 128, M=%d of 128, leading dimensions of 16", Ny, Nx, M);
  const int cus = num_cus_w();
  const int tiles = ((Ny + 255) / 256) * ((Nx + 255) / 256);
  const int stages = M / BKT8;
  int S = cus / tiles;
  if (S > 32) S = 32;
  while (S > 1 && (stages / S < 4 || (size_t)S * Ny * Nx > ws_elems)) --S;
  if (S < 1) S = 1;
  Wgrad8P p{};
  p.P = (const uint8_t*)P8; p.Q = (const uint8_t*)Q8; p.ldp = ldp; p.ldq = ldq; p.Ny = Ny; p.Nx = Nx; p.M = M;
  p.split_k = S; p.split_stride = (size_t)Ny * Nx; p.out = S > 1 ? ws : out; p.inv_p = inv_p; p.inv_q = inv_q;
  constexpr size_t lds = 2 * 65536 + 8 * 4096;
  static bool attr_set = false;
  if (!attr_set) {
    OSUD_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad8_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  hipLaunchKernelGGL(wgrad8_kernel, dim3(tiles, S), dim3(512), lds, st, p);
  OSUD_HIP(hipGetLastError());
  if (S > 1) OSUD_TRY(launch_splitk_reduce(ws, S, (size_t)Ny * Nx, out, (size_t)Ny * Nx, st));
  return OSUD_OK;
}

int launch_colsum_bf16(const void* a, int ld, int M, int N, float* out, hipStream_t st, float* part, size_t part_elems) {
  OSUD_CHECK_ARG(N % 8 == 0 && ld % 8 == 0, "colsum: N=%d and ld=%d must be multiples of 8", N, ld);
  const int rb = (M + CS_ROWS - 1) / CS_ROWS;
  OSUD_CHECK_ARG(part != nullptr && (size_t)rb * N <= part_elems, "colsum: %d x %d floats of scratch needed for the row blocks' shares", rb, N);
  hipLaunchKernelGGL(colsum_bf16_kernel<false>, dim3((N + 255) / 256, rb), dim3(256), 0, st,
                     (const bf16_t*)a, ld, M, N, part, (fp8_t*)nullptr, (float*)nullptr);
  OSUD_HIP(hipGetLastError());
  return launch_colsum_f32(part, rb, N, out, st);
}

// column sums + e4m3 twin (q8 may be null: record the amax only) + amax of a DENSE bf16 [M][N] matrix (ld == N) in one pass
int launch_colsum_quant_bf16(const void* a, int M, int N, float* out, void* q8, float* slot, hipStream_t st, float* part, size_t part_elems) {
  OSUD_CHECK_ARG(N % 8 == 0 && slot != nullptr, "colsum_quant: N=%d must be a multiple of 8 and a scale slot is needed", N);
  const int rb = (M + CS_ROWS - 1) / CS_ROWS;
  OSUD_CHECK_ARG(part != nullptr && (size_t)rb * N <= part_elems, "colsum_quant: %d x %d floats of scratch needed for the row blocks' shares", rb, N);
  hipLaunchKernelGGL(colsum_bf16_kernel<true>, dim3((N + 255) / 256, rb), dim3(256), 0, st,
                     (const bf16_t*)a, N, M, N, part, (fp8_t*)q8, slot);
  OSUD_HIP(hipGetLastError());
  return launch_colsum_f32(part, rb, N, out, st);
}

}  // namespace osud
