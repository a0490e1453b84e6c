// Backward of the attention core (autograd of nn.MultiheadAttention's softmax(QK^T/sqrt(hd))V,
// models.py:164-170, as used in training: no mask, T == Tp).
//
// Inputs (training layouts): qkv [M][3D] (Q | K | V row-major, head h = columns h*hd..),
// dO = d(attention output) [M][D], O [M][D], lse [N][H][T] saved by the forward kernel
// (log2 domain in the bf16 tier, natural log in the f32 tier).  Output dqkv [M][3D].
//   P = exp(S - lse), delta = rowsum(dO * O), dS = P * (dO V^T - delta) * scale
//   dQ = dS K, dK = dS^T Q, dV = P^T dO
//
// bf16 tier: one workgroup per (n, h), T <= 128.  Two register-resident passes so that no two
// waves ever share an output: pass A — each wave owns 32 queries and walks the keys (dQ);
// pass B — each wave owns 32 keys and walks the queries (dK, dV).  S and dO.V^T are recomputed
// in the orientation each pass needs (lane = the owned index), which keeps P / dS in registers
// as MFMA B operands exactly like the forward kernel; the A operands that need the contraction
// index contiguous (K^T, Q^T, dO^T) are read straight from the row-major LDS tiles with
// ds_read_b64_tr_b16, so no transposed copies exist (LDS 66 KB for T = 128 -> two workgroups per CU).
// f32 tier: plain VALU kernels (one thread per query / per key).
#include <stdlib.h>

#include "attn_frag.h"
#include "gemm.h"
#include "kernels.h"

namespace osud {

namespace {

// The two register-resident passes on row-major LDS tiles of one (sample, head) (see the file comment), shared by the
// one-workgroup-per-head kernel and the persistent streaming kernel below.
// pass A: dQ for queries own..own+31, walking the keys
template <int HDP>
__device__ __forceinline__ void attn_bwd_pass_a(const char* Qs, const char* Ks, const char* Vs, const char* Os,
                                                const float* lse_s, const float* del_s, int T, int own, int lane, float c1,
                                                float scale, f32x16 (&dq)[HDP / 32]) {
  constexpr int KS = HDP / 16, DT = HDP / 32;
  const int frow = lane & 31, fhalf = lane >> 5;
  u32x4 qf[KS], of[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    qf[ks] = rowfrag<HDP>(Qs, own + frow, 2 * ks + fhalf);
    of[ks] = rowfrag<HDP>(Os, own + frow, 2 * ks + fhalf);
  }
  const float my_lse = lse_s[own + frow], my_del = del_s[own + frow];
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dq[i][r] = 0.f;
  for (int kt = 0; kt < T / 32; ++kt) {
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      s = mfma_bf16(rowfrag<HDP>(Ks, kt * 32 + frow, 2 * ks + fhalf), qf[ks], s);    // D[key][query]
      dp = mfma_bf16(rowfrag<HDP>(Vs, kt * 32 + frow, 2 * ks + fhalf), of[ks], dp);  // dO . V^T
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __builtin_amdgcn_exp2f(s[r] * c1 - my_lse);
      s[r] = p * (dp[r] - my_del) * scale;  // dS
    }
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      const u32x4 dsf = pack8(s, 8 * ss);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
        dq[dt] = mfma_bf16(trfrag<HDP>(Ks, kt * 32 + 16 * ss, dt * 32, lane), dsf, dq[dt]);
    }
  }
}
// pass B: dK, dV for keys own..own+31, walking the queries
template <int HDP>
__device__ __forceinline__ void attn_bwd_pass_b(const char* Qs, const char* Ks, const char* Vs, const char* Os,
                                                const float* lse_s, const float* del_s, int T, int own, int lane, float c1,
                                                float scale, f32x16 (&dk)[HDP / 32], f32x16 (&dv)[HDP / 32]) {
  constexpr int KS = HDP / 16, DT = HDP / 32;
  const int frow = lane & 31, fhalf = lane >> 5;
  u32x4 kf[KS], vf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    kf[ks] = rowfrag<HDP>(Ks, own + frow, 2 * ks + fhalf);
    vf[ks] = rowfrag<HDP>(Vs, own + frow, 2 * ks + fhalf);
  }
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) dk[i][r] = dv[i][r] = 0.f;
  for (int qt = 0; qt < T / 32; ++qt) {
    f32x16 s, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      s = mfma_bf16(rowfrag<HDP>(Qs, qt * 32 + frow, 2 * ks + fhalf), kf[ks], s);    // D[query][key]
      dp = mfma_bf16(rowfrag<HDP>(Os, qt * 32 + frow, 2 * ks + fhalf), vf[ks], dp);
    }
    f32x16 p;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + qt * 32 + 8 * g + 4 * fhalf);
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(del_s + qt * 32 + 8 * g + 4 * fhalf);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float pv = __builtin_amdgcn_exp2f(s[4 * g + i] * c1 - l4[i]);
        p[4 * g + i] = pv;
        s[4 * g + i] = pv * (dp[4 * g + i] - d4[i]) * scale;
      }
    }
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      const u32x4 pf = pack8(p, 8 * ss), dsf = pack8(s, 8 * ss);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        dv[dt] = mfma_bf16(trfrag<HDP>(Os, qt * 32 + 16 * ss, dt * 32, lane), pf, dv[dt]);
        dk[dt] = mfma_bf16(trfrag<HDP>(Qs, qt * 32 + 16 * ss, dt * 32, lane), dsf, dk[dt]);
      }
    }
  }
}
// accumulator rows (lane = the owned row, 4 consecutive columns per register quad) -> bf16 row `orow`
template <int HD, int HDP>
__device__ __forceinline__ void attn_bwd_store(bf16_t* orow, const f32x16 (&acc)[HDP / 32], int fhalf) {
#pragma unroll
  for (int dt = 0; dt < HDP / 32; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int d = dt * 32 + 8 * g + 4 * fhalf;
      if (d < HD) store4(orow + d, acc[dt][4 * g], acc[dt][4 * g + 1], acc[dt][4 * g + 2], acc[dt][4 * g + 3]);
    }
}

template <int HD, int HDP, int NT>
__global__ __launch_bounds__(NT) void attn_bwd_bf16_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dO,
                                                            const bf16_t* __restrict__ O, const float* __restrict__ lse,
                                                            bf16_t* __restrict__ dqkv, int T, int D, float c1,
                                                            float scale) {
  using TL = AttnTile<HDP>;
  constexpr int DT = HDP / 32, CPR = TL::CPR, NWV = NT / 64;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Qs = smem;
  char* Ks = Qs + T * TL::RS;
  char* Vs = Ks + T * TL::RS;
  char* Os = Vs + T * TL::RS;  // dO rows
  float* lse_s = reinterpret_cast<float*>(Os + T * TL::RS);
  float* del_s = lse_s + T;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 31, fhalf = lane >> 5;
  const int h = blockIdx.x, n = blockIdx.y, H = gridDim.x;
  const size_t ld3 = 3 * (size_t)D;
  const size_t m0 = (size_t)n * T;

  // ---- load row-major tiles (coalesced, 16 bytes per lane; pad columns of a 72-wide head are zero) + delta = rowsum(dO * O):
  //      LPR lanes per row (8 for a 64-wide head, 16 for the padded 96), so that a row's partial sums meet in one lane group
  constexpr int LPR = CPR <= 8 ? 8 : 16;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  for (int idx = tid; idx < T * LPR; idx += NT) {
    const int r = idx / LPR, cp = idx % LPR;
    float part = 0.f;
    if (cp < CPR) {
      const bool real = cp * 8 < HD;
      const bf16_t* src = qkv + (m0 + r) * ld3 + h * HD + cp * 8;
      *reinterpret_cast<u32x4*>(Qs + TL::off(r, cp)) = real ? *reinterpret_cast<const u32x4*>(src) : zero4;
      *reinterpret_cast<u32x4*>(Ks + TL::off(r, cp)) = real ? *reinterpret_cast<const u32x4*>(src + D) : zero4;
      *reinterpret_cast<u32x4*>(Vs + TL::off(r, cp)) = real ? *reinterpret_cast<const u32x4*>(src + 2 * D) : zero4;
      const u32x4 dov = real ? *reinterpret_cast<const u32x4*>(dO + (m0 + r) * D + h * HD + cp * 8) : zero4;
      const u32x4 ov = real ? *reinterpret_cast<const u32x4*>(O + (m0 + r) * D + h * HD + cp * 8) : zero4;
      *reinterpret_cast<u32x4*>(Os + TL::off(r, cp)) = dov;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        part += bf2f((bf16_t)(dov[e] & 0xffff)) * bf2f((bf16_t)(ov[e] & 0xffff));
        part += bf2f((bf16_t)(dov[e] >> 16)) * bf2f((bf16_t)(ov[e] >> 16));
      }
    }
    part += __shfl_xor(part, 1, 64);
    part += __shfl_xor(part, 2, 64);
    part += __shfl_xor(part, 4, 64);
    if (LPR == 16) part += __shfl_xor(part, 8, 64);
    if (cp == 0) del_s[r] = part;
  }
  for (int r = tid; r < T; r += NT) lse_s[r] = lse[((size_t)n * H + h) * T + r];
  __syncthreads();
  const int own = wave * 32;  // first query (pass A) / key (pass B) this wave owns
  if (own < T) {
    bf16_t* orow = dqkv + (m0 + own + frow) * ld3 + h * HD;
    {
      f32x16 dq[DT];
      attn_bwd_pass_a<HDP>(Qs, Ks, Vs, Os, lse_s, del_s, T, own, lane, c1, scale, dq);
      attn_bwd_store<HD, HDP>(orow, dq, fhalf);
    }
    {
      f32x16 dk[DT], dv[DT];
      attn_bwd_pass_b<HDP>(Qs, Ks, Vs, Os, lse_s, del_s, T, own, lane, c1, scale, dk, dv);
      attn_bwd_store<HD, HDP>(orow + D, dk, fhalf);
      attn_bwd_store<HD, HDP>(orow + 2 * D, dv, fhalf);
    }
  }
}

// ------------------------------------------------------------------------- T == 128, head_dim 64: persistent, streamed
// The kernel above keeps the memory system idle while a workgroup computes and the matrix pipe idle while it loads: with two
// workgroups per CU (66 KiB of LDS, 228 registers each) a head took 11.4 us of CU time against 6.2 us of HBM time (131 KB per head
// at the 5.4 TB/s a device copy reaches).  Here ONE workgroup per CU walks heads b, b + G, ...: the four tiles of the next head
// (Q | K | V | dO, 64 KiB) travel global -> LDS by LDS-DMA (no registers; wave w fetches tile w, the XOR swizzle of AttnTile is
// applied on the source side) into the other half of a 128 KiB double buffer while the two passes run on this head's tiles.
// delta = rowsum(dO . O) needs O, which is not a tile: its rows are fetched into registers one head ahead.
#ifndef OSUD_ATTN_EXP
#define OSUD_ATTN_EXP 0
#endif
template <int T, bool DBIAS>
__global__ __launch_bounds__(512) void attn_bwd_stream_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dO,
                                                              const bf16_t* __restrict__ O, const float* __restrict__ lse,
                                                              bf16_t* __restrict__ dqkv, int D, int H, int items, float c1,
                                                              float scale, unsigned* __restrict__ queue, float* __restrict__ bias_part) {
  constexpr int HD = 64, HDP = 64, DT = 2;
  using TL = AttnTile<HDP>;
  constexpr int TILE = T * TL::RS;  // bytes
  static_assert(T == 128, "eight waves: four walk the keys (dQ), four the queries (dK, dV), 32 rows each");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2][4][TILE] | lse_s[2][T] | del_s[2][T] | store patches [8][2 KiB]
  float* lse_s = reinterpret_cast<float*>(smem + 8 * TILE);
  float* del_s = lse_s + 2 * T;
  char* patch = reinterpret_cast<char*>(del_s + 2 * T) + (threadIdx.x >> 6) * 2048;
  // Shared-GPU mode (queue != nullptr: collectives hold compute units, so some of the G workgroups start late): a workgroup's first
  // two heads are b and b + G, every later one is 2 G + a ticket drawn one head ahead (requested behind the barrier of head i,
  // published through LDS for the barrier of head i + 1, where the DMA of head i + 2 is issued) -- a late workgroup finds the queue
  // drained instead of doubling the launch.  Which workgroup computes a head does not change its result.
  volatile lds_u32* tword = reinterpret_cast<volatile lds_u32*>((lds_void*)(smem + 8 * TILE + 4 * T * 4 + 8 * 2048));
  const uint32_t lds0 = (uint32_t)(size_t)(const __attribute__((address_space(3))) void*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 31, fhalf = lane >> 5;
  // DBIAS: the in_proj bias gradient (column sums of dQ | dK | dV over all tokens) rides along.  A wave's 32-row column sums come
  // out of store_rows_patch<true> (an MFMA against ones on the rows already sitting in the LDS patch); they meet their three
  // sibling waves in cs_s at the next head's barrier, where one wave per part adds the four in a fixed order and writes the head's
  // 64 sums to bias_part[n][part * D + h * 64 ..] -- every element exactly once; a fixed-order column pass over the N rows follows
  // the kernel.  (Before: a pass over the whole dqkv tensor, 24 us per launch, summed with atomics.)
  float* cs_s = reinterpret_cast<float*>(smem + 8 * TILE + 4 * T * 4 + 8 * 2048 + 16);  // [2][3 parts][4 waves][64]
  auto put_colsums = [&](const f32x16 (&cacc)[2], int slot, int part) {
    int lo = lane;
    asm volatile("" : "+v"(lo));  // (opaque, as above)
    if ((lo & 31) == 0) {
      float* dst = cs_s + ((slot * 3 + part) * 4 + (wave & 3)) * 64 + 4 * (lo >> 5);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<f32x4*>(dst + dt * 32 + 8 * g) = f32x4{cacc[dt][4 * g], cacc[dt][4 * g + 1], cacc[dt][4 * g + 2], cacc[dt][4 * g + 3]};
    }
  };
  auto flush_colsums = [&](int slot, int head) {  // waves 0..2: one part each, lane = column
    if (wave < 3) {
      const float* src = cs_s + (slot * 3 + wave) * 4 * 64;
      const int n = head / H, h = head - n * H;
      bias_part[(size_t)n * 3 * D + (size_t)wave * D + h * HD + lane] = src[lane] + src[64 + lane] + src[128 + lane] + src[192 + lane];
    }
  };
  const size_t ld3 = 3 * (size_t)D;
  // LDS-DMA: one instruction = 64 lanes x 16 bytes = 8 tile rows, written linearly; lane (lr = lane / 8, pc = lane % 8) therefore
  // fetches the chunk that AttnTile::off places at physical position pc of row 8 p + lr: pc ^ ((row >> 1) & 7).
  // Wave w fetches half w % 2 of tile w / 2 (Q, K, V, dO).
  const int tile = wave >> 1;
  const uint32_t ldb = (uint32_t)((tile == 3 ? (size_t)D : ld3) * 2);  // source row stride of this wave's tile, bytes
  const int lr = lane >> 3, pc = lane & 7;
  const uint32_t voff_even = (uint32_t)lr * ldb + (uint32_t)((pc ^ (lr >> 1)) << 4);
  const uint32_t voff_odd = (uint32_t)lr * ldb + (uint32_t)((pc ^ (4 + (lr >> 1))) << 4);
  auto issue = [&](int item, int buf) {
    const int n = item / H, h = item - n * H;
    const char* base = tile == 3 ? reinterpret_cast<const char*>(dO + (size_t)n * T * D + h * HD)
                                 : reinterpret_cast<const char*>(qkv + (size_t)n * T * ld3 + (size_t)tile * D + h * HD);
    const int p0 = (wave & 1) * (T / 16);  // first 8-row piece of this wave's half
    base += (size_t)p0 * 8 * ldb;
    const uint32_t dst0 = lds0 + (uint32_t)((buf * 4 + tile) * TILE + p0 * 1024);
#pragma unroll
    for (int pp = 0; pp < T / 16; ++pp) {  // (T / 16 is even: piece parity = pp parity)
      const char* sb = base + (size_t)pp * 8 * ldb;
      const uint32_t dst = dst0 + pp * 1024;
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"((pp & 1) ? voff_odd : voff_even), "s"(sb), "s"(dst) : "memory");
    }
  };
  // delta = rowsum(dO . O) and lse of the NEXT head are made while this head's stores drain: thread (row r = tid / 4, quarter =
  // tid % 4) fetches 16 columns of O's and dO's row r into registers right after the next head's DMA is issued, and reduces them
  // after the two passes -- by then everything older (that DMA included) has landed, and the compiler's wait for these
  // registers, which counts only what it can see, has nothing left to wait for.  The statistics are double-buffered like the tiles.
  const int dr = tid >> 2, dqt = tid & 3;
  u32x4 oreg[2], doreg[2];
  float lreg = 0.f;
  auto fetch_stats = [&](int item) {
    const int n = item / H, h = item - n * H;
    const size_t off = ((size_t)n * T + dr) * D + h * HD + dqt * 16;
    oreg[0] = *reinterpret_cast<const u32x4*>(O + off);
    oreg[1] = *reinterpret_cast<const u32x4*>(O + off + 8);
    doreg[0] = *reinterpret_cast<const u32x4*>(dO + off);
    doreg[1] = *reinterpret_cast<const u32x4*>(dO + off + 8);
    if (tid < T) lreg = lse[((size_t)n * H + h) * T + tid];
  };
  auto put_stats = [&](int sb) {
    float part = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        part += bf2f((bf16_t)(doreg[j][e] & 0xffff)) * bf2f((bf16_t)(oreg[j][e] & 0xffff));
        part += bf2f((bf16_t)(doreg[j][e] >> 16)) * bf2f((bf16_t)(oreg[j][e] >> 16));
      }
    part += __shfl_xor(part, 1, 64);
    part += __shfl_xor(part, 2, 64);
    if (dqt == 0) del_s[sb * T + dr] = part;
    if (tid < T) lse_s[sb * T + tid] = lreg;
  };
  const int G = gridDim.x;
  const bool ticket_lane = queue != nullptr && tid == 0;
  int it = blockIdx.x, nx = blockIdx.x + G, buf = 0, iter = 0;
  if (it < items) {
    issue(it, 0);
    fetch_stats(it);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the first head's tiles
    put_stats(0);
  }
  int prev = 0;  // the head of the previous iteration (DBIAS)
  for (; it < items; prev = it, it = nx, buf ^= 1, ++iter) {
    const char* Qs = smem + (size_t)buf * 4 * TILE;
    const char* Ks = Qs + TILE;
    const char* Vs = Ks + TILE;
    const char* Os = Vs + TILE;
    const float* lse_b = lse_s + buf * T;
    const float* del_b = del_s + buf * T;
    // every wave has waited for its own pieces of this head and written its statistics; the other buffers are free from here on
    __syncthreads();
    if (DBIAS && iter > 0) flush_colsums((iter - 1) & 1, prev);
    if (iter > 0) nx = queue != nullptr ? 2 * G + __builtin_amdgcn_readfirstlane((int)tword[iter & 1]) : it + G;  // (iteration 0: b + G)
    uint32_t tk;  // (no initialiser: writing the register at the loop top would first wait for last head's ticket AND stores)
    if (ticket_lane) tk = __hip_atomic_fetch_add(queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (nx < items && !(OSUD_ATTN_EXP & 8)) {
      issue(nx, buf ^ 1);
      fetch_stats(nx);
    }
    const int n = it / H, h = it - n * H;
    const int own = (wave & 3) * 32;
    bf16_t* orow = dqkv + ((size_t)n * T + own + frow) * ld3 + h * HD;
    bf16_t* orows = dqkv + ((size_t)n * T + own) * ld3 + h * HD;
#ifndef OSUD_ATTN_EXP
#define OSUD_ATTN_EXP 0
#endif
    if (wave < 4) {  // the two passes are independent once the tiles are in LDS: they run side by side, two waves per SIMD
      f32x16 dq[DT];
      if (OSUD_ATTN_EXP & 2) {
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) dq[i][r] = lse_b[own + frow];
      } else
      attn_bwd_pass_a<HDP>(Qs, Ks, Vs, Os, lse_b, del_b, T, own, lane, c1, scale, dq);
      if (nx < items) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the next head (issued a whole pass ago)
        put_stats(buf ^ 1);
      }
      if (ticket_lane) tword[(iter + 1) & 1] = tk;  // (the ticket has had the whole pass to return)
      if (DBIAS) {
        f32x16 cacc[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) cacc[i][r] = 0.f;
        store_rows_patch<true>(patch, orows, ld3, dq, lane, cacc);
        put_colsums(cacc, iter & 1, 0);
      } else if (!(OSUD_ATTN_EXP & 1)) store_rows_patch(patch, orows, ld3, dq, lane);
      else if (dq[0][0] == 12345.f) orow[0] = 1;
    } else {
      f32x16 dk[DT], dv[DT];
      if (OSUD_ATTN_EXP & 4) {
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) dk[i][r] = dv[i][r] = lse_b[own + frow];
      } else
      attn_bwd_pass_b<HDP>(Qs, Ks, Vs, Os, lse_b, del_b, T, own, lane, c1, scale, dk, dv);
      if (nx < items) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        put_stats(buf ^ 1);
      }
      if (!(OSUD_ATTN_EXP & 1)) {
      if (DBIAS) {
        f32x16 cacc[2];
#pragma unroll
        for (int part = 1; part < 3; ++part) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) cacc[i][r] = 0.f;
          store_rows_patch<true>(patch, orows + part * D, ld3, part == 1 ? dk : dv, lane, cacc);
          put_colsums(cacc, iter & 1, part);
        }
      } else {
      store_rows_patch(patch, orows + D, ld3, dk, lane);
      store_rows_patch(patch, orows + 2 * D, ld3, dv, lane);
      }
      } else if (dk[0][0] + dv[0][0] == 12345.f) orow[D] = 1;
    }
  }
  if (DBIAS && iter > 0) {  // the last head's column sums
    __syncthreads();
    flush_colsums((iter - 1) & 1, prev);
  }
  if (ticket_lane) {  // the last workgroup out re-arms the counters
    const unsigned done = __hip_atomic_fetch_add(queue + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done == (unsigned)G - 1) {
      __hip_atomic_store(queue, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(queue + 8, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ------------------------------------------------------------------------- any T (bf16 tier)
// delta[n][h][t] = sum_d dO[m][h*HD + d] * O[m][h*HD + d]  (one thread per (row, head), 16-byte loads)
template <int HD>
__global__ __launch_bounds__(256) void attn_delta_kernel(const bf16_t* __restrict__ dO, const bf16_t* __restrict__ O,
                                                         float* __restrict__ delta, int N, int T, int H) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N * T * H) return;
  const int h = i % H, m = i / H, n = m / T, t = m % T;
  const bf16_t* a = dO + (size_t)m * H * HD + h * HD;
  const bf16_t* b = O + (size_t)m * H * HD + h * HD;
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < HD / 8; ++c) {
    float x[8], y[8];
    load8(a + 8 * c, x);
    load8(b + 8 * c, y);
#pragma unroll
    for (int e = 0; e < 8; ++e) s = fmaf(x[e], y[e], s);
  }
  delta[((size_t)n * H + h) * T + t] = s;
}

// The same two passes as attn_bwd_bf16_kernel, but a workgroup owns only 128 queries (pass A) / 128 keys (pass B) and
// STREAMS the other side through two LDS tiles of 128 rows, so that T is unbounded (DiT-XL at T = 256, long sequences):
// LDS = 2 x 128 x row stride + 1 KiB.  The owned rows' fragments come straight from global memory; delta is precomputed.
template <int HD, int HDP>
__global__ __launch_bounds__(256, 2) void attn_bwd_tiled_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dO,
                                                             const float* __restrict__ lse, const float* __restrict__ delta,
                                                             bf16_t* __restrict__ dqkv, int T, int D, float c1, float scale) {
  using TL = AttnTile<HDP>;
  constexpr int KS = HDP / 16, DT = HDP / 32, CPR = TL::CPR;
  __shared__ __attribute__((aligned(16))) char Xs[128 * TL::RS];  // K (pass A) / Q (pass B) rows of the streamed block
  __shared__ __attribute__((aligned(16))) char Ys[128 * TL::RS];  // V (pass A) / dO (pass B)
  __shared__ float lse_s[128], del_s[128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 31, fhalf = lane >> 5;
  const int h = blockIdx.y, n = blockIdx.z, H = gridDim.y;
  const size_t ld3 = 3 * (size_t)D;
  const size_t m0 = (size_t)n * T;
  const int own = blockIdx.x * 128 + wave * 32;  // first query (pass A) / key (pass B) this wave owns
  const bool live = own < T;                     // T % 32 == 0: a wave is either fully inside or fully outside
  const int orow_i = live ? own + frow : 0;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  const int nblk = (T + 127) / 128;
  auto gfrag = [&](const bf16_t* base, size_t ld, int col0, int ks) -> u32x4 {  // chunk 2ks+fhalf of the owned row
    const int c = 2 * ks + fhalf;
    return (live && c * 8 < HD) ? *reinterpret_cast<const u32x4*>(base + (m0 + orow_i) * ld + col0 + h * HD + c * 8) : zero4;
  };
  auto load_block = [&](int blk, int colX, const bf16_t* ysrc, size_t yld, int colY, bool stats) {
    const int rows = T - blk * 128 < 128 ? T - blk * 128 : 128;
    __syncthreads();  // the previous block is fully consumed
    for (int idx = tid; idx < 128 * CPR; idx += 256) {
      const int r = idx / CPR, cp = idx % CPR;
      const bool real = r < rows && cp * 8 < HD;
      const size_t m = m0 + blk * 128 + r;
      *reinterpret_cast<u32x4*>(Xs + TL::off(r, cp)) = real ? *reinterpret_cast<const u32x4*>(qkv + m * ld3 + colX + h * HD + cp * 8) : zero4;
      *reinterpret_cast<u32x4*>(Ys + TL::off(r, cp)) = real ? *reinterpret_cast<const u32x4*>(ysrc + m * yld + colY + h * HD + cp * 8) : zero4;
    }
    if (stats && tid < 128) {
      const bool real = tid < rows;
      lse_s[tid] = real ? lse[((size_t)n * H + h) * T + blk * 128 + tid] : 0.f;
      del_s[tid] = real ? delta[((size_t)n * H + h) * T + blk * 128 + tid] : 0.f;
    }
    __syncthreads();
    return rows;
  };

  // =============================== pass A: dQ for the owned queries; K, V streamed ======================
  {
    u32x4 qf[KS], of[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qf[ks] = gfrag(qkv, ld3, 0, ks);
      of[ks] = gfrag(dO, (size_t)D, 0, ks);
    }
    const float my_lse = live ? lse[((size_t)n * H + h) * T + orow_i] : 0.f;
    const float my_del = live ? delta[((size_t)n * H + h) * T + orow_i] : 0.f;
    f32x16 dq[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) dq[i][r] = 0.f;
    for (int kb = 0; kb < nblk; ++kb) {
      const int rows = load_block(kb, D, qkv, ld3, 2 * D, false);
      if (!live) continue;
      for (int kt = 0; kt < rows / 32; ++kt) {
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          s = mfma_bf16(rowfrag<HDP>(Xs, kt * 32 + frow, 2 * ks + fhalf), qf[ks], s);    // D[key][query]
          dp = mfma_bf16(rowfrag<HDP>(Ys, kt * 32 + frow, 2 * ks + fhalf), of[ks], dp);  // dO . V^T
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = __builtin_amdgcn_exp2f(s[r] * c1 - my_lse);
          s[r] = p * (dp[r] - my_del) * scale;  // dS
        }
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
          const u32x4 dsf = pack8(s, 8 * ss);
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) dq[dt] = mfma_bf16(trfrag<HDP>(Xs, kt * 32 + 16 * ss, dt * 32, lane), dsf, dq[dt]);
        }
      }
    }
    if (live) {
      bf16_t* orow = dqkv + (m0 + own + frow) * ld3 + h * HD;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          if (dt * 32 + 8 * g + 4 * fhalf < HD)
            store4(orow + dt * 32 + 8 * g + 4 * fhalf, dq[dt][4 * g], dq[dt][4 * g + 1], dq[dt][4 * g + 2], dq[dt][4 * g + 3]);
    }
  }
  // =============================== pass B: dK, dV for the owned keys; Q, dO streamed =====================
  {
    u32x4 kf[KS], vf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      kf[ks] = gfrag(qkv, ld3, D, ks);
      vf[ks] = gfrag(qkv, ld3, 2 * D, ks);
    }
    f32x16 dk[DT], dv[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) dk[i][r] = dv[i][r] = 0.f;
    for (int qb = 0; qb < nblk; ++qb) {
      const int rows = load_block(qb, 0, dO, (size_t)D, 0, true);
      if (!live) continue;
      for (int qt = 0; qt < rows / 32; ++qt) {
        f32x16 s, dp;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          s = mfma_bf16(rowfrag<HDP>(Xs, qt * 32 + frow, 2 * ks + fhalf), kf[ks], s);    // D[query][key]
          dp = mfma_bf16(rowfrag<HDP>(Ys, qt * 32 + frow, 2 * ks + fhalf), vf[ks], dp);
        }
        f32x16 p;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + qt * 32 + 8 * g + 4 * fhalf);
          const f32x4 d4 = *reinterpret_cast<const f32x4*>(del_s + qt * 32 + 8 * g + 4 * fhalf);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float pv = __builtin_amdgcn_exp2f(s[4 * g + i] * c1 - l4[i]);
            p[4 * g + i] = pv;
            s[4 * g + i] = pv * (dp[4 * g + i] - d4[i]) * scale;
          }
        }
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
          const u32x4 pf = pack8(p, 8 * ss), dsf = pack8(s, 8 * ss);
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            dv[dt] = mfma_bf16(trfrag<HDP>(Ys, qt * 32 + 16 * ss, dt * 32, lane), pf, dv[dt]);
            dk[dt] = mfma_bf16(trfrag<HDP>(Xs, qt * 32 + 16 * ss, dt * 32, lane), dsf, dk[dt]);
          }
        }
      }
    }
    if (live) {
      bf16_t* orow = dqkv + (m0 + own + frow) * ld3 + h * HD;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = dt * 32 + 8 * g + 4 * fhalf;
          if (d < HD) {
            store4(orow + D + d, dk[dt][4 * g], dk[dt][4 * g + 1], dk[dt][4 * g + 2], dk[dt][4 * g + 3]);
            store4(orow + 2 * D + d, dv[dt][4 * g], dv[dt][4 * g + 1], dv[dt][4 * g + 2], dv[dt][4 * g + 3]);
          }
        }
    }
  }
}

// ------------------------------------------------------------------------- T == 256, head_dim 72 (DiT-XL): persistent, streamed
// Four tiles of 256 x 96 (padded) columns do not fit the LDS, so the tiled kernel above gives a workgroup 128 queries / 128 keys
// and lets it stream the other side: every tile is read twice, through registers, by workgroups that alternate between loading
// and computing -- 426 us for 600 MB of algorithmic traffic (a device copy of those bytes: 110 us).  Here one eight-wave workgroup
// per CU walks heads; a head is four block steps -- K|V rows 0..127, K|V 128..255 (all eight waves in the dQ pass, 32 queries
// each), then Q|dO 0..127, Q|dO 128..255 (the dK/dV pass, 32 keys each) -- whose 128-row tile pairs (52 KiB) arrive by LDS-DMA in
// a two-stage ring, one step ahead, across head boundaries.  The rows a wave owns (B operands of the score products) and the next
// head's lse / delta come straight from global memory one pass ahead.  EVERY vector-memory instruction in the loop except the
// stores is inline asm, so that the compiler inserts no wait of its own (it cannot see the DMA pieces and would wait for them,
// or for the previous pass's stores); the waits below are counted by hand -- vector memory retires in issue order:
//   per wave and head:  step 0: P DMA pieces, 10 fragment loads (K, V rows) | step 1: P pieces ... 6 stores (dQ)
//                       step 2: P pieces | step 3: P pieces ... 10 fragment loads (next head's Q, dO rows), 2 statistics loads,
//                       12 stores (dK, dV)
template <int T>
__global__ __launch_bounds__(512, 2) void attn_bwd_stream72_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ dO,
                                                                   const float* __restrict__ lse, const float* __restrict__ delta,
                                                                   bf16_t* __restrict__ dqkv, int D, int H, int items, float c1,
                                                                   float scale, unsigned* __restrict__ queue) {
  constexpr int HD = 72, HDP = 96, KS = 5, DT = 3, BLK = 128;  // KS: 16-column k-steps that hold real columns (72 -> 5; the sixth is all pad)
  using TL = AttnTile<HDP>;
  constexpr int TILE = BLK * TL::RS, STAGE = 2 * TILE;  // 26 624 / 53 248 bytes
  static_assert(T == 256 && TL::RS == 208, "eight waves x 32 rows, two 128-row blocks per side");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][X | Y][TILE] | lse_s[2][T] | del_s[2][T] | patches [8][16 x 208]
  float* lse_s = reinterpret_cast<float*>(smem + 2 * STAGE);
  float* del_s = lse_s + 2 * T;
  char* patch = reinterpret_cast<char*>(del_s + 2 * T) + (threadIdx.x >> 6) * (16 * 208);
  // shared-GPU mode (queue != nullptr): heads b, b + G, then 2 G + ticket, requested one head ahead in step 2 (it has returned by the
  // vmcnt(0) of step 3 and does not enter any of the counted waits) and published through LDS for the next head's first barrier
  volatile lds_u32* tword = reinterpret_cast<volatile lds_u32*>((lds_void*)(smem + 2 * STAGE + 4 * T * 4 + 8 * 16 * 208));
  const uint32_t lds0 = (uint32_t)(size_t)(const __attribute__((address_space(3))) void*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 31, fhalf = lane >> 5;
  const size_t ld3 = 3 * (size_t)D;
  const int own = wave * 32;
  // ---- LDS-DMA of one stage: 52 pieces of 1 KiB (64 lanes x 16 bytes, written linearly = 13 chunks per 208-byte row);
  //      waves 0..3 issue 7 pieces, waves 4..7 six.  Lane -> (row, chunk); chunks 9..12 (pad columns, stride pad) read zeros.
  const int p_first = wave < 4 ? 7 * wave : 28 + 6 * (wave - 4), p_count = wave < 4 ? 7 : 6;
  const char* zsrc = reinterpret_cast<const char*>(&g_attn_zero16);
  auto issue = [&](int head, int step, int stage) {  // step 0,1: K|V block step; 2,3: Q|dO block step - 2
    const int n = head / H, h = head - n * H;
    const int blk = step & 1;
    const char* xb = reinterpret_cast<const char*>(qkv + ((size_t)n * T + blk * BLK) * ld3 + (step < 2 ? D : 0) + h * HD);
    const char* yb = step < 2 ? reinterpret_cast<const char*>(qkv + ((size_t)n * T + blk * BLK) * ld3 + 2 * D + h * HD)
                              : reinterpret_cast<const char*>(dO + ((size_t)n * T + blk * BLK) * D + h * HD);
    const size_t xld = ld3 * 2, yld = (step < 2 ? ld3 : (size_t)D) * 2;
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));  // (opaque: the per-piece addresses are recomputed here, not kept in 30 registers across the passes)
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      if (q < p_count) {
        const int p = p_first + q, t = p >= 26 ? 1 : 0, pp = p - 26 * t;  // wave-uniform
        const int gi = pp * 64 + lane_o, row = (gi * 5042) >> 16, c = gi - 13 * row;
        const char* src = c < 9 ? (t ? yb + (size_t)row * yld : xb + (size_t)row * xld) + c * 16 : zsrc;
        const uint32_t dst = lds0 + (uint32_t)(stage * STAGE + t * TILE + pp * 1024);
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(dst) : "memory");
      }
    }
  };
  // ---- the rows this wave owns, as B-operand fragments straight from global memory: chunk 2 ks + fhalf of row own + frow
  //      (five loads per tensor: k-steps 0..4; chunk 9 of k-step 4 is a pad column = zero; k-step 5 would be all pad: skipped)
  auto fetch_frags = [&](int head, const bf16_t* a, size_t lda, int cola, const bf16_t* b, size_t ldb, int colb, u32x4 (&fa)[KS],
                         u32x4 (&fb)[KS]) {
    const int n = head / H, h = head - n * H;
    const bf16_t* pa = a + ((size_t)n * T + own + frow) * lda + cola + h * HD;
    const bf16_t* pb = b + ((size_t)n * T + own + frow) * ldb + colb + h * HD;
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
      const int c = (ks == 4 && fhalf) ? 8 : 2 * ks + fhalf;  // (the pad chunk's lanes read a valid chunk and drop it)
      gload16(fa[ks], pa + c * 8);
      gload16(fb[ks], pb + c * 8);
    }
  };
  auto settle_frags = [&](u32x4 (&fa)[KS], u32x4 (&fb)[KS]) {  // after the counted wait: the registers are defined from here
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
      asm volatile("" : "+v"(fa[ks]), "+v"(fb[ks]));
    }
    if (fhalf) fa[4] = fb[4] = zero4;
  };
  u32x4 qf[KS], of[KS], kf[KS], vf[KS];
  float lreg = 0.f, dreg = 0.f;
  auto fetch_stats = [&](int head) {  // thread t < T: lse / delta of query t (the other threads read entry 0 and drop it)
    const int n = head / H, h = head - n * H;
    const size_t o = ((size_t)n * H + h) * T + (tid < T ? tid : 0);
    gload4(lreg, lse + o);
    gload4(dreg, delta + o);
  };
  auto put_stats = [&](int sb) {
    asm volatile("" : "+v"(lreg), "+v"(dreg));
    if (tid < T) {
      lse_s[sb * T + tid] = lreg;
      del_s[sb * T + tid] = dreg;
    }
  };

  const int G = gridDim.x;
  const bool ticket_lane = queue != nullptr && tid == 0;
  int it = blockIdx.x, nx = blockIdx.x + G, iter = 0, hb = 0;  // hb: statistics buffer of this head
  if (it < items) {
    issue(it, 0, 0);
    fetch_frags(it, qkv, ld3, 0, dO, (size_t)D, 0, qf, of);
    fetch_stats(it);
    OSUD_VM_WAIT(0);
    settle_frags(qf, of);
    put_stats(0);
  }
  uint32_t tk;  // (no initialiser: see attn_bwd_stream_kernel)
  for (; it < items; it = nx, hb ^= 1, ++iter) {
    const int n = it / H, h = it - n * H;
    const float* lse_b = lse_s + hb * T;
    const float* del_b = del_s + hb * T;
    bf16_t* orows = dqkv + ((size_t)n * T + own) * ld3 + h * HD;
    // =============================== pass A: dQ for queries own..own+31; K | V blocks streamed ======================
    {
      f32x16 dq[DT];
#pragma unroll
      for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[i][r] = 0.f;
      float my_lse = 0.f, my_del = 0.f;
#pragma unroll
      for (int step = 0; step < 2; ++step) {
        // this step's tiles have landed (step 0: issued in step 3 of the previous head, with this head's fragments and statistics
        // and the 12 dK / dV stores behind them; step 1: issued in step 0, 10 fragment loads behind them)
        if (step == 0) {
          OSUD_VM_WAIT(12);
          if (it != (int)blockIdx.x) {  // (the first head's were settled in the prologue)
            settle_frags(qf, of);
            put_stats(hb);
          }
        } else {
          OSUD_VM_WAIT(10);
        }
        __syncthreads();
        if (step == 0 && iter > 0) nx = queue != nullptr ? 2 * G + __builtin_amdgcn_readfirstlane((int)tword[iter & 1]) : it + G;
        if (step == 0) {
          my_lse = lse_b[own + frow];
          my_del = del_b[own + frow];
        }
        issue(it, step + 1, (step + 1) & 1);
        if (step == 0) fetch_frags(it, qkv, ld3, D, qkv, ld3, 2 * D, kf, vf);
        const char* Xs = smem + (step & 1) * STAGE;
        const char* Ys = Xs + TILE;
        for (int kt = 0; kt < BLK / 32; ++kt) {
          f32x16 s, dp;
#pragma unroll
          for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            s = mfma_bf16(rowfrag<HDP>(Xs, kt * 32 + frow, 2 * ks + fhalf), qf[ks], s);    // D[key][query]
            dp = mfma_bf16(rowfrag<HDP>(Ys, kt * 32 + frow, 2 * ks + fhalf), of[ks], dp);  // dO . V^T
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(s[r] * c1 - my_lse);
            s[r] = p * (dp[r] - my_del) * scale;  // dS
          }
#pragma unroll
          for (int ss = 0; ss < 2; ++ss) {
            const u32x4 dsf = pack8(s, 8 * ss);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) dq[dt] = mfma_bf16(trfrag<HDP>(Xs, kt * 32 + 16 * ss, dt * 32, lane), dsf, dq[dt]);
          }
        }
      }
      store_rows_patch72(patch, orows, ld3, dq, lane);  // 6 stores
    }
    // =============================== pass B: dK, dV for keys own..own+31; Q | dO blocks streamed =====================
    {
      f32x16 dk[DT], dv[DT];
#pragma unroll
      for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) dk[i][r] = dv[i][r] = 0.f;
#pragma unroll
      for (int step = 2; step < 4; ++step) {
        // step 2: tiles issued in step 1, the 6 dQ stores behind them (this pass's fragments are older: landed as well);
        // step 3: tiles issued in step 2, nothing behind them
        if (step == 2) { OSUD_VM_WAIT(6); } else { OSUD_VM_WAIT(0); }
        __syncthreads();
        if (step == 2) {
          settle_frags(kf, vf);
          if (ticket_lane) tk = __hip_atomic_fetch_add(queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          issue(it, 3, 1);
        } else {
          if (ticket_lane) tword[(iter + 1) & 1] = tk;  // (returned: the wait above was vmcnt(0))
          issue(nx < items ? nx : it, 0, 0);  // (past the last head: a harmless re-read into the free stage)
        }
        const char* Xs = smem + (step & 1) * STAGE;
        const char* Ys = Xs + TILE;
        const int q0 = (step - 2) * BLK;
        for (int qt = 0; qt < BLK / 32; ++qt) {
          f32x16 s, dp;
#pragma unroll
          for (int r = 0; r < 16; ++r) s[r] = dp[r] = 0.f;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            s = mfma_bf16(rowfrag<HDP>(Xs, qt * 32 + frow, 2 * ks + fhalf), kf[ks], s);    // D[query][key]
            dp = mfma_bf16(rowfrag<HDP>(Ys, qt * 32 + frow, 2 * ks + fhalf), vf[ks], dp);
          }
          f32x16 p;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_b + q0 + qt * 32 + 8 * g + 4 * fhalf);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(del_b + q0 + qt * 32 + 8 * g + 4 * fhalf);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float pv = __builtin_amdgcn_exp2f(s[4 * g + i] * c1 - l4[i]);
              p[4 * g + i] = pv;
              s[4 * g + i] = pv * (dp[4 * g + i] - d4[i]) * scale;
            }
          }
#pragma unroll
          for (int ss = 0; ss < 2; ++ss) {
            const u32x4 pf = pack8(p, 8 * ss), dsf = pack8(s, 8 * ss);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
              dv[dt] = mfma_bf16(trfrag<HDP>(Ys, qt * 32 + 16 * ss, dt * 32, lane), pf, dv[dt]);
              dk[dt] = mfma_bf16(trfrag<HDP>(Xs, qt * 32 + 16 * ss, dt * 32, lane), dsf, dk[dt]);
            }
          }
        }
      }
      // the next head's own rows and statistics: issued here, where the score registers are dead, in front of this head's stores;
      // they are waited for together with the next head's first tiles (the last head re-reads its own: same instruction counts)
      fetch_frags(nx < items ? nx : it, qkv, ld3, 0, dO, (size_t)D, 0, qf, of);
      fetch_stats(nx < items ? nx : it);
      store_rows_patch72(patch, orows + D, ld3, dk, lane);      // 6 stores
      store_rows_patch72(patch, orows + 2 * D, ld3, dv, lane);  // 6 stores
    }
  }
  if (ticket_lane) {  // the last workgroup out re-arms the counters
    const unsigned done = __hip_atomic_fetch_add(queue + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done == (unsigned)G - 1) {
      __hip_atomic_store(queue, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(queue + 8, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ------------------------------------------------------------------------- f32 tier (VALU)
template <int HD>
__global__ __launch_bounds__(64) void attn_bwd_dq_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ dO,
                                                             const float* __restrict__ O, const float* __restrict__ lse,
                                                             float* __restrict__ dqkv, int T, int D, float scale) {
  __shared__ float Ks[64][HD];
  __shared__ float Vs[64][HD];
  const int tid = threadIdx.x, n = blockIdx.z, h = blockIdx.y, H = gridDim.y;
  const int q = blockIdx.x * 64 + tid;
  const size_t ld3 = 3 * (size_t)D, m = (size_t)n * T + q;
  float qv[HD], dov[HD], dq[HD];
  float delta = 0.f;
#pragma unroll
  for (int d = 0; d < HD; ++d) {
    qv[d] = qkv[m * ld3 + h * HD + d] * scale;
    dov[d] = dO[m * D + h * HD + d];
    delta += dov[d] * O[m * D + h * HD + d];
    dq[d] = 0.f;
  }
  const float my_lse = lse[((size_t)n * H + h) * T + q];
  for (int kb = 0; kb * 64 < T; ++kb) {
    __syncthreads();
    for (int idx = tid; idx < 64 * HD; idx += 64) {
      const int r = idx / HD, d = idx % HD;
      const size_t mk = (size_t)n * T + kb * 64 + r;
      Ks[r][d] = qkv[mk * ld3 + D + h * HD + d];
      Vs[r][d] = qkv[mk * ld3 + 2 * D + h * HD + d];
    }
    __syncthreads();
    for (int j = 0; j < 64; ++j) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) {
        s = fmaf(qv[d], Ks[j][d], s);
        dp = fmaf(dov[d], Vs[j][d], dp);
      }
      const float ds = expf(s - my_lse) * (dp - delta) * scale;
#pragma unroll
      for (int d = 0; d < HD; ++d) dq[d] = fmaf(ds, Ks[j][d], dq[d]);
    }
  }
#pragma unroll
  for (int d = 0; d < HD; ++d) dqkv[m * ld3 + h * HD + d] = dq[d];
}

template <int HD>
__global__ __launch_bounds__(64) void attn_bwd_dkv_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ dO,
                                                              const float* __restrict__ O, const float* __restrict__ lse,
                                                              float* __restrict__ dqkv, int T, int D, float scale) {
  __shared__ float Qs[64][HD];
  __shared__ float Os[64][HD];
  __shared__ float ls[64], dl[64];
  const int tid = threadIdx.x, n = blockIdx.z, h = blockIdx.y, H = gridDim.y;
  const int key = blockIdx.x * 64 + tid;
  const size_t ld3 = 3 * (size_t)D, m = (size_t)n * T + key;
  float kv[HD], vv[HD], dk[HD], dv[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) {
    kv[d] = qkv[m * ld3 + D + h * HD + d];
    vv[d] = qkv[m * ld3 + 2 * D + h * HD + d];
    dk[d] = dv[d] = 0.f;
  }
  for (int qb = 0; qb * 64 < T; ++qb) {
    __syncthreads();
    for (int idx = tid; idx < 64 * HD; idx += 64) {
      const int r = idx / HD, d = idx % HD;
      const size_t mq = (size_t)n * T + qb * 64 + r;
      Qs[r][d] = qkv[mq * ld3 + h * HD + d] * scale;
      Os[r][d] = dO[mq * D + h * HD + d];
    }
    {
      const size_t mq = (size_t)n * T + qb * 64 + tid;
      float delta = 0.f;
      for (int d = 0; d < HD; ++d) delta += dO[mq * D + h * HD + d] * O[mq * D + h * HD + d];
      dl[tid] = delta;
      ls[tid] = lse[((size_t)n * H + h) * T + qb * 64 + tid];
    }
    __syncthreads();
    for (int j = 0; j < 64; ++j) {
      float s = 0.f, dp = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) {
        s = fmaf(Qs[j][d], kv[d], s);
        dp = fmaf(Os[j][d], vv[d], dp);
      }
      const float p = expf(s - ls[j]);
      const float ds = p * (dp - dl[j]);  // Qs already carries `scale`
#pragma unroll
      for (int d = 0; d < HD; ++d) {
        dv[d] = fmaf(p, Os[j][d], dv[d]);
        dk[d] = fmaf(ds, Qs[j][d], dk[d]);
      }
    }
  }
#pragma unroll
  for (int d = 0; d < HD; ++d) {
    dqkv[m * ld3 + D + h * HD + d] = dk[d];
    dqkv[m * ld3 + 2 * D + h * HD + d] = dv[d];
  }
}

}  // namespace

int launch_attention_bwd(int prec, const void* qkv, const void* dO, const void* O, const float* lse, void* dqkv, int N,
                         int T, int heads, int head_dim, hipStream_t st, float* delta_ws, float* dbias, float* bias_scratch,
                         size_t bias_scratch_elems, int* bias_rows_pending) {
  if (bias_rows_pending != nullptr) *bias_rows_pending = 0;
  OSUD_CHECK_ARG(N > 0 && T > 0 && T % 64 == 0, "attention backward: T=%d must be a multiple of 64", T);
  const int D = heads * head_dim;
  const float scale = 1.0f / sqrtf((float)head_dim);
  if (prec == OSUD_PREC_BF16) {
    // a workgroup keeps the whole sequence of one (sample, head) in LDS: 4 tiles of T rows (one wave per 32 rows)
    const int rs = head_dim == 64 ? AttnTile<64>::RS : AttnTile<96>::RS;
    const size_t lds = (size_t)4 * T * rs + (size_t)2 * T * 4;
    if (head_dim != 64 && head_dim != 72) {
      set_error("attention backward (bf16 tier) is built for head_dim 64 and 72 (got %d)", head_dim);
      return OSUD_ERR_UNSUPPORTED;
    }
    const float c1t = scale * 1.4426950408889634f;
    const int kernel_opt = opt(OPT_ATTN_BWD_KERNEL);  // osud_set_option("attn_bwd_kernel", 1: no streamed kernels | 2: the tiled kernel) -- tests
    const bool force_tiled = kernel_opt == 2;
    if (T > 256 || lds > 160 * 1024 || force_tiled) {  // the sequence of a head does not fit the LDS: streamed variant
      OSUD_CHECK_ARG(delta_ws != nullptr, "attention backward: T=%d needs the delta workspace", T);
      const int rows = N * T * heads;
      const dim3 grid((T + 127) / 128, heads, N);
      if (head_dim == 72 && T == 256 && kernel_opt == 0) {  // DiT-XL: persistent streamed kernel
        hipLaunchKernelGGL((attn_delta_kernel<72>), dim3((rows + 255) / 256), dim3(256), 0, st, (const bf16_t*)dO, (const bf16_t*)O,
                           delta_ws, N, T, heads);
        constexpr size_t lds72 = (size_t)4 * 128 * AttnTile<96>::RS + 4 * 256 * 4 + 8 * 16 * 208 + 16;
        OSUD_BIG_LDS_ONCE(attn_bwd_stream72_kernel<256>);
        const int cus = device_cus();
        const int items = N * heads;
        hipLaunchKernelGGL((attn_bwd_stream72_kernel<256>), dim3(items < cus ? items : cus), dim3(512), lds72, st, (const bf16_t*)qkv,
                           (const bf16_t*)dO, lse, delta_ws, (bf16_t*)dqkv, D, heads, items, c1t, scale,
                           (gemm_dynamic_tiles_on() && items > 2 * cus) ? gemm_ticket_slot() : nullptr);
        OSUD_HIP(hipGetLastError());
        if (dbias != nullptr) OSUD_TRY(launch_colsum_bf16(dqkv, 3 * D, N * T, 3 * D, dbias, st, bias_scratch, bias_scratch_elems));
        return OSUD_OK;
      }
      if (head_dim == 64) {
        hipLaunchKernelGGL((attn_delta_kernel<64>), dim3((rows + 255) / 256), dim3(256), 0, st, (const bf16_t*)dO, (const bf16_t*)O,
                           delta_ws, N, T, heads);
        hipLaunchKernelGGL((attn_bwd_tiled_kernel<64, 64>), grid, dim3(256), 0, st, (const bf16_t*)qkv, (const bf16_t*)dO, lse,
                           delta_ws, (bf16_t*)dqkv, T, D, c1t, scale);
      } else {
        hipLaunchKernelGGL((attn_delta_kernel<72>), dim3((rows + 255) / 256), dim3(256), 0, st, (const bf16_t*)dO, (const bf16_t*)O,
                           delta_ws, N, T, heads);
        hipLaunchKernelGGL((attn_bwd_tiled_kernel<72, 96>), grid, dim3(256), 0, st, (const bf16_t*)qkv, (const bf16_t*)dO, lse,
                           delta_ws, (bf16_t*)dqkv, T, D, c1t, scale);
      }
      OSUD_HIP(hipGetLastError());
      if (dbias != nullptr) OSUD_TRY(launch_colsum_bf16(dqkv, 3 * D, N * T, 3 * D, dbias, st, bias_scratch, bias_scratch_elems));  // streamed variant: separate pass
      return OSUD_OK;
    }
    OSUD_BIG_LDS_ONCE((attn_bwd_bf16_kernel<64, 64, 256>));
    OSUD_BIG_LDS_ONCE((attn_bwd_bf16_kernel<64, 64, 512>));
    OSUD_BIG_LDS_ONCE((attn_bwd_bf16_kernel<72, 96, 256>));
    const float c1 = scale * 1.4426950408889634f;
    const bool stream_on = kernel_opt == 0;  // (else: the one-workgroup-per-head kernel)
    if (head_dim == 64 && T == 128 && stream_on) {
      constexpr size_t slds = (size_t)8 * 128 * AttnTile<64>::RS + 4 * 128 * 4 + 8 * 2048 + 16 + 2 * 3 * 4 * 64 * 4;
      const int cus = device_cus();
      const int items = N * heads;
      unsigned* queue = (gemm_dynamic_tiles_on() && items > 2 * cus) ? gemm_ticket_slot() : nullptr;
      const bool fuse_bias = dbias != nullptr && bias_scratch != nullptr && (size_t)N * 3 * D <= bias_scratch_elems && N >= 64;
      if (fuse_bias) {
        OSUD_BIG_LDS_ONCE((attn_bwd_stream_kernel<128, true>));
        hipLaunchKernelGGL((attn_bwd_stream_kernel<128, true>), dim3(items < cus ? items : cus), dim3(512), slds, st, (const bf16_t*)qkv,
                           (const bf16_t*)dO, (const bf16_t*)O, lse, (bf16_t*)dqkv, D, heads, items, c1, scale, queue, bias_scratch);
        OSUD_HIP(hipGetLastError());
        if (bias_rows_pending != nullptr) {  // (the caller batches the fixed-order sums of a whole backward call into one launch)
          *bias_rows_pending = N;
          return OSUD_OK;
        }
        return launch_colsum_f32(bias_scratch, N, 3 * D, dbias, st);  // fixed-order sum over the samples
      }
      OSUD_BIG_LDS_ONCE((attn_bwd_stream_kernel<128, false>));
      hipLaunchKernelGGL((attn_bwd_stream_kernel<128, false>), dim3(items < cus ? items : cus), dim3(512), slds, st, (const bf16_t*)qkv,
                         (const bf16_t*)dO, (const bf16_t*)O, lse, (bf16_t*)dqkv, D, heads, items, c1, scale, queue, nullptr);
      OSUD_HIP(hipGetLastError());
      if (dbias != nullptr) OSUD_TRY(launch_colsum_bf16(dqkv, 3 * D, N * T, 3 * D, dbias, st, bias_scratch, bias_scratch_elems));
      return OSUD_OK;
    }
    // (the in_proj bias gradient by lane butterflies inside these kernels measured slower than the column-sum pass behind them)
    if (head_dim == 64 && T <= 128)
      hipLaunchKernelGGL((attn_bwd_bf16_kernel<64, 64, 256>), dim3(heads, N), dim3(256), lds, st, (const bf16_t*)qkv,
                         (const bf16_t*)dO, (const bf16_t*)O, lse, (bf16_t*)dqkv, T, D, c1, scale);
    else if (head_dim == 64)
      hipLaunchKernelGGL((attn_bwd_bf16_kernel<64, 64, 512>), dim3(heads, N), dim3(512), lds, st, (const bf16_t*)qkv,
                         (const bf16_t*)dO, (const bf16_t*)O, lse, (bf16_t*)dqkv, T, D, c1, scale);
    else
      hipLaunchKernelGGL((attn_bwd_bf16_kernel<72, 96, 256>), dim3(heads, N), dim3(256), lds, st, (const bf16_t*)qkv,
                         (const bf16_t*)dO, (const bf16_t*)O, lse, (bf16_t*)dqkv, T, D, c1, scale);
    OSUD_HIP(hipGetLastError());
    if (dbias != nullptr) OSUD_TRY(launch_colsum_bf16(dqkv, 3 * D, N * T, 3 * D, dbias, st, bias_scratch, bias_scratch_elems));
    return OSUD_OK;
  } else {
    const dim3 grid(T / 64, heads, N);
#define OSUD_ABWD(HD)                                                                                                   \
  hipLaunchKernelGGL((attn_bwd_dq_f32_kernel<HD>), grid, dim3(64), 0, st, (const float*)qkv, (const float*)dO,       \
                     (const float*)O, lse, (float*)dqkv, T, D, scale);                                                  \
  hipLaunchKernelGGL((attn_bwd_dkv_f32_kernel<HD>), grid, dim3(64), 0, st, (const float*)qkv, (const float*)dO,      \
                     (const float*)O, lse, (float*)dqkv, T, D, scale)
    if (head_dim == 64) { OSUD_ABWD(64); }
    else if (head_dim == 72) { OSUD_ABWD(72); }
    else {
      set_error("attention backward: head_dim %d not built (64, 72)", head_dim);
      return OSUD_ERR_UNSUPPORTED;
    }
#undef OSUD_ABWD
  }
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

}  // namespace osud
