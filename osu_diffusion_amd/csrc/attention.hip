// Self-attention core for the DiT blocks (reference: nn.MultiheadAttention inside
// DiTBlock.forward, models.py:130-135,164-170): softmax(q k^T / sqrt(hd) + mask) v per
// (sample, head), non-causal, optional (T,T) boolean mask shared by all samples/heads.
//
// Input comes straight from the packed in_proj GEMM (no head-major reshuffle, no transposed V):
//   qkv [Mp][3D]  row m = n*Tp + t : Q in columns [0,D), K in [D,2D), V in [2D,3D); head h = 64 columns
// The P.V product needs its contraction index (the key) contiguous per lane; V stays row-major [key][d] in LDS
// and the fragments come out of ds_read_b64_tr_b16 (attn_frag.h), as in the backward kernel.
// Output ao [Mp][D].  Tp (tokens per sample incl. padding) is a multiple of 64; keys >= T
// are masked out.
//
// bf16 tier: one workgroup = 128 queries of one (n,h); 4 waves x 32 queries; K / V blocks of
// 64 keys staged in LDS (XOR-swizzled 16-byte chunks); S^T = K.Q^T on MFMA 32x32x16 so every
// lane owns ONE query (its 32 scores of the tile sit in its own registers + the partner lane
// l^32) -> softmax needs one cross-lane exchange; P goes back into the MFMA as the B operand
// without leaving registers (the key order inside a k-step is permuted identically on the
// V side).  fp32 statistics, online softmax across key blocks.
// f32 (parity) tier: plain one-thread-per-query VALU kernel, fp32 everywhere.
#include <stdlib.h>

#include "attn_frag.h"
#include "gemm.h"

namespace osud {

namespace {

__device__ __forceinline__ float fast_exp2(float v) { return __builtin_amdgcn_exp2f(v); }

// Tile map of an attention mask, one workgroup per 64-query x 64-key tile:
//   kb_class[qb][kb] = 0 if every (query, key) pair of the tile is masked, 2 if none is (and all its keys < T), else 1.
// The reference's long-sequence sampling uses a banded mask (sample.py:81-84): with this map the attention kernels skip
// the fully masked tiles (O(T x band) work instead of O(T^2)) and read mask bytes only on the band's edge tiles.
__global__ __launch_bounds__(256) void mask_tiles_kernel(const uint8_t* __restrict__ mask, int T, int nkb,
                                                         uint8_t* __restrict__ kb_class) {
  __shared__ int cnt;
  const int q0 = blockIdx.y * 64, kb = blockIdx.x;
  if (threadIdx.x == 0) cnt = 0;
  __syncthreads();
  int c = 0;
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int r = q0 + (e >> 6), k = kb * 64 + (e & 63);
    if (r < T && k < T && mask[(size_t)r * T + k] == 0) ++c;
  }
  if (c) atomicAdd(&cnt, c);
  __syncthreads();
  if (threadIdx.x == 0) {
    const int rows = T - q0 < 64 ? (T - q0 > 0 ? T - q0 : 0) : 64;  // real queries of this block
    kb_class[(size_t)blockIdx.y * nkb + kb] = cnt == 0 ? 0 : ((kb * 64 + 64 <= T && cnt == rows * 64) ? 2 : 1);
  }
}

// HD = head_dim (64, or 72 for DiT-XL), HDP = HD padded to a multiple of 32 (zero columns: they add nothing to q.k and give
// zero output columns that are not stored)
// (4 waves per SIMD for the 64-wide head: the kernel has only two key blocks per workgroup at T = 128 and hides its load
// latency through occupancy; prefetching the next block into registers instead was measured neutral-to-slower)
// KB = key blocks (of 64 keys) staged per round.  KB = 2 puts ALL loads of a T <= 128 workgroup (Q fragments, both K / V
// blocks) in flight at once -- one memory round trip and one barrier instead of two.  Built and measured SLOWER (same box,
// alternating runs: sampling step 4.41 vs 4.21 ms, training step 28.91 vs 28.59 ms): 32 KiB of LDS per workgroup and eight
// staging loads per thread cost more occupancy-side latency hiding than the saved round trip buys.  KB = 1 stays the default
// (OSUD_ATTN_KB=2 selects the other form for A/B runs).
// X3 (split-bf16 tier): qk and out are plane pairs (rows [hi | lo], see common.h): ld_qk and D count LOGICAL columns, the lo plane
// sits ld_qk (resp. D) elements behind the hi plane inside a row of twice that length.  Both products take the three-term form
//   S = K_hi.q_hi + K_lo.q_hi + K_hi.q_lo        O += V_hi.p_hi + V_lo.p_hi + V_hi.p_lo      (p = exp2(..) in fp32, split in registers)
// with K / V tiles of both planes in LDS; softmax statistics in fp32 as before.
template <int HD, int HDP, int KB, bool X3 = false, bool F16 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((HD == 64 && !X3) ? 4 : 2))) void attn_bf16_kernel(const bf16_t* __restrict__ qk,
                                                        const uint8_t* __restrict__ mask, bf16_t* __restrict__ out,
                                                        float* __restrict__ lse, int T, int Tp, int Mp, int D,
                                                        int ld_qk, float c1 /* scale*log2(e) */,
                                                        const uint8_t* __restrict__ kb_class, float fp8_scale) {
  using TL = AttnTile<HDP>;
  constexpr int KS = HDP / 16, DT = HDP / 32, CPR = TL::CPR;
  constexpr int NP = X3 ? 2 : 1;  // planes
  constexpr int TILE = 64 * TL::RS;
  __shared__ __attribute__((aligned(16))) char Ks_all[KB * NP * TILE];
  __shared__ __attribute__((aligned(16))) char Vs_all[KB * NP * TILE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 31, fhalf = lane >> 5;
  const int n = blockIdx.z, h = blockIdx.y;
  const int q = blockIdx.x * 128 + wave * 32 + frow;
  const int qc = q < Tp ? q : Tp - 1;
  const size_t ldq = (size_t)ld_qk * NP;  // row stride: 3D (Q|K|V), twice that for a plane pair
  const size_t lo_q = (size_t)ld_qk;      // X3: distance of the lo plane inside a row
  const size_t mrow = (size_t)n * Tp + qc;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  // Q fragments (B operand of S^T = K.Q^T): 8 consecutive d per lane and k-step
  u32x4 qf[KS], ql[X3 ? KS : 1];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    qf[ks] = (ks * 16 + fhalf * 8) < HD ? *reinterpret_cast<const u32x4*>(qk + mrow * ldq + h * HD + ks * 16 + fhalf * 8) : zero4;
    if constexpr (X3)
      ql[ks] = (ks * 16 + fhalf * 8) < HD ? *reinterpret_cast<const u32x4*>(qk + mrow * ldq + lo_q + h * HD + ks * 16 + fhalf * 8) : zero4;
  }

  f32x16 o[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const int qm = qc < T ? qc : T - 1;  // row of the mask this query reads

  const int nkb = Tp / 64, nkb_all = nkb;
  for (int kb0 = 0; kb0 < nkb; kb0 += KB) {
    if (kb0 * 64 >= T) break;  // whole round is padding
    // which blocks of this round are live (workgroup-uniform): not padding, not fully masked
    bool live[KB], chk[KB];
    bool any = false;
#pragma unroll
    for (int j = 0; j < KB; ++j) {
      const int kb = kb0 + j;
      live[j] = kb < nkb && kb * 64 < T;
      chk[j] = mask != nullptr;
      if (live[j] && kb_class != nullptr) {  // skip fully masked tiles, read no mask bytes on fully open ones
        const int ca = kb_class[(size_t)(2 * blockIdx.x) * nkb_all + kb];
        const int cb = 2 * blockIdx.x + 1 < nkb_all ? kb_class[(size_t)(2 * blockIdx.x + 1) * nkb_all + kb] : ca;
        if (ca == 0 && cb == 0) live[j] = false;
        chk[j] = !(ca == 2 && cb == 2);
      }
      any = any || live[j];
    }
    if (!any) continue;
    __syncthreads();          // previous round fully consumed
#pragma unroll
    for (int j = 0; j < KB; ++j) {
      if (!live[j]) continue;
      for (int idx = tid; idx < 64 * CPR; idx += 256) {
        const int r = idx / CPR, cp = idx % CPR;
        const bf16_t* src = qk + ((size_t)n * Tp + (kb0 + j) * 64 + r) * ldq + h * HD + cp * 8;
        const bool real = cp * 8 < HD;
        *reinterpret_cast<u32x4*>(Ks_all + j * NP * TILE + TL::off(r, cp)) = real ? *reinterpret_cast<const u32x4*>(src + D) : zero4;
        *reinterpret_cast<u32x4*>(Vs_all + j * NP * TILE + TL::off(r, cp)) = real ? *reinterpret_cast<const u32x4*>(src + 2 * D) : zero4;
        if constexpr (X3) {
          *reinterpret_cast<u32x4*>(Ks_all + j * NP * TILE + TILE + TL::off(r, cp)) = real ? *reinterpret_cast<const u32x4*>(src + lo_q + D) : zero4;
          *reinterpret_cast<u32x4*>(Vs_all + j * NP * TILE + TILE + TL::off(r, cp)) = real ? *reinterpret_cast<const u32x4*>(src + lo_q + 2 * D) : zero4;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KB; ++j) {
    if (!live[j]) continue;
    const int kb = kb0 + j;
    const bool check_mask = chk[j];
    const char* Ks = Ks_all + j * NP * TILE;
    const char* Vs = Vs_all + j * NP * TILE;

    // ---- S^T tiles: s[kt][4g+i] = score(key = kb*64 + kt*32 + 8g + 4*fhalf + i, query q)
    f32x16 s[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const u32x4 kh = rowfrag<HDP>(Ks, kt * 32 + frow, 2 * ks + fhalf);
        if constexpr (X3) {  // small terms first
          s[kt] = mfma_h<F16>(rowfrag<HDP>(Ks + TILE, kt * 32 + frow, 2 * ks + fhalf), qf[ks], s[kt]);
          s[kt] = mfma_h<F16>(kh, ql[ks], s[kt]);
        }
        s[kt] = mfma_h<F16>(kh, qf[ks], s[kt]);
      }
    }
    // ---- scale, mask, block max
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int key0 = kb * 64 + kt * 32 + 8 * g + 4 * fhalf;
        // the 4 mask bytes of this lane's 4 consecutive keys: one 32-bit load when rows are 4-byte aligned (T % 4 == 0)
        // (-7...-10 % per long-sequence sampling step vs byte loads; bit-packing the mask first was slower again: the
        // packing pass runs every forward)
        uint32_t m4 = 0;
        if (check_mask && key0 < T) {
          const uint8_t* mp = mask + (size_t)qm * T + key0;
          if ((T & 3) == 0) {
            m4 = *reinterpret_cast<const uint32_t*>(mp);
          } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (key0 + i < T) m4 |= (uint32_t)mp[i] << (8 * i);
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int key = key0 + i;
          bool dead = key >= T;
          if (check_mask && !dead) dead = ((m4 >> (8 * i)) & 0xffu) != 0;
          const float v = dead ? -INFINITY : s[kt][4 * g + i] * c1;
          s[kt][4 * g + i] = v;
          mx = fmaxf(mx, v);
        }
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float m_use = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = fast_exp2(m_run - m_use);  // m_run = -inf -> 0
    float psum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = fast_exp2(s[kt][r] - m_use);
        s[kt][r] = p;
        psum += p;
      }
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[i][r] *= alpha;

    // ---- O^T += V^T . P^T ; k-step (kt, ss) covers keys kt*32 + 16ss + {4*fhalf + 0..3, 8 + 4*fhalf + 0..3}
    //      (V^T fragments = transposing reads of the row-major V tile)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const u32x4 pf = pack8_h<F16>(s[kt], 8 * ss);
        if constexpr (X3) {
          const u32x4 pl = pack8_lo(s[kt], 8 * ss, pf);
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            const u32x4 vh = trfrag<HDP>(Vs, kt * 32 + 16 * ss, dt * 32, lane);
            o[dt] = mfma_h<F16>(trfrag<HDP>(Vs + TILE, kt * 32 + 16 * ss, dt * 32, lane), pf, o[dt]);
            o[dt] = mfma_h<F16>(vh, pl, o[dt]);
            o[dt] = mfma_h<F16>(vh, pf, o[dt]);
          }
        } else {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = mfma_h<F16>(trfrag<HDP>(Vs, kt * 32 + 16 * ss, dt * 32, lane), pf, o[dt]);
        }
      }
    }  // blocks of the round
  }

  if (q < Tp) {
    if (lse != nullptr && fhalf == 0)  // log2-domain logsumexp of the scaled scores, for the backward pass
      lse[((size_t)n * gridDim.y + h) * Tp + q] = m_run + __builtin_amdgcn_logf(l_run);
    const float inv = (fp8_scale > 0.f ? fp8_scale : 1.0f) / l_run;
    const bool h8_out = X3 && fp8_scale < 0.f && fp8_scale > -1.5f;  // split-bf16 arithmetic, output in the fp16 + e4m3 operand form
    const bool w8_out = X3 && fp8_scale <= -1.5f;                    // ... or as fp16 + e4m3(v) activation rows (OSUD_PREC_F16W8)
    bf16_t* orow = out + ((size_t)n * Tp + q) * D * NP + h * HD;
    fp8_t* orow8 = reinterpret_cast<fp8_t*>(out) + ((size_t)n * Tp + q) * D + h * HD;  // fp8 tier: e4m3 operand of out_proj
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = dt * 32 + 8 * g + 4 * fhalf;
        if (d < HD) {
          if constexpr (X3) {
            // (mul_rn: the streamed kernel of the window shape must produce the same bits, see attn_frag.h)
            const float v0 = mul_rn(o[dt][4 * g + 0], inv), v1 = mul_rn(o[dt][4 * g + 1], inv), v2 = mul_rn(o[dt][4 * g + 2], inv),
                        v3 = mul_rn(o[dt][4 * g + 3], inv);
            if (w8_out) store4_w8<false>(reinterpret_cast<w8_t*>(out) + ((size_t)n * Tp + q) * D, h * HD + d, v0, v1, v2, v3);
            else if (h8_out) store4_h8<false>(reinterpret_cast<h8_t*>(out) + ((size_t)n * Tp + q) * D, h * HD + d, v0, v1, v2, v3);  // (OSUD_PREC_F16F8: out_proj reads fp16 + e4m3 rows)
            else store4_x3(orow + d, (size_t)D, v0, v1, v2, v3);
          }
          else if (fp8_scale > 0.f) store4(orow8 + d, o[dt][4 * g + 0] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
          else if (F16) store4(reinterpret_cast<f16_t*>(orow) + d, o[dt][4 * g + 0] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
          else store4(orow + d, o[dt][4 * g + 0] * inv, o[dt][4 * g + 1] * inv, o[dt][4 * g + 2] * inv, o[dt][4 * g + 3] * inv);
        }
      }
  }
}

// ---------------------------------------------------------------- any T, masks, head_dim 64: key blocks double-buffered by LDS-DMA
// The kernel above pays a full memory round trip per 64-key block (barrier, global -> registers -> LDS, barrier, compute): one long
// beatmap of T = 2048 spends 7.6 us per block and workgroup, of which the MFMAs need 0.25.  Same work split here (128 queries per
// workgroup, a wave owns 32; online softmax over 64-key blocks; tile classes of the mask), but the K | V rows of the NEXT live block
// travel by LDS-DMA into the other half of a 32 KiB double buffer while this block is computed: one barrier per block, no staging
// registers.  The mask bytes of an edge tile (eight 32-bit loads per lane) are inline asm, issued in front of the DMA and waited
// for with a count that leaves the DMA in flight -- a compiler-visible load there would wait for the pieces it cannot see.
// Mask rows must be 4-byte aligned (T % 4 == 0; other T take the kernel above).  Output rows leave as full lines through the
// (by then idle) stage memory.
template <bool F16 = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) void attn_bf16_dma_kernel(
    const bf16_t* __restrict__ qk, const uint8_t* __restrict__ mask, bf16_t* __restrict__ out, float* __restrict__ lse, int T, int Tp,
    int D, int ld_qk, float c1 /* scale*log2(e) */, const uint8_t* __restrict__ kb_class) {
  constexpr int HD = 64, HDP = 64, KS = 4, DT = 2;
  using TL = AttnTile<HDP>;
  constexpr int TILE = 64 * TL::RS, STAGE = 2 * TILE;  // 8 KiB K + 8 KiB V
  __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];
  const uint32_t lds0 = (uint32_t)(size_t)(const __attribute__((address_space(3))) void*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 31, fhalf = lane >> 5;
  const int n = blockIdx.z, h = blockIdx.y;
  const int q = blockIdx.x * 128 + wave * 32 + frow;
  const int qc = q < Tp ? q : Tp - 1;
  const size_t ldq = (size_t)ld_qk;
  const size_t mrow = (size_t)n * Tp + qc;
  const int nkb = Tp / 64;
  // tile classes of this workgroup's two 64-query rows of the mask map: live = not padding, not fully masked; check = read mask bytes
  auto classify = [&](int kb, bool& check) -> bool {
    if (kb >= nkb || kb * 64 >= T) return false;
    check = mask != nullptr;
    if (kb_class != nullptr) {
      const int ca = kb_class[(size_t)(2 * blockIdx.x) * nkb + kb];
      const int cb = 2 * (int)blockIdx.x + 1 < nkb ? kb_class[(size_t)(2 * blockIdx.x + 1) * nkb + kb] : ca;
      if (ca == 0 && cb == 0) return false;
      check = !(ca == 2 && cb == 2);
    }
    return true;
  };
  auto next_live = [&](int from, bool& check) -> int {  // first live block >= from, nkb if none (workgroup-uniform)
    int kb = from;
    while (kb < nkb && !classify(kb, check)) ++kb;
    return kb < nkb && kb * 64 < T ? kb : nkb;
  };
  // LDS-DMA of one key block: wave w fetches 8-row pieces 4 w .. 4 w + 3 of K | V (16 pieces); source-side XOR swizzle of AttnTile<64>
  const uint32_t ldb = (uint32_t)(ldq * 2);
  const int lr = lane >> 3, pc = lane & 7;
  const uint32_t voff_even = (uint32_t)lr * ldb + (uint32_t)((pc ^ (lr >> 1)) << 4);
  const uint32_t voff_odd = (uint32_t)lr * ldb + (uint32_t)((pc ^ (4 + (lr >> 1))) << 4);
  auto issue = [&](int kb, int stage) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int p = 4 * wave + i, t = p >> 3, pp = p & 7;  // tile (K, V), piece inside the tile
      const char* sb = reinterpret_cast<const char*>(qk + ((size_t)n * Tp + kb * 64 + pp * 8) * ldq + (size_t)(1 + t) * D + h * HD);
      const uint32_t dst = lds0 + (uint32_t)(stage * STAGE + t * TILE + pp * 1024);
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"((pp & 1) ? voff_odd : voff_even), "s"(sb), "s"(dst) : "memory");
    }
  };
  bool chk_cur = false, chk_nxt = false;
  int cur = next_live(0, chk_cur);
  if (cur < nkb) issue(cur, 0);
  // Q fragments (B operand of S^T = K.Q^T): 8 consecutive d per lane and k-step (compiler loads: their wait also covers the
  // first block's pieces, which the first iteration needs anyway)
  u32x4 qf[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const u32x4*>(qk + mrow * ldq + h * HD + ks * 16 + fhalf * 8);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));  // (the compiler's wait for them sits HERE, not at their first use
                                                                    //  inside the loop, where it would drain the DMA every block)
  f32x16 o[DT];
#pragma unroll
  for (int i = 0; i < DT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const int qm = qc < T ? qc : T - 1;  // row of the mask this query reads
  int stage = 0;
  while (cur < nkb) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this block's pieces (nothing else is in flight here)
    __syncthreads();
    const int nxt = next_live(cur + 1, chk_nxt);
    const char* Ks = smem + stage * STAGE;
    const char* Vs = Ks + TILE;
    // the 4 mask bytes of this lane's 4 consecutive keys per (kt, g): eight 32-bit loads, in front of the next block's DMA
    uint32_t m4[2][4];
    if (chk_cur) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          int key0 = cur * 64 + kt * 32 + 8 * g + 4 * fhalf;
          if (key0 > T - 4) key0 = T - 4;  // (keys past T are dead below whatever is read here; T % 4 == 0)
          const uint8_t* mp = mask + (size_t)qm * T + key0;
          asm volatile("global_load_dword %0, %1, off" : "=&v"(m4[kt][g]) : "v"(mp) : "memory");
        }
    }
    if (nxt < nkb) issue(nxt, stage ^ 1);
    // ---- S^T tiles: s[kt][4g+i] = score(key = cur*64 + kt*32 + 8g + 4*fhalf + i, query q)
    f32x16 s[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) s[kt] = mfma_h<F16>(rowfrag<HDP>(Ks, kt * 32 + frow, 2 * ks + fhalf), qf[ks], s[kt]);
    }
    if (chk_cur) {
      if (nxt < nkb) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");  // the mask words; the 4 DMA pieces behind them stay in flight
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(m4[kt][g]));
    }
    // ---- scale, mask, block max
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int key0 = cur * 64 + kt * 32 + 8 * g + 4 * fhalf;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          bool dead = key0 + i >= T;
          if (chk_cur && !dead) dead = ((m4[kt][g] >> (8 * i)) & 0xffu) != 0;
          const float v = dead ? -INFINITY : s[kt][4 * g + i] * c1;
          s[kt][4 * g + i] = v;
          mx = fmaxf(mx, v);
        }
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float m_use = m_new == -INFINITY ? 0.f : m_new;
    const float alpha = fast_exp2(m_run - m_use);  // m_run = -inf -> 0
    float psum = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = fast_exp2(s[kt][r] - m_use);
        s[kt][r] = p;
        psum += p;
      }
    psum += __shfl_xor(psum, 32, 64);
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
    // ---- O^T += V^T . P^T (V^T fragments = transposing reads of the row-major V tile)
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const u32x4 pf = pack8_h<F16>(s[kt], 8 * ss);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) o[dt] = mfma_h<F16>(trfrag<HDP>(Vs, kt * 32 + 16 * ss, dt * 32, lane), pf, o[dt]);
      }
    cur = nxt;
    chk_cur = chk_nxt;
    stage ^= 1;
  }
  __syncthreads();  // the stages are idle: they become the waves' output patches
  if (blockIdx.x * 128 + wave * 32 < Tp) {  // (Tp % 64 == 0: a wave's 32 rows are all inside or all outside)
    if (lse != nullptr && fhalf == 0)  // log2-domain logsumexp of the scaled scores, for the backward pass
      lse[((size_t)n * gridDim.y + h) * Tp + q] = m_run + __builtin_amdgcn_logf(l_run);
    const float inv = 1.0f / l_run;
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[i][r] *= inv;
    store_rows_patch<false, F16>(smem + wave * 2048, out + ((size_t)n * Tp + blockIdx.x * 128 + wave * 32) * D + h * HD, (size_t)D, o, lane);
  }
}

// ---------------------------------------------------------------- T == Tp == 128, head_dim 64, no mask: persistent, streamed
// The shape of training and of window sampling.  The kernel above is a 48 KB-in / 16 KB-out copy per head with a little MFMA
// work: at four workgroups per CU it reached 4.3 TB/s in training and 3.1 TB/s at the sampling batch (a device copy: 5.4).  Here one
// eight-wave workgroup per CU walks PAIRS of heads (waves 0-3 the first, 4-7 the second; a wave owns 32 queries): the K and V
// tiles of the next pair travel global -> LDS by LDS-DMA (XOR swizzle applied on the source side, see attention_bwd.hip) into the
// other half of a 128 KiB double buffer while this pair is computed; Q fragments are fetched into registers one pair ahead; the
// output rows leave through per-wave LDS patches as full 128-byte lines.  All 128 keys are scored before the softmax (four
// 32 x 32 score tiles in registers), so there is no running maximum to rescale by.
template <int T, bool F16 = false>
__global__ __launch_bounds__(512) void attn_fwd_stream_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                              float* __restrict__ lse, int D, int H, int items, float c1,
                                                              unsigned* __restrict__ queue) {
  constexpr int HD = 64, HDP = 64, KS = 4, DT = 2;
  using TL = AttnTile<HDP>;
  constexpr int TILE = T * TL::RS;  // bytes
  static_assert(T == 128, "four waves x 32 queries per head, four 32-key score tiles");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][2 heads][K | V][TILE] | store patches [8][2 KiB]
  char* patch = smem + 8 * TILE + (threadIdx.x >> 6) * 2048;
  // shared-GPU mode (queue != nullptr): pairs b, b + G, then 2 G + ticket, drawn one pair ahead -- see attn_bwd_stream_kernel
  volatile lds_u32* tword = reinterpret_cast<volatile lds_u32*>((lds_void*)(smem + 8 * TILE + 8 * 2048));
  const uint32_t lds0 = (uint32_t)(size_t)(const __attribute__((address_space(3))) void*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 31, fhalf = lane >> 5;
  const size_t ld3 = 3 * (size_t)D;
  const int npairs = (items + 1) / 2;
  // LDS-DMA: wave w fetches half w % 2 of tile w / 2 (head (w / 2) / 2 of the pair, K or V = (w / 2) % 2)
  const int tile = wave >> 1, thead = tile >> 1, tkv = tile & 1;
  const uint32_t ldb = (uint32_t)(ld3 * 2);
  const int lr = lane >> 3, pc = lane & 7;
  const uint32_t voff_even = (uint32_t)lr * ldb + (uint32_t)((pc ^ (lr >> 1)) << 4);
  const uint32_t voff_odd = (uint32_t)lr * ldb + (uint32_t)((pc ^ (4 + (lr >> 1))) << 4);
  auto issue = [&](int pair, int buf) {
    const int item = 2 * pair + thead;
    if (item >= items) return;  // odd head count: the last pair is half empty
    const int n = item / H, h = item - n * H;
    const int p0 = (wave & 1) * (T / 16);
    const char* base = reinterpret_cast<const char*>(qkv + (size_t)n * T * ld3 + (size_t)(1 + tkv) * D + h * HD) + (size_t)p0 * 8 * ldb;
    const uint32_t dst0 = lds0 + (uint32_t)((buf * 4 + tile) * TILE + p0 * 1024);
#pragma unroll
    for (int pp = 0; pp < T / 16; ++pp) {
      const char* sb = base + (size_t)pp * 8 * ldb;
      const uint32_t dst = dst0 + pp * 1024;
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"((pp & 1) ? voff_odd : voff_even), "s"(sb), "s"(dst) : "memory");
    }
  };
  const int hw = wave >> 2, own = (wave & 3) * 32;  // which head of the pair, first query of this wave
  auto fetch_q = [&](int pair, u32x4 (&qf)[KS]) {
    int item = 2 * pair + hw;
    if (item >= items) item = items - 1;  // (the empty half of an odd last pair reads a valid head and stores nothing: no select
                                          //  on the loaded value, which would pull the wait for it right behind the load)
    const int n = item / H, h = item - n * H;
    const bf16_t* src = qkv + ((size_t)n * T + own + frow) * ld3 + h * HD + fhalf * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const u32x4*>(src + ks * 16);
  };
  const int G = gridDim.x;
  const bool ticket_lane = queue != nullptr && tid == 0;
  int it = blockIdx.x, nx = blockIdx.x + G, buf = 0, iter = 0;
  u32x4 qf[KS], qn[KS];
  if (it < npairs) {
    issue(it, 0);
    fetch_q(it, qf);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));  // (the compiler's wait for these registers: here, see below)
  }
  for (; it < npairs; it = nx, buf ^= 1, ++iter) {
    const char* Ks = smem + (size_t)((buf * 2 + hw) * 2) * TILE;
    const char* Vs = Ks + TILE;
    __syncthreads();  // every wave has waited for its pieces of this pair; the other buffer is free from here on
    if (iter > 0) nx = queue != nullptr ? 2 * G + __builtin_amdgcn_readfirstlane((int)tword[iter & 1]) : it + G;  // (iteration 0: b + G)
    uint32_t tk;  // (no initialiser: writing the register at the loop top would first wait for last head's ticket AND stores)
    if (ticket_lane) tk = __hip_atomic_fetch_add(queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (nx < npairs) {
      issue(nx, buf ^ 1);
      fetch_q(nx, qn);
    }
    const int item = 2 * it + hw;
    f32x16 o[DT];
    float m_row = 0.f, l_row = 1.f;
    if (item < items) {
      // ---- S^T tiles: s[kt][4g+i] = score(key = kt*32 + 8g + 4*fhalf + i, query own + frow)
      f32x16 s[4];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) s[kt] = mfma_h<F16>(rowfrag<HDP>(Ks, kt * 32 + frow, 2 * ks + fhalf), qf[ks], s[kt]);
      }
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s[kt][r] *= c1;
          mx = fmaxf(mx, s[kt][r]);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float psum = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = fast_exp2(s[kt][r] - mx);
          s[kt][r] = p;
          psum += p;
        }
      psum += __shfl_xor(psum, 32, 64);
      m_row = mx;
      l_row = psum;
#pragma unroll
      for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
      // ---- O^T += V^T . P^T (V^T fragments = transposing reads of the row-major V tile)
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
          const u32x4 pf = pack8_h<F16>(s[kt], 8 * ss);
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) o[dt] = mfma_h<F16>(trfrag<HDP>(Vs, kt * 32 + 16 * ss, dt * 32, lane), pf, o[dt]);
        }
      const float inv = 1.0f / l_row;
#pragma unroll
      for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] *= inv;
    }
    // Everything issued so far -- the next pair's DMA pieces and Q fragments -- has had the whole pair to land: wait for it HERE,
    // before this pair's stores are issued, and let the compiler see the fragments used (its own wait for them then sits here as
    // well and not behind the stores and the DMA it cannot see), so that the stores drain while the next pair is computed.
    if (nx < npairs) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        asm volatile("" : "+v"(qn[ks]));
        qf[ks] = qn[ks];
      }
    }
    if (ticket_lane) tword[(iter + 1) & 1] = tk;  // (the ticket has had the whole pair to return)
    if (item < items) {
      const int n = item / H, h = item - n * H;
      if (lse != nullptr && fhalf == 0)  // log2-domain logsumexp of the scaled scores, for the backward pass
        lse[((size_t)n * H + h) * T + own + frow] = m_row + __builtin_amdgcn_logf(l_row);
      store_rows_patch<false, F16>(patch, out + ((size_t)n * T + own) * D + h * HD, (size_t)D, o, lane);
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

// ---------------------------------------------------------------- the same shape on split-bf16 operands (the tolerance tiers' attention)
// attn_bf16_kernel<64, 64, 1, true> pays a memory round trip per 64-key block behind two barriers: 60 us per launch at the sampling
// batch where its 201 MB need 37 at the rate of a device copy.  This is the streamed structure of attn_fwd_stream_kernel for plane
// pairs: one eight-wave workgroup per CU walks PAIRS of heads (waves 0-3 the first, 4-7 the second; a wave owns 32 queries) in
// STAGES of 64 keys -- K_hi | K_lo | V_hi | V_lo of both heads = eight 8 KiB tiles = 64 KiB, one tile per wave by LDS-DMA -- into the
// other half of a 128 KiB double buffer while this stage is computed, with the online softmax of the general kernel across a head's
// two stages.  The arithmetic -- three-term products with the small terms first, the rescale, the order of the key blocks -- is the
// general kernel's, instruction for instruction: the results are bit-identical (tests/test_gpu_x3.py).  Output forms as there:
// plane pairs (bf16x3), fp16 + e4m3 rows (fp16f8: mode -1) or fp16 + e4m3(v) rows (fp16w8: mode -2).  Rows leave through per-wave
// LDS patches as whole 64- / 128-byte segments (see the end of the kernel).
template <int T>
__global__ __launch_bounds__(512) void attn_fwd_stream_x3_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int D, int H,
                                                                 int items, float c1, float mode) {
  constexpr int HD = 64, HDP = 64, KS = 4, DT = 2, KB = 64;
  using TL = AttnTile<HDP>;
  constexpr int TILE = KB * TL::RS;  // 8 KiB
  static_assert(T == 128, "four waves x 32 queries per head, two 64-key stages");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 buffers][2 heads][K_hi | K_lo | V_hi | V_lo][TILE]
  const uint32_t lds0 = (uint32_t)(size_t)(const __attribute__((address_space(3))) void*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 31, fhalf = lane >> 5;
  const size_t ld3 = 3 * (size_t)D, ldrow = 2 * ld3;  // a row is [hi plane | lo plane], each 3 D long
  const int npairs = (items + 1) / 2;
  // LDS-DMA: wave w fetches tile w of a stage: head w / 4 of the pair, K or V = (w / 2) % 2, plane w % 2; 8 pieces of 8 rows
  const int thead = wave >> 2, tkv = (wave >> 1) & 1, tpl = wave & 1;
  const uint32_t ldb = (uint32_t)(ldrow * 2);
  const int lr = lane >> 3, pc = lane & 7;
  const uint32_t voff_even = (uint32_t)lr * ldb + (uint32_t)((pc ^ (lr >> 1)) << 4);
  const uint32_t voff_odd = (uint32_t)lr * ldb + (uint32_t)((pc ^ (4 + (lr >> 1))) << 4);
  auto issue = [&](int pair, int half, int buf) {
    const int item = 2 * pair + thead;
    if (item >= items) return;  // odd head count: the last pair is half empty
    const int n = item / H, h = item - n * H;
    const char* base = reinterpret_cast<const char*>(qkv + ((size_t)n * T + half * KB) * ldrow + (size_t)tpl * ld3 + (size_t)(1 + tkv) * D + h * HD);
    const uint32_t dst0 = lds0 + (uint32_t)((buf * 8 + wave) * TILE);
#pragma unroll
    for (int pp = 0; pp < KB / 8; ++pp) {
      const char* sb = base + (size_t)pp * 8 * ldb;
      const uint32_t dst = dst0 + pp * 1024;
      asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"((pp & 1) ? voff_odd : voff_even), "s"(sb), "s"(dst) : "memory");
    }
  };
  const int hw = wave >> 2, own = (wave & 3) * 32;  // which head of the pair, first query of this wave
  auto fetch_q = [&](int pair, u32x4 (&qh)[KS], u32x4 (&ql)[KS]) {
    int item = 2 * pair + hw;
    if (item >= items) item = items - 1;  // (the empty half of an odd last pair reads a valid head and stores nothing)
    const int n = item / H, h = item - n * H;
    const bf16_t* src = qkv + ((size_t)n * T + own + frow) * ldrow + h * HD + fhalf * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      qh[ks] = *reinterpret_cast<const u32x4*>(src + ks * 16);
      ql[ks] = *reinterpret_cast<const u32x4*>(src + ld3 + ks * 16);
    }
  };
  const int G = gridDim.x;
  int it = blockIdx.x;
  u32x4 qf[KS], qlo[KS], qn[KS], qnl[KS];
  if (it < npairs) {
    issue(it, 0, 0);
    fetch_q(it, qf, qlo);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]), "+v"(qlo[ks]));  // (the compiler's wait for these registers: here)
  }
  int buf = 0;
  for (; it < npairs; it += G) {
    const int nx = it + G;
    const int item = 2 * it + hw;
    f32x16 o[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
#pragma unroll
    for (int half = 0; half < 2; ++half, buf ^= 1) {
      __syncthreads();  // every wave has waited for its tile of this stage; the other buffer is free from here on
      if (half == 0) issue(it, 1, buf ^ 1);
      else if (nx < npairs) {
        issue(nx, 0, buf ^ 1);
        fetch_q(nx, qn, qnl);
      }
      if (item < items) {
        const char* Ks = smem + (size_t)((buf * 8 + hw * 4) * TILE);
        const char* Vs = Ks + 2 * TILE;
        // ---- S^T tiles: s[kt][4g+i] = score(key = half*64 + kt*32 + 8g + 4*fhalf + i, query own + frow)
        f32x16 s[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
          for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
          for (int ks = 0; ks < KS; ++ks) {
            const u32x4 kh = rowfrag<HDP>(Ks, kt * 32 + frow, 2 * ks + fhalf);
            s[kt] = mfma_h<false>(rowfrag<HDP>(Ks + TILE, kt * 32 + frow, 2 * ks + fhalf), qf[ks], s[kt]);  // small terms first
            s[kt] = mfma_h<false>(kh, qlo[ks], s[kt]);
            s[kt] = mfma_h<false>(kh, qf[ks], s[kt]);
          }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = mul_rn(s[kt][r], c1);  // (the ROUNDED product, as in the general kernel, whose select on the mask keeps the compiler
            s[kt][r] = v;                           //  from contracting it into the subtraction below; here it would, and the bits would differ)
            mx = fmaxf(mx, v);
          }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float m_use = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = fast_exp2(m_run - m_use);  // m_run = -inf -> 0
        float psum = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float pv = fast_exp2(s[kt][r] - m_use);
            s[kt][r] = pv;
            psum += pv;
          }
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < DT; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
        // ---- O^T += V^T . P^T (V^T fragments = transposing reads of the row-major V tiles)
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
          for (int ss = 0; ss < 2; ++ss) {
            const u32x4 pf = pack8_h<false>(s[kt], 8 * ss);
            const u32x4 pl = pack8_lo(s[kt], 8 * ss, pf);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
              const u32x4 vh = trfrag<HDP>(Vs, kt * 32 + 16 * ss, dt * 32, lane);
              o[dt] = mfma_h<false>(trfrag<HDP>(Vs + TILE, kt * 32 + 16 * ss, dt * 32, lane), pf, o[dt]);
              o[dt] = mfma_h<false>(vh, pl, o[dt]);
              o[dt] = mfma_h<false>(vh, pf, o[dt]);
            }
          }
      }
      // everything issued in this stage -- the next stage's tiles, the next pair's Q rows -- has had the whole stage to land: wait for
      // it HERE, before the pair's stores are issued, so that the stores drain under the next stage
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (half == 1 && nx < npairs) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          asm volatile("" : "+v"(qn[ks]), "+v"(qnl[ks]));
          qf[ks] = qn[ks];
          qlo[ks] = qnl[ks];
        }
      }
    }
    if (item < items) {
      // Output rows through the wave's 4 KiB LDS patch, one 32-column half (dt) at a time: a lane owns a ROW of the accumulators, so direct
      // stores touch 32 rows with 4-16 bytes each per instruction -- 16-22 of the kernel's 46-52 us (tools/attn_x3_bench.py, stores
      // compiled out: 29-30 us).  The half-row image is 128 bytes in every form -- plane pairs: hi 64 | lo 64; fp16 + e4m3 rows: exactly one
      // K-blocked group of 32 (fp16 64 | e4m3 32 | e4m3 32); fp16 + e4m3(v) rows: fp16 64 | e4m3 32 -- its 16-byte chunks XOR-swizzled
      // with (row >> 1) & 7, and leaves as 16 bytes per lane, eight lanes per row: whole 64- / 128-byte segments per store instruction.
      // (mul_rn: the rounded product, as in the general kernel -- attn_frag.h; tests/test_gpu_x3.py holds the two kernels to the same bits.)
      const int n = item / H, h = item - n * H;
      const size_t row0 = (size_t)n * T + own;
      const float inv = 1.0f / l_run;
      const bool h8_out = mode < 0.f && mode > -1.5f, w8_out = mode <= -1.5f;
      char* const patch = smem + 2 * 8 * TILE + wave * 4096;
      char* const prow = patch + frow * 128;
      const int sw = (frow >> 1) & 7;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const float v0 = mul_rn(o[dt][4 * g + 0], inv), v1 = mul_rn(o[dt][4 * g + 1], inv), v2 = mul_rn(o[dt][4 * g + 2], inv),
                      v3 = mul_rn(o[dt][4 * g + 3], inv);
          uint2 hi;
          if (w8_out) {
            uint32_t p8;
            pack4_w8<false>(v0, v1, v2, v3, hi, p8);
            *reinterpret_cast<uint32_t*>(prow + (((4 + (g >> 1)) ^ sw) << 4) + 8 * (g & 1) + 4 * fhalf) = p8;
          } else if (h8_out) {
            uint32_t p64, p96;
            pack4_h8<false>(v0, v1, v2, v3, hi, p64, p96);
            *reinterpret_cast<uint32_t*>(prow + (((4 + (g >> 1)) ^ sw) << 4) + 8 * (g & 1) + 4 * fhalf) = p64;
            *reinterpret_cast<uint32_t*>(prow + (((6 + (g >> 1)) ^ sw) << 4) + 8 * (g & 1) + 4 * fhalf) = p96;
          } else {
            uint2 lo;
            pack4_x3(v0, v1, v2, v3, hi, lo);
            *reinterpret_cast<uint2*>(prow + (((4 + g) ^ sw) << 4) + 8 * fhalf) = lo;
          }
          *reinterpret_cast<uint2*>(prow + ((g ^ sw) << 4) + 8 * fhalf) = hi;
        }
        asm volatile("" ::: "memory");  // (LDS operations of a wave execute in order; this only pins the compiler's order)
        const int x0 = h * HD + dt * 32;  // first logical column of this half
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int rr = 8 * k + (lane >> 3), c = lane & 7;
          const u32x4 v = *reinterpret_cast<const u32x4*>(patch + rr * 128 + ((c ^ ((rr >> 1) & 7)) << 4));
          const size_t row = row0 + rr;
          if (w8_out) {
            char* gsg = reinterpret_cast<char*>(out) + row * D * 3 + (size_t)(x0 >> 7) * 384;
            const int i = x0 & 127;
            if (c < 4) *reinterpret_cast<u32x4*>(gsg + 2 * i + c * 16) = v;
            else if (c < 6) *reinterpret_cast<u32x4*>(gsg + 256 + i + (c - 4) * 16) = v;
          } else if (h8_out) {
            *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(out) + row * D * 4 + (size_t)(x0 >> 5) * 128 + c * 16) = v;
          } else {
            *reinterpret_cast<u32x4*>(out + row * D * 2 + (c < 4 ? 0 : D) + x0 + (c & 3) * 8) = v;
          }
        }
        asm volatile("" ::: "memory");
      }
    }
  }
}

// ---------------------------------------------------------------- T == Tp == 256, head_dim 72 (DiT-XL), no mask: persistent, streamed
// One eight-wave workgroup per CU walks heads (a wave owns 32 of the 256 queries); the K | V tile pairs of 128 keys (52 KiB, rows of
// 208 bytes, pad columns read from a zero chunk) arrive by LDS-DMA in a two-stage ring, one step ahead, across head boundaries;
// the next head's Q rows are fetched behind the second step; every load in the loop is inline asm and the waits are counted by
// hand (see attn_bwd_stream72_kernel); online softmax across the two key blocks; the sixth k-step (columns 80..95: all pad) is
// skipped; output rows leave as whole 144-byte rows through LDS patches.
//   per wave and head:  step 0: P DMA pieces | step 1: P pieces ... 5 fragment loads (next head's Q rows), 6 stores (+1: lse)
template <int T>
__global__ __launch_bounds__(512, 2) void attn_fwd_stream72_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                                   float* __restrict__ lse, int D, int H, int items, float c1,
                                                                   unsigned* __restrict__ queue) {
  constexpr int HD = 72, HDP = 96, KS = 5, DT = 3, BLK = 128;
  using TL = AttnTile<HDP>;
  constexpr int TILE = BLK * TL::RS, STAGE = 2 * TILE;
  static_assert(T == 256 && TL::RS == 208, "eight waves x 32 queries, two 128-key blocks");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 stages][K | V][TILE] | patches [8][16 x 208]
  char* patch = smem + 2 * STAGE + (threadIdx.x >> 6) * (16 * 208);
  // shared-GPU mode (queue != nullptr): heads b, b + G, then 2 G + ticket, requested one head ahead in step 0 (returned by the
  // vmcnt(0) of step 1, outside the counted wait) and published through LDS for the next head's first barrier
  volatile lds_u32* tword = reinterpret_cast<volatile lds_u32*>((lds_void*)(smem + 2 * STAGE + 8 * 16 * 208));
  const uint32_t lds0 = (uint32_t)(size_t)(const __attribute__((address_space(3))) void*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int frow = lane & 31, fhalf = lane >> 5;
  const size_t ld3 = 3 * (size_t)D;
  const int own = wave * 32;
  const int p_first = wave < 4 ? 7 * wave : 28 + 6 * (wave - 4), p_count = wave < 4 ? 7 : 6;
  const char* zsrc = reinterpret_cast<const char*>(&g_attn_zero16);
  auto issue = [&](int head, int blk, int stage) {
    const int n = head / H, h = head - n * H;
    const char* xb = reinterpret_cast<const char*>(qkv + ((size_t)n * T + blk * BLK) * ld3 + D + h * HD);
    const char* yb = xb + (size_t)D * 2;
    const size_t ldb = ld3 * 2;
    int lane_o = lane;
    asm volatile("" : "+v"(lane_o));  // (opaque: the per-piece addresses are recomputed here, not kept in registers)
#pragma unroll
    for (int q = 0; q < 7; ++q) {
      if (q < p_count) {
        const int p = p_first + q, t = p >= 26 ? 1 : 0, pp = p - 26 * t;  // wave-uniform
        const int gi = pp * 64 + lane_o, row = (gi * 5042) >> 16, c = gi - 13 * row;
        const char* src = c < 9 ? (t ? yb : xb) + (size_t)row * ldb + c * 16 : zsrc;
        const uint32_t dst = lds0 + (uint32_t)(stage * STAGE + t * TILE + pp * 1024);
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(dst) : "memory");
      }
    }
  };
  u32x4 qf[KS];
  auto fetch_q = [&](int head) {
    const int n = head / H, h = head - n * H;
    const bf16_t* pq = qkv + ((size_t)n * T + own + frow) * ld3 + h * HD;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) gload16(qf[ks], pq + ((ks == 4 && fhalf) ? 8 : 2 * ks + fhalf) * 8);
  };
  auto settle_q = [&]() {
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(qf[ks]));
    if (fhalf) qf[4] = zero4;  // chunk 9: pad columns 72..79
  };
  const int G = gridDim.x;
  const bool ticket_lane = queue != nullptr && tid == 0;
  int it = blockIdx.x, nx = blockIdx.x + G, iter = 0;
  uint32_t tk;  // (no initialiser: see attn_bwd_stream_kernel)
  if (it < items) {
    issue(it, 0, 0);
    fetch_q(it);
    OSUD_VM_WAIT(0);
  }
  for (; it < items; it = nx, ++iter) {
    const int n = it / H, h = it - n * H;
    f32x16 o[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
#pragma unroll
    for (int step = 0; step < 2; ++step) {
      // step 0: this head's first tiles (issued in step 1 of the previous head) and Q rows, the 6 output stores behind them;
      // step 1: tiles issued in step 0, nothing behind them
      if (step == 0) { OSUD_VM_WAIT(6); } else { OSUD_VM_WAIT(0); }
      if (step == 0) settle_q();
      __syncthreads();
      if (step == 0 && iter > 0) nx = queue != nullptr ? 2 * G + __builtin_amdgcn_readfirstlane((int)tword[iter & 1]) : it + G;
      if (step == 0) {
        if (ticket_lane) tk = __hip_atomic_fetch_add(queue, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        issue(it, 1, 1);
      } else {
        if (ticket_lane) tword[(iter + 1) & 1] = tk;  // (returned: the wait above was vmcnt(0))
        issue(nx < items ? nx : it, 0, 0);  // (past the last head: a harmless re-read into the free stage)
      }
      const char* Ks = smem + step * STAGE;
      const char* Vs = Ks + TILE;
      f32x16 s[4];
#pragma unroll
      for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kt][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) s[kt] = mfma_bf16(rowfrag<HDP>(Ks, kt * 32 + frow, 2 * ks + fhalf), qf[ks], s[kt]);
      }
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s[kt][r] *= c1;
          mx = fmaxf(mx, s[kt][r]);
        }
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      const float alpha = fast_exp2(m_run - m_new);  // first block: exp2(-inf) = 0
      float psum = 0.f;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = fast_exp2(s[kt][r] - m_new);
          s[kt][r] = p;
          psum += p;
        }
      psum += __shfl_xor(psum, 32, 64);
      l_run = l_run * alpha + psum;
      m_run = m_new;
#pragma unroll
      for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int ss = 0; ss < 2; ++ss) {
          const u32x4 pf = pack8(s[kt], 8 * ss);
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) o[dt] = mfma_bf16(trfrag<HDP>(Vs, kt * 32 + 16 * ss, dt * 32, lane), pf, o[dt]);
        }
    }
    fetch_q(nx < items ? nx : it);  // in front of the stores; waited for with the next head's first tiles
    const float inv = 1.0f / l_run;
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[i][r] *= inv;
    store_rows_patch72(patch, out + ((size_t)n * T + own) * D + h * HD, (size_t)D, o, lane);
    if (lse != nullptr && fhalf == 0) lse[((size_t)n * H + h) * T + own + frow] = m_run + __builtin_amdgcn_logf(l_run);
  }
  if (ticket_lane) {  // the last workgroup out re-arms the counters
    const unsigned done = __hip_atomic_fetch_add(queue + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done == (unsigned)G - 1) {
      __hip_atomic_store(queue, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(queue + 8, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---------------------------------------------------------------- parity tier (fp32, VALU)
template <int HD>
__global__ __launch_bounds__(64) void attn_f32_kernel(const float* __restrict__ qk,
                                                      const uint8_t* __restrict__ mask, float* __restrict__ out,
                                                      float* __restrict__ lse, int T, int Tp, int Mp, int D, int ld_qk,
                                                      float scale, const uint8_t* __restrict__ kb_class) {
  __shared__ float Ks[64][HD];
  __shared__ float Vs[HD][64];
  const int tid = threadIdx.x, n = blockIdx.z, h = blockIdx.y;
  const int q = blockIdx.x * 64 + tid;  // Tp % 64 == 0 -> always < Tp
  const size_t ldq = (size_t)ld_qk;
  float qv[HD], o[HD];
  const float* qrow = qk + ((size_t)n * Tp + q) * ldq + h * HD;
#pragma unroll
  for (int d = 0; d < HD; ++d) {
    qv[d] = qrow[d] * scale;  // reference scales q before the product (F.multi_head_attention_forward)
    o[d] = 0.f;
  }
  float m_run = -INFINITY, l_run = 0.f;
  const int qm = q < T ? q : T - 1;
  for (int kb = 0; kb * 64 < T; ++kb) {
    bool check_mask = mask != nullptr;
    if (kb_class != nullptr) {
      const int cls = kb_class[(size_t)blockIdx.x * (Tp / 64) + kb];
      if (cls == 0) continue;
      check_mask = cls != 2;
    }
    __syncthreads();
    for (int idx = tid; idx < 64 * HD; idx += 64) {
      const int r = idx / HD, d = idx % HD;
      Ks[r][d] = qk[((size_t)n * Tp + kb * 64 + r) * ldq + D + h * HD + d];
    }
    for (int idx = tid; idx < 64 * HD; idx += 64) {
      const int j = idx / HD, d = idx % HD;
      Vs[d][j] = qk[((size_t)n * Tp + kb * 64 + j) * ldq + 2 * D + h * HD + d];
    }
    __syncthreads();
    for (int j = 0; j < 64; ++j) {
      const int key = kb * 64 + j;
      if (key >= T) break;
      if (check_mask && mask[(size_t)qm * T + key]) continue;
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) s = fmaf(qv[d], Ks[j][d], s);
      float p;
      if (s > m_run) {
        const float a = expf(m_run - s);
        l_run = l_run * a + 1.0f;
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] *= a;
        m_run = s;
        p = 1.0f;
      } else {
        p = expf(s - m_run);
        l_run += p;
      }
#pragma unroll
      for (int d = 0; d < HD; ++d) o[d] = fmaf(p, Vs[d][j], o[d]);
    }
  }
  if (lse != nullptr) lse[((size_t)n * gridDim.y + h) * Tp + q] = m_run + logf(l_run);
  const float inv = 1.0f / l_run;
  float* orow = out + ((size_t)n * Tp + q) * D + h * HD;
#pragma unroll
  for (int d = 0; d < HD; ++d) orow[d] = o[d] * inv;
}

}  // namespace

int launch_mask_tiles(const uint8_t* mask, int T, int Tp, uint8_t* kb_class, hipStream_t st) {
  OSUD_CHECK_ARG(mask && kb_class && T > 0 && Tp % 64 == 0, "mask_tiles: bad arguments");
  hipLaunchKernelGGL(mask_tiles_kernel, dim3(Tp / 64, Tp / 64), dim3(256), 0, st, mask, T, Tp / 64, kb_class);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

int launch_attention(int prec, const void* qk, int ld_qk, const uint8_t* mask, void* out, float* lse, int N, int T, int Tp,
                     int Mp, int heads, int head_dim, hipStream_t st, const uint8_t* kb_class, float fp8_scale) {
  OSUD_CHECK_ARG(N > 0 && T > 0 && Tp >= T && Tp % 64 == 0 && Mp >= N * Tp, "attention: bad sizes N=%d T=%d Tp=%d Mp=%d", N,
                 T, Tp, Mp);
  const int D = heads * head_dim;
  OSUD_CHECK_ARG(ld_qk >= 3 * D, "attention: packed q|k|v rows need ld >= 3*hidden (got %d)", ld_qk);
  const float scale = 1.0f / sqrtf((float)head_dim);
  if (prec == OSUD_PREC_BF16) {
    if (head_dim != 64 && head_dim != 72) {
      set_error("attention: the bf16 tier is built for head_dim 64 and 72 (got %d); use the f32 tier", head_dim);
      return OSUD_ERR_UNSUPPORTED;
    }
    const bool stream_ok = opt(OPT_ATTN_FWD_KERNEL) == 0;  // osud_set_option("attn_fwd_kernel", 1 | 2): the general kernels on this shape too (tests)
    if (head_dim == 64 && T == 128 && Tp == 128 && mask == nullptr && fp8_scale <= 0.f && ld_qk == 3 * D && stream_ok) {
      constexpr size_t slds = (size_t)8 * 128 * AttnTile<64>::RS + 8 * 2048 + 16;
      OSUD_BIG_LDS_ONCE(attn_fwd_stream_kernel<128>);
      const int cus = device_cus();
      const int items = N * heads, npairs = (items + 1) / 2;
      hipLaunchKernelGGL((attn_fwd_stream_kernel<128>), dim3(npairs < cus ? npairs : cus), dim3(512), slds, st, (const bf16_t*)qk,
                         (bf16_t*)out, lse, D, heads, items, scale * 1.4426950408889634f,
                         (gemm_dynamic_tiles_on() && npairs > 2 * cus) ? gemm_ticket_slot() : nullptr);
      OSUD_HIP(hipGetLastError());
      return OSUD_OK;
    }
    if (head_dim == 72 && T == 256 && Tp == 256 && mask == nullptr && fp8_scale <= 0.f && ld_qk == 3 * D && stream_ok) {
      constexpr size_t lds72 = (size_t)4 * 128 * AttnTile<96>::RS + 8 * 16 * 208 + 16;
      OSUD_BIG_LDS_ONCE(attn_fwd_stream72_kernel<256>);
      const int cus = device_cus();
      const int items = N * heads;
      hipLaunchKernelGGL((attn_fwd_stream72_kernel<256>), dim3(items < cus ? items : cus), dim3(512), lds72, st, (const bf16_t*)qk,
                         (bf16_t*)out, lse, D, heads, items, scale * 1.4426950408889634f,
                         (gemm_dynamic_tiles_on() && items > 2 * cus) ? gemm_ticket_slot() : nullptr);
      OSUD_HIP(hipGetLastError());
      return OSUD_OK;
    }
    dim3 grid((Tp + 127) / 128, heads, N);
    const bool dma_ok = opt(OPT_ATTN_FWD_KERNEL) < 2;  // (2: the register-staged kernel)
    if (head_dim == 64 && fp8_scale <= 0.f && (mask == nullptr || T % 4 == 0) && T >= 4 && dma_ok) {
      hipLaunchKernelGGL(attn_bf16_dma_kernel<false>, grid, dim3(256), 0, st, (const bf16_t*)qk, mask, (bf16_t*)out, lse, T, Tp, D, ld_qk,
                         scale * 1.4426950408889634f, mask ? kb_class : nullptr);
      OSUD_HIP(hipGetLastError());
      return OSUD_OK;
    }
    // (two key blocks per round trip -- attn_bf16_kernel<.., 2> -- measured slower, 4.41 vs 4.21 ms per sampling step: not instantiated)
    if (head_dim == 64)
      hipLaunchKernelGGL((attn_bf16_kernel<64, 64, 1>), grid, dim3(256), 0, st, (const bf16_t*)qk, mask, (bf16_t*)out, lse, T, Tp,
                         Mp, D, ld_qk, scale * 1.4426950408889634f, mask ? kb_class : nullptr, fp8_scale);
    else
      hipLaunchKernelGGL((attn_bf16_kernel<72, 96, 1>), grid, dim3(256), 0, st, (const bf16_t*)qk, mask, (bf16_t*)out, lse, T, Tp,
                         Mp, D, ld_qk, scale * 1.4426950408889634f, mask ? kb_class : nullptr, fp8_scale);
  } else if (prec == OSUD_PREC_F16) {
    // fp16 tier (inference): the bf16 tier's forward kernels on half operands -- the streamed kernel at the window shape, the LDS-DMA
    // kernel for other lengths and masks at head_dim 64, the general kernel for the rest
    OSUD_CHECK_ARG(lse == nullptr && fp8_scale <= 0.f, "attention: the fp16 tier is inference only");
    if (head_dim != 64 && head_dim != 72) {
      set_error("attention: the fp16 tier is built for head_dim 64 and 72 (got %d); use the f32 tier", head_dim);
      return OSUD_ERR_UNSUPPORTED;
    }
    if (head_dim == 64 && T == 128 && Tp == 128 && mask == nullptr && ld_qk == 3 * D) {
      constexpr size_t slds = (size_t)8 * 128 * AttnTile<64>::RS + 8 * 2048 + 16;
      {
        auto* kern = attn_fwd_stream_kernel<128, true>;
        OSUD_BIG_LDS_ONCE(kern);
      }
      const int cus = device_cus();
      const int items = N * heads, npairs = (items + 1) / 2;
      hipLaunchKernelGGL((attn_fwd_stream_kernel<128, true>), dim3(npairs < cus ? npairs : cus), dim3(512), slds, st, (const bf16_t*)qk,
                         (bf16_t*)out, lse, D, heads, items, scale * 1.4426950408889634f,
                         (gemm_dynamic_tiles_on() && npairs > 2 * cus) ? gemm_ticket_slot() : nullptr);
      OSUD_HIP(hipGetLastError());
      return OSUD_OK;
    }
    dim3 grid((Tp + 127) / 128, heads, N);
    if (head_dim == 64 && (mask == nullptr || T % 4 == 0) && T >= 4)
      hipLaunchKernelGGL(attn_bf16_dma_kernel<true>, grid, dim3(256), 0, st, (const bf16_t*)qk, mask, (bf16_t*)out, lse, T, Tp, D, ld_qk,
                         scale * 1.4426950408889634f, mask ? kb_class : nullptr);
    else if (head_dim == 64)
      hipLaunchKernelGGL((attn_bf16_kernel<64, 64, 1, false, true>), grid, dim3(256), 0, st, (const bf16_t*)qk, mask, (bf16_t*)out, lse, T, Tp,
                         Mp, D, ld_qk, scale * 1.4426950408889634f, mask ? kb_class : nullptr, 0.f);
    else
      hipLaunchKernelGGL((attn_bf16_kernel<72, 96, 1, false, true>), grid, dim3(256), 0, st, (const bf16_t*)qk, mask, (bf16_t*)out, lse, T, Tp,
                         Mp, D, ld_qk, scale * 1.4426950408889634f, mask ? kb_class : nullptr, 0.f);
  } else if (prec == OSUD_PREC_BF16X3 || prec == OSUD_PREC_F16F8 || prec == OSUD_PREC_F16W8) {
    // (F16F8: the split-bf16 kernel on split-bf16 q | k | v, its output written as fp16 + e4m3 rows for out_proj: flag = scale < 0)
    OSUD_CHECK_ARG(lse == nullptr && fp8_scale <= 0.f, "attention: the split-bf16 tier is inference only");
    const float flag = prec == OSUD_PREC_F16F8 ? -1.0f : (prec == OSUD_PREC_F16W8 ? -2.0f : 0.f);  // (the output form of the split-bf16 kernel)
    if (head_dim == 64 && T == 128 && Tp == 128 && mask == nullptr && ld_qk == 3 * D && opt(OPT_ATTN_FWD_KERNEL) == 0) {  // the window shape: streamed
      constexpr size_t slds = (size_t)16 * 64 * AttnTile<64>::RS + 8 * 4096;  // double buffer + a 4 KiB output patch per wave = 160 KiB
      OSUD_BIG_LDS_ONCE(attn_fwd_stream_x3_kernel<128>);
      const int cus = device_cus();
      const int items = N * heads, npairs = (items + 1) / 2;
      hipLaunchKernelGGL((attn_fwd_stream_x3_kernel<128>), dim3(npairs < cus ? npairs : cus), dim3(512), slds, st, (const bf16_t*)qk, (bf16_t*)out, D,
                         heads, items, scale * 1.4426950408889634f, flag);
      OSUD_HIP(hipGetLastError());
      return OSUD_OK;
    }
    dim3 grid((Tp + 127) / 128, heads, N);
    if (head_dim == 64)
      hipLaunchKernelGGL((attn_bf16_kernel<64, 64, 1, true>), grid, dim3(256), 0, st, (const bf16_t*)qk, mask, (bf16_t*)out, lse, T, Tp,
                         Mp, D, ld_qk, scale * 1.4426950408889634f, mask ? kb_class : nullptr, flag);
    else if (head_dim == 72)
      hipLaunchKernelGGL((attn_bf16_kernel<72, 96, 1, true>), grid, dim3(256), 0, st, (const bf16_t*)qk, mask, (bf16_t*)out, lse, T, Tp,
                         Mp, D, ld_qk, scale * 1.4426950408889634f, mask ? kb_class : nullptr, flag);
    else {
      set_error("attention: head_dim %d not built (64, 72)", head_dim);
      return OSUD_ERR_UNSUPPORTED;
    }
  } else {
    dim3 grid(Tp / 64, heads, N);
    if (head_dim == 64)
      hipLaunchKernelGGL(attn_f32_kernel<64>, grid, dim3(64), 0, st, (const float*)qk, mask,
                         (float*)out, lse, T, Tp, Mp, D, ld_qk, scale, mask ? kb_class : nullptr);
    else if (head_dim == 72)
      hipLaunchKernelGGL(attn_f32_kernel<72>, grid, dim3(64), 0, st, (const float*)qk, mask,
                         (float*)out, lse, T, Tp, Mp, D, ld_qk, scale, mask ? kb_class : nullptr);
    else {
      set_error("attention: head_dim %d not built (64, 72)", head_dim);
      return OSUD_ERR_UNSUPPORTED;
    }
  }
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

}  // namespace osud
