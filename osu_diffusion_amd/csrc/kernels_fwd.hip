// Memory-bound forward kernels of the DiT path (everything that is not a GEMM or the
// attention core).  All are templated on the tier's element type TE (bf16 storage / f32).
//
// Activation row layout: row m = n * Tp + t, Tp = tokens per sample rounded up to 64, rows
// with t >= T (and rows >= N*Tp up to Mp) are padding: finite, never read back by the caller.
#include <type_traits>

#include "kernels.h"

namespace osud {

namespace {

// ------------------------------------------------------------------------------------------
// Token embedding features (reference FirstLayer.forward, models.py:227-235 with
// positional_embedding.py:29-77): per token
//   [cos,sin](512*x0*f) | [cos,sin](384*x1*f) | [cos,sin]((o/10)*f) | c[0..E)      f = 64 freqs
// written as one TE row of Kp (>= 384+E, zero padded) — the A operand of the first GEMM.
// x, c are channel-major (N,2,T) / (N,E,T): T is the contiguous axis, so each block takes 32
// consecutive tokens (TOK = 16), reads along T coalesced, transposes through LDS and writes whole rows.
// Accurate sincosf on purpose: arguments reach ~1.4e4 rad.
// SPLIT (bf16 tier): the features are fine-grained functions of the sampled coordinates (512 rad per unit x), and rounding them
// to bf16 was the largest single contribution to the fast tier's deviation from the fp32 path (tools/analysis/
// bf16_error_budget.py: as much as all 48 trunk GEMMs of DiT-S together).  The row is therefore written as
// [hi | lo | hi] (3 x Kp columns, hi = bf16(v), lo = bf16(v - hi)) against weights packed as [w_hi | w_hi | w_lo]: one bf16
// GEMM with K = 3 Kp computes hi*w_hi + lo*w_hi + hi*w_lo, i.e. the fp32 product to ~2^-17 relative.
// SPLIT = 2 (split-bf16 tier): the row is the tier's plane pair [hi | lo] (2 x Kp columns) against weights packed [w_hi | w_lo]; the
// GEMM forms the three products itself (gemm_kernel.h).
template <typename TE, int SPLIT>
__global__ __launch_bounds__(256) void embed_kernel(const float* __restrict__ x, const float* __restrict__ o,
                                                    const float* __restrict__ c, const float* __restrict__ freqs64,
                                                    float pf0, float pf1, TE* __restrict__ out, int N, int T, int Tp,
                                                    int E, int Kp, int x_dup_half, int mode) {
  // mode 0: the whole row.  The sampler loop splits the first linear into its step-invariant part (offsets and context: mode 2 =
  // the whole row with the coordinate features zeroed, multiplied ONCE per loop) and the part that follows x (mode 1 = a compact row
  // of the 256 coordinate features only, Kp = 256, E = 0: 40 % of the columns and of the sincos work per step)
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int TOK = 16;
  const int ldr = SPLIT == 1 ? 3 * Kp : (SPLIT == 2 ? 2 * Kp : Kp);  // row length in elements
  TE* tile = reinterpret_cast<TE*>(smem_raw);  // [TOK][ldr]
  const int tid = threadIdx.x;
  const int m0 = blockIdx.x * TOK;
  const int n = m0 / Tp, t0 = m0 % Tp;  // Tp % TOK == 0 -> one sample per block
  if (n >= N) {                          // rows past the last sample: zero
    for (int i = tid; i < TOK * ldr; i += 256) store_elem(out + (size_t)m0 * ldr + i, 0.f);
    return;
  }
  auto put = [&](int tok, int col, float v) {
    TE* r = tile + tok * ldr;
    store_elem(r + col, v);
    if (SPLIT) {
      const float hi = load_elem(r + col);
      store_elem(r + Kp + col, v - hi);
      if (SPLIT == 1) r[2 * Kp + col] = r[col];
    }
  };
  const int nx = (x_dup_half > 0 && n >= x_dup_half) ? n - x_dup_half : n;  // forward_with_cfg: cat([half, half])
  // (a) sin/cos features: TOK tokens x 3 scalars x 64 frequencies
  const int nfeat = mode == 1 ? 128 : 192;  // scalar features x 64 frequencies per token
  for (int idx = tid; idx < TOK * nfeat; idx += 256) {
    const int tok = idx / nfeat, rem = idx % nfeat, which = rem >> 6, k = rem & 63;
    const int t = t0 + tok;
    float cs = 0.f, sn = 0.f;
    if (t < T && !(mode == 2 && which < 2)) {
      float v;
      if (which == 0) v = x[((size_t)nx * 2 + 0) * T + t] * pf0;        // models.py:229
      else if (which == 1) v = x[((size_t)nx * 2 + 1) * T + t] * pf1;
      else v = o[(size_t)n * T + t] / 10.0f;                            // models.py:232
      const float arg = v * freqs64[k];
      sincosf(arg, &sn, &cs);
    }
    put(tok, which * 128 + k, cs);       // cos first (positional_embedding.py:46)
    put(tok, which * 128 + 64 + k, sn);
  }
  // (b) context rows, read along T
  for (int idx = tid; idx < (mode == 1 ? 0 : TOK * E); idx += 256) {
    const int e = idx / TOK, tok = idx % TOK;
    const int t = t0 + tok;
    const float v = t < T ? c[((size_t)n * E + e) * T + t] : 0.f;
    put(tok, 384 + e, v);
  }
  if (mode != 1)
    for (int idx = tid; idx < TOK * (Kp - 384 - E); idx += 256) {
      const int tok = idx / (Kp - 384 - E), k = idx % (Kp - 384 - E);
      put(tok, 384 + E + k, 0.f);
    }
  __syncthreads();
  // (c) whole rows out, 16 bytes per lane
  const int n16 = TOK * ldr * (int)sizeof(TE) / 16;
  const uint4* src = reinterpret_cast<const uint4*>(smem_raw);
  uint4* dst = reinterpret_cast<uint4*>(out + (size_t)m0 * ldr);
  for (int i = tid; i < n16; i += 256) dst[i] = src[i];
}

// Timestep frequency embedding (models.py:35-36): [cos,sin](t * f_k), 128 frequencies.
template <typename TE>
__global__ void temb_kernel(const int64_t* __restrict__ t, const float* __restrict__ freqs128, TE* __restrict__ out,
                            int N) {
  const int n = blockIdx.x, k = threadIdx.x;  // 128 threads
  float cs = 0.f, sn = 0.f;
  if (n < N) sincosf((float)t[n] * freqs128[k], &sn, &cs);
  if constexpr (std::is_same<TE, x3_t>::value) {
    bf16_t* row = reinterpret_cast<bf16_t*>(out) + (size_t)n * 512;
    store_elem_x3(row + k, 256, cs);
    store_elem_x3(row + 128 + k, 256, sn);
  } else {
  store_elem(out + (size_t)n * 256 + k, cs);
  store_elem(out + (size_t)n * 256 + 128 + k, sn);
  }
}

// b = t_emb + table[y]; keep b (fp32, for backward) and silu(b) (TE, operand of every adaLN GEMM).
template <typename TE>
__global__ void cond_kernel(const float* __restrict__ tvec, const float* __restrict__ table,
                            const int64_t* __restrict__ y, int table_rows, float* __restrict__ b_out,
                            TE* __restrict__ sb_out, int N, int D, const int64_t* __restrict__ t_index) {
  // t_index != nullptr (sampler loops): tvec is a table with one row per schedule index, made once per loop; row n reads its step's
  constexpr bool kX3 = std::is_same<TE, x3_t>::value;
  constexpr bool FAST = sizeof(TE) == 2 && !kX3;
  const int n = blockIdx.x;
  auto put_sb = [&](int d, float v) {
    if constexpr (kX3) store_elem_x3(reinterpret_cast<bf16_t*>(sb_out) + (size_t)n * 2 * D + d, (size_t)D, v);
    else store_elem(sb_out + (size_t)n * D + d, v);
  };
  if (n >= N) {
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
      b_out[(size_t)n * D + d] = 0.f;
      put_sb(d, 0.f);
    }
    return;
  }
  int64_t cls = y[n];
  cls = cls < 0 ? 0 : (cls >= table_rows ? table_rows - 1 : cls);
  const float* trow = tvec + (size_t)(t_index != nullptr ? t_index[n] : (int64_t)n) * D;
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    const float b = trow[d] + table[(size_t)cls * D + d];  // models.py:320
    b_out[(size_t)n * D + d] = b;
    put_sb(d, silu_t<FAST>(b));
  }
}

// LayerNorm (eps 1e-6, no affine, biased variance) + adaLN modulate (models.py:12-13,160,173):
//   u = LN(h) * (1 + scale[n]) + shift[n]        one wave per token row, D/64 values per lane
// The gated residual update of the PREVIOUS branch rides along (models.py:161-175): with br != nullptr the row is
//   h' = h + gate[n] * br,  written to h_out (may be h itself), and u = LN(h') ...
// so the branch GEMMs (attention out-projection, fc2) keep a plain bias epilogue and the residual stream is read
// once instead of twice.  `out` may alias `br` (a wave reads its whole row before it writes).
// TWIN (fp8 training, TE = bf16): the e4m3 operand of the next GEMM is written here as well -- the bf16 value times the slot's
// delayed scale, exactly what the stand-alone f8_quantize pass produced -- and this step's amax goes out as one partial maximum per
// workgroup (amax_part[blockIdx.x], reduced by f8_update_kernel: no atomics on the slot's single word).  The grid is capped at
// kF8AmaxParts workgroups there, each wave walking rows blockIdx.x * 4 + wave, + 4 * gridDim.x, ...
constexpr int kF8AmaxParts = 8192;  // (= rows / 4 of a 32 768-token step: one row per wave, as in the plain kernel)
template <typename TE, int VPL, bool TWIN = false>
__global__ __launch_bounds__(256) void ln_mod_kernel(const float* h, const float* __restrict__ ada, int ld_ada,
                                                     int off_shift, int off_scale, TE* out, float* __restrict__ stats,
                                                     int M, int Tp, int N, const TE* br, int off_gate, float* h_out,
                                                     float out_scale, fp8_t* __restrict__ out8 = nullptr,
                                                     const float* __restrict__ slot = nullptr, float* __restrict__ amax_part = nullptr) {
  // lane l owns columns W*l + 64*W*g + {0..W-1}: 16-byte fp32 accesses where the per-lane count allows (W = 4)
  constexpr int D = VPL * 64, W = (VPL % 4 == 0) ? 4 : 2, NG = VPL / W;
  const int lane = threadIdx.x & 63;
  float amax = 0.f;
  const float q_scale = TWIN ? slot[0] : 1.0f;
  for (int m = blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += TWIN ? 4 * (int)gridDim.x : M) {
  int n = m / Tp;
  if (n >= N) n = N - 1;
  const float* hr = h + (size_t)m * D;
  float v[VPL];
  float sum = 0.f;
#pragma unroll
  for (int g = 0; g < NG; ++g) loadw<W>(hr + W * lane + 64 * W * g, v + g * W);
  if (br != nullptr) {
    const TE* brow = br + (size_t)m * D;
    const float* gt = ada + (size_t)n * ld_ada + off_gate;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int d = W * lane + 64 * W * g;
      float b[W], gg[W];
      loadw<W>(brow + d, b);
      loadw<W>(gt + d, gg);
#pragma unroll
      for (int e = 0; e < W; ++e) v[g * W + e] += gg[e] * b[e];
    }
    if (h_out != nullptr) {
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        // (streaming store: the updated residual stream is read again four launches later, behind 300 MB of other outputs -- stored
        //  normally it only displaces the rows the next GEMM reads; training step -0.10 ms, profiles/r05_ab_runs.md)
        if constexpr (W == 4) {
          typedef float f4v __attribute__((ext_vector_type(4)));
          const f4v t = {v[g * W + 0], v[g * W + 1], v[g * W + 2], v[g * W + 3]};
          __builtin_nontemporal_store(t, reinterpret_cast<f4v*>(h_out + (size_t)m * D + W * lane + 64 * W * g));
        } else {
          storew<W>(h_out + (size_t)m * D + W * lane + 64 * W * g, v + g * W);
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < VPL; ++i) sum += v[i];
  const float mu = wave_sum(sum) * (1.0f / D);
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const float d = v[i] - mu;
    sq += d * d;
  }
  const float rstd = 1.0f / sqrtf(wave_sum(sq) * (1.0f / D) + 1e-6f);
  if (stats != nullptr && lane == 0) {
    stats[2 * (size_t)m] = mu;
    stats[2 * (size_t)m + 1] = rstd;
  }
  const float* sh = ada + (size_t)n * ld_ada + off_shift;
  const float* sc = ada + (size_t)n * ld_ada + off_scale;
  TE* orow = out + (size_t)m * D * Planes<TE>::k;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int d = W * lane + 64 * W * g;
    float s4[W], h4[W], r[W];
    loadw<W>(sc + d, s4);
    loadw<W>(sh + d, h4);
#pragma unroll
    for (int e = 0; e < W; ++e) {
      r[e] = (v[g * W + e] - mu) * rstd * (1.0f + s4[e]) + h4[e];
      if (sizeof(TE) == 1) r[e] *= out_scale;  // fp8 operand of the next GEMM, statically scaled
    }
    if constexpr (std::is_same<TE, x3_t>::value) storew_x3<W>(reinterpret_cast<bf16_t*>(orow) + d, (size_t)D, r);
    else if constexpr (std::is_same<TE, h8_t>::value) storew_h8<W>(orow, d, r);
    else if constexpr (std::is_same<TE, w8_t>::value) storew_w8<W>(orow, d, r);
    else if (!TWIN || out != nullptr) storew<W>(orow + d, r);  // (twin only: every consumer of this step reads the e4m3 form)
    if constexpr (TWIN) {
      float q[W];
#pragma unroll
      for (int e = 0; e < W; ++e) {
        const float rb = bf2f(f2bf(r[e]));  // the stored bf16 value: what a separate pass over `out` would have read
        amax = fmaxf(amax, fabsf(rb));
        q[e] = rb * q_scale;
      }
      if (out8 != nullptr) storew<W>(out8 + (size_t)m * D + d, q);
    }
  }
  }  // rows of this wave
  if constexpr (TWIN) {
    __shared__ float red[4];
    amax = wave_max(amax);
    if (lane == 0) red[threadIdx.x >> 6] = amax;
    __syncthreads();
    if (threadIdx.x == 0) amax_part[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  }
}

// Final layer (models.py:192-196,324): LN -> modulate -> Linear(D -> C) -> channel-major (N,C,T).
// Memory bound (one read of h); one wave per token, C (<= 4) dot products reduced in-wave.  A block takes 16 consecutive tokens of
// one sample (four per wave) and keeps the sample's 1 + scale, shift and the C weight rows in LDS: re-reading those 18 KB per token
// from the caches cost more than the token's own 3 KB (32 -> 13 us per sampling step, 78 -> 55 us in training).
// The last block's gated MLP branch is added here (br != nullptr), as in ln_mod_kernel.
template <typename TE, int VPL>
__global__ __launch_bounds__(256) void final_kernel(const float* h, const float* __restrict__ ada, int ld_ada,
                                                    int off_shift, int off_scale, const float* __restrict__ w,
                                                    const float* __restrict__ bias, float* __restrict__ out,
                                                    float* __restrict__ u_save, float* __restrict__ stats, int N, int T,
                                                    int Tp, int C, const TE* __restrict__ br, int off_gate, float* h_out) {
  constexpr int D = VPL * 64, ROWS = 16;
  __shared__ float cst[7][D];  // 1 + scale | shift | gate of the pending branch | weight rows (zero beyond C)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m0 = blockIdx.x * ROWS, n = m0 / Tp;  // Tp % 16 == 0: one sample per block
  if (n >= N) return;
  {
    const float* arow = ada + (size_t)n * ld_ada;
    for (int d = threadIdx.x; d < D; d += 256) {
      cst[0][d] = 1.0f + arow[off_scale + d];
      cst[1][d] = arow[off_shift + d];
      cst[2][d] = br != nullptr ? arow[off_gate + d] : 0.f;
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) cst[3 + ch][d] = ch < C ? w[(size_t)ch * D + d] : 0.f;
    }
  }
  __syncthreads();
  for (int r = wave; r < ROWS; r += 4) {
    const int m = m0 + r, t = m % Tp;
    if (t >= T) continue;
    const float* hr = h + (size_t)m * D;
    float v[VPL];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < VPL / 2; ++i) {
      const float2 p = *reinterpret_cast<const float2*>(hr + 2 * lane + 128 * i);
      v[2 * i] = p.x;
      v[2 * i + 1] = p.y;
    }
    if (br != nullptr) {
      const TE* brow = br + (size_t)m * D;
#pragma unroll
      for (int i = 0; i < VPL / 2; ++i) {
        const int d = 2 * lane + 128 * i;
        float b0, b1;
        load2(brow + d, b0, b1);
        const float2 g2 = *reinterpret_cast<const float2*>(&cst[2][d]);
        v[2 * i] += g2.x * b0;
        v[2 * i + 1] += g2.y * b1;
      }
      if (h_out != nullptr) {
#pragma unroll
        for (int i = 0; i < VPL / 2; ++i)
          *reinterpret_cast<float2*>(h_out + (size_t)m * D + 2 * lane + 128 * i) = make_float2(v[2 * i], v[2 * i + 1]);
      }
    }
#pragma unroll
    for (int i = 0; i < VPL; ++i) sum += v[i];
    const float mu = wave_sum(sum) * (1.0f / D);
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const float d = v[i] - mu;
      sq += d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) * (1.0f / D) + 1e-6f);
    if (stats != nullptr && lane == 0) {
      stats[2 * (size_t)m] = mu;
      stats[2 * (size_t)m + 1] = rstd;
    }
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < VPL / 2; ++i) {
      const int d = 2 * lane + 128 * i;
      const float2 s2 = *reinterpret_cast<const float2*>(&cst[0][d]);
      const float2 h2 = *reinterpret_cast<const float2*>(&cst[1][d]);
      const float a = (v[2 * i] - mu) * rstd * s2.x + h2.x;
      const float b = (v[2 * i + 1] - mu) * rstd * s2.y + h2.y;
      if (u_save != nullptr) *reinterpret_cast<float2*>(u_save + (size_t)m * D + d) = make_float2(a, b);
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) {
        const float2 w2 = *reinterpret_cast<const float2*>(&cst[3 + ch][d]);
        acc[ch] = fmaf(a, w2.x, fmaf(b, w2.y, acc[ch]));
      }
    }
#pragma unroll
    for (int ch = 0; ch < 4; ++ch)
      if (ch < C) {
        const float rr = wave_sum(acc[ch]);
        if (lane == 0) out[((size_t)n * C + ch) * T + t] = rr + bias[ch];
      }
  }
}

// forward_with_cfg tail (models.py:338-343), in place on (N, C2, T): eps channels [0,C) of both
// halves become uncond + s * (cond - uncond); the remaining channels are untouched.
__global__ void cfg_combine_kernel(float* __restrict__ out, int half, int C, int C2, int T, float s) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= half * C * T) return;
  const int t = i % T, ch = (i / T) % C, n = i / (T * C);
  float* pc = out + ((size_t)n * C2 + ch) * T + t;
  float* pu = out + ((size_t)(n + half) * C2 + ch) * T + t;
  const float ce = *pc, ue = *pu;
  const float he = ue + s * (ce - ue);
  *pc = he;
  *pu = he;
}

template <typename TE> __global__ void convert_kernel(const float* __restrict__ src, TE* __restrict__ dst, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    store_elem(dst + i, src[i]);
}

// dst[r][c] (ld_dst, zero padded to cols_dst) = src[r][c] (ld_src) for r < rows — weight packing
template <typename TE>
__global__ void pack_rows_kernel(const float* __restrict__ src, int ld_src, int cols_src, TE* __restrict__ dst,
                                 int ld_dst, int cols_dst, int rows) {
  const size_t total = (size_t)rows * cols_dst;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols_dst), cc = (int)(i % cols_dst);
    store_elem(dst + (size_t)r * ld_dst + cc, cc < cols_src ? src[(size_t)r * ld_src + cc] : 0.f);
  }
}

// bf16 tier, first linear: dst row = [w_hi | w_hi | w_lo], each part cols_dst wide (zero padded) -- the weight side of the
// split product of embed_kernel<bf16, SPLIT>
template <typename TE>  // bf16_t, or f16_t (the fp16 tier's first linear: the same three-term form on half operands)
__global__ void pack_rows_split_kernel(const float* __restrict__ src, int ld_src, int cols_src, TE* __restrict__ dst,
                                       int cols_dst, int rows) {
  const size_t total = (size_t)rows * cols_dst;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols_dst), cc = (int)(i % cols_dst);
    const float v = cc < cols_src ? src[(size_t)r * ld_src + cc] : 0.f;
    TE* d = dst + (size_t)r * 3 * cols_dst + cc;
    store_elem(d, v);
    d[cols_dst] = d[0];
    store_elem(d + 2 * cols_dst, v - load_elem(d));
  }
}

// split-bf16 tier: dst row = [w_hi | w_lo], each plane cols_dst wide (zero padded)
__global__ void pack_rows_x3_kernel(const float* __restrict__ src, int ld_src, int cols_src, bf16_t* __restrict__ dst,
                                    int cols_dst, int rows) {
  const size_t total = (size_t)rows * cols_dst;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols_dst), cc = (int)(i % cols_dst);
    store_elem_x3(dst + (size_t)r * 2 * cols_dst + cc, (size_t)cols_dst, cc < cols_src ? src[(size_t)r * ld_src + cc] : 0.f);
  }
}

// fp16 + e4m3 tier: dst row = K-blocked groups of 32 logical columns (common.h: h8_t), zero padded to cols_dst
template <bool WEIGHT> __global__ void pack_rows_h8_kernel(const float* __restrict__ src, int ld_src, int cols_src, h8_t* __restrict__ dst,
                                                            int cols_dst, int rows) {
  const size_t total = (size_t)rows * (cols_dst / 4);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / (cols_dst / 4)), cc = (int)(i % (cols_dst / 4)) * 4;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = cc + e < cols_src ? src[(size_t)r * ld_src + cc + e] : 0.f;
    store4_h8<WEIGHT>(dst + (size_t)r * cols_dst, cc, v[0], v[1], v[2], v[3]);
  }
}

// fp16 x (fp16 + e4m3) tier: dst row = K-blocked super-groups of 128 logical columns (common.h: w8_t), zero padded to cols_dst
template <bool WEIGHT> __global__ void pack_rows_w8_kernel(const float* __restrict__ src, int ld_src, int cols_src, w8_t* __restrict__ dst,
                                                            int cols_dst, int rows) {
  const size_t total = (size_t)rows * (cols_dst / 4);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / (cols_dst / 4)), cc = (int)(i % (cols_dst / 4)) * 4;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = cc + e < cols_src ? src[(size_t)r * ld_src + cc + e] : 0.f;
    store4_w8<WEIGHT>(dst + (size_t)r * cols_dst, cc, v[0], v[1], v[2], v[3]);
  }
}

// fp8 tier: one weight row (output channel) per block -> e4m3 with the row's own scale 448 / max|w|;
// dequant[x] = max|w_x| / (448 * act_scale) is what the GEMM epilogue multiplies the accumulator with
__global__ __launch_bounds__(256) void quantize_rows_kernel(const float* __restrict__ w, int cols, fp8_t* __restrict__ q,
                                                            float* __restrict__ dequant, float act_scale) {
  __shared__ float red[4];
  const float* row = w + (size_t)blockIdx.x * cols;
  float amax = 0.f;
  for (int c = threadIdx.x; c < cols; c += 256) amax = fmaxf(amax, fabsf(row[c]));
  amax = wave_max(amax);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float s = amax > 0.f ? 448.0f / amax : 1.0f;
  for (int c = threadIdx.x * 4; c < cols; c += 1024) store4(q + (size_t)blockIdx.x * cols + c, row[c] * s, row[c + 1] * s, row[c + 2] * s, row[c + 3] * s);
  if (threadIdx.x == 0) dequant[blockIdx.x] = 1.0f / (s * act_scale);
}

// bf16 source variant (the transposed weight copies of the data-gradient products exist only in bf16)
__global__ __launch_bounds__(256) void quantize_rows_bf16_kernel(const bf16_t* __restrict__ w, int cols, fp8_t* __restrict__ q,
                                                                 float* __restrict__ dequant) {
  __shared__ float red[4];
  const bf16_t* row = w + (size_t)blockIdx.x * cols;
  float amax = 0.f;
  for (int c = threadIdx.x; c < cols; c += 256) amax = fmaxf(amax, fabsf(bf2f(row[c])));
  amax = wave_max(amax);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
  __syncthreads();
  amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float s = amax > 0.f ? 448.0f / amax : 1.0f;
  for (int c = threadIdx.x * 4; c < cols; c += 1024)
    store4(q + (size_t)blockIdx.x * cols + c, bf2f(row[c]) * s, bf2f(row[c + 1]) * s, bf2f(row[c + 2]) * s, bf2f(row[c + 3]) * s);
  if (threadIdx.x == 0) dequant[blockIdx.x] = 1.0f / s;
}

// the same two kernels over a list of matrices (kernels.h: QuantList); one wave per row, 16 bytes per lane and load
template <bool SRC_BF16> __global__ __launch_bounds__(256) void quantize_rows_many_kernel(QuantList L) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int grow = blockIdx.x * 4 + wave;
  if (grow >= L.row_begin[L.count]) return;
  int s = 0;
  while (s + 1 < L.count && grow >= L.row_begin[s + 1]) ++s;
  const int r = grow - L.row_begin[s], cols = L.cols[s];
  fp8_t* q = reinterpret_cast<fp8_t*>(L.q[s]) + (size_t)r * cols;
  float amax = 0.f;
  if constexpr (SRC_BF16) {
    const bf16_t* row = reinterpret_cast<const bf16_t*>(L.src[s]) + (size_t)r * cols;
    for (int c = lane * 8; c < cols; c += 512) {
      float v[8];
      load8(row + c, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) amax = fmaxf(amax, fabsf(v[e]));
    }
    amax = wave_max(amax);
    const float sc = amax > 0.f ? 448.0f / amax : 1.0f;
    for (int c = lane * 8; c < cols; c += 512) {
      float v[8];
      load8(row + c, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] *= sc;
      store8(q + c, v);
    }
    if (lane == 0) L.dq[s][r] = 1.0f / sc;
  } else {
    const float* row = reinterpret_cast<const float*>(L.src[s]) + (size_t)r * cols;
    for (int c = lane * 4; c < cols; c += 256) {
      const float4 v = *reinterpret_cast<const float4*>(row + c);
      amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    amax = wave_max(amax);
    const float sc = amax > 0.f ? 448.0f / amax : 1.0f;
    for (int c = lane * 4; c < cols; c += 256) {
      const float4 v = *reinterpret_cast<const float4*>(row + c);
      store4(q + c, v.x * sc, v.y * sc, v.z * sc, v.w * sc);
    }
    if (lane == 0) L.dq[s][r] = 1.0f / sc;
  }
}

// fp8 training: bf16 tensor -> e4m3 with the slot's scale (delayed scaling: the scale comes from the previous step's amax),
// recording this step's amax.  16 bytes in, 8 bytes out per lane; HBM-bound (3 bytes per element).
__global__ __launch_bounds__(256) void f8_quantize_kernel(const bf16_t* __restrict__ src, fp8_t* __restrict__ dst, size_t n8,
                                                          float* __restrict__ slot) {
  __shared__ float red[4];
  const float scale = slot[0];
  float amax = 0.f;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    float v[8];
    load8(src + i * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      amax = fmaxf(amax, fabsf(v[e]));
      v[e] *= scale;
    }
    if (dst != nullptr) store8(dst + i * 8, v);
  }
  // ONE atomic per workgroup (a per-wave atomic on the single amax word cost 160 us per launch: 16 K serialised L2 atomics)
  amax = wave_max(amax);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
  __syncthreads();
  if (threadIdx.x == 0) {
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (amax > 0.f) atomicMax(reinterpret_cast<unsigned*>(slot) + 2, __float_as_uint(amax));  // amax >= 0: bit order = value order
  }
}

// one workgroup per slot: this step's amax = max(the slot's atomic word, the per-workgroup partial maxima of the producers that
// write them), scale for the next step, both cleared
__global__ __launch_bounds__(256) void f8_update_kernel(float* __restrict__ slots, int n_slots, float* __restrict__ parts) {
  __shared__ float red[4];
  const int i = blockIdx.x;
  float* s = slots + 4 * i;
  float amax = 0.f;
  if (parts != nullptr) {
    float* p = parts + (size_t)i * kF8AmaxParts;
    for (int j = threadIdx.x; j < kF8AmaxParts; j += 256) {
      amax = fmaxf(amax, p[j]);
      p[j] = 0.f;
    }
  }
  amax = wave_max(amax);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = amax;
  __syncthreads();
  if (threadIdx.x != 0) return;
  amax = fmaxf(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])), s[2]);
  if (amax > 0.f && amax < 3.0e38f) {
    const float sc = 448.0f / (2.0f * amax);
    s[0] = sc;
    s[1] = 1.0f / sc;
  }
  s[2] = 0.f;
}

}  // namespace

int launch_quantize_rows(const float* w, int rows, int cols, void* q, float* dequant, float act_scale, hipStream_t st) {
  OSUD_CHECK_ARG(cols % 4 == 0, "quantize_rows: cols=%d must be a multiple of 4", cols);
  hipLaunchKernelGGL(quantize_rows_kernel, dim3(rows), dim3(256), 0, st, w, cols, (fp8_t*)q, dequant, act_scale);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

int launch_quantize_rows_many(bool src_bf16, const QuantList& L, hipStream_t st) {
  if (L.count == 0) return OSUD_OK;
  OSUD_CHECK_ARG(L.count <= QuantList::kMax, "quantize_rows_many: list too long (%d)", L.count);
  for (int i = 0; i < L.count; ++i)
    OSUD_CHECK_ARG(L.cols[i] % 8 == 0 && L.src[i] && L.q[i] && L.dq[i], "quantize_rows_many: cols=%d must be a multiple of 8, no null pointers", L.cols[i]);
  const int rows = L.row_begin[L.count];
  if (src_bf16) hipLaunchKernelGGL(quantize_rows_many_kernel<true>, dim3((rows + 3) / 4), dim3(256), 0, st, L);
  else hipLaunchKernelGGL(quantize_rows_many_kernel<false>, dim3((rows + 3) / 4), dim3(256), 0, st, L);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

int launch_quantize_rows_bf16(const void* w, int rows, int cols, void* q, float* dequant, hipStream_t st) {
  OSUD_CHECK_ARG(cols % 4 == 0, "quantize_rows: cols=%d must be a multiple of 4", cols);
  hipLaunchKernelGGL(quantize_rows_bf16_kernel, dim3(rows), dim3(256), 0, st, (const bf16_t*)w, cols, (fp8_t*)q, dequant);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}
int launch_f8_quantize(const void* src, void* dst, size_t n, float* slot, hipStream_t st) {
  OSUD_CHECK_ARG(n % 8 == 0 && slot != nullptr, "f8_quantize: n must be a multiple of 8");
  const size_t n8 = n / 8;
  const int grid = (int)((n8 + 255) / 256 > 1024 ? 1024 : (n8 + 255) / 256);
  hipLaunchKernelGGL(f8_quantize_kernel, dim3(grid), dim3(256), 0, st, (const bf16_t*)src, (fp8_t*)dst, n8, slot);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}
int f8_amax_parts() { return kF8AmaxParts; }
int launch_f8_update(float* slots, int n_slots, hipStream_t st, float* parts) {
  hipLaunchKernelGGL(f8_update_kernel, dim3(n_slots), dim3(256), 0, st, slots, n_slots, parts);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

// ---------------------------------------------------------------------------------- launchers
template <typename TE, int SPLIT>
static int embed_t(const float* x, const float* o, const float* c, const float* freqs64, float pf0, float pf1, void* out,
                   int N, int T, int Tp, int Mp, int E, int Kp, int x_dup_half, hipStream_t st, int mode) {
  const size_t lds = (size_t)16 * Kp * sizeof(TE) * (SPLIT == 1 ? 3 : (SPLIT == 2 ? 2 : 1));
  hipLaunchKernelGGL((embed_kernel<TE, SPLIT>), dim3(Mp / 16), dim3(256), lds, st, x, o, c, freqs64, pf0, pf1, (TE*)out, N, T,
                     Tp, E, Kp, x_dup_half, mode);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}
int launch_embed(int prec, const float* x, const float* o, const float* c, const float* freqs64, float pf0, float pf1,
                 void* out, int N, int T, int Tp, int Mp, int E, int Kp, int x_dup_half, hipStream_t st, bool split, int mode) {
  const bool x3 = prec == OSUD_PREC_BF16X3;
  if (mode == 1) {  // coordinate features only: a compact row of 256 (x 3 in the split form, x 2 planes in the split-bf16 tier) columns
    OSUD_CHECK_ARG(x3 || (split && (prec == OSUD_PREC_BF16 || prec == OSUD_PREC_F16)), "embed: the coordinate-only row exists in the split forms");
    E = 0;
    Kp = 256;
  }
  OSUD_CHECK_ARG(Tp % 16 == 0 && Mp % 16 == 0 && Kp >= (mode == 1 ? 256 : 384 + E) && (Kp * elem_size(prec)) % 16 == 0, "embed: bad sizes");
  if (x3) {
    OSUD_CHECK_ARG((size_t)16 * Kp * 4 <= 64 * 1024, "embed: a plane-pair row of %d columns does not fit the 64 KiB LDS tile (context too wide)", Kp);
    return embed_t<bf16_t, 2>(x, o, c, freqs64, pf0, pf1, out, N, T, Tp, Mp, E, Kp, x_dup_half, st, mode);
  }
  OSUD_CHECK_ARG(!split || prec == OSUD_PREC_BF16 || prec == OSUD_PREC_F16, "embed: the split row form exists in the 16-bit tiers only");
  OSUD_CHECK_ARG(!split || (size_t)16 * Kp * 6 <= 64 * 1024, "embed: a split row of %d columns does not fit the 64 KiB LDS tile (context too wide)", Kp);
  if (split && prec == OSUD_PREC_F16) return embed_t<f16_t, 1>(x, o, c, freqs64, pf0, pf1, out, N, T, Tp, Mp, E, Kp, x_dup_half, st, mode);
  if (split) return embed_t<bf16_t, 1>(x, o, c, freqs64, pf0, pf1, out, N, T, Tp, Mp, E, Kp, x_dup_half, st, mode);
  if (prec == OSUD_PREC_F16) return embed_t<f16_t, 0>(x, o, c, freqs64, pf0, pf1, out, N, T, Tp, Mp, E, Kp, x_dup_half, st, mode);
  return prec == OSUD_PREC_BF16 ? embed_t<bf16_t, 0>(x, o, c, freqs64, pf0, pf1, out, N, T, Tp, Mp, E, Kp, x_dup_half, st, mode)
                                : embed_t<float, 0>(x, o, c, freqs64, pf0, pf1, out, N, T, Tp, Mp, E, Kp, x_dup_half, st, mode);
}

int launch_pack_rows_x3(const float* src, int ld_src, int cols_src, void* dst, int cols_dst, int rows, hipStream_t st) {
  const size_t total = (size_t)rows * cols_dst;
  if (total == 0) return OSUD_OK;
  const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  hipLaunchKernelGGL(pack_rows_x3_kernel, dim3(grid), dim3(256), 0, st, src, ld_src, cols_src, (bf16_t*)dst, cols_dst, rows);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

int launch_pack_rows_h8(const float* src, int ld_src, int cols_src, void* dst, int cols_dst, int rows, bool weight, hipStream_t st) {
  OSUD_CHECK_ARG(cols_dst % 32 == 0 && cols_src <= cols_dst, "pack_rows_h8: cols_dst=%d must be a multiple of 32 and >= cols_src=%d", cols_dst, cols_src);
  const size_t total = (size_t)rows * (cols_dst / 4);
  if (total == 0) return OSUD_OK;
  const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  if (weight) hipLaunchKernelGGL(pack_rows_h8_kernel<true>, dim3(grid), dim3(256), 0, st, src, ld_src, cols_src, (h8_t*)dst, cols_dst, rows);
  else hipLaunchKernelGGL(pack_rows_h8_kernel<false>, dim3(grid), dim3(256), 0, st, src, ld_src, cols_src, (h8_t*)dst, cols_dst, rows);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

int launch_pack_rows_w8(const float* src, int ld_src, int cols_src, void* dst, int cols_dst, int rows, bool weight, hipStream_t st) {
  OSUD_CHECK_ARG(cols_dst % 128 == 0 && cols_src <= cols_dst, "pack_rows_w8: cols_dst=%d must be a multiple of 128 and >= cols_src=%d", cols_dst, cols_src);
  const size_t total = (size_t)rows * (cols_dst / 4);
  if (total == 0) return OSUD_OK;
  const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  if (weight) hipLaunchKernelGGL(pack_rows_w8_kernel<true>, dim3(grid), dim3(256), 0, st, src, ld_src, cols_src, (w8_t*)dst, cols_dst, rows);
  else hipLaunchKernelGGL(pack_rows_w8_kernel<false>, dim3(grid), dim3(256), 0, st, src, ld_src, cols_src, (w8_t*)dst, cols_dst, rows);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

int launch_pack_rows_split(const float* src, int ld_src, int cols_src, void* dst, int cols_dst, int rows, hipStream_t st, int prec) {
  const size_t total = (size_t)rows * cols_dst;
  if (total == 0) return OSUD_OK;
  const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  if (prec == OSUD_PREC_F16) hipLaunchKernelGGL(pack_rows_split_kernel<f16_t>, dim3(grid), dim3(256), 0, st, src, ld_src, cols_src, (f16_t*)dst, cols_dst, rows);
  else hipLaunchKernelGGL(pack_rows_split_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, src, ld_src, cols_src, (bf16_t*)dst, cols_dst, rows);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

int launch_temb(int prec, const int64_t* t, const float* freqs128, void* out, int N, int Np, hipStream_t st) {
  if (prec == OSUD_PREC_BF16)
    hipLaunchKernelGGL((temb_kernel<bf16_t>), dim3(Np), dim3(128), 0, st, t, freqs128, (bf16_t*)out, N);
  else if (prec == OSUD_PREC_F16)
    hipLaunchKernelGGL((temb_kernel<f16_t>), dim3(Np), dim3(128), 0, st, t, freqs128, (f16_t*)out, N);
  else if (prec == OSUD_PREC_BF16X3)
    hipLaunchKernelGGL((temb_kernel<x3_t>), dim3(Np), dim3(128), 0, st, t, freqs128, (x3_t*)out, N);
  else
    hipLaunchKernelGGL((temb_kernel<float>), dim3(Np), dim3(128), 0, st, t, freqs128, (float*)out, N);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

int launch_cond(int prec, const float* tvec, const float* table, const int64_t* y, int table_rows, float* b_out,
                void* sb_out, int N, int Np, int D, hipStream_t st, const int64_t* t_index) {
  if (prec == OSUD_PREC_BF16)
    hipLaunchKernelGGL((cond_kernel<bf16_t>), dim3(Np), dim3(256), 0, st, tvec, table, y, table_rows, b_out,
                       (bf16_t*)sb_out, N, D, t_index);
  else if (prec == OSUD_PREC_F16)
    hipLaunchKernelGGL((cond_kernel<f16_t>), dim3(Np), dim3(256), 0, st, tvec, table, y, table_rows, b_out,
                       (f16_t*)sb_out, N, D, t_index);
  else if (prec == OSUD_PREC_BF16X3)
    hipLaunchKernelGGL((cond_kernel<x3_t>), dim3(Np), dim3(256), 0, st, tvec, table, y, table_rows, b_out,
                       (x3_t*)sb_out, N, D, t_index);
  else
    hipLaunchKernelGGL((cond_kernel<float>), dim3(Np), dim3(256), 0, st, tvec, table, y, table_rows, b_out,
                       (float*)sb_out, N, D, t_index);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

template <typename TE>
static int ln_mod_t(const float* h, const float* ada, int ld_ada, int off_shift, int off_scale, void* out, float* stats,
                    int M, int Tp, int N, int D, hipStream_t st, const void* br, int off_gate, float* h_out, float out_scale = 1.0f) {
  const dim3 grid((M + 3) / 4), block(256);
#define OSUD_LN(V)                                                                                                  \
  hipLaunchKernelGGL((ln_mod_kernel<TE, V>), grid, block, 0, st, h, ada, ld_ada, off_shift, off_scale, (TE*)out, \
                     stats, M, Tp, N, (const TE*)br, off_gate, h_out, out_scale)
  switch (D) {
    case 128: OSUD_LN(2); break;
    case 384: OSUD_LN(6); break;
    case 768: OSUD_LN(12); break;
    case 1024: OSUD_LN(16); break;
    case 1152: OSUD_LN(18); break;
    default: set_error("ln_mod: hidden size %d not built (128, 384, 768, 1024, 1152)", D); return OSUD_ERR_UNSUPPORTED;
  }
#undef OSUD_LN
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}
// bf16 output + its e4m3 twin and amax partials (fp8 training)
int launch_ln_mod_twin(const float* h, const float* ada, int ld_ada, int off_shift, int off_scale, void* out, float* stats, int M,
                       int Tp, int N, int D, hipStream_t st, const void* br, int off_gate, float* h_out, void* out8, const float* slot,
                       float* amax_part) {
  OSUD_CHECK_ARG(slot != nullptr && amax_part != nullptr, "ln_mod twin: slot and amax partials are required");
  const int wgs = (M + 3) / 4;
  const dim3 grid(wgs < kF8AmaxParts ? wgs : kF8AmaxParts), block(256);
#define OSUD_LN(V)                                                                                                                \
  hipLaunchKernelGGL((ln_mod_kernel<bf16_t, V, true>), grid, block, 0, st, h, ada, ld_ada, off_shift, off_scale, (bf16_t*)out, stats, \
                     M, Tp, N, (const bf16_t*)br, off_gate, h_out, 1.0f, (fp8_t*)out8, slot, amax_part)
  switch (D) {
    case 128: OSUD_LN(2); break;
    case 384: OSUD_LN(6); break;
    case 768: OSUD_LN(12); break;
    case 1024: OSUD_LN(16); break;
    case 1152: OSUD_LN(18); break;
    default: set_error("ln_mod: hidden size %d not built (128, 384, 768, 1024, 1152)", D); return OSUD_ERR_UNSUPPORTED;
  }
#undef OSUD_LN
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}
int launch_ln_mod(int prec, const float* h, const float* ada, int ld_ada, int off_shift, int off_scale, void* out,
                  float* stats, int M, int Tp, int N, int D, hipStream_t st, const void* br, int off_gate, float* h_out,
                  float fp8_scale) {
  if (fp8_scale > 0.f)  // fp8 tier: the LayerNorm output is the e4m3 operand of the next GEMM
    return ln_mod_t<fp8_t>(h, ada, ld_ada, off_shift, off_scale, out, stats, M, Tp, N, D, st, br, off_gate, h_out, fp8_scale);
  if (prec == OSUD_PREC_BF16X3) {
    OSUD_CHECK_ARG(br == nullptr, "ln_mod: the split-bf16 tier is inference only (no pending branch operand)");
    return ln_mod_t<x3_t>(h, ada, ld_ada, off_shift, off_scale, out, stats, M, Tp, N, D, st, nullptr, 0, h_out);
  }
  if (prec == OSUD_PREC_F16) {
    OSUD_CHECK_ARG(br == nullptr, "ln_mod: the fp16 tier is inference only (no pending branch operand)");
    return ln_mod_t<f16_t>(h, ada, ld_ada, off_shift, off_scale, out, stats, M, Tp, N, D, st, nullptr, 0, h_out);
  }
  if (prec == OSUD_PREC_F16F8) {  // (the trunk GEMMs' operand form inside the split-bf16 tier)
    OSUD_CHECK_ARG(br == nullptr, "ln_mod: the fp16 + e4m3 tier is inference only (no pending branch operand)");
    return ln_mod_t<h8_t>(h, ada, ld_ada, off_shift, off_scale, out, stats, M, Tp, N, D, st, nullptr, 0, h_out);
  }
  if (prec == OSUD_PREC_F16W8) {
    OSUD_CHECK_ARG(br == nullptr && D % 128 == 0, "ln_mod: the fp16 x (fp16 + e4m3) form is inference only and K-blocked in groups of 128 (D=%d)", D);
    return ln_mod_t<w8_t>(h, ada, ld_ada, off_shift, off_scale, out, stats, M, Tp, N, D, st, nullptr, 0, h_out);
  }
  return prec == OSUD_PREC_BF16
             ? ln_mod_t<bf16_t>(h, ada, ld_ada, off_shift, off_scale, out, stats, M, Tp, N, D, st, br, off_gate, h_out)
             : ln_mod_t<float>(h, ada, ld_ada, off_shift, off_scale, out, stats, M, Tp, N, D, st, br, off_gate, h_out);
}

int launch_final(const float* h, const float* ada, int ld_ada, int off_shift, int off_scale, const float* w,
                 const float* bias, float* out, float* u_save, float* stats, int N, int T, int Tp, int D, int C,
                 hipStream_t st, int prec, const void* br, int off_gate, float* h_out) {
  OSUD_CHECK_ARG(C >= 1 && C <= 4 && Tp % 16 == 0, "final layer: out channels %d not in 1..4 / Tp %% 16", C);
  if (prec == OSUD_PREC_BF16X3 || prec == OSUD_PREC_F16) {  // (the kernel reads fp32 h; only the pending-branch operand has the tier's type, and inference has none)
    OSUD_CHECK_ARG(br == nullptr, "final layer: the split-bf16 and fp16 tiers are inference only");
    prec = OSUD_PREC_F32;
  }
  const dim3 grid(N * Tp / 16), block(256);
#define OSUD_FIN(V)                                                                                                     \
  do {                                                                                                                  \
    if (prec == OSUD_PREC_BF16)                                                                                         \
      hipLaunchKernelGGL((final_kernel<bf16_t, V>), grid, block, 0, st, h, ada, ld_ada, off_shift, off_scale, w, bias,  \
                         out, u_save, stats, N, T, Tp, C, (const bf16_t*)br, off_gate, h_out);                          \
    else                                                                                                                \
      hipLaunchKernelGGL((final_kernel<float, V>), grid, block, 0, st, h, ada, ld_ada, off_shift, off_scale, w, bias,   \
                         out, u_save, stats, N, T, Tp, C, (const float*)br, off_gate, h_out);                           \
  } while (0)
  switch (D) {
    case 128: OSUD_FIN(2); break;
    case 384: OSUD_FIN(6); break;
    case 768: OSUD_FIN(12); break;
    case 1024: OSUD_FIN(16); break;
    case 1152: OSUD_FIN(18); break;
    default: set_error("final layer: hidden size %d not built", D); return OSUD_ERR_UNSUPPORTED;
  }
#undef OSUD_FIN
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

int launch_cfg_combine(float* out, int N, int C, int C2, int T, float s, hipStream_t st) {
  OSUD_CHECK_ARG(N % 2 == 0, "forward_with_cfg needs an even batch (cond rows then uncond rows), got %d", N);
  const int half = N / 2, total = half * C * T;
  hipLaunchKernelGGL(cfg_combine_kernel, dim3((total + 255) / 256), dim3(256), 0, st, out, half, C, C2, T, s);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

int launch_convert(int prec, const float* src, void* dst, size_t n, hipStream_t st) {
  if (n == 0) return OSUD_OK;
  const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  if (prec == OSUD_PREC_BF16)
    hipLaunchKernelGGL((convert_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, src, (bf16_t*)dst, n);
  else if (prec == OSUD_PREC_F16)
    hipLaunchKernelGGL((convert_kernel<f16_t>), dim3(grid), dim3(256), 0, st, src, (f16_t*)dst, n);
  else
    hipLaunchKernelGGL((convert_kernel<float>), dim3(grid), dim3(256), 0, st, src, (float*)dst, n);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

int launch_pack_rows(int prec, const float* src, int ld_src, int cols_src, void* dst, int ld_dst, int cols_dst, int rows,
                     hipStream_t st) {
  const size_t total = (size_t)rows * cols_dst;
  if (total == 0) return OSUD_OK;
  const int grid = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
  if (prec == OSUD_PREC_BF16)
    hipLaunchKernelGGL((pack_rows_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, src, ld_src, cols_src, (bf16_t*)dst,
                       ld_dst, cols_dst, rows);
  else if (prec == OSUD_PREC_F16)
    hipLaunchKernelGGL((pack_rows_kernel<f16_t>), dim3(grid), dim3(256), 0, st, src, ld_src, cols_src, (f16_t*)dst,
                       ld_dst, cols_dst, rows);
  else
    hipLaunchKernelGGL((pack_rows_kernel<float>), dim3(grid), dim3(256), 0, st, src, ld_src, cols_src, (float*)dst,
                       ld_dst, cols_dst, rows);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

}  // namespace osud
