// Backward-pass kernels of the DiT path that are not GEMMs (autograd of models.py:151-196,
// 306-325 written by hand).  Gradients flow as fp32 through the residual stream / LayerNorm /
// modulation and as TE (bf16 | f32) into the MFMA products.  Training layouts have no padding
// rows (T % 64 == 0 and N*T % 128 == 0 are enforced by the caller).
#include "kernels.h"

namespace osud {

namespace {

// ------------------------------------------------------------------------------------------
// out[c][r] = in[r][c] for a [R][C] TE matrix (R, C multiples of 64), plus optional column sums (bias gradients: one pass
// over dC feeds both the transposed operand of the weight-gradient GEMM and db): row block y writes its 64-row partial sums to
// colpart[y][C]; the launcher's fixed-order column pass over the R / 64 partial rows makes the sum -- no float atomics, so a
// gradient never depends on the order in which workgroups arrive.  64x64 tile through LDS, padded rows.
template <typename TE>
__global__ __launch_bounds__(256) void transpose_kernel(const TE* __restrict__ in, int ld_in, TE* __restrict__ out,
                                                        int ld_out, float* __restrict__ colsum) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 4 row-groups
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = ty + 4 * i;
    const float v = load_elem(in + (size_t)(r0 + r) * ld_in + c0 + tx);
    tile[r][tx] = v;
    s += v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = ty + 4 * i;
    store_elem(out + (size_t)(c0 + c) * ld_out + r0 + tx, tile[tx][c]);
  }
  if (colsum != nullptr) {
    __shared__ float part[4][64];
    part[ty][tx] = s;
    __syncthreads();
    if (ty == 0) colsum[(size_t)blockIdx.y * (gridDim.x * 64) + c0 + tx] = part[0][tx] + part[1][tx] + part[2][tx] + part[3][tx];
  }
}

// f32 [R][C] -> TE [C][R] (+ column sums): same, source fp32 (gradient of the residual stream)
template <typename TE>
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ in, int ld_in,
                                                            TE* __restrict__ out, int ld_out,
                                                            float* __restrict__ colsum) {
  __shared__ float tile[64][65];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = ty + 4 * i;
    const float v = in[(size_t)(r0 + r) * ld_in + c0 + tx];
    tile[r][tx] = v;
    s += v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = ty + 4 * i;
    store_elem(out + (size_t)(c0 + c) * ld_out + r0 + tx, tile[tx][c]);
  }
  if (colsum != nullptr) {
    __shared__ float part[4][64];
    part[ty][tx] = s;
    __syncthreads();
    if (ty == 0) colsum[(size_t)blockIdx.y * (gridDim.x * 64) + c0 + tx] = part[0][tx] + part[1][tx] + part[2][tx] + part[3][tx];
  }
}

// ------------------------------------------------------------------------------------------
// Gated residual branch backward:  h_out = h_in + gate[n] * br   (models.py:161-175)
//   dbr[m][d]  = gate[n][d] * dh[m][d]                       (TE, operand of the next GEMMs)
//   dgate[n][d] = sum_t dh[m][d] * br[m][d]
//   db[d]      = sum_n gate[n][d] * sum_t dh[m][d]           (= column sums of dbr: the branch Linear's bias gradient)
// Block = 64 consecutive rows (one sample: Tp % 64 == 0), wave w takes rows w, w+4, ...  The block's share of the two sums goes
// to part[block][2][D] (row 0: dgate, row 1: db); row_reduce_kernel adds the blocks' rows in a fixed order.
template <typename TE, int VPL>
__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* __restrict__ dh, const TE* __restrict__ br,
                                                       const float* __restrict__ gate, int ld_ada,
                                                       TE* __restrict__ dbr, float* __restrict__ part, int Tp) {
  constexpr int D = VPL * 64;
  __shared__ float red[4][D];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int m0 = blockIdx.x * 64, n = m0 / Tp;
  const float* g = gate + (size_t)n * ld_ada;
  float gv[VPL], acc[VPL], cs[VPL];
#pragma unroll
  for (int i = 0; i < VPL / 2; ++i) {
    const float2 t = *reinterpret_cast<const float2*>(g + 2 * lane + 128 * i);
    gv[2 * i] = t.x; gv[2 * i + 1] = t.y;
    acc[2 * i] = acc[2 * i + 1] = cs[2 * i] = cs[2 * i + 1] = 0.f;
  }
  for (int r = wave; r < 64; r += 4) {
    const size_t row = (size_t)(m0 + r) * D;
#pragma unroll
    for (int i = 0; i < VPL / 2; ++i) {
      const int d = 2 * lane + 128 * i;
      const float2 dv = *reinterpret_cast<const float2*>(dh + row + d);
      float b0, b1;
      load2(br + row + d, b0, b1);
      acc[2 * i] += dv.x * b0;
      acc[2 * i + 1] += dv.y * b1;
      cs[2 * i] += dv.x;
      cs[2 * i + 1] += dv.y;
      store2(dbr + row + d, gv[2 * i] * dv.x, gv[2 * i + 1] * dv.y);
    }
  }
#pragma unroll
  for (int i = 0; i < VPL / 2; ++i) {
    red[wave][2 * lane + 128 * i] = acc[2 * i];
    red[wave][2 * lane + 128 * i + 1] = acc[2 * i + 1];
  }
  __syncthreads();
  float* prow = part + (size_t)blockIdx.x * 2 * D;
  for (int d = threadIdx.x; d < D; d += 256) prow[d] = red[0][d] + red[1][d] + red[2][d] + red[3][d];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < VPL / 2; ++i) {
    red[wave][2 * lane + 128 * i] = gv[2 * i] * cs[2 * i];
    red[wave][2 * lane + 128 * i + 1] = gv[2 * i + 1] * cs[2 * i + 1];
  }
  __syncthreads();
  for (int d = threadIdx.x; d < D; d += 256) prow[D + d] = red[0][d] + red[1][d] + red[2][d] + red[3][d];
}

// ------------------------------------------------------------------------------------------
// LayerNorm + modulate backward:  u = xhat * (1 + sc[n]) + sh[n],  xhat = (h - mu) * rstd
//   dsh[n] = sum_t du ; dsc[n] = sum_t du * xhat ; dy = du * (1 + sc)
//   dh_out = dh_skip + rstd * (dy - mean(dy) - xhat * mean(dy * xhat))
// dh_out is the gradient of the residual stream in front of this LayerNorm, i.e. exactly the input of the gate_bwd of the
// branch that was added just before it; with br_next != nullptr that step is done here on the rows still in registers
// (dbr, dgate, db as in gate_bwd_kernel), saving a second pass over dh.
// The block's share of the per-sample / per-column sums goes to part[block][Q][D] (Q = 2: dshift, dscale; with the gate step Q = 4:
// + dgate of the next branch, + its bias gradient); row_reduce_kernel adds the blocks' rows in a fixed order (no float atomics:
// two runs of any schedule give the same bits).
#ifndef OSUD_LNB_ROWS
#define OSUD_LNB_ROWS 64
#endif
#ifdef OSUD_LNB_OCCN
#define OSUD_LNB_OCC(VPL) OSUD_LNB_OCCN
#else
#define OSUD_LNB_OCC(VPL) ((VPL) <= 12 ? 3 : 2)  // waves per SIMD the register allocation is held to (wider rows: 2, or it spills)
#endif
// W consecutive TE elements as they come from memory: bf16 stays packed (half the registers) until it is used
template <typename TE, int W> struct RawW;
template <int W> struct RawW<float, W> {
  float v[W];
  __device__ __forceinline__ void load(const float* p) { loadw<W>(p, v); }
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int e = 0; e < W; ++e) v[e] = 0.f;
  }
  __device__ __forceinline__ void get(float* o) const {
#pragma unroll
    for (int e = 0; e < W; ++e) o[e] = v[e];
  }
};
template <int W> struct RawW<bf16_t, W> {
  uint32_t u[W / 2];
  __device__ __forceinline__ void load(const bf16_t* p) {
    if constexpr (W == 4) {
      const uint2 t = *reinterpret_cast<const uint2*>(p);
      u[0] = t.x; u[1] = t.y;
    } else {
      u[0] = *reinterpret_cast<const uint32_t*>(p);
    }
  }
  __device__ __forceinline__ void zero() {
#pragma unroll
    for (int e = 0; e < W / 2; ++e) u[e] = 0u;
  }
  __device__ __forceinline__ void get(float* o) const {
#pragma unroll
    for (int e = 0; e < W / 2; ++e) {
      o[2 * e] = __uint_as_float(u[e] << 16);
      o[2 * e + 1] = __uint_as_float(u[e] & 0xffff0000u);
    }
  }
};

// Block = ROWS consecutive rows of one sample, wave w takes rows w, w+4, ...; every load of a row is issued before the first
// reduction (one memory round trip per row).  The kernel is a six-stream copy with a little arithmetic, so what matters is
// how many rows the chip has in flight, i.e. registers per wave: the per-column constants live in LDS, bf16 rows stay
// packed until they are used, the gate step is a template switch (no branches in the row loop) -> 168 registers and all 512
// blocks resident at D = 768.  (The first version held two fp32 row sets and the constants in 274 registers: one wave per
// SIMD, i.e. half of the blocks waiting for a slot -- 107 us per launch inside a training step at D = 768 and 248 us at
// D = 1152; now 86 and 141 us, a plain device copy of the same bytes takes 84 / 127 us.  Measured and not kept: 16- or 32-row
// blocks for more waves (the column-sum atomics grow with the block count: 103 us at 16 rows; at 32 rows 83 us but four
// blocks per sample add in arrival order, and gradients must not depend on that), a second register set per wave
// (equal at D = 768, spills at D = 1152), four waves per SIMD by a register cap (22 spills in the row loop, 128 us).)
// (Tried and dropped: column sums in LDS via ds_add_f32 with 16-wave blocks -- 2.4x slower, the LDS atomics serialise.)
template <typename TE, int VPL, bool GATE>
__global__ __launch_bounds__(256, OSUD_LNB_OCC(VPL)) void ln_mod_bwd_kernel(const float* __restrict__ h, const float* __restrict__ stats,
                                                         const TE* __restrict__ du, const float* __restrict__ ada,
                                                         int ld_ada, int off_shift, int off_scale,
                                                         const float* __restrict__ dh_skip, float* __restrict__ dh_out,
                                                         float* __restrict__ part, int Tp, const TE* __restrict__ br_next,
                                                         int off_gate_next, TE* __restrict__ dbr,
                                                         fp8_t* __restrict__ dbr8, const float* __restrict__ slot8,
                                                         float* __restrict__ amax_part) {
  // lane l owns columns W*l + 64*W*g + {0..W-1}, g < NG: 16-byte fp32 / 8-byte bf16 accesses where VPL allows (W = 4)
  constexpr int D = VPL * 64, W = (VPL % 4 == 0) ? 4 : 2, NG = VPL / W;
  constexpr int ROWS = OSUD_LNB_ROWS;
  // fp8 training (slot8 != nullptr, bf16 tier): dbr's e4m3 twin -- the operand of the next block's fc2 data-gradient GEMM -- and
  // this step's amax as one partial maximum per workgroup, as in ln_mod_kernel's TWIN form
  const bool twin = GATE && sizeof(TE) == 2 && slot8 != nullptr;
  const float q_scale = twin ? slot8[0] : 1.0f;
  float amax8 = 0.f;
  __shared__ float red[2][4][D];
  __shared__ float cst[2][D];  // 1 + scale | gate of the next branch
  // (the wave index as a scalar: row bases then live in SGPRs and every access is base + one per-lane offset + immediate)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m0 = blockIdx.x * ROWS, n = m0 / Tp;
  const float* arow = ada + (size_t)n * ld_ada;
  for (int d = threadIdx.x; d < D; d += 256) {
    cst[0][d] = 1.0f + arow[off_scale + d];
    cst[1][d] = GATE ? arow[off_gate_next + d] : 0.f;
  }
  float a_sh[VPL], a_sc[VPL], a_g[VPL], a_cs[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) a_sh[i] = a_sc[i] = a_g[i] = a_cs[i] = 0.f;
  __syncthreads();
  struct Row {
    float hv[VPL], sk[VPL];
    RawW<TE, W> dv[NG], bn[GATE ? NG : 1];
    float mu, rstd;
  };
  auto load_row = [&](Row& R, int m) {
    const size_t row = (size_t)m * D;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int d = W * lane + 64 * W * g;
      // h (the forward's saved residual stream) and br_next (the saved branch output) are read here for the last time: streaming
      // loads, so that they do not displace what the next launches read (training step -0.1 ... -0.2 ms in three same-box
      // alternations; the same hint on dh_skip as well: no further gain -- profiles/r05_ab_runs.md)
      if constexpr (W == 4) {
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(h + row + d));
        R.hv[g * W + 0] = t[0]; R.hv[g * W + 1] = t[1]; R.hv[g * W + 2] = t[2]; R.hv[g * W + 3] = t[3];
      } else loadw<W>(h + row + d, R.hv + g * W);
      R.dv[g].load(du + row + d);
      loadw<W>(dh_skip + row + d, R.sk + g * W);
      if constexpr (GATE) {
        if constexpr (W == 4 && sizeof(TE) == 2) {
          typedef uint32_t u2v __attribute__((ext_vector_type(2)));
          const u2v t = __builtin_nontemporal_load(reinterpret_cast<const u2v*>(br_next + row + d));
          R.bn[g].u[0] = t[0]; R.bn[g].u[1] = t[1];
        } else R.bn[g].load(br_next + row + d);
      }
    }
    R.mu = stats[2 * (size_t)m];
    R.rstd = stats[2 * (size_t)m + 1];
  };
  auto process_row = [&](Row& R, int m) {
    const size_t row = (size_t)m * D;
    const float mu = R.mu, rstd = R.rstd;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int d = W * lane + 64 * W * g;
      float dvf[W], c[W];
      R.dv[g].get(dvf);
      loadw<W>(&cst[0][d], c);
#pragma unroll
      for (int e = 0; e < W; ++e) {
        const int i = g * W + e;
        const float xh = (R.hv[i] - mu) * rstd;
        R.hv[i] = xh;
        a_sh[i] += dvf[e];
        a_sc[i] += dvf[e] * xh;
        const float dy = dvf[e] * c[e];
        s1 += dy;
        s2 += dy * xh;
      }
    }
    const float m1 = wave_sum(s1) * (1.0f / D), m2 = wave_sum(s2) * (1.0f / D);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int d = W * lane + 64 * W * g;
      float o[W], b[W], dvf[W], c[W], bnf[W], gt[W];
      R.dv[g].get(dvf);  // dy again from the packed row: cheaper than VPL live registers across the reductions
      loadw<W>(&cst[0][d], c);
      if constexpr (GATE) {
        R.bn[g].get(bnf);
        loadw<W>(&cst[1][d], gt);
      }
#pragma unroll
      for (int e = 0; e < W; ++e) {
        const int i = g * W + e;
        o[e] = R.sk[i] + rstd * (dvf[e] * c[e] - m1 - R.hv[i] * m2);
        if constexpr (GATE) {
          a_g[i] += o[e] * bnf[e];
          a_cs[i] += o[e];
          b[e] = gt[e] * o[e];
        }
      }
      if constexpr (W == 4) {  // (streaming store, as ln_mod_kernel's residual stream: its next reader is the LayerNorm backward after next)
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v t = {o[0], o[1], o[2], o[3]};
        __builtin_nontemporal_store(t, reinterpret_cast<f4v*>(dh_out + row + d));
      } else {
        storew<W>(dh_out + row + d, o);
      }
      if constexpr (GATE) {
        if (dbr != nullptr) storew<W>(dbr + row + d, b);  // (null: only the e4m3 twin is consumed this step)
        if (twin) {
          float q[W];
#pragma unroll
          for (int e = 0; e < W; ++e) {
            const float rb = bf2f(f2bf(b[e]));  // the stored bf16 value
            amax8 = fmaxf(amax8, fabsf(rb));
            q[e] = rb * q_scale;
          }
          if (dbr8 != nullptr) storew<W>(dbr8 + row + d, q);
        }
      }
    }
  };
  // rows wave, wave+4, ...: one row per wave in flight, three waves per SIMD (a second register set per wave costs a wave)
  Row R;
#pragma unroll 1
  for (int i = 0; i < ROWS / 4; ++i) {
    load_row(R, m0 + wave + 4 * i);
    process_row(R, m0 + wave + 4 * i);
  }
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int e = 0; e < W; ++e) {
      const int d = W * lane + 64 * W * g + e;
      red[0][wave][d] = a_sh[g * W + e];
      red[1][wave][d] = a_sc[g * W + e];
    }
  __syncthreads();
  float* prow = part + (size_t)blockIdx.x * (GATE ? 4 : 2) * D;
  for (int d = threadIdx.x; d < D; d += 256) {
    prow[d] = red[0][0][d] + red[0][1][d] + red[0][2][d] + red[0][3][d];
    prow[D + d] = red[1][0][d] + red[1][1][d] + red[1][2][d] + red[1][3][d];
  }
  if constexpr (GATE) {
    __syncthreads();
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int e = 0; e < W; ++e) {
        const int d = W * lane + 64 * W * g + e;
        red[0][wave][d] = a_g[g * W + e];
        red[1][wave][d] = cst[1][d] * a_cs[g * W + e];
      }
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += 256) {
      prow[2 * D + d] = red[0][0][d] + red[0][1][d] + red[0][2][d] + red[0][3][d];
      prow[3 * D + d] = red[1][0][d] + red[1][1][d] + red[1][2][d] + red[1][3][d];
    }
    if (twin) {
      __syncthreads();
      amax8 = wave_max(amax8);
      if (lane == 0) red[0][wave][0] = amax8;
      __syncthreads();
      if (threadIdx.x == 0) amax_part[blockIdx.x] = fmaxf(fmaxf(red[0][0][0], red[0][1][0]), fmaxf(red[0][2][0], red[0][3][0]));
    }
  }
}

// ------------------------------------------------------------------------------------------
// Final layer backward (models.py:192-196): out[n][ch][t] = sum_d uF[m][d] Wf[ch][d] + bf[ch]
//   duF = sum_ch dout * Wf ; dWf[ch][d] += sum_m dout * uF ; dbf[ch] += sum_m dout
// then the same LN+modulate backward as above (dh_skip = 0), all in one pass over h.
// Per-column constants (1 + scale, shift, the C weight rows) live in LDS, the next row's h is in flight while this one is
// reduced, and every partial sum of a block goes to its row of `part` [blocks][6 D + 64] = [dW rows 4 D | bias 4 (+ 60 pad) |
// dshift D | dscale D]: a fixed-order column pass makes dW and the bias gradient, row_reduce_kernel the per-sample modulation
// gradients (512 blocks adding atomically into the same addresses made the result depend on arrival order, and slow).
template <int VPL>
__global__ __launch_bounds__(256, 2) void final_bwd_kernel(const float* __restrict__ h, const float* __restrict__ stats,
                                                           const float* __restrict__ dout, const float* __restrict__ w,
                                                           const float* __restrict__ ada, int ld_ada, int off_shift,
                                                           int off_scale, float* __restrict__ dh_out, int T, int Tp, int C,
                                                           float* __restrict__ part) {
  constexpr int D = VPL * 64;
  __shared__ float red[4][D];  // one quantity at a time (6 of them)
  __shared__ float cst[6][D];  // 1 + scale | shift | weight rows (zero beyond C)
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int m0 = blockIdx.x * 64, n = m0 / Tp;
  const float* sc = ada + (size_t)n * ld_ada + off_scale;
  const float* sh = ada + (size_t)n * ld_ada + off_shift;
  for (int d = threadIdx.x; d < D; d += 256) {
    cst[0][d] = 1.0f + sc[d];
    cst[1][d] = sh[d];
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) cst[2 + ch][d] = ch < C ? w[(size_t)ch * D + d] : 0.f;
  }
  float a_sh[VPL], a_sc[VPL], a_w[4][VPL];
  float a_b[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    a_sh[i] = a_sc[i] = 0.f;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) a_w[ch][i] = 0.f;
  }
  __syncthreads();
  struct Row {
    float hv[VPL], go[4], mu, rstd;
  };
  auto load_row = [&](Row& R, int m) {
    const int t = m % Tp;
    const size_t row = (size_t)m * D;
#pragma unroll
    for (int i = 0; i < VPL / 2; ++i) {
      const float2 v = *reinterpret_cast<const float2*>(h + row + 2 * lane + 128 * i);
      R.hv[2 * i] = v.x; R.hv[2 * i + 1] = v.y;
    }
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) R.go[ch] = (t < T && ch < C) ? dout[((size_t)n * C + ch) * T + t] : 0.f;
    R.mu = stats[2 * (size_t)m];
    R.rstd = stats[2 * (size_t)m + 1];
  };
  auto process_row = [&](Row& R, int m) {
    const size_t row = (size_t)m * D;
    const float mu = R.mu, rstd = R.rstd;
    float dy[VPL];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) a_b[ch] += R.go[ch];
#pragma unroll
    for (int i = 0; i < VPL / 2; ++i) {
      const int d = 2 * lane + 128 * i;
      float c0[2], c1[2], wr[4][2];
      loadw<2>(&cst[0][d], c0);
      loadw<2>(&cst[1][d], c1);
#pragma unroll
      for (int ch = 0; ch < 4; ++ch) loadw<2>(&cst[2 + ch][d], wr[ch]);
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int k = 2 * i + e;
        const float xh = (R.hv[k] - mu) * rstd;
        R.hv[k] = xh;
        const float uF = xh * c0[e] + c1[e];
        float du = 0.f;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
          du += R.go[ch] * wr[ch][e];
          a_w[ch][k] += R.go[ch] * uF;
        }
        a_sh[k] += du;
        a_sc[k] += du * xh;
        dy[k] = du * c0[e];
        s1 += dy[k];
        s2 += dy[k] * xh;
      }
    }
    const float m1 = wave_sum(s1) * (1.0f / D), m2 = wave_sum(s2) * (1.0f / D);
#pragma unroll
    for (int i = 0; i < VPL / 2; ++i) {
      const int d = 2 * lane + 128 * i;
      *reinterpret_cast<float2*>(dh_out + row + d) =
          make_float2(rstd * (dy[2 * i] - m1 - R.hv[2 * i] * m2), rstd * (dy[2 * i + 1] - m1 - R.hv[2 * i + 1] * m2));
    }
  };
  Row ra, rb;
  load_row(ra, m0 + wave);
#pragma unroll 1
  for (int r = wave; r < 64; r += 8) {
    load_row(rb, m0 + r + 4);
    process_row(ra, m0 + r);
    if (r + 8 < 64) load_row(ra, m0 + r + 8);
    process_row(rb, m0 + r + 4);
  }
  float* prow = part + (size_t)blockIdx.x * (6 * D + 64);
#pragma unroll
  for (int q = 0; q < 6; ++q) {  // 0: dshift, 1: dscale, 2..5: dW rows (zero beyond C: their dout is)
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int d = 2 * lane + 128 * (i >> 1) + (i & 1);
      red[wave][d] = q == 0 ? a_sh[i] : (q == 1 ? a_sc[i] : a_w[q >= 2 ? q - 2 : 0][i]);
    }
    __syncthreads();
    float* dst = q < 2 ? prow + 4 * D + 64 + (size_t)q * D : prow + (size_t)(q - 2) * D;
    for (int d = threadIdx.x; d < D; d += 256) dst[d] = red[0][d] + red[1][d] + red[2][d] + red[3][d];
    __syncthreads();
  }
  // bias partial sums behind the weight rows: [4 D + 0..3], the pad columns are never read back
  if (lane == 0) {
#pragma unroll
    for (int ch = 0; ch < 4; ++ch) red[wave][ch] = a_b[ch];  // every lane of a wave carries the same a_b
  }
  __syncthreads();
  if (threadIdx.x < 4) prow[4 * D + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// ------------------------------------------------------------------------------------------
// Conditioning path backward.  sb = silu(b), b = tvec + table[y]  (models.py:318-320):
//   db = dsb * silu'(b) ; dtvec = db (TE copy for the GEMMs + f32) ; dtable[y[n]] = sum of db[n'] over the samples n' with that label
// Two kernels.  cond_bwd_kernel makes the rows db[n].  table_rows_kernel adds the rows that share a label in a FIXED order (no float
// atomics: with label dropout a fifth of a training batch carries the null class, and 51 atomic adds per element arrive in any
// order): block n finds out in parallel whether it is the first sample with its label (the row's owner) and which later samples
// share it (a bit mask in LDS, turned into an ascending index list); the owner's threads are (column, part) pairs, part p adds rows
// list[p], list[p + P], ... and the P partial rows meet in LDS and are added in part order.  The table is zeroed by the caller and
// every touched row has exactly one writer.  (One block walking its 51 rows as a dependent chain per column: 55-100 us; this: ~8.)
template <typename TE>
__global__ void cond_bwd_kernel(const float* __restrict__ dsb, const float* __restrict__ b, float* __restrict__ db_out,
                                TE* __restrict__ db_te, int N, int D) {
  const int n = blockIdx.x;
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    float g = 0.f;
    if (n < N) {
      const float bv = b[(size_t)n * D + d];
      const float s = 1.0f / (1.0f + expf(-bv));
      g = dsb[(size_t)n * D + d] * (s + bv * s * (1.0f - s));
    }
    db_out[(size_t)n * D + d] = g;
    store_elem(db_te + (size_t)n * D + d, g);
  }
}

constexpr int kTableThreads = 1024;
__global__ __launch_bounds__(kTableThreads) void table_rows_kernel(const float* __restrict__ db, const int64_t* __restrict__ y, int table_rows,
                                                                   float* __restrict__ dtable, int N, int D) {
  constexpr int kMaskWords = 128, kList = 1024;  // batches of up to 4096 samples / 1024 sharers of one label; beyond: the plain walk
  __shared__ unsigned later[kMaskWords];
  __shared__ int list[kList];
  __shared__ int list_n;
  extern __shared__ float parts[];  // [P][D]
  const int n = blockIdx.x;
  auto label = [&](int i) {
    const int64_t cls = y[i];
    return cls < 0 ? (int64_t)0 : (cls >= table_rows ? (int64_t)table_rows - 1 : cls);
  };
  const int64_t cls = label(n);
  const bool masked = N <= kMaskWords * 32;
  for (int w = threadIdx.x; w < kMaskWords; w += blockDim.x) later[w] = 0u;
  __syncthreads();
  int before = 0;
  for (int i = threadIdx.x; i < N; i += blockDim.x)
    if (i != n && label(i) == cls) {
      if (i < n) before = 1;
      else if (masked) atomicOr(&later[i >> 5], 1u << (i & 31));
    }
  if (__syncthreads_or(before)) return;  // an earlier sample owns this label's row
  if (threadIdx.x == 0) {
    int k = 0;
    list[k++] = n;
    if (masked)
      for (int w = n >> 5; w < (N + 31) / 32; ++w) {
        unsigned bits = later[w];
        while (bits) {
          if (k < kList) list[k] = 32 * w + __builtin_ctz(bits);
          ++k;
          bits &= bits - 1;
        }
      }
    list_n = k;
  }
  __syncthreads();
  const int cols4 = D / 4, P = kTableThreads / cols4 > 8 ? 8 : kTableThreads / cols4;
  const int c4 = threadIdx.x % cols4, part = threadIdx.x / cols4;
  float* row = dtable + (size_t)cls * D;
  if (!masked || list_n > kList) {  // (not reached by any realistic batch: the plain walk in sample order)
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
      float s = db[(size_t)n * D + d];
      for (int i = n + 1; i < N; ++i)
        if (label(i) == cls) s += db[(size_t)i * D + d];
      row[d] = s;
    }
    return;
  }
  const int nl = list_n;
  if (part < P) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int k = part;
    for (; k + 3 * P < nl; k += 4 * P) {  // four rows' loads in flight
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(db + (size_t)list[k + u * P] * D + 4 * c4);
#pragma unroll
      for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
    }
    for (; k < nl; k += P) {
      const float4 v = *reinterpret_cast<const float4*>(db + (size_t)list[k] * D + 4 * c4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(parts + (size_t)part * D + 4 * c4) = s;
  }
  __syncthreads();
  if (part == 0) {
    float4 s = *reinterpret_cast<const float4*>(parts + 4 * c4);
    for (int p = 1; p < P; ++p) {
      const float4 v = *reinterpret_cast<const float4*>(parts + (size_t)p * D + 4 * c4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    *reinterpret_cast<float4*>(row + 4 * c4) = s;
  }
}

// dz = dth * silu'(z)  (TimestepEmbedder's SiLU, models.py:29-31); z is the saved pre-activation
template <typename TE>
__global__ void silu_bwd_kernel(const float* __restrict__ dth, const TE* __restrict__ z, TE* __restrict__ dz, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float zv = load_elem(z + i);
    const float s = 1.0f / (1.0f + expf(-zv));
    store_elem(dz + i, dth[i] * (s + zv * s * (1.0f - s)));
  }
}

// zero rows >= N of an f32 [Np][C] matrix and emit the TE copy (padding samples carry no gradient)
template <typename TE>
__global__ void mask_rows_kernel(float* __restrict__ a, TE* __restrict__ a_te, int N, int Np, int C, int ld) {
  const size_t total = (size_t)Np * C;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int n = (int)(i / C);
    const size_t j = (size_t)n * ld + (i % C);  // a column slice of a wider matrix when ld > C
    float v = a[j];
    if (n >= N) {
      v = 0.f;
      a[j] = 0.f;
    }
    store_elem(a_te + j, v);
  }
}

// dst[r][0..cols) = src[r][0..cols) from a padded f32 [rows][ld_src] (first-layer weight gradient)
__global__ void unpad_rows_kernel(const float* __restrict__ src, int ld_src, float* __restrict__ dst, int cols, int rows) {
  const size_t total = (size_t)rows * cols;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(i / cols), c = (int)(i % cols);
    dst[i] = src[(size_t)r * ld_src + c];
  }
}

// column sums of an f32 [R][C] matrix over rows < R_valid (bias grads of the conditioning path)
__global__ void colsum_f32_kernel(const float* __restrict__ a, int R_valid, int C, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int r = 0; r < R_valid; ++r) s += a[(size_t)r * C + c];
  out[c] = s;
}

// ------------------------------------------------------------------------------------------
// The fixed-order sums behind gate_bwd / ln_mod_bwd / final_bwd: item i holds per-block partial rows part[block][..] (row q of
// block b at part + b * stride + q * D), `bps` consecutive blocks per sample.  Rows q < nq_sample are per-sample sums:
//   dada[n][off[q] + d] = sum_{j < bps} part[n * bps + j][q][d]                      (written, not added: every slot has one writer)
// and row nq_sample (bias != nullptr) is summed over ALL blocks: bias[d] = sum_b part[b][nq_sample][d].
// Grid (D / 64, items); 16 sample groups x 16 float4 columns per block: group g walks samples g, g + 16, ... and their blocks in
// order, the 16 groups' bias shares meet in LDS and are added in group order.  One launch per backward call for all the
// LayerNorm / gate kernels of the call (the descriptor list travels in the kernel arguments).
// BPS / NQ: compile-time blocks per sample and rows per block (0 = run-time loops): with both known, the 2 x BPS x NQ loads of two
// samples are issued before the first sum (the kernel is a latency-bound gather of 6 MB: 80 us with one sample's loads in flight)
template <int BPS, int NQ>
__global__ __launch_bounds__(256) void row_reduce_kernel(const RowRedList L) {
  __shared__ float4 sh[16][16];
  const int it = blockIdx.y;
  const float* part = L.part[it];
  const int stride = L.stride[it], bps = BPS ? BPS : L.bps[it], nqs = L.nq_sample[it], samples = L.blocks[it] / bps, D = L.D;
  const int tx = threadIdx.x & 15, tg = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + tx * 4;
  float* bias = L.bias[it];
  const int nq = nqs + (bias != nullptr ? 1 : 0);
  float4 bsum = make_float4(0.f, 0.f, 0.f, 0.f);
  auto finish = [&](int n, int q, const float4& s) {
    if (q < nqs) *reinterpret_cast<float4*>(L.dada[it] + (size_t)n * L.ld_ada + L.off[it][q] + c) = s;
    else { bsum.x += s.x; bsum.y += s.y; bsum.z += s.z; bsum.w += s.w; }
  };
  if constexpr (BPS > 0 && NQ > 0) {
    constexpr int U = 2;  // samples per round
    for (int n0 = tg; n0 < samples; n0 += 16 * U) {
      float4 v[U][BPS][NQ];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int n = n0 + 16 * u;
#pragma unroll
        for (int j = 0; j < BPS; ++j)
#pragma unroll
          for (int q = 0; q < NQ; ++q)
            v[u][j][q] = (n < samples && q < nq) ? *reinterpret_cast<const float4*>(part + ((size_t)n * BPS + j) * stride + (size_t)q * D + c)
                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int n = n0 + 16 * u;
        if (n >= samples) break;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          if (q >= nq) break;
          float4 s = v[u][0][q];
#pragma unroll
          for (int j = 1; j < BPS; ++j) { s.x += v[u][j][q].x; s.y += v[u][j][q].y; s.z += v[u][j][q].z; s.w += v[u][j][q].w; }
          finish(n, q, s);
        }
      }
    }
  } else {
    for (int n = tg; n < samples; n += 16) {
      const float* p0 = part + (size_t)n * bps * stride + c;
      for (int q = 0; q < nq; ++q) {
        float4 s = *reinterpret_cast<const float4*>(p0 + (size_t)q * D);
        for (int j = 1; j < bps; ++j) {
          const float4 v = *reinterpret_cast<const float4*>(p0 + (size_t)j * stride + (size_t)q * D);
          s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        finish(n, q, s);
      }
    }
  }
  if (bias != nullptr) {
    sh[tg][tx] = bsum;
    __syncthreads();
    if (tg == 0) {
      float4 t = sh[0][tx];
#pragma unroll
      for (int g = 1; g < 16; ++g) { t.x += sh[g][tx].x; t.y += sh[g][tx].y; t.z += sh[g][tx].z; t.w += sh[g][tx].w; }
      *reinterpret_cast<float4*>(bias + c) = t;
    }
  }
}

}  // namespace

// ---------------------------------------------------------------------------------- launchers
int launch_transpose(int prec, const void* in, int ld_in, void* out, int ld_out, int R, int C, float* colsum,
                     hipStream_t st, float* colpart, size_t colpart_elems) {
  OSUD_CHECK_ARG(R % 64 == 0 && C % 64 == 0, "transpose: %dx%d must be multiples of 64", R, C);
  OSUD_CHECK_ARG(colsum == nullptr || (colpart != nullptr && (size_t)(R / 64) * C <= colpart_elems),
                 "transpose: the column sums need %d x %d floats of scratch for their partial rows", R / 64, C);
  const dim3 grid(C / 64, R / 64);
  float* cp = colsum != nullptr ? colpart : nullptr;
  if (prec == OSUD_PREC_BF16)
    hipLaunchKernelGGL((transpose_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)in, ld_in, (bf16_t*)out, ld_out, cp);
  else
    hipLaunchKernelGGL((transpose_kernel<float>), grid, dim3(256), 0, st, (const float*)in, ld_in, (float*)out, ld_out, cp);
  OSUD_HIP(hipGetLastError());
  if (colsum != nullptr) return launch_colsum_f32(colpart, R / 64, C, colsum, st);  // fixed-order sum of the row blocks' shares
  return OSUD_OK;
}

int launch_transpose_f32(int prec, const float* in, int ld_in, void* out, int ld_out, int R, int C, float* colsum,
                         hipStream_t st, float* colpart, size_t colpart_elems) {
  OSUD_CHECK_ARG(R % 64 == 0 && C % 64 == 0, "transpose: %dx%d must be multiples of 64", R, C);
  OSUD_CHECK_ARG(colsum == nullptr || (colpart != nullptr && (size_t)(R / 64) * C <= colpart_elems),
                 "transpose: the column sums need %d x %d floats of scratch for their partial rows", R / 64, C);
  const dim3 grid(C / 64, R / 64);
  float* cp = colsum != nullptr ? colpart : nullptr;
  if (prec == OSUD_PREC_BF16)
    hipLaunchKernelGGL((transpose_f32_kernel<bf16_t>), grid, dim3(256), 0, st, in, ld_in, (bf16_t*)out, ld_out, cp);
  else
    hipLaunchKernelGGL((transpose_f32_kernel<float>), grid, dim3(256), 0, st, in, ld_in, (float*)out, ld_out, cp);
  OSUD_HIP(hipGetLastError());
  if (colsum != nullptr) return launch_colsum_f32(colpart, R / 64, C, colsum, st);
  return OSUD_OK;
}

#define OSUD_BY_D(D, CALL)                                                              \
  switch (D) {                                                                          \
    case 128: CALL(2); break;                                                           \
    case 384: CALL(6); break;                                                           \
    case 768: CALL(12); break;                                                          \
    case 1024: CALL(16); break;                                                         \
    case 1152: CALL(18); break;                                                         \
    default: set_error("hidden size %d not built", D); return OSUD_ERR_UNSUPPORTED;    \
  }

int launch_gate_bwd(int prec, const float* dh, const void* br, const float* gate, int ld_ada, void* dbr, float* part,
                    int M, int Tp, int D, hipStream_t st) {
  OSUD_CHECK_ARG(M % 64 == 0 && Tp % 64 == 0, "gate_bwd: rows must come in blocks of 64");
  OSUD_CHECK_ARG(part != nullptr, "gate_bwd: no room for the partial rows");
  const dim3 grid(M / 64), block(256);
  if (prec == OSUD_PREC_BF16) {
#define CALL(V) hipLaunchKernelGGL((gate_bwd_kernel<bf16_t, V>), grid, block, 0, st, dh, (const bf16_t*)br, gate, ld_ada, (bf16_t*)dbr, part, Tp)
    OSUD_BY_D(D, CALL)
#undef CALL
  } else {
#define CALL(V) hipLaunchKernelGGL((gate_bwd_kernel<float, V>), grid, block, 0, st, dh, (const float*)br, gate, ld_ada, (float*)dbr, part, Tp)
    OSUD_BY_D(D, CALL)
#undef CALL
  }
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

int launch_ln_mod_bwd(int prec, const float* h, const float* stats, const void* du, const float* ada, int ld_ada,
                      int off_shift, int off_scale, const float* dh_skip, float* dh_out, float* part, int M, int Tp, int D,
                      hipStream_t st, const void* br_next, int off_gate_next, void* dbr, void* dbr8, const float* slot8,
                      float* amax_part) {
  OSUD_CHECK_ARG(part != nullptr, "ln_mod_bwd: no room for the partial rows");
  OSUD_CHECK_ARG(M % 64 == 0 && Tp % 64 == 0, "ln_mod_bwd: rows must come in blocks of 64");
  OSUD_CHECK_ARG(dh_skip != nullptr, "ln_mod_bwd: the gradient of the residual stream behind the LayerNorm is required");
  OSUD_CHECK_ARG(slot8 == nullptr || (amax_part != nullptr && br_next != nullptr && prec == OSUD_PREC_BF16 && M / OSUD_LNB_ROWS <= f8_amax_parts()),
                 "ln_mod_bwd: the e4m3 twin rides with the gate step of the bf16 tier");
  OSUD_CHECK_ARG(br_next == nullptr || dbr != nullptr || (dbr8 != nullptr && slot8 != nullptr),
                 "ln_mod_bwd: the gate step needs somewhere to put the branch gradient (bf16 rows, or the e4m3 twin with its scale slot)");
  const dim3 grid(M / OSUD_LNB_ROWS), block(256);
#define ARGS(T) h, stats, (const T*)du, ada, ld_ada, off_shift, off_scale, dh_skip, dh_out, part, Tp, (const T*)br_next, off_gate_next, (T*)dbr, (fp8_t*)dbr8, slot8, amax_part
  if (prec == OSUD_PREC_BF16) {
#define CALL(V)                                                                                                      \
  if (br_next != nullptr) hipLaunchKernelGGL((ln_mod_bwd_kernel<bf16_t, V, true>), grid, block, 0, st, ARGS(bf16_t)); \
  else hipLaunchKernelGGL((ln_mod_bwd_kernel<bf16_t, V, false>), grid, block, 0, st, ARGS(bf16_t));
    OSUD_BY_D(D, CALL)
#undef CALL
  } else {
#define CALL(V)                                                                                                    \
  if (br_next != nullptr) hipLaunchKernelGGL((ln_mod_bwd_kernel<float, V, true>), grid, block, 0, st, ARGS(float)); \
  else hipLaunchKernelGGL((ln_mod_bwd_kernel<float, V, false>), grid, block, 0, st, ARGS(float));
    OSUD_BY_D(D, CALL)
#undef CALL
  }
#undef ARGS
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

int launch_final_bwd(const float* h, const float* stats, const float* dout, const float* w, const float* ada, int ld_ada,
                     int off_shift, int off_scale, float* dh_out, float* dw, float* dbias, int N, int T, int Tp,
                     int D, int C, hipStream_t st, float* part) {
  OSUD_CHECK_ARG(Tp % 64 == 0 && C == 4, "final_bwd: bad sizes (Tp=%d, %d output channels; built for 4)", Tp, C);
  OSUD_CHECK_ARG(part != nullptr, "final_bwd: no room for the partial rows");
  const dim3 grid(N * Tp / 64), block(256);
  // every partial sum of a block -> its row of part [blocks][6 D + 64]; dW and the bias gradient by a fixed-order column pass over
  // the first 4 D + 64 columns here, the per-sample dshift / dscale rows by the caller's row_reduce item
#define CALL(V) hipLaunchKernelGGL((final_bwd_kernel<V>), grid, block, 0, st, h, stats, dout, w, ada, ld_ada, off_shift, off_scale, dh_out, T, Tp, C, part)
  OSUD_BY_D(D, CALL)
#undef CALL
  OSUD_HIP(hipGetLastError());
  return launch_colsum_f32(part, (int)grid.x, 4 * D + 64, dw, st, 4 * D, dbias, 4, 6 * D + 64);
}

int launch_cond_bwd(int prec, const float* dsb, const float* b, const int64_t* y, int table_rows, float* db_out,
                    void* db_te, float* dtable, int N, int Np, int D, hipStream_t st) {
  OSUD_CHECK_ARG(D % 4 == 0 && D / 4 <= kTableThreads, "cond_bwd: hidden size %d", D);
  if (prec == OSUD_PREC_BF16)
    hipLaunchKernelGGL((cond_bwd_kernel<bf16_t>), dim3(Np), dim3(256), 0, st, dsb, b, db_out, (bf16_t*)db_te, N, D);
  else
    hipLaunchKernelGGL((cond_bwd_kernel<float>), dim3(Np), dim3(256), 0, st, dsb, b, db_out, (float*)db_te, N, D);
  OSUD_HIP(hipGetLastError());
  const int P = kTableThreads / (D / 4) > 8 ? 8 : kTableThreads / (D / 4);
  hipLaunchKernelGGL(table_rows_kernel, dim3(N), dim3(kTableThreads), (size_t)P * D * sizeof(float), st, db_out, y, table_rows, dtable, N, D);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

int launch_silu_bwd(int prec, const float* dth, const void* z, void* dz, size_t n, hipStream_t st) {
  const int grid = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
  if (prec == OSUD_PREC_BF16)
    hipLaunchKernelGGL((silu_bwd_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, dth, (const bf16_t*)z, (bf16_t*)dz, n);
  else
    hipLaunchKernelGGL((silu_bwd_kernel<float>), dim3(grid), dim3(256), 0, st, dth, (const float*)z, (float*)dz, n);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

int launch_mask_rows(int prec, float* a, void* a_te, int N, int Np, int C, hipStream_t st, int ld) {
  if (ld <= 0) ld = C;
  const size_t total = (size_t)Np * C;
  const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  if (prec == OSUD_PREC_BF16)
    hipLaunchKernelGGL((mask_rows_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, a, (bf16_t*)a_te, N, Np, C, ld);
  else
    hipLaunchKernelGGL((mask_rows_kernel<float>), dim3(grid), dim3(256), 0, st, a, (float*)a_te, N, Np, C, ld);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

int launch_unpad_rows(const float* src, int ld_src, float* dst, int cols, int rows, hipStream_t st) {
  const size_t total = (size_t)rows * cols;
  const int grid = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(unpad_rows_kernel, dim3(grid), dim3(256), 0, st, src, ld_src, dst, cols, rows);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

// the same for tall partial-sum slabs (R in the hundreds): 16 row groups x 16 float4 columns per block, fixed-order combine
// (columns >= split go to out2[c - split], of which only n2 exist: the final layer's bias gradient rides behind its weight gradient)
__global__ __launch_bounds__(256) void colsum_f32_tall_kernel(const float* __restrict__ a, int R, int C /* row stride */, float* __restrict__ out,
                                                              int split, float* __restrict__ out2, int n2) {
  __shared__ float4 part[16][16];
  const int tx = threadIdx.x & 15, tg = threadIdx.x >> 4;
  const int c = blockIdx.x * 64 + tx * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r = tg; r < R; r += 64) {  // four independent loads in flight per thread
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int rr = r + 16 * u;
      v[u] = rr < R ? *reinterpret_cast<const float4*>(a + (size_t)rr * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
  }
  part[tg][tx] = s;
  __syncthreads();
  if (tg == 0) {
    float4 t = part[0][tx];
#pragma unroll
    for (int g = 1; g < 16; ++g) { t.x += part[g][tx].x; t.y += part[g][tx].y; t.z += part[g][tx].z; t.w += part[g][tx].w; }
    if (c < split) *reinterpret_cast<float4*>(out + c) = t;
    else if (c - split < n2) *reinterpret_cast<float4*>(out2 + (c - split)) = t;
  }
}

int launch_colsum_f32(const float* a, int R_valid, int C, float* out, hipStream_t st, int split, float* out2, int n2, int ld) {
  if (ld <= 0) ld = C;
  if (out2 == nullptr) split = C;
  OSUD_CHECK_ARG(out2 == nullptr || (C % 64 == 0 && split % 4 == 0 && n2 % 4 == 0), "colsum: bad split");
  OSUD_CHECK_ARG(ld == C || (C % 64 == 0 && ld % 4 == 0), "colsum: a column window needs 64-column blocks");
  if (C % 64 == 0 && ld % 4 == 0 && (R_valid >= 64 || out2 != nullptr || ld != C)) {
    hipLaunchKernelGGL(colsum_f32_tall_kernel, dim3(C / 64), dim3(256), 0, st, a, R_valid, ld, out, split, out2, n2);
    OSUD_HIP(hipGetLastError());
    return OSUD_OK;
  }
  hipLaunchKernelGGL(colsum_f32_kernel, dim3((C + 255) / 256), dim3(256), 0, st, a, R_valid, C, out);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

namespace {
__global__ __launch_bounds__(256) void colsum_many_kernel(const ColsumList L) {
  __shared__ float4 part[16][16];
  int it = 0;
  while (it + 1 < L.count && (int)blockIdx.x >= L.blk_begin[it + 1]) ++it;
  const float* a = L.src[it];
  const int R = L.R[it], ld = L.ld[it];
  const int tx = threadIdx.x & 15, tg = threadIdx.x >> 4;
  const int c = ((int)blockIdx.x - L.blk_begin[it]) * 64 + tx * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r = tg; r < R; r += 64) {  // four independent loads in flight per thread (as colsum_f32_tall_kernel: the same order)
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int rr = r + 16 * u;
      v[u] = rr < R ? *reinterpret_cast<const float4*>(a + (size_t)rr * ld + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
  }
  part[tg][tx] = s;
  __syncthreads();
  if (tg == 0) {
    float4 t = part[0][tx];
#pragma unroll
    for (int g = 1; g < 16; ++g) { t.x += part[g][tx].x; t.y += part[g][tx].y; t.z += part[g][tx].z; t.w += part[g][tx].w; }
    *reinterpret_cast<float4*>(L.out[it] + c) = t;
  }
}
}  // namespace

int launch_colsum_many(const ColsumList& L, hipStream_t st) {
  if (L.count == 0) return OSUD_OK;
  hipLaunchKernelGGL(colsum_many_kernel, dim3(L.blk_begin[L.count]), dim3(256), 0, st, L);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

int launch_row_reduce(const RowRedList& L, hipStream_t st) {
  if (L.count == 0) return OSUD_OK;
  OSUD_CHECK_ARG(L.D % 64 == 0 && L.ld_ada % 4 == 0, "row_reduce: D=%d, ld=%d", L.D, L.ld_ada);
  // (every item of a list comes from the same backward call: the same blocks per sample; rows per block: at most 4 = 3 per-sample + bias)
  const int bps = L.bps[0];
  bool uniform = true;
  for (int i = 1; i < L.count; ++i) uniform = uniform && L.bps[i] == bps;
  if (uniform && bps == 2) hipLaunchKernelGGL((row_reduce_kernel<2, 4>), dim3(L.D / 64, L.count), dim3(256), 0, st, L);
  else if (uniform && bps == 4) hipLaunchKernelGGL((row_reduce_kernel<4, 4>), dim3(L.D / 64, L.count), dim3(256), 0, st, L);
  else if (uniform && bps == 1) hipLaunchKernelGGL((row_reduce_kernel<1, 4>), dim3(L.D / 64, L.count), dim3(256), 0, st, L);
  else hipLaunchKernelGGL((row_reduce_kernel<0, 0>), dim3(L.D / 64, L.count), dim3(256), 0, st, L);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

}  // namespace osud
