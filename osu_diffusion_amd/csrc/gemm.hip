// See gemm.h for the design.  gfx950 only.
#include "gemm.h"

namespace osud {

namespace {

constexpr int BM = 128, BN = 128, SLAB = 128;  // SLAB in bytes along K
constexpr int TILE_BYTES = BM * SLAB;          // 16 KiB per operand per buffer

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

// HBM -> LDS, one 128-row x 128-byte operand slab, 4 x 1 KiB pieces per wave.
__device__ __forceinline__ void stage_tile(const char* gsrc, size_t ld_bytes, char* lds_tile, int wave, int lane) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int r0 = 32 * wave + 8 * q;        // wave-uniform
    const int R = r0 + (lane >> 3);          // this lane's row
    const int c = (lane & 7) ^ ((R >> 1) & 7);  // source chunk for LDS position lane&7
    const char* g = gsrc + (size_t)R * ld_bytes + c * 16;
    char* dst = lds_tile + __builtin_amdgcn_readfirstlane(r0 * SLAB);
    __builtin_amdgcn_global_load_lds((glb_void*)g, (lds_void*)dst, 16, 0, 0);
  }
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Fragment reads are inline asm on purpose: hipcc cannot prove that a compiler-visible
// ds_read does not alias the in-flight LDS-DMA of the NEXT slab and would put
// `s_waitcnt vmcnt(0)` in front of every read, serialising load and MFMA.  The asm reads are
// ordered by the explicit vmcnt(0)+barrier at the top of each slab and by counted lgkmcnt.
template <int OFF> __device__ __forceinline__ u32x4 ds_read16(uint32_t addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "i"(OFF));
  return v;
}

template <typename TE> __device__ __forceinline__ void mma(f32x16& acc, const u32x4& a, const u32x4& b);
template <> __device__ __forceinline__ void mma<bf16_t>(f32x16& acc, const u32x4& a, const u32x4& b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0,
                                                0, 0);
}
template <> __device__ __forceinline__ void mma<float>(f32x16& acc, const u32x4& a, const u32x4& b) {
  const f32x4 af = __builtin_bit_cast(f32x4, a), bf = __builtin_bit_cast(f32x4, b);
#pragma unroll
  for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[j], bf[j], acc, 0, 0, 0);
}

struct FragSet {
  u32x4 y[2], x[2];
};
// ya/xa: this lane's LDS byte address of (first Y / X row of the wave, k-substep s) in buffer 0
template <int BUF> __device__ __forceinline__ void read_set(FragSet& f, uint32_t ya, uint32_t xa) {
  constexpr int B = BUF * 2 * TILE_BYTES;
  f.y[0] = ds_read16<B>(ya);
  f.y[1] = ds_read16<B + 32 * SLAB>(ya);
  f.x[0] = ds_read16<B + TILE_BYTES>(xa);
  f.x[1] = ds_read16<B + TILE_BYTES + 32 * SLAB>(xa);
}
template <typename TE> __device__ __forceinline__ void mma_set(f32x16 (&acc)[2][2], const FragSet& f) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) mma<TE>(acc[i][j], f.x[j], f.y[i]);
}
#define OSUD_LGKM_WAIT(n)                                  \
  asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory"); \
  __builtin_amdgcn_sched_barrier(0)

// One 128-byte K slab from LDS buffer BUF: 4 sub-steps, reads of sub-step s+1 in flight under
// the MFMAs of sub-step s (LDS returns in order, so lgkmcnt(4) == "all but the newest 4").
template <typename TE, int BUF>
__device__ __forceinline__ void compute_slab(f32x16 (&acc)[2][2], const uint32_t (&ya)[4], const uint32_t (&xa)[4]) {
  FragSet f0, f1;
  read_set<BUF>(f0, ya[0], xa[0]);
  read_set<BUF>(f1, ya[1], xa[1]);
  OSUD_LGKM_WAIT(4);
  mma_set<TE>(acc, f0);
  read_set<BUF>(f0, ya[2], xa[2]);
  OSUD_LGKM_WAIT(4);
  mma_set<TE>(acc, f1);
  read_set<BUF>(f1, ya[3], xa[3]);
  OSUD_LGKM_WAIT(4);
  mma_set<TE>(acc, f0);
  OSUD_LGKM_WAIT(0);
  mma_set<TE>(acc, f1);
}

template <typename TE, int EPI>
__global__ __launch_bounds__(256) void gemm_kernel(GemmP p) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE_BYTES];  // [buf][Y|X]
  constexpr bool FAST = sizeof(TE) == 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wy = wave >> 1, wx = wave & 1;

  // Workgroup -> tile.  Blocks are dispatched round-robin over the 8 XCDs (b % 8); give each
  // XCD a contiguous run of tiles (x fastest) so its private L2 sees whole Y row-panels.
  const int ntx = p.Nx / BN, nwg = gridDim.x;
  int b = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7, idx = b >> 3;
    b = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int ty = b / ntx, tx = b % ntx;

  const size_t ldy_b = (size_t)p.ldy * sizeof(TE), ldx_b = (size_t)p.ldx * sizeof(TE);
  const char* gy = reinterpret_cast<const char*>(p.Y) + (size_t)ty * BM * ldy_b;
  const char* gx = reinterpret_cast<const char*>(p.X) + (size_t)tx * BN * ldx_b;
  const int nk = (int)((size_t)p.K * sizeof(TE) / SLAB);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // per-lane LDS byte addresses of the wave's first Y/X row for the 4 k-substeps (buffer 0)
  const int frow = lane & 31, fhalf = lane >> 5;
  const uint32_t lds0 = (uint32_t)(size_t)(lds_void*)smem;
  uint32_t ya[4], xa[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const uint32_t sw = (uint32_t)(((2 * s + fhalf) ^ ((frow >> 1) & 7)) << 4);
    ya[s] = lds0 + (wy * 64 + frow) * SLAB + sw;
    xa[s] = lds0 + (wx * 64 + frow) * SLAB + sw;
  }

  stage_tile(gy, ldy_b, smem, wave, lane);
  stage_tile(gx, ldx_b, smem + TILE_BYTES, wave, lane);
  int kt = 0;
  for (; kt + 2 <= nk; kt += 2) {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // slab kt landed for every wave; buffer 1 is free again
    stage_tile(gy + (size_t)(kt + 1) * SLAB, ldy_b, smem + 2 * TILE_BYTES, wave, lane);
    stage_tile(gx + (size_t)(kt + 1) * SLAB, ldx_b, smem + 3 * TILE_BYTES, wave, lane);
    compute_slab<TE, 0>(acc, ya, xa);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (kt + 2 < nk) {
      stage_tile(gy + (size_t)(kt + 2) * SLAB, ldy_b, smem, wave, lane);
      stage_tile(gx + (size_t)(kt + 2) * SLAB, ldx_b, smem + TILE_BYTES, wave, lane);
    }
    compute_slab<TE, 1>(acc, ya, xa);
  }
  if (kt < nk) {  // odd slab count: the last slab sits in buffer 0
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    compute_slab<TE, 0>(acc, ya, xa);
  }

  // ---- epilogue: lane holds, for y = ..+frow, x = xb + 8g + 4*fhalf + {0..3}, g = 0..3 -----
  // All loads of a batch (bias / gate / residual / aux) are issued BEFORE its stores: on gfx950
  // vmcnt counts stores too, so a load waited for between stores would drain every earlier store.
  constexpr bool kBias = EPI == EPI_BIAS_F32 || EPI == EPI_BIAS_TE || EPI == EPI_BIAS_SILU_TE ||
                         EPI == EPI_BIAS_GELU_TE || EPI == EPI_GATE_RES;
  float4 bv[2][4];
  if (kBias) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        bv[j][g] = *reinterpret_cast<const float4*>(p.bias + tx * BN + wx * 64 + j * 32 + 8 * g + 4 * fhalf);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int y = ty * BM + wy * 64 + i * 32 + frow;
    float rb = 0.f;
    if (EPI == EPI_ROWBIAS_TE) rb = p.bias[y];
    float4 gv[2][4], rv[2][4];
    if (EPI == EPI_GATE_RES) {
      int sample = y / p.rows_per_sample;
      if (sample >= p.n_samples) sample = p.n_samples - 1;  // padding rows
      const float* rsrc = p.res ? p.res : reinterpret_cast<const float*>(p.out);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int x = tx * BN + wx * 64 + j * 32 + 8 * g + 4 * fhalf;
          gv[j][g] = *reinterpret_cast<const float4*>(p.gate + (size_t)sample * p.ld_gate + x);
          rv[j][g] = *reinterpret_cast<const float4*>(rsrc + (size_t)y * p.ldo + x);
        }
    }
    if (EPI == EPI_ACCUM_F32) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          rv[j][g] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.out) + (size_t)y * p.ldo +
                                                      tx * BN + wx * 64 + j * 32 + 8 * g + 4 * fhalf);
    }
    if (EPI == EPI_GELUGRAD_TE) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const TE* a = reinterpret_cast<const TE*>(p.aux) + (size_t)y * p.ldo + tx * BN + wx * 64 + j * 32 + 8 * g + 4 * fhalf;
          rv[j][g] = make_float4(load_elem(a), load_elem(a + 1), load_elem(a + 2), load_elem(a + 3));
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int x = tx * BN + wx * 64 + j * 32 + 8 * g + 4 * fhalf;
        float v0 = acc[i][j][4 * g + 0], v1 = acc[i][j][4 * g + 1], v2 = acc[i][j][4 * g + 2],
              v3 = acc[i][j][4 * g + 3];
        if (kBias) { v0 += bv[j][g].x; v1 += bv[j][g].y; v2 += bv[j][g].z; v3 += bv[j][g].w; }
        if (EPI == EPI_ROWBIAS_TE) { v0 += rb; v1 += rb; v2 += rb; v3 += rb; }
        const size_t o = (size_t)y * p.ldo + x;
        if (EPI == EPI_BIAS_F32 || EPI == EPI_NONE_F32) {
          store4(reinterpret_cast<float*>(p.out) + o, v0, v1, v2, v3);
        } else if (EPI == EPI_ACCUM_F32) {
          store4(reinterpret_cast<float*>(p.out) + o, rv[j][g].x + v0, rv[j][g].y + v1, rv[j][g].z + v2,
                 rv[j][g].w + v3);
        } else if (EPI == EPI_GATE_RES) {
          if (p.out2) store4(reinterpret_cast<TE*>(p.out2) + o, v0, v1, v2, v3);  // branch output (training)
          store4(reinterpret_cast<float*>(p.out) + o, rv[j][g].x + gv[j][g].x * v0, rv[j][g].y + gv[j][g].y * v1,
                 rv[j][g].z + gv[j][g].z * v2, rv[j][g].w + gv[j][g].w * v3);
        } else if (EPI == EPI_BIAS_SILU_TE) {
          if (p.out2) store4(reinterpret_cast<TE*>(p.out2) + o, v0, v1, v2, v3);  // pre-activation (training)
          store4(reinterpret_cast<TE*>(p.out) + o, silu_t<FAST>(v0), silu_t<FAST>(v1), silu_t<FAST>(v2),
                 silu_t<FAST>(v3));
        } else if (EPI == EPI_BIAS_GELU_TE) {
          if (p.out2) store4(reinterpret_cast<TE*>(p.out2) + o, v0, v1, v2, v3);
          store4(reinterpret_cast<TE*>(p.out) + o, gelu_tanh_t<FAST>(v0), gelu_tanh_t<FAST>(v1),
                 gelu_tanh_t<FAST>(v2), gelu_tanh_t<FAST>(v3));
        } else if (EPI == EPI_GELUGRAD_TE) {
          store4(reinterpret_cast<TE*>(p.out) + o, v0 * gelu_tanh_grad_t<FAST>(rv[j][g].x),
                 v1 * gelu_tanh_grad_t<FAST>(rv[j][g].y), v2 * gelu_tanh_grad_t<FAST>(rv[j][g].z),
                 v3 * gelu_tanh_grad_t<FAST>(rv[j][g].w));
        } else {  // EPI_BIAS_TE, EPI_ROWBIAS_TE, EPI_NONE_TE
          store4(reinterpret_cast<TE*>(p.out) + o, v0, v1, v2, v3);
        }
      }
    }
  }
}

template <typename TE, int EPI> int launch_t(const GemmP& p, hipStream_t st) {
  const int nwg = (p.My / BM) * (p.Nx / BN);
  hipLaunchKernelGGL((gemm_kernel<TE, EPI>), dim3(nwg), dim3(256), 0, st, p);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

template <typename TE> int launch_e(int epi, const GemmP& p, hipStream_t st) {
  switch (epi) {
    case EPI_BIAS_F32: return launch_t<TE, EPI_BIAS_F32>(p, st);
    case EPI_BIAS_TE: return launch_t<TE, EPI_BIAS_TE>(p, st);
    case EPI_BIAS_SILU_TE: return launch_t<TE, EPI_BIAS_SILU_TE>(p, st);
    case EPI_ROWBIAS_TE: return launch_t<TE, EPI_ROWBIAS_TE>(p, st);
    case EPI_BIAS_GELU_TE: return launch_t<TE, EPI_BIAS_GELU_TE>(p, st);
    case EPI_GATE_RES: return launch_t<TE, EPI_GATE_RES>(p, st);
    case EPI_NONE_F32: return launch_t<TE, EPI_NONE_F32>(p, st);
    case EPI_NONE_TE: return launch_t<TE, EPI_NONE_TE>(p, st);
    case EPI_ACCUM_F32: return launch_t<TE, EPI_ACCUM_F32>(p, st);
    case EPI_GELUGRAD_TE: return launch_t<TE, EPI_GELUGRAD_TE>(p, st);
  }
  set_error("gemm: unknown epilogue %d", epi);
  return OSUD_ERR_ARG;
}

}  // namespace

int launch_gemm(int prec, int epi, const GemmP& p, hipStream_t st) {
  const int esz = (int)elem_size(prec);
  OSUD_CHECK_ARG(p.My > 0 && p.Nx > 0 && p.K > 0 && p.My % BM == 0 && p.Nx % BN == 0 && (p.K * esz) % SLAB == 0,
                 "gemm: My=%d Nx=%d must be multiples of 128 and K=%d a multiple of %d", p.My, p.Nx, p.K, SLAB / esz);
  OSUD_CHECK_ARG((p.ldy * esz) % 16 == 0 && (p.ldx * esz) % 16 == 0 && p.ldo % 4 == 0,
                 "gemm: leading dimensions must keep 16-byte alignment (ldy=%d ldx=%d ldo=%d)", p.ldy, p.ldx, p.ldo);
  OSUD_CHECK_ARG(p.Y && p.X && p.out, "gemm: null operand");
  if (epi == EPI_GATE_RES)
    OSUD_CHECK_ARG(p.gate && p.bias && p.rows_per_sample > 0 && p.n_samples > 0, "gemm: gated epilogue needs gate/bias");
  if (epi == EPI_BIAS_F32 || epi == EPI_BIAS_TE || epi == EPI_BIAS_SILU_TE || epi == EPI_ROWBIAS_TE ||
      epi == EPI_BIAS_GELU_TE)
    OSUD_CHECK_ARG(p.bias != nullptr, "gemm: epilogue %d needs a bias", epi);
  if (epi == EPI_GELUGRAD_TE) OSUD_CHECK_ARG(p.aux != nullptr, "gemm: epilogue %d needs aux", epi);
  return prec == OSUD_PREC_BF16 ? launch_e<bf16_t>(epi, p, st) : launch_e<float>(epi, p, st);
}

}  // namespace osud
