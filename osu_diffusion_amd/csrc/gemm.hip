// Host side of the GEMM: argument checks, dispatch to the per-element-type translation units, the process-wide state of the
// dynamic tile queues, and the split-K combine.  See gemm.h for the design, gemm_kernel.h for the kernel.  gfx950 only.
#include <atomic>
#include <mutex>
#include <string>

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gemm.h"

namespace osud {

// one entry per operand element type (gemm_<type>.hip)
int launch_gemm_bf16(int epi, const GemmP& p, hipStream_t st);
int launch_gemm_f32(int epi, const GemmP& p, hipStream_t st);
int launch_gemm_fp8(int epi, const GemmP& p, hipStream_t st);
int launch_gemm_x3(int epi, const GemmP& p, hipStream_t st);
int launch_gemm_h8(int epi, const GemmP& p, hipStream_t st);
int launch_gemm_w8(int epi, const GemmP& p, hipStream_t st);
int launch_gemm_f16(int epi, const GemmP& p, hipStream_t st);

namespace {
constexpr int SLAB = 128;
// Per-device state (one process normally drives one GPU, but nothing here may point a second device's kernels at the first
// device's memory): CU count and the tile-queue counter pool, indexed by the current device.
constexpr int kMaxDevices = 16;
int cur_device() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) dev = 0;
  return dev;
}
int num_cus() {
  static int n[kMaxDevices] = {};
  const int dev = cur_device();
  if (n[dev] == 0) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) == hipSuccess) n[dev] = prop.multiProcessorCount;
    if (n[dev] <= 0) n[dev] = 256;
  }
  return n[dev];
}

// Counter sets ({8 per-XCD tickets, done}, 64-byte slots) of the dynamic tile queue: zeroed once, re-armed by the last workgroup of every launch that used one.  Launches
// take slots round-robin; kernels on one stream run in order, so a slot is idle again long before it comes round (the library is
// single-stream per device by contract -- include/osud.h: handles are not thread-safe, all work goes to the caller's stream; a
// slot baked into a captured graph belongs to that graph's launch and is re-armed by it on every replay).
constexpr int kSchedSlots = 256;
// Off by default: alone on the GPU the queue costs 0.7 % of a training step (a one-off drain for the first ticket, less regular
// tile order).  Data-parallel training switches it on (osud_set_gemm_dynamic_tiles), where collectives share the compute units.
std::atomic<int> g_dynamic_tiles{0};
unsigned* g_sched_pool[kMaxDevices] = {};
int gemm_sched_init_impl() {
  const int dev = cur_device();
  if (g_sched_pool[dev]) return OSUD_OK;
  unsigned* pool = nullptr;
  OSUD_HIP(hipMalloc(&pool, kSchedSlots * 16 * sizeof(unsigned)));
  OSUD_HIP(hipMemset(pool, 0, kSchedSlots * 16 * sizeof(unsigned)));
  g_sched_pool[dev] = pool;
  return OSUD_OK;
}
unsigned* sched_slot() {
  static std::atomic<unsigned> seq{0};
  unsigned* pool = g_sched_pool[cur_device()];
  return pool ? pool + 16 * (seq.fetch_add(1) % kSchedSlots) : nullptr;
}

}  // namespace

// ---- options ------------------------------------------------------------------------------------------------------------
namespace {
struct OptDef {
  const char* name;
  int def, lo, hi;
};
constexpr OptDef kOpts[OPT_COUNT] = {
    {"wgrad_side_stream", 1, 0, 1}, {"sample_graph", 1, 0, 1},    {"embed_const", 1, 0, 1},  {"tvec_table", 1, 0, 1},
    {"split_first", 1, 0, 1},       {"attn_fwd_kernel", 0, 0, 2}, {"attn_bwd_kernel", 0, 0, 2}, {"gemm_tile", 0, 0, 1256},
    {"f8_twins_only", 1, 0, 1},     {"debug_sync", 0, 0, 1},      {"f16m8_forms", 11, 0, 15},    {"gemm_loop", 1, 0, 1},
    {"gelu_code", 1, 0, 1},
};
std::atomic<int> g_opt[OPT_COUNT];
std::atomic<unsigned> g_opt_epoch{0};
// The defaults, overridden ONCE by the environment variable OSUD_OPTIONS="name=value,name=value" (for command-line A/B runs of an
// unmodified script; tests and hosts call osud_set_option).  Unknown names there are an error on stderr, not silently ignored.
void opt_init() {
  static std::once_flag once;
  std::call_once(once, [] {
    for (int i = 0; i < OPT_COUNT; ++i) g_opt[i].store(kOpts[i].def, std::memory_order_relaxed);
    const char* e = getenv("OSUD_OPTIONS");
    if (!e) return;
    std::string s(e);
    size_t pos = 0;
    while (pos < s.size()) {
      size_t end = s.find(',', pos);
      if (end == std::string::npos) end = s.size();
      const std::string item = s.substr(pos, end - pos);
      const size_t eq = item.find('=');
      bool ok = false;
      if (eq != std::string::npos)
        for (int i = 0; i < OPT_COUNT; ++i)
          if (item.substr(0, eq) == kOpts[i].name) {
            const char* vs = item.c_str() + eq + 1;
            char* endp = nullptr;
            const long v = strtol(vs, &endp, 10);  // ("sample_graph=off" or "gemm_tile=" must not become a silent 0)
            if (endp != vs && *endp == '\0' && v >= kOpts[i].lo && v <= kOpts[i].hi) {
              g_opt[i].store((int)v, std::memory_order_relaxed);
              ok = true;
            }
          }
      if (!ok && !item.empty()) fprintf(stderr, "[osud] OSUD_OPTIONS: ignoring '%s' (unknown option, or a value that is not a number in range)\n", item.c_str());
      pos = end + 1;
    }
  });
}
}  // namespace

int opt(Opt o) {
  opt_init();
  return g_opt[o].load(std::memory_order_relaxed);
}
unsigned opt_epoch() { return g_opt_epoch.load(std::memory_order_relaxed); }
int opt_set(const char* name, int value) {
  opt_init();
  OSUD_CHECK_ARG(name != nullptr, "set_option: null name");
  for (int i = 0; i < OPT_COUNT; ++i)
    if (strcmp(name, kOpts[i].name) == 0) {
      OSUD_CHECK_ARG(value == -1 || (value >= kOpts[i].lo && value <= kOpts[i].hi), "set_option: %s takes %d..%d (or -1 = default), got %d", name,
                     kOpts[i].lo, kOpts[i].hi, value);
      g_opt[i].store(value == -1 ? kOpts[i].def : value, std::memory_order_relaxed);
      // (the epoch invalidates captured sampler steps: only options a captured step reads move it -- not the training / triage switches)
      if (i != OPT_WGRAD_SIDE_STREAM && i != OPT_DEBUG_SYNC && i != OPT_F8_TWINS_ONLY && i != OPT_GELU_CODE)
        g_opt_epoch.fetch_add(1, std::memory_order_relaxed);
      return OSUD_OK;
    }
  set_error("set_option: unknown option '%s'", name);
  return OSUD_ERR_ARG;
}
int opt_get(const char* name, int* value) {
  opt_init();
  OSUD_CHECK_ARG(name != nullptr && value != nullptr, "get_option: null argument");
  for (int i = 0; i < OPT_COUNT; ++i)
    if (strcmp(name, kOpts[i].name) == 0) {
      *value = g_opt[i].load(std::memory_order_relaxed);
      return OSUD_OK;
    }
  set_error("get_option: unknown option '%s'", name);
  return OSUD_ERR_ARG;
}

int gemm_num_cus() { return num_cus(); }
unsigned* gemm_sched_slot() { return sched_slot(); }
bool gemm_dynamic_tiles_wanted() { return gemm_dynamic_tiles_on(); }

int gemm_sched_init() { return gemm_sched_init_impl(); }
unsigned* gemm_ticket_slot() {
  if (!g_sched_pool[cur_device()] && gemm_sched_init_impl() != OSUD_OK) return nullptr;
  return sched_slot();
}
bool gemm_dynamic_tiles_on() { return g_dynamic_tiles.load(std::memory_order_relaxed) != 0; }
void gemm_set_dynamic_tiles(int on) { g_dynamic_tiles.store(on > 0 ? 1 : 0, std::memory_order_relaxed); }

int launch_gemm(int prec, int epi, const GemmP& p_in, hipStream_t st) {
  if (!g_sched_pool[cur_device()]) {  // normally done at handle creation; never reached while a stream is being captured
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone) OSUD_TRY(gemm_sched_init_impl());
  }
  GemmP p = p_in;
  const int esz = (int)elem_size(prec);
  OSUD_CHECK_ARG(p.My > 0 && p.Nx > 0 && p.K > 0 && p.My % 128 == 0 && p.Nx % 128 == 0 && (p.K * esz) % SLAB == 0,
                 "gemm: My=%d Nx=%d must be multiples of 128 and K=%d a multiple of %d", p.My, p.Nx, p.K, SLAB / esz);
  OSUD_CHECK_ARG((p.ldy * esz) % 16 == 0 && (p.ldx * esz) % 16 == 0 && p.ldo % 8 == 0,
                 "gemm: leading dimensions must keep 16-byte alignment (ldy=%d ldx=%d ldo=%d)", p.ldy, p.ldx, p.ldo);
  // (fp8 training: the two epilogues that write an e4m3 twin may drop their bf16 output when nothing reads it)
  const bool twin_only = prec == OSUD_PREC_FP8 && p.out8 != nullptr && (epi == EPI_BIAS_GELU_BF || epi == EPI_GELUGRAD_TE);
  OSUD_CHECK_ARG(p.Y && p.X && (p.out || twin_only), "gemm: null operand");
  OSUD_CHECK_ARG(p.seg_rows == 0 || (epi == EPI_NONE_F32 && p.split_k <= 1 && p.seg_rows % 32 == 0 && p.seg_out != nullptr),
                 "gemm: segmented output needs the plain f32 epilogue, no split-K and a device table of segment pointers");
  OSUD_CHECK_ARG((size_t)p.ldy * esz * 256 < (1ull << 31) && (size_t)p.ldx * esz * 256 < (1ull << 31),
                 "gemm: leading dimension too large for 32-bit panel offsets");
  if (epi == EPI_GATE_RES)
    OSUD_CHECK_ARG(p.gate && p.bias && p.rows_per_sample > 0 && p.rows_per_sample % 32 == 0 && p.n_samples > 0,
                   "gemm: gated epilogue needs gate/bias and rows_per_sample %% 32 == 0");
  if (epi == EPI_BIAS_F32 || epi == EPI_BIAS_TE || epi == EPI_BIAS_SILU_TE || epi == EPI_ROWBIAS_TE ||
      epi == EPI_BIAS_GELU_TE || epi == EPI_BIAS_GELU_BF || epi == EPI_BIAS_GELU_ALT)
    OSUD_CHECK_ARG(p.bias != nullptr, "gemm: epilogue %d needs a bias", epi);
  if (epi == EPI_GELUGRAD_TE) OSUD_CHECK_ARG(p.aux != nullptr, "gemm: epilogue %d needs aux", epi);
  if (p.aux_code)
    OSUD_CHECK_ARG((prec == OSUD_PREC_BF16 || prec == OSUD_PREC_FP8) && p.ldo % 32 == 0 &&
                       (epi == EPI_BIAS_GELU_TE || epi == EPI_BIAS_GELU_BF || epi == EPI_GELUGRAD_TE),
                   "gemm: the 8-bit code of the saved GELU derivative exists for the bf16-output GELU epilogues and ldo %% 32 == 0 (prec %d, epilogue %d, ldo %d)",
                   prec, epi, p.ldo);
  if (p.split_k > 1) {
    OSUD_CHECK_ARG(epi == EPI_NONE_F32 || epi == EPI_NONE_TE, "gemm: split-K needs a plain epilogue");
    OSUD_CHECK_ARG((size_t)p.K * esz / SLAB >= (size_t)p.split_k, "gemm: K=%d does not split %d ways", p.K, p.split_k);
  }
  if (prec == OSUD_PREC_BF16X3) {
    OSUD_CHECK_ARG(p.split_k <= 1, "gemm: the split-bf16 operand form has no split-K");
    OSUD_CHECK_ARG(p.K % 64 == 0 && p.ldy % 8 == 0 && p.ldx % 8 == 0 && p.ldy >= p.K && p.ldx >= p.K,
                   "gemm: split-bf16 operands need K %% 64 == 0 and plane widths (ld) >= K, multiples of 8 (K=%d ldy=%d ldx=%d)", p.K, p.ldy, p.ldx);
    return launch_gemm_x3(epi, p, st);
  }
  if (prec == OSUD_PREC_F16F8) {
    OSUD_CHECK_ARG(p.split_k <= 1, "gemm: the fp16 + e4m3 operand form has no split-K");
    OSUD_CHECK_ARG(p.K % 32 == 0 && p.ldy % 32 == 0 && p.ldx % 32 == 0 && p.ldy >= p.K && p.ldx >= p.K,
                   "gemm: fp16 + e4m3 operands are K-blocked in groups of 32 (K=%d ldy=%d ldx=%d)", p.K, p.ldy, p.ldx);
    return launch_gemm_h8(epi, p, st);
  }
  if (prec == OSUD_PREC_F16W8) {
    OSUD_CHECK_ARG(p.split_k <= 1, "gemm: the fp16 x (fp16 + e4m3) operand form has no split-K");
    OSUD_CHECK_ARG(p.K % 128 == 0 && p.ldy % 128 == 0 && p.ldx % 128 == 0 && p.ldy >= p.K && p.ldx >= p.K,
                   "gemm: fp16 x (fp16 + e4m3) operands are K-blocked in super-groups of 128 (K=%d ldy=%d ldx=%d)", p.K, p.ldy, p.ldx);
    return launch_gemm_w8(epi, p, st);
  }
  if (prec == OSUD_PREC_F16) {
    OSUD_CHECK_ARG(p.split_k <= 1, "gemm: the fp16 operand form is built for the forward pass (no split-K)");
    return launch_gemm_f16(epi, p, st);
  }
  if (prec == 2) return launch_gemm_fp8(epi, p, st);
  return prec == OSUD_PREC_BF16 ? launch_gemm_bf16(epi, p, st) : launch_gemm_f32(epi, p, st);
}

namespace {
__global__ void splitk_reduce_kernel(const float4* __restrict__ part, int splits, size_t stride4, float4* __restrict__ out,
                                     size_t n4) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    // the partial slabs are read exactly once: streaming loads (in-step, same box: 23.43 -> 23.31 ms per training step; a streaming store
    // of the sum -- read by the optimizer at the end of the step -- made no difference: profiles/r05_ab_runs.md)
    typedef float f4v __attribute__((ext_vector_type(4)));
    auto ld = [&](size_t k) -> float4 {
      const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(part + k));
      return make_float4(t[0], t[1], t[2], t[3]);
    };
    // eight slabs in flight per lane, added in slab order (the order of the sum is part of the result: gradients are bit-reproducible)
    float4 a = ld(i);
    int s = 1;
    for (; s + 8 <= splits; s += 8) {
      float4 b[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) b[j] = ld((size_t)(s + j) * stride4 + i);
#pragma unroll
      for (int j = 0; j < 8; ++j) { a.x += b[j].x; a.y += b[j].y; a.z += b[j].z; a.w += b[j].w; }
    }
    if (s < splits) {
      float4 b[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (s + j < splits) b[j] = ld((size_t)(s + j) * stride4 + i);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (s + j < splits) { a.x += b[j].x; a.y += b[j].y; a.z += b[j].z; a.w += b[j].w; }
    }
    out[i] = a;
  }
}
}  // namespace

int launch_splitk_reduce(const float* part, int splits, size_t stride, float* out, size_t n, hipStream_t st) {
  OSUD_CHECK_ARG(n % 4 == 0 && stride % 4 == 0, "splitk_reduce: sizes must be multiples of 4");
  const size_t n4 = n / 4;
  const int grid = (int)((n4 + 255) / 256 > 2048 ? 2048 : (n4 + 255) / 256);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid), dim3(256), 0, st, reinterpret_cast<const float4*>(part), splits,
                     stride / 4, reinterpret_cast<float4*>(out), n4);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

}  // namespace osud
