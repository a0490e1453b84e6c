// Diffusion schedule tables and the per-step sampler update.
//
// Host side (fp64, exactly the reference's arithmetic):
//   GaussianDiffusion.__init__   diffusion/gaussian_diffusion.py:167-211
//   SpacedDiffusion.__init__     diffusion/respace.py:72-86   (beta re-derivation + timestep_map)
// Device side: one elementwise kernel that fuses forward_with_cfg's combine
// (models.py:338-343), p_mean_variance's LEARNED_RANGE / epsilon branch
// (gaussian_diffusion.py:312-358), and p_sample / ddim_sample (:454-467, :589-610).  The
// coefficient table lives on the device (the reference re-uploads 8 numpy scalars per step,
// gaussian_diffusion.py:960); values are the fp64 tables cast to fp32, as `.float()` does.
// FP contraction is off in this file so the update is evaluated with the same roundings as the
// reference's unfused torch ops.
#include <math.h>

#include <vector>

#include "kernels.h"

#pragma clang fp contract(off)

struct osud_sched {
  int n = 0;
  int n_base = 0;
  std::vector<double> betas, ac, ac_prev, sqrt_ac, sqrt_1m_ac, sqrt_recip, sqrt_recipm1, post_var, post_logvar,
      coef1, coef2, log_betas;
  std::vector<int64_t> tmap;
  float* d_coefs = nullptr;  // [n][8]
  float* d_train = nullptr;  // [n][8] sqrt_ac, sqrt_1m_ac, sqrt_recip, sqrt_recipm1, coef1, coef2, post_logvar, log_beta
  int64_t* d_tmap = nullptr;
  int device = -1;
};

namespace osud {

int sched_upload(osud_sched* s) {
  if (s->d_coefs) return OSUD_OK;
  std::vector<float> h((size_t)s->n * 8);
  for (int i = 0; i < s->n; ++i) {
    float* r = &h[(size_t)i * 8];
    r[0] = (float)s->sqrt_recip[i];
    r[1] = (float)s->sqrt_recipm1[i];
    r[2] = (float)s->coef1[i];
    r[3] = (float)s->coef2[i];
    r[4] = (float)s->post_logvar[i];
    r[5] = (float)s->log_betas[i];
    r[6] = (float)s->ac[i];
    r[7] = (float)s->ac_prev[i];
  }
  OSUD_HIP(hipGetDevice(&s->device));
  OSUD_HIP(hipMalloc(&s->d_coefs, h.size() * sizeof(float)));
  OSUD_HIP(hipMalloc(&s->d_tmap, (size_t)s->n * sizeof(int64_t)));
  OSUD_HIP(hipMemcpy(s->d_coefs, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  OSUD_HIP(hipMemcpy(s->d_tmap, s->tmap.data(), (size_t)s->n * sizeof(int64_t), hipMemcpyHostToDevice));
  return OSUD_OK;
}
const float* sched_coefs(const osud_sched* s) { return s->d_coefs; }
const float* sched_train_coefs(const osud_sched* s) { return s->d_train; }
int sched_upload_train(osud_sched* s) {
  if (s->d_train) return OSUD_OK;
  std::vector<float> h((size_t)s->n * 8);
  for (int i = 0; i < s->n; ++i) {
    float* r = &h[(size_t)i * 8];
    r[0] = (float)s->sqrt_ac[i];
    r[1] = (float)s->sqrt_1m_ac[i];
    r[2] = (float)s->sqrt_recip[i];
    r[3] = (float)s->sqrt_recipm1[i];
    r[4] = (float)s->coef1[i];
    r[5] = (float)s->coef2[i];
    r[6] = (float)s->post_logvar[i];
    r[7] = (float)s->log_betas[i];
  }
  OSUD_HIP(hipMalloc(&s->d_train, h.size() * sizeof(float)));
  OSUD_HIP(hipMemcpy(s->d_train, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  return OSUD_OK;
}
const int64_t* sched_tmap_dev(const osud_sched* s) { return s->d_tmap; }

namespace {

// Philox4x32-10 (Salmon et al. 2011) -> two N(0,1) via Box-Muller.  Own stream keyed by
// (seed, step, element); NOT torch's generator (parity runs pass the noise in instead).
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
  const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
  const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
  const uint32_t n0 = hi1 ^ c[1] ^ k0, n1 = lo1, n2 = hi0 ^ c[3] ^ k1, n3 = lo0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ float philox_normal(uint64_t seed, uint32_t step, uint32_t idx) {
  uint32_t c[4] = {idx, step, 0x6f737564u /* "osud" */, 0u};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  const float u1 = ((c[0] >> 8) + 1) * (1.0f / 16777216.0f);  // (0, 1]
  const float u2 = (c[1] >> 8) * (1.0f / 16777216.0f);        // [0, 1)
  return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}

__global__ void sampler_step_kernel(const float* __restrict__ coefs, int mode, float eta,
                                    const float* __restrict__ model_out, const float* __restrict__ x,
                                    const int64_t* __restrict__ t_index, const int* __restrict__ step_state,
                                    const float* __restrict__ noise, size_t noise_step_stride, uint64_t seed, int N,
                                    int T, float cfg_scale, int clip, const uint8_t* __restrict__ keep,
                                    const float* __restrict__ known, float* __restrict__ x_out,
                                    float* __restrict__ pred_xstart) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * 2 * T) return;
  const int t = i % T, ch = (i / T) % 2, n = i / (2 * T);
  int step, exec_k = 0;
  if (step_state != nullptr) {  // graph-replayed loop: step index, noise offset and the Philox seed live on the device, so ONE
    step = step_state[1];       // captured graph serves every step of every run (the seed is not baked into the launch)
    exec_k = step_state[2];
    seed = (uint64_t)(uint32_t)step_state[4] | ((uint64_t)(uint32_t)step_state[5] << 32);
  } else {
    step = (int)t_index[n];
  }
  const float* cf = coefs + (size_t)step * 8;
  const float A = cf[0], B = cf[1], c1 = cf[2], c2 = cf[3], min_log = cf[4], max_log = cf[5];
  float eps;
  if (cfg_scale >= 0.f) {  // models.py:338-343
    const int half = N / 2, nn = n % half;
    const float ce = model_out[((size_t)nn * 4 + ch) * T + t];
    const float ue = model_out[((size_t)(nn + half) * 4 + ch) * T + t];
    eps = ue + cfg_scale * (ce - ue);
  } else {
    eps = model_out[((size_t)n * 4 + ch) * T + t];
  }
  const float v = model_out[((size_t)n * 4 + 2 + ch) * T + t];
  const float xv = x[i];
  const float frac = (v + 1.0f) / 2.0f;                              // gaussian_diffusion.py:322
  const float log_var = frac * max_log + (1.0f - frac) * min_log;    // :323
  float x0 = A * xv - B * eps;                                       // :373-376
  if (keep != nullptr && keep[i] == 0) x0 = known[i];                // denoised_fn (in-paint mask), before the clamp :341-344
  if (clip) x0 = fminf(fmaxf(x0, -1.0f), 2.0f);                      // :345
  float nz;
  if (noise != nullptr) nz = noise[(size_t)exec_k * noise_step_stride + i];
  else nz = philox_normal(seed, (uint32_t)step, (uint32_t)i);
  const float nonzero = step != 0 ? 1.0f : 0.0f;                     // :455-457
  float sample;
  if (mode == OSUD_SAMPLER_P) {
    const float mean = c1 * x0 + c2 * xv;                            // :255-258
    sample = mean + nonzero * expf(0.5f * log_var) * nz;             // :466
  } else {
    const float ab = cf[6], abp = cf[7];
    const float eps2 = (A * xv - x0) / B;                            // :378-382
    const float sigma = eta * sqrtf((1.0f - abp) / (1.0f - ab)) * sqrtf(1.0f - ab / abp);  // :595-599
    const float mean = x0 * sqrtf(abp) + sqrtf(1.0f - abp - sigma * sigma) * eps2;         // :602-605
    sample = mean + nonzero * sigma * nz;                            // :609
  }
  x_out[i] = sample;
  if (pred_xstart != nullptr) pred_xstart[i] = x0;
}

// Loop bookkeeping, one block: state[0] = next step index, state[1] = current, state[2] = number
// of steps executed before this one, state[4..5] = Philox seed of the run.  Fills the model's timestep (timestep_map[i], respace.py:127-132).
__global__ void step_begin_kernel(int* state, const int64_t* __restrict__ tmap, int64_t* __restrict__ t_model,
                                  int64_t* __restrict__ t_index, int N) {
  const int cur = state[0];
  const int k = state[3];
  __syncthreads();
  for (int n = threadIdx.x; n < N; n += blockDim.x) {
    t_model[n] = tmap[cur];
    if (t_index) t_index[n] = cur;
  }
  if (threadIdx.x == 0) {
    state[1] = cur;
    state[2] = k;
    state[0] = cur - state[6];  // (1: the next step of a loop; 0: the same step again -- osud_sample_repeat)
    state[3] = k + 1;
  }
}

__global__ void step_init_kernel(int* state, int first, uint64_t seed, int dec) {
  state[0] = first;
  state[1] = first;
  state[2] = 0;
  state[3] = 0;
  state[4] = (int)(uint32_t)seed;
  state[5] = (int)(uint32_t)(seed >> 32);
  state[6] = dec;
}

}  // namespace

int launch_step_init(int* step_state, int first, uint64_t seed, hipStream_t st, int dec) {
  hipLaunchKernelGGL(step_init_kernel, dim3(1), dim3(1), 0, st, step_state, first, seed, dec);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

int launch_sampler_step(const float* coefs, int mode, float eta, const float* model_out, const float* x,
                        const int64_t* t_index, const int* step_state, const float* noise, size_t noise_step_stride,
                        uint64_t seed, int N, int T, float cfg_scale, int clip, const osud_inpaint* inpaint, float* x_out,
                        float* pred_xstart, hipStream_t st) {
  OSUD_CHECK_ARG(inpaint == nullptr || (inpaint->keep != nullptr && inpaint->known != nullptr),
                 "sampler: in-painting needs both the keep mask and the known values");
  OSUD_CHECK_ARG(mode == OSUD_SAMPLER_P || mode == OSUD_SAMPLER_DDIM, "sampler: unknown mode %d", mode);
  OSUD_CHECK_ARG(cfg_scale < 0.f || N % 2 == 0, "sampler: classifier-free guidance needs an even batch, got %d", N);
  const int total = N * 2 * T;
  hipLaunchKernelGGL(sampler_step_kernel, dim3((total + 255) / 256), dim3(256), 0, st, coefs, mode, eta, model_out, x,
                     t_index, step_state, noise, noise_step_stride, seed, N, T, cfg_scale, clip,
                     inpaint ? inpaint->keep : nullptr, inpaint ? inpaint->known : nullptr, x_out, pred_xstart);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

int launch_step_begin(int* step_state, const int64_t* tmap_dev, int64_t* t_model, int64_t* t_index, int N,
                      hipStream_t st) {
  hipLaunchKernelGGL(step_begin_kernel, dim3(1), dim3(256), 0, st, step_state, tmap_dev, t_model, t_index, N);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

}  // namespace osud

// ------------------------------------------------------------------------------- C ABI
using namespace osud;

extern "C" int osud_sched_create(const double* betas, int n_base, const int64_t* use_timesteps, int n_use,
                                 osud_sched** out) {
  OSUD_CHECK_ARG(betas && use_timesteps && out && n_base > 0 && n_use > 0, "sched_create: null/empty argument");
  for (int i = 0; i < n_base; ++i)  // gaussian_diffusion.py:176
    OSUD_CHECK_ARG(betas[i] > 0 && betas[i] <= 1, "sched_create: betas must be in (0, 1], betas[%d]=%g", i, betas[i]);
  std::vector<char> keep((size_t)n_base, 0);
  for (int i = 0; i < n_use; ++i) {
    OSUD_CHECK_ARG(use_timesteps[i] >= 0 && use_timesteps[i] < n_base, "sched_create: timestep %lld outside [0, %d)",
                   (long long)use_timesteps[i], n_base);
    keep[(size_t)use_timesteps[i]] = 1;
  }
  osud_sched* s = new osud_sched();
  s->n_base = n_base;
  // respace.py:76-84: walk the base alphas_cumprod, re-derive betas for the kept steps
  double acp = 1.0, last = 1.0;
  for (int i = 0; i < n_base; ++i) {
    acp *= (1.0 - betas[i]);  // np.cumprod is a left-to-right product
    if (keep[(size_t)i]) {
      s->betas.push_back(1 - acp / last);
      last = acp;
      s->tmap.push_back(i);
    }
  }
  const int n = s->n = (int)s->betas.size();
  s->ac.resize(n); s->ac_prev.resize(n); s->sqrt_ac.resize(n); s->sqrt_1m_ac.resize(n); s->sqrt_recip.resize(n);
  s->sqrt_recipm1.resize(n); s->post_var.resize(n); s->post_logvar.resize(n); s->coef1.resize(n); s->coef2.resize(n);
  s->log_betas.resize(n);
  double cp = 1.0;
  for (int i = 0; i < n; ++i) {  // gaussian_diffusion.py:180-211
    const double b = s->betas[i], alpha = 1.0 - b;
    const double prev = cp;
    cp *= alpha;
    s->ac[i] = cp;
    s->ac_prev[i] = i == 0 ? 1.0 : prev;
    s->sqrt_ac[i] = sqrt(cp);
    s->sqrt_1m_ac[i] = sqrt(1.0 - cp);
    s->sqrt_recip[i] = sqrt(1.0 / cp);
    s->sqrt_recipm1[i] = sqrt(1.0 / cp - 1);
    s->post_var[i] = b * (1.0 - s->ac_prev[i]) / (1.0 - cp);
    s->coef1[i] = b * sqrt(s->ac_prev[i]) / (1.0 - cp);
    s->coef2[i] = (1.0 - s->ac_prev[i]) * sqrt(alpha) / (1.0 - cp);
    s->log_betas[i] = log(b);
  }
  for (int i = 0; i < n; ++i)  // :198-202: log of [pv[1], pv[1], pv[2], ...]
    s->post_logvar[i] = n > 1 ? log(s->post_var[i == 0 ? 1 : i]) : 0.0;
  *out = s;
  return OSUD_OK;
}

extern "C" void osud_sched_destroy(osud_sched* s) {
  if (!s) return;
  if (s->d_coefs) (void)hipFree(s->d_coefs);
  if (s->d_tmap) (void)hipFree(s->d_tmap);
  if (s->d_train) (void)hipFree(s->d_train);
  delete s;
}

extern "C" int osud_sched_num_timesteps(const osud_sched* s) { return s ? s->n : 0; }

extern "C" int osud_sched_table(const osud_sched* s, const char* name, double* out, int n) {
  OSUD_CHECK_ARG(s && name && out, "sched_table: null argument");
  OSUD_CHECK_ARG(n == s->n, "sched_table: expected %d entries, got %d", s->n, n);
  const std::string k(name);
  const std::vector<double>* v = nullptr;
  if (k == "betas") v = &s->betas;
  else if (k == "alphas_cumprod") v = &s->ac;
  else if (k == "alphas_cumprod_prev") v = &s->ac_prev;
  else if (k == "sqrt_alphas_cumprod") v = &s->sqrt_ac;
  else if (k == "sqrt_one_minus_alphas_cumprod") v = &s->sqrt_1m_ac;
  else if (k == "sqrt_recip_alphas_cumprod") v = &s->sqrt_recip;
  else if (k == "sqrt_recipm1_alphas_cumprod") v = &s->sqrt_recipm1;
  else if (k == "posterior_variance") v = &s->post_var;
  else if (k == "posterior_log_variance_clipped") v = &s->post_logvar;
  else if (k == "posterior_mean_coef1") v = &s->coef1;
  else if (k == "posterior_mean_coef2") v = &s->coef2;
  else if (k == "log_betas") v = &s->log_betas;
  OSUD_CHECK_ARG(v != nullptr, "sched_table: unknown table '%s'", name);
  for (int i = 0; i < n; ++i) out[i] = (*v)[(size_t)i];
  return OSUD_OK;
}

extern "C" int osud_sched_timestep_map(const osud_sched* s, int64_t* out, int n) {
  OSUD_CHECK_ARG(s && out && n == s->n, "sched_timestep_map: expected %d entries", s ? s->n : 0);
  for (int i = 0; i < n; ++i) out[i] = s->tmap[(size_t)i];
  return OSUD_OK;
}

extern "C" int osud_sampler_step_inpaint(const osud_sched* s, int mode, float eta, const float* model_out,
                                         const float* x, const int64_t* t_index, const float* noise, int N, int T,
                                         float cfg_scale, int clip, const osud_inpaint* inpaint, float* x_out,
                                         float* pred_xstart, osud_stream stream) {
  OSUD_CHECK_ARG(s && model_out && x && t_index && noise && x_out && N > 0 && T > 0, "sampler_step: null/empty argument");
  OSUD_CHECK_ARG(inpaint == nullptr || (inpaint->keep != nullptr && inpaint->known != nullptr),
                 "sampler_step: in-painting needs both the keep mask and the known values");
  OSUD_TRY(sched_upload(const_cast<osud_sched*>(s)));
  return launch_sampler_step(s->d_coefs, mode, eta, model_out, x, t_index, nullptr, noise, 0, 0, N, T, cfg_scale, clip,
                             inpaint, x_out, pred_xstart, (hipStream_t)stream);
}

extern "C" int osud_sampler_step(const osud_sched* s, int mode, float eta, const float* model_out, const float* x,
                                 const int64_t* t_index, const float* noise, int N, int T, float cfg_scale, int clip,
                                 float* x_out, float* pred_xstart, osud_stream stream) {
  return osud_sampler_step_inpaint(s, mode, eta, model_out, x, t_index, noise, N, T, cfg_scale, clip, nullptr, x_out,
                                   pred_xstart, stream);
}
