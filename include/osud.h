/*
 * osud.h — C ABI of libosud.so, the MI355X-native (gfx950) DiT denoising path for
 * osu-diffusion.  Plain pointers and sizes only; no torch / C++ types cross this
 * boundary.  Every entry point returns 0 on success or an OSUD_ERR_* code;
 * osud_last_error() gives the thread-local message.  Handles are bound to the device
 * that was current at create time and are NOT thread-safe; all work is enqueued on the
 * caller-supplied stream (a hipStream_t passed as void*; NULL = the default stream), so
 * it is ordered with the caller's other work on that stream.  The caller owns every
 * I/O buffer (device memory); the library owns its handles, its packed low-precision
 * weight copies and its activation workspaces (allocated in osud_dit_reserve /
 * first use of a new shape — no allocation in steady state, so steps are hipGraph-
 * capturable).
 *
 * Reference interfaces replaced (paths relative to the osu-diffusion repository):
 *   models.py:243-273   DiT.__init__            -> osud_dit_create / osud_dit_set_param
 *   models.py:306-325   DiT.forward             -> osud_dit_forward
 *   models.py:327-343   DiT.forward_with_cfg    -> osud_dit_forward (cfg_scale >= 0)
 *   diffusion/gaussian_diffusion.py:167-211 + diffusion/respace.py:72-86
 *                       schedule tables         -> osud_sched_create
 *   diffusion/gaussian_diffusion.py:273-369,420-467  p_mean_variance + p_sample
 *                                               -> osud_sampler_step (mode P)
 *   diffusion/gaussian_diffusion.py:563-610     ddim_sample -> osud_sampler_step (mode DDIM)
 *   diffusion/gaussian_diffusion.py:514-561,686-733  *_sample_loop_progressive
 *                                               -> osud_sample_loop
 *   diffusion/gaussian_diffusion.py:785-874 (+735-783, diffusion_utils.py:9-89)
 *                       training_losses         -> osud_train_loss / osud_dit_backward
 *   train.py:243-261    optimizer + EMA step    -> osud_adamw_ema_step
 *   train.py:106-115,152,257,274  DDP init broadcast / gradient all-reduce
 *                                               -> osud_comm_init / osud_broadcast_params / osud_allreduce_grads (or
 *                                                  osud_reduce_scatter_grads + osud_allgather_params) on the flat gradient
 *                                                  arena: RCCL behind this ABI; the host only decides WHEN a slice is
 *                                                  final (osud_dit_backward_phases) -- see INTEGRATION.md
 *
 * Run-time switches: ONE table, osud_set_option / osud_get_option (below).  The library reads two environment variables and no
 * others: OSUD_OPTIONS ("name=value,..": initial values of that table, for A/B runs of unmodified scripts) and OSUD_RCCL_LIB (path
 * of librccl, resolved with dlopen).
 */
#ifndef OSUD_H
#define OSUD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OSUD_OK 0
#define OSUD_ERR_ARG 1         /* bad argument / shape (reference: assert / IndexError) */
#define OSUD_ERR_HIP 2         /* a HIP runtime call failed */
#define OSUD_ERR_STATE 3       /* missing parameter, not reserved, wrong call order */
#define OSUD_ERR_UNSUPPORTED 4 /* configuration not built (e.g. head_dim != 64 in bf16) */

/* arithmetic tiers */
#define OSUD_PREC_BF16 0 /* fast tier: bf16 MFMA operands, fp32 accumulate / residual / LN / softmax */
#define OSUD_PREC_F32 1  /* parity tier: exact-f32 MFMA (v_mfma_f32_32x32x2_f32), fp32 everywhere */
#define OSUD_PREC_FP8 2  /* inference only: the bf16 tier with the four big per-block GEMMs on OCP e4m3 operands
                            (v_mfma_scale_f32_32x32x64_f8f6f4, per-output-channel weight scales, static activation scales) */

#define OSUD_PREC_BF16X3 3 /* tolerance tier, inference only: every GEMM and the attention products on split-bf16 operands
                            (v = hi + lo, two bf16 each; three bf16 MFMAs per product: hi*hi + lo*hi + hi*lo, fp32 accumulate) --
                            16 significand bits per operand, finer than the TF32 matmuls of the reference's sampling path
                            (sample.py:25-26), at a third of the bf16 tier's MFMA rate instead of the f32 tier's sixteenth */

#define OSUD_PREC_F16F8 4 /* the tolerance tier's faster form, inference only: OSUD_PREC_BF16X3 with the four big GEMMs of every block
                            (in_proj, out_proj, fc1, fc2) on fp16 + e4m3-residual operands: v = hi (fp16) + 2^-12 lo8 (e4m3), product =
                            hi*hi on v_mfma_f32_32x32x16_f16 + both cross terms in ONE block-scaled v_mfma_scale_f32_32x32x64_f8f6f4
                            -- 15 significand bits per operand at 2/3 of the split-bf16 form's matrix-pipe passes */

#define OSUD_PREC_F16 5 /* fast tier of the sampling path, inference only: the bf16 tier's kernels on IEEE half operands
                          (v_mfma_f32_32x32x16_f16): 11 significand bits -- the precision of the TF32 matmuls the reference's own
                          sampling path uses (sample.py:25-26) -- at the bf16 tier's speed */

#define OSUD_PREC_F16W8 6 /* a faster form of the tolerance tier, inference only: OSUD_PREC_F16F8 with the ACTIVATION operand of the four big
                            GEMMs rounded to fp16 (11 bits) and only the WEIGHT carrying its e4m3 residual (15 bits): a weight's rounding
                            error repeats in every product of every step, an activation's is a fresh one each time.  Per 128 k: eight
                            v_mfma_f32_32x32x16_f16 + two block-scaled v_mfma_scale_f32_32x32x64_f8f6f4 = 3/4 of F16F8's matrix-pipe passes,
                            3 instead of 4 operand bytes per element */

#define OSUD_PREC_F16M8 7 /* the tolerance tier with a per-GEMM choice between the two operand forms above: option "f16m8_forms" (read at
                            osud_dit_create) has bit i set where GEMM i of a block (0 in_proj, 1 out_proj, 2 fc1, 3 fc2) takes fp16 activations
                            (F16W8's form) and clear where it keeps the activation's residual (F16F8's); default 11 = every big GEMM but fc1 */

typedef struct osud_dit osud_dit;
typedef struct osud_sched osud_sched;
typedef void* osud_stream; /* hipStream_t */

typedef struct osud_dit_cfg {
  int32_t hidden;      /* D: 384 / 768 / 1024 / 1152      models.py:410-423 */
  int32_t depth;       /* number of DiTBlocks */
  int32_t heads;       /* D / heads = head_dim */
  int32_t context;     /* E, rows of c (144) */
  int32_t in_channels; /* 2 */
  int32_t table_rows;  /* num_classes + 1 (null class last)   models.py:48-52 */
  int32_t learn_sigma; /* 1 -> out channels = 2 * in_channels */
  int32_t precision;   /* OSUD_PREC_* */
} osud_dit_cfg;

const char* osud_last_error(void);
/* Library/ABI version and the gfx target it was built for ("gfx950"). */
int osud_version(void);
const char* osud_build_arch(void);

/* ------------------------------------------------------------------ model */
int osud_dit_create(const osud_dit_cfg* cfg, osud_dit** out);
void osud_dit_destroy(osud_dit* m);

/* Upload one parameter under its reference state-dict key (e.g.
 * "blocks.3.attn.in_proj_weight").  `dev_f32` is fp32 device memory laid out as the
 * reference tensor (row-major, shape given); the library makes its own packed copy in the
 * handle's precision — the caller keeps the fp32 master. */
int osud_dit_set_param(osud_dit* m, const char* key, const float* dev_f32, const int64_t* shape, int ndim,
                       osud_stream stream);
/* Number of parameters still missing before forward may run (0 = ready). */
int osud_dit_missing_params(const osud_dit* m);

/* Allocate workspaces for batches up to N rows of T tokens (training != 0 also keeps the
 * per-layer activations the backward pass needs). */
int osud_dit_reserve(osud_dit* m, int max_N, int max_T, int training);

/* Noise-prediction forward.
 *   x (N,2,T) f32 channel-major; t (N) i64; o (N,T) f32 ms; c (N,E,T) f32 channel-major;
 *   y (N) i64 class index (already label-dropped in training); attn_mask (T,T) u8, 1 = masked,
 *   or NULL; out (N,4,T) f32 channel-major.
 *   cfg_scale < 0: plain forward (models.py:306-325).
 *   cfg_scale >= 0: forward_with_cfg (models.py:327-343): rows N/2.. reuse x of rows 0..N/2-1,
 *   eps channels of both halves become uncond + s*(cond - uncond), other channels untouched. */
int osud_dit_forward(osud_dit* m, const float* x, const int64_t* t, const float* o, const float* c,
                     const int64_t* y, const uint8_t* attn_mask, int N, int T, float cfg_scale, float* out,
                     osud_stream stream);

/* fp8 tier, inference: measure the activation scales of the e4m3 GEMM operands on a representative batch instead of using the
 * built-in constants (no reference counterpart: the reference has no fp8 path).  Same arguments as osud_dit_forward; the forward
 * runs in bf16 arithmetic and records, per block, the amax of the LayerNorm, attention and GELU outputs; scale = 448 / (2 amax).
 * accumulate != 0: keep the maximum over this and earlier calls (e.g. a few timesteps of the schedule). */
int osud_dit_calibrate_fp8(osud_dit* m, const float* x, const int64_t* t, const float* o, const float* c, const int64_t* y,
                           const uint8_t* attn_mask, int N, int T, float cfg_scale, int accumulate, osud_stream stream);

/* ------------------------------------------------------------------ diffusion schedule */
/* `betas` are the BASE process's betas (fp64, n_base of them); `use_timesteps` the sorted kept
 * indices (n_use).  Re-derives the spaced betas and every fp64 table exactly as
 * SpacedDiffusion/GaussianDiffusion.__init__ do. */
int osud_sched_create(const double* betas, int n_base, const int64_t* use_timesteps, int n_use, osud_sched** out);
void osud_sched_destroy(osud_sched* s);
int osud_sched_num_timesteps(const osud_sched* s);
/* Copy one fp64 host table out (for parity checks).  Names: "betas", "alphas_cumprod",
 * "alphas_cumprod_prev", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
 * "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2",
 * "log_betas", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod", "posterior_variance". */
int osud_sched_table(const osud_sched* s, const char* name, double* out, int n);
int osud_sched_timestep_map(const osud_sched* s, int64_t* out, int n);

#define OSUD_SAMPLER_P 0    /* ancestral p_sample */
#define OSUD_SAMPLER_DDIM 1 /* ddim_sample with eta */

/* One sampler update given the raw model output (N,4,T) (pre-CFG when cfg_scale >= 0):
 *   x (N,2,T) in; t_index (N) i64 step indices into the (spaced) schedule; noise (N,2,T);
 *   x_out (N,2,T) (may alias x); pred_xstart (N,2,T) or NULL.  clip != 0 clamps x0 to [-1, 2]. */
int osud_sampler_step(const osud_sched* s, int mode, float eta, const float* model_out, const float* x,
                      const int64_t* t_index, const float* noise, int N, int T, float cfg_scale, int clip,
                      float* x_out, float* pred_xstart, osud_stream stream);

/* In-painting: the reference passes `denoised_fn = lambda x0: torch.where(mask, x0, known)` to the sampling loops
 * (testing/test_toy.py:56-74; applied to the predicted x0 before the clamp, gaussian_diffusion.py:341-346).  Both arrays are
 * (N,2,T) device buffers; where keep == 0 the prediction is replaced by `known`.  A NULL `inpaint` = no in-painting. */
typedef struct osud_inpaint {
  const uint8_t* keep;
  const float* known;
} osud_inpaint;
int osud_sampler_step_inpaint(const osud_sched* s, int mode, float eta, const float* model_out, const float* x,
                              const int64_t* t_index, const float* noise, int N, int T, float cfg_scale, int clip,
                              const osud_inpaint* inpaint, float* x_out, float* pred_xstart, osud_stream stream);

/* Whole sampling loop for steps first_step, first_step-1, ..., last_step (inclusive; the full
 * loop is first = num_timesteps-1, last = 0), each step = forward (model sees
 * timestep_map[i]) + sampler update, replayed from one captured hipGraph.
 *   x (N,2,T) in/out.  noise: (n_steps, N, 2, T) f32 used in execution order, or NULL to draw
 *   N(0,1) in-kernel from Philox4x32-10 keyed by `seed` (own stream, not torch's). */
int osud_sample_loop(osud_dit* m, const osud_sched* s, int mode, float eta, float* x, const float* o, const float* c,
                     const int64_t* y, const uint8_t* attn_mask, int N, int T, float cfg_scale, int clip,
                     int first_step, int last_step, const float* noise, uint64_t seed, osud_stream stream);
/* the same loop with in-painting fused into every sampler update */
int osud_sample_loop_inpaint(osud_dit* m, const osud_sched* s, int mode, float eta, float* x, const float* o,
                             const float* c, const int64_t* y, const uint8_t* attn_mask, int N, int T, float cfg_scale,
                             int clip, int first_step, int last_step, const float* noise, uint64_t seed,
                             const osud_inpaint* inpaint, osud_stream stream);

/* `iters` sampler steps at ONE schedule index `step`, x in/out: the refine pass of sample.py:186-205 (p_sample at t = 0, refine_iters
 * times, after the weights of --refine-ckpt were loaded) without a host round trip per iteration -- the captured step of
 * osud_sample_loop replayed with its device-side step counter standing still.  noise: (iters, N, 2, T) or NULL (in-kernel Philox;
 * unused at step 0, where the sampler adds none).  inpaint may be NULL. */
int osud_sample_repeat(osud_dit* m, const osud_sched* s, int mode, float eta, float* x, const float* o, const float* c,
                       const int64_t* y, const uint8_t* attn_mask, int N, int T, float cfg_scale, int clip, int step, int iters,
                       const float* noise, uint64_t seed, const osud_inpaint* inpaint, osud_stream stream);

/* ------------------------------------------------------------------ training
 * Sizes must satisfy T % 64 == 0 and N*T % 128 == 0 (no padding rows in the gradient products). */
/* Where the backward pass writes the gradient of parameter `key` (fp32, same shape as the
 * parameter).  Backward OVERWRITES these buffers (it does not accumulate). */
int osud_dit_bind_grad(osud_dit* m, const char* key, float* grad_f32);
/* Re-pack every parameter from the fp32 master pointer last passed to osud_dit_set_param
 * (call after the optimizer changed the masters in place). */
int osud_dit_refresh(osud_dit* m, osud_stream stream);
/* The same re-pack for the parameters of phases [phase_lo, phase_hi] only (phases as in osud_dit_backward_phases: 0 = embedders and
 * conditioning path, p = 1..depth = block p - 1 with its adaLN pair, depth + 1 = final layer), e.g. on a side stream as the sharded
 * optimizer's all-gather delivers them; and a gate: the NEXT forward waits for `hip_event` (a hipEvent_t, recorded behind that
 * phase's re-pack) before the first kernel of `phase` -- so the gather / re-pack of later blocks overlaps the forward of earlier ones.
 * Gates are one-shot; NULL clears one. */
int osud_dit_refresh_phases(osud_dit* m, int phase_lo, int phase_hi, osud_stream stream);
int osud_dit_forward_gate(osud_dit* m, int phase, void* hip_event);
/* forward that keeps the per-layer activations (plain forward, no CFG, no mask) */
int osud_dit_forward_train(osud_dit* m, const float* x, const int64_t* t, const float* o, const float* c,
                           const int64_t* y, int N, int T, float* out, osud_stream stream);
/* backward of the last osud_dit_forward_train: dout (N,4,T) = dLoss/d(out) */
int osud_dit_backward(osud_dit* m, const float* dout, osud_stream stream);
/* The same backward in phases, so the host can start reducing finished gradient slices while the rest is
 * still being computed: phase 0 = final layer, phase p (1..depth) = block depth-p, phase depth+1 = first
 * linear + conditioning path (adaLN / embedder / class-table gradients).  Phases must run in order. */
int osud_dit_backward_phases(osud_dit* m, const float* dout, int phase_lo, int phase_hi, osud_stream stream);
/* x_t = sqrt(ac_t) x_0 + sqrt(1-ac_t) noise   (gaussian_diffusion.py:231-247); t = step indices */
int osud_q_sample(const osud_sched* s, const float* x_start, const int64_t* t, const float* noise, int N, int T,
                  float* x_t, osud_stream stream);
/* training_losses for EPSILON / LEARNED_RANGE (gaussian_diffusion.py:785-874): terms is (3,N) =
 * [l1|mse ; vb ; loss]; dout (N,4,T) = d(mean_n loss_n)/d(model_out).  use_l1: 1 = L1, 0 = MSE. */
int osud_train_loss(const osud_sched* s, int use_l1, const float* model_out, const float* x_start, const float* x_t,
                    const float* noise, const int64_t* t, int N, int T, float* terms, float* dout, osud_stream stream);
/* torch.optim.AdamW step (bias-corrected, decoupled weight decay) fused with the EMA update
 * (train.py:36-45,258-261) over flat fp32 arenas of n elements; elements [skip_begin, skip_end)
 * are frozen (EMA only); grads are multiplied by grad_scale first (1/world after a SUM all-reduce).
 * ema may be NULL. */
int osud_adamw_ema_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* ema, size_t n,
                        float lr, float beta1, float beta2, float eps, float weight_decay, int step, float ema_decay,
                        size_t skip_begin, size_t skip_end, float grad_scale, osud_stream stream);

/* ------------------------------------------------------------------ data-parallel exchange (RCCL over xGMI)
 * The collectives of train.py's DDP wrapper on the library's own RCCL communicator (one per process = per GPU):
 *   train.py:106  dist.init_process_group     -> osud_comm_unique_id (rank 0) + osud_comm_init (every rank)
 *   train.py:152  DDP(...) parameter broadcast -> osud_broadcast_params on the flat parameter arena
 *   train.py:257  gradient all-reduce (mean)   -> osud_allreduce_grads per finished slice of the flat gradient arena (SUM; the
 *                                                 1/world goes into osud_adamw_ema_step's grad_scale), or the sharded form
 *                                                 osud_reduce_scatter_grads -> optimizer on the own shard -> osud_allgather_params
 *   train.py:274  loss all-reduce, :297 barrier -> osud_allreduce_grads on a 1-element buffer
 * The 128-byte id is created by rank 0 and handed to every rank by the host (torchrun's store, a file, MPI).  Collectives are
 * enqueued on the given stream; overlap with the backward phases (osud_dit_backward_phases) is the host's choice of streams.
 * librccl is loaded at run time (the process's own copy, e.g. PyTorch-ROCm's): without it these calls return
 * OSUD_ERR_UNSUPPORTED and everything else works. */
#define OSUD_COMM_UID_BYTES 128
#define OSUD_WIRE_F32 0
#define OSUD_WIRE_BF16 1 /* gradients rounded to bf16 for the exchange (half the bytes; opt-in) */
typedef struct osud_comm osud_comm;
int osud_comm_unique_id(void* uid128);
int osud_comm_init(int rank, int world, const void* uid128, osud_comm** out);
void osud_comm_destroy(osud_comm* c);
int osud_comm_rank(const osud_comm* c);
int osud_comm_world(const osud_comm* c);
int osud_comm_rccl_version(void); /* 0 if librccl is not available */
int osud_allreduce_grads(osud_comm* c, void* buf, size_t n, int wire, osud_stream stream);
int osud_broadcast_params(osud_comm* c, float* buf, size_t n, int root, osud_stream stream);
int osud_reduce_scatter_grads(osud_comm* c, const void* in, void* out_shard, size_t n_per_rank, int wire, osud_stream stream);
int osud_allgather_params(osud_comm* c, const float* shard, float* full, size_t n_per_rank, osud_stream stream);

/* Row exchange of the class-table gradient (52 671 x D; only the step's label rows are non-zero, the reference's DDP reduces it
 * densely: train.py:257).  pack: labels (B, as fed to the forward) -> idx_out (B, sorted) + rows_out (B x D: the gradient row of
 * every label's first occurrence, zeros for duplicates).  The host all-gathers both; apply: all_idx (W x B) / all_rows (W x B x D)
 * in rank order -> every touched row of table_grad becomes the sum of the ranks' rows added in rank order (bit-identical on
 * every rank).  scratch: W * B + 1 int64.  At most 4096 labels over all ranks. */
int osud_table_rows_pack(const float* table_grad, int rows, int D, const int64_t* labels, int B, int64_t* idx_out, float* rows_out,
                         osud_stream stream);
int osud_table_rows_apply(float* table_grad, int rows, int D, const int64_t* all_idx, const float* all_rows, int W, int B,
                          int64_t* scratch, osud_stream stream);

/* ------------------------------------------------------------------ op-level entry points
 * (the fused building blocks, exported so each can be parity-tested on its own) */
/* Process-wide switch for multi-round GEMM launches (more output tiles than compute units): 1 = workgroups draw their tiles
 * from per-XCD ticket queues instead of a fixed stride, so that a launch does not wait for workgroups whose compute unit is
 * held by another kernel (RCCL collectives overlapped with the backward: 8 held CUs stretch a fixed-stride launch 1.5x, a
 * queued one 1.1x), and the split-K weight-gradient kernel draws K-chunks from per-tile queues; 0 = fixed schedules (0.3 %
 * faster per training step, 2 % per sampling step, when the GPU is not shared: the default).
 * GEMM results are identical either way; weight gradients differ by the order the chunks are summed in (the only place where
 * two runs of the backward pass may differ in the last bits: everything else is summed in a fixed order).  Data-parallel
 * trainers switch it on. */
int osud_set_gemm_dynamic_tiles(int on);

/* Process-wide options: every run-time choice between two built and tested forms of the same computation.  value -1 restores the
 * default.  Unknown names / out-of-range values: OSUD_ERR_ARG.
 *   name                default  values
 *   wgrad_side_stream   1        1: a block's weight gradients run on the library's side stream next to the block's data-gradient
 *                                chain (bf16 and fp8 training tiers); 0: single stream.  Same bits.
 *   sample_graph        1        1: osud_sample_loop captures one step as a hipGraph and replays it; 0: eager launches.  Same bits.
 *   embed_const         1        sampler loops: the offset / context part of the first linear is multiplied once per loop
 *   tvec_table          1        sampler loops: the timestep-embedding MLP is evaluated once per loop for every schedule index
 *   split_first         1        bf16 / fp16 tiers: first linear at fp32 accuracy ([hi | lo | hi] rows); read by osud_dit_create
 *   attn_fwd_kernel     0        0 auto; 1: never the streamed window kernels; 2: the register-staged general kernel
 *   attn_bwd_kernel     0        0 auto; 1: never the streamed kernels; 2: the tiled kernel (any T)
 *   gemm_tile           0        0 auto; 64 | 128 | 192 | 256 | 1192 (192 x 256) | 1256 (128 x 256): force a geometry where it divides
 *   f8_twins_only       1        fp8 training: tensors whose bf16 form has no reader are written as e4m3 only
 *   debug_sync          0        1: synchronise after every stage of the backward pass and name it on stderr (fault triage)
 *   f16m8_forms         11       OSUD_PREC_F16M8: bit i = 1 puts GEMM i of a block (0 in_proj, 1 out_proj, 2 fc1, 3 fc2) on fp16-activation operands;
 *                                read by osud_dit_create
 *   gemm_loop           1        main loop of the 256-row GEMM tiles (bf16, fp16, fp16 + e4m3 and e4m3 operands) and of the 256 x 256 weight-gradient kernel: 1: the phased
 *                                schedule (the two waves of a SIMD one barrier apart, counted vmcnt: csrc/gemm_phased.h); 0: one barrier per K
 *                                slab.  Same bits.
 *   gelu_code           1        bf16 / fp8 training tiers: the GELU derivative saved by the fc1 epilogue for the backward pass is an 8-bit code
 *                                (step 1 / 200, absolute error <= 2.5e-3, 0 and 1 exact; 32 x 32 blocks of 1 KiB) instead of bf16 rows: half the
 *                                bytes out of fc1's epilogue and into the fc2 data gradient's.  0: bf16 rows.  Read by osud_dit_create. */
int osud_set_option(const char* name, int value);
int osud_get_option(const char* name, int* value);

/* out[y][x] = epilogue(sum_k Y[y][k] * X[x][k]); see csrc/gemm.h for the epilogue codes.  Operand / output forms per precision: 0 bf16,
 * 1 f32, 2 e4m3 (experimental), 3 split-bf16 plane pairs [hi | lo] (ld = logical columns), 4 fp16 + e4m3 rows (osud_op_pack_h8; the
 * bias epilogue writes split-bf16 planes -- the attention kernel's input --, the bias + GELU epilogue writes fp16 + e4m3 rows). */
int osud_op_gemm(int precision, int epilogue, const void* Y, int ldy, const void* X, int ldx, int My, int Nx, int K,
                 void* out, int ldo, const float* bias, const float* gate, int ld_gate, int rows_per_sample,
                 int n_samples, osud_stream stream);
/* osud_op_gemm with the second tensor of the TRAINING epilogues (csrc/gemm.h; reference: Mlp.forward models.py:112-119 with nn.GELU(approximate="tanh")
 * :138 -- fc1 -> GELU -> fc2 -- and autograd's derivative of that GELU in train.py:257's backward): out2 = what the epilogue saves for the backward pass
 * (4 bias + GELU: the GELU derivative; 2 bias + SiLU: the pre-activation), aux = what epilogue 9 multiplies the product with (the saved GELU
 * derivative), colpart (epilogue 9, optional) = [My / 128 or My / 64][Nx] partial column sums of the output.  aux_code = 1: the derivative
 * is the 8-bit block code of option gelu_code (one byte per element; 32 x 32 blocks of 1 KiB, block (y / 32, x / 32) at ((y / 32) * (ldo / 32)
 * + x / 32) * 1024, lane l = 4 * (row & 15) + ((col & 31) >> 3) holds 16 bytes at 16 l: columns (col & ~7) .. + 7 of row (row & 15), then of row
 * 16 + (row & 15); value = (code - 26) / 200) instead of rows of the tier's element type.  bf16 and fp32 tiers (fp32: aux_code = 0). */
int osud_op_gemm_ex(int precision, int epilogue, const void* Y, int ldy, const void* X, int ldx, int My, int Nx, int K, void* out, int ldo,
                    const float* bias, void* out2, const void* aux, int aux_code, float* colpart, osud_stream stream);
/* fp16 + e4m3 operand form of OSUD_PREC_F16F8 (csrc/common.h: h8_t): src fp32 [rows][ld_src] (cols_src used, zero padded to cols_dst,
 * a multiple of 32) -> dst [rows][4 * cols_dst bytes], K-blocked groups of 32; weight = 1 for the weight flavour (planes swapped). */
int osud_op_pack_h8(const float* src, int ld_src, int cols_src, void* dst, int cols_dst, int rows, int weight, osud_stream stream);
/* fp16 x (fp16 + e4m3) operand form of OSUD_PREC_F16W8 (csrc/common.h: w8_t): dst [rows][3 * cols_dst bytes], K-blocked super-groups of
 * 128 (cols_dst % 128 == 0); weight = 1: the e4m3 plane holds the fp16 residual x 2^12, weight = 0 (an activation): e4m3(v). */
int osud_op_pack_w8(const float* src, int ld_src, int cols_src, void* dst, int cols_dst, int rows, int weight, osud_stream stream);
/* Convert n fp32 values to the tier's element type (bf16 round-to-nearest-even or f32 copy). */
int osud_op_convert(int precision, const float* src, void* dst, size_t n, osud_stream stream);
/* Attention core on the packed in_proj output qkv [Mp][ld_qkv] (Q | K | V, head h = hd columns): out [Mp][hidden]. */
int osud_op_attention(int precision, const void* qkv, int ld_qkv, const uint8_t* mask, void* out, int N, int T, int Tp,
                      int Mp, int heads, int head_dim, osud_stream stream);

/* Weight gradient of a Linear without operand transposes (bf16 tier): out[y][x] = sum_m P[m][y] * Q[m][x] over M token rows
 * (P = gradient of the Linear's output [M][ldp], Q = its input [M][ldq], both bf16 row-major; out fp32 [Ny][Nx]).  `ws` holds the
 * fp32 partial slabs of the split over the token axis (ws_elems floats; at least Ny * Nx * (CUs / tiles) for a full split). */
int osud_op_wgrad(const void* P, int ldp, const void* Q, int ldq, int Ny, int Nx, int M, float* out, float* ws, size_t ws_elems,
                  osud_stream stream);

/* The same product on OCP e4m3 operands (fp8 training, BASELINE config 5): P8 / Q8 are one byte per element, quantised with
 * per-tensor scales; out = inv_p[0] * inv_q[0] * P8^T . Q8 with inv_* the operands' 1 / scale (DEVICE scalars).  M % 128 == 0. */
int osud_op_wgrad8(const void* P8, int ldp, const void* Q8, int ldq, int Ny, int Nx, int M, float* out, float* ws, size_t ws_elems,
                   const float* inv_p, const float* inv_q, osud_stream stream);

/* Backward of the attention core in the training layout (T == Tp, T % 64 == 0, no mask): qkv [N*T][3*hidden], d_out / out [N*T][hidden],
 * lse [N][heads][T] as saved by the training forward (log2 domain in the bf16 tier, natural log in the f32 tier) -> dqkv [N*T][3*hidden].
 * delta_ws: [N][heads][T] floats of scratch, needed when a head's sequence does not fit the LDS (T > 256); may be NULL otherwise. */
int osud_op_attention_bwd(int precision, const void* qkv, const void* d_out, const void* out, const float* lse, void* dqkv,
                          int N, int T, int heads, int head_dim, float* delta_ws, osud_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* OSUD_H */
