// MFMA tile GEMM for gfx950:  out[y][x] = epilogue( sum_k Y[y][k] * X[x][k] )
//
// Both operands are "row operands" with the contraction index contiguous (PyTorch Linear
// weights are (out,in) = exactly that), so the forward (Y = activations, X = weights), the
// transposed V projection (Y = W_v, X = activations -> V^T) and the backward dgrad/wgrad
// products all run through this one kernel with the roles of Y and X chosen by the caller.
//
// Tile 128(y) x 128(x) per 256-thread workgroup, 4 waves as 2x2, each wave 64x64 = 2x2
// MFMA 32x32 tiles.  K is walked in 128-BYTE slabs (64 bf16 or 32 f32) so the two tiers
// share every address computation: a 16-byte chunk is one MFMA operand fragment in both
// (bf16: v_mfma_f32_32x32x16_bf16 once; f32: v_mfma_f32_32x32x2_f32 four times, exact f32).
// Slabs go HBM -> LDS with global_load_lds_dwordx4 (no VGPR round trip), double buffered,
// one barrier per slab.  The LDS image is [row][8 chunks] with the chunk index XOR-swizzled
// by (row>>1)&7 on the SOURCE address (the LDS side of an LDS-DMA is lane-linear), which
// makes the ds_read_b128 fragment reads bank-conflict free for 128-byte rows.
// The MFMA is issued as mfma(Xfrag, Yfrag): D[i=x][j=y], so each lane ends up with 4
// consecutive x for one y -> 8/16-byte row-major stores.
#pragma once
#include "common.h"

namespace osud {

enum GemmEpilogue {
  EPI_BIAS_F32 = 0,       // out f32 = acc + bias[x]
  EPI_BIAS_TE = 1,        // out TE  = acc + bias[x]
  EPI_BIAS_SILU_TE = 2,   // out TE  = silu(acc + bias[x]); out2 TE (optional) = acc + bias[x]
  EPI_ROWBIAS_TE = 3,     // out TE  = acc + bias[y]            (transposed products)
  EPI_BIAS_GELU_TE = 4,   // out TE  = gelu_tanh(acc + bias[x]); out2 TE (optional) = gelu_tanh'(acc + bias[x])
  EPI_GATE_RES = 5,       // out f32 = res + gate[sample(y)][x] * (acc + bias[x]); out2 TE (optional) = acc + bias[x]
  EPI_NONE_F32 = 6,       // out f32 = acc
  EPI_NONE_TE = 7,        // out TE  = acc
  EPI_ACCUM_F32 = 8,      // out f32 += acc                      (gradient accumulation)
  EPI_GELUGRAD_TE = 9,    // out TE  = acc * aux[y][x], aux = the saved gelu' of fc1's pre-activation (dgrad through it)
  EPI_BIAS_GELU_BF = 10,  // fp8 operands only (fp8 training): EPI_BIAS_GELU_TE with bf16 outputs: out = gelu, out2 (optional) = gelu'
  EPI_BIAS_GELU_ALT = 11, // h8_t / w8_t operands only: EPI_BIAS_GELU_TE with the output rows in the OTHER K-blocked form (h8_t operands ->
                          // w8_t activation rows and vice versa): fc1 and fc2 of the mixed tolerance tier need not share an operand form
  EPI_COUNT
};

struct GemmP {
  const void* Y;
  const void* X;
  int ldy, ldx;  // elements
  int My, Nx, K; // My, Nx multiples of 128; K multiple of the 128-byte slab
  void* out;
  void* out2;
  int ldo;
  const float* bias;
  const float* gate;
  int ld_gate;
  int rows_per_sample;  // Tp
  int n_samples;
  const void* aux;   // TE [My][ldo] (EPI_GELUGRAD_TE: saved GELU derivative)
  int aux_code;      // 1 (bf16-typed outputs only): the saved GELU derivative -- out2 of the GELU epilogues, aux of EPI_GELUGRAD_TE -- is the 8-bit
                     // block code of common.h (gelu_code: one byte per element, 32 x 32 blocks of 1 KiB in the epilogue's lane order) instead of TE rows
  const float* res;  // EPI_GATE_RES: residual input [My][ldo]; nullptr = update `out` in place
  const float* colscale;  // fp8 operands only: out = epilogue(acc * colscale[x]) -- the product of the activation and per-output-channel
                          // weight de-quantisation factors
  float act_inv_host;     // fp8 operands: host scalar multiplied into colscale (1 / a static activation scale); 0 = none
  const float* act_inv;   // fp8 operands, optional: DEVICE scalar multiplied into colscale (1 / the activation's dynamic quantisation scale)
  void* out8;             // fp8 operands, EPI_BIAS_GELU_BF / EPI_GELUGRAD_TE, optional: e4m3 twin of `out` [My][ldo], quantised with the
  float* out8_slot;       //   device slot {scale, 1/scale, amax}: value * slot[0], and slot[2] = max(slot[2], max|value|) (fp8 training)
  float out_scale;        // fp8 OUTPUT (EPI_BIAS_GELU_TE with fp8 operands): the value is multiplied by this before quantisation
  int split_k;       // > 1: blockIdx.y walks K in split_k equal ranges, range s writes out + s * split_stride (elements)
  size_t split_stride;
  float* colpart;     // EPI_GELUGRAD_TE, optional: f32 [R][Nx] partial column sums of the OUTPUT (before rounding), one row per wave-row
                      // of the grid (R = My / rows per wave, reported through colpart_rows); summed over R they are the bias gradient
  int* colpart_rows;  // host pointer, written at launch
  int seg_rows;           // EPI_NONE_F32, optional (> 0): output row y goes to seg_out[y / seg_rows] + (y % seg_rows) * ldo instead of `out`
  float* const* seg_out;  //   (DEVICE table) -- one product whose row panels land in different tensors: the adaLN weight gradients of all blocks
  unsigned* sched;    // set by the launcher for multi-round launches: {ticket, done} counters of the dynamic tile queue
  int tile_order;  // 2: banded tile order (TileMap); experiments only otherwise (-DOSUD_GEMM_TIMING builds: flag bits of the timing variants)
};

int launch_gemm(int prec, int epi, const GemmP& p, hipStream_t st);
int gemm_num_cus();                     // compute units of the current device
bool gemm_dynamic_tiles_on();            // the effective setting
void gemm_set_dynamic_tiles(int on);  // 1 / 0 (default off; -1 = back to the default)

// Process-wide options (include/osud.h: osud_set_option).  Everything that selects between two BUILT AND TESTED forms of the same
// computation lives here -- one table, one setter, one getter; there are no other run-time switches in the library.
enum Opt {
  OPT_WGRAD_SIDE_STREAM = 0,  // 1: a block's weight gradients on the library's side stream next to its data-gradient chain (train.hip)
  OPT_SAMPLE_GRAPH,           // 1: osud_sample_loop replays one captured hipGraph per step; 0: eager launches (same bits)
  OPT_EMBED_CONST,            // 1: sampler loops multiply the offset / context part of the first linear once per loop
  OPT_TVEC_TABLE,             // 1: sampler loops make the timestep-embedding MLP once per loop for every schedule index
  OPT_SPLIT_FIRST,            // 1: bf16 / fp16 tiers keep the first linear at fp32 accuracy ([hi | lo | hi] rows); read at osud_dit_create
  OPT_ATTN_FWD_KERNEL,        // 0 auto; 1: never the streamed T = 128 kernel; 2: neither that nor the LDS-DMA kernel (register-staged)
  OPT_ATTN_BWD_KERNEL,        // 0 auto; 1: never the streamed kernels (one workgroup per head / tiled); 2: the tiled kernel
  OPT_GEMM_TILE,              // 0 auto; 64 / 128 / 192 / 256 / 1192 (192 x 256) / 1256 (128 x 256): force a tile geometry where it divides
  OPT_F8_TWINS_ONLY,          // 1: fp8 training writes only the e4m3 forms of tensors whose bf16 forms have no reader
  OPT_DEBUG_SYNC,             // 1: synchronise and name every stage of the backward pass (fault triage)
  OPT_F16M8_FORMS,            // OSUD_PREC_F16M8: bit i = 1 puts GEMM i of a block (0 in_proj, 1 out_proj, 2 fc1, 3 fc2) on fp16-activation operands (w8_t),
                              // 0 on fp16 + e4m3 operands (h8_t); read by osud_dit_create
  OPT_GEMM_LOOP,              // 1: the 256-row GEMM tiles (bf16 / fp16 / fp16 + e4m3 / e4m3 operands) and the 256 x 256 weight-gradient kernel run the
                              // phased main loop (gemm_phased.h: the two waves of a SIMD one barrier apart); 0: one barrier per K slab.  Same bits.
  OPT_GELU_CODE,              // 1: bf16 / fp8 training tiers save the GELU derivative for the backward pass as an 8-bit code (step 1 / 200 over [-0.13, 1.145]:
                              // absolute error <= 2.5e-3, 0 and 1 exact) in 32 x 32 blocks: half the bytes of the bf16 rows.  Read by osud_dit_create.
  OPT_COUNT
};
int opt(Opt o);
unsigned opt_epoch();  // bumped by every successful opt_set: whatever caches a kernel choice (a captured sampler step) keys on it
int opt_set(const char* name, int value);   // OSUD_OK / OSUD_ERR_ARG (unknown name or value out of range)
int opt_get(const char* name, int* value);
// one 16-word counter set {[0] tickets, [8] finished workgroups} of the same pool, for other persistent kernels that draw their
// work items from a queue while the GPU is shared (the attention kernels); zero on entry, re-armed by the kernel's last workgroup
unsigned* gemm_ticket_slot();
int gemm_sched_init();  // allocates the tile-queue counters (call outside stream capture; launch_gemm does it lazily otherwise)
// out[i] = sum_s part[s * stride + i], i < n (n % 4 == 0): deterministic split-K combine
int launch_splitk_reduce(const float* part, int splits, size_t stride, float* out, size_t n, hipStream_t st);

}  // namespace osud
