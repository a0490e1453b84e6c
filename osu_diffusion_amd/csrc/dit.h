// The osud_dit handle: packed weights, activation workspaces, forward orchestration.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "gemm.h"
#include "kernels.h"

struct osud_sched;

namespace osud {
int sched_upload(osud_sched* s);
const float* sched_coefs(const osud_sched* s);
const int64_t* sched_tmap_dev(const osud_sched* s);
}  // namespace osud

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
};

struct BlockWeights {
  // TE, [out][in].  w_qkv is the packed in_proj ([Wq;Wk;Wv], 3D x D).
  void *w_qkv = nullptr, *w_o = nullptr, *w1 = nullptr, *w2 = nullptr;
  float *b_qkv = nullptr, *b_o = nullptr, *b1 = nullptr, *b2 = nullptr;
  // fp8 tier: e4m3 copies of the four big weights + per-output-channel de-quantisation factors (incl. the activation scale)
  void *w8_qkv = nullptr, *w8_o = nullptr, *w8_1 = nullptr, *w8_2 = nullptr;
  float *dq_qkv = nullptr, *dq_o = nullptr, *dq_1 = nullptr, *dq_2 = nullptr;
  // transposed copies ([in][out]) for the data-gradient products (training only)
  void *w_qkv_t = nullptr, *w_o_t = nullptr, *w1_t = nullptr, *w2_t = nullptr;
  // fp8 training: e4m3 copies of the transposed weights (quantised per row = per output column of the data-gradient product)
  void *w_qkv_t8 = nullptr, *w1_t8 = nullptr, *w2_t8 = nullptr, *w_o_t8 = nullptr;
  float *dq_qkv_t = nullptr, *dq_1_t = nullptr, *dq_2_t = nullptr, *dq_o_t = nullptr;
};

// per-layer activations kept for the backward pass (training only)
struct LayerSaved {
  float* h_in = nullptr;     // [Mp][D] residual stream entering the block
  float* h_mid = nullptr;    // [Mp][D] after the attention branch
  float* stats1 = nullptr;   // [Mp][2] mean, rstd of LN1
  float* stats2 = nullptr;   // [Mp][2]
  void *u1 = nullptr, *qk = nullptr /* [Mp][3D] q|k|v */, *ao = nullptr, *u2 = nullptr, *z1 = nullptr /* gelu'(fc1 pre-activation) */,
       *g = nullptr, *br1 = nullptr /* attention branch output */, *br2 = nullptr /* MLP branch output */;
  float* lse = nullptr;  // [N][H][Tp]
  // fp8 training: the e4m3 twins of this layer's GEMM inputs, kept for the weight-gradient products of the backward pass
  // (u1 -> in_proj, ao -> out_proj, u2 -> fc1, g -> fc2); each quantised with its slot's delayed scale of THIS step
  void *u1_8 = nullptr, *ao_8 = nullptr, *u2_8 = nullptr, *g_8 = nullptr;
};
// fp8 training: quantised tensors (= delayed-scale slots) per block:
//   0 u1 (LN1 out)  1 u2 (LN2 out)  2 g (GELU out)  3 d(MLP branch)  4 d(fc1 pre-activation)  5 dqkv  6 ao (attention out)  7 d(attention branch)
constexpr int kF8Slots = 8;

// workspaces of the backward pass
struct BwdWs {
  float *dhA = nullptr, *dhB = nullptr, *du = nullptr, *dada = nullptr, *dWada = nullptr, *dbada = nullptr, *dsb = nullptr,
        *db = nullptr, *dth = nullptr, *dWe = nullptr, *splitk = nullptr, *attn_delta = nullptr /* [N][H][Tp] */;
  size_t splitk_elems = 0;
  // deterministic reductions (kernels.h): per-workgroup partial rows of the LayerNorm / gate / final-layer backward kernels, one slot
  // per launch of a backward pass ([2 L + 2][M / 64][6 D + 64]), and scratch for the row blocks' shares of a column sum (main / side stream)
  float *rowpart = nullptr, *colpart = nullptr, *colpart2 = nullptr;
  size_t colpart_elems = 0;
  // per-layer partial rows of the two bias gradients that ride in epilogues (fc1: one row per wave-row of the data-gradient GEMM's grid;
  // in_proj: one row per sample from the streamed attention backward), summed by ONE launch_colsum_many per backward call
  float *b1part = nullptr, *bqkvpart = nullptr;
  size_t b1part_stride = 0, bqkvpart_stride = 0;  // floats per layer
  // bf16 tier: the weight gradients of a block run on a side stream next to the block's data-gradient chain (train.hip): their own
  // split-K slab area, the stream, and the events that hand the operands over {fc2 dgrad done, LN2 backward + out_proj dgrad done,
  // attention backward done, side stream drained}
  float* splitk2 = nullptr;
  hipStream_t side = nullptr;
  hipEvent_t side_ev[4] = {nullptr, nullptr, nullptr, nullptr};  // [0..2] chain -> side, [3] side -> chain
  bool side_busy = false;  // weight gradients enqueued on the side stream since the chain last joined it
  float** seg_tbl = nullptr;       // device: the per-block adaLN weight-gradient tensors (GemmP::seg_out), 32 entries
  float* seg_tbl_host[32] = {};    // what the device table holds (uploaded again when a gradient tensor is re-bound)
  void* dbr2 = nullptr;  // d(attention branch output): its own buffer, so that the MLP branch's `dbr` lives to the end of the block's
                         // phase (all four weight gradients of a block are formed there in one grouped launch)
  void *dbr = nullptr, *dz1 = nullptr, *dqkv = nullptr, *dao = nullptr, *tA = nullptr, *tB = nullptr, *dada_te = nullptr,
       *db_te = nullptr, *dz0 = nullptr, *small_t1 = nullptr, *small_t2 = nullptr, *sb_t = nullptr /* silu(b)^T [D][Np] */;
};

struct GraphKey {
  int N, T, mode, clip, has_mask, has_noise;
  float cfg, eta;
  const void *o, *c, *y, *mask, *x, *noise;
  const void* sched;
  const void *keep = nullptr, *known = nullptr;
  int embed_const = 0;  // the captured step used the split-off constant part of the first linear
  unsigned opt_epoch = 0;  // the process-wide option table (gemm.h: opt_epoch) picks kernels inside the captured step: a change re-captures
  bool operator==(const GraphKey& r) const {
    return embed_const == r.embed_const && opt_epoch == r.opt_epoch && N == r.N && T == r.T && mode == r.mode && clip == r.clip && has_mask == r.has_mask &&
           has_noise == r.has_noise && cfg == r.cfg && eta == r.eta && o == r.o && c == r.c && y == r.y &&
           mask == r.mask && x == r.x && noise == r.noise && sched == r.sched && keep == r.keep && known == r.known;
  }
};

struct osud_dit {
  osud_dit_cfg cfg{};
  int D = 0, L = 0, H = 0, hd = 0, E = 0, C = 0, C2 = 0, Kp = 0, prec = 0, esz = 0, ada_cols = 0;
  int Ke = 0;                // row length of e0 / w_e: Kp, or 3 * Kp in the split form of the bf16 tier's first linear
  bool split_first = false;
  bool z1_code = false;  // the saved GELU derivative (saved[l].z1) is the 8-bit block code of common.h (option gelu_code at create; bf16 / fp8 training tiers)
  int device = -1;
  bool training = false;
  bool h8 = false;   // OSUD_PREC_F16F8: prec == BF16X3 (x3 set), and the four big GEMMs of every block on fp16 + e4m3 operands (h8_t): their
                     // weights, the LayerNorm / attention / GELU outputs that feed them
  bool w8 = false;   // OSUD_PREC_F16W8: as h8, with the activation operand of those GEMMs rounded to fp16 (w8_t: fp16 + ONE e4m3 plane)
  // the operand form of each of a block's four big GEMMs {in_proj, out_proj, fc1, fc2} (and of the kernel that writes its activation
  // operand: LN1, attention, LN2, fc1's GELU epilogue): the handle's prec, OSUD_PREC_F16F8 or OSUD_PREC_F16W8.  OSUD_PREC_F16F8 / _F16W8
  // set all four alike; OSUD_PREC_F16M8 takes them from the option "f16m8_forms" at create time
  int bform[4] = {0, 0, 0, 0};
  bool x3 = false;   // OSUD_PREC_BF16X3: prec == BF16X3, every TE matrix a plane pair [hi | lo] (common.h); inference only
  bool fp8 = false;  // OSUD_PREC_FP8: prec == BF16 everywhere except the e4m3 operands of qkv / out_proj / fc1 / fc2

  // weights
  void* w_e = nullptr;  float* b_e = nullptr;
  // sampler loops (o, c fixed over the steps; split first linear only): the offset / context part of the first linear is made once
  // per loop (h0c, without the bias) and every step multiplies only the 256 coordinate features (w_ex = columns 0..255 of the weight
  // in the split form) and adds it: out = h0c + 1 * (acc + bias) through the gated-residual epilogue with a gate of ones
  void* w_ex = nullptr;
  float* h0c = nullptr;
  float* ones_d = nullptr;
  bool embed_const_on = false;
  // ... and the timestep-embedding MLP (models.py:29-36) once per loop for every schedule index (tv_all [steps][D]); a step's
  // conditioning kernel reads its row through the per-row schedule index the step-begin kernel writes
  float* tv_all = nullptr;
  void *temb_all = nullptr, *th_all = nullptr;
  int tv_cap = 0;  // rows the three buffers hold
  bool tvec_table_on = false;
  void* w_t0 = nullptr; float* b_t0 = nullptr;
  void* w_t2 = nullptr; float* b_t2 = nullptr;
  float* table = nullptr;
  const float* table_ref = nullptr;  // what the forward reads: `table`, or -- after osud_dit_refresh -- the caller's fp32 master itself
                                     // (the class table is 162 MB at 52 670 classes: no second copy per optimizer step)
  std::vector<BlockWeights> blk;
  void* w_ada = nullptr; float* b_ada = nullptr;
  float* w_f = nullptr;  float* b_f = nullptr;
  float* freqs64 = nullptr; float* freqs128 = nullptr;
  float pf[2] = {512.f, 384.f};
  std::map<std::string, bool> have;
  std::map<std::string, const float*> master;  // caller's fp32 parameter (as last passed to set_param)
  std::map<std::string, float*> grad;          // caller's fp32 gradient buffer (osud_dit_bind_grad)
  void *w_ada_t = nullptr, *w_t2_t = nullptr;  // transposed copies (training)
  // osud_dit_refresh re-packs every parameter: while `defer` is set, set_param records its copies / conversions
  // here instead of launching them one by one, and refresh sends each list as one batched launch (batch.hip)
  osud::SegBatch* defer_copy = nullptr;
  osud::SegBatch* defer_convert = nullptr;
  osud::QuantBatch* defer_quant = nullptr;  // fp8: the per-row e4m3 forms of the refreshed weights, one launch per list
  bool transposed_ready = false;
  BwdWs bw;
  int bw_dh_cur = 0;  // which residual-gradient buffer currently holds d(loss)/d(h) (phased backward)
  void* z0 = nullptr;  // [Np][D] TE: TimestepEmbedder pre-activation
  // last training forward (inputs needed again by the backward pass)
  const int64_t* last_y = nullptr;
  int last_N = 0, last_T = 0;
  std::vector<void*> owned;  // everything hipMalloc'ed by this handle

  // workspaces (reserve)
  int cap_N = 0, cap_T = 0, cap_Mp = 0, cap_Np = 0, cap_Tp = 0;
  void *u8 = nullptr, *ao8 = nullptr, *g8 = nullptr;  // fp8 tier: e4m3 activations [Mp][D], [Mp][D], [Mp][4D]
  // fp8 TRAINING (delayed per-tensor scaling): per block 6 quantised tensors (u1, u2, gelu out, d(mlp branch), d(fc1 pre-act),
  // dqkv), each with a slot {scale in use, 1/scale, amax seen this step, -}; e4m3 staging buffers [Mp][D] and [Mp][4D]
  float* f8_slots = nullptr;
  float* f8_parts = nullptr;  // [slots][f8_amax_parts()] per-workgroup partial maxima of producers that do not use the slot's atomic word
  void *q8a = nullptr, *q8b = nullptr, *q8c = nullptr;  // q8c [Mp][D]: d(attention branch)
  // fp8 INFERENCE: per-block activation scales {LN1 out, attention out, LN2 out, GELU out}; defaults are the static constants, 
  // osud_dit_calibrate_fp8 replaces them by 448 / (2 * amax) measured on the caller's batch
  std::vector<float> f8_inf;
  bool f8_calibrating = false;
  float* f8_cal_slots = nullptr;  // [L][4] slots (device), used during calibration only
  int f8_steps = 0;  // training forwards so far: the first one runs its GEMMs in bf16 and only records the amax history
  void *e0 = nullptr, *u = nullptr, *qk = nullptr /* [Mp][3D] q|k|v */, *ao = nullptr, *g = nullptr;
  float *h = nullptr, *tvec = nullptr, *bvec = nullptr, *ada = nullptr, *out_ws = nullptr;
  void *temb = nullptr, *th = nullptr, *sb = nullptr;
  int64_t *t_model = nullptr, *t_index = nullptr;
  int* step_state = nullptr;
  uint8_t* kb_class = nullptr;  // [cap_Tp / 64][cap_Tp / 64] tile classes of the current attention mask (banded long-sequence sampling)
  std::vector<LayerSaved> saved;
  std::vector<void*> ws_owned;

  // osud_dit_forward_gate: events the NEXT forward waits for before the first kernel of a phase (0 = embedders / conditioning,
  // 1..L = block p - 1, L + 1 = final layer); one-shot.  With any gate set the adaLN product runs per phase instead of batched.
  std::vector<hipEvent_t> gate_ev;

  // sample-loop graph cache
  hipStream_t cap_stream = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  GraphKey graph_key{};
  bool graph_valid = false;
};

namespace osud {

template <typename P> inline int dev_alloc(std::vector<void*>& owner, P** out, size_t bytes, bool zero = true) {
  void* p = nullptr;
  // 512 bytes of slack behind every buffer: the weight-gradient kernel's edge tiles (feature counts that are odd multiples of 128,
  // e.g. DiT-XL's 1152 / 3456) stage up to 128 features past a row's end -- the next row's first bytes, and for the LAST token
  // row up to 256 bytes past the buffer; the slack keeps that read inside the allocation (the values are never stored)
  bytes = (bytes ? bytes : 16) + 512;
  OSUD_HIP(hipMalloc(&p, bytes));
  if (zero) OSUD_HIP(hipMemset(p, 0, bytes));
  owner.push_back(p);
  *out = reinterpret_cast<P*>(p);
  return OSUD_OK;
}

inline int gemm(osud_dit* m, int epi, const void* Y, int ldy, const void* X, int ldx, int My, int Nx, int K, void* out,
                int ldo, const float* bias, hipStream_t st, const float* gate = nullptr, int ld_gate = 0, int Tp = 0,
                int N = 0, void* out2 = nullptr, const float* res = nullptr, const void* aux = nullptr, int prec = -1) {
  GemmP p{};
  p.Y = Y; p.X = X; p.ldy = ldy; p.ldx = ldx; p.My = My; p.Nx = Nx; p.K = K;
  p.out = out; p.out2 = out2; p.ldo = ldo; p.bias = bias; p.gate = gate; p.ld_gate = ld_gate;
  p.rows_per_sample = Tp; p.n_samples = N; p.res = res; p.aux = aux;
  p.aux_code = (m->z1_code && ((epi == EPI_BIAS_GELU_TE && out2 != nullptr) || (epi == EPI_GELUGRAD_TE && aux != nullptr))) ? 1 : 0;
  return launch_gemm(prec >= 0 ? prec : m->prec, epi, p, st);
}
// one of the four big GEMMs of a block (in_proj, out_proj, fc1, fc2): OSUD_PREC_F16F8 runs these -- and only these -- on fp16 + e4m3
// operands (m->h8; everything else of that tier is the split-bf16 tier)
inline int gemm_blk(osud_dit* m, int which /* 0 in_proj, 1 out_proj, 2 fc1, 3 fc2 */, int epi, const void* Y, int ldy, const void* X, int ldx, int My,
                    int Nx, int K, void* out, int ldo, const float* bias, hipStream_t st, const float* gate = nullptr, int ld_gate = 0, int Tp = 0,
                    int N = 0) {
  return gemm(m, epi, Y, ldy, X, ldx, My, Nx, K, out, ldo, bias, st, gate, ld_gate, Tp, N, nullptr, nullptr, nullptr, m->bform[which]);
}

// GEMM on e4m3 operands (fp8 tier): Y [My][K] and X [Nx][K] are fp8, `dequant` the per-column factors, out per epilogue
// `dequant` = the weight's per-output-channel factors; the activation's factor is a host scalar (static scale, inference) or a
// device scalar (dynamic scale, training)
inline int gemm8(osud_dit* m, int epi, const void* Y, const void* X, int My, int Nx, int K, void* out, int ldo, const float* bias,
                 const float* dequant, float out_scale, hipStream_t st, const float* gate = nullptr, int ld_gate = 0, int Tp = 0,
                 int N = 0, float act_inv_host = 0.f, const float* act_inv_dev = nullptr, void* out2 = nullptr,
                 const void* aux = nullptr, float* colpart = nullptr, int* colpart_rows = nullptr, void* out8 = nullptr,
                 float* out8_slot = nullptr) {
  GemmP p{};
  p.Y = Y; p.X = X; p.ldy = K; p.ldx = K; p.My = My; p.Nx = Nx; p.K = K;
  p.out = out; p.ldo = ldo; p.bias = bias; p.gate = gate; p.ld_gate = ld_gate;
  p.rows_per_sample = Tp; p.n_samples = N; p.colscale = dequant; p.out_scale = out_scale;
  p.act_inv_host = act_inv_host; p.act_inv = act_inv_dev; p.out2 = out2; p.aux = aux; p.colpart = colpart; p.colpart_rows = colpart_rows;
  p.out8 = out8; p.out8_slot = out8_slot;
  p.aux_code = (m->z1_code && ((epi == EPI_BIAS_GELU_BF && out2 != nullptr) || (epi == EPI_GELUGRAD_TE && aux != nullptr))) ? 1 : 0;
  return launch_gemm(OSUD_PREC_FP8, epi, p, st);
}

// fp8 training, live steps (`live`: this step's GEMMs run on e4m3 operands): true when EVERY consumer of the block's GEMM operands
// -- forward products, data gradients and weight gradients (train.hip: the `weight_grad8` conditions) -- reads their e4m3 twins, so
// that the bf16 forms of u1 / u2 / gelu(z1) / dz1 / the branch gradients are not written at all (0.9 GB per DiT-XL block and step).
// osud_set_option("f8_twins_only", 0) keeps them (a test holds the two settings to 1e-7).
inline bool f8_twins_only(const osud_dit* m, bool live, int Mp) {
  const bool off = opt(OPT_F8_TWINS_ONLY) == 0;  // (read per call: a test switches it inside one process)
  return !off && live && m->prec == OSUD_PREC_BF16 && Mp % 128 == 0;
}

int dit_forward_impl(osud_dit* m, const float* x, const int64_t* t, const float* o, const float* c, const int64_t* y,
                     const uint8_t* mask, int N, int T, float cfg_scale, bool combine_cfg, float* out, bool save,
                     hipStream_t st);
int dit_ensure_ws(osud_dit* m, int N, int T, bool training);
}
