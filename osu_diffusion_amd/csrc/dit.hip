// osud_dit: parameter packing, workspaces, the forward pass and the graph-replayed sampling
// loop.  Reference behaviour: models.py (DiT.forward / forward_with_cfg) and
// diffusion/gaussian_diffusion.py (p_sample_loop_progressive / ddim_sample_loop_progressive).
#include "dit.h"

#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

namespace osud {

static thread_local std::string g_err;
void set_error(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
}
int hip_fail(hipError_t e, const char* what, const char* file, int line) {
  set_error("HIP error %d (%s) in `%s` at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
  return OSUD_ERR_HIP;
}

int dit_ensure_ws(osud_dit* m, int N, int T, bool training) {
  // the final layer's backward writes its weight / bias gradient through a fixed-order column pass built for 4 output channels
  // (learn_sigma = True: what every script of the reference trains, models.py:243-254); learn_sigma = False runs forward / sampling only.
  // Checked HERE so that every training entry point (reserve, forward_train) rejects such a handle before anything is allocated or run.
  OSUD_CHECK_ARG(!training || m->C2 == 4, "the backward pass is built for learn_sigma=True (4 output channels), this handle has %d", m->C2);
  OSUD_TRY(gemm_sched_init());  // tile-queue counters: must exist before any launch is captured into a graph
  if (N <= m->cap_N && T <= m->cap_T && (!training || m->training)) return OSUD_OK;
  // grow: free the old set, allocate for the max of old/new
  for (void* p : m->ws_owned) (void)hipFree(p);
  m->ws_owned.clear();
  m->saved.clear();
  m->graph_valid = false;
  const int nN = N > m->cap_N ? N : m->cap_N, nT = T > m->cap_T ? T : m->cap_T;
  const int Tp = round_up(nT, 64), Mp = round_up(nN * Tp, 128), Np = round_up(nN, 128);
  const size_t es = m->esz, D = m->D;
  auto& W = m->ws_owned;
  OSUD_CHECK_ARG(!(training && m->x3), "reserve: the split-bf16 tier is inference only (train in bf16 or fp32)");
  OSUD_CHECK_ARG(!(training && m->prec == OSUD_PREC_F16), "reserve: the fp16 tier is inference only (train in bf16 or fp32)");
  OSUD_TRY(dev_alloc(W, &m->e0, (size_t)Mp * m->Ke * es));
  if (m->split_first || m->x3) OSUD_TRY(dev_alloc(W, &m->h0c, (size_t)Mp * D * 4));
  OSUD_TRY(dev_alloc(W, &m->temb, (size_t)Np * 256 * es));
  OSUD_TRY(dev_alloc(W, &m->th, (size_t)Np * D * es));
  OSUD_TRY(dev_alloc(W, &m->sb, (size_t)Np * D * es));
  OSUD_TRY(dev_alloc(W, &m->tvec, (size_t)Np * D * 4));
  OSUD_TRY(dev_alloc(W, &m->bvec, (size_t)Np * D * 4));
  OSUD_TRY(dev_alloc(W, &m->ada, (size_t)Np * m->ada_cols * 4));
  OSUD_TRY(dev_alloc(W, &m->out_ws, (size_t)nN * m->C2 * nT * 4));
  OSUD_TRY(dev_alloc(W, &m->t_model, (size_t)nN * 8));
  OSUD_TRY(dev_alloc(W, &m->t_index, (size_t)nN * 8));
  OSUD_TRY(dev_alloc(W, &m->step_state, 32));
  OSUD_TRY(dev_alloc(W, &m->kb_class, (size_t)(Tp / 64) * (Tp / 64)));
  const bool tr = training || m->training;
  if (!tr) {
    OSUD_TRY(dev_alloc(W, &m->h, (size_t)Mp * D * 4));
    OSUD_TRY(dev_alloc(W, &m->u, (size_t)Mp * D * es));
    OSUD_TRY(dev_alloc(W, &m->qk, (size_t)Mp * 3 * D * es));
    OSUD_TRY(dev_alloc(W, &m->ao, (size_t)Mp * D * es));
    OSUD_TRY(dev_alloc(W, &m->g, (size_t)Mp * 4 * D * es));
    if (m->fp8) {
      OSUD_TRY(dev_alloc(W, &m->u8, (size_t)Mp * D));
      OSUD_TRY(dev_alloc(W, &m->ao8, (size_t)Mp * D));
      OSUD_TRY(dev_alloc(W, &m->g8, (size_t)Mp * 4 * D));
    }
  } else {
    m->saved.resize((size_t)m->L + 1);
    for (int l = 0; l <= m->L; ++l) {
      LayerSaved& s = m->saved[(size_t)l];
      OSUD_TRY(dev_alloc(W, &s.h_in, (size_t)Mp * D * 4));  // saved[L].h_in = output of the last block
      if (l == m->L) {
        OSUD_TRY(dev_alloc(W, &s.stats1, (size_t)Mp * 2 * 4));  // final-layer LN statistics
        break;
      }
      OSUD_TRY(dev_alloc(W, &s.h_mid, (size_t)Mp * D * 4));
      OSUD_TRY(dev_alloc(W, &s.stats1, (size_t)Mp * 2 * 4));
      OSUD_TRY(dev_alloc(W, &s.stats2, (size_t)Mp * 2 * 4));
      OSUD_TRY(dev_alloc(W, &s.u1, (size_t)Mp * D * es));
      OSUD_TRY(dev_alloc(W, &s.qk, (size_t)Mp * 3 * D * es));
      OSUD_TRY(dev_alloc(W, &s.ao, (size_t)Mp * D * es));
      OSUD_TRY(dev_alloc(W, &s.u2, (size_t)Mp * D * es));
      OSUD_TRY(dev_alloc(W, &s.z1, (size_t)Mp * 4 * D * (m->z1_code ? 1 : es)));  // (Mp % 128 == 0: whole 32 x 32 code blocks)
      OSUD_TRY(dev_alloc(W, &s.g, (size_t)Mp * 4 * D * es));
      OSUD_TRY(dev_alloc(W, &s.br1, (size_t)Mp * D * es));
      OSUD_TRY(dev_alloc(W, &s.br2, (size_t)Mp * D * es));
      OSUD_TRY(dev_alloc(W, &s.lse, (size_t)nN * m->H * Tp * 4));
      if (m->fp8) {
        OSUD_TRY(dev_alloc(W, &s.u1_8, (size_t)Mp * D));
        OSUD_TRY(dev_alloc(W, &s.ao_8, (size_t)Mp * D));
        OSUD_TRY(dev_alloc(W, &s.u2_8, (size_t)Mp * D));
        OSUD_TRY(dev_alloc(W, &s.g_8, (size_t)Mp * 4 * D));
      }
    }
    m->h = m->saved[0].h_in;
    m->training = true;
    if (m->fp8) {  // fp8 training: e4m3 staging of the GEMM operands + the scale slots (scale 1 until a history exists)
      OSUD_TRY(dev_alloc(W, &m->q8a, (size_t)Mp * D));
      OSUD_TRY(dev_alloc(W, &m->q8b, (size_t)Mp * 4 * D));
      OSUD_TRY(dev_alloc(W, &m->q8c, (size_t)Mp * D));
      OSUD_TRY(dev_alloc(W, &m->f8_slots, (size_t)m->L * kF8Slots * 4 * sizeof(float)));
      OSUD_TRY(dev_alloc(W, &m->f8_parts, (size_t)m->L * kF8Slots * f8_amax_parts() * sizeof(float)));  // (zeroed)
      std::vector<float> init((size_t)m->L * kF8Slots * 4, 0.f);
      for (size_t i = 0; i < init.size(); i += 4) init[i] = init[i + 1] = 1.0f;
      OSUD_HIP(hipMemcpy(m->f8_slots, init.data(), init.size() * sizeof(float), hipMemcpyHostToDevice));
      m->f8_steps = 0;
    }
    OSUD_TRY(dev_alloc(W, &m->z0, (size_t)Np * D * es));
    BwdWs& b = m->bw;
    const size_t AC = m->ada_cols;
    OSUD_TRY(dev_alloc(W, &b.dhA, (size_t)Mp * D * 4));
    OSUD_TRY(dev_alloc(W, &b.dhB, (size_t)Mp * D * 4));
    OSUD_TRY(dev_alloc(W, &b.du, (size_t)Mp * D * 4));
    OSUD_TRY(dev_alloc(W, &b.dbr, (size_t)Mp * D * es));
    OSUD_TRY(dev_alloc(W, &b.dbr2, (size_t)Mp * D * es));
    OSUD_TRY(dev_alloc(W, &b.dz1, (size_t)Mp * 4 * D * es));
    OSUD_TRY(dev_alloc(W, &b.dqkv, (size_t)Mp * 3 * D * es));
    OSUD_TRY(dev_alloc(W, &b.dao, (size_t)Mp * D * es));
    const size_t tcols = 4 * D > (size_t)m->Kp ? 4 * D : (size_t)m->Kp;  // widest matrix that gets transposed
    OSUD_TRY(dev_alloc(W, &b.tA, (size_t)Mp * tcols * es));
    OSUD_TRY(dev_alloc(W, &b.tB, (size_t)Mp * tcols * es));
    OSUD_TRY(dev_alloc(W, &b.dada, (size_t)Np * AC * 4));
    OSUD_TRY(dev_alloc(W, &b.dada_te, (size_t)Np * AC * es * 2));  // [Np][AC] + its transpose [AC][Np]
    OSUD_TRY(dev_alloc(W, &b.dWada, AC * D * 4));
    OSUD_TRY(dev_alloc(W, &b.dbada, AC * 4));
    OSUD_TRY(dev_alloc(W, &b.dsb, (size_t)Np * D * 4));
    OSUD_TRY(dev_alloc(W, &b.db, (size_t)Np * D * 4));
    OSUD_TRY(dev_alloc(W, &b.dth, (size_t)Np * D * 4));
    OSUD_TRY(dev_alloc(W, &b.db_te, (size_t)Np * D * es));
    OSUD_TRY(dev_alloc(W, &b.dz0, (size_t)Np * D * es));
    OSUD_TRY(dev_alloc(W, &b.small_t1, (size_t)Np * (D > 256 ? D : 256) * es));
    OSUD_TRY(dev_alloc(W, &b.small_t2, (size_t)Np * (D > 256 ? D : 256) * es));
    OSUD_TRY(dev_alloc(W, &b.sb_t, (size_t)Np * D * es));
    OSUD_TRY(dev_alloc(W, &b.dWe, (size_t)D * m->Kp * 4));
    b.splitk_elems = (size_t)16 * 4 * D * D;  // up to 16 partial slabs of the largest weight gradient
    OSUD_TRY(dev_alloc(W, &b.splitk, b.splitk_elems * 4, false));
    OSUD_TRY(dev_alloc(W, &b.splitk2, b.splitk_elems * 4, false));
    OSUD_TRY(dev_alloc(W, &b.rowpart, (size_t)(2 * m->L + 2) * (Mp / 64) * (6 * D + 64) * 4, false));
    b.b1part_stride = (size_t)(Mp / 32) * 4 * D;
    b.bqkvpart_stride = (size_t)(nN > (Mp + 255) / 256 ? nN : (Mp + 255) / 256) * 3 * D;  // one row per sample (streamed kernel) or per 256-token row block (column-sum pass)
    // (fc1's bias gradient rides in the GELU' epilogue in the bf16 tier only -- train.hip: fused_b1 -- the fp32 tier never touches these rows;
    //  DiT-XL at 32 768 tokens: 528 MB here + 829 MB of rowpart, stated in DESIGN.md section 3)
    if (m->prec == OSUD_PREC_BF16) OSUD_TRY(dev_alloc(W, &b.b1part, (size_t)m->L * b.b1part_stride * 4, false));
    OSUD_TRY(dev_alloc(W, &b.bqkvpart, (size_t)m->L * b.bqkvpart_stride * 4, false));
    {  // the widest column sum: a transpose's (rows / 64) shares of 4 D (or the padded first-layer width) columns, or (Np / 64) x AC
      const size_t a = (size_t)(Mp / 64) * tcols, c = (size_t)(Np / 64) * AC;
      b.colpart_elems = a > c ? a : c;
      OSUD_TRY(dev_alloc(W, &b.colpart, b.colpart_elems * 4, false));
      OSUD_TRY(dev_alloc(W, &b.colpart2, b.colpart_elems * 4, false));
    }
    OSUD_TRY(dev_alloc(W, &b.attn_delta, (size_t)nN * m->H * Tp * 4));
    OSUD_TRY(dev_alloc(W, &b.seg_tbl, 32 * sizeof(float*)));
    for (float*& q : b.seg_tbl_host) q = nullptr;
  }
  m->cap_N = nN; m->cap_T = nT; m->cap_Tp = Tp; m->cap_Mp = Mp; m->cap_Np = Np;
  return OSUD_OK;
}

// fp8 tier: static activation scales (value * scale -> e4m3, saturating at +-448).  LayerNorm outputs are O(1) with a few
// sigma of headroom after modulation, attention outputs are convex combinations of V rows, GELU outputs are >= -0.17.
constexpr float kF8ScaleLN = 8.0f, kF8ScaleAttn = 16.0f, kF8ScaleGelu = 8.0f;

int dit_forward_impl(osud_dit* m, const float* x, const int64_t* t, const float* o, const float* c, const int64_t* y,
                     const uint8_t* mask, int N, int T, float cfg_scale, bool combine_cfg, float* out, bool save,
                     hipStream_t st) {
  OSUD_CHECK_ARG(m && x && t && o && c && y && out, "forward: null argument");
  OSUD_CHECK_ARG(N > 0 && T > 0, "forward: empty batch (N=%d, T=%d)", N, T);
  int missing = 0;
  for (auto& kv : m->have) missing += kv.second ? 0 : 1;
  if (missing) {
    for (auto& kv : m->have)
      if (!kv.second) {
        set_error("forward: %d parameter(s) not set, first missing: %s", missing, kv.first.c_str());
        break;
      }
    return OSUD_ERR_STATE;
  }
  const bool cfg = cfg_scale >= 0.f;
  OSUD_CHECK_ARG(!cfg || N % 2 == 0, "forward_with_cfg: batch must be [cond; uncond] halves, got N=%d", N);
  OSUD_CHECK_ARG(!save || m->training, "forward(save): workspaces were not reserved for training");
  OSUD_TRY(dit_ensure_ws(m, N, T, m->training));
  // fp8 TRAINING (BASELINE config 5): in_proj / out_proj / fc1 / fc2 of every block on e4m3 operands with delayed per-tensor
  // scaling -- a tensor is quantised with the scale derived from the amax it showed in the previous step (launch_f8_update,
  // once per forward).  The very first training forward has no history: it runs its GEMMs in bf16 and only records.  Weights
  // carry per-output-channel scales.  The e4m3 twins of the four GEMM inputs stay in per-layer buffers: the backward pass
  // forms the weight gradients from them (wgrad8_kernel).
  const bool f8_train = m->fp8 && save;
  const bool f8_live = f8_train && m->f8_steps > 0;
  if (f8_train) OSUD_TRY(launch_f8_update(m->f8_slots, m->L * kF8Slots, st, m->f8_parts));
  auto slot = [&](int l, int which) { return m->f8_slots + ((size_t)l * kF8Slots + which) * 4; };
  auto parts = [&](int l, int which) { return m->f8_parts + ((size_t)l * kF8Slots + which) * f8_amax_parts(); };
  // fp8 inference: per-block activation scales (static defaults, or calibrated: osud_dit_calibrate_fp8)
  if (m->fp8 && m->f8_inf.empty()) {
    m->f8_inf.resize((size_t)m->L * 4);
    for (int l = 0; l < m->L; ++l) {
      m->f8_inf[(size_t)l * 4 + 0] = kF8ScaleLN; m->f8_inf[(size_t)l * 4 + 1] = kF8ScaleAttn;
      m->f8_inf[(size_t)l * 4 + 2] = kF8ScaleLN; m->f8_inf[(size_t)l * 4 + 3] = kF8ScaleGelu;
    }
  }
  const bool cal = m->fp8 && m->f8_calibrating && !save;  // calibration forward: bf16 arithmetic, amax of the four tensors recorded
  auto cslot = [&](int l, int which) { return m->f8_cal_slots + ((size_t)l * 4 + which) * 4; };

  const int D = m->D, L = m->L, Tp = round_up(T, 64), M = N * Tp, Mp = round_up(M, 128), Np = round_up(N, 128);
  const int prec = m->prec, AC = m->ada_cols;
  const int* bf = m->bform;  // the operand forms of the blocks' four big GEMMs (and of what feeds them): dit.h
  const bool f8_slim = f8_twins_only(m, f8_live, Mp);  // the bf16 forms of the GEMM inputs have no reader this step: not written
  // Gates (osud_dit_forward_gate): the sharded optimizer's all-gather of the updated master weights and their re-pack run on a
  // side stream while this forward is already under way -- phase p's kernels wait for phase p's event only.
  bool gated = false;
  for (hipEvent_t e : m->gate_ev) gated = gated || e != nullptr;
  auto gate = [&](int phase) -> int {
    if (phase < (int)m->gate_ev.size() && m->gate_ev[(size_t)phase] != nullptr) {
      OSUD_HIP(hipStreamWaitEvent(st, m->gate_ev[(size_t)phase], 0));
      m->gate_ev[(size_t)phase] = nullptr;
    }
    return OSUD_OK;
  };
  OSUD_TRY(gate(0));

  // token embedding + first linear (models.py:315-317)
  // (bf16 tier: rows and weights in the split [hi | lo | hi] x [w_hi | w_hi | w_lo] form, Ke = 3 Kp -- see embed_kernel)
  float* h = m->training ? m->saved[0].h_in : m->h;
  if (m->embed_const_on) {  // inside a sampler loop: h = h0c (made by embed_const_prepare) + coordinate features x their 256 columns
    const int kx = m->x3 ? 256 : 768;  // 256 coordinate features: a plane pair (split-bf16 tier) or [hi | lo | hi] (bf16 tier's split first linear)
    OSUD_TRY(launch_embed(prec, x, o, c, m->freqs64, m->pf[0], m->pf[1], m->e0, N, T, Tp, Mp, m->E, m->Kp, cfg ? N / 2 : 0, st, true, 1));
    OSUD_TRY(gemm(m, EPI_GATE_RES, m->e0, kx, m->w_ex, kx, Mp, D, kx, h, D, m->b_e, st, m->ones_d, 0, Tp, N, nullptr, m->h0c));
  } else {
  OSUD_TRY(launch_embed(prec, x, o, c, m->freqs64, m->pf[0], m->pf[1], m->e0, N, T, Tp, Mp, m->E, m->Kp, cfg ? N / 2 : 0, st,
                        m->split_first));
  OSUD_TRY(gemm(m, EPI_BIAS_F32, m->e0, m->Ke, m->w_e, m->Ke, Mp, D, m->Ke, h, D, m->b_e, st));
  }
  // conditioning vector b = t_emb + y_emb (models.py:318-320) and ALL adaLN modulations in one GEMM:
  // b is the same for every block, so the 12 x (D -> 6D) + (D -> 2D) linears are one (Np x D) x (D x AC) product.
  if (m->tvec_table_on) {  // inside a sampler loop: the MLP's output for every schedule index was made by tvec_table_prepare
    OSUD_TRY(launch_cond(prec, m->tv_all, m->table_ref ? m->table_ref : m->table, y, m->cfg.table_rows, m->bvec, m->sb, N, Np, D, st,
                         m->t_index));
  } else {
  OSUD_TRY(launch_temb(prec, t, m->freqs128, m->temb, N, Np, st));
  OSUD_TRY(gemm(m, EPI_BIAS_SILU_TE, m->temb, 256, m->w_t0, 256, Np, D, 256, m->th, D, m->b_t0, st, nullptr, 0, 0, 0,
                m->training ? m->z0 : nullptr));
  OSUD_TRY(gemm(m, EPI_BIAS_F32, m->th, D, m->w_t2, D, Np, D, D, m->tvec, D, m->b_t2, st));
  OSUD_TRY(launch_cond(prec, m->tvec, m->table_ref ? m->table_ref : m->table, y, m->cfg.table_rows, m->bvec, m->sb, N, Np, D, st));
  }
  // (gated forward: a block's adaLN weights arrive with the block, so its 6D modulation columns are produced in front of it)
  auto ada_part = [&](int l) -> int {  // l == L: the final layer's 2D columns
    const size_t off = (size_t)l * 6 * D;
    return gemm(m, EPI_BIAS_F32, m->sb, D, (const char*)m->w_ada + off * D * m->esz, D, Np, l < L ? 6 * D : 2 * D, D, m->ada + off, AC,
                m->b_ada + off, st);
  };
  if (!gated) OSUD_TRY(gemm(m, EPI_BIAS_F32, m->sb, D, m->w_ada, D, Np, AC, D, m->ada, AC, m->b_ada, st));

  if (mask != nullptr) OSUD_TRY(launch_mask_tiles(mask, T, Tp, m->kb_class, st));  // once per forward, shared by all blocks
  // Training: the gated residual updates (h += gate * branch, models.py:161-175) are folded into the NEXT LayerNorm
  // kernel (`pend` = branch output not yet added to the residual stream `h`): the branch must be materialised for the
  // backward pass anyway, and the GEMM keeps a lean epilogue (measured -1.9 % step time).  Inference adds the branch in
  // the GEMM epilogue, in place, which moves fewer bytes there (the folded form was 1.7 % slower per sampling step).
  const void* pend = nullptr;
  int pend_gate = 0;
  for (int l = 0; l < L; ++l) {  // DiTBlock.forward, models.py:151-175
    const BlockWeights& w = m->blk[(size_t)l];
    LayerSaved* sv = m->training ? &m->saved[(size_t)l] : nullptr;
    void* u1 = sv ? sv->u1 : m->u;
    void* qk = sv ? sv->qk : m->qk;
    void* ao = sv ? sv->ao : m->ao;
    void* u2 = sv ? sv->u2 : m->u;
    void* g = sv ? sv->g : m->g;
    void* br1 = sv ? sv->br1 : nullptr;
    void* br2 = sv ? sv->br2 : nullptr;
    float* h_in = sv ? sv->h_in : h;    // residual stream entering the block (training keeps every version)
    float* h_mid = sv ? sv->h_mid : h;
    const int base = l * 6 * D;
    const int qcols = 3 * D;
    const float *fs = m->fp8 ? &m->f8_inf[(size_t)l * 4] : nullptr;  // this block's {LN1, attention, LN2, GELU} output scales
    if (gated) {
      OSUD_TRY(gate(l + 1));
      OSUD_TRY(ada_part(l));
    }
    if (!sv && m->fp8 && !cal) {
      OSUD_TRY(launch_ln_mod(prec, h, m->ada, AC, base, base + D, m->u8, nullptr, Mp, Tp, N, D, st, nullptr, 0, nullptr, fs[0]));
      OSUD_TRY(gemm8(m, EPI_BIAS_TE, m->u8, w.w8_qkv, Mp, qcols, D, qk, qcols, w.b_qkv, w.dq_qkv, 0.f, st, nullptr, 0, 0, 0, 1.0f / fs[0]));
      OSUD_TRY(launch_attention(prec, qk, qcols, mask, m->ao8, nullptr, N, T, Tp, Mp, m->H, m->hd, st, m->kb_class, fs[1]));
    } else {
      // (fp8 training: LayerNorm writes the e4m3 twin of its output and this step's amax itself)
      if (f8_train)
        OSUD_TRY(launch_ln_mod_twin(h, m->ada, AC, base, base + D, f8_slim ? nullptr : u1, sv ? sv->stats1 : nullptr, Mp, Tp, N, D, st, pend, pend_gate,
                                    pend ? h_in : nullptr, f8_live ? sv->u1_8 : nullptr, slot(l, 0), parts(l, 0)));
      else
      OSUD_TRY(launch_ln_mod(bf[0], h, m->ada, AC, base, base + D, u1, sv ? sv->stats1 : nullptr, Mp, Tp, N, D, st, pend,
                             pend_gate, pend ? h_in : nullptr));
      // packed in_proj: one 3D-wide product, Q | K | V row-major (the attention kernels transpose V on the LDS read)
      if (f8_live) OSUD_TRY(gemm8(m, EPI_BIAS_TE, sv->u1_8, w.w8_qkv, Mp, qcols, D, qk, qcols, w.b_qkv, w.dq_qkv, 0.f, st, nullptr, 0, 0, 0, 0.f,
                                  slot(l, 0) + 1));
      else
      OSUD_TRY(gemm_blk(m, 0, EPI_BIAS_TE, u1, D, w.w_qkv, D, Mp, qcols, D, qk, qcols, w.b_qkv, st));
      OSUD_TRY(launch_attention(bf[1], qk, qcols, mask, ao, sv ? sv->lse : nullptr, N, T, Tp, Mp, m->H, m->hd, st, m->kb_class));
      if (cal) {
        OSUD_TRY(launch_f8_quantize(u1, nullptr, (size_t)Mp * D, cslot(l, 0), st));
        OSUD_TRY(launch_f8_quantize(ao, nullptr, (size_t)Mp * D, cslot(l, 1), st));
      }
    }
    if (!sv && m->fp8 && !cal) {
      // the four big GEMMs of the block on e4m3 operands (this block's LN1 / qkv / attention were emitted above in fp8 form)
      OSUD_TRY(gemm8(m, EPI_GATE_RES, m->ao8, w.w8_o, Mp, D, D, h, D, w.b_o, w.dq_o, 0.f, st, m->ada + base + 2 * D, AC, Tp, N, 1.0f / fs[1]));
      OSUD_TRY(launch_ln_mod(prec, h, m->ada, AC, base + 3 * D, base + 4 * D, m->u8, nullptr, Mp, Tp, N, D, st, nullptr, 0, nullptr, fs[2]));
      OSUD_TRY(gemm8(m, EPI_BIAS_GELU_TE, m->u8, w.w8_1, Mp, 4 * D, D, m->g8, 4 * D, w.b1, w.dq_1, fs[3], st, nullptr, 0, 0, 0, 1.0f / fs[2]));
      OSUD_TRY(gemm8(m, EPI_GATE_RES, m->g8, w.w8_2, Mp, D, 4 * D, h, D, w.b2, w.dq_2, 0.f, st, m->ada + base + 5 * D, AC, Tp, N, 1.0f / fs[3]));
      continue;
    }
    if (!sv) {
      OSUD_TRY(gemm_blk(m, 1, EPI_GATE_RES, ao, D, w.w_o, D, Mp, D, D, h, D, w.b_o, st, m->ada + base + 2 * D, AC, Tp, N));
      OSUD_TRY(launch_ln_mod(bf[2], h, m->ada, AC, base + 3 * D, base + 4 * D, u2, nullptr, Mp, Tp, N, D, st));
      // (fc1's GELU epilogue writes fc2's operand: in the other K-blocked form where the two GEMMs differ)
      OSUD_TRY(gemm_blk(m, 2, bf[2] == bf[3] ? EPI_BIAS_GELU_TE : EPI_BIAS_GELU_ALT, u2, D, w.w1, D, Mp, 4 * D, D, g, 4 * D, w.b1, st));
      if (cal) {
        OSUD_TRY(launch_f8_quantize(u2, nullptr, (size_t)Mp * D, cslot(l, 2), st));
        OSUD_TRY(launch_f8_quantize(g, nullptr, (size_t)Mp * 4 * D, cslot(l, 3), st));
      }
      OSUD_TRY(gemm_blk(m, 3, EPI_GATE_RES, g, 4 * D, w.w2, 4 * D, Mp, D, 4 * D, h, D, w.b2, st, m->ada + base + 5 * D, AC, Tp, N));
      continue;
    }
    if (f8_train) {  // attention output: e4m3 twin for out_proj (forward here, weight gradient in the backward pass) + this step's amax
      OSUD_TRY(launch_f8_quantize(ao, f8_live ? sv->ao_8 : nullptr, (size_t)Mp * D, slot(l, 6), st));
    }
    if (f8_live) OSUD_TRY(gemm8(m, EPI_BIAS_TE, sv->ao_8, w.w8_o, Mp, D, D, br1, D, w.b_o, w.dq_o, 0.f, st, nullptr, 0, 0, 0, 0.f, slot(l, 6) + 1));
    else
    OSUD_TRY(gemm(m, EPI_BIAS_TE, ao, D, w.w_o, D, Mp, D, D, br1, D, w.b_o, st));
    if (f8_train)
      OSUD_TRY(launch_ln_mod_twin(h_in, m->ada, AC, base + 3 * D, base + 4 * D, f8_slim ? nullptr : u2, sv->stats2, Mp, Tp, N, D, st, br1, base + 2 * D,
                                  h_mid, f8_live ? sv->u2_8 : nullptr, slot(l, 1), parts(l, 1)));
    else
    OSUD_TRY(launch_ln_mod(prec, h_in, m->ada, AC, base + 3 * D, base + 4 * D, u2, sv->stats2, Mp, Tp, N, D, st, br1,
                           base + 2 * D, h_mid));
    if (f8_train) {
      // (live steps: the fc1 epilogue writes the e4m3 twin of its GELU output and records its amax itself)
      if (f8_live) OSUD_TRY(gemm8(m, EPI_BIAS_GELU_BF, sv->u2_8, w.w8_1, Mp, 4 * D, D, f8_slim ? nullptr : g, 4 * D, w.b1, w.dq_1, 0.f, st, nullptr, 0, 0, 0, 0.f,
                                  slot(l, 1) + 1, sv->z1, nullptr, nullptr, nullptr, sv->g_8, slot(l, 2)));
      else {
        OSUD_TRY(gemm(m, EPI_BIAS_GELU_TE, u2, D, w.w1, D, Mp, 4 * D, D, g, 4 * D, w.b1, st, nullptr, 0, 0, 0, sv->z1));
        OSUD_TRY(launch_f8_quantize(g, nullptr, (size_t)Mp * 4 * D, slot(l, 2), st));
      }
      if (f8_live) OSUD_TRY(gemm8(m, EPI_BIAS_TE, sv->g_8, w.w8_2, Mp, D, 4 * D, br2, D, w.b2, w.dq_2, 0.f, st, nullptr, 0, 0, 0, 0.f,
                                  slot(l, 2) + 1));
      else OSUD_TRY(gemm(m, EPI_BIAS_TE, g, 4 * D, w.w2, 4 * D, Mp, D, 4 * D, br2, D, w.b2, st));
    } else {
    OSUD_TRY(gemm(m, EPI_BIAS_GELU_TE, u2, D, w.w1, D, Mp, 4 * D, D, g, 4 * D, w.b1, st, nullptr, 0, 0, 0, sv->z1));
    OSUD_TRY(gemm(m, EPI_BIAS_TE, g, 4 * D, w.w2, 4 * D, Mp, D, 4 * D, br2, D, w.b2, st));
    }
    pend = br2;
    pend_gate = base + 5 * D;
    h = h_mid;
  }
  if (gated) {
    OSUD_TRY(gate(L + 1));
    OSUD_TRY(ada_part(L));
  }
  // FinalLayer (models.py:192-196) + swapaxes (:324); adds the last MLP branch first
  OSUD_TRY(launch_final(h, m->ada, AC, L * 6 * D, L * 6 * D + D, m->w_f, m->b_f, out, nullptr,
                        m->training ? m->saved[(size_t)L].stats1 : nullptr, N, T, Tp, D, m->C2, st, prec, pend, pend_gate,
                        m->training ? m->saved[(size_t)L].h_in : nullptr));
  if (cfg && combine_cfg) OSUD_TRY(launch_cfg_combine(out, N, m->C, m->C2, T, cfg_scale, st));
  if (save) {
    m->last_y = y;
    m->last_N = N;
    m->last_T = T;
    if (f8_train) ++m->f8_steps;
  }
  return OSUD_OK;
}

}  // namespace osud

using namespace osud;

// ------------------------------------------------------------------------------- C ABI
extern "C" const char* osud_last_error(void) { return g_err.c_str(); }
extern "C" int osud_version(void) { return 1; }
extern "C" const char* osud_build_arch(void) { return "gfx950"; }

static int upload_f32(osud_dit* m, float** dst, const float* src, size_t n, hipStream_t st) {
  if (!*dst) OSUD_TRY(dev_alloc(m->owned, dst, n * 4, false));
  if (m->defer_copy && n % 4 == 0) return m->defer_copy->add(src, *dst, n / 4);
  OSUD_HIP(hipMemcpyAsync(*dst, src, n * 4, hipMemcpyDeviceToDevice, st));
  return OSUD_OK;
}
// fp32 master -> TE copy of a weight (n % 4 == 0 for every DiT weight)
static int convert_w(osud_dit* m, const float* src, void* dst, size_t rows, size_t cols, hipStream_t st, int block_gemm = -1 /* 0..3: that GEMM's weight */) {
  const size_t n = rows * cols;
  const int form = block_gemm >= 0 ? m->bform[block_gemm] : m->prec;
  if (form == OSUD_PREC_F16F8) return launch_pack_rows_h8(src, (int)cols, (int)cols, dst, (int)cols, (int)rows, true, st);  // (same 4 bytes per element)
  if (form == OSUD_PREC_F16W8) return launch_pack_rows_w8(src, (int)cols, (int)cols, dst, (int)cols, (int)rows, true, st);  // (3 of the buffer's 4 bytes per element)
  if (m->x3) return launch_pack_rows_x3(src, (int)cols, (int)cols, dst, (int)cols, (int)rows, st);  // rows of [w_hi | w_lo]
  if (m->defer_convert && n % 4 == 0) return m->defer_convert->add(src, dst, n / 4);
  return launch_convert(m->prec, src, dst, n, st);
}

extern "C" int osud_dit_create(const osud_dit_cfg* cfg, osud_dit** out) {
  OSUD_CHECK_ARG(cfg && out, "dit_create: null argument");
  OSUD_CHECK_ARG(cfg->hidden > 0 && cfg->hidden % 128 == 0, "dit_create: hidden size %d must be a multiple of 128",
                 cfg->hidden);
  OSUD_CHECK_ARG(cfg->heads > 0 && cfg->hidden % cfg->heads == 0, "dit_create: heads=%d does not divide hidden=%d",
                 cfg->heads, cfg->hidden);
  OSUD_CHECK_ARG(cfg->depth > 0 && cfg->context > 0 && cfg->in_channels == 2 && cfg->table_rows > 0,
                 "dit_create: bad depth/context/in_channels/table_rows");
  OSUD_CHECK_ARG(cfg->precision == OSUD_PREC_BF16 || cfg->precision == OSUD_PREC_F32 || cfg->precision == OSUD_PREC_FP8 ||
                     cfg->precision == OSUD_PREC_BF16X3 || cfg->precision == OSUD_PREC_F16F8 || cfg->precision == OSUD_PREC_F16 ||
                     cfg->precision == OSUD_PREC_F16W8 || cfg->precision == OSUD_PREC_F16M8,
                 "dit_create: unknown precision %d", cfg->precision);
  const int hd = cfg->hidden / cfg->heads;
  if (hd != 64 && hd != 72) {
    set_error("dit_create: head_dim %d not built (64, 72)", hd);
    return OSUD_ERR_UNSUPPORTED;
  }
  osud_dit* m = new osud_dit();
  m->cfg = *cfg;
  m->D = cfg->hidden; m->L = cfg->depth; m->H = cfg->heads; m->hd = hd; m->E = cfg->context;
  m->C = cfg->in_channels; m->C2 = cfg->learn_sigma ? 2 * cfg->in_channels : cfg->in_channels;
  m->fp8 = cfg->precision == OSUD_PREC_FP8;
  m->h8 = cfg->precision == OSUD_PREC_F16F8;
  m->w8 = cfg->precision == OSUD_PREC_F16W8;
  const bool m8 = cfg->precision == OSUD_PREC_F16M8;  // per-GEMM choice between the two forms: option "f16m8_forms", bit i = GEMM i on w8_t
  m->prec = m->fp8 ? OSUD_PREC_BF16 : ((m->h8 || m->w8 || m8) ? OSUD_PREC_BF16X3 : cfg->precision);
  for (int i = 0; i < 4; ++i)
    m->bform[i] = m->h8 ? OSUD_PREC_F16F8 : (m->w8 ? OSUD_PREC_F16W8 : (m8 ? (((opt(OPT_F16M8_FORMS) >> i) & 1) ? OSUD_PREC_F16W8 : OSUD_PREC_F16F8) : m->prec));
  if (m8) m->h8 = m->w8 = true;  // (both forms occur: buffers are sized for the wider one either way)
  m->x3 = m->prec == OSUD_PREC_BF16X3;  // every TE matrix is a plane pair [hi | lo] (common.h): esz = 4 bytes per logical element
  m->esz = (int)elem_size(m->prec);
  m->Kp = round_up(cfg->in_channels * 128 + 128 + cfg->context, 128);  // 528 -> 640
  {  // osud_set_option("split_first", 0) before the handle is created: plain bf16 first linear (A/B measurements of the fast tier's deviation)
    // (embed_kernel stages 16 rows of [hi | lo | hi] in LDS: 16 * Kp * 6 bytes must fit 64 KiB, i.e. context sizes up to 256;
    //  wider contexts keep the plain bf16 first linear)
    m->split_first = (m->prec == OSUD_PREC_BF16 || m->prec == OSUD_PREC_F16) && opt(OPT_SPLIT_FIRST) != 0 && (size_t)16 * m->Kp * 6 <= 64 * 1024;
  }
  m->Ke = m->split_first ? 3 * m->Kp : m->Kp;
  m->z1_code = m->prec == OSUD_PREC_BF16 && opt(OPT_GELU_CODE) != 0 && (4 * m->D) % 32 == 0;
  m->ada_cols = 6 * m->D * m->L + 2 * m->D;
  if (hipGetDevice(&m->device) != hipSuccess) {
    delete m;
    return hip_fail(hipErrorNoDevice, "hipGetDevice", __FILE__, __LINE__);
  }
  const size_t D = m->D, es = m->esz;
  int rc = OSUD_OK;
  auto A = [&](auto** p, size_t bytes) { if (rc == OSUD_OK) rc = dev_alloc(m->owned, p, bytes); };
  A(&m->w_e, D * m->Ke * es); A(&m->b_e, D * 4);
  if (m->split_first) { A(&m->w_ex, D * 768 * es); A(&m->ones_d, D * 4); }
  if (m->x3) { A(&m->w_ex, D * 256 * es); A(&m->ones_d, D * 4); }
  A(&m->w_t0, D * 256 * es);  A(&m->b_t0, D * 4);
  A(&m->w_t2, D * D * es);    A(&m->b_t2, D * 4);
  A(&m->table, (size_t)cfg->table_rows * D * 4);
  A(&m->w_ada, (size_t)m->ada_cols * D * es); A(&m->b_ada, (size_t)m->ada_cols * 4);
  A(&m->w_f, (size_t)m->C2 * D * 4); A(&m->b_f, 16);
  A(&m->freqs64, 64 * 4); A(&m->freqs128, 128 * 4);
  m->blk.resize((size_t)m->L);
  for (auto& b : m->blk) {
    A(&b.w_qkv, 3 * D * D * es); A(&b.b_qkv, 3 * D * 4);
    A(&b.w_o, D * D * es);      A(&b.b_o, D * 4);
    A(&b.w1, 4 * D * D * es);   A(&b.b1, 4 * D * 4);
    A(&b.w2, 4 * D * D * es);   A(&b.b2, D * 4);
    if (m->fp8) {
      A(&b.w8_qkv, 3 * D * D); A(&b.dq_qkv, 3 * D * 4);
      A(&b.w8_o, D * D);       A(&b.dq_o, D * 4);
      A(&b.w8_1, 4 * D * D);   A(&b.dq_1, 4 * D * 4);
      A(&b.w8_2, 4 * D * D);   A(&b.dq_2, D * 4);
    }
  }
  if (rc == OSUD_OK && hipStreamCreateWithFlags(&m->cap_stream, hipStreamNonBlocking) != hipSuccess) rc = OSUD_ERR_HIP;
  if (rc != OSUD_OK) {
    osud_dit_destroy(m);
    return rc;
  }
  if (m->ones_d) {
    std::vector<float> ones((size_t)m->D, 1.0f);
    (void)hipMemcpy(m->ones_d, ones.data(), ones.size() * sizeof(float), hipMemcpyHostToDevice);
  }
  // default frequency tables: exp(-ln(1e4) * k / half) in fp32 (positional_embedding.py:39-44).
  // The host may overwrite them with torch's own values through the "const.freqs64/128" keys.
  float f64[64], f128[128];
  for (int k = 0; k < 64; ++k) f64[k] = expf((float)(-log(10000.0)) * (float)k / 64.0f);
  for (int k = 0; k < 128; ++k) f128[k] = expf((float)(-log(10000.0)) * (float)k / 128.0f);
  (void)hipMemcpy(m->freqs64, f64, sizeof f64, hipMemcpyHostToDevice);
  (void)hipMemcpy(m->freqs128, f128, sizeof f128, hipMemcpyHostToDevice);
  // expected state-dict keys
  const char* top[] = {"xoc_embedder.playfield_size", "xoc_embedder.mlp.0.weight", "xoc_embedder.mlp.0.bias",
                       "t_embedder.mlp.0.weight", "t_embedder.mlp.0.bias", "t_embedder.mlp.2.weight",
                       "t_embedder.mlp.2.bias", "y_embedder.embedding_table.weight", "final_layer.linear.weight",
                       "final_layer.linear.bias", "final_layer.adaLN_modulation.1.weight",
                       "final_layer.adaLN_modulation.1.bias"};
  for (const char* k : top) m->have[k] = false;
  const char* per[] = {"attn.in_proj_weight", "attn.in_proj_bias", "attn.out_proj.weight", "attn.out_proj.bias",
                       "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias",
                       "adaLN_modulation.1.weight", "adaLN_modulation.1.bias"};
  for (int l = 0; l < m->L; ++l)
    for (const char* k : per) m->have["blocks." + std::to_string(l) + "." + k] = false;
  *out = m;
  return OSUD_OK;
}

extern "C" void osud_dit_destroy(osud_dit* m) {
  if (!m) return;
  if (m->graph_exec) (void)hipGraphExecDestroy(m->graph_exec);
  if (m->cap_stream) (void)hipStreamDestroy(m->cap_stream);
  for (hipEvent_t& e : m->bw.side_ev)
    if (e) (void)hipEventDestroy(e);
  if (m->bw.side) (void)hipStreamDestroy(m->bw.side);
  for (void* p : m->owned) (void)hipFree(p);
  for (void* p : m->ws_owned) (void)hipFree(p);
  delete m;
}

extern "C" int osud_dit_missing_params(const osud_dit* m) {
  if (!m) return -1;
  int n = 0;
  for (auto& kv : m->have) n += kv.second ? 0 : 1;
  return n;
}

static bool shape_is(const int64_t* s, int nd, int64_t a, int64_t b = -1) {
  return b < 0 ? (nd == 1 && s[0] == a) : (nd == 2 && s[0] == a && s[1] == b);
}

// fp8 tier: per-output-channel e4m3 form of a weight; inside a refresh the launch is deferred into the refresh's list
static int quant_w(osud_dit* m, const float* src, int rows, int cols, void* q, float* dq, hipStream_t st) {
  if (m->defer_quant && cols % 8 == 0) return m->defer_quant->add(src, rows, cols, q, dq);
  return launch_quantize_rows(src, rows, cols, q, dq, 1.0f, st);
}

extern "C" int osud_dit_set_param(osud_dit* m, const char* key, const float* src, const int64_t* shape, int ndim,
                                  osud_stream stream) {
  OSUD_CHECK_ARG(m && key && src && shape, "set_param: null argument");
  hipStream_t st = (hipStream_t)stream;
  const std::string k(key);
  const int64_t D = m->D;
  const int prec = m->prec;
#define SHAPE(...) OSUD_CHECK_ARG(shape_is(shape, ndim, __VA_ARGS__), "set_param(%s): unexpected shape", key)
  if (k == "const.freqs64") { SHAPE(64); return upload_f32(m, &m->freqs64, src, 64, st); }
  if (k == "const.freqs128") { SHAPE(128); return upload_f32(m, &m->freqs128, src, 128, st); }
  auto it = m->have.find(k);
  if (it == m->have.end()) {
    set_error("set_param: unexpected key '%s'", key);  // load_state_dict(strict=True): unexpected key
    return OSUD_ERR_ARG;
  }
  int rc = OSUD_OK;
  if (k == "xoc_embedder.playfield_size") {
    SHAPE(2);
    OSUD_HIP(hipStreamSynchronize(st));
    OSUD_HIP(hipMemcpy(m->pf, src, 8, hipMemcpyDeviceToHost));
  } else if (k == "xoc_embedder.mlp.0.weight") {
    SHAPE(D, 384 + m->E);
    rc = m->x3 ? launch_pack_rows_x3(src, 384 + m->E, 384 + m->E, m->w_e, m->Kp, (int)D, st)
         : m->split_first ? launch_pack_rows_split(src, 384 + m->E, 384 + m->E, m->w_e, m->Kp, (int)D, st, m->prec)
                          : launch_pack_rows(prec, src, 384 + m->E, 384 + m->E, m->w_e, m->Kp, m->Kp, (int)D, st);
    if (rc == OSUD_OK && m->split_first) rc = launch_pack_rows_split(src, 384 + m->E, 256, m->w_ex, 256, (int)D, st, m->prec);  // coordinate columns
    if (rc == OSUD_OK && m->x3) rc = launch_pack_rows_x3(src, 384 + m->E, 256, m->w_ex, 256, (int)D, st);
  } else if (k == "xoc_embedder.mlp.0.bias") { SHAPE(D); rc = upload_f32(m, &m->b_e, src, D, st);
  } else if (k == "t_embedder.mlp.0.weight") { SHAPE(D, 256); rc = convert_w(m, src, m->w_t0, D, 256, st);
  } else if (k == "t_embedder.mlp.0.bias") { SHAPE(D); rc = upload_f32(m, &m->b_t0, src, D, st);
  } else if (k == "t_embedder.mlp.2.weight") { SHAPE(D, D); rc = convert_w(m, src, m->w_t2, D, D, st);
  } else if (k == "t_embedder.mlp.2.bias") { SHAPE(D); rc = upload_f32(m, &m->b_t2, src, D, st);
  } else if (k == "y_embedder.embedding_table.weight") {
    SHAPE(m->cfg.table_rows, D);
    if (m->defer_copy) {  // osud_dit_refresh: the master stays where it is (the caller keeps it alive between refreshes)
      m->table_ref = src;
    } else {
      rc = upload_f32(m, &m->table, src, (size_t)m->cfg.table_rows * D, st);
      m->table_ref = m->table;
    }
  } else if (k == "final_layer.linear.weight") { SHAPE(m->C2, D); rc = upload_f32(m, &m->w_f, src, (size_t)m->C2 * D, st);
  } else if (k == "final_layer.linear.bias") { SHAPE(m->C2); rc = upload_f32(m, &m->b_f, src, m->C2, st);
  } else if (k == "final_layer.adaLN_modulation.1.weight") {
    SHAPE(2 * D, D);
    rc = convert_w(m, src, (char*)m->w_ada + (size_t)m->L * 6 * D * D * m->esz, 2 * D, D, st);
  } else if (k == "final_layer.adaLN_modulation.1.bias") {
    SHAPE(2 * D);
    float* dst = m->b_ada + (size_t)m->L * 6 * D;
    rc = upload_f32(m, &dst, src, 2 * D, st);
  } else {  // blocks.<l>.<name>
    const size_t p1 = k.find('.', 7);
    const int l = atoi(k.substr(7, p1 - 7).c_str());
    const std::string name = k.substr(p1 + 1);
    BlockWeights& b = m->blk[(size_t)l];
    const size_t es = m->esz;
    if (name == "attn.in_proj_weight") {
      SHAPE(3 * D, D);  // rows [Wq; Wk; Wv]
      rc = convert_w(m, src, b.w_qkv, 3 * D, D, st, 0);
      if (rc == OSUD_OK && m->fp8) rc = quant_w(m, src, (int)(3 * D), (int)D, b.w8_qkv, b.dq_qkv, st);
    } else if (name == "attn.in_proj_bias") {
      SHAPE(3 * D);
      rc = upload_f32(m, &b.b_qkv, src, 3 * D, st);
    } else if (name == "attn.out_proj.weight") {
      SHAPE(D, D);
      rc = convert_w(m, src, b.w_o, D, D, st, 1);
      if (rc == OSUD_OK && m->fp8) rc = quant_w(m, src, (int)D, (int)D, b.w8_o, b.dq_o, st);
    } else if (name == "attn.out_proj.bias") { SHAPE(D); rc = upload_f32(m, &b.b_o, src, D, st);
    } else if (name == "mlp.fc1.weight") {
      SHAPE(4 * D, D);
      rc = convert_w(m, src, b.w1, 4 * D, D, st, 2);
      if (rc == OSUD_OK && m->fp8) rc = quant_w(m, src, (int)(4 * D), (int)D, b.w8_1, b.dq_1, st);
    } else if (name == "mlp.fc1.bias") { SHAPE(4 * D); rc = upload_f32(m, &b.b1, src, 4 * D, st);
    } else if (name == "mlp.fc2.weight") {
      SHAPE(D, 4 * D);
      rc = convert_w(m, src, b.w2, D, 4 * D, st, 3);
      if (rc == OSUD_OK && m->fp8) rc = quant_w(m, src, (int)D, (int)(4 * D), b.w8_2, b.dq_2, st);
    } else if (name == "mlp.fc2.bias") { SHAPE(D); rc = upload_f32(m, &b.b2, src, D, st);
    } else if (name == "adaLN_modulation.1.weight") {
      SHAPE(6 * D, D);
      rc = convert_w(m, src, (char*)m->w_ada + (size_t)l * 6 * D * D * es, 6 * D, D, st);
    } else if (name == "adaLN_modulation.1.bias") {
      SHAPE(6 * D);
      float* dst = m->b_ada + (size_t)l * 6 * D;
      rc = upload_f32(m, &dst, src, 6 * D, st);
    } else {
      set_error("set_param: unexpected key '%s'", key);
      return OSUD_ERR_ARG;
    }
  }
#undef SHAPE
  if (rc == OSUD_OK) {
    it->second = true;
    m->master[k] = src;
    m->transposed_ready = false;
  }
  return rc;
}

extern "C" int osud_dit_reserve(osud_dit* m, int max_N, int max_T, int training) {
  OSUD_CHECK_ARG(m && max_N > 0 && max_T > 0, "reserve: bad argument");
  return dit_ensure_ws(m, max_N, max_T, training != 0);
}

extern "C" int osud_dit_forward(osud_dit* m, const float* x, const int64_t* t, const float* o, const float* c,
                                const int64_t* y, const uint8_t* attn_mask, int N, int T, float cfg_scale, float* out,
                                osud_stream stream) {
  return dit_forward_impl(m, x, t, o, c, y, attn_mask, N, T, cfg_scale, true, out, false, (hipStream_t)stream);
}

// one loop iteration: timestep bookkeeping -> forward -> sampler update (x updated in place)
// The step-invariant part of the first linear, once per sampler loop: offsets and context (coordinate features zeroed) times the
// whole weight, no bias.  (x is only read for its shape here.)
static int embed_const_prepare(osud_dit* m, const float* x, const float* o, const float* c, int N, int T, bool cfg, hipStream_t st) {
  const int Tp = round_up(T, 64), Mp = round_up(N * Tp, 128);
  OSUD_TRY(launch_embed(m->prec, x, o, c, m->freqs64, m->pf[0], m->pf[1], m->e0, N, T, Tp, Mp, m->E, m->Kp, cfg ? N / 2 : 0, st, true, 2));
  return gemm(m, EPI_NONE_F32, m->e0, m->Ke, m->w_e, m->Ke, Mp, m->D, m->Ke, m->h0c, m->D, nullptr, st);
}

// The timestep-embedding MLP for every schedule index of the loop's schedule, once per loop (the steps of a loop share their t over
// the rows, and there are at most a thousand different ones): temb -> Linear + SiLU -> Linear, as in the forward.
static int tvec_table_prepare(osud_dit* m, const osud_sched* s, hipStream_t st) {
  const int nt = osud_sched_num_timesteps(s), rows = round_up(nt, 128), D = m->D;
  if (rows > m->tv_cap) {  // (first loop on this handle, or a longer schedule: the old buffers stay owned until the handle goes)
    OSUD_TRY(dev_alloc(m->owned, &m->tv_all, (size_t)rows * D * 4));
    OSUD_TRY(dev_alloc(m->owned, &m->temb_all, (size_t)rows * 256 * m->esz));
    OSUD_TRY(dev_alloc(m->owned, &m->th_all, (size_t)rows * D * m->esz));
    m->tv_cap = rows;
  }
  OSUD_TRY(launch_temb(m->prec, sched_tmap_dev(s), m->freqs128, m->temb_all, nt, rows, st));
  OSUD_TRY(gemm(m, EPI_BIAS_SILU_TE, m->temb_all, 256, m->w_t0, 256, rows, D, 256, m->th_all, D, m->b_t0, st));
  return gemm(m, EPI_BIAS_F32, m->th_all, D, m->w_t2, D, rows, D, D, m->tv_all, D, m->b_t2, st);
}

static int loop_body(osud_dit* m, const osud_sched* s, int mode, float eta, float* x, const float* o, const float* c,
                     const int64_t* y, const uint8_t* mask, int N, int T, float cfg_scale, int clip, const float* noise,
                     uint64_t seed, const osud_inpaint* inpaint, hipStream_t st) {
  OSUD_TRY(launch_step_begin(m->step_state, sched_tmap_dev(s), m->t_model, m->t_index, N, st));
  OSUD_TRY(dit_forward_impl(m, x, m->t_model, o, c, y, mask, N, T, cfg_scale, false, m->out_ws, false, st));
  OSUD_TRY(launch_sampler_step(sched_coefs(s), mode, eta, m->out_ws, x, nullptr, m->step_state, noise,
                               (size_t)N * 2 * T, seed, N, T, cfg_scale, clip, inpaint, x, nullptr, st));
  return OSUD_OK;
}

extern "C" int osud_sample_loop(osud_dit* m, const osud_sched* s, int mode, float eta, float* x, const float* o,
                                const float* c, const int64_t* y, const uint8_t* attn_mask, int N, int T,
                                float cfg_scale, int clip, int first_step, int last_step, const float* noise,
                                uint64_t seed, osud_stream stream) {
  return osud_sample_loop_inpaint(m, s, mode, eta, x, o, c, y, attn_mask, N, T, cfg_scale, clip, first_step, last_step,
                                  noise, seed, nullptr, stream);
}

// n_steps sampler steps from schedule index first_step, the index going down by `dec` per step (1: a sampling loop; 0: the same
// step again and again)
static int sample_steps(osud_dit* m, const osud_sched* s, int mode, float eta, float* x, const float* o, const float* c, const int64_t* y,
                        const uint8_t* attn_mask, int N, int T, float cfg_scale, int clip, int first_step, int n_steps, int dec,
                        const float* noise, uint64_t seed, const osud_inpaint* inpaint_in, osud_stream stream);

extern "C" int osud_sample_loop_inpaint(osud_dit* m, const osud_sched* s, int mode, float eta, float* x, const float* o,
                                        const float* c, const int64_t* y, const uint8_t* attn_mask, int N, int T,
                                        float cfg_scale, int clip, int first_step, int last_step, const float* noise,
                                        uint64_t seed, const osud_inpaint* inpaint_in, osud_stream stream) {
  OSUD_CHECK_ARG(s != nullptr, "sample_loop: null argument");
  OSUD_CHECK_ARG(first_step < osud_sched_num_timesteps(s) && last_step >= 0 && first_step >= last_step,
                 "sample_loop: steps %d..%d outside the schedule's %d steps", first_step, last_step, osud_sched_num_timesteps(s));
  return sample_steps(m, s, mode, eta, x, o, c, y, attn_mask, N, T, cfg_scale, clip, first_step, first_step - last_step + 1, 1, noise, seed,
                      inpaint_in, stream);
}

// The refine pass of sample.py:186-205 -- `refine_iters` calls of p_sample at t = 0 on the weights of a second checkpoint -- and in
// general `iters` sampler steps at ONE schedule index: the captured step of osud_sample_loop replayed with a step counter that
// stands still (the step index lives in device memory, so it is the very same graph).
extern "C" int osud_sample_repeat(osud_dit* m, const osud_sched* s, int mode, float eta, float* x, const float* o, const float* c,
                                  const int64_t* y, const uint8_t* attn_mask, int N, int T, float cfg_scale, int clip, int step,
                                  int iters, const float* noise, uint64_t seed, const osud_inpaint* inpaint, osud_stream stream) {
  OSUD_CHECK_ARG(s != nullptr, "sample_repeat: null argument");
  OSUD_CHECK_ARG(step >= 0 && step < osud_sched_num_timesteps(s) && iters >= 0, "sample_repeat: step %d (of %d), %d iterations", step,
                 osud_sched_num_timesteps(s), iters);
  if (iters == 0) return OSUD_OK;
  return sample_steps(m, s, mode, eta, x, o, c, y, attn_mask, N, T, cfg_scale, clip, step, iters, 0, noise, seed, inpaint, stream);
}

static int sample_steps(osud_dit* m, const osud_sched* s, int mode, float eta, float* x, const float* o, const float* c, const int64_t* y,
                        const uint8_t* attn_mask, int N, int T, float cfg_scale, int clip, int first_step, int n_steps, int dec,
                        const float* noise, uint64_t seed, const osud_inpaint* inpaint_in, osud_stream stream) {
  OSUD_CHECK_ARG(m && s && x && o && c && y, "sample_loop: null argument");
  OSUD_CHECK_ARG(inpaint_in == nullptr || (inpaint_in->keep && inpaint_in->known),
                 "sample_loop: in-painting needs both the keep mask and the known values");
  const osud_inpaint held = inpaint_in ? *inpaint_in : osud_inpaint{nullptr, nullptr};  // the caller's struct may be a temporary
  const osud_inpaint* inpaint = inpaint_in ? &held : nullptr;
  hipStream_t st = (hipStream_t)stream;
  OSUD_TRY(sched_upload(const_cast<osud_sched*>(s)));
  OSUD_TRY(dit_ensure_ws(m, N, T, m->training));
  OSUD_TRY(launch_step_init(m->step_state, first_step, seed, st, dec));  // the seed travels in device memory, not in the graph
  // o and c do not change over the steps of a loop: their share of the first linear is computed here, once (option embed_const = 0: off)
  struct ConstGuard {
    osud_dit* m;
    ~ConstGuard() { m->embed_const_on = m->tvec_table_on = false; }
  } const_guard{m};
  if (opt(OPT_TVEC_TABLE)) {
    OSUD_TRY(tvec_table_prepare(m, s, st));
    m->tvec_table_on = true;
  }
  {
    if ((m->split_first || m->x3) && m->h0c != nullptr && opt(OPT_EMBED_CONST)) {
      OSUD_TRY(embed_const_prepare(m, x, o, c, N, T, cfg_scale >= 0.f, st));
      m->embed_const_on = true;  // (consulted while the step is captured / run eagerly; replays of the graph read h0c)
    }
  }
  if (!opt(OPT_SAMPLE_GRAPH)) {  // eager launches: the same kernels on the same arguments (tests hold the two to bit equality)
    for (int k = 0; k < n_steps; ++k)
      OSUD_TRY(loop_body(m, s, mode, eta, x, o, c, y, attn_mask, N, T, cfg_scale, clip, noise, seed, inpaint, st));
    return OSUD_OK;
  }
  GraphKey key{N, T, mode, clip, attn_mask != nullptr, noise != nullptr, cfg_scale, eta, o, c, y, attn_mask, x, noise, s,
               held.keep, held.known, (m->embed_const_on ? 1 : 0) | (m->tvec_table_on ? 2 : 0), opt_epoch()};
  if (!(m->graph_valid && m->graph_key == key)) {
    if (m->graph_exec) {
      (void)hipGraphExecDestroy(m->graph_exec);
      m->graph_exec = nullptr;
    }
    m->graph_valid = false;
    hipGraph_t graph = nullptr;
    OSUD_HIP(hipStreamBeginCapture(m->cap_stream, hipStreamCaptureModeThreadLocal));
    const int rc = loop_body(m, s, mode, eta, x, o, c, y, attn_mask, N, T, cfg_scale, clip, noise, seed, inpaint, m->cap_stream);
    const hipError_t e = hipStreamEndCapture(m->cap_stream, &graph);
    if (rc != OSUD_OK) {
      if (graph) (void)hipGraphDestroy(graph);
      return rc;
    }
    OSUD_HIP(e);
    OSUD_HIP(hipGraphInstantiate(&m->graph_exec, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
    m->graph_key = key;
    m->graph_valid = true;
  }
  for (int k = 0; k < n_steps; ++k) OSUD_HIP(hipGraphLaunch(m->graph_exec, st));
  return OSUD_OK;
}

// fp8 inference tier: replace the static activation scales by ones measured on the caller's batch.  Runs one forward in bf16
// arithmetic recording the amax of every block's LayerNorm / attention / GELU outputs; scale = 448 / (2 * amax).  `accumulate`
// != 0 keeps the running maximum of earlier calls (several timesteps); the scales take effect at once (cached graphs are dropped).
extern "C" int osud_dit_calibrate_fp8(osud_dit* m, const float* x, const int64_t* t, const float* o, const float* c, const int64_t* y,
                                      const uint8_t* attn_mask, int N, int T, float cfg_scale, int accumulate, osud_stream stream) {
  OSUD_CHECK_ARG(m && m->fp8, "calibrate_fp8: the handle is not an fp8-tier model");
  OSUD_CHECK_ARG(!m->training, "calibrate_fp8: inference handles only (fp8 training scales follow their own amax history)");
  hipStream_t st = (hipStream_t)stream;
  const size_t nslots = (size_t)m->L * 4;
  if (!m->f8_cal_slots) OSUD_TRY(dev_alloc(m->owned, &m->f8_cal_slots, nslots * 4 * sizeof(float)));
  std::vector<float> host(nslots * 4, 0.f);
  if (accumulate) OSUD_HIP(hipMemcpy(host.data(), m->f8_cal_slots, host.size() * sizeof(float), hipMemcpyDeviceToHost));
  for (size_t i = 0; i < nslots; ++i) {
    host[4 * i] = host[4 * i + 1] = 1.0f;
    if (!accumulate) host[4 * i + 2] = 0.f;
  }
  OSUD_HIP(hipMemcpyAsync(m->f8_cal_slots, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice, st));
  OSUD_HIP(hipStreamSynchronize(st));  // `host` goes out of scope / is reused below
  float* scratch = nullptr;  // the forward's (N, C2, T) output: not needed
  OSUD_HIP(hipMalloc(&scratch, (size_t)N * m->C2 * T * sizeof(float)));
  m->f8_calibrating = true;
  const int rc = dit_forward_impl(m, x, t, o, c, y, attn_mask, N, T, cfg_scale, true, scratch, false, st);
  m->f8_calibrating = false;
  hipError_t e = hipStreamSynchronize(st);
  (void)hipFree(scratch);
  if (rc != OSUD_OK) return rc;
  OSUD_HIP(e);
  OSUD_HIP(hipMemcpy(host.data(), m->f8_cal_slots, host.size() * sizeof(float), hipMemcpyDeviceToHost));
  for (size_t i = 0; i < nslots; ++i) {
    const float amax = host[4 * i + 2];
    if (amax > 0.f && amax < 3.0e38f) m->f8_inf[i] = 448.0f / (2.0f * amax);
  }
  m->graph_valid = false;  // the scales are launch arguments of the captured kernels
  return OSUD_OK;
}

extern "C" int osud_set_gemm_dynamic_tiles(int on) {
  gemm_set_dynamic_tiles(on);
  return OSUD_OK;
}
extern "C" int osud_set_option(const char* name, int value) { return opt_set(name, value); }
extern "C" int osud_get_option(const char* name, int* value) { return opt_get(name, value); }

// ---- op-level exports ---------------------------------------------------------------------
extern "C" int osud_op_gemm(int precision, int epilogue, const void* Y, int ldy, const void* X, int ldx, int My, int Nx,
                            int K, void* out, int ldo, const float* bias, const float* gate, int ld_gate,
                            int rows_per_sample, int n_samples, osud_stream stream) {
  OSUD_CHECK_ARG(precision == OSUD_PREC_BF16 || precision == OSUD_PREC_F32 || precision == 2 /* experimental fp8 e4m3 operands */ ||
                     precision == OSUD_PREC_BF16X3 /* plane pairs [hi | lo]: ld counts logical columns */ ||
                     precision == OSUD_PREC_F16F8 /* K-blocked fp16 + e4m3 groups: ld counts logical columns */ ||
                     precision == OSUD_PREC_F16W8 /* K-blocked super-groups of 128: ld counts logical columns */ ||
                     precision == OSUD_PREC_F16 /* IEEE half operands */,
                 "op_gemm: unknown precision");
  GemmP p{};
  p.Y = Y; p.X = X; p.ldy = ldy; p.ldx = ldx; p.My = My; p.Nx = Nx; p.K = K; p.out = out; p.ldo = ldo; p.bias = bias;
  p.gate = gate; p.ld_gate = ld_gate; p.rows_per_sample = rows_per_sample; p.n_samples = n_samples;
  return launch_gemm(precision, epilogue, p, (hipStream_t)stream);
}
extern "C" int osud_op_gemm_ex(int precision, int epilogue, const void* Y, int ldy, const void* X, int ldx, int My, int Nx, int K, void* out,
                               int ldo, const float* bias, void* out2, const void* aux, int aux_code, float* colpart, osud_stream stream) {
  OSUD_CHECK_ARG(precision == OSUD_PREC_BF16 || precision == OSUD_PREC_F32, "op_gemm_ex: the training epilogues exist in the bf16 and fp32 tiers");
  GemmP p{};
  p.Y = Y; p.X = X; p.ldy = ldy; p.ldx = ldx; p.My = My; p.Nx = Nx; p.K = K; p.out = out; p.ldo = ldo; p.bias = bias;
  p.out2 = out2; p.aux = aux; p.aux_code = aux_code;
#ifdef OSUD_PH_TIMING  // (timing builds, tools/gemm_phase_stamps.py: the cycle stamps leave through the gate pointer, which these epilogues do not use)
  p.gate = colpart;
  colpart = nullptr;
#endif
  int rows = 0;
  if (colpart != nullptr) {
    p.colpart = colpart;
    p.colpart_rows = &rows;
  }
  return launch_gemm(precision, epilogue, p, (hipStream_t)stream);
}
extern "C" int osud_op_pack_w8(const float* src, int ld_src, int cols_src, void* dst, int cols_dst, int rows, int weight, osud_stream stream) {
  OSUD_CHECK_ARG(src && dst, "op_pack_w8: null argument");
  return launch_pack_rows_w8(src, ld_src, cols_src, dst, cols_dst, rows, weight != 0, (hipStream_t)stream);
}
extern "C" int osud_op_pack_h8(const float* src, int ld_src, int cols_src, void* dst, int cols_dst, int rows, int weight, osud_stream stream) {
  OSUD_CHECK_ARG(src && dst, "op_pack_h8: null argument");
  return launch_pack_rows_h8(src, ld_src, cols_src, dst, cols_dst, rows, weight != 0, (hipStream_t)stream);
}
extern "C" int osud_op_convert(int precision, const float* src, void* dst, size_t n, osud_stream stream) {
  OSUD_CHECK_ARG(src && dst, "op_convert: null argument");
  return launch_convert(precision, src, dst, n, (hipStream_t)stream);
}
extern "C" int osud_op_attention_bwd(int precision, const void* qkv, const void* d_out, const void* out, const float* lse,
                                     void* dqkv, int N, int T, int heads, int head_dim, float* delta_ws, osud_stream stream) {
  OSUD_CHECK_ARG(qkv && d_out && out && lse && dqkv, "op_attention_bwd: null argument");
  return launch_attention_bwd(precision, qkv, d_out, out, lse, dqkv, N, T, heads, head_dim, (hipStream_t)stream, delta_ws, nullptr);
}
extern "C" int osud_op_wgrad(const void* P, int ldp, const void* Q, int ldq, int Ny, int Nx, int M, float* out, float* ws, size_t ws_elems,
                             osud_stream stream) {
  OSUD_CHECK_ARG(P && Q && out && ws, "op_wgrad: null argument");
  return launch_wgrad_tr(P, ldp, Q, ldq, Ny, Nx, M, out, ws, ws_elems, (hipStream_t)stream);
}
extern "C" int osud_op_wgrad8(const void* P8, int ldp, const void* Q8, int ldq, int Ny, int Nx, int M, float* out, float* ws, size_t ws_elems,
                              const float* inv_p, const float* inv_q, osud_stream stream) {
  OSUD_CHECK_ARG(P8 && Q8 && out && ws, "op_wgrad8: null argument");
  return launch_wgrad8_tr(P8, ldp, Q8, ldq, Ny, Nx, M, out, ws, ws_elems, inv_p, inv_q, (hipStream_t)stream);
}
extern "C" int osud_op_attention(int precision, const void* qkv, int ld_qkv, const uint8_t* mask, void* out, int N,
                                 int T, int Tp, int Mp, int heads, int head_dim, osud_stream stream) {
  OSUD_CHECK_ARG(qkv && out, "op_attention: null argument");
  return launch_attention(precision, qkv, ld_qkv, mask, out, nullptr, N, T, Tp, Mp, heads, head_dim,
                          (hipStream_t)stream);  // no key-block ranges: the op-level entry scans every block
}
