// Training path: backward of the DiT forward (hand-written autograd of models.py:306-325),
// the fused diffusion loss (gaussian_diffusion.py:785-874 + :735-783, diffusion_utils.py:9-89),
// q_sample (:231-247) and the AdamW + EMA update (train.py:161,258-261, update_ema :36-45).
//
// Every matrix product of the backward pass runs through the same MFMA tile kernel as the
// forward (gemm.hip): data gradients use transposed weight copies made at pack time, weight
// gradients use transposed activation / gradient copies produced by one fused
// transpose+column-sum pass (the column sums are the bias gradients).
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "dit.h"

#pragma clang fp contract(off)

struct osud_sched;
namespace osud {
const float* sched_train_coefs(const osud_sched* s);
int sched_upload_train(osud_sched* s);
}  // namespace osud

namespace osud {

namespace {

// transposed weight copies ([in][out]) for the data-gradient products
int build_transposed(osud_dit* m, hipStream_t st) {
  if (m->transposed_ready) return OSUD_OK;
  const int D = m->D, prec = m->prec;
  const size_t es = m->esz;
  TransposeList tl{};
  auto T_ = [&](void* src, int R, int C, void** dst) -> int {
    if (!*dst) OSUD_TRY(dev_alloc(m->owned, dst, (size_t)R * C * es, false));
    if (tl.count == TransposeList::kMax) {
      OSUD_TRY(launch_transpose_many(prec, tl, st));
      tl.count = 0;
    }
    tl.src[tl.count] = src; tl.dst[tl.count] = *dst; tl.R[tl.count] = R; tl.C[tl.count] = C;
    ++tl.count;
    return OSUD_OK;
  };
  for (auto& b : m->blk) {
    OSUD_TRY(T_(b.w_qkv, 3 * D, D, &b.w_qkv_t));
    OSUD_TRY(T_(b.w_o, D, D, &b.w_o_t));
    OSUD_TRY(T_(b.w1, 4 * D, D, &b.w1_t));
    OSUD_TRY(T_(b.w2, D, 4 * D, &b.w2_t));
  }
  OSUD_TRY(T_(m->w_ada, m->ada_cols, D, &m->w_ada_t));
  OSUD_TRY(T_(m->w_t2, D, D, &m->w_t2_t));
  OSUD_TRY(launch_transpose_many(prec, tl, st));
  if (m->fp8) {  // fp8 training: e4m3 twins of the transposed weights, one scale per row (= per output column of the dgrad product)
    QuantBatch quants(true, st);
    for (auto& b : m->blk) {
      auto Q8 = [&](void* src, int rows, int cols, void** q, float** dq) -> int {
        if (!*q) OSUD_TRY(dev_alloc(m->owned, q, (size_t)rows * cols, false));
        if (!*dq) OSUD_TRY(dev_alloc(m->owned, dq, (size_t)rows * 4, false));
        if (cols % 8 == 0) return quants.add(src, rows, cols, *q, *dq);
        return launch_quantize_rows_bf16(src, rows, cols, *q, *dq, st);
      };
      OSUD_TRY(Q8(b.w_qkv_t, D, 3 * D, &b.w_qkv_t8, &b.dq_qkv_t));
      OSUD_TRY(Q8(b.w_o_t, D, D, &b.w_o_t8, &b.dq_o_t));
      OSUD_TRY(Q8(b.w1_t, D, 4 * D, &b.w1_t8, &b.dq_1_t));
      OSUD_TRY(Q8(b.w2_t, 4 * D, D, &b.w2_t8, &b.dq_2_t));
    }
    OSUD_TRY(quants.flush());
  }
  m->transposed_ready = true;
  return OSUD_OK;
}

// osud_set_option("debug_sync", 1): synchronise after every stage of the backward pass and name it (fault triage)
int dbg_sync(hipStream_t st, const char* stage) {
  if (!opt(OPT_DEBUG_SYNC)) return OSUD_OK;
  fprintf(stderr, "[osud] %s ...", stage);
  fflush(stderr);
  const hipError_t e = hipStreamSynchronize(st);
  fprintf(stderr, " %s\n", e == hipSuccess ? "ok" : hipGetErrorString(e));
  fflush(stderr);
  return e == hipSuccess ? OSUD_OK : hip_fail(e, stage, __FILE__, __LINE__);
}

// Weight-gradient product out[My][Nx] = Y[My][K] . X[Nx][K]^T with K = number of tokens: few output
// tiles, very long contraction -> split K over workgroups (partial slabs + deterministic combine).
int wgrad(osud_dit* m, const void* Y, const void* X, int My, int Nx, int K, float* out, int ldo, hipStream_t st) {
  const int tiles = (My / 128) * (Nx / 128);
  const int slabs = (int)((size_t)K * m->esz / 128);
  // fill the chip in one round: splits = CUs / tiles (uneven K ranges are fine), each at least 8 slabs
  int S = 256 / (tiles > 0 ? tiles : 1);
  if (S > 32) S = 32;
  while (S > 1 && (slabs / S < 8 || (size_t)S * My * Nx > m->bw.splitk_elems)) --S;
  if (S < 1) S = 1;
  if (S == 1 || ldo != Nx) return gemm(m, EPI_NONE_F32, Y, K, X, K, My, Nx, K, out, ldo, nullptr, st);
  GemmP p{};
  p.Y = Y; p.X = X; p.ldy = K; p.ldx = K; p.My = My; p.Nx = Nx; p.K = K;
  p.out = m->bw.splitk; p.ldo = Nx; p.split_k = S; p.split_stride = (size_t)My * Nx;
  OSUD_TRY(launch_gemm(m->prec, EPI_NONE_F32, p, st));
  return launch_splitk_reduce(m->bw.splitk, S, (size_t)My * Nx, out, (size_t)My * Nx, st);
}

// dW[Ny][Nx] = dC[:, :Ny]^T . A[:, :Nx] and db = column sums of dC.
//   bf16 tier: transpose-free kernel (wgrad.hip) + a column-sum pass;
//   f32 tier : transposes (the column sums ride along) + the generic GEMM.
int weight_grad(osud_dit* m, const void* dC, int ld_dc, const void* A, int ld_a, int Ny, int Nx, int M, float* dW,
                float* db, hipStream_t st, float* slabs = nullptr /* bf16 tier: split-K slab area (default: the main stream's) */) {
  BwdWs& w = m->bw;
  if (m->prec == OSUD_PREC_BF16) {
    OSUD_TRY(launch_wgrad_tr(dC, ld_dc, A, ld_a, Ny, Nx, M, dW, slabs ? slabs : w.splitk, w.splitk_elems, st));
    // (a call on the side stream -- it brings its own slab area -- gets its own scratch for the column sums' partial rows too)
    if (db) OSUD_TRY(launch_colsum_bf16(dC, ld_dc, M, Ny, db, st, slabs ? w.colpart2 : w.colpart, w.colpart_elems));
    return OSUD_OK;
  }
  OSUD_TRY(launch_transpose(m->prec, dC, ld_dc, w.tB, M, M, Ny, db, st, w.colpart, w.colpart_elems));
  OSUD_TRY(launch_transpose(m->prec, A, ld_a, w.tA, M, M, Nx, nullptr, st));
  return wgrad(m, w.tB, w.tA, Ny, Nx, M, dW, Nx, st);
}

float* grad_of(osud_dit* m, const std::string& key) {
  auto it = m->grad.find(key);
  return it == m->grad.end() ? nullptr : it->second;
}

}  // namespace

// Phases: 0 = zero the atomic accumulators + final layer; p in 1..L = block L-p; L+1 = first linear +
// conditioning path.  Running them in order [0, L+1] is the whole backward; the host may interleave
// gradient all-reduces of finished parameter slices between phases.
int dit_backward_impl(osud_dit* m, const float* dout, int phase_lo, int phase_hi, hipStream_t st) {
  OSUD_CHECK_ARG(m && dout, "backward: null argument");
  OSUD_CHECK_ARG(m->training && m->last_N > 0, "backward: no training forward to differentiate");
  const int N = m->last_N, T = m->last_T, D = m->D, L = m->L, prec = m->prec, AC = m->ada_cols;
  const int Tp = round_up(T, 64), M = N * Tp, Mp = round_up(M, 128), Np = round_up(N, 128);
  OSUD_CHECK_ARG(T == Tp && M == Mp, "training needs seq_len %% 64 == 0 and batch*seq_len %% 128 == 0 (got N=%d, T=%d)", N, T);
  OSUD_CHECK_ARG(phase_lo >= 0 && phase_hi <= L + 1 && phase_lo <= phase_hi, "backward: phases %d..%d outside 0..%d", phase_lo,
                 phase_hi, L + 1);
  for (auto& kv : m->have)
    if (kv.first != "xoc_embedder.playfield_size" && !grad_of(m, kv.first)) {
      set_error("backward: no gradient buffer bound for '%s'", kv.first.c_str());
      return OSUD_ERR_STATE;
    }
  OSUD_TRY(build_transposed(m, st));
  BwdWs& w = m->bw;
  const size_t es = m->esz;
  auto zero = [&](void* p, size_t bytes) -> int {
    OSUD_HIP(hipMemsetAsync(p, 0, bytes, st));
    return OSUD_OK;
  };
  auto G = [&](const std::string& k) { return grad_of(m, k); };
  // Phased invocation (the host reduces / updates finished slices between phases): the adaLN weight and bias gradients of a
  // block are produced inside that block's phase -- its 6D modulation-gradient columns are final by then -- so that they can
  // travel with the block's slice instead of waiting for the last phase.  A one-call backward keeps the single batched GEMM.
  const bool per_block_ada = !(phase_lo == 0 && phase_hi == L + 1);
  char* dada_te = (char*)w.dada_te;
  char* dada_t = dada_te + (size_t)Np * AC * es;  // [AC][Np]
  auto ada_slice = [&](int l) -> int {  // l == L: the final layer's two chunks
    const std::string key = l < L ? "blocks." + std::to_string(l) + ".adaLN_modulation.1." : "final_layer.adaLN_modulation.1.";
    const int rows = l < L ? 6 * D : 2 * D;
    const size_t off = (size_t)l * 6 * D;
    OSUD_TRY(launch_mask_rows(prec, w.dada + off, dada_te + off * es, N, Np, rows, st, AC));
    OSUD_TRY(launch_transpose(prec, dada_te + off * es, AC, dada_t + off * Np * es, Np, Np, rows, w.dbada + off, st, w.colpart, w.colpart_elems));
    OSUD_TRY(gemm(m, EPI_NONE_F32, dada_t + off * Np * es, Np, w.sb_t, Np, rows, D, Np, G(key + "weight"), D, nullptr, st));
    SegBatch cb(SEG_COPY, prec, st);
    OSUD_TRY(cb.add(w.dbada + off, G(key + "bias"), (size_t)rows / 4));
    return cb.flush();
  };

  // Per-workgroup partial rows of the LayerNorm / gate kernels (kernels.h: RowRedList): slot 2 l = block l's LN1 backward,
  // 2 l + 1 = its LN2 backward, 2 L = the final layer, 2 L + 1 = the last block's gate step; each (M / 64) x (6 D + 64) floats.  The
  // fixed-order sums over them are deferred to ONE launch per call (or per phase where a block's adaLN slice needs them at once).
  const size_t rowpart_stride = (size_t)(M / 64) * (6 * D + 64);
  auto rowpart = [&](int slot) { return w.rowpart + (size_t)slot * rowpart_stride; };
  RowRedList rr{};
  rr.D = D; rr.ld_ada = AC;
  auto rr_flush = [&]() -> int {
    const int rc = launch_row_reduce(rr, st);
    rr.count = 0;
    return rc;
  };
  // ... and the bias gradients whose partial rows come out of GEMM / attention epilogues (fc1, in_proj): one launch per call too
  ColsumList cs{};
  auto cs_flush = [&]() -> int {
    const int rc = launch_colsum_many(cs, st);
    cs.count = 0;
    return rc;
  };
  auto cs_add = [&](const float* src, int R, int C, float* out) -> int {
    if (cs.count == ColsumList::kMax) OSUD_TRY(cs_flush());
    const int i = cs.count++;
    if (i == 0) cs.blk_begin[0] = 0;
    cs.src[i] = src; cs.out[i] = out; cs.R[i] = R; cs.ld[i] = C;
    cs.blk_begin[i + 1] = cs.blk_begin[i] + C / 64;
    return OSUD_OK;
  };
  auto rr_add = [&](const float* part, int stride, int nq_sample, int off0, int off1, int off2, float* bias) -> int {
    if (rr.count == RowRedList::kMax) OSUD_TRY(rr_flush());
    const int i = rr.count++;
    rr.part[i] = part; rr.dada[i] = w.dada; rr.bias[i] = bias; rr.stride[i] = stride; rr.blocks[i] = M / 64; rr.bps[i] = Tp / 64;
    rr.nq_sample[i] = nq_sample; rr.off[i][0] = off0; rr.off[i][1] = off1; rr.off[i][2] = off2;
    return OSUD_OK;
  };

  float* dh = m->bw_dh_cur ? w.dhB : w.dhA;
  float* dh_other = m->bw_dh_cur ? w.dhA : w.dhB;
  if (phase_lo == 0) {
  // ---- nothing is accumulated by atomics (kernels.h): every gradient element has one writer -- a GEMM, or a fixed-order sum of
  // per-workgroup partial rows -- so nothing needs zeroing but the class table, to which the conditioning path ADDS the step's
  // label rows (one writer per row there too)
  OSUD_TRY(zero(G("y_embedder.embedding_table.weight"), (size_t)m->cfg.table_rows * D * 4));
  // ---- final layer
  const LayerSaved& fin = m->saved[(size_t)L];
  dh = w.dhA;
  dh_other = w.dhB;
  OSUD_TRY(launch_final_bwd(fin.h_in, fin.stats1, dout, m->w_f, m->ada, AC, L * 6 * D, L * 6 * D + D, dh,
                            G("final_layer.linear.weight"), G("final_layer.linear.bias"), N, T, Tp, D, m->C2, st, rowpart(2 * L)));
    OSUD_TRY(rr_add(rowpart(2 * L) + 4 * D + 64, 6 * D + 64, 2, L * 6 * D, L * 6 * D + D, 0, nullptr));
    OSUD_TRY(dbg_sync(st, "final_bwd"));
    if (per_block_ada) {
      OSUD_TRY(rr_flush());
      OSUD_TRY(launch_transpose(prec, m->sb, D, w.sb_t, Np, Np, D, nullptr, st));  // silu(b)^T [D][Np], shared by every slice
      OSUD_TRY(ada_slice(L));
      OSUD_TRY(dbg_sync(st, "wgrad ada (final layer)"));
    }

  }  // phase 0

  // ---- blocks, last to first
  for (int l = L - 1; l >= 0; --l) {
    const int phase = L - l;
    if (phase < phase_lo || phase > phase_hi) continue;
    const BlockWeights& bw = m->blk[(size_t)l];
    const LayerSaved& sv = m->saved[(size_t)l];
    const std::string p = "blocks." + std::to_string(l) + ".";
    const int base = l * 6 * D;
    float *g_b2 = G(p + "mlp.fc2.bias"), *g_b1 = G(p + "mlp.fc1.bias"), *g_bo = G(p + "attn.out_proj.bias"),
          *g_bqkv = G(p + "attn.in_proj_bias");
    // MLP branch: h_out = h_mid + g2 * (gelu(u2 W1^T + b1) W2^T + b2).  Its gate step (dbr = g2 * dh, dg2, db2) was done
    // by the kernel that produced dh: final_bwd's successor below for the last block, the LN1 backward of block l+1 otherwise.
    if (l == L - 1) {
      OSUD_TRY(launch_gate_bwd(prec, dh, sv.br2, m->ada + base + 5 * D, AC, w.dbr, rowpart(2 * L + 1), M, Tp, D, st));
      OSUD_TRY(rr_add(rowpart(2 * L + 1), 2 * D, 1, base + 5 * D, 0, 0, g_b2));
      OSUD_TRY(dbg_sync(st, "gate_bwd mlp"));
    }
    // dz1 = (dbr . W2) * gelu'(z1); in the bf16 tier the fc1 bias gradient (column sums of dz1) rides in the same epilogue
    // as per-wave-row partial sums (scratch: the split-K slab area, free until the weight gradients below)
    const bool fused_b1 = prec == OSUD_PREC_BF16;
    float* b1part = fused_b1 ? w.b1part + (size_t)l * w.b1part_stride : nullptr;  // this layer's partial rows (summed at the end of the call; allocated where fused_b1 holds: dit.hip)
    float* bqkvpart = w.bqkvpart + (size_t)l * w.bqkvpart_stride;
    // fp8 training: the data-gradient products of fc2, fc1 and in_proj run on e4m3 operands (gradient tensors quantised with the
    // scale from their previous step's amax, transposed weights per row); the very first step only records (see dit_forward_impl)
    const bool f8_train = m->fp8, f8_live = m->fp8 && m->f8_steps > 1;
    const bool f8_slim = f8_twins_only(m, f8_live, Mp);  // (implies the three weight_grad8 conditions below)
    auto slot = [&](int which) { return m->f8_slots + ((size_t)l * kF8Slots + which) * 4; };
    // fp8 training, live steps: a weight gradient from the e4m3 twins of its two operands (twice the bf16 kernel's rate); dW is
    // de-quantised by the two slots' 1 / scale.  P8: the gradient twin in its staging buffer, Q8: the layer's saved activation twin
    auto weight_grad8 = [&](const void* P8, int ldp, int slot_p, const void* Q8, int ldq, int slot_q, int Ny, int Nx, float* dW,
                            hipStream_t s8 = nullptr /* the side stream (its own slab area) */) -> int {
      return launch_wgrad8_tr(P8, ldp, Q8, ldq, Ny, Nx, Mp, dW, s8 ? w.splitk2 : w.splitk, w.splitk_elems, slot(slot_p) + 1, slot(slot_q) + 1,
                              s8 ? s8 : st);
    };
    // bf16 tier: the four weight gradients of the block leave the data-gradient chain and run on a side stream (their own slab
    // area), each as soon as its gradient operand exists; the chain -- data-gradient GEMMs with the HBM-bound LayerNorm / attention
    // backward kernels between them -- goes on at once, and waits for the side stream only before the kernel that overwrites the
    // branch gradients (the LN1 backward).  The weight gradients then run under the chain's HBM-bound kernels, which leave the
    // matrix pipe -- and the power budget, DESIGN.md section 4.1 -- idle: 12 blocks of the pattern with device copies standing in
    // for the HBM-bound kernels 14.2 -> 13.2 ms (tools/overlap_wgrad_probe.py); the real step 26.2 -> 25.7 ms on one box, 26.0 -> 25.8
    // on another (five alternating pairs, every one in favour).  Less than the stand-in promised: a LayerNorm backward fills every
    // compute unit's registers, so a weight-gradient workgroup only starts where its blocks have finished -- the gain is kernel
    // heads and tails filling each other, not two kernels sharing compute units.  osud_set_option("wgrad_side_stream", 0) restores
    // the single stream (same bits: tests/test_gpu_train.py; bench.py takes its per-kernel table there).
    // (Built, gradients equal to 2e-7, measured neutral and removed: the four products of a block in one or two GROUPED launches --
    //  equal runs of stages per workgroup, one combine pass per group; HISTORY.md section 4 "round 3".)
    const bool side_env = opt(OPT_WGRAD_SIDE_STREAM) != 0;  // (read per block)
    bool side_on = side_env && prec == OSUD_PREC_BF16 && !f8_train;
    // (fp8 training, live steps: the same for the e4m3 weight gradients, with one more join -- the twin of dqkv re-uses the staging
    //  buffer the fc1 weight gradient reads dz1's twin from)
    bool side8 = side_env && prec == OSUD_PREC_BF16 && f8_live && fused_b1 && Mp % 128 == 0;
    if ((side_on || side8) && w.side == nullptr) {
      bool ok = hipStreamCreateWithFlags(&w.side, hipStreamNonBlocking) == hipSuccess;
      if (!ok) w.side = nullptr;
      for (hipEvent_t& e : w.side_ev)
        if (ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { e = nullptr; ok = false; }
      if (!ok) side_on = side8 = false;
    }
    if (w.side == nullptr || w.side_ev[3] == nullptr) side_on = side8 = false;
    auto side_after = [&](int e) -> int {  // the side stream continues behind everything the chain has enqueued so far
      OSUD_HIP(hipEventRecord(w.side_ev[e], st));
      OSUD_HIP(hipStreamWaitEvent(w.side, w.side_ev[e], 0));
      w.side_busy = true;
      return OSUD_OK;
    };
    // side -> chain: the chain joins the side stream before the LN1 backward, the first kernel that overwrites an operand of this block's
    // weight gradients (dbr; the next block then overwrites dz1, dbr2, dqkv).  Measured against per-buffer events that let in_proj's
    // weight gradient run on under the LN1 backward and the next block's first data gradient: 25.6-25.8 vs 26.0 ms per step (single
    // stream 26.2) -- two MFMA kernels side by side only take compute units from each other.
    auto chain_joins_side = [&]() -> int {
      OSUD_HIP(hipEventRecord(w.side_ev[3], w.side));
      OSUD_HIP(hipStreamWaitEvent(st, w.side_ev[3], 0));
      w.side_busy = false;
      return OSUD_OK;
    };
    {
      int part_rows = 0;
      // (the e4m3 twin of dbr and its amax come from the kernel that produced dbr -- the LN1 backward of block l + 1 -- except
      //  for the last block, whose gate step is its own kernel)
      if (f8_train && l == L - 1) OSUD_TRY(launch_f8_quantize(w.dbr, f8_live ? m->q8a : nullptr, (size_t)Mp * D, slot(3), st));
      if (f8_live) {
        OSUD_TRY(gemm8(m, EPI_GELUGRAD_TE, m->q8a, bw.w2_t8, Mp, 4 * D, D, f8_slim ? nullptr : w.dz1, 4 * D, nullptr, bw.dq_2_t, 0.f, st, nullptr, 0, 0, 0, 0.f,
                       slot(3) + 1, nullptr, sv.z1, fused_b1 ? b1part : nullptr, fused_b1 ? &part_rows : nullptr, m->q8b, slot(4)));
      } else {
      GemmP gp{};
      gp.Y = w.dbr; gp.X = bw.w2_t; gp.ldy = D; gp.ldx = D; gp.My = Mp; gp.Nx = 4 * D; gp.K = D;
      gp.out = w.dz1; gp.ldo = 4 * D; gp.aux = sv.z1; gp.aux_code = m->z1_code ? 1 : 0;
      if (fused_b1) {
        gp.colpart = b1part;
        gp.colpart_rows = &part_rows;
      }
      OSUD_TRY(launch_gemm(prec, EPI_GELUGRAD_TE, gp, st));
      }
      if (fused_b1) OSUD_TRY(cs_add(b1part, part_rows, 4 * D, g_b1));
    }
    OSUD_TRY(dbg_sync(st, "dgrad fc2 (gelu grad)"));
    hipStream_t ws_ = side_on ? w.side : st;
    float* slabs_ = side_on ? w.splitk2 : nullptr;
    auto wg_mlp = [&]() -> int {
      OSUD_TRY(weight_grad(m, w.dz1, 4 * D, sv.u2, D, 4 * D, D, Mp, G(p + "mlp.fc1.weight"), fused_b1 ? nullptr : g_b1, ws_, slabs_));
      OSUD_TRY(dbg_sync(ws_, "wgrad fc1"));
      OSUD_TRY(weight_grad(m, w.dbr, D, sv.g, 4 * D, D, 4 * D, Mp, G(p + "mlp.fc2.weight"), nullptr, ws_, slabs_));
      return dbg_sync(ws_, "wgrad fc2");
    };
    auto wg_proj = [&]() -> int {
      OSUD_TRY(weight_grad(m, w.dbr2, D, sv.ao, D, D, D, Mp, G(p + "attn.out_proj.weight"), nullptr, ws_, slabs_));
      return dbg_sync(ws_, "wgrad out_proj");
    };
    if (side_on) {  // dz1 and this block's dbr exist: fc1 / fc2 weight gradients start next to the fc1 data gradient (starting them
                    // behind it, under the LN2 backward, measured no better than the single stream)
      OSUD_TRY(side_after(0));
      OSUD_TRY(wg_mlp());
    }
    if (side8) {  // (q8b = dz1's twin, q8a = the MLP branch gradient's)
      OSUD_TRY(side_after(0));
      OSUD_TRY(weight_grad8(m->q8b, 4 * D, 4, sv.u2_8, D, 1, 4 * D, D, G(p + "mlp.fc1.weight"), w.side));
      OSUD_TRY(weight_grad8(m->q8a, D, 3, sv.g_8, 4 * D, 2, D, 4 * D, G(p + "mlp.fc2.weight"), w.side));
      OSUD_TRY(dbg_sync(w.side, "wgrad fc1, fc2 (e4m3, side stream)"));
    }
    // consumers of dz1 (201 MB, fresh in the Infinity Cache) first, the fc2 weight gradient (dbr, g) after them
    if (f8_train && !f8_live) OSUD_TRY(launch_f8_quantize(w.dz1, nullptr, (size_t)Mp * 4 * D, slot(4), st));  // live: written by the epilogue above
    if (f8_live) OSUD_TRY(gemm8(m, EPI_NONE_TE, m->q8b, bw.w1_t8, Mp, D, 4 * D, w.du, D, nullptr, bw.dq_1_t, 0.f, st, nullptr, 0, 0, 0, 0.f,
                                slot(4) + 1));
    else
    OSUD_TRY(gemm(m, EPI_NONE_TE, w.dz1, 4 * D, bw.w1_t, 4 * D, Mp, D, 4 * D, w.du, D, nullptr, st));
    OSUD_TRY(dbg_sync(st, "dgrad fc1"));
    if (side8) {  // (enqueued on the side stream behind the fc2 data gradient, above)
    } else if (f8_live && fused_b1 && Mp % 128 == 0) {  // (q8b = dz1's twin, q8a = the MLP branch gradient's: both still in place)
      OSUD_TRY(weight_grad8(m->q8b, 4 * D, 4, sv.u2_8, D, 1, 4 * D, D, G(p + "mlp.fc1.weight")));
      OSUD_TRY(weight_grad8(m->q8a, D, 3, sv.g_8, 4 * D, 2, D, 4 * D, G(p + "mlp.fc2.weight")));
      OSUD_TRY(dbg_sync(st, "wgrad fc1, fc2 (e4m3)"));
    } else if (!side_on) {
      OSUD_TRY(wg_mlp());  // (side stream: enqueued right behind the fc2 data gradient, above)
    }
    // LN2 backward -> dh = grad wrt h_mid, and on the same rows the gate step of the attention branch
    // (h_mid = h_in + g1 * (attn(u1) Wo^T + bo)): dbr = g1 * dh, dg1, dbo
    // (fp8 training: the attention branch's gradient gets its e4m3 twin -- q8c -- and amax from this kernel too)
    OSUD_TRY(launch_ln_mod_bwd(prec, sv.h_mid, sv.stats2, w.du, m->ada, AC, base + 3 * D, base + 4 * D, dh, dh_other, rowpart(2 * l + 1),
                               M, Tp, D, st, sv.br1, base + 2 * D, f8_slim ? nullptr : w.dbr2, f8_live ? m->q8c : nullptr, f8_train ? slot(7) : nullptr,
                               f8_train ? m->f8_parts + ((size_t)l * kF8Slots + 7) * f8_amax_parts() : nullptr));
    OSUD_TRY(rr_add(rowpart(2 * l + 1), 4 * D, 3, base + 3 * D, base + 4 * D, base + 2 * D, g_bo));
    OSUD_TRY(dbg_sync(st, "ln2 bwd + gate_bwd attn"));
    std::swap(dh, dh_other);
    if (side_on) {  // dbr2 exists: out_proj's weight gradient starts next to its data gradient
      OSUD_TRY(side_after(1));
      OSUD_TRY(wg_proj());
    }
    if (side8) {
      OSUD_TRY(side_after(1));
      OSUD_TRY(weight_grad8(m->q8c, D, 7, sv.ao_8, D, 6, D, D, G(p + "attn.out_proj.weight"), w.side));
      OSUD_TRY(dbg_sync(w.side, "wgrad out_proj (e4m3, side stream)"));
    }
    if (f8_live) OSUD_TRY(gemm8(m, EPI_NONE_TE, m->q8c, bw.w_o_t8, Mp, D, D, w.dao, D, nullptr, bw.dq_o_t, 0.f, st, nullptr, 0, 0, 0, 0.f, slot(7) + 1));
    else
    OSUD_TRY(gemm(m, EPI_NONE_TE, w.dbr2, D, bw.w_o_t, D, Mp, D, D, w.dao, D, nullptr, st));
    if (side8) {  // (enqueued on the side stream behind the LN2 backward, above)
    } else if (f8_live && Mp % 128 == 0) {
      OSUD_TRY(weight_grad8(m->q8c, D, 7, sv.ao_8, D, 6, D, D, G(p + "attn.out_proj.weight")));
      OSUD_TRY(dbg_sync(st, "wgrad out_proj (e4m3)"));
    } else if (!side_on) {
      OSUD_TRY(wg_proj());  // (side stream: enqueued right behind the LN2 backward, above)
    }
    // (bf16 tier: the in_proj bias gradient is the attention backward's job -- inside the streamed kernel at T = 128, a column-sum
    //  pass over dqkv behind the other kernels; scratch: the split-K slab area, idle between two weight gradients)
    const bool fused_bqkv = prec == OSUD_PREC_BF16;
    // (fp8 training: dqkv's bias gradient, its e4m3 twin and its amax come from ONE pass over it instead of the attention
    //  backward's column-sum pass + a quantisation pass)
    {
      int pending = 0;  // (T = 128: the streamed kernel leaves one row of column sums per sample; their sum joins the call's batch)
      OSUD_TRY(launch_attention_bwd(prec, sv.qk, w.dao, sv.ao, sv.lse, w.dqkv, N, T, m->H, m->hd, st, w.attn_delta,
                                    (fused_bqkv && !f8_train) ? g_bqkv : nullptr, bqkvpart, w.bqkvpart_stride, &pending));
      if (pending > 0) OSUD_TRY(cs_add(bqkvpart, pending, 3 * D, g_bqkv));
    }
    OSUD_TRY(dbg_sync(st, "attention bwd"));
    auto wg_qkv = [&]() -> int {
      OSUD_TRY(weight_grad(m, w.dqkv, 3 * D, sv.u1, D, 3 * D, D, Mp, G(p + "attn.in_proj_weight"), fused_bqkv ? nullptr : g_bqkv, ws_, slabs_));
      return dbg_sync(ws_, "wgrad qkv");
    };
    if (side_on) {  // dqkv exists: in_proj's weight gradient starts next to its data gradient
      OSUD_TRY(side_after(2));
      OSUD_TRY(wg_qkv());
    }
    if (side8) OSUD_TRY(chain_joins_side());  // (q8b: dz1's twin, read by fc1's weight gradient, is overwritten by dqkv's twin next)
    if (f8_train) OSUD_TRY(launch_colsum_quant_bf16(w.dqkv, Mp, 3 * D, g_bqkv, f8_live ? m->q8b : nullptr, slot(5), st, w.colpart, w.colpart_elems));
    if (side8) {  // q8b = dqkv's twin: in_proj's weight gradient starts next to its data gradient
      OSUD_TRY(side_after(2));
      OSUD_TRY(weight_grad8(m->q8b, 3 * D, 5, sv.u1_8, D, 0, 3 * D, D, G(p + "attn.in_proj_weight"), w.side));
      OSUD_TRY(dbg_sync(w.side, "wgrad in_proj (e4m3, side stream)"));
    }
    if (f8_live) OSUD_TRY(gemm8(m, EPI_NONE_TE, m->q8b, bw.w_qkv_t8, Mp, D, 3 * D, w.du, D, nullptr, bw.dq_qkv_t, 0.f, st, nullptr, 0, 0, 0, 0.f,
                                slot(5) + 1));
    else
    OSUD_TRY(gemm(m, EPI_NONE_TE, w.dqkv, 3 * D, bw.w_qkv_t, 3 * D, Mp, D, 3 * D, w.du, D, nullptr, st));
    OSUD_TRY(dbg_sync(st, "dgrad qkv"));
    if (side8) {  // (enqueued on the side stream behind the dqkv pass, above)
    } else if (f8_live && fused_bqkv && Mp % 128 == 0) {  // (q8b = dqkv's twin)
      OSUD_TRY(weight_grad8(m->q8b, 3 * D, 5, sv.u1_8, D, 0, 3 * D, D, G(p + "attn.in_proj_weight")));
      OSUD_TRY(dbg_sync(st, "wgrad in_proj (e4m3)"));
    } else if (!side_on) {
      OSUD_TRY(wg_qkv());  // (side stream: enqueued right behind the attention backward, above)
    }
    if (side_on || side8) OSUD_TRY(chain_joins_side());  // (fp8: the LN1 backward also rewrites q8a, the twin fc2's weight gradient reads)
    // LN1 backward -> dh = grad wrt h_in = grad wrt the output of block l-1, whose MLP gate step rides along
    if (l > 0) {
      const LayerSaved& svp = m->saved[(size_t)l - 1];
      const int basep = (l - 1) * 6 * D;
      float* slot_prev = m->f8_slots + ((size_t)(l - 1) * kF8Slots + 3) * 4;  // fp8 training: block l - 1's dbr slot
      OSUD_TRY(launch_ln_mod_bwd(prec, sv.h_in, sv.stats1, w.du, m->ada, AC, base, base + D, dh, dh_other, rowpart(2 * l), M, Tp, D,
                                 st, svp.br2, basep + 5 * D, f8_slim ? nullptr : w.dbr,
                                 f8_live ? m->q8a : nullptr, f8_train ? slot_prev : nullptr,
                                 f8_train ? m->f8_parts + ((size_t)(l - 1) * kF8Slots + 3) * f8_amax_parts() : nullptr));
      OSUD_TRY(rr_add(rowpart(2 * l), 4 * D, 3, base, base + D, basep + 5 * D, G("blocks." + std::to_string(l - 1) + ".mlp.fc2.bias")));
    } else {
      OSUD_TRY(launch_ln_mod_bwd(prec, sv.h_in, sv.stats1, w.du, m->ada, AC, base, base + D, dh, dh_other, rowpart(2 * l), M, Tp, D,
                                 st));
      OSUD_TRY(rr_add(rowpart(2 * l), 2 * D, 2, base, base + D, 0, nullptr));
    }
    OSUD_TRY(dbg_sync(st, "ln1 bwd"));
    std::swap(dh, dh_other);  // dh = grad wrt h_in
    if (per_block_ada) {
      OSUD_TRY(cs_flush());
      OSUD_TRY(rr_flush());  // (the slice's modulation-gradient columns are final now; the two bias gradients travel with the phase)
      OSUD_TRY(ada_slice(l));
      OSUD_TRY(dbg_sync(st, "wgrad ada (block)"));
    }
  }
  OSUD_TRY(cs_flush());
  OSUD_TRY(rr_flush());  // one launch for every LayerNorm / gate kernel of this call

  if (w.side != nullptr && w.side_busy) {  // (not reached: every block joins) the caller's stream owns every gradient again
    OSUD_HIP(hipEventRecord(w.side_ev[3], w.side));
    OSUD_HIP(hipStreamWaitEvent(st, w.side_ev[3], 0));
    w.side_busy = false;
  }
  m->bw_dh_cur = dh == w.dhB ? 1 : 0;
  if (phase_hi < L + 1) return OSUD_OK;

  // ---- token embedding linear: h0 = e0 We^T + be   (inputs need no gradient)
  {
    float* g_be = G("xoc_embedder.mlp.0.bias");
    OSUD_TRY(launch_transpose_f32(prec, dh, D, w.tB, Mp, Mp, D, g_be, st, w.colpart, w.colpart_elems));
    OSUD_TRY(launch_transpose(prec, m->e0, m->Ke, w.tA, Mp, Mp, m->Kp, nullptr, st));  // the hi part of a split row
    OSUD_TRY(wgrad(m, w.tB, w.tA, D, m->Kp, Mp, w.dWe, m->Kp, st));
    OSUD_TRY(launch_unpad_rows(w.dWe, m->Kp, G("xoc_embedder.mlp.0.weight"), 384 + m->E, D, st));
    OSUD_TRY(dbg_sync(st, "first layer"));
  }

  // ---- conditioning path: ada = silu(b) Wada^T + bada ; b = t_emb + table[y]
  {
    if (!per_block_ada) {
    OSUD_TRY(launch_mask_rows(prec, w.dada, dada_te, N, Np, AC, st));
    OSUD_TRY(launch_transpose(prec, dada_te, AC, dada_t, Np, Np, AC, w.dbada, st, w.colpart, w.colpart_elems));
    OSUD_TRY(launch_transpose(prec, m->sb, D, w.small_t1, Np, Np, D, nullptr, st));  // sb^T [D][Np]
    {
      // one product for every block's adaLN weight gradient; its 6D-row panels land straight in the per-block gradient tensors
      // (a staging buffer + 13 segment copies of 170 MB cost 0.18 ms per DiT-B step)
      const bool direct = L + 1 <= 32;
      GemmP gp{};
      gp.Y = dada_t; gp.X = w.small_t1; gp.ldy = Np; gp.ldx = Np; gp.My = AC; gp.Nx = D; gp.K = Np; gp.out = w.dWada; gp.ldo = D;
      SegBatch cb(SEG_COPY, prec, st);
      float* tbl[32] = {};
      for (int l = 0; l <= L; ++l) {
        const std::string key = l < L ? "blocks." + std::to_string(l) + ".adaLN_modulation.1." : "final_layer.adaLN_modulation.1.";
        const size_t rows = l < L ? 6 * (size_t)D : 2 * (size_t)D, off = (size_t)l * 6 * D;
        if (direct) tbl[l] = G(key + "weight");
        else OSUD_TRY(cb.add(w.dWada + off * D, G(key + "weight"), rows * D / 4));
        OSUD_TRY(cb.add(w.dbada + off, G(key + "bias"), rows / 4));
      }
      if (direct) {
        if (memcmp(tbl, w.seg_tbl_host, sizeof(tbl)) != 0) {  // first use, or a gradient tensor was re-bound
          memcpy(w.seg_tbl_host, tbl, sizeof(tbl));
          OSUD_HIP(hipMemcpyAsync(w.seg_tbl, w.seg_tbl_host, sizeof(tbl), hipMemcpyHostToDevice, st));
        }
        gp.seg_rows = 6 * D;
        gp.seg_out = w.seg_tbl;
      }
      OSUD_TRY(launch_gemm(prec, EPI_NONE_F32, gp, st));
      OSUD_TRY(dbg_sync(st, "wgrad ada"));
      OSUD_TRY(cb.flush());
    }
    }  // else: every slice was converted and differentiated in its own phase; dada_te is complete
    OSUD_TRY(wgrad(m, dada_te, m->w_ada_t, Np, D, AC, w.dsb, D, st));  // K = 6(L+1)D: split over workgroups
    OSUD_TRY(launch_cond_bwd(prec, w.dsb, m->bvec, m->last_y, m->cfg.table_rows, w.db, w.db_te,
                             G("y_embedder.embedding_table.weight"), N, Np, D, st));
    OSUD_TRY(dbg_sync(st, "cond bwd"));
    // TimestepEmbedder: tvec = silu(temb W0^T + b0) W2^T + b2
    float *g_bt2 = G("t_embedder.mlp.2.bias"), *g_bt0 = G("t_embedder.mlp.0.bias");
    OSUD_TRY(launch_transpose(prec, w.db_te, D, w.small_t1, Np, Np, D, g_bt2, st, w.colpart, w.colpart_elems));  // db^T [D][Np]
    OSUD_TRY(launch_transpose(prec, m->th, D, w.small_t2, Np, Np, D, nullptr, st));  // th^T [D][Np]
    OSUD_TRY(gemm(m, EPI_NONE_F32, w.small_t1, Np, w.small_t2, Np, D, D, Np, G("t_embedder.mlp.2.weight"), D, nullptr, st));
    OSUD_TRY(gemm(m, EPI_NONE_F32, w.db_te, D, m->w_t2_t, D, Np, D, D, w.dth, D, nullptr, st));
    OSUD_TRY(launch_silu_bwd(prec, w.dth, m->z0, w.dz0, (size_t)Np * D, st));
    OSUD_TRY(launch_transpose(prec, w.dz0, D, w.small_t1, Np, Np, D, g_bt0, st, w.colpart, w.colpart_elems));  // dz0^T [D][Np]
    OSUD_TRY(launch_transpose(prec, m->temb, 256, w.small_t2, Np, Np, 256, nullptr, st));  // temb^T [256][Np]
    OSUD_TRY(gemm(m, EPI_NONE_F32, w.small_t1, Np, w.small_t2, Np, D, 256, Np, G("t_embedder.mlp.0.weight"), 256, nullptr, st));
    OSUD_TRY(dbg_sync(st, "t-embedder"));
  }
  return OSUD_OK;
}

// ------------------------------------------------------------------------------------------
namespace {

// x_t = sqrt(ac_t) x0 + sqrt(1 - ac_t) noise        (gaussian_diffusion.py:231-247)
__global__ void q_sample_kernel(const float* __restrict__ tc, const float* __restrict__ x0, const int64_t* __restrict__ t,
                                const float* __restrict__ noise, float* __restrict__ xt, int N, int CT) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * CT) return;
  const float* c = tc + (size_t)t[i / CT] * 8;
  xt[i] = c[0] * x0[i] + c[1] * noise[i];
}

__device__ __forceinline__ float approx_cdf(float v) {  // diffusion_utils.py:38-43
  return 0.5f * (1.0f + tanhf(0.7978845608028654f * (v + 0.044715f * (v * v * v))));
}
__device__ __forceinline__ float approx_cdf_grad(float v) {
  const float th = tanhf(0.7978845608028654f * (v + 0.044715f * (v * v * v)));
  return 0.5f * (1.0f - th * th) * 0.7978845608028654f * (1.0f + 3.0f * 0.044715f * v * v);
}

// One block per sample.  terms[0][n] = main (l1 | mse), terms[1][n] = vb, terms[2][n] = loss;
// dout[n] = d(mean_n loss)/d(model_out[n]).  The vb term sees eps DETACHED (gaussian_diffusion.py:833),
// so channels 0:2 get only the l1/mse gradient and channels 2:4 only the vb gradient.
__global__ __launch_bounds__(256) void train_loss_kernel(const float* __restrict__ tc, int use_l1,
                                                         const float* __restrict__ out, const float* __restrict__ x0,
                                                         const float* __restrict__ xt, const float* __restrict__ noise,
                                                         const int64_t* __restrict__ t, float* __restrict__ terms,
                                                         float* __restrict__ dout, int N, int T) {
  __shared__ float red[2][4];
  const int n = blockIdx.x, tid = threadIdx.x;
  const int CT = 2 * T;
  const int step = (int)t[n];
  const float* c = tc + (size_t)step * 8;
  const float A = c[2], B = c[3], c1 = c[4], c2 = c[5], min_log = c[6], max_log = c[7];
  const float inv_ln2 = 1.4426950408889634f;
  const float wgt = 1.0f / ((float)CT * (float)N);  // d(mean over batch of mean_flat)
  float s_main = 0.f, s_vb = 0.f;
  for (int i = tid; i < CT; i += 256) {
    const int ch = i / T, tt = i % T;
    const size_t io = ((size_t)n * 4 + ch) * T + tt, iv = ((size_t)n * 4 + 2 + ch) * T + tt, ix = (size_t)n * CT + i;
    const float eps = out[io], v = out[iv], x_start = x0[ix], x_t = xt[ix], nz = noise[ix];
    // main term
    const float diff = nz - eps;
    float g_eps;
    if (use_l1) {
      s_main += fabsf(diff);
      g_eps = diff > 0.f ? -1.0f : (diff < 0.f ? 1.0f : 0.0f);
    } else {
      s_main += diff * diff;
      g_eps = -2.0f * diff;
    }
    dout[io] = g_eps * wgt;
    // vb term (p_mean_variance with clip_denoised=False on the frozen eps)
    const float frac = (v + 1.0f) / 2.0f;
    const float lv = frac * max_log + (1.0f - frac) * min_log;
    const float xs = A * x_t - B * eps;
    const float mean = c1 * xs + c2 * x_t;
    const float true_mean = c1 * x_start + c2 * x_t;
    float term, g_lv;
    if (step == 0) {  // decoder NLL, discretized Gaussian (diffusion_utils.py:63-89)
      const float ls = 0.5f * lv;
      const float cx = x_start - mean;
      const float inv = expf(-ls);
      const float pin = inv * (cx + 1.0f / 255.0f), nin = inv * (cx - 1.0f / 255.0f);
      const float cp = approx_cdf(pin), cm = approx_cdf(nin);
      float lp, g_ls;  // d lp / d ls ; d pin/d ls = -pin, d nin/d ls = -nin
      if (x_start < -0.999f) {
        lp = logf(fmaxf(cp, 1e-12f));
        g_ls = cp > 1e-12f ? approx_cdf_grad(pin) * (-pin) / cp : 0.f;
      } else if (x_start > 0.999f) {
        const float om = 1.0f - cm;
        lp = logf(fmaxf(om, 1e-12f));
        g_ls = om > 1e-12f ? -approx_cdf_grad(nin) * (-nin) / om : 0.f;
      } else {
        const float dl = cp - cm;
        lp = logf(fmaxf(dl, 1e-12f));
        g_ls = dl > 1e-12f ? (approx_cdf_grad(pin) * (-pin) - approx_cdf_grad(nin) * (-nin)) / dl : 0.f;
      }
      term = -lp;
      g_lv = -g_ls * 0.5f;
    } else {  // KL(q(x_{t-1}|x_t,x_0) || p)   (diffusion_utils.py:9-35)
      const float dm = true_mean - mean;
      const float e1 = expf(min_log - lv), e2 = expf(-lv);
      term = 0.5f * (-1.0f + lv - min_log + e1 + (dm * dm) * e2);
      g_lv = 0.5f * (1.0f - e1 - (dm * dm) * e2);
    }
    s_vb += term;
    dout[iv] = g_lv * 0.5f * (max_log - min_log) * inv_ln2 * wgt;
  }
  s_main = wave_sum(s_main);
  s_vb = wave_sum(s_vb);
  if ((tid & 63) == 0) {
    red[0][tid >> 6] = s_main;
    red[1][tid >> 6] = s_vb;
  }
  __syncthreads();
  if (tid == 0) {
    const float mainv = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / (float)CT;
    const float vb = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / (float)CT * inv_ln2;
    terms[n] = mainv;
    terms[N + n] = vb;
    terms[2 * N + n] = mainv + vb;
  }
}

// AdamW (torch.optim.AdamW semantics, decoupled weight decay) + EMA in one pass over flat arenas.
// Elements in [skip_begin, skip_end) (the frozen playfield_size parameter) get only the EMA update.
struct AdamC {
  float lr, beta1, beta2, eps, wd, bc1, bc2_sqrt, decay, grad_scale;
};
__device__ __forceinline__ void adamw_one(float& pv, float gv, float& a, float& b, float& e, const AdamC& c, bool frozen, bool has_ema) {
  if (!frozen) {
    gv *= c.grad_scale;
    pv = pv * (1.0f - c.lr * c.wd);
    a = a * c.beta1 + (1.0f - c.beta1) * gv;  // exp_avg.lerp_(grad, 1 - beta1)
    b = b * c.beta2 + (1.0f - c.beta2) * gv * gv;
    const float denom = sqrtf(b) / c.bc2_sqrt + c.eps;
    pv = pv - (c.lr / c.bc1) * (a / denom);
  }
  if (has_ema) e = e * c.decay + pv * (1.0f - c.decay);  // update_ema, train.py:36-45
}
// Nine 4-byte streams per element (p, g, m1, m2, ema in; p, m1, m2, ema out): HBM-bound.  16 bytes per lane and stream in the
// body; `head` scalar elements in front bring all five arrays (same element offset into equally aligned arenas) to a 16-byte
// boundary, the remainder is scalar again.
__global__ __launch_bounds__(256) void adamw_ema_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m1,
                                                        float* __restrict__ m2, float* __restrict__ ema, size_t n, AdamC c,
                                                        size_t skip_begin, size_t skip_end, size_t head) {
  const bool has_ema = ema != nullptr;
  const size_t n4 = (n - head) / 4, tail0 = head + n4 * 4;
  const size_t gid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, gstride = (size_t)gridDim.x * blockDim.x;
  for (size_t v = gid; v < n4; v += gstride) {
    const size_t i = head + v * 4;
    float4 pv = *reinterpret_cast<float4*>(p + i);
    const float4 gv = *reinterpret_cast<const float4*>(g + i);
    float4 a = *reinterpret_cast<float4*>(m1 + i), b = *reinterpret_cast<float4*>(m2 + i);
    float4 e = has_ema ? *reinterpret_cast<float4*>(ema + i) : make_float4(0.f, 0.f, 0.f, 0.f);
    const bool any_frozen = i < skip_end && i + 4 > skip_begin;
    adamw_one(pv.x, gv.x, a.x, b.x, e.x, c, any_frozen && i + 0 >= skip_begin && i + 0 < skip_end, has_ema);
    adamw_one(pv.y, gv.y, a.y, b.y, e.y, c, any_frozen && i + 1 >= skip_begin && i + 1 < skip_end, has_ema);
    adamw_one(pv.z, gv.z, a.z, b.z, e.z, c, any_frozen && i + 2 >= skip_begin && i + 2 < skip_end, has_ema);
    adamw_one(pv.w, gv.w, a.w, b.w, e.w, c, any_frozen && i + 3 >= skip_begin && i + 3 < skip_end, has_ema);
    *reinterpret_cast<float4*>(p + i) = pv;
    *reinterpret_cast<float4*>(m1 + i) = a;
    *reinterpret_cast<float4*>(m2 + i) = b;
    if (has_ema) *reinterpret_cast<float4*>(ema + i) = e;
  }
  for (size_t k = gid; k < head + (n - tail0); k += gstride) {  // unaligned head and tail, one element per thread
    const size_t i = k < head ? k : tail0 + (k - head);
    float pv = p[i], a = m1[i], b = m2[i], e = has_ema ? ema[i] : 0.f;
    adamw_one(pv, g[i], a, b, e, c, i >= skip_begin && i < skip_end, has_ema);
    p[i] = pv;
    m1[i] = a;
    m2[i] = b;
    if (has_ema) ema[i] = e;
  }
}

}  // namespace
}  // namespace osud

using namespace osud;

// ------------------------------------------------------------------------------- C ABI
extern "C" int osud_dit_bind_grad(osud_dit* m, const char* key, float* grad_f32) {
  OSUD_CHECK_ARG(m && key && grad_f32, "bind_grad: null argument");
  OSUD_CHECK_ARG(m->have.count(key) != 0, "bind_grad: unknown parameter '%s'", key);
  m->grad[key] = grad_f32;
  return OSUD_OK;
}

// phase of a parameter: 0 = embedders / conditioning path, 1..L = block p - 1 (with its adaLN pair), L + 1 = final layer
static int phase_of_key(const osud_dit* m, const std::string& k) {
  if (k.compare(0, 7, "blocks.") == 0) return atoi(k.c_str() + 7) + 1;
  if (k.compare(0, 12, "final_layer.") == 0) return m->L + 1;
  return 0;
}

extern "C" int osud_dit_forward_gate(osud_dit* m, int phase, void* hip_event) {
  OSUD_CHECK_ARG(m && phase >= 0 && phase <= m->L + 1, "forward_gate: phase outside 0..depth+1");
  if (m->gate_ev.size() < (size_t)m->L + 2) m->gate_ev.assign((size_t)m->L + 2, nullptr);
  m->gate_ev[(size_t)phase] = (hipEvent_t)hip_event;
  return OSUD_OK;
}

extern "C" int osud_dit_refresh(osud_dit* m, osud_stream stream) {
  OSUD_CHECK_ARG(m, "refresh: null handle");
  OSUD_TRY(osud_dit_refresh_phases(m, 0, m->L + 1, stream));
  if (m->training) OSUD_TRY(build_transposed(m, (hipStream_t)stream));
  return OSUD_OK;
}

extern "C" int osud_dit_refresh_phases(osud_dit* m, int phase_lo, int phase_hi, osud_stream stream) {
  OSUD_CHECK_ARG(m && phase_lo >= 0 && phase_hi <= m->L + 1 && phase_lo <= phase_hi, "refresh: phases outside 0..depth+1");
  // re-pack the parameters of these phases from the caller's fp32 masters (after an optimizer step; the transposed copies of the
  // data-gradient products are rebuilt lazily by the next backward pass)
  std::vector<std::pair<std::string, const float*>> items;
  for (auto& kv : m->master) {
    const int ph = phase_of_key(m, kv.first);
    if (ph >= phase_lo && ph <= phase_hi) items.push_back(kv);
  }
  // set_param only RECORDS its copies / conversions while these are set; each list then goes out as one launch
  SegBatch copies(SEG_COPY, m->prec, (hipStream_t)stream), converts(SEG_CONVERT, m->prec, (hipStream_t)stream);
  struct Guard {
    osud_dit* m;
    ~Guard() {
      m->defer_copy = m->defer_convert = nullptr;
      m->defer_quant = nullptr;
    }
  } guard{m};
  QuantBatch quants(false, (hipStream_t)stream);
  m->defer_copy = &copies;
  m->defer_convert = &converts;
  const bool quant_batched = true;
  if (quant_batched) m->defer_quant = &quants;
  for (auto& kv : items) {
    int64_t shape[2];
    int nd = 0;
    const int64_t D = m->D;
    const std::string& k = kv.first;
    auto ends = [&](const char* s) { const std::string e(s); return k.size() >= e.size() && k.compare(k.size() - e.size(), e.size(), e) == 0; };
    if (k == "xoc_embedder.playfield_size") continue;
    else if (k == "xoc_embedder.mlp.0.weight") { shape[0] = D; shape[1] = 384 + m->E; nd = 2; }
    else if (k == "t_embedder.mlp.0.weight") { shape[0] = D; shape[1] = 256; nd = 2; }
    else if (k == "t_embedder.mlp.2.weight") { shape[0] = D; shape[1] = D; nd = 2; }
    else if (k == "y_embedder.embedding_table.weight") { shape[0] = m->cfg.table_rows; shape[1] = D; nd = 2; }
    else if (k == "final_layer.linear.weight") { shape[0] = m->C2; shape[1] = D; nd = 2; }
    else if (k == "final_layer.linear.bias") { shape[0] = m->C2; nd = 1; }
    else if (k == "final_layer.adaLN_modulation.1.weight") { shape[0] = 2 * D; shape[1] = D; nd = 2; }
    else if (k == "final_layer.adaLN_modulation.1.bias") { shape[0] = 2 * D; nd = 1; }
    else if (ends("attn.in_proj_weight")) { shape[0] = 3 * D; shape[1] = D; nd = 2; }
    else if (ends("attn.in_proj_bias")) { shape[0] = 3 * D; nd = 1; }
    else if (ends("attn.out_proj.weight")) { shape[0] = D; shape[1] = D; nd = 2; }
    else if (ends("mlp.fc1.weight")) { shape[0] = 4 * D; shape[1] = D; nd = 2; }
    else if (ends("mlp.fc1.bias")) { shape[0] = 4 * D; nd = 1; }
    else if (ends("mlp.fc2.weight")) { shape[0] = D; shape[1] = 4 * D; nd = 2; }
    else if (ends("adaLN_modulation.1.weight")) { shape[0] = 6 * D; shape[1] = D; nd = 2; }
    else if (ends("adaLN_modulation.1.bias")) { shape[0] = 6 * D; nd = 1; }
    else { shape[0] = D; nd = 1; }  // every remaining bias is (D,)
    OSUD_TRY(osud_dit_set_param(m, k.c_str(), kv.second, shape, nd, stream));
  }
  OSUD_TRY(copies.flush());
  OSUD_TRY(converts.flush());
  OSUD_TRY(quants.flush());
  m->defer_copy = m->defer_convert = nullptr;
  m->defer_quant = nullptr;
  return OSUD_OK;
}

extern "C" int osud_dit_forward_train(osud_dit* m, const float* x, const int64_t* t, const float* o, const float* c,
                                      const int64_t* y, int N, int T, float* out, osud_stream stream) {
  OSUD_CHECK_ARG(m, "forward_train: null handle");
  OSUD_CHECK_ARG(T % 64 == 0 && (N * T) % 128 == 0,
                 "training needs seq_len %% 64 == 0 and batch*seq_len %% 128 == 0 (got N=%d, T=%d)", N, T);
  OSUD_TRY(dit_ensure_ws(m, N, T, true));  // (rejects learn_sigma = False handles up front)
  return dit_forward_impl(m, x, t, o, c, y, nullptr, N, T, -1.0f, false, out, true, (hipStream_t)stream);
}

extern "C" int osud_dit_backward(osud_dit* m, const float* dout, osud_stream stream) {
  OSUD_CHECK_ARG(m, "backward: null handle");
  return dit_backward_impl(m, dout, 0, m->L + 1, (hipStream_t)stream);
}

extern "C" int osud_dit_backward_phases(osud_dit* m, const float* dout, int phase_lo, int phase_hi, osud_stream stream) {
  OSUD_CHECK_ARG(m, "backward: null handle");
  return dit_backward_impl(m, dout, phase_lo, phase_hi, (hipStream_t)stream);
}

extern "C" int osud_q_sample(const osud_sched* s, const float* x_start, const int64_t* t, const float* noise, int N,
                             int T, float* x_t, osud_stream stream) {
  OSUD_CHECK_ARG(s && x_start && t && noise && x_t && N > 0 && T > 0, "q_sample: null/empty argument");
  OSUD_TRY(sched_upload_train(const_cast<osud_sched*>(s)));
  const int total = N * 2 * T;
  hipLaunchKernelGGL(q_sample_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, sched_train_coefs(s),
                     x_start, t, noise, x_t, N, 2 * T);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

extern "C" int osud_train_loss(const osud_sched* s, int use_l1, const float* model_out, const float* x_start,
                               const float* x_t, const float* noise, const int64_t* t, int N, int T, float* terms,
                               float* dout, osud_stream stream) {
  OSUD_CHECK_ARG(s && model_out && x_start && x_t && noise && t && terms && dout && N > 0 && T > 0,
                 "train_loss: null/empty argument");
  OSUD_TRY(sched_upload_train(const_cast<osud_sched*>(s)));
  hipLaunchKernelGGL(train_loss_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, sched_train_coefs(s), use_l1, model_out,
                     x_start, x_t, noise, t, terms, dout, N, T);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

extern "C" int osud_adamw_ema_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, float* ema,
                                   size_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                                   float ema_decay, size_t skip_begin, size_t skip_end, float grad_scale,
                                   osud_stream stream) {
  OSUD_CHECK_ARG(params && grads && exp_avg && exp_avg_sq && n > 0 && step >= 1, "adamw_ema_step: bad argument");
  // bias corrections in double, as torch evaluates them in Python floats (1 - 0.999f loses 4-5 digits in fp32 at small steps)
  const float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  const float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  // all five arenas are indexed by the same element offset and are equally aligned: `head` elements reach a 16-byte boundary
  size_t head = (size_t)((16 - (reinterpret_cast<uintptr_t>(params) & 15)) & 15) / 4;
  const bool same = ((reinterpret_cast<uintptr_t>(params) ^ reinterpret_cast<uintptr_t>(grads)) & 15) == 0 &&
                    ((reinterpret_cast<uintptr_t>(params) ^ reinterpret_cast<uintptr_t>(exp_avg)) & 15) == 0 &&
                    ((reinterpret_cast<uintptr_t>(params) ^ reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15) == 0 &&
                    (ema == nullptr || ((reinterpret_cast<uintptr_t>(params) ^ reinterpret_cast<uintptr_t>(ema)) & 15) == 0);
  if (!same || head > n) head = n;  // differently aligned buffers: everything on the one-element-per-thread path
  // (the grid follows whichever path carries the elements: a misaligned call must still use the whole chip)
  const size_t work = head == n ? n : (n - head) / 4 + 8;
  // two workgroups per compute unit (512 on this part), long strides (130 M elements, same box: 512 blocks 787-800 us, 256: 821-845, 1024: 831-846, 4096: 820-844;
  // non-temporal loads / stores and two 16-byte sets per lane in flight were slower -- tools/adam_bench.py, profiles/r05_ab_runs.md)
  const size_t cap = 2 * (size_t)gemm_num_cus();
  const int grid = (int)((work + 255) / 256 > cap ? cap : (work + 255) / 256);
  const AdamC c{lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, ema_decay, grad_scale};
  hipLaunchKernelGGL(adamw_ema_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, ema, n, c,
                     skip_begin, skip_end, head);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}
