// Launchers of the non-GEMM kernels (definitions in kernels_fwd.hip, attention.hip,
// sampler.hip, kernels_bwd.hip).  Every launcher enqueues on `st` and returns an OSUD_* code.
#pragma once
#include "common.h"

namespace osud {

int launch_embed(int prec, const float* x, const float* o, const float* c, const float* freqs64, float pf0, float pf1,
                 void* out, int N, int T, int Tp, int Mp, int E, int Kp, int x_dup_half, hipStream_t st,
                 bool split = false /* bf16 only: rows are [hi | lo | hi], 3 * Kp columns (see embed_kernel) */,
                 int mode = 0 /* 1: coordinate features only (compact 256-column row), 2: whole row, coordinate features zeroed */);
int launch_temb(int prec, const int64_t* t, const float* freqs128, void* out, int N, int Np, hipStream_t st);
int launch_cond(int prec, const float* tvec, const float* table, const int64_t* y, int table_rows, float* b_out,
                void* sb_out, int N, int Np, int D, hipStream_t st, const int64_t* t_index = nullptr);
// br != nullptr: the row is first updated to h + ada[n][off_gate..] * br (written to h_out if given, may be h itself)
int launch_ln_mod(int prec, const float* h, const float* ada, int ld_ada, int off_shift, int off_scale, void* out,
                  float* stats, int M, int Tp, int N, int D, hipStream_t st, const void* br = nullptr, int off_gate = 0,
                  float* h_out = nullptr, float fp8_scale = 0.f /* > 0: out is e4m3, values multiplied by this */);
// fp8 tier: per-output-channel e4m3 quantisation of a weight (rows x cols fp32) and its de-quantisation factors
int launch_quantize_rows(const float* w, int rows, int cols, void* q, float* dequant, float act_scale, hipStream_t st);
int launch_quantize_rows_bf16(const void* w_bf16, int rows, int cols, void* q, float* dequant, hipStream_t st);
// fp8 training, delayed per-tensor scaling.  slot = {scale, 1/scale, amax of this step, -}:
//   quantize: dst[i] = e4m3(src[i] * slot[0]) (dst may be null: record only), slot[2] = max(slot[2], max|src|)
//   update  : for every slot with a recorded amax: scale = 448 / (2 * amax) (one binade of headroom), amax cleared
int launch_f8_quantize(const void* src_bf16, void* dst_fp8, size_t n, float* slot, hipStream_t st);
int launch_f8_update(float* slots, int n_slots, hipStream_t st, float* amax_parts = nullptr /* [n_slots][f8_amax_parts()] */);
int f8_amax_parts();
int launch_ln_mod_twin(const float* h, const float* ada, int ld_ada, int off_shift, int off_scale, void* out, float* stats, int M,
                       int Tp, int N, int D, hipStream_t st, const void* br, int off_gate, float* h_out, void* out8, const float* slot,
                       float* amax_part);
int launch_final(const float* h, const float* ada, int ld_ada, int off_shift, int off_scale, const float* w,
                 const float* bias, float* out, float* u_save, float* stats, int N, int T, int Tp, int D, int C,
                 hipStream_t st, int prec = OSUD_PREC_F32, const void* br = nullptr, int off_gate = 0,
                 float* h_out = nullptr);
int launch_cfg_combine(float* out, int N, int C, int C2, int T, float s, hipStream_t st);
int launch_convert(int prec, const float* src, void* dst, size_t n, hipStream_t st);
int launch_pack_rows(int prec, const float* src, int ld_src, int cols_src, void* dst, int ld_dst, int cols_dst, int rows,
                     hipStream_t st);
// split-bf16 tier: dst [rows][2 * cols_dst] = [w_hi | w_lo] (zero padded to cols_dst per plane)
int launch_pack_rows_x3(const float* src, int ld_src, int cols_src, void* dst, int cols_dst, int rows, hipStream_t st);
// fp16 + e4m3 operand form (common.h: h8_t; cols_dst % 32 == 0, zero padded): weight = the weight flavour (hi8 | lo8 planes)
int launch_pack_rows_h8(const float* src, int ld_src, int cols_src, void* dst, int cols_dst, int rows, bool weight, hipStream_t st);
// fp16 x (fp16 + e4m3) operand form (common.h: w8_t; cols_dst % 128 == 0): weight = the flavour whose e4m3 plane is the fp16 residual
int launch_pack_rows_w8(const float* src, int ld_src, int cols_src, void* dst, int cols_dst, int rows, bool weight, hipStream_t st);
// first linear of the bf16 tier: dst [rows][3 * cols_dst] = [w_hi | w_hi | w_lo]
int launch_pack_rows_split(const float* src, int ld_src, int cols_src, void* dst, int cols_dst, int rows, hipStream_t st,
                           int prec = OSUD_PREC_BF16 /* or OSUD_PREC_F16 */);
// qkv: packed in_proj output [Mp][ld_qkv], Q | K | V in columns [0,D) [D,2D) [2D,3D)
// kb_class (optional, from launch_mask_tiles): class of every 64-query x 64-key tile of the mask (0 fully masked, 1 mixed,
// 2 fully open): masked tiles are skipped, open tiles read no mask bytes
int launch_attention(int prec, const void* qkv, int ld_qkv, const uint8_t* mask, void* out, float* lse, int N, int T, int Tp,
                     int Mp, int heads, int head_dim, hipStream_t st, const uint8_t* kb_class = nullptr,
                     float fp8_scale = 0.f /* > 0 (bf16 tier only): out is e4m3 [Mp][D], values multiplied by this */);
int launch_mask_tiles(const uint8_t* mask, int T, int Tp, uint8_t* kb_class, hipStream_t st);
// delta_ws: [N][heads][T] fp32 scratch, needed when the sequence of one head does not fit the LDS (streamed variant)
// dbias (optional, bf16 tier): [3 * hidden] fp32, += column sums of dqkv over all tokens (the in_proj bias gradient)
int launch_attention_bwd(int prec, const void* qkv, const void* dO, const void* O, const float* lse, void* dqkv, int N,
                         int T, int heads, int head_dim, hipStream_t st, float* delta_ws = nullptr, float* dbias = nullptr,
                         float* bias_scratch = nullptr /* N x 3 hidden floats: the streamed kernel's per-sample column sums */,
                         size_t bias_scratch_elems = 0,
                         int* bias_rows_pending = nullptr /* optional: where the per-sample sums of the streamed kernel were written to
                                                             bias_scratch, the final column sum is LEFT TO THE CALLER and *bias_rows_pending =
                                                             the number of rows (else it is 0 and dbias is complete) */);  // dbias != nullptr (bf16 tier): column sums of dqkv are ADDED to / written into it

// kernels_bwd.hip
// No kernel of the backward pass adds floats atomically: sums that several workgroups contribute to are written as per-workgroup
// partial rows and added by a fixed-order pass, so a gradient is the same bits on every run and under every stream schedule.
// colsum (optional) = column sums of `in` (a bias gradient), written; needs colpart: (R / 64) x C floats of scratch
int launch_transpose(int prec, const void* in, int ld_in, void* out, int ld_out, int R, int C, float* colsum,
                     hipStream_t st, float* colpart = nullptr, size_t colpart_elems = 0);
int launch_transpose_f32(int prec, const float* in, int ld_in, void* out, int ld_out, int R, int C, float* colsum,
                         hipStream_t st, float* colpart = nullptr, size_t colpart_elems = 0);
// Descriptor list of launch_row_reduce (kernels_bwd.hip: row_reduce_kernel): item i = the per-block partial rows one gate_bwd /
// ln_mod_bwd / final_bwd launch left behind; rows q < nq_sample become dada[n][off[q]..] (sum over the sample's bps blocks), row
// nq_sample -- if bias is set -- becomes bias[..] (sum over all blocks)
struct RowRedList {
  static constexpr int kMax = 40;
  const float* part[kMax];
  float* dada[kMax];
  float* bias[kMax];
  int off[kMax][3];
  int stride[kMax], blocks[kMax], bps[kMax], nq_sample[kMax];
  int count, D, ld_ada;
};
int launch_row_reduce(const RowRedList& L, hipStream_t st);
// A list of fixed-order column sums out[c] = sum_{r < R} src[r][c] (C % 64 == 0 columns, row stride ld) in ONE launch: the bias
// gradients whose partial rows come out of GEMM / attention epilogues (fc1's, in_proj's), deferred to the end of a backward call
struct ColsumList {
  static constexpr int kMax = 64;
  const float* src[kMax];
  float* out[kMax];
  int R[kMax], ld[kMax], blk_begin[kMax + 1];  // blk_begin: first 64-column block of item i in the launch's grid
  int count;
};
int launch_colsum_many(const ColsumList& L, hipStream_t st);
// part: (M / 64) x 2 D floats, rows {dgate share, share of db = the branch Linear's bias gradient}
int launch_gate_bwd(int prec, const float* dh, const void* br, const float* gate, int ld_ada, void* dbr, float* part,
                    int M, int Tp, int D, hipStream_t st);
// du is TE.  br_next != nullptr: also run the gate_bwd of the branch added in front of this LayerNorm on the fresh dh_out
// rows (dbr = gate * dh_out).  part: (M / 64) x Q D floats, rows {dshift, dscale} and with the gate step (Q = 4) {dgate of the next
// branch, its bias gradient}: the block's shares, to be summed by launch_row_reduce
int launch_ln_mod_bwd(int prec, const float* h, const float* stats, const void* du, const float* ada, int ld_ada,
                      int off_shift, int off_scale, const float* dh_skip, float* dh_out, float* part, int M, int Tp, int D,
                      hipStream_t st, const void* br_next = nullptr, int off_gate_next = 0, void* dbr = nullptr,
                      void* dbr8 = nullptr, const float* slot8 = nullptr, float* amax_part = nullptr);
// part: (N Tp / 64) x (6 D + 64) floats; dw / dbias are written here, the dshift | dscale rows (at + 4 D + 64 of a block's row) are
// left for launch_row_reduce
int launch_final_bwd(const float* h, const float* stats, const float* dout, const float* w, const float* ada, int ld_ada,
                     int off_shift, int off_scale, float* dh_out, float* dw, float* dbias, int N, int T, int Tp,
                     int D, int C, hipStream_t st, float* part);
int launch_cond_bwd(int prec, const float* dsb, const float* b, const int64_t* y, int table_rows, float* db_out,
                    void* db_te, float* dtable, int N, int Np, int D, hipStream_t st);
int launch_silu_bwd(int prec, const float* dth, const void* z, void* dz, size_t n, hipStream_t st);
int launch_mask_rows(int prec, float* a, void* a_te, int N, int Np, int C, hipStream_t st, int ld = 0);  // ld > C: a column slice
int launch_unpad_rows(const float* src, int ld_src, float* dst, int cols, int rows, hipStream_t st);
// out[c] = sum_r a[r][c] in a fixed order (columns >= split go to out2[c - split], n2 of them); ld: row stride when only the first C
// columns of wider rows are summed
int launch_colsum_f32(const float* a, int R_valid, int C, float* out, hipStream_t st, int split = 0, float* out2 = nullptr, int n2 = 0,
                      int ld = 0);

// wgrad.hip (bf16 tier): transpose-free weight-gradient product and bias-gradient column sums
int launch_wgrad_tr(const void* P, int ldp, const void* Q, int ldq, int Ny, int Nx, int M, float* out, float* ws,
                    size_t ws_elems, hipStream_t st);
// out[c] = sum_m a[m][c] (written; fixed order); part: ceil(M / 256) x N floats of scratch for the row blocks' shares
int launch_colsum_bf16(const void* a, int ld, int M, int N, float* out, hipStream_t st, float* part, size_t part_elems);
// fp8 training: column sums AND e4m3 twin (x slot[0]; q8 may be null) AND amax (slot[2]) of a dense bf16 [M][N] matrix
int launch_colsum_quant_bf16(const void* a, int M, int N, float* out, void* q8, float* slot, hipStream_t st, float* part, size_t part_elems);
// the same product on e4m3 twins (fp8 training): out = inv_p[0] * inv_q[0] * P8^T . Q8 (device scalars: the operands' 1 / scale)
int launch_wgrad8_tr(const void* P8, int ldp, const void* Q8, int ldq, int Ny, int Nx, int M, float* out, float* ws, size_t ws_elems,
                     const float* inv_p, const float* inv_q, hipStream_t st);
// batch.hip: one launch for a list of small buffers (passed by value in the kernel arguments)
enum SegOp { SEG_ZERO = 0, SEG_COPY = 1, SEG_CONVERT = 2 };
struct SegList {
  static constexpr int kMax = 64;
  const void* src[kMax];
  void* dst[kMax];
  size_t n[kMax];  // SEG_ZERO / SEG_COPY: 16-byte chunks; SEG_CONVERT (fp32 -> TE): groups of 4 elements
  int count;
};
struct TransposeList {  // dst[c][r] = src[r][c], TE, dense (ld = C / R), R and C multiples of 64
  static constexpr int kMax = 64;
  const void* src[kMax];
  void* dst[kMax];
  int R[kMax], C[kMax], tile_begin[kMax];
  int count;
};
int launch_segments(int op, int prec, const SegList& L, hipStream_t st);
int launch_transpose_many(int prec, TransposeList& L, hipStream_t st);
// host-side accumulator: add() launches by itself whenever the list is full, flush() sends the rest
struct SegBatch {
  int op, prec;
  hipStream_t st;
  SegList L{};
  SegBatch(int op_, int prec_, hipStream_t st_) : op(op_), prec(prec_), st(st_) {}
  int flush() {
    const int rc = launch_segments(op, prec, L, st);
    L.count = 0;
    return rc;
  }
  int add(const void* src, void* dst, size_t n) {
    if (L.count == SegList::kMax) OSUD_TRY(flush());
    L.src[L.count] = src; L.dst[L.count] = dst; L.n[L.count] = n;
    ++L.count;
    return OSUD_OK;
  }
};

// fp8 tier: per-row e4m3 quantisation (launch_quantize_rows / _bf16) of a whole LIST of weights in one launch; block b finds its
// matrix by its row range.  224 weights per DiT-XL step were 224 launches of ~9 us (each too short to fill the memory system).
struct QuantList {
  static constexpr int kMax = 48;
  const void* src[kMax];  // fp32 (SRC_BF16 = false) or bf16 rows, dense
  void* q[kMax];
  float* dq[kMax];
  int cols[kMax], row_begin[kMax + 1];
  int count;
};
int launch_quantize_rows_many(bool src_bf16, const QuantList& L, hipStream_t st);
struct QuantBatch {
  bool src_bf16;
  hipStream_t st;
  QuantList L{};
  QuantBatch(bool src_bf16_, hipStream_t st_) : src_bf16(src_bf16_), st(st_) {}
  int flush() {
    const int rc = launch_quantize_rows_many(src_bf16, L, st);
    L.count = 0;
    return rc;
  }
  int add(const void* src, int rows, int cols, void* q, float* dq) {
    if (L.count == QuantList::kMax) OSUD_TRY(flush());
    if (L.count == 0) L.row_begin[0] = 0;
    L.src[L.count] = src; L.q[L.count] = q; L.dq[L.count] = dq; L.cols[L.count] = cols;
    L.row_begin[L.count + 1] = L.row_begin[L.count] + rows;
    ++L.count;
    return OSUD_OK;
  }
};

// sampler.hip
struct StepCoefs;  // device table, 8 floats per step
int launch_sampler_step(const float* coefs, int mode, float eta, const float* model_out, const float* x,
                        const int64_t* t_index, const int* step_state, const float* noise, size_t noise_step_stride,
                        uint64_t seed, int N, int T, float cfg_scale, int clip, const osud_inpaint* inpaint, float* x_out,
                        float* pred_xstart, hipStream_t st);
// step_state: 8 ints {next index, current index, k-th step of the loop (current), k (next), seed lo, seed hi, decrement per step, -}
int launch_step_init(int* step_state, int first, uint64_t seed, hipStream_t st, int dec = 1);
int launch_step_begin(int* step_state, const int64_t* tmap_dev, int64_t* t_model, int64_t* t_index, int N,
                      hipStream_t st);

}  // namespace osud
