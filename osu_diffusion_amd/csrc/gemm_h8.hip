// GEMM instantiations for fp16 + e4m3 operands (h8_t, common.h: fp16 hi product + one block-scaled e4m3 MFMA for the two cross terms;
// kernel: gemm_kernel.h; dispatch: gemm.hip).  The four big per-block GEMMs of the tolerance tier's fast form (OSUD_PREC_F16F8).
#include "gemm_kernel.h"

namespace osud {

// inference only: in_proj (bias, split-bf16 output for the attention kernel), fc1 (bias + GELU, h8 output), out_proj / fc2 (gated
// residual, fp32), and the plain fp32 forms the operator tests use
int launch_gemm_h8(int epi, const GemmP& p, hipStream_t st) {
  switch (epi) {
    case EPI_BIAS_F32: return launch_t<h8_t, EPI_BIAS_F32>(p, st);
    case EPI_BIAS_TE: return launch_t<h8_t, EPI_BIAS_TE>(p, st);
    case EPI_BIAS_GELU_TE: return launch_t<h8_t, EPI_BIAS_GELU_TE>(p, st);
    case EPI_BIAS_GELU_ALT: return launch_t<h8_t, EPI_BIAS_GELU_ALT>(p, st);
    case EPI_GATE_RES: return launch_t<h8_t, EPI_GATE_RES>(p, st);
    case EPI_NONE_F32: return launch_t<h8_t, EPI_NONE_F32>(p, st);
  }
  set_error("gemm: epilogue %d is not built for fp16 + e4m3 operands (inference tier)", epi);
  return OSUD_ERR_UNSUPPORTED;
}

}  // namespace osud
