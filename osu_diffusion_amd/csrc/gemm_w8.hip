// GEMM instantiations for fp16 activations x (fp16 + e4m3 residual) weights (w8_t, common.h: per 128 k eight fp16 MFMAs + two
// block-scaled e4m3 MFMAs; kernel: gemm_kernel.h; dispatch: gemm.hip).  The big per-block GEMMs of OSUD_PREC_F16W8.
#include "gemm_kernel.h"

namespace osud {

// inference only: in_proj (bias, split-bf16 output for the attention kernel), fc1 (bias + GELU, w8 activation rows out), out_proj / fc2
// (gated residual, fp32), and the plain fp32 forms the operator tests use
int launch_gemm_w8(int epi, const GemmP& p, hipStream_t st) {
  switch (epi) {
    case EPI_BIAS_F32: return launch_t<w8_t, EPI_BIAS_F32>(p, st);
    case EPI_BIAS_TE: return launch_t<w8_t, EPI_BIAS_TE>(p, st);
    case EPI_BIAS_GELU_TE: return launch_t<w8_t, EPI_BIAS_GELU_TE>(p, st);
    case EPI_BIAS_GELU_ALT: return launch_t<w8_t, EPI_BIAS_GELU_ALT>(p, st);
    case EPI_GATE_RES: return launch_t<w8_t, EPI_GATE_RES>(p, st);
    case EPI_NONE_F32: return launch_t<w8_t, EPI_NONE_F32>(p, st);
  }
  set_error("gemm: epilogue %d is not built for fp16 x (fp16 + e4m3) operands (inference tier)", epi);
  return OSUD_ERR_UNSUPPORTED;
}

}  // namespace osud
