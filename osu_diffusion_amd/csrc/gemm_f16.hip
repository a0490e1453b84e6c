// GEMM instantiations for IEEE half operands (f16_t: the fp16 tier, OSUD_PREC_F16 -- the bf16 tier's loop on v_mfma_f32_32x32x16_f16;
// kernel: gemm_kernel.h; dispatch: gemm.hip).
#include "gemm_kernel.h"

namespace osud {

// inference only: the epilogues of the forward pass
int launch_gemm_f16(int epi, const GemmP& p, hipStream_t st) {
  switch (epi) {
    case EPI_BIAS_F32: return launch_t<f16_t, EPI_BIAS_F32>(p, st);
    case EPI_BIAS_TE: return launch_t<f16_t, EPI_BIAS_TE>(p, st);
    case EPI_BIAS_SILU_TE: return launch_t<f16_t, EPI_BIAS_SILU_TE>(p, st);
    case EPI_BIAS_GELU_TE: return launch_t<f16_t, EPI_BIAS_GELU_TE>(p, st);
    case EPI_GATE_RES: return launch_t<f16_t, EPI_GATE_RES>(p, st);
    case EPI_NONE_F32: return launch_t<f16_t, EPI_NONE_F32>(p, st);
  }
  set_error("gemm: epilogue %d is not built for fp16 operands (inference tier)", epi);
  return OSUD_ERR_UNSUPPORTED;
}

}  // namespace osud
