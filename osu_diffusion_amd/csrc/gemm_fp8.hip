// GEMM instantiations for OCP e4m3 (fp8_t) operands (kernel: gemm_kernel.h; dispatch: gemm.hip).
#include "gemm_kernel.h"

namespace osud {

// fp32, bf16 (EPI_BIAS_TE) or fp8 (EPI_BIAS_GELU_TE) outputs
int launch_gemm_fp8(int epi, const GemmP& p, hipStream_t st) {
  switch (epi) {
    case EPI_NONE_F32: return launch_t<fp8_t, EPI_NONE_F32>(p, st);
    case EPI_BIAS_F32: return launch_t<fp8_t, EPI_BIAS_F32>(p, st);
    case EPI_BIAS_TE: return launch_t<fp8_t, EPI_BIAS_TE>(p, st);
    case EPI_BIAS_GELU_TE: return launch_t<fp8_t, EPI_BIAS_GELU_TE>(p, st);
    case EPI_GATE_RES: return launch_t<fp8_t, EPI_GATE_RES>(p, st);
    case EPI_BIAS_GELU_BF: return launch_t<fp8_t, EPI_BIAS_GELU_BF>(p, st);  // training: bf16 gelu + gelu' outputs
    case EPI_NONE_TE: return launch_t<fp8_t, EPI_NONE_TE>(p, st);            // data gradients
    case EPI_GELUGRAD_TE: return launch_t<fp8_t, EPI_GELUGRAD_TE>(p, st);
  }
  set_error("gemm: epilogue %d is not built for fp8 operands", epi);
  return OSUD_ERR_UNSUPPORTED;
}

}  // namespace osud
