// GEMM instantiations for bf16_t operands (kernel: gemm_kernel.h; dispatch: gemm.hip).
#include "gemm_kernel.h"

namespace osud {
namespace {
template <typename TE> int launch_e(int epi, const GemmP& p, hipStream_t st) {
  switch (epi) {
    case EPI_BIAS_F32: return launch_t<TE, EPI_BIAS_F32>(p, st);
    case EPI_BIAS_TE: return launch_t<TE, EPI_BIAS_TE>(p, st);
    case EPI_BIAS_SILU_TE: return launch_t<TE, EPI_BIAS_SILU_TE>(p, st);
    case EPI_ROWBIAS_TE: return launch_t<TE, EPI_ROWBIAS_TE>(p, st);
    case EPI_BIAS_GELU_TE: return launch_t<TE, EPI_BIAS_GELU_TE>(p, st);
    case EPI_GATE_RES: return launch_t<TE, EPI_GATE_RES>(p, st);
    case EPI_NONE_F32: return launch_t<TE, EPI_NONE_F32>(p, st);
    case EPI_NONE_TE: return launch_t<TE, EPI_NONE_TE>(p, st);
    case EPI_ACCUM_F32: return launch_t<TE, EPI_ACCUM_F32>(p, st);
    case EPI_GELUGRAD_TE: return launch_t<TE, EPI_GELUGRAD_TE>(p, st);
  }
  set_error("gemm: unknown epilogue %d", epi);
  return OSUD_ERR_ARG;
}

}  // namespace

int launch_gemm_bf16(int epi, const GemmP& p, hipStream_t st) { return launch_e<bf16_t>(epi, p, st); }

}  // namespace osud
