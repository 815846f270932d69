// fp8 (OCP e4m3) operand instances of the forward dense-conv GEMM (v_mfma_f32_16x16x32_fp8_fp8, fp32 accumulation) and the reduced-
// precision (bf16 / fp8) instances of the short-K streaming kernel and of the small-map long-K kernel; see conv_gemm_kernels.hpp.  A translation unit of its own so the
// instantiation sets compile in parallel.
#include "conv_gemm_kernels.hpp"

namespace mliis {

void launch_gemm_fp8(const GemmPlan& g, const ConvGemmParams& p, hipStream_t stream) { launch_gemm_t<2>(g, p, stream); }

template <int PREC, bool AIN>
static bool launch_stream_prec(int kc, int nt, dim3 grid, const ConvGemmParams& p, int row_groups, hipStream_t stream) {
  dim3 block(64 * kStreamWaves);
#define S(KC_, NT_) hipLaunchKernelGGL((conv1x1_stream_k<KC_, NT_, PREC, AIN>), grid, block, 0, stream, p, row_groups); break;
#define SN(KC_, NT_) if constexpr (AIN) return false; else { S(KC_, NT_) }   // (the planner never gives more than four column tiles: no AIN instance)
  switch (kc) {
    case 1: switch (nt) { case 1: S(1, 1) case 2: S(1, 2) case 3: S(1, 3) case 4: S(1, 4) case 5: SN(1, 5) case 6: SN(1, 6) case 7: SN(1, 7) case 8: SN(1, 8) default: return false; } break;
    case 2: switch (nt) { case 1: S(2, 1) case 2: S(2, 2) case 3: S(2, 3) case 4: S(2, 4) default: return false; } break;
    case 3: switch (nt) { case 1: S(3, 1) case 2: S(3, 2) default: return false; } break;
    case 4: switch (nt) { case 1: S(4, 1) case 2: S(4, 2) default: return false; } break;
    case 5: switch (nt) { case 1: S(5, 1) default: return false; } break;
    case 6: switch (nt) { case 1: S(6, 1) default: return false; } break;
    case 7: switch (nt) { case 1: S(7, 1) default: return false; } break;
    default: return false;
  }
#undef SN
#undef S
  return true;
}

bool launch_stream_lowp(int precision, int kc, int nt, dim3 grid, const ConvGemmParams& p, int row_groups, hipStream_t stream, bool ain) {
  if (ain)
    return precision == MLIIS_PREC_FP8 ? launch_stream_prec<2, true>(kc, nt, grid, p, row_groups, stream)
                                       : launch_stream_prec<1, true>(kc, nt, grid, p, row_groups, stream);
  return precision == MLIIS_PREC_FP8 ? launch_stream_prec<2, false>(kc, nt, grid, p, row_groups, stream)
                                     : launch_stream_prec<1, false>(kc, nt, grid, p, row_groups, stream);
}

bool launch_ksplit_lowp(int precision, int kc, int nt, dim3 grid, const ConvGemmParams& p, int row_groups, hipStream_t stream) {
  return precision == MLIIS_PREC_FP8 ? launch_ksplit_t<2>(kc, nt, grid, p, row_groups, stream)
                                     : launch_ksplit_t<1>(kc, nt, grid, p, row_groups, stream);
}

}  // namespace mliis
