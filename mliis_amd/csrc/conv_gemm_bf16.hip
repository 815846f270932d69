// bf16-operand instances of the dense-conv GEMM kernels (v_mfma_f32_16x16x32_bf16, fp32 accumulation); see conv_gemm_kernels.hpp.
#include "conv_gemm_kernels.hpp"

namespace mliis {

void launch_gemm_bf16(const GemmPlan& g, const ConvGemmParams& p, hipStream_t stream) { launch_gemm_t<1>(g, p, stream); }
void launch_gemm_sk_bf16(const GemmPlan& g, const ConvGemmParams& p, float* slab, hipStream_t stream) { launch_gemm_sk_t<1>(g, p, slab, stream); }
void launch_filter_bf16(const FilterPlan& f, const FilterGradParams& p, hipStream_t stream) { launch_filter_t<true>(f, p, stream); }
bool launch_filter_batched_bf16(int tmf, int nt, bool sc, const long long* desc, int nprob, int blocks, hipStream_t stream) {
  return launch_filter_batched_t<true>(tmf, nt, sc, desc, nprob, blocks, stream);
}

}  // namespace mliis
