// Flat-arena parameter kernels: fused SGD (+L2) over all trainable tensors in one launch, the arena algebra of the
// meta-learner (snapshot / axpby -- meta_learners/variables.py:9-45,58-80 become device-side ops on one flat fp32
// buffer), the multiplicative weight-decay pre-step (variables.py:48-55), plus library plumbing (errors, HIP-graph
// capture of an inner step).
// Reference: GradientDescentOptimizer apply under UPDATE_OPS (models/efficientlab.py:301,315-317; args.py:151-152),
// L2 = 5e-4 * sum ||w||^2 / 2 over non-BN trainables (models/regularizers.py:4-10) folded in as grad += 5e-4 * w.
#include <stdarg.h>
#include <string.h>

#include "common.hpp"

namespace mliis {

static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

__device__ __forceinline__ float sign_f(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }   // tf.abs' gradient: sign(0) = 0
// gradient of the weight regularisers of models/regularizers.py:4-19 on the masked (non batch-norm) quads: l2 * w + l1 * sign(w)
__device__ __forceinline__ float4 add_reg(float4 gv, float4 wv, float l2, float l1) {
  gv = f4fma(make_float4(l2, l2, l2, l2), wv, gv);
  if (l1 != 0.f) {
    gv.x += l1 * sign_f(wv.x); gv.y += l1 * sign_f(wv.y); gv.z += l1 * sign_f(wv.z); gv.w += l1 * sign_f(wv.w);
  }
  return gv;
}

// w -= lr * (g + (l2 * w + l1 * sign(w)) * l2mask[quad]) ; lr read from device memory when lr_dev != null (graph replays re-read it)
__global__ __launch_bounds__(256) void sgd_k(float* __restrict__ w, const float* __restrict__ g, const uint8_t* __restrict__ l2mask,
                                             long long nquads, float lr_host, const float* __restrict__ lr_dev, float l2, float l1) {
  const float lr = lr_dev ? *lr_dev : lr_host;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nquads; i += (long long)gridDim.x * blockDim.x) {
    float4 wv = ld4(w + i * 4);
    float4 gv = ld4(g + i * 4);
    if (l2mask && (l2 != 0.f || l1 != 0.f) && l2mask[i]) gv = add_reg(gv, wv, l2, l1);
    wv.x -= lr * gv.x;
    wv.y -= lr * gv.y;
    wv.z -= lr * gv.z;
    wv.w -= lr * gv.w;
    st4(w + i * 4, wv);
  }
}

// Adam with beta1 = 0 (models/efficientlab.py:16; args.py:153-154): v = b2 v + (1-b2) g^2 ; w -= lr_t * g / (sqrt(v) + eps)
// lr_t = lr * sqrt(1 - b2^t) (beta1^t term is 1 - 0 = 1).  step_dev holds the number of steps applied so far (float).  ticket ==
// nullptr: the caller has already advanced it (t = *step_dev).  ticket != nullptr (a zeroed device counter): this launch is step
// *step_dev + 1 and advances the count itself -- every workgroup reads it before it takes a ticket, the last one to finish stores
// t + 1 and re-arms the counter -- so that a captured HIP graph can replay the optimizer step with no host-side bookkeeping.
__global__ __launch_bounds__(256) void adam_b1zero_k(float* __restrict__ w, const float* __restrict__ g, float* __restrict__ v,
                                                     const uint8_t* __restrict__ l2mask, long long nquads, float lr_host,
                                                     const float* __restrict__ lr_dev, float l2, float l1, float beta2, float eps,
                                                     float* __restrict__ step_dev, unsigned* __restrict__ ticket) {
  const float lr = lr_dev ? *lr_dev : lr_host;
  const float tstep = *step_dev + (ticket != nullptr ? 1.0f : 0.0f);
  const float lr_t = lr * sqrtf(1.f - powf(beta2, tstep));
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nquads; i += (long long)gridDim.x * blockDim.x) {
    float4 wv = ld4(w + i * 4), gv = ld4(g + i * 4), vv = ld4(v + i * 4);
    if (l2mask && (l2 != 0.f || l1 != 0.f) && l2mask[i]) gv = add_reg(gv, wv, l2, l1);
    vv.x = beta2 * vv.x + (1.f - beta2) * gv.x * gv.x;
    vv.y = beta2 * vv.y + (1.f - beta2) * gv.y * gv.y;
    vv.z = beta2 * vv.z + (1.f - beta2) * gv.z * gv.z;
    vv.w = beta2 * vv.w + (1.f - beta2) * gv.w * gv.w;
    wv.x -= lr_t * gv.x / (sqrtf(vv.x) + eps);
    wv.y -= lr_t * gv.y / (sqrtf(vv.y) + eps);
    wv.z -= lr_t * gv.z / (sqrtf(vv.z) + eps);
    wv.w -= lr_t * gv.w / (sqrtf(vv.w) + eps);
    st4(v + i * 4, vv);
    st4(w + i * 4, wv);
  }
  if (ticket != nullptr) {   // (uniform)
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned tk = atomicAdd(ticket, 1u);
      if (tk == gridDim.x - 1u) {
        *ticket = 0u;
        *step_dev = tstep;
      }
    }
  }
}

__global__ void copy_words_k(const unsigned* __restrict__ src, unsigned* __restrict__ dst, int n) {
  const int i = threadIdx.x;
  if (i < n) dst[i] = src[i];
}

// y = a * x + b * y   (x nullable when a == 0)
__global__ __launch_bounds__(256) void axpby_k(float a, const float* __restrict__ x, float b, float* __restrict__ y, long long nquads) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nquads; i += (long long)gridDim.x * blockDim.x) {
    float4 yv = ld4(y + i * 4);
    float4 r = f4scale(yv, b);
    if (x) r = f4fma(make_float4(a, a, a, a), ld4(x + i * 4), r);
    st4(y + i * 4, r);
  }
}

// out = a * x + b * y  (three-address; out may alias x or y)
__global__ __launch_bounds__(256) void lincomb_k(float a, const float* __restrict__ x, float b, const float* __restrict__ y,
                                                 float* __restrict__ out, long long nquads) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nquads; i += (long long)gridDim.x * blockDim.x) {
    const float4 xv = ld4(x + i * 4), yv = ld4(y + i * 4);
    st4(out + i * 4, make_float4(a * xv.x + b * yv.x, a * xv.y + b * yv.y, a * xv.z + b * yv.z, a * xv.w + b * yv.w));
  }
}

static inline int flat_blocks(long long nquads) {
  long long b = (nquads + 255) / 256;
  if (b > 2048) b = 2048;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace mliis

using namespace mliis;

extern "C" {

int mliis_version(void) { return 100; }

const char* mliis_last_error(void) { return g_err; }

int mliis_sgd_fused(float* w, const float* g, const uint8_t* l2_quad_mask, long long n, float lr, const float* lr_dev, float l2, float l1,
                    hipStream_t stream) {
  MLIIS_REQUIRE(w && g, MLIIS_ERR_ARG, "sgd_fused: null pointer");
  MLIIS_REQUIRE(n > 0 && (n & 3) == 0, MLIIS_ERR_ARG, "sgd_fused: n must be a positive multiple of 4 (arena is padded)");
  MLIIS_REQUIRE(aligned16(w) && aligned16(g), MLIIS_ERR_ALIGN, "sgd_fused: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(sgd_k, dim3(flat_blocks(n / 4)), dim3(256), 0, stream, w, g, l2_quad_mask, n / 4, lr, lr_dev, l2, l1);
  MLIIS_CHECK_LAUNCH("sgd_fused");
  return MLIIS_OK;
}

int mliis_adam_b1zero_fused(float* w, const float* g, float* v, const uint8_t* l2_quad_mask, long long n, float lr, const float* lr_dev,
                            float l2, float l1, float beta2, float eps, float* step_dev, unsigned* step_ticket, hipStream_t stream) {
  MLIIS_REQUIRE(w && g && v && step_dev, MLIIS_ERR_ARG, "adam_b1zero_fused: null pointer");
  MLIIS_REQUIRE(n > 0 && (n & 3) == 0, MLIIS_ERR_ARG, "adam_b1zero_fused: n must be a positive multiple of 4");
  MLIIS_REQUIRE(aligned16(w) && aligned16(g) && aligned16(v), MLIIS_ERR_ALIGN, "adam_b1zero_fused: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(adam_b1zero_k, dim3(flat_blocks(n / 4)), dim3(256), 0, stream, w, g, v, l2_quad_mask, n / 4, lr, lr_dev, l2, l1,
                     beta2, eps, step_dev, step_ticket);
  MLIIS_CHECK_LAUNCH("adam_b1zero_fused");
  return MLIIS_OK;
}

int mliis_axpby(float a, const float* x, float b, float* y, long long n, hipStream_t stream) {
  MLIIS_REQUIRE(y && (x || a == 0.f), MLIIS_ERR_ARG, "axpby: null pointer");
  MLIIS_REQUIRE(n > 0 && (n & 3) == 0, MLIIS_ERR_ARG, "axpby: n must be a positive multiple of 4");
  MLIIS_REQUIRE(aligned16(x) && aligned16(y), MLIIS_ERR_ALIGN, "axpby: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(axpby_k, dim3(flat_blocks(n / 4)), dim3(256), 0, stream, a, a == 0.f ? nullptr : x, b, y, n / 4);
  MLIIS_CHECK_LAUNCH("axpby");
  return MLIIS_OK;
}

int mliis_copy_words(const void* src, void* dst, int n, hipStream_t stream) {
  MLIIS_REQUIRE(src && dst && n > 0 && n <= 1024, MLIIS_ERR_ARG, "copy_words: 1..1024 words");
  hipLaunchKernelGGL(copy_words_k, dim3(1), dim3(((n + 63) / 64) * 64), 0, stream, reinterpret_cast<const unsigned*>(src), reinterpret_cast<unsigned*>(dst), n);
  MLIIS_CHECK_LAUNCH("copy_words");
  return MLIIS_OK;
}

int mliis_lincomb(float a, const float* x, float b, const float* y, float* out, long long n, hipStream_t stream) {
  MLIIS_REQUIRE(x && y && out, MLIIS_ERR_ARG, "lincomb: null pointer");
  MLIIS_REQUIRE(n > 0 && (n & 3) == 0, MLIIS_ERR_ARG, "lincomb: n must be a positive multiple of 4");
  MLIIS_REQUIRE(aligned16(x) && aligned16(y) && aligned16(out), MLIIS_ERR_ALIGN, "lincomb: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(lincomb_k, dim3(flat_blocks(n / 4)), dim3(256), 0, stream, a, x, b, y, out, n / 4);
  MLIIS_CHECK_LAUNCH("lincomb");
  return MLIIS_OK;
}

// ---- HIP-graph capture of a launch sequence (one inner step = a few hundred launches; replay removes host launch cost)
int mliis_graph_begin_capture(hipStream_t stream) {
  hipError_t e = hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal);
  MLIIS_REQUIRE(e == hipSuccess, MLIIS_ERR_LAUNCH, "graph_begin_capture: %s", hipGetErrorString(e));
  return MLIIS_OK;
}

int mliis_graph_end_capture(hipStream_t stream, void** graph_exec_out) {
  MLIIS_REQUIRE(graph_exec_out, MLIIS_ERR_ARG, "graph_end_capture: null output");
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamEndCapture(stream, &graph);
  MLIIS_REQUIRE(e == hipSuccess && graph, MLIIS_ERR_LAUNCH, "graph_end_capture: %s", hipGetErrorString(e));
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  MLIIS_REQUIRE(e == hipSuccess && exec, MLIIS_ERR_LAUNCH, "graph_instantiate: %s", hipGetErrorString(e));
  *graph_exec_out = (void*)exec;
  return MLIIS_OK;
}

int mliis_graph_launch(void* graph_exec, hipStream_t stream) {
  MLIIS_REQUIRE(graph_exec, MLIIS_ERR_ARG, "graph_launch: null graph");
  hipError_t e = hipGraphLaunch((hipGraphExec_t)graph_exec, stream);
  MLIIS_REQUIRE(e == hipSuccess, MLIIS_ERR_LAUNCH, "graph_launch: %s", hipGetErrorString(e));
  return MLIIS_OK;
}

int mliis_graph_destroy(void* graph_exec) {
  if (graph_exec) (void)hipGraphExecDestroy((hipGraphExec_t)graph_exec);
  return MLIIS_OK;
}
}
