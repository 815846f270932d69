// Pure-MFMA rate probe: every wave issues `iters` x 8 independent v_mfma_f32_16x16x4_f32 (no memory traffic in the loop).
#include <hip/hip_runtime.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void mfma_probe_k(float* out, int iters) {
  f32x4 acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
// same with v_mfma_f32_32x32x2_f32 (4096 flop per instruction, 64 cycles): 4 independent accumulators of 16 registers
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_probe32_k(float* out, int iters) {
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
  float a = threadIdx.x * 1e-3f, b = 1.0f + blockIdx.x * 1e-6f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) s += acc[j][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
extern "C" int mfma_probe32(float* out, int blocks, int iters, hipStream_t stream) {
  hipLaunchKernelGGL(mfma_probe32_k, dim3(blocks), dim3(256), 0, stream, out, iters);
  return (int)hipGetLastError();
}
extern "C" int mfma_probe(float* out, int blocks, int iters, hipStream_t stream) {
  hipLaunchKernelGGL(mfma_probe_k, dim3(blocks), dim3(256), 0, stream, out, iters);
  return (int)hipGetLastError();
}
