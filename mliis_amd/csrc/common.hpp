// Shared device/host helpers for libmliis_hip (gfx950 / CDNA4 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/mliis_hip.h"

namespace mliis {

int set_error(int code, const char* fmt, ...);

#define MLIIS_REQUIRE(cond, code, ...)                 \
  do {                                                 \
    if (!(cond)) return ::mliis::set_error((code), __VA_ARGS__); \
  } while (0)

#define MLIIS_CHECK_LAUNCH(name)                                                              \
  do {                                                                                        \
    hipError_t e__ = hipGetLastError();                                                       \
    if (e__ != hipSuccess)                                                                    \
      return ::mliis::set_error(MLIIS_ERR_LAUNCH, "%s: launch failed: %s", (name), hipGetErrorString(e__)); \
  } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int ceil_div(long long a, long long b) { return (int)((a + b - 1) / b); }

constexpr int kWave = 64;

// sigmoid through the hardware transcendentals: v_exp_f32 (base 2, ~1 ulp) and v_rcp_f32 (~1 ulp) -- four instructions instead of
// the ~40 of expf() + an IEEE division.  Every activation / BN kernel evaluates one or two of these per element; with the library
// forms the apply passes of the 112x112 layers were as much VALU-bound as memory-bound.  Error: |x| * 2^-24 relative in the
// exponential (rounding of x * log2 e) + 2 ulp, i.e. < 2e-6 relative for |x| <= 20 -- far inside the parity tolerances (2e-5).
__device__ __forceinline__ float sigmoid_f(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float swish_f(float x) { return x * sigmoid_f(x); }
// d/dx [x * sigmoid(x)] = s * (1 + x * (1 - s))
__device__ __forceinline__ float swish_grad_f(float x) {
  float s = sigmoid_f(x);
  return s * (1.0f + x * (1.0f - s));
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// source row/column pair and weight of one output coordinate of tf.image.resize_images(BILINEAR, align_corners=True)
// (scale = (in - 1) / (out - 1)); head.hip's resize kernels and rsd.hip's concat share it
__device__ __forceinline__ void src_coord(int o, float scale, int in_size, int& i0, int& i1, float& l) {
  const float f = (float)o * scale;
  i0 = (int)floorf(f);
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + 1 < in_size ? i0 + 1 : in_size - 1;
  l = f - (float)i0;
}
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4mul(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 f4fma(float4 a, float4 b, float4 c) {
  return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
__device__ __forceinline__ float4 f4scale(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }

// XCD-aware block remap (bijective for any grid size): workgroups are dealt round-robin over the 8 XCDs, each with a private
// L2, so logical neighbours (adjacent rows / row tiles that share halo data) would land on different L2s.  Logical block
// xcd_remap(b, n) gives every XCD one contiguous 1/8 of the logical range, turning halo re-reads into same-XCD L2 hits.
// Placement is only a speed matter: results never depend on it.
__device__ __forceinline__ unsigned xcd_remap(unsigned b, unsigned n) {
  const unsigned q = n >> 3, r = n & 7u, x = b & 7u, l = b >> 3;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + l;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---------------------------------------------------------------------------------------------------------------
// Second stage of the deterministic two-stage reductions: fold `nblk` partial vectors.  Launch with blockDim = (16, 16):
// threadIdx.x = column inside the block, threadIdx.y = one of 16 lanes that stride over the partial blocks; the lanes are
// combined through LDS in double precision in a fixed order, so the result does not depend on scheduling.
// Returns the column sum in every thread with threadIdx.y == 0 (others return garbage-free partials; ignore them).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kFoldX = 16, kFoldY = 16;

__device__ __forceinline__ double fold_partials(const float* __restrict__ part, int nblk, long long blk_stride, long long col_off,
                                                bool col_valid, double* __restrict__ sm /* [kFoldY][kFoldX + 1] */) {
  double s = 0.0;
  if (col_valid)
    for (int b = threadIdx.y; b < nblk; b += kFoldY) s += (double)part[(long long)b * blk_stride + col_off];
  sm[threadIdx.y * (kFoldX + 1) + threadIdx.x] = s;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.y == 0) {
#pragma unroll
    for (int j = 0; j < kFoldY; ++j) r += sm[j * (kFoldX + 1) + threadIdx.x];
  }
  __syncthreads();
  return r;
}

// out[map(i)] (+)= scale * sum_z part[z * total + i], map(i) = (i / seg_len) * seg_stride + seg_off + i % seg_len
// (seg_len = total, seg_off = 0 for a dense output; the segmented form writes a channel window of a larger weight tensor)
__global__ __launch_bounds__(256) void fold_flat_k(const float* __restrict__ part, int nblk, long long total, float scale,
                                                   float* __restrict__ out, int accumulate, long long seg_len, long long seg_stride,
                                                   long long seg_off);

}  // namespace mliis
