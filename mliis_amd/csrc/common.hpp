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

// ---- storage element types of the expanded MBConv tensors (z0, z1, a1 and their gradients): float, or bf16 in HBM with fp32
//      arithmetic everywhere (BASELINE configs[3], `--precision bf16-storage`).  Loads widen exactly (bf16 -> fp32 is a 16-bit shift),
//      stores round to nearest even (v_cvt_pk_bf16_f32) -- the rounding the oracle's storage points apply (oracle/efficientlab_ref.py).
//      A quad of 4 consecutive channels is 16 bytes (float) or 8 bytes (bf16s); leading dimensions are in ELEMENTS.
struct bf16s {
  unsigned short u;
};
#define MLIIS_DT_F32 0
#define MLIIS_DT_BF16 1
typedef __bf16 mliis_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16_pair(float lo, float hi) {
  mliis_bf16x2 t;
  t[0] = (__bf16)lo;
  t[1] = (__bf16)hi;
  return __builtin_bit_cast(unsigned, t);
}
__device__ __forceinline__ float bf16_round_f(float v) { return __uint_as_float(pack_bf16_pair(v, 0.f) << 16); }
__device__ __forceinline__ float4 unpack_bf16x4(uint2 u) {
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ uint2 pack_bf16x4(float4 v) { return make_uint2(pack_bf16_pair(v.x, v.y), pack_bf16_pair(v.z, v.w)); }
__device__ __forceinline__ float4 ldq(const float* p) { return ld4(p); }
__device__ __forceinline__ float4 ldq(const bf16s* p) { return unpack_bf16x4(*reinterpret_cast<const uint2*>(p)); }
__device__ __forceinline__ void stq(float* p, float4 v) { st4(p, v); }
__device__ __forceinline__ void stq(bf16s* p, float4 v) { *reinterpret_cast<uint2*>(p) = pack_bf16x4(v); }
// the value a consumer will read back after stq(): identity for float, the bf16 rounding for bf16s
__device__ __forceinline__ float4 stored_value(const float*, float4 v) { return v; }
__device__ __forceinline__ float4 stored_value(const bf16s*, float4 v) {
  return make_float4(bf16_round_f(v.x), bf16_round_f(v.y), bf16_round_f(v.z), bf16_round_f(v.w));
}
template <typename T> struct ElemBytes { static constexpr int value = (int)sizeof(T); };

// A value the optimiser must treat as a lone 32-bit register.  MI355X: while a wave that mixes bf16 / fp8 matrix instructions with memory
// instructions is resident on a CU, a packed fp32 instruction (v_pk_mul / add / fma_f32) of another wave whose op_sel is [0,1] can return a
// wrong low result (profiles/r06_notes.md).  The compiler emits that form when a scalar that is the HIGH half of a register pair (one
// component of a float2 load, of a packed result) is broadcast over the components of a float2 / float4; passing the scalar through lone()
// breaks the pair.  The built library must not contain the form at all: tests/test_build_cpu.py disassembles it (tools/check_packed_forms.py).
__device__ __forceinline__ float lone(float v) {
  asm volatile("" : "+v"(v));
  return v;
}

// source row/column pair and weight of one output coordinate of tf.image.resize_images(BILINEAR, align_corners=True)
// (scale = (in - 1) / (out - 1)); head.hip's resize kernels and rsd.hip's concat share it
__device__ __forceinline__ void src_coord(int o, float scale, int in_size, int& i0, int& i1, float& l) {
  const float f = (float)o * scale;
  i0 = (int)floorf(f);
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + 1 < in_size ? i0 + 1 : in_size - 1;
  l = lone(f - (float)i0);   // (the callers form products of two of these weights and broadcast them over channel pairs: see lone())
}
// the four corner weights of a bilinear sample (top-left, top-right, bottom-left, bottom-right), the products the resize kernels have always
// formed; the complements pass through lone() (the compiler otherwise computes them as one packed pair and takes cross products of it)
__device__ __forceinline__ void bilinear_weights(float ly, float lx, float& wtl, float& wtr, float& wbl, float& wbr) {
  const float my = lone(1.f - ly), mx = lone(1.f - lx);
  wtl = my * mx;
  wtr = my * lx;
  wbl = ly * mx;
  wbr = ly * lx;
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
