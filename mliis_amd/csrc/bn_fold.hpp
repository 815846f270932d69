// Stage 2 of the deterministic batch-norm statistics: fold the per-block partial sums a producer left behind ([nblk][2][C]) for
// the 32 channels a workgroup owns.  Shared by the BN apply / backward kernels (bn.hip) and the row-marching depthwise kernels
// (dwmarch.hip), which apply the preceding batch norm while they stage their input.
#pragma once
#include "common.hpp"

namespace mliis {

struct BnFold {
  const float* part;  // [nblk][2][C]
  int nblk;
  double inv_n;
  float eps, one_minus_momentum, ema_var_factor;
  float* mean;        // [C] out
  float* rstd;        // [C] out
  float* moving_mean; // nullable
  float* moving_var;
};

// 256 threads = 8 channel quads x 32 fold lanes; each lane strides over the partial blocks with float4 loads, the 32 lanes
// are combined through LDS in double precision in a fixed order.  Result (threads 0..31 <-> channels c0..c0+31) in s / ss.
template <int kFoldBatch = 8>
__device__ __forceinline__ void fold32(const float* __restrict__ part, int nblk, int C, int c0, double* smd /*[2][32][32]*/, double& s,
                                       double& ss) {
  const int t = threadIdx.x, q = t & 7, bl = t >> 3;
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0;
  const int c = c0 + q * 4;
  if (c < C) {
    // kFoldBatch partial blocks per round trip: the big maps hand over 400-1600 partials (25-50 per lane), and one block per
    // iteration -- load, wait, add -- cost 15 us of pure latency in every apply kernel of a 112x112 layer
    for (int k = bl; k < nblk; k += 32 * kFoldBatch) {
      float4 u[kFoldBatch], v[kFoldBatch];
#pragma unroll
      for (int j = 0; j < kFoldBatch; ++j) {
        const int kk = k + 32 * j;
        const long long kr = kk < nblk ? kk : bl;   // surplus slots re-read this lane's first block (valid address) and are masked below
        u[j] = ld4(part + (kr * 2 + 0) * C + c);
        v[j] = ld4(part + (kr * 2 + 1) * C + c);
      }
#pragma unroll
      for (int j = 0; j < kFoldBatch; ++j)
        if (k + 32 * j < nblk) {
          a0 += u[j].x; a1 += u[j].y; a2 += u[j].z; a3 += u[j].w;
          b0 += v[j].x; b1 += v[j].y; b2 += v[j].z; b3 += v[j].w;
        }
    }
  }
  double* p0 = smd + (0 * 32 + bl) * 32 + q * 4;
  double* p1 = smd + (1 * 32 + bl) * 32 + q * 4;
  p0[0] = a0; p0[1] = a1; p0[2] = a2; p0[3] = a3;
  p1[0] = b0; p1[1] = b1; p1[2] = b2; p1[3] = b3;
  __syncthreads();
  s = ss = 0.0;
  if (t < 32) {
#pragma unroll 8
    for (int k = 0; k < 32; ++k) {
      s += smd[(0 * 32 + k) * 32 + t];
      ss += smd[(1 * 32 + k) * 32 + t];
    }
  }
}

}  // namespace mliis
