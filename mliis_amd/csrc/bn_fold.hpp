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
// The pieces (issue the loads of one round / add them / finish) are separate so that a kernel can put its own memory traffic between
// the loads of the last round and their use (dwmarch.hip); fold32 strings them together (bn.hip, mbconv_small.hip).
struct FoldAcc {
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0;
};

// loads of blocks k, k + 32, ... (kBatch of them) of this lane's channel quad; surplus slots re-read block `safe` (a valid one) and are
// masked in fold_add
template <int kBatch>
__device__ __forceinline__ void fold_issue(const float* __restrict__ part, int nblk, int C, int c, int k, int safe, float4 (&u)[kBatch],
                                           float4 (&v)[kBatch]) {
#pragma unroll
  for (int j = 0; j < kBatch; ++j) {
    const int kk = k + 32 * j;
    const long long kr = kk < nblk ? kk : safe;
    u[j] = ld4(part + (kr * 2 + 0) * C + c);
    v[j] = ld4(part + (kr * 2 + 1) * C + c);
  }
}
template <int kBatch>
__device__ __forceinline__ void fold_add(FoldAcc& f, int nblk, int k, const float4 (&u)[kBatch], const float4 (&v)[kBatch]) {
#pragma unroll
  for (int j = 0; j < kBatch; ++j)
    if (k + 32 * j < nblk) {
      f.a0 += u[j].x; f.a1 += u[j].y; f.a2 += u[j].z; f.a3 += u[j].w;
      f.b0 += v[j].x; f.b1 += v[j].y; f.b2 += v[j].z; f.b3 += v[j].w;
    }
}
// the 32 lanes of every channel through LDS, summed in lane order: 64 threads, one statistic of one channel each (the second
// statistic comes back to threads 0..31 by a lane shift inside wave 0).  One barrier inside; the caller owns the one after its reads.
__device__ __forceinline__ void fold_finish(const FoldAcc& f, double* smd /*[2][32][32]*/, double& s, double& ss) {
  const int t = threadIdx.x, q = t & 7, bl = t >> 3;
  double* p0 = smd + (0 * 32 + bl) * 32 + q * 4;
  double* p1 = smd + (1 * 32 + bl) * 32 + q * 4;
  p0[0] = f.a0; p0[1] = f.a1; p0[2] = f.a2; p0[3] = f.a3;
  p1[0] = f.b0; p1[1] = f.b1; p1[2] = f.b2; p1[3] = f.b3;
  __syncthreads();
  double x = 0.0;
  if (t < 64) {
    const double* col = smd + (t >> 5) * 32 * 32 + (t & 31);
#pragma unroll 8
    for (int k = 0; k < 32; ++k) x += col[k * 32];
  }
  s = x;
  ss = __shfl_down(x, 32, 64);   // (threads 32..63 of wave 0 hold the second statistic)
}

template <int kFoldBatch = 8>
__device__ __forceinline__ void fold32(const float* __restrict__ part, int nblk, int C, int c0, double* smd /*[2][32][32]*/, double& s,
                                       double& ss) {
  const int t = threadIdx.x, q = t & 7, bl = t >> 3;
  FoldAcc f;
  const int c = c0 + q * 4;
  if (c < C) {
    // kFoldBatch partial blocks per round trip: the big maps hand over 400-1600 partials (25-50 per lane), and one block per
    // iteration -- load, wait, add -- cost 15 us of pure latency in every apply kernel of a 112x112 layer
    for (int k = bl; k < nblk; k += 32 * kFoldBatch) {
      float4 u[kFoldBatch], v[kFoldBatch];
      fold_issue<kFoldBatch>(part, nblk, C, c, k, bl, u, v);
      fold_add<kFoldBatch>(f, nblk, k, u, v);
    }
  }
  fold_finish(f, smd, s, ss);
}

}  // namespace mliis
