// Depthwise k x k convolution (k in {3,5}, stride in {1,2}, TF-SAME padding), NHWC fp32: forward, backward-data,
// backward-filter.
// WHO STILL CALLS THIS FILE (since round 3 the MBConv blocks of the training step run the row-marching kernels of dwmarch.hip and the
// fused small-map kernels of mbconv_small.hip): the depthwise convs of the `--skip_decoding` decoder's separable convs, blocks in
// inference / evaluation at shapes neither fused family takes (their *_supported queries), and Learner(small_fused=False, dw_march=False),
// the op-by-op path that tests/test_step_gpu.py::test_one_step_grads_params_bn[small_fused=False] keeps covered.  Reference call sites: models/efficientnet/efficientnet_model.py:190-196,271 and
// models/efficientnet/utils.py:219-222 (keras DepthwiseConv2D, depth_multiplier 1, no bias).
//
// HBM-bound (arithmetic intensity 0.9-6.2 flop/B).  Lanes run along C in float4, so one wave
// instruction reads whole contiguous channel spans; every thread owns a strip of TW outputs along W and keeps the
// (TW-1)*S+K input columns of the current filter row in registers (sliding window), so each input element is fetched
// K (not K*K) times per strip and those re-reads are L1/L2 hits.  Filter taps are wave-uniform-per-quad loads that stay
// in L1.  Backward-filter is a deterministic two-stage reduction (block partials -> double-precision fold).
#include "common.hpp"

namespace mliis {

template <int K, int S, int TW>
__global__ __launch_bounds__(256) void dwconv_fwd_k(const float* __restrict__ x, const float* __restrict__ w,
                                                    float* __restrict__ y, int N, int Hi, int Wi, int Ho, int Wo, int C, int pt,
                                                    int pl) {
  constexpr int IW = (TW - 1) * S + K;
  const int Q = C >> 2;
  const int strips = (Wo + TW - 1) / TW;
  const unsigned total = (unsigned)N * Ho * strips * Q;   // host guarantees < 2^31: 32-bit index math (64-bit div is ~4x dearer)
  const unsigned i = xcd_remap(blockIdx.x, gridDim.x) * 256u + threadIdx.x;
  if (i >= total) return;
  const int cq = (int)(i % (unsigned)Q);
  unsigned r = i / (unsigned)Q;
  const int sx = (int)(r % (unsigned)strips);
  r /= (unsigned)strips;
  const int ho = (int)(r % (unsigned)Ho);
  const int n = (int)(r / (unsigned)Ho);
  const int c = cq << 2;
  const int wo0 = sx * TW;
  const int hi0 = ho * S - pt, wi0 = wo0 * S - pl;
  float4 acc[TW];
#pragma unroll
  for (int t = 0; t < TW; ++t) acc[t] = f4zero();
  if constexpr (K == 5) {
    // 5x5 layers live on small maps (one wave per SIMD): issue the whole 5 x IW window before any FMA so the loads form ONE
    // memory round trip instead of five (out-of-range taps read a safe address and are zeroed by a select)
    float4 in[K][IW];
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
      const int hi = hi0 + ky;
      const bool rok = (hi >= 0) && (hi < Hi);
      const float* xrow = x + ((long long)(n * Hi + (rok ? hi : 0)) * Wi) * C + c;
#pragma unroll
      for (int j = 0; j < IW; ++j) {
        const int wi = wi0 + j;
        const bool ok = rok && (wi >= 0) && (wi < Wi);
        const float4 v = ld4(ok ? xrow + (long long)wi * C : x + c);
        in[ky][j] = ok ? v : f4zero();
      }
    }
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const float4 wv = ld4(w + (ky * K + kx) * C + c);
#pragma unroll
        for (int t = 0; t < TW; ++t) acc[t] = f4fma(in[ky][t * S + kx], wv, acc[t]);
      }
  } else {
#pragma unroll
  for (int ky = 0; ky < K; ++ky) {
    const int hi = hi0 + ky;
    if (hi < 0 || hi >= Hi) continue;
    const float* xrow = x + ((long long)(n * Hi + hi) * Wi) * C + c;
    float4 in[IW];
#pragma unroll
    for (int j = 0; j < IW; ++j) {
      const int wi = wi0 + j;
      in[j] = (wi >= 0 && wi < Wi) ? ld4(xrow + (long long)wi * C) : f4zero();
    }
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      const float4 wv = ld4(w + (ky * K + kx) * C + c);
#pragma unroll
      for (int t = 0; t < TW; ++t) acc[t] = f4fma(in[t * S + kx], wv, acc[t]);
    }
  }
  }
  float* yrow = y + ((long long)(n * Ho + ho) * Wo) * C + c;
#pragma unroll
  for (int t = 0; t < TW; ++t)
    if (wo0 + t < Wo) st4(yrow + (long long)(wo0 + t) * C, acc[t]);
}

// Forward + the following batch norm's stage-1 statistics (training): grid = (strip blocks, channel groups of 32), 256 threads =
// 8 channel quads x 32 strips, so the strips of a block share their 32 channels and their {sum y, sum y^2} are folded in the block
// (lane butterfly, then the 4 waves through LDS in a fixed order) into stats_part [strip block][2][C] -- the statistics pass over
// the depthwise output and its launch disappear.
template <int K, int S, int TW>
__global__ __launch_bounds__(256) void dwconv_fwd_stats_k(const float* __restrict__ x, const float* __restrict__ w,
                                                          float* __restrict__ y, int N, int Hi, int Wi, int Ho, int Wo, int C, int pt,
                                                          int pl, float* __restrict__ stats_part) {
  constexpr int IW = (TW - 1) * S + K;
  __shared__ float4 sm[2][4][8];
  const int t = threadIdx.x, q = t & 7, sl = t >> 3;
  const int strips = (Wo + TW - 1) / TW;
  const unsigned total = (unsigned)N * Ho * strips;
  const unsigned bx = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned item = bx * 32u + sl;
  const int c = blockIdx.y * 32 + q * 4;
  const bool active = (item < total) && (c < C);
  float4 s1 = f4zero(), s2 = f4zero();
  if (active) {
    const int sx = (int)(item % (unsigned)strips);
    const unsigned r = item / (unsigned)strips;
    const int ho = (int)(r % (unsigned)Ho);
    const int n = (int)(r / (unsigned)Ho);
    const int wo0 = sx * TW;
    const int hi0 = ho * S - pt, wi0 = wo0 * S - pl;
    float4 acc[TW];
#pragma unroll
    for (int j = 0; j < TW; ++j) acc[j] = f4zero();
    if constexpr (K == 5) {   // whole window first: one memory round trip (see dwconv_fwd_k)
      float4 in[K][IW];
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        const int hi = hi0 + ky;
        const bool rok = (hi >= 0) && (hi < Hi);
        const float* xrow = x + ((long long)(n * Hi + (rok ? hi : 0)) * Wi) * C + c;
#pragma unroll
        for (int j = 0; j < IW; ++j) {
          const int wi = wi0 + j;
          const bool ok = rok && (wi >= 0) && (wi < Wi);
          const float4 v = ld4(ok ? xrow + (long long)wi * C : x + c);
          in[ky][j] = ok ? v : f4zero();
        }
      }
#pragma unroll
      for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const float4 wv = ld4(w + (ky * K + kx) * C + c);
#pragma unroll
          for (int j = 0; j < TW; ++j) acc[j] = f4fma(in[ky][j * S + kx], wv, acc[j]);
        }
    } else {
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        const int hi = hi0 + ky;
        if (hi < 0 || hi >= Hi) continue;
        const float* xrow = x + ((long long)(n * Hi + hi) * Wi) * C + c;
        float4 in[IW];
#pragma unroll
        for (int j = 0; j < IW; ++j) {
          const int wi = wi0 + j;
          in[j] = (wi >= 0 && wi < Wi) ? ld4(xrow + (long long)wi * C) : f4zero();
        }
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const float4 wv = ld4(w + (ky * K + kx) * C + c);
#pragma unroll
          for (int j = 0; j < TW; ++j) acc[j] = f4fma(in[j * S + kx], wv, acc[j]);
        }
      }
    }
    float* yrow = y + ((long long)(n * Ho + ho) * Wo) * C + c;
#pragma unroll
    for (int j = 0; j < TW; ++j)
      if (wo0 + j < Wo) {
        st4(yrow + (long long)(wo0 + j) * C, acc[j]);
        s1 = f4add(s1, acc[j]);
        s2 = f4fma(acc[j], acc[j], s2);
      }
  }
#pragma unroll
  for (int off = 8; off < 64; off <<= 1) {
    s1.x += __shfl_xor(s1.x, off); s1.y += __shfl_xor(s1.y, off); s1.z += __shfl_xor(s1.z, off); s1.w += __shfl_xor(s1.w, off);
    s2.x += __shfl_xor(s2.x, off); s2.y += __shfl_xor(s2.y, off); s2.z += __shfl_xor(s2.z, off); s2.w += __shfl_xor(s2.w, off);
  }
  if ((t & 63) < 8) {
    sm[0][t >> 6][q] = s1;
    sm[1][t >> 6][q] = s2;
  }
  __syncthreads();
  if (t < 16) {
    const int v = t >> 3, qq = t & 7;
    const int cc = blockIdx.y * 32 + qq * 4;
    if (cc < C)
      st4(stats_part + ((long long)bx * 2 + v) * C + cc, f4add(f4add(sm[v][0][qq], sm[v][1][qq]), f4add(sm[v][2][qq], sm[v][3][qq])));
  }
}

// 5x5 layers through an LDS tile.  The sliding-window kernels above fetch every input element K times per strip (10 window loads per
// output for K = 5: 23-42 % of the HBM roofline even on large launches); here a block stages the (TOH*S + K - S) x (TOW*S + K - S)
// input window of TOH x TOW outputs for 32 channels in LDS once (raw buffer loads, out-of-image pixels come back as zeros) and every
// output reads its K*K taps from there (ds_read_b128, 8 channel quads x 8 pixels per wave = contiguous 1 KB, conflict-free) against
// filter taps.  256 threads = 8 channel quads x 32 pixel lanes.  FLIP: taps reversed = backward-data (a stride-1 correlation with the
// rotated filter); DIL with it: over the zero-interleaved gradient of a stride-2 layer (75 % of the taps then multiply zeros, which
// is still 2.4x faster than the gather kernel: the launch is latency-bound, not FMA-bound).  STATS: also emits the following batch norm's stage-1 statistics [block][2][C].
typedef unsigned dw_u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kDwOob = 0xFFFFFFF0u;

// BNB (backward-data launches): the output is the gradient w.r.t. a = swish(bn(z)); the launch then also emits stage 1 of that batch
// norm's backward -- per-block column sums {sum g, sum g * xhat}, g = dx * swish'(gamma * xhat + beta), xhat = (z - mean) * rstd -- into
// stats_part [block][2][C], so mliis_bn_bwd needs no reduce pass of its own over (z, dx).
struct DwBn {
  const float* z;       // [N, Ho, Wo, C] the batch norm's input (same shape as this launch's output)
  const float *mean, *rstd, *gamma, *beta;
};

template <int K, int S, int TOH, int TOW, int TS, bool FLIP, bool STATS, bool DIL, bool BNB>
__global__ __launch_bounds__(256) void dwconv_tile_k(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                     int Hi, int Wi, int Ho, int Wo, int C, int pt, int pl, int tiles_y, int tiles_x,
                                                     float* __restrict__ stats_part, DwBn bn) {
  // every thread produces a strip of TS outputs along W from a sliding (TS-1)*S+K wide register window: K*((TS-1)*S+K)/TS LDS reads per
  // output instead of K*K (10 instead of 25 for K = 5, TS = 4 at stride 1 -- the LDS port is the bottleneck otherwise)
  constexpr int IH = (TOH - 1) * S + K, IW = (TOW - 1) * S + K, NPIX = IH * IW;
  constexpr int SPR = TOW / TS, NITEM = TOH * SPR;            // strips per row, strips per tile
  constexpr int LD_PER = (NPIX + 31) / 32, IT_PER = (NITEM + 31) / 32, WW = (TS - 1) * S + K;
  static_assert(TOW % TS == 0, "tile width must be a multiple of the strip");
  __shared__ float4 sm[NPIX * 8];
  __shared__ float4 red[2][4][8];
  const int t = threadIdx.x, q = t & 7, lane_p = t >> 3;
  const unsigned bx = blockIdx.x;
  const int tx = (int)(bx % (unsigned)tiles_x);
  const unsigned r = bx / (unsigned)tiles_x;
  const int ty = (int)(r % (unsigned)tiles_y), n = (int)(r / (unsigned)tiles_y);
  const int c = blockIdx.y * 32 + q * 4;
  const bool cok = c < C;
  const int oy0 = ty * TOH, ox0 = tx * TOW;
  const int hi0 = oy0 * S - pt, wi0 = ox0 * S - pl;
  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)x, 0, 0x80000000u, 0x00020000);
  float4 v[LD_PER];
#pragma unroll
  for (int i = 0; i < LD_PER; ++i) {
    const int idx = lane_p + 32 * i;
    const int iy = idx / IW, ix = idx - iy * IW;
    int hi = hi0 + iy, wi = wi0 + ix;
    bool ok = cok & (idx < NPIX);
    if (DIL) {   // the input is read as if a zero row / column stood between its pixels (backward-data of a stride-2 layer)
      ok &= ((hi | wi) & 1) == 0;
      hi >>= 1;
      wi >>= 1;
    }
    ok &= ((unsigned)hi < (unsigned)Hi) & ((unsigned)wi < (unsigned)Wi);
    const unsigned off = (unsigned)((((n * Hi + hi) * Wi + wi) * C + c) * 4);   // host guarantees the tensor is below 2 GiB
    const dw_u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(rX, (int)(ok ? off : kDwOob), 0, 0);
    v[i] = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
  }
  const float* wc = w + (cok ? c : 0);
#pragma unroll
  for (int i = 0; i < LD_PER; ++i) {
    const int idx = lane_p + 32 * i;
    if (idx < NPIX) sm[idx * 8 + q] = v[i];
  }
  __syncthreads();
  float4 s1 = f4zero(), s2 = f4zero();
  float4 bm = f4zero(), br = f4zero(), bg = f4zero(), bb = f4zero();
  if (BNB) {
    const int cc = cok ? c : 0;
    bm = ld4(bn.mean + cc); br = ld4(bn.rstd + cc); bg = ld4(bn.gamma + cc); bb = ld4(bn.beta + cc);
  }
#pragma unroll
  for (int i = 0; i < IT_PER; ++i) {
    const int item = lane_p + 32 * i;
    const int it = item < NITEM ? item : 0;
    const int oy = it / SPR, ox = (it - oy * SPR) * TS;
    const float4* base = sm + ((oy * S) * IW + ox * S) * 8 + q;
    float4 acc[TS];
#pragma unroll
    for (int j = 0; j < TS; ++j) acc[j] = f4zero();
#pragma unroll 1   // (fully unrolled the compiler hoists all K*WW window reads: 260 VGPRs, one wave per SIMD)
    for (int ky = 0; ky < K; ++ky) {
      float4 in[WW], wr[K];   // one filter row at a time (L1-resident): all K*K taps in registers cost 100 VGPRs and the occupancy
#pragma unroll
      for (int kx = 0; kx < K; ++kx) wr[kx] = ld4(wc + (FLIP ? K * K - 1 - (ky * K + kx) : ky * K + kx) * C);
#pragma unroll
      for (int j = 0; j < WW; ++j) in[j] = base[(ky * IW + j) * 8];
#pragma unroll
      for (int kx = 0; kx < K; ++kx)
#pragma unroll
        for (int j = 0; j < TS; ++j) acc[j] = f4fma(in[j * S + kx], wr[kx], acc[j]);
    }
    const bool rok = cok && item < NITEM && oy0 + oy < Ho;
#pragma unroll
    for (int j = 0; j < TS; ++j)
      if (rok && ox0 + ox + j < Wo) {
        const long long o4 = ((long long)((n * Ho + oy0 + oy) * Wo + ox0 + ox + j)) * C + c;
        st4(y + o4, acc[j]);
        if (STATS) {
          s1 = f4add(s1, acc[j]);
          s2 = f4fma(acc[j], acc[j], s2);
        }
        if (BNB) {
          const float4 zv = ld4(bn.z + o4);
          const float4 xh = make_float4((zv.x - bm.x) * br.x, (zv.y - bm.y) * br.y, (zv.z - bm.z) * br.z, (zv.w - bm.w) * br.w);
          float4 gg = acc[j];
          gg.x *= swish_grad_f(fmaf(xh.x, bg.x, bb.x));
          gg.y *= swish_grad_f(fmaf(xh.y, bg.y, bb.y));
          gg.z *= swish_grad_f(fmaf(xh.z, bg.z, bb.z));
          gg.w *= swish_grad_f(fmaf(xh.w, bg.w, bb.w));
          s1 = f4add(s1, gg);
          s2 = f4fma(gg, xh, s2);
        }
      }
  }
  if (!STATS && !BNB) return;
#pragma unroll
  for (int off = 8; off < 64; off <<= 1) {
    s1.x += __shfl_xor(s1.x, off); s1.y += __shfl_xor(s1.y, off); s1.z += __shfl_xor(s1.z, off); s1.w += __shfl_xor(s1.w, off);
    s2.x += __shfl_xor(s2.x, off); s2.y += __shfl_xor(s2.y, off); s2.z += __shfl_xor(s2.z, off); s2.w += __shfl_xor(s2.w, off);
  }
  if ((t & 63) < 8) {
    red[0][t >> 6][q] = s1;
    red[1][t >> 6][q] = s2;
  }
  __syncthreads();
  if (t < 16) {
    const int vv = t >> 3, qq = t & 7;
    const int cc = blockIdx.y * 32 + qq * 4;
    if (cc < C)
      st4(stats_part + ((long long)bx * 2 + vv) * C + cc, f4add(f4add(red[vv][0][qq], red[vv][1][qq]), f4add(red[vv][2][qq], red[vv][3][qq])));
  }
}

// dx[n,hi,wi,c] = sum_{ky,kx : (hi+pt-ky) % S == 0, (wi+pl-kx) % S == 0} dy[n,(hi+pt-ky)/S,(wi+pl-kx)/S,c] * w[ky,kx,c]
template <int K, int S, int TW>
__global__ __launch_bounds__(256) void dwconv_bwd_data_k(const float* __restrict__ dy, const float* __restrict__ w,
                                                         float* __restrict__ dx, int N, int Hi, int Wi, int Ho, int Wo, int C,
                                                         int pt, int pl) {
  constexpr int JW = TW + K - 1;
  const int Q = C >> 2;
  const int strips = (Wi + TW - 1) / TW;
  const unsigned total = (unsigned)N * Hi * strips * Q;
  const unsigned i = xcd_remap(blockIdx.x, gridDim.x) * 256u + threadIdx.x;
  if (i >= total) return;
  const int cq = (int)(i % (unsigned)Q);
  unsigned r = i / (unsigned)Q;
  const int sx = (int)(r % (unsigned)strips);
  r /= (unsigned)strips;
  const int hi = (int)(r % (unsigned)Hi);
  const int n = (int)(r / (unsigned)Hi);
  const int c = cq << 2;
  const int wi0 = sx * TW;
  const int base = wi0 + pl - (K - 1);  // wx of window slot j is base + j ; tap kx pairs output t with slot t + K-1-kx
  float4 acc[TW];
#pragma unroll
  for (int t = 0; t < TW; ++t) acc[t] = f4zero();
  if constexpr (K == 5) {   // all loads of the 5 x JW window first (see dwconv_fwd_k)
    float4 dv[K][JW];
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
      const int hy = hi + pt - ky;
      const bool rok = (hy >= 0) && ((hy % S) == 0) && ((hy / S) < Ho);
      const float* drow = dy + ((long long)(n * Ho + (rok ? hy / S : 0)) * Wo) * C + c;
#pragma unroll
      for (int j = 0; j < JW; ++j) {
        const int wx = base + j;
        const bool ok = rok && (wx >= 0) && ((wx % S) == 0) && ((wx / S) < Wo);
        const float4 v = ld4(ok ? drow + (long long)(wx / S) * C : dy + c);
        dv[ky][j] = ok ? v : f4zero();
      }
    }
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        const float4 wv = ld4(w + (ky * K + kx) * C + c);
#pragma unroll
        for (int t = 0; t < TW; ++t) acc[t] = f4fma(dv[ky][t + K - 1 - kx], wv, acc[t]);
      }
  } else {
#pragma unroll
  for (int ky = 0; ky < K; ++ky) {
    const int hy = hi + pt - ky;
    if (hy < 0 || (hy % S) != 0) continue;
    const int ho = hy / S;
    if (ho >= Ho) continue;
    const float* drow = dy + ((long long)(n * Ho + ho) * Wo) * C + c;
    float4 dv[JW];
#pragma unroll
    for (int j = 0; j < JW; ++j) {
      const int wx = base + j;
      const bool ok = (wx >= 0) && ((wx % S) == 0) && ((wx / S) < Wo);
      dv[j] = ok ? ld4(drow + (long long)(wx / S) * C) : f4zero();
    }
#pragma unroll
    for (int kx = 0; kx < K; ++kx) {
      const float4 wv = ld4(w + (ky * K + kx) * C + c);
#pragma unroll
      for (int t = 0; t < TW; ++t) acc[t] = f4fma(dv[t + K - 1 - kx], wv, acc[t]);
    }
  }
  }
  float* xrow = dx + ((long long)(n * Hi + hi) * Wi) * C + c;
#pragma unroll
  for (int t = 0; t < TW; ++t)
    if (wi0 + t < Wi) st4(xrow + (long long)(wi0 + t) * C, acc[t]);
}

// Stage 1 of dw[ky,kx,c] = sum_{n,ho,wo} x[n,ho*S-pt+ky,wo*S-pl+kx,c] * dy[n,ho,wo,c].
// Block = QB channel quads x RP strip-lanes; every thread walks strips item = rl, rl+RP, ... of the block's range keeping
// K*K float4 accumulators, then the RP lanes of each quad are folded through LDS.  part layout [blk][K*K][C].
template <int K, int S, int TW>
__global__ __launch_bounds__(256) void dwconv_bwd_filter_k(const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ part, int N, int Hi, int Wi, int Ho, int Wo,
                                                           int C, int pt, int pl, int QB, int RP, int items_per_block) {
  constexpr int IW = (TW - 1) * S + K;
  __shared__ float4 sm[8 * 256];
  const int t = threadIdx.x;
  const int ql = t % QB, rl = t / QB;
  const int Q = C >> 2;
  const int q = blockIdx.y * QB + ql;
  const int c = q << 2;
  const int strips = (Wo + TW - 1) / TW;
  const long long items = (long long)N * Ho * strips;
  const bool active = (rl < RP) && (q < Q);
  float4 acc[K * K];
#pragma unroll
  for (int k = 0; k < K * K; ++k) acc[k] = f4zero();
  if (active) {
    const long long i0 = (long long)xcd_remap(blockIdx.x, gridDim.x) * items_per_block;
    long long i1 = i0 + items_per_block;
    if (i1 > items) i1 = items;
    for (long long it = i0 + rl; it < i1; it += RP) {
      const int sx = (int)(it % strips);
      const long long r = it / strips;
      const int ho = (int)(r % Ho);
      const int n = (int)(r / Ho);
      const int wo0 = sx * TW;
      const int hi0 = ho * S - pt, wi0 = wo0 * S - pl;
      const float* drow = dy + ((long long)(n * Ho + ho) * Wo) * C + c;
      float4 dv[TW];
#pragma unroll
      for (int j = 0; j < TW; ++j) dv[j] = (wo0 + j < Wo) ? ld4(drow + (long long)(wo0 + j) * C) : f4zero();
#pragma unroll
      for (int ky = 0; ky < K; ++ky) {
        const int hi = hi0 + ky;
        if (hi < 0 || hi >= Hi) continue;
        const float* xrow = x + ((long long)(n * Hi + hi) * Wi) * C + c;
        float4 in[IW];
#pragma unroll
        for (int j = 0; j < IW; ++j) {
          const int wi = wi0 + j;
          in[j] = (wi >= 0 && wi < Wi) ? ld4(xrow + (long long)wi * C) : f4zero();
        }
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
#pragma unroll
          for (int j = 0; j < TW; ++j) acc[ky * K + kx] = f4fma(in[j * S + kx], dv[j], acc[ky * K + kx]);
        }
      }
    }
  }
  // fold the RP strip lanes of every quad through LDS, TB taps per barrier pair (K*K barriers would dominate small layers)
  constexpr int TB = 8;
#pragma unroll
  for (int k0 = 0; k0 < K * K; k0 += TB) {
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < TB; ++kk)
      if (k0 + kk < K * K) sm[kk * 256 + t] = acc[k0 + kk];
    __syncthreads();
    for (int item = t; item < TB * QB; item += 256) {
      const int kk = item / QB, q2 = item - kk * QB;
      const int qq = blockIdx.y * QB + q2;
      if (k0 + kk < K * K && qq < Q) {
        float4 s4 = sm[kk * 256 + q2];
        for (int j = 1; j < RP; ++j) s4 = f4add(s4, sm[kk * 256 + j * QB + q2]);
        st4(part + ((long long)blockIdx.x * (K * K) + k0 + kk) * C + (qq << 2), s4);
      }
    }
  }
}

struct DwGeom {
  int Ho, Wo, pt, pl;
};
static inline DwGeom dw_geom(int Hi, int Wi, int K, int S) {
  DwGeom g;
  g.Ho = (Hi + S - 1) / S;
  g.Wo = (Wi + S - 1) / S;
  int th = (g.Ho - 1) * S + K - Hi;
  int tw = (g.Wo - 1) * S + K - Wi;
  if (th < 0) th = 0;
  if (tw < 0) tw = 0;
  g.pt = th / 2;
  g.pl = tw / 2;
  return g;
}

struct DwFilterGeom {
  int QB, RP, items_per_block, nblk, ny;
};
static inline DwFilterGeom dw_filter_geom(int N, int Ho, int Wo, int C, int TW) {
  DwFilterGeom g;
  int Q = C / 4;
  g.QB = Q < 64 ? Q : 64;
  g.RP = 256 / g.QB;
  g.ny = ceil_div(Q, g.QB);
  long long items = (long long)N * Ho * ((Wo + TW - 1) / TW);
  // measured (tools/dwf_sweep.py): the kernel is latency-bound on every EfficientLab layer -- as many blocks as there are strips to
  // hand out (one strip per thread on the small maps) beats longer per-thread walks by 1.4-2x; the extra slabs go to the batched fold
  long long want = 512 / g.ny;   // (whole step on one box: 128 -> 2395, 256 -> 2418, 384 -> 2414, 512 -> 2429, 768 -> 2417, 1024 -> 2407,
                                 //  2048 -> 2411 images/s: the slabs are read again by the batched fold)
  if (want < 1) want = 1;
  long long ipb = (items + want - 1) / want;
  long long minr = (long long)g.RP;
  if (ipb < minr) ipb = minr;
  ipb = (ipb + g.RP - 1) / g.RP * g.RP;
  g.items_per_block = (int)ipb;
  g.nblk = ceil_div(items, ipb);
  return g;
}

// LDS-tile kernel geometry for the 5x5 layers (dwconv_tile_k): 7 x 16 output tiles in strips of 4 at stride 1, 7 x 7 singles at stride 2
struct DwTile {
  int toh, tow, tiles_y, tiles_x;
  long long blocks;
};
static inline DwTile dw_tile(int N, int Ho, int Wo, int stride) {
  DwTile d;
  d.toh = 7;
  d.tow = stride == 1 ? 16 : 7;
  d.tiles_y = ceil_div(Ho, d.toh);
  d.tiles_x = ceil_div(Wo, d.tow);
  d.blocks = (long long)N * d.tiles_y * d.tiles_x;
  return d;
}
template <bool FLIP, bool STATS, bool DIL = false, bool BNB = false>
static void launch_dw_tile(const DwTile& d, int stride, int C, const float* x, const float* w, float* y, int Hi, int Wi, int Ho, int Wo, int pt,
                           int pl, float* stats_part, hipStream_t stream, int k = 5, DwBn bn = DwBn{nullptr, nullptr, nullptr, nullptr, nullptr}) {
  dim3 grid((unsigned)d.blocks, ceil_div(C, 32));
  if (k == 3) {   // (stride-1 geometry: backward-data launches only)
    hipLaunchKernelGGL((dwconv_tile_k<3, 1, 7, 16, 4, FLIP, STATS, DIL, BNB>), grid, dim3(256), 0, stream, x, w, y, Hi, Wi, Ho, Wo, C, pt, pl,
                       d.tiles_y, d.tiles_x, stats_part, bn);
    return;
  }
  if (stride == 1)
    hipLaunchKernelGGL((dwconv_tile_k<5, 1, 7, 16, 4, FLIP, STATS, DIL, BNB>), grid, dim3(256), 0, stream, x, w, y, Hi, Wi, Ho, Wo, C, pt, pl,
                       d.tiles_y, d.tiles_x, stats_part, bn);
  else
    hipLaunchKernelGGL((dwconv_tile_k<5, 2, 7, 7, 1, FLIP, STATS, false, false>), grid, dim3(256), 0, stream, x, w, y, Hi, Wi, Ho, Wo, C, pt, pl,
                       d.tiles_y, d.tiles_x, stats_part, bn);
}

constexpr int kTW = 4;

}  // namespace mliis

using namespace mliis;

#define DW_DISPATCH(KERNEL, ...)                                                                              \
  do {                                                                                                        \
    if (k == 3 && stride == 1) hipLaunchKernelGGL((KERNEL<3, 1, kTW>), grid, dim3(256), 0, stream, __VA_ARGS__);      \
    else if (k == 3 && stride == 2) hipLaunchKernelGGL((KERNEL<3, 2, kTW>), grid, dim3(256), 0, stream, __VA_ARGS__); \
    else if (k == 5 && stride == 1) hipLaunchKernelGGL((KERNEL<5, 1, kTW>), grid, dim3(256), 0, stream, __VA_ARGS__); \
    else hipLaunchKernelGGL((KERNEL<5, 2, kTW>), grid, dim3(256), 0, stream, __VA_ARGS__);                            \
  } while (0)

static int dw_check(const char* name, const void* a, const void* b, const void* c, int N, int H, int W, int C, int k, int stride) {
  MLIIS_REQUIRE(a && b && c, MLIIS_ERR_ARG, "%s: null pointer", name);
  MLIIS_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0, MLIIS_ERR_ARG, "%s: bad shape N=%d H=%d W=%d C=%d (C %% 4 must be 0)",
                name, N, H, W, C);
  MLIIS_REQUIRE((k == 3 || k == 5) && (stride == 1 || stride == 2), MLIIS_ERR_UNSUPPORTED, "%s: unsupported k=%d stride=%d", name, k,
                stride);
  MLIIS_REQUIRE(aligned16(a) && aligned16(b) && aligned16(c), MLIIS_ERR_ALIGN, "%s: pointers must be 16-byte aligned", name);
  return MLIIS_OK;
}

extern "C" {

int mliis_dwconv_fwd(const float* x, const float* w, float* y, int N, int H, int W, int C, int k, int stride, float* stats_part,
                     size_t stats_floats, int* stats_nblk, hipStream_t stream) {
  int rc = dw_check("dwconv_fwd", x, w, y, N, H, W, C, k, stride);
  if (rc) return rc;
  DwGeom g = dw_geom(H, W, k, stride);
  const long long strips = (long long)N * g.Ho * ((g.Wo + kTW - 1) / kTW);
  long long total = strips * (C / 4);
  MLIIS_REQUIRE(total < (1LL << 31), MLIIS_ERR_UNSUPPORTED, "dwconv_fwd: tensor too large for 32-bit indexing");
  // 5x5 layers: LDS-tile kernel (tensors below 2 GiB: 32-bit buffer offsets; with statistics only while its block count fits the
  // documented statistics buffer)
  const DwTile tile = dw_tile(N, g.Ho, g.Wo, stride);
  // (stride-1 layers; the stride-2 forward measured 11.2 us through the tile against 10.0 us through the sliding window inside the step)
  const bool tiled = k == 5 && stride == 1 && (long long)N * H * W * C * 4 < (1LL << 31) && tile.blocks < (1LL << 31);
  if (stats_part != nullptr) {   // training: the following batch norm's stage-1 statistics come out of the same launch
    MLIIS_REQUIRE(stats_nblk && aligned16(stats_part), MLIIS_ERR_ARG, "dwconv_fwd: statistics need a 16-byte aligned buffer and a stats_nblk output");
    const int nblk = (int)((strips + 31) / 32);
    MLIIS_REQUIRE((size_t)nblk * 2 * C <= stats_floats, MLIIS_ERR_WORKSPACE, "dwconv_fwd: statistics buffer too small (%zu floats needed, %zu given)",
                  (size_t)nblk * 2 * C, stats_floats);
    if (tiled && (size_t)tile.blocks * 2 * C <= stats_floats) {
      launch_dw_tile<false, true>(tile, stride, C, x, w, y, H, W, g.Ho, g.Wo, g.pt, g.pl, stats_part, stream, k);
      MLIIS_CHECK_LAUNCH("dwconv_fwd_tile_stats");
      *stats_nblk = (int)tile.blocks;
      return MLIIS_OK;
    }
    dim3 grid(nblk, ceil_div(C, 32));
    DW_DISPATCH(dwconv_fwd_stats_k, x, w, y, N, H, W, g.Ho, g.Wo, C, g.pt, g.pl, stats_part);
    MLIIS_CHECK_LAUNCH("dwconv_fwd_stats");
    *stats_nblk = nblk;
    return MLIIS_OK;
  }
  if (stats_nblk) *stats_nblk = 0;
  if (tiled) {
    launch_dw_tile<false, false>(tile, stride, C, x, w, y, H, W, g.Ho, g.Wo, g.pt, g.pl, nullptr, stream, k);
    MLIIS_CHECK_LAUNCH("dwconv_fwd_tile");
    return MLIIS_OK;
  }
  dim3 grid(ceil_div(total, 256));
  DW_DISPATCH(dwconv_fwd_k, x, w, y, N, H, W, g.Ho, g.Wo, C, g.pt, g.pl);
  MLIIS_CHECK_LAUNCH("dwconv_fwd");
  return MLIIS_OK;
}

// bn_z != NULL: dx is the gradient w.r.t. swish(bn(bn_z)) (the expand branch of an MBConv block): the launch also emits stage 1 of that
// batch norm's backward into part [*nblk][2][C] (see DwBn); *nblk == 0 means "not produced" (no LDS-tile path for this call or part
// too small) and the caller lets mliis_bn_bwd run its own reduce pass.
static int dw_bwd_data(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int k, int stride, const DwBn* bn, float* part,
                       size_t part_floats, int* nblk, hipStream_t stream) {
  int rc = dw_check("dwconv_bwd_data", dy, w, dx, N, H, W, C, k, stride);
  if (rc) return rc;
  DwGeom g = dw_geom(H, W, k, stride);
  long long total = (long long)N * H * ((W + kTW - 1) / kTW) * (C / 4);
  MLIIS_REQUIRE(total < (1LL << 31), MLIIS_ERR_UNSUPPORTED, "dwconv_bwd_data: tensor too large for 32-bit indexing");
  if (nblk) *nblk = 0;
  const DwTile tile = dw_tile(N, H, W, 1);
  const bool small = (long long)N * H * W * C * 4 < (1LL << 31) && tile.blocks < (1LL << 31);
  const bool fuse = bn != nullptr && small && nblk != nullptr && part != nullptr && aligned16(part) && (size_t)tile.blocks * 2 * C <= part_floats;
  // dx = dy (zero-interleaved when the layer has stride 2) correlated with the rotated filter, padding K-1-pt / K-1-pl: the LDS-tile
  // kernel with FLIP (and DIL) -- for the 5x5 layers always, for the 3x3 layers (where it merely matches the sliding-window kernel)
  // when the BN-backward statistics ride along
  if (small && (k == 5 || fuse)) {
    const int qt = k - 1 - g.pt, ql = k - 1 - g.pl;
    if (fuse) {
      if (stride == 1) launch_dw_tile<true, false, false, true>(tile, 1, C, dy, w, dx, g.Ho, g.Wo, H, W, qt, ql, part, stream, k, *bn);
      else launch_dw_tile<true, false, true, true>(tile, 1, C, dy, w, dx, g.Ho, g.Wo, H, W, qt, ql, part, stream, k, *bn);
      *nblk = (int)tile.blocks;
    } else {
      if (stride == 1) launch_dw_tile<true, false, false>(tile, 1, C, dy, w, dx, g.Ho, g.Wo, H, W, qt, ql, nullptr, stream, k);
      else launch_dw_tile<true, false, true>(tile, 1, C, dy, w, dx, g.Ho, g.Wo, H, W, qt, ql, nullptr, stream, k);
    }
    MLIIS_CHECK_LAUNCH("dwconv_bwd_data_tile");
    return MLIIS_OK;
  }
  dim3 grid(ceil_div(total, 256));
  DW_DISPATCH(dwconv_bwd_data_k, dy, w, dx, N, H, W, g.Ho, g.Wo, C, g.pt, g.pl);
  MLIIS_CHECK_LAUNCH("dwconv_bwd_data");
  return MLIIS_OK;
}

int mliis_dwconv_bwd_data(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int k, int stride,
                          hipStream_t stream) {
  return dw_bwd_data(dy, w, dx, N, H, W, C, k, stride, nullptr, nullptr, 0, nullptr, stream);
}

int mliis_dwconv_bwd_data_bn(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int k, int stride, const float* bn_z,
                             const float* bn_mean, const float* bn_rstd, const float* bn_gamma, const float* bn_beta, float* part,
                             size_t part_floats, int* nblk, hipStream_t stream) {
  MLIIS_REQUIRE(bn_z && bn_mean && bn_rstd && bn_gamma && bn_beta && part && nblk, MLIIS_ERR_ARG, "dwconv_bwd_data_bn: null pointer");
  MLIIS_REQUIRE(aligned16(bn_z) && aligned16(bn_mean) && aligned16(bn_rstd) && aligned16(bn_gamma) && aligned16(bn_beta), MLIIS_ERR_ALIGN,
                "dwconv_bwd_data_bn: pointers must be 16-byte aligned");
  const DwBn bn{bn_z, bn_mean, bn_rstd, bn_gamma, bn_beta};
  return dw_bwd_data(dy, w, dx, N, H, W, C, k, stride, &bn, part, part_floats, nblk, stream);
}

size_t mliis_dwconv_bwd_filter_workspace_floats(int N, int H, int W, int C, int k, int stride) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return 0;
  DwGeom g = dw_geom(H, W, k, stride);
  DwFilterGeom f = dw_filter_geom(N, g.Ho, g.Wo, C, kTW);
  return (size_t)f.nblk * k * k * C;
}

int mliis_dwconv_bwd_filter(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int k, int stride, float* ws,
                            size_t ws_floats, hipStream_t stream) {
  int rc = dw_check("dwconv_bwd_filter", x, dy, dw ? dw : ws, N, H, W, C, k, stride);   // dw == NULL: slabs stay in ws
  if (rc) return rc;
  MLIIS_REQUIRE(ws && aligned16(ws), MLIIS_ERR_ARG, "dwconv_bwd_filter: workspace null/unaligned");
  DwGeom g = dw_geom(H, W, k, stride);
  DwFilterGeom f = dw_filter_geom(N, g.Ho, g.Wo, C, kTW);
  size_t need = (size_t)f.nblk * k * k * C;
  MLIIS_REQUIRE(need <= ws_floats, MLIIS_ERR_WORKSPACE, "dwconv_bwd_filter: workspace too small (%zu needed, %zu given)", need,
                ws_floats);
  dim3 grid(f.nblk, f.ny);
  DW_DISPATCH(dwconv_bwd_filter_k, x, dy, ws, N, H, W, g.Ho, g.Wo, C, g.pt, g.pl, f.QB, f.RP, f.items_per_block);
  MLIIS_CHECK_LAUNCH("dwconv_bwd_filter");
  if (dw == nullptr) return MLIIS_OK;
  hipLaunchKernelGGL(fold_flat_k, dim3(ceil_div(k * k * C, kFoldX)), dim3(kFoldX, kFoldY), 0, stream, ws, f.nblk, (long long)k * k * C, 1.0f,
                     dw, 0, (long long)k * k * C, 0LL, 0LL);
  MLIIS_CHECK_LAUNCH("dwconv_bwd_filter_finalize");
  return MLIIS_OK;
}
}
