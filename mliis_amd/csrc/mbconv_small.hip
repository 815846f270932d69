// The depthwise half of an MBConv block on SMALL maps (14x14 at 224x224 inputs: blocks 6-10 of EfficientLab-6-3), one launch per
// direction.  Reference ops: models/efficientnet/efficientnet_model.py:183-200,266-271 (expand BN -> swish -> depthwise k x k -> BN ->
// swish -> squeeze-excite mean) and models/efficientnet/utils.py:87-134 (training-mode batch norm + moving averages).
//
// Why: on a [8,14,14,480..672] tensor (3-4 MB) every launch of the op-by-op chain  bn0 apply -> depthwise(+statistics) -> bn1
// apply(+pooling)  and, backward,  bn1 reduce -> bn1 apply -> depthwise filter gradient -> depthwise backward-data(+bn0 sums) -> bn0
// apply  costs 5-10 us of dependent memory round trips for < 1 us of traffic (profiles/r01_final_profile.md).  Everything in those
// chains is PER CHANNEL: batch-norm statistics and gradient sums run over (N, H, W) of one channel, the depthwise stencil stays inside
// a channel.  So a workgroup that owns a group of 8 channels for the whole [N, H, W] extent needs no grid-wide dependency at all:
//   forward : fold the expand conv's stage-1 statistics -> a0 = swish(bn0(z0)) into LDS -> stencil out of LDS -> exact two-pass statistics
//             of z1 in the workgroup -> a1 = swish(bn1(z1)) -> per-image means for the squeeze-excite -> both moving averages;
//   backward: bn1 backward (both sums + apply) -> dz1 tile in LDS -> depthwise filter gradient (complete, no slabs) and backward-data out
//             of LDS -> bn0 backward (both sums + apply) -> dz0.
// Layout: 1024 threads = 2 channel quads x 512 pixel lanes; a lane owns a vertical strip of 4 pixels (lanes run along W, so the LDS
// tiles [pixel][8 channels] are read as contiguous, conflict-free 16-byte words); LDS holds two [N*H*W][8] fp32 tiles.  Eligible:
// stride 1, N*H*W <= 2048 pixels, N*ceil(H/4)*W <= 512 strips, C % 8 == 0 -- anything else takes the op-by-op path.
// Workgroup -> channel-group mapping keeps the four groups of a 128-byte line (32 channels) on one XCD (speed only).
#include "common.hpp"

namespace mliis {

constexpr int kSmThreads = 1024;
constexpr int kSmLanes = 512;      // pixel lanes (strips) per workgroup
constexpr int kSmMaxPix = 2048;    // N*H*W: the backward kernel keeps two [N*H*W][8] fp32 tiles (128 KB) + 14 KB of scratch in LDS

struct SmallGeom {
  int N, H, W, C, HS, nitems, npix;
};

// channel group of this workgroup (or -1): groups of one 32-channel line share an XCD under round-robin dispatch
__device__ __forceinline__ int sm_channel_group(int C) {
  const int lines = (C + 31) >> 5;
  const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
  const int line = (j >> 2) * 8 + xcd, sub = j & 3;
  if (line >= lines) return -1;
  const int cg = line * 4 + sub;
  return cg * 8 < C ? cg : -1;
}
static inline int sm_grid(int C) { return ((((C + 31) / 32) + 7) / 8) * 8 * 4; }

// sum of a float4 pair over all threads with the same quad parity (t & 1); every thread gets the totals.
// red: LDS float4 [16 waves][2 quads][2]; two barriers.
__device__ __forceinline__ void sm_block_sum2(float4& a, float4& b, float4* red) {
#pragma unroll
  for (int off = 2; off < 64; off <<= 1) {
    a.x += __shfl_xor(a.x, off); a.y += __shfl_xor(a.y, off); a.z += __shfl_xor(a.z, off); a.w += __shfl_xor(a.w, off);
    b.x += __shfl_xor(b.x, off); b.y += __shfl_xor(b.y, off); b.z += __shfl_xor(b.z, off); b.w += __shfl_xor(b.w, off);
  }
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, q = t & 1;
  if (lane < 2) {
    red[(wave * 2 + q) * 2 + 0] = a;
    red[(wave * 2 + q) * 2 + 1] = b;
  }
  __syncthreads();
  float4 sa = f4zero(), sb = f4zero();
#pragma unroll 4   // (fully unrolled the 32 LDS reads are issued together: 128 VGPRs)
  for (int w = 0; w < kSmThreads / 64; ++w) {
    sa = f4add(sa, red[(w * 2 + q) * 2 + 0]);
    sb = f4add(sb, red[(w * 2 + q) * 2 + 1]);
  }
  a = sa;
  b = sb;
  __syncthreads();
}

// raw buffer access with 32-bit byte offsets (tensors < 2 GiB, checked on the host): one VGPR per address instead of a 64-bit pointer
// pair -- the kernels below run at the 128-VGPR budget of 1024-thread workgroups -- and an out-of-range offset reads zeros / drops the store
typedef unsigned sm_u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kSmOob = 0xFFFFFFF0u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sm_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x80000000u, 0x00020000);
}
__device__ __forceinline__ float4 sm_ld(__amdgpu_buffer_rsrc_t r, unsigned off) {
  const sm_u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
  return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
}
__device__ __forceinline__ void sm_st(__amdgpu_buffer_rsrc_t r, unsigned off, float4 v) {
  sm_u32x4 u;
  u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y); u.z = __float_as_uint(v.z); u.w = __float_as_uint(v.w);
  __builtin_amdgcn_raw_buffer_store_b128(u, r, (int)off, 0, 0);
}

__device__ __forceinline__ float4 f4swish(float4 v) { return make_float4(swish_f(v.x), swish_f(v.y), swish_f(v.z), swish_f(v.w)); }
__device__ __forceinline__ float4 f4swish_grad(float4 v) {
  return make_float4(swish_grad_f(v.x), swish_grad_f(v.y), swish_grad_f(v.z), swish_grad_f(v.w));
}
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

// ------------------------------------------------------------------------------------------------ forward
struct SmallFwd {
  const float* z0;      // [N,H,W,C] expand conv output
  const float* part0;   // [nblk0][2][C] stage-1 statistics of z0 {sum, sum of squares}
  int nblk0;
  const float *gamma0, *beta0;
  float *mean0, *rstd0, *mm0, *mv0;   // out: batch statistics (backward), moving averages updated in place (mm0 nullable)
  const float* w;       // [K,K,C]
  const float *gamma1, *beta1;
  float *mean1, *rstd1, *mm1, *mv1;
  float* a0;            // [N,H,W,C] out: swish(bn0(z0)) (nullable: the backward kernel recomputes it from z0)
  float* z1;            // [N,H,W,C] out: depthwise output
  float* a1;            // [N,H,W,C] out: swish(bn1(z1))
  float* s;             // [N,C] out: per-image mean of a1 (squeeze-excite input)
  SmallGeom g;
  float eps, one_minus_momentum;
};

template <int K>
__global__ __launch_bounds__(kSmThreads) void mbconv_dw_fwd_small_k(SmallFwd p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const SmallGeom g = p.g;
  float4* tileA = reinterpret_cast<float4*>(smem);                       // [npix][2] a0
  float4* wl = tileA + (size_t)g.npix * 2;                               // [K*K][2] filter taps
  float4* red = wl + K * K * 2;                                          // [16][2][2]
  float4* pool = red + 64;                                               // [512][2] per-strip sums of a1
  double* fold = reinterpret_cast<double*>(pool + kSmLanes * 2);         // [16][16]
  float* stat = reinterpret_cast<float*>(fold + 256);                    // mean0[8] rstd0[8]
  const int cg = sm_channel_group(g.C);
  if (cg < 0) return;
  const int t = threadIdx.x, q = t & 1, pl = t >> 1;
  const int c = cg * 8 + q * 4;
  const bool act = pl < g.nitems;
  int w_ = 0, hs = 0, n = 0;
  if (act) {
    w_ = pl % g.W;
    const int r = pl / g.W;
    hs = r % g.HS;
    n = r / g.HS;
  }
  const int h0 = hs * 4;
  // byte offset of the own strip's first pixel (same in every [N,H,W,C] tensor), row stride; rows beyond H / idle lanes: out of range
  const unsigned off0 = (unsigned)((((n * g.H + h0) * g.W + w_) * g.C + c) * 4), rstride = (unsigned)(g.W * g.C * 4);
  auto poff = [&](int j) { return (act && h0 + j < g.H) ? off0 + (unsigned)j * rstride : kSmOob; };
  // own pixels' z0 first (does not depend on the statistics fold)
  const __amdgpu_buffer_rsrc_t rz0 = sm_rsrc(p.z0);
  float4 zin[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) zin[j] = sm_ld(rz0, poff(j));
  if (t < K * K * 2) wl[t] = ld4(p.w + (long long)(t >> 1) * g.C + cg * 8 + (t & 1) * 4);
  // ---- fold the expand conv's stage-1 statistics of this group's 8 channels (double precision, fixed order)
  if (t < 256) {
    const int col = t & 15, lane16 = t >> 4;          // col = v * 8 + channel
    const int v = col >> 3, ch = col & 7;
    double acc = 0.0;
    for (int b = lane16; b < p.nblk0; b += 16) acc += (double)p.part0[((long long)b * 2 + v) * g.C + cg * 8 + ch];
    fold[lane16 * 16 + col] = acc;
  }
  __syncthreads();
  if (t < 8) {
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      s1 += fold[k * 16 + t];
      s2 += fold[k * 16 + 8 + t];
    }
    const double inv_n = 1.0 / (double)g.npix;
    const double m = s1 * inv_n;
    double var = s2 * inv_n - m * m;
    if (var < 0.0) var = 0.0;
    const float mf = (float)m, rf = (float)(1.0 / sqrt(var + (double)p.eps));
    stat[t] = mf;
    stat[8 + t] = rf;
    const int cc = cg * 8 + t;
    p.mean0[cc] = mf;
    p.rstd0[cc] = rf;
    if (p.mm0 != nullptr) {   // non-fused TpuBatchNormalization: the biased variance enters the moving average (utils.py:87-134)
      const float mm = p.mm0[cc], mv = p.mv0[cc];
      p.mm0[cc] = mm - (mm - mf) * p.one_minus_momentum;
      p.mv0[cc] = mv - (mv - (float)var) * p.one_minus_momentum;
    }
  }
  __syncthreads();
  {
    const float4 m0 = ld4(stat + q * 4), r0 = ld4(stat + 8 + q * 4), ga = ld4(p.gamma0 + c), be = ld4(p.beta0 + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (act && h0 + j < g.H) {
        const int pix = (n * g.H + h0 + j) * g.W + w_;
        float4 y;
        y.x = fmaf((zin[j].x - m0.x) * r0.x, ga.x, be.x);
        y.y = fmaf((zin[j].y - m0.y) * r0.y, ga.y, be.y);
        y.z = fmaf((zin[j].z - m0.z) * r0.z, ga.z, be.z);
        y.w = fmaf((zin[j].w - m0.w) * r0.w, ga.w, be.w);
        const float4 a = f4swish(y);
        tileA[pix * 2 + q] = a;
        if (p.a0 != nullptr) sm_st(sm_rsrc(p.a0), poff(j), a);
      }
    }
  }
  __syncthreads();
  // ---- depthwise stencil out of LDS: outputs (h0 + j, w_), j < 4; TF-SAME, stride 1.  One filter row per trip of a ROLLED loop (its K
  //      taps and one K-wide window row in registers): fully unrolled, the compiler hoists all K*K taps and spills (128-VGPR budget at
  //      1024 threads)
  constexpr int P = (K - 1) / 2;
  float4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = f4zero();
  if (act) {
    const float4* imgA = tileA + (size_t)n * g.H * g.W * 2 + q;
#pragma unroll 1
    for (int ky = 0; ky < K; ++ky) {
      float4 wr[K];
#pragma unroll
      for (int kx = 0; kx < K; ++kx) wr[kx] = wl[(ky * K + kx) * 2 + q];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int hi = h0 + j + ky - P;
        const bool rok = (unsigned)hi < (unsigned)g.H;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const int wi = w_ - P + kx;
          const float4 v = (rok && (unsigned)wi < (unsigned)g.W) ? imgA[(hi * g.W + wi) * 2] : f4zero();
          acc[j] = f4fma(v, wr[kx], acc[j]);
        }
        __builtin_amdgcn_sched_barrier(0);   // one window row in flight at a time (hoisting all 4 costs 60 more VGPRs and spills)
      }
    }
  }
  // ---- statistics of z1: exact two-pass in the workgroup (mean, then centred second moment)
  float4 s1 = f4zero(), s2 = f4zero();
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (act && h0 + j < g.H) {
      sm_st(sm_rsrc(p.z1), poff(j), acc[j]);
      s1 = f4add(s1, acc[j]);
    }
  sm_block_sum2(s1, s2, red);
  const float inv_n = 1.0f / (float)g.npix;
  const float4 m1 = f4scale(s1, inv_n);
  float4 d2 = f4zero(), dummy = f4zero();
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (act && h0 + j < g.H) {
      const float4 d = f4sub(acc[j], m1);
      d2 = f4fma(d, d, d2);
    }
  sm_block_sum2(d2, dummy, red);
  const float4 var1 = f4scale(d2, inv_n);
  const float4 r1 = make_float4(1.0f / sqrtf(var1.x + p.eps), 1.0f / sqrtf(var1.y + p.eps), 1.0f / sqrtf(var1.z + p.eps),
                                1.0f / sqrtf(var1.w + p.eps));
  if (pl == 0) {
    st4(p.mean1 + c, m1);
    st4(p.rstd1 + c, r1);
    if (p.mm1 != nullptr) {
      const float4 mm = ld4(p.mm1 + c), mv = ld4(p.mv1 + c);
      const float om = p.one_minus_momentum;
      st4(p.mm1 + c, make_float4(mm.x - (mm.x - m1.x) * om, mm.y - (mm.y - m1.y) * om, mm.z - (mm.z - m1.z) * om, mm.w - (mm.w - m1.w) * om));
      st4(p.mv1 + c, make_float4(mv.x - (mv.x - var1.x) * om, mv.y - (mv.y - var1.y) * om, mv.z - (mv.z - var1.z) * om,
                                 mv.w - (mv.w - var1.w) * om));
    }
  }
  // ---- a1 = swish(bn1(z1)), per-image pooled mean
  {
    const float4 ga = ld4(p.gamma1 + c), be = ld4(p.beta1 + c);
    float4 ps = f4zero();
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (act && h0 + j < g.H) {
        float4 y;
        y.x = fmaf((acc[j].x - m1.x) * r1.x, ga.x, be.x);
        y.y = fmaf((acc[j].y - m1.y) * r1.y, ga.y, be.y);
        y.z = fmaf((acc[j].z - m1.z) * r1.z, ga.z, be.z);
        y.w = fmaf((acc[j].w - m1.w) * r1.w, ga.w, be.w);
        const float4 a = f4swish(y);
        sm_st(sm_rsrc(p.a1), poff(j), a);
        ps = f4add(ps, a);
      }
    pool[pl * 2 + q] = ps;
  }
  __syncthreads();
  // image n owns strips [n * HS * W, (n + 1) * HS * W): 16 lanes per (image, quad), fixed order
  {
    const int per = g.HS * g.W;
    const int grp = t >> 4, l16 = t & 15;        // 64 groups of 16 lanes
    if (grp < g.N * 2) {
      const int img = grp >> 1, qq = grp & 1;
      float4 a = f4zero();
      for (int i = l16; i < per; i += 16) a = f4add(a, pool[(img * per + i) * 2 + qq]);
#pragma unroll
      for (int off = 1; off < 16; off <<= 1) {
        a.x += __shfl_xor(a.x, off); a.y += __shfl_xor(a.y, off); a.z += __shfl_xor(a.z, off); a.w += __shfl_xor(a.w, off);
      }
      if (l16 == 0) st4(p.s + (long long)img * g.C + cg * 8 + qq * 4, f4scale(a, 1.0f / (float)(g.H * g.W)));
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward
struct SmallBwd {
  const float* da2;        // [N,H,W,C] gradient w.r.t. a1 * gate (the project conv's backward-data)
  const float* gate;       // [N,C] squeeze-excite gate (nullable)
  const float* chan_add;   // [N,C] (dL/ds) / (H*W) (nullable)
  const float* z1;
  const float *mean1, *rstd1, *gamma1, *beta1;
  const float* w;          // [K,K,C]
  const float* z0;
  const float *mean0, *rstd0, *gamma0, *beta0;
  float *dgamma1, *dbeta1, *dw, *dgamma0, *dbeta0;
  float* dz0;              // [N,H,W,C] out: gradient w.r.t. the expand conv's output
  SmallGeom g;
};

template <int K>
__global__ __launch_bounds__(kSmThreads) void mbconv_dw_bwd_small_k(SmallBwd p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const SmallGeom g = p.g;
  float4* tileA = reinterpret_cast<float4*>(smem);         // [npix][2] a0 = swish(bn0(z0))
  float4* tileD = tileA + (size_t)g.npix * 2;              // [npix][2] dz1
  float4* wl = tileD + (size_t)g.npix * 2;                 // [K*K][2]
  float4* red = wl + K * K * 2;                            // [16][2][2]
  float4* wred = red + 64;                                 // [16 waves][2 quads][K*K] filter-gradient partials
  const int cg = sm_channel_group(g.C);
  if (cg < 0) return;
  const int t = threadIdx.x, q = t & 1, pl = t >> 1, lane = t & 63, wave = t >> 6;
  const int c = cg * 8 + q * 4;
  const bool act = pl < g.nitems;
  int w_ = 0, hs = 0, n = 0;
  if (act) {
    w_ = pl % g.W;
    const int r = pl / g.W;
    hs = r % g.HS;
    n = r / g.HS;
  }
  const int h0 = hs * 4;
  const float inv_n = 1.0f / (float)g.npix;
  // all global loads of the own strip up front: one memory round trip
  const unsigned off0 = (unsigned)((((n * g.H + h0) * g.W + w_) * g.C + c) * 4), rstride = (unsigned)(g.W * g.C * 4);
  auto poff = [&](int j) { return (act && h0 + j < g.H) ? off0 + (unsigned)j * rstride : kSmOob; };
  float4 zv[4], dv[4], z0v[4];
  {
    const __amdgpu_buffer_rsrc_t rz1 = sm_rsrc(p.z1), rd = sm_rsrc(p.da2);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      zv[j] = sm_ld(rz1, poff(j));
      dv[j] = sm_ld(rd, poff(j));
    }
  }
  if (t < K * K * 2) wl[t] = ld4(p.w + (long long)(t >> 1) * g.C + cg * 8 + (t & 1) * 4);
  float4 gt = make_float4(1.f, 1.f, 1.f, 1.f), ca = f4zero();
  if (act && p.gate != nullptr) gt = ld4(p.gate + (long long)n * g.C + c);
  if (act && p.chan_add != nullptr) ca = ld4(p.chan_add + (long long)n * g.C + c);
  // ---- bn1 backward: g = (da2 * gate + chan_add) * swish'(gamma1 * xhat + beta1); sums; dz1 -> LDS
  {
    const float4 m1 = ld4(p.mean1 + c), r1 = ld4(p.rstd1 + c), ga = ld4(p.gamma1 + c), be = ld4(p.beta1 + c);
    float4 s1 = f4zero(), s2 = f4zero();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = act && h0 + j < g.H;
      const float4 xh = make_float4((zv[j].x - m1.x) * r1.x, (zv[j].y - m1.y) * r1.y, (zv[j].z - m1.z) * r1.z, (zv[j].w - m1.w) * r1.w);
      float4 gg = f4fma(dv[j], gt, ca);
      gg = f4mul(gg, f4swish_grad(make_float4(fmaf(xh.x, ga.x, be.x), fmaf(xh.y, ga.y, be.y), fmaf(xh.z, ga.z, be.z), fmaf(xh.w, ga.w, be.w))));
      if (!ok) gg = f4zero();
      zv[j] = xh;      // (reuse: xhat)
      dv[j] = gg;      // (reuse: g)
      s1 = f4add(s1, gg);
      s2 = f4fma(gg, xh, s2);
      __builtin_amdgcn_sched_barrier(0);   // one row at a time: sixteen interleaved swish' evaluations spill
    }
    // z0 of the own strip: in flight across the workgroup reduction below
    {
      const __amdgpu_buffer_rsrc_t rz0 = sm_rsrc(p.z0);
#pragma unroll
      for (int j = 0; j < 4; ++j) z0v[j] = sm_ld(rz0, poff(j));
    }
    sm_block_sum2(s1, s2, red);
    if (pl == 0) {
      st4(p.dbeta1 + c, s1);
      st4(p.dgamma1 + c, s2);
    }
    const float4 a = f4scale(s1, inv_n), b = f4scale(s2, inv_n);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float4 d;
      d.x = ga.x * r1.x * (dv[j].x - a.x - zv[j].x * b.x);
      d.y = ga.y * r1.y * (dv[j].y - a.y - zv[j].y * b.y);
      d.z = ga.z * r1.z * (dv[j].z - a.z - zv[j].z * b.z);
      d.w = ga.w * r1.w * (dv[j].w - a.w - zv[j].w * b.w);
      const bool ok = act && h0 + j < g.H;
      dv[j] = ok ? d : f4zero();   // dz1 of the own strip stays in registers for the filter gradient (zero on rows beyond H)
      if (ok) tileD[((n * g.H + h0 + j) * g.W + w_) * 2 + q] = d;
    }
  }
  // ---- a0 = swish(bn0(z0)) -> LDS; xhat0 kept
  const float4 m0 = ld4(p.mean0 + c), r0 = ld4(p.rstd0 + c), ga0 = ld4(p.gamma0 + c), be0 = ld4(p.beta0 + c);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 xh = make_float4((z0v[j].x - m0.x) * r0.x, (z0v[j].y - m0.y) * r0.y, (z0v[j].z - m0.z) * r0.z, (z0v[j].w - m0.w) * r0.w);
    z0v[j] = xh;
    if (act && h0 + j < g.H) {
      const float4 y = make_float4(fmaf(xh.x, ga0.x, be0.x), fmaf(xh.y, ga0.y, be0.y), fmaf(xh.z, ga0.z, be0.z), fmaf(xh.w, ga0.w, be0.w));
      tileA[((n * g.H + h0 + j) * g.W + w_) * 2 + q] = f4swish(y);
    }
  }
  __syncthreads();
  constexpr int P = (K - 1) / 2;
  // ---- depthwise filter gradient: dw[ky][kx] = sum_pixels a0[h + ky - P][w + kx - P] * dz1[h][w]; one filter row per trip of a rolled
  //      loop (K accumulators live), reduced over the lanes of equal quad parity right away, over the 16 waves through LDS at the end
  {
    const float4* imgA = tileA + (size_t)n * g.H * g.W * 2 + q;
#pragma unroll 1
    for (int ky = 0; ky < K; ++ky) {
      float4 wacc[K];
#pragma unroll
      for (int kx = 0; kx < K; ++kx) wacc[kx] = f4zero();
      if (act) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int hi = h0 + j + ky - P;
          const bool rok = (unsigned)hi < (unsigned)g.H;
#pragma unroll
          for (int kx = 0; kx < K; ++kx) {
            const int wi = w_ - P + kx;
            const float4 v = (rok && (unsigned)wi < (unsigned)g.W) ? imgA[(hi * g.W + wi) * 2] : f4zero();
            wacc[kx] = f4fma(v, dv[j], wacc[kx]);   // (dv = 0 on rows beyond H)
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int kx = 0; kx < K; ++kx) {
        float4 v = wacc[kx];
#pragma unroll
        for (int off = 2; off < 64; off <<= 1) {
          v.x += __shfl_xor(v.x, off); v.y += __shfl_xor(v.y, off); v.z += __shfl_xor(v.z, off); v.w += __shfl_xor(v.w, off);
        }
        if (lane < 2) wred[(wave * 2 + q) * (K * K) + ky * K + kx] = v;
      }
    }
    __syncthreads();
    if (t < 2 * K * K) {
      const int qq = t / (K * K), tap = t - qq * (K * K);
      float4 v = f4zero();
#pragma unroll 4
      for (int wv = 0; wv < kSmThreads / 64; ++wv) v = f4add(v, wred[(wv * 2 + qq) * (K * K) + tap]);
      st4(p.dw + (long long)tap * g.C + cg * 8 + qq * 4, v);
    }
  }
  // ---- depthwise backward-data out of LDS: da0[h][w] = sum dz1[h + P - ky][w + P - kx] * w[ky][kx]  (stride 1, SAME: P = (K-1)/2)
  float4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = f4zero();
  if (act) {
    const float4* imgD = tileD + (size_t)n * g.H * g.W * 2 + q;
#pragma unroll 1
    for (int ky = 0; ky < K; ++ky) {
      float4 wr[K];
#pragma unroll
      for (int kx = 0; kx < K; ++kx) wr[kx] = wl[(ky * K + kx) * 2 + q];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int hi = h0 + j + P - ky;
        const bool rok = (unsigned)hi < (unsigned)g.H;
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const int wi = w_ + P - kx;
          const float4 v = (rok && (unsigned)wi < (unsigned)g.W) ? imgD[(hi * g.W + wi) * 2] : f4zero();
          acc[j] = f4fma(v, wr[kx], acc[j]);
        }
        __builtin_amdgcn_sched_barrier(0);   // one window row in flight at a time (hoisting all 4 costs 60 more VGPRs and spills)
      }
    }
  }
  // ---- bn0 backward: g0 = da0 * swish'(gamma0 * xhat0 + beta0); sums; dz0
  {
    float4 s1 = f4zero(), s2 = f4zero();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const bool ok = act && h0 + j < g.H;
      const float4 xh = z0v[j];
      float4 gg = f4mul(acc[j], f4swish_grad(make_float4(fmaf(xh.x, ga0.x, be0.x), fmaf(xh.y, ga0.y, be0.y), fmaf(xh.z, ga0.z, be0.z),
                                                          fmaf(xh.w, ga0.w, be0.w))));
      if (!ok) gg = f4zero();
      acc[j] = gg;
      s1 = f4add(s1, gg);
      s2 = f4fma(gg, xh, s2);
      __builtin_amdgcn_sched_barrier(0);
    }
    sm_block_sum2(s1, s2, red);
    if (pl == 0) {
      st4(p.dbeta0 + c, s1);
      st4(p.dgamma0 + c, s2);
    }
    const float4 a = f4scale(s1, inv_n), b = f4scale(s2, inv_n);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (act && h0 + j < g.H) {
        float4 d;
        d.x = ga0.x * r0.x * (acc[j].x - a.x - z0v[j].x * b.x);
        d.y = ga0.y * r0.y * (acc[j].y - a.y - z0v[j].y * b.y);
        d.z = ga0.z * r0.z * (acc[j].z - a.z - z0v[j].z * b.z);
        d.w = ga0.w * r0.w * (acc[j].w - a.w - z0v[j].w * b.w);
        sm_st(sm_rsrc(p.dz0), poff(j), d);
      }
  }
}

static bool small_geom(int N, int H, int W, int C, int k, int stride, SmallGeom* g) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 7) || stride != 1 || (k != 3 && k != 5)) return false;
  const long long npix = (long long)N * H * W;
  const int HS = (H + 3) / 4;
  const long long nitems = (long long)N * HS * W;
  if (npix > kSmMaxPix || nitems > kSmLanes || N * 2 > kSmThreads / 16) return false;
  *g = SmallGeom{N, H, W, C, HS, (int)nitems, (int)npix};
  return true;
}
static size_t small_fwd_lds(const SmallGeom& g, int k) {
  return ((size_t)g.npix * 2 + (size_t)k * k * 2 + 64 + kSmLanes * 2) * 16 + 256 * 8 + 16 * 4;
}
static size_t small_bwd_lds(const SmallGeom& g, int k) { return ((size_t)g.npix * 4 + (size_t)k * k * 2 + 64 + (size_t)32 * k * k) * 16; }

// dynamic LDS above the default limit needs the function attribute: raised once per instantiation to the largest eligible request
// (not a stream operation; done before the first launch, i.e. before any HIP-graph capture of the inner step)
template <typename Kern>
static int small_attr(Kern kern, size_t lds_max, int* done) {
  if (*done) return MLIIS_OK;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
  if (e != hipSuccess) return set_error(MLIIS_ERR_LAUNCH, "mbconv_small: cannot reserve %zu bytes of LDS: %s", lds_max, hipGetErrorString(e));
  *done = 1;
  return MLIIS_OK;
}
static SmallGeom small_max_geom() { return SmallGeom{1, 1, 1, 8, 1, kSmLanes, kSmMaxPix}; }

}  // namespace mliis

using namespace mliis;

extern "C" {

int mliis_mbconv_dw_small_supported(int N, int H, int W, int C, int k, int stride) {
  SmallGeom g;
  return small_geom(N, H, W, C, k, stride, &g) ? 1 : 0;
}

int mliis_mbconv_dw_fwd_small(const float* z0, const float* part0, int nblk0, const float* gamma0, const float* beta0, float* mean0,
                              float* rstd0, float* moving_mean0, float* moving_var0, const float* w, const float* gamma1, const float* beta1,
                              float* mean1, float* rstd1, float* moving_mean1, float* moving_var1, float* a0, float* z1, float* a1, float* s,
                              int N, int H, int W, int C, int k, float eps, float momentum, hipStream_t stream) {
  SmallGeom g;
  MLIIS_REQUIRE(small_geom(N, H, W, C, k, 1, &g), MLIIS_ERR_UNSUPPORTED,
                "mbconv_dw_fwd_small: shape N=%d H=%d W=%d C=%d k=%d is not eligible (mliis_mbconv_dw_small_supported)", N, H, W, C, k);
  MLIIS_REQUIRE(z0 && part0 && nblk0 > 0 && gamma0 && beta0 && mean0 && rstd0 && w && gamma1 && beta1 && mean1 && rstd1 && z1 && a1 && s,
                MLIIS_ERR_ARG, "mbconv_dw_fwd_small: null pointer");
  MLIIS_REQUIRE((moving_mean0 == nullptr) == (moving_var0 == nullptr) && (moving_mean1 == nullptr) == (moving_var1 == nullptr), MLIIS_ERR_ARG,
                "mbconv_dw_fwd_small: moving mean / variance come in pairs");
  MLIIS_REQUIRE(aligned16(z0) && aligned16(part0) && aligned16(gamma0) && aligned16(beta0) && aligned16(mean0) && aligned16(rstd0) &&
                    aligned16(w) && aligned16(gamma1) && aligned16(beta1) && aligned16(mean1) && aligned16(rstd1) && aligned16(a0) &&
                    aligned16(z1) && aligned16(a1) && aligned16(s) && aligned16(moving_mean1) && aligned16(moving_var1),
                MLIIS_ERR_ALIGN, "mbconv_dw_fwd_small: pointers must be 16-byte aligned");
  SmallFwd p{z0, part0, nblk0, gamma0, beta0, mean0, rstd0, moving_mean0, moving_var0, w, gamma1, beta1, mean1, rstd1, moving_mean1,
             moving_var1, a0, z1, a1, s, g, eps, 1.0f - momentum};
  const size_t lds = small_fwd_lds(g, k);
  static int attr3 = 0, attr5 = 0;
  int rc;
  if (k == 3) {
    if ((rc = small_attr(mbconv_dw_fwd_small_k<3>, small_fwd_lds(small_max_geom(), 3), &attr3))) return rc;
    hipLaunchKernelGGL(mbconv_dw_fwd_small_k<3>, dim3(sm_grid(C)), dim3(kSmThreads), lds, stream, p);
  } else {
    if ((rc = small_attr(mbconv_dw_fwd_small_k<5>, small_fwd_lds(small_max_geom(), 5), &attr5))) return rc;
    hipLaunchKernelGGL(mbconv_dw_fwd_small_k<5>, dim3(sm_grid(C)), dim3(kSmThreads), lds, stream, p);
  }
  MLIIS_CHECK_LAUNCH("mbconv_dw_fwd_small");
  return MLIIS_OK;
}

int mliis_mbconv_dw_bwd_small(const float* da2, const float* gate, const float* chan_add, const float* z1, const float* mean1,
                              const float* rstd1, const float* gamma1, const float* beta1, const float* w, const float* z0, const float* mean0,
                              const float* rstd0, const float* gamma0, const float* beta0, float* dgamma1, float* dbeta1, float* dw,
                              float* dgamma0, float* dbeta0, float* dz0, int N, int H, int W, int C, int k, hipStream_t stream) {
  SmallGeom g;
  MLIIS_REQUIRE(small_geom(N, H, W, C, k, 1, &g), MLIIS_ERR_UNSUPPORTED,
                "mbconv_dw_bwd_small: shape N=%d H=%d W=%d C=%d k=%d is not eligible (mliis_mbconv_dw_small_supported)", N, H, W, C, k);
  MLIIS_REQUIRE(da2 && z1 && mean1 && rstd1 && gamma1 && beta1 && w && z0 && mean0 && rstd0 && gamma0 && beta0 && dgamma1 && dbeta1 && dw &&
                    dgamma0 && dbeta0 && dz0,
                MLIIS_ERR_ARG, "mbconv_dw_bwd_small: null pointer");
  MLIIS_REQUIRE(aligned16(da2) && aligned16(gate) && aligned16(chan_add) && aligned16(z1) && aligned16(mean1) && aligned16(rstd1) &&
                    aligned16(gamma1) && aligned16(beta1) && aligned16(w) && aligned16(z0) && aligned16(mean0) && aligned16(rstd0) &&
                    aligned16(gamma0) && aligned16(beta0) && aligned16(dgamma1) && aligned16(dbeta1) && aligned16(dw) && aligned16(dgamma0) &&
                    aligned16(dbeta0) && aligned16(dz0),
                MLIIS_ERR_ALIGN, "mbconv_dw_bwd_small: pointers must be 16-byte aligned");
  SmallBwd p{da2, gate, chan_add, z1, mean1, rstd1, gamma1, beta1, w, z0, mean0, rstd0, gamma0, beta0, dgamma1, dbeta1, dw, dgamma0, dbeta0,
             dz0, g};
  const size_t lds = small_bwd_lds(g, k);
  static int attr3 = 0, attr5 = 0;
  int rc;
  if (k == 3) {
    if ((rc = small_attr(mbconv_dw_bwd_small_k<3>, small_bwd_lds(small_max_geom(), 3), &attr3))) return rc;
    hipLaunchKernelGGL(mbconv_dw_bwd_small_k<3>, dim3(sm_grid(C)), dim3(kSmThreads), lds, stream, p);
  } else {
    if ((rc = small_attr(mbconv_dw_bwd_small_k<5>, small_bwd_lds(small_max_geom(), 5), &attr5))) return rc;
    hipLaunchKernelGGL(mbconv_dw_bwd_small_k<5>, dim3(sm_grid(C)), dim3(kSmThreads), lds, stream, p);
  }
  MLIIS_CHECK_LAUNCH("mbconv_dw_bwd_small");
  return MLIIS_OK;
}
}
