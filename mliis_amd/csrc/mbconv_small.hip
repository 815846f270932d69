// The depthwise half of an MBConv block on SMALL maps (14x14 at 224x224 inputs: blocks 6-10 of EfficientLab-6-3), one launch per
// direction.  Reference ops: models/efficientnet/efficientnet_model.py:183-200,266-271 (expand BN -> swish -> depthwise k x k -> BN ->
// swish -> squeeze-excite mean) and models/efficientnet/utils.py:87-134 (training-mode batch norm + moving averages).
//
// Why: on a [8,14,14,480..672] tensor (3-4 MB) every launch of the op-by-op chain  bn0 apply -> depthwise(+statistics) -> bn1
// apply(+pooling)  and, backward,  bn1 reduce -> bn1 apply -> depthwise filter gradient -> depthwise backward-data(+bn0 sums) -> bn0
// apply  costs 5-10 us of dependent memory round trips for < 1 us of traffic (profiles/r01_final_profile.md).  Everything in those
// chains is PER CHANNEL: batch-norm statistics and gradient sums run over (N, H, W) of one channel, the depthwise stencil stays inside
// a channel.  So a workgroup that owns one channel QUAD for the whole [N, H, W] extent needs no grid-wide dependency at all:
//   forward : fold the expand conv's stage-1 statistics -> a0 = swish(bn0(z0)) into LDS -> stencil out of LDS -> exact two-pass statistics
//             of z1 in the workgroup -> a1 = swish(bn1(z1)) -> per-image means for the squeeze-excite -> both moving averages;
//   backward: bn1 backward (both sums + apply) -> dz1 tile in LDS -> depthwise filter gradient (complete, no slabs) and backward-data out
//             of LDS -> bn0 backward (both sums + apply) -> dz0.
// Layout: 512 threads = 512 pixel lanes of ONE channel quad (120-168 workgroups for C = 480-672; 256 VGPRs per thread: the K x K taps,
// the K x K filter-gradient accumulators and a K-wide window row all live in registers); a lane owns a vertical strip of 4 pixels and
// lanes run along W, so the LDS tiles [pixel] (float4) are read as contiguous, conflict-free 16-byte words and every window row is read
// once for its up to four (output row, filter row) pairs.  Wave-level sums are DPP adds (VALU), not LDS permutes.  Eligible: stride 1,
// N*H*W <= 2048 pixels, N*ceil(H/4)*W <= 512 strips, C % 4 == 0 -- anything else takes the op-by-op path.
// Workgroup -> channel-quad mapping keeps the eight quads of a 128-byte line (32 channels) on one XCD (speed only).
// (First version -- 8 channels x 1024 threads, 128-VGPR budget, filter rows in a rolled loop, ds_bpermute reductions -- measured 24 us
// forward / 67 us backward per 14x14x672 layer against 17 / 44 us op by op: spills and ~880 LDS instructions per wave.)
#include "common.hpp"

namespace mliis {

constexpr int kSmThreads = 512;    // pixel lanes (strips) per workgroup
constexpr int kSmWaves = kSmThreads / 64;
constexpr int kSmMaxPix = 2048;    // N*H*W: two [N*H*W] float4 tiles (64 KB) + scratch in LDS

struct SmallGeom {
  int N, H, W, C, HS, nitems, npix;
};

// channel quad of this workgroup (or -1): the eight quads of one 32-channel line share an XCD under round-robin dispatch
__device__ __forceinline__ int sm_channel_quad(int C) {
  const int lines = (C + 31) >> 5;
  const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
  const int line = (j >> 3) * 8 + xcd, sub = j & 7;
  if (line >= lines) return -1;
  const int cq = line * 8 + sub;
  return cq * 4 < C ? cq : -1;
}
static inline int sm_grid(int C) { return ((((C + 31) / 32) + 7) / 8) * 8 * 8; }

// ---- wave sum by DPP adds (VALU): quad xor 1, xor 2, half-row mirror, row mirror -> every lane holds its 16-lane row sum; row_bcast15
//      adds row 0 into row 1 and row 2 into row 3, row_bcast31 adds rows 0+1 into rows 2, 3: lane 63 ends with the wave total.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float sm_dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, true));
}
__device__ __forceinline__ float sm_wave_sum63(float v) {
  v = sm_dpp_add<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
  v = sm_dpp_add<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
  v = sm_dpp_add<0x141, 0xF>(v);   // row_half_mirror
  v = sm_dpp_add<0x140, 0xF>(v);   // row_mirror
  v = sm_dpp_add<0x142, 0xA>(v);   // row_bcast15 -> rows 1, 3
  v = sm_dpp_add<0x143, 0xC>(v);   // row_bcast31 -> rows 2, 3
  return v;                        // valid in lane 63
}
__device__ __forceinline__ float4 sm_wave_sum63(float4 v) {
  return make_float4(sm_wave_sum63(v.x), sm_wave_sum63(v.y), sm_wave_sum63(v.z), sm_wave_sum63(v.w));
}

// sum of a float4 pair over the workgroup (fixed order); every thread gets the totals.  red: LDS float4 [kSmWaves][2]; two barriers.
__device__ __forceinline__ void sm_block_sum2(float4& a, float4& b, float4* red) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float4 wa = sm_wave_sum63(a), wb = sm_wave_sum63(b);
  if (lane == 63) {
    red[wave * 2 + 0] = wa;
    red[wave * 2 + 1] = wb;
  }
  __syncthreads();
  float4 sa = red[0], sb = red[1];
#pragma unroll
  for (int w = 1; w < kSmWaves; ++w) {
    sa = f4add(sa, red[w * 2 + 0]);
    sb = f4add(sb, red[w * 2 + 1]);
  }
  a = sa;
  b = sb;
  __syncthreads();
}

// raw buffer access with 32-bit byte offsets (tensors < 2 GiB, checked on the host): an out-of-range offset reads zeros / drops the store
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
// xhat = (x - mean) * rstd;  y = gamma * xhat + beta
__device__ __forceinline__ float4 f4xhat(float4 x, float4 m, float4 r) {
  return make_float4((x.x - m.x) * r.x, (x.y - m.y) * r.y, (x.z - m.z) * r.z, (x.w - m.w) * r.w);
}
__device__ __forceinline__ float4 f4affine(float4 xh, float4 g, float4 b) {
  return make_float4(fmaf(xh.x, g.x, b.x), fmaf(xh.y, g.y, b.y), fmaf(xh.z, g.z, b.z), fmaf(xh.w, g.w, b.w));
}
// gamma * rstd * (g - a - xhat * b)
__device__ __forceinline__ float4 f4bn_dx(float4 g, float4 xh, float4 a, float4 b, float4 ga, float4 rs) {
  return make_float4(ga.x * rs.x * (g.x - a.x - xh.x * b.x), ga.y * rs.y * (g.y - a.y - xh.y * b.y), ga.z * rs.z * (g.z - a.z - xh.z * b.z),
                     ga.w * rs.w * (g.w - a.w - xh.w * b.w));
}

// strip decode shared by both kernels
struct SmStrip {
  bool act;
  int n, h0, w;
  unsigned off0, rstride;   // byte offset of the strip's first pixel / of one image row, in any [N,H,W,C] tensor
  int H;
  __device__ __forceinline__ unsigned poff(int j) const { return (act && h0 + j < H) ? off0 + (unsigned)j * rstride : kSmOob; }
  __device__ __forceinline__ bool ok(int j) const { return act && h0 + j < H; }
};
__device__ __forceinline__ SmStrip sm_strip(const SmallGeom& g, int c) {
  SmStrip s;
  const int pl = threadIdx.x;
  s.act = pl < g.nitems;
  int w_ = 0, hs = 0, n = 0;
  if (s.act) {
    w_ = pl % g.W;
    const int r = pl / g.W;
    hs = r % g.HS;
    n = r / g.HS;
  }
  s.n = n;
  s.h0 = hs * 4;
  s.w = w_;
  s.H = g.H;
  s.off0 = (unsigned)((((n * g.H + s.h0) * g.W + w_) * g.C + c) * 4);
  s.rstride = (unsigned)(g.W * g.C * 4);
  return s;
}

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
  float4* tileA = reinterpret_cast<float4*>(smem);                       // [npix] a0
  float4* wl = tileA + g.npix;                                           // [K*K] filter taps
  float4* red = wl + K * K;                                              // [kSmWaves][2]
  float4* pool = red + kSmWaves * 2;                                     // [512] per-strip sums of a1
  double* fold = reinterpret_cast<double*>(pool + kSmThreads);           // [8 columns]
  float* stat = reinterpret_cast<float*>(fold + 64);                     // mean0[4] rstd0[4] var0[4]
  const int cq = sm_channel_quad(g.C);
  if (cq < 0) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int c = cq * 4;
  const SmStrip st = sm_strip(g, c);
  // own pixels' z0 first (does not depend on the statistics fold)
  const __amdgpu_buffer_rsrc_t rz0 = sm_rsrc(p.z0);
  float4 zin[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) zin[j] = sm_ld(rz0, st.poff(j));
  if (t < K * K) wl[t] = ld4(p.w + (long long)t * g.C + c);
  // ---- fold the expand conv's stage-1 statistics of this quad (double precision, fixed order): wave w owns column w of the 8
  //      {sum, sum of squares} x 4 channels; its lanes take the partial blocks (ONE memory round trip for up to 64 of them)
  {
    const int v = wave >> 2, ch = wave & 3;
    double acc = 0.0;
    for (int b = lane; b < p.nblk0; b += 64) acc += (double)p.part0[((long long)b * 2 + v) * g.C + c + ch];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) fold[wave] = acc;
  }
  // the moving averages are read now (their latency hides behind everything below) and written at the very end
  float mmv0 = 0.f, mvv0 = 0.f;
  if (t < 4 && p.mm0 != nullptr) {
    mmv0 = p.mm0[c + t];
    mvv0 = p.mv0[c + t];
  }
  float4 mmv1 = f4zero(), mvv1 = f4zero();
  if (t == 0 && p.mm1 != nullptr) {
    mmv1 = ld4(p.mm1 + c);
    mvv1 = ld4(p.mv1 + c);
  }
  __syncthreads();
  if (t < 4) {
    const double s1 = fold[t], s2 = fold[4 + t];
    const double inv_n = 1.0 / (double)g.npix;
    const double m = s1 * inv_n;
    double var = s2 * inv_n - m * m;
    if (var < 0.0) var = 0.0;
    stat[t] = (float)m;
    stat[4 + t] = (float)(1.0 / sqrt(var + (double)p.eps));
    stat[8 + t] = (float)var;
  }
  __syncthreads();
  const float4 m0 = ld4(stat), r0 = ld4(stat + 4);
  {
    const float4 ga = ld4(p.gamma0 + c), be = ld4(p.beta0 + c);
    const __amdgpu_buffer_rsrc_t ra0 = sm_rsrc(p.a0 != nullptr ? p.a0 : p.z0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (st.ok(j)) {
        const float4 a = f4swish(f4affine(f4xhat(zin[j], m0, r0), ga, be));
        tileA[(st.n * g.H + st.h0 + j) * g.W + st.w] = a;
        if (p.a0 != nullptr) sm_st(ra0, st.poff(j), a);
      }
    }
  }
  __syncthreads();
  // ---- depthwise stencil out of LDS: outputs (h0 + j, w), j < 4; TF-SAME, stride 1.  Window row r (image row h0 - P + r) is read once
  //      and feeds the (output row j, filter row ky = r - j) pairs; all K*K taps in registers.
  constexpr int P = (K - 1) / 2;
  float4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = f4zero();
  if (st.act) {
    const float4* imgA = tileA + st.n * g.H * g.W;
    if constexpr (K == 3) {   // all taps and the window rows in registers, fully unrolled (134 VGPRs, no spills)
      float4 wreg[K * K];
#pragma unroll
      for (int i = 0; i < K * K; ++i) wreg[i] = wl[i];
#pragma unroll
      for (int r = 0; r < K + 3; ++r) {
        const int hi = st.h0 - P + r;
        const bool rok = (unsigned)hi < (unsigned)g.H;
        float4 row[K];
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const int wi = st.w - P + kx;
          row[kx] = (rok && (unsigned)wi < (unsigned)g.W) ? imgA[hi * g.W + wi] : f4zero();
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int ky = r - j;
          if (ky >= 0 && ky < K) {
#pragma unroll
            for (int kx = 0; kx < K; ++kx) acc[j] = f4fma(row[kx], wreg[ky * K + kx], acc[j]);
          }
        }
      }
    } else {
      // 5x5: a ROLLED loop over the window rows; the taps of the (output row j, filter row r - j) pairs are read from LDS where they
      // are used (one address for the whole wave: a broadcast).  Unrolled, the compiler hoists the 25 taps (100 VGPRs) and the window
      // rows on top of each other and spills 65-320 VGPRs -- and every spill is a memory round trip that the next barrier waits for.
#pragma unroll 1
      for (int r = 0; r < K + 3; ++r) {
        const int hi = st.h0 - P + r;
        const bool rok = (unsigned)hi < (unsigned)g.H;
        float4 row[K];
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const int wi = st.w - P + kx;
          row[kx] = (rok && (unsigned)wi < (unsigned)g.W) ? imgA[hi * g.W + wi] : f4zero();
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int ky = r - j;
          if (ky >= 0 && ky < K) {
#pragma unroll
            for (int kx = 0; kx < K; ++kx) acc[j] = f4fma(row[kx], wl[ky * K + kx], acc[j]);
          }
        }
      }
    }
  }
  // ---- statistics of z1: exact two-pass in the workgroup (mean, then centred second moment).  z1 / a1 are STORED at the very end: a
  //      barrier waits for the wave's outstanding stores too (vmcnt), and there are five barriers between here and there.
  float4 s1 = f4zero(), s2 = f4zero();
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (st.ok(j)) s1 = f4add(s1, acc[j]);
  sm_block_sum2(s1, s2, red);
  const float inv_n = 1.0f / (float)g.npix;
  const float4 m1 = f4scale(s1, inv_n);
  float4 d2 = f4zero(), dummy = f4zero();
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (st.ok(j)) {
      const float4 d = f4sub(acc[j], m1);
      d2 = f4fma(d, d, d2);
    }
  sm_block_sum2(d2, dummy, red);
  const float4 var1 = f4scale(d2, inv_n);
  const float4 r1 = make_float4(1.0f / sqrtf(var1.x + p.eps), 1.0f / sqrtf(var1.y + p.eps), 1.0f / sqrtf(var1.z + p.eps),
                                1.0f / sqrtf(var1.w + p.eps));
  // ---- a1 = swish(bn1(z1)), per-image pooled mean
  float4 a1v[4];
  {
    const float4 ga = ld4(p.gamma1 + c), be = ld4(p.beta1 + c);
    float4 ps = f4zero();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a1v[j] = f4swish(f4affine(f4xhat(acc[j], m1, r1), ga, be));
      if (st.ok(j)) ps = f4add(ps, a1v[j]);
    }
    pool[t] = ps;
  }
  __syncthreads();
  // image n owns strips [n * HS * W, (n + 1) * HS * W): one wave per image (fixed order), DPP wave sum
  {
    const int per = g.HS * g.W;
    for (int img = wave; img < g.N; img += kSmWaves) {
      float4 a = f4zero();
      for (int i = lane; i < per; i += 64) a = f4add(a, pool[img * per + i]);
      a = sm_wave_sum63(a);
      if (lane == 63) st4(p.s + (long long)img * g.C + c, f4scale(a, 1.0f / (float)(g.H * g.W)));
    }
  }
  {
    const __amdgpu_buffer_rsrc_t rz1 = sm_rsrc(p.z1), ra1 = sm_rsrc(p.a1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sm_st(rz1, st.poff(j), acc[j]);    // (rows beyond H / idle lanes: out-of-range offset, the store is dropped)
      sm_st(ra1, st.poff(j), a1v[j]);
    }
  }
  // ---- batch statistics for the backward pass, moving averages (biased variance: non-fused TpuBatchNormalization, utils.py:87-134)
  if (t < 4) {
    const float mf = stat[t], vf = stat[8 + t];
    p.mean0[c + t] = mf;
    p.rstd0[c + t] = stat[4 + t];
    if (p.mm0 != nullptr) {
      p.mm0[c + t] = mmv0 - (mmv0 - mf) * p.one_minus_momentum;
      p.mv0[c + t] = mvv0 - (mvv0 - vf) * p.one_minus_momentum;
    }
  }
  if (t == 0) {
    st4(p.mean1 + c, m1);
    st4(p.rstd1 + c, r1);
    if (p.mm1 != nullptr) {
      const float om = p.one_minus_momentum;
      st4(p.mm1 + c, make_float4(mmv1.x - (mmv1.x - m1.x) * om, mmv1.y - (mmv1.y - m1.y) * om, mmv1.z - (mmv1.z - m1.z) * om,
                                 mmv1.w - (mmv1.w - m1.w) * om));
      st4(p.mv1 + c, make_float4(mvv1.x - (mvv1.x - var1.x) * om, mvv1.y - (mvv1.y - var1.y) * om, mvv1.z - (mvv1.z - var1.z) * om,
                                 mvv1.w - (mvv1.w - var1.w) * om));
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
  float4* tileA = reinterpret_cast<float4*>(smem);         // [npix] a0 = swish(bn0(z0))
  float4* tileD = tileA + g.npix;                          // [npix] dz1
  float4* wl = tileD + g.npix;                             // [K*K]
  float4* red = wl + K * K;                                // [kSmWaves][2]
  float4* wred = red + kSmWaves * 2;                       // [kSmWaves][K*K] filter-gradient partials
  const int cq = sm_channel_quad(g.C);
  if (cq < 0) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int c = cq * 4;
  const SmStrip st = sm_strip(g, c);
  const float inv_n = 1.0f / (float)g.npix;
  // all global loads of the own strip up front: one memory round trip
  float4 zv[4], dv[4], z0v[4];
  {
    const __amdgpu_buffer_rsrc_t rz1 = sm_rsrc(p.z1), rd = sm_rsrc(p.da2), rz0 = sm_rsrc(p.z0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      zv[j] = sm_ld(rz1, st.poff(j));
      dv[j] = sm_ld(rd, st.poff(j));
      z0v[j] = sm_ld(rz0, st.poff(j));
    }
  }
  if (t < K * K) wl[t] = ld4(p.w + (long long)t * g.C + c);
  float4 gt = make_float4(1.f, 1.f, 1.f, 1.f), ca = f4zero();
  if (st.act && p.gate != nullptr) gt = ld4(p.gate + (long long)st.n * g.C + c);
  if (st.act && p.chan_add != nullptr) ca = ld4(p.chan_add + (long long)st.n * g.C + c);
  const float4 m0 = ld4(p.mean0 + c), r0 = ld4(p.rstd0 + c), ga0 = ld4(p.gamma0 + c), be0 = ld4(p.beta0 + c);
  // ---- bn1 backward: g = (da2 * gate + chan_add) * swish'(gamma1 * xhat + beta1); sums; dz1 -> LDS
  {
    const float4 m1 = ld4(p.mean1 + c), r1 = ld4(p.rstd1 + c), ga = ld4(p.gamma1 + c), be = ld4(p.beta1 + c);
    float4 s1 = f4zero(), s2 = f4zero();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 xh = f4xhat(zv[j], m1, r1);
      float4 gg = f4mul(f4fma(dv[j], gt, ca), f4swish_grad(f4affine(xh, ga, be)));
      if (!st.ok(j)) gg = f4zero();
      zv[j] = xh;      // (reuse: xhat)
      dv[j] = gg;      // (reuse: g)
      s1 = f4add(s1, gg);
      s2 = f4fma(gg, xh, s2);
    }
    sm_block_sum2(s1, s2, red);
    if (t == 0) {
      st4(p.dbeta1 + c, s1);
      st4(p.dgamma1 + c, s2);
    }
    const float4 a = f4scale(s1, inv_n), b = f4scale(s2, inv_n);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 d = f4bn_dx(dv[j], zv[j], a, b, ga, r1);
      dv[j] = st.ok(j) ? d : f4zero();   // dz1 of the own strip stays in registers for the filter gradient (zero on rows beyond H)
      if (st.ok(j)) tileD[(st.n * g.H + st.h0 + j) * g.W + st.w] = d;
    }
  }
  // ---- a0 = swish(bn0(z0)) -> LDS; xhat0 kept
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 xh = f4xhat(z0v[j], m0, r0);
    z0v[j] = xh;
    if (st.ok(j)) tileA[(st.n * g.H + st.h0 + j) * g.W + st.w] = f4swish(f4affine(xh, ga0, be0));
  }
  __syncthreads();
  constexpr int P = (K - 1) / 2;
  // ---- depthwise filter gradient: dw[ky][kx] = sum_pixels a0[h + ky - P][w + kx - P] * dz1[h][w]; window row r (image row h0 - P + r)
  //      read once for its (j, ky = r - j) pairs; K*K accumulators, DPP wave sums, the 8 waves through LDS
  {
    float4 wacc[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) wacc[i] = f4zero();
    if (st.act) {
      const float4* imgA = tileA + st.n * g.H * g.W;
#pragma unroll
      for (int r = 0; r < K + 3; ++r) {
        const int hi = st.h0 - P + r;
        const bool rok = (unsigned)hi < (unsigned)g.H;
        float4 row[K];
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const int wi = st.w - P + kx;
          row[kx] = (rok && (unsigned)wi < (unsigned)g.W) ? imgA[hi * g.W + wi] : f4zero();
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int ky = r - j;
          if (ky >= 0 && ky < K) {
#pragma unroll
            for (int kx = 0; kx < K; ++kx) wacc[ky * K + kx] = f4fma(row[kx], dv[j], wacc[ky * K + kx]);
          }
        }
        if (K == 5) __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int i = 0; i < K * K; ++i) {
      const float4 v = sm_wave_sum63(wacc[i]);
      if (lane == 63) wred[wave * (K * K) + i] = v;
    }
  }
  // ---- depthwise backward-data out of LDS: da0[h][w] = sum dz1[h + P - ky][w + P - kx] * w[ky][kx]  (stride 1, SAME: P = (K-1)/2)
  float4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = f4zero();
  if (st.act) {
    const float4* imgD = tileD + st.n * g.H * g.W;
    // window row r = image row h0 - P + r; pairs with (j, ky): r - j = (K - 1) - ky.  3x3: unrolled, taps in registers; 5x5: rolled, taps
    // from LDS where they are used (see the forward kernel)
    float4 wreg[K == 3 ? K * K : 1];
    if constexpr (K == 3) {
#pragma unroll
      for (int i = 0; i < K * K; ++i) wreg[i] = wl[i];
    }
    auto body = [&](int r) {
      const int hi = st.h0 - P + r;
      const bool rok = (unsigned)hi < (unsigned)g.H;
      float4 row[K];
#pragma unroll
      for (int x = 0; x < K; ++x) {
        const int wi = st.w - P + x;
        row[x] = (rok && (unsigned)wi < (unsigned)g.W) ? imgD[hi * g.W + wi] : f4zero();
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ky = (K - 1) - (r - j);
        if (ky >= 0 && ky < K) {
#pragma unroll
          for (int x = 0; x < K; ++x) {
            const float4 wv = K == 3 ? wreg[K == 3 ? ky * K + (K - 1 - x) : 0] : wl[ky * K + (K - 1 - x)];
            acc[j] = f4fma(row[x], wv, acc[j]);
          }
        }
      }
    };
    if constexpr (K == 3) {
#pragma unroll
      for (int r = 0; r < K + 3; ++r) body(r);
    } else {
#pragma unroll 1
      for (int r = 0; r < K + 3; ++r) body(r);
    }
  }
  __syncthreads();   // (the filter-gradient partials of all waves are in LDS)
  if (t < K * K) {
    float4 v = wred[t];
#pragma unroll
    for (int wv = 1; wv < kSmWaves; ++wv) v = f4add(v, wred[wv * (K * K) + t]);
    st4(p.dw + (long long)t * g.C + c, v);
  }
  // ---- bn0 backward: g0 = da0 * swish'(gamma0 * xhat0 + beta0); sums; dz0
  {
    float4 s1 = f4zero(), s2 = f4zero();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float4 gg = f4mul(acc[j], f4swish_grad(f4affine(z0v[j], ga0, be0)));
      if (!st.ok(j)) gg = f4zero();
      acc[j] = gg;
      s1 = f4add(s1, gg);
      s2 = f4fma(gg, z0v[j], s2);
    }
    sm_block_sum2(s1, s2, red);
    if (t == 0) {
      st4(p.dbeta0 + c, s1);
      st4(p.dgamma0 + c, s2);
    }
    const float4 a = f4scale(s1, inv_n), b = f4scale(s2, inv_n);
    const __amdgpu_buffer_rsrc_t rdz = sm_rsrc(p.dz0);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (st.ok(j)) sm_st(rdz, st.poff(j), f4bn_dx(acc[j], z0v[j], a, b, ga0, r0));
  }
}

static bool small_geom(int N, int H, int W, int C, int k, int stride, SmallGeom* g) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || stride != 1 || (k != 3 && k != 5)) return false;
  const long long npix = (long long)N * H * W;
  const int HS = (H + 3) / 4;
  const long long nitems = (long long)N * HS * W;
  if (npix > kSmMaxPix || nitems > kSmThreads || npix * C * 4 >= (1LL << 31)) return false;
  *g = SmallGeom{N, H, W, C, HS, (int)nitems, (int)npix};
  return true;
}
static size_t small_fwd_lds(const SmallGeom& g, int k) {
  return ((size_t)g.npix + (size_t)k * k + kSmWaves * 2 + kSmThreads) * 16 + 64 * 8 + 16 * 4;
}
static size_t small_bwd_lds(const SmallGeom& g, int k) { return ((size_t)g.npix * 2 + (size_t)k * k + kSmWaves * 2 + (size_t)kSmWaves * k * k) * 16; }

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
static SmallGeom small_max_geom() { return SmallGeom{1, 1, 1, 4, 1, kSmThreads, kSmMaxPix}; }

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
