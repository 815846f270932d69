// The depthwise half of an MBConv block on SMALL maps (14x14 at 224x224 inputs: blocks 6-10 of EfficientLab-6-3), one launch per
// direction.  Reference ops: models/efficientnet/efficientnet_model.py:183-200,266-271 (expand BN -> swish -> depthwise k x k -> BN ->
// swish -> squeeze-excite mean) and models/efficientnet/utils.py:87-134 (training-mode batch norm + moving averages).
//
// Why: on a [8,14,14,480..672] tensor (3-4 MB) every launch of the op-by-op chain  bn0 apply -> depthwise(+statistics) -> bn1
// apply(+pooling)  and, backward,  bn1 reduce -> bn1 apply -> depthwise filter gradient -> depthwise backward-data(+bn0 sums) -> bn0
// apply  costs 5-10 us of dependent memory round trips for < 1 us of traffic (profiles/r01_final_profile.md).  Everything in those
// chains is PER CHANNEL: batch-norm statistics and gradient sums run over (N, H, W) of one channel, the depthwise stencil stays inside
// a channel.  So a workgroup that owns a GROUP OF V CHANNELS for the whole [N, H, W] extent needs no grid-wide dependency at all:
//   forward : fold the expand conv's stage-1 statistics -> a0 = swish(bn0(z0)) into LDS -> stencil out of LDS -> exact two-pass statistics
//             of z1 in the workgroup -> a1 = swish(bn1(z1)) -> per-image means for the squeeze-excite -> both moving averages;
//   backward: bn1 backward (both sums + apply) -> dz1 tile in LDS -> depthwise filter gradient (complete, no slabs) and backward-data out
//             of LDS -> bn0 backward (both sums + apply) -> dz0.
// Layout: 512 threads = 512 pixel lanes of ONE channel group; a lane owns a vertical strip of 4 pixels and lanes run along W, so the
// LDS tiles [pixel] are read as contiguous, conflict-free words and every window row is read once for its up to four (output row,
// filter row) pairs.  Wave-level sums are DPP adds (VALU), not LDS permutes.  Eligible: stride 1, N*H*W <= 2048 pixels,
// N*ceil(H/4)*W <= 512 strips, C % 4 == 0 -- anything else takes the op-by-op path.
//
// Round 4 -- V channels per workgroup, V = 4 (quads) or 2 (pairs), chosen per layer (sm_group_width), and a stamps build (-DSM_DBG,
// profiles/r04_notes.md) that showed where the 10-23 us of a launch go.  Rounds 2-3 ran quads only: 120 / 168 workgroups on 256 CUs.
// Filling the chip with 240 / 336 workgroups of pairs does NOT shorten the memory phases -- a workgroup touches the same 1568 lines
// per tensor whatever V is, 8-byte accesses run at 0.55-0.7 of the 16-byte rate, and channel triples (12-byte accesses) were 1.3-1.8x
// slower -- so pairs are used only where the compute phases dominate: the 5x5 layers whose C / 2 workgroups fit one round (25 taps:
// with V = 2 the taps and the filter-gradient accumulators fit the register file and the 600 DPP adds of the accumulators' wave
// sums halve): 14x14x480 5x5 13.1 -> 10.4 us forward, 22.3 -> 17.0 us backward.  For every form: the workgroup sums fold their eight
// partial vectors in one wave instead of in every thread, barriers that follow a global store order LDS traffic only (the outputs
// drain under the statistics instead of at the end of the kernel), and the 5x5 quads backward parks xhat0 in a third LDS tile instead
// of spilling 26 registers.
// Workgroup -> channel-group mapping keeps the groups of one 128-byte line (32 channels) on one XCD (speed only).
// (First version -- 8 channels x 1024 threads, 128-VGPR budget, filter rows in a rolled loop, ds_bpermute reductions -- measured 24 us
// forward / 67 us backward per 14x14x672 layer against 17 / 44 us op by op: spills and ~880 LDS instructions per wave.)
#include "common.hpp"

namespace mliis {

constexpr int kSmThreads = 512;    // pixel lanes (strips) per workgroup
constexpr int kSmWaves = kSmThreads / 64;
constexpr int kSmMaxPix = 2048;    // N*H*W: two [N*H*W] tiles of <= 16-byte slots (64 KB) + scratch in LDS

struct SmallGeom {
  int N, H, W, C, HS, nitems, npix;
};

// -DSM_DBG (diagnosis builds only, tools/_exp): wall-clock stamps (100 MHz) of thread 0 at the phase boundaries of both kernels
#ifdef SM_DBG
#define SM_STAMP_DECL unsigned long long stamp[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define SM_STAMP(k) do { stamp[k] = wall_clock64(); } while (0)
#define SM_STAMP_FLUSH(ptr)                                                              \
  do {                                                                                   \
    __builtin_amdgcn_s_waitcnt(0);                                                       \
    stamp[11] = wall_clock64();                                                          \
    if (threadIdx.x == 0 && (ptr) != nullptr)                                            \
      for (int k = 0; k < 12; ++k) (ptr)[(long long)blockIdx.x * 12 + k] = stamp[k];     \
  } while (0)
#else
#define SM_STAMP_DECL
#define SM_STAMP(k) do { } while (0)
#define SM_STAMP_FLUSH(ptr) do { } while (0)
#endif

// ---- V channels of one pixel: plain float arrays, every operation an unrolled loop (V = 2, 3, 4)
template <int V>
struct SmV {
  float v[V];
};
#define SMV_FOR _Pragma("unroll") for (int i = 0; i < V; ++i)
template <int V> __device__ __forceinline__ SmV<V> vzero() { SmV<V> r; SMV_FOR r.v[i] = 0.f; return r; }
template <int V> __device__ __forceinline__ SmV<V> vfill(float x) { SmV<V> r; SMV_FOR r.v[i] = x; return r; }
template <int V> __device__ __forceinline__ SmV<V> vadd(SmV<V> a, SmV<V> b) { SMV_FOR a.v[i] += b.v[i]; return a; }
template <int V> __device__ __forceinline__ SmV<V> vsub(SmV<V> a, SmV<V> b) { SMV_FOR a.v[i] -= b.v[i]; return a; }
template <int V> __device__ __forceinline__ SmV<V> vmul(SmV<V> a, SmV<V> b) { SMV_FOR a.v[i] *= b.v[i]; return a; }
template <int V> __device__ __forceinline__ SmV<V> vscale(SmV<V> a, float s) { SMV_FOR a.v[i] *= s; return a; }
template <int V> __device__ __forceinline__ SmV<V> vfma(SmV<V> a, SmV<V> b, SmV<V> c) { SMV_FOR c.v[i] = fmaf(a.v[i], b.v[i], c.v[i]); return c; }
template <int V> __device__ __forceinline__ SmV<V> vswish(SmV<V> a) { SMV_FOR a.v[i] = swish_f(a.v[i]); return a; }
template <int V> __device__ __forceinline__ SmV<V> vswish_grad(SmV<V> a) { SMV_FOR a.v[i] = swish_grad_f(a.v[i]); return a; }
// xhat = (x - mean) * rstd;  y = gamma * xhat + beta
template <int V> __device__ __forceinline__ SmV<V> vxhat(SmV<V> x, SmV<V> m, SmV<V> r) { SMV_FOR x.v[i] = (x.v[i] - m.v[i]) * r.v[i]; return x; }
template <int V> __device__ __forceinline__ SmV<V> vaffine(SmV<V> xh, SmV<V> g, SmV<V> b) { SMV_FOR xh.v[i] = fmaf(xh.v[i], g.v[i], b.v[i]); return xh; }
// gamma * rstd * (g - a - xhat * b)
template <int V> __device__ __forceinline__ SmV<V> vbn_dx(SmV<V> g, SmV<V> xh, SmV<V> a, SmV<V> b, SmV<V> ga, SmV<V> rs) {
  SMV_FOR g.v[i] = ga.v[i] * rs.v[i] * (g.v[i] - a.v[i] - xh.v[i] * b.v[i]);
  return g;
}
// per-channel parameter vectors in global memory (c = first channel of the group: 4 V bytes aligned)
template <int V> __device__ __forceinline__ SmV<V> vld(const float* __restrict__ p) { SmV<V> r; SMV_FOR r.v[i] = p[i]; return r; }
template <int V> __device__ __forceinline__ void vst(float* __restrict__ p, SmV<V> a) { SMV_FOR p[i] = a.v[i]; }

// ---- LDS slots: V floats moved as ONE 8- or 16-byte word
template <int V> struct SmSlot { static constexpr int F = V; };
template <int V> __device__ __forceinline__ SmV<V> lds_ld(const float* __restrict__ base, int slot) {
  SmV<V> r;
  if constexpr (V == 2) {
    const float2 t = *reinterpret_cast<const float2*>(base + slot * 2);
    r.v[0] = t.x; r.v[1] = t.y;
  } else {
    const float4 t = *reinterpret_cast<const float4*>(base + slot * 4);
    r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[V - 1] = t.w;
  }
  return r;
}
template <int V> __device__ __forceinline__ void lds_st(float* __restrict__ base, int slot, SmV<V> a) {
  if constexpr (V == 2) {
    *reinterpret_cast<float2*>(base + slot * 2) = make_float2(a.v[0], a.v[1]);
  } else {
    *reinterpret_cast<float4*>(base + slot * 4) = make_float4(a.v[0], a.v[1], a.v[2], a.v[V - 1]);
  }
}

// channel group of this workgroup (or -1).  Runs of R consecutive groups (the groups of one 128-byte line of 32 channels) share an XCD
// under round-robin dispatch, so a line crosses HBM once and is re-read through that XCD's L2.
template <int V> struct SmRun { static constexpr int R = 32 / V; };
template <int V>
__device__ __forceinline__ int sm_channel_group(int C) {
  constexpr int R = SmRun<V>::R;
  const int G = C / V, runs = (G + R - 1) / R;
  const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
  const int run = (j / R) * 8 + xcd, sub = j % R;
  if (run >= runs) return -1;
  const int gq = run * R + sub;
  return gq < G ? gq : -1;
}
static inline int sm_grid(int C, int V) {
  const int R = 32 / V, G = C / V, runs = (G + R - 1) / R;
  return ((runs + 7) / 8) * 8 * R;
}
// channels per workgroup.  Measured (profiles/r04_notes.md): the 16-byte strided accesses of a quad are what the memory side likes
// (8-byte pairs run at 0.55-0.7 of the rate, 12-byte triples are split), so pairs pay only where the compute phases dominate -- the
// 5x5 layers (25 taps: the filter-gradient accumulators' wave sums, the taps in registers) when C / 2 workgroups fit one round.
static inline int sm_group_width(int C, int k, int cus) { return (k == 5 && C % 2 == 0 && C / 2 <= cus) ? 2 : 4; }

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
template <int V> __device__ __forceinline__ SmV<V> sm_wave_sum63(SmV<V> a) { SMV_FOR a.v[i] = sm_wave_sum63(a.v[i]); return a; }

// workgroup barrier that orders LDS traffic only: global stores issued before it stay in flight (a __syncthreads() also waits for the
// wave's outstanding vector-memory operations, i.e. for every store issued so far)
__device__ __forceinline__ void sm_lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// sum of a vector pair over the workgroup (fixed order); every thread gets the totals.  red: LDS floats [kSmWaves + 1][2][4]; the
// partials of the 8 waves meet in LDS, lanes 0 .. 2 V - 1 of wave 0 add them in wave order and leave the totals in the last slot pair
// (rounds 2-3: every thread read and added all 16 partial vectors itself -- 16 ds_read_b128 per thread, 0.4 us of LDS time per sum).
// Two barriers; the caller must not touch `red` before its next barrier (every call site has one in between).
template <int V>
__device__ __forceinline__ void sm_block_sum2(SmV<V>& a, SmV<V>& b, float* red) {
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const SmV<V> wa = sm_wave_sum63(a), wb = sm_wave_sum63(b);
  if (lane == 63) {
    SMV_FOR {
      red[(wave * 2 + 0) * 4 + i] = wa.v[i];
      red[(wave * 2 + 1) * 4 + i] = wb.v[i];
    }
  }
  sm_lds_barrier();
  if (t < 8) {   // thread t: component t & 3 of vector t >> 2
    float sacc = red[t];
#pragma unroll
    for (int w = 1; w < kSmWaves; ++w) sacc += red[w * 8 + t];
    red[kSmWaves * 8 + t] = sacc;
  }
  sm_lds_barrier();
  SMV_FOR {
    a.v[i] = red[kSmWaves * 8 + i];
    b.v[i] = red[kSmWaves * 8 + 4 + i];
  }
}

// raw buffer access with 32-bit byte offsets (tensors < 2 GiB, checked on the host): an out-of-range offset reads zeros / drops the store
typedef unsigned sm_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned sm_u32x2 __attribute__((ext_vector_type(2)));
constexpr unsigned kSmOob = 0xFFFFFFF0u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t sm_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, 0x80000000u, 0x00020000);
}
// T = storage type of the tensor (float, or bf16s: `--precision bf16-storage`); `off` in BYTES
template <int V, typename T = float>
__device__ __forceinline__ SmV<V> sm_ld(__amdgpu_buffer_rsrc_t r, unsigned off) {
  SmV<V> o;
  if constexpr (sizeof(T) == 2 && V == 2) {
    const unsigned u = __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, 0);
    o.v[0] = __uint_as_float(u << 16); o.v[1] = __uint_as_float(u & 0xffff0000u);
  } else if constexpr (sizeof(T) == 2) {
    const sm_u32x2 u = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0);
    o.v[0] = __uint_as_float(u.x << 16); o.v[1] = __uint_as_float(u.x & 0xffff0000u);
    o.v[2] = __uint_as_float(u.y << 16); o.v[V - 1] = __uint_as_float(u.y & 0xffff0000u);
  } else if constexpr (V == 2) {
    const sm_u32x2 u = __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, 0);
    o.v[0] = __uint_as_float(u.x); o.v[1] = __uint_as_float(u.y);
  } else {
    const sm_u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
    o.v[0] = __uint_as_float(u.x); o.v[1] = __uint_as_float(u.y); o.v[2] = __uint_as_float(u.z); o.v[3] = __uint_as_float(u.w);
  }
  return o;
}
template <int V, typename T = float>
__device__ __forceinline__ void sm_st(__amdgpu_buffer_rsrc_t r, unsigned off, SmV<V> a) {
  if constexpr (sizeof(T) == 2 && V == 2) {
    __builtin_amdgcn_raw_buffer_store_b32(pack_bf16_pair(a.v[0], a.v[1]), r, (int)off, 0, 0);
  } else if constexpr (sizeof(T) == 2) {
    sm_u32x2 u;
    u.x = pack_bf16_pair(a.v[0], a.v[1]); u.y = pack_bf16_pair(a.v[2], a.v[V - 1]);
    __builtin_amdgcn_raw_buffer_store_b64(u, r, (int)off, 0, 0);
  } else if constexpr (V == 2) {
    sm_u32x2 u;
    u.x = __float_as_uint(a.v[0]); u.y = __float_as_uint(a.v[1]);
    __builtin_amdgcn_raw_buffer_store_b64(u, r, (int)off, 0, 0);
  } else {
    sm_u32x4 u;
    u.x = __float_as_uint(a.v[0]); u.y = __float_as_uint(a.v[1]); u.z = __float_as_uint(a.v[2]); u.w = __float_as_uint(a.v[V - 1]);
    __builtin_amdgcn_raw_buffer_store_b128(u, r, (int)off, 0, 0);
  }
}

// the values as a consumer reads them back from a tensor of storage type T
template <int V, typename T> __device__ __forceinline__ SmV<V> sm_stored(SmV<V> a) {
  if constexpr (sizeof(T) == 2) {
#pragma unroll
    for (int i = 0; i < V; ++i) a.v[i] = bf16_round_f(a.v[i]);
  }
  return a;
}

// strip decode shared by both kernels
struct SmStrip {
  bool act;
  int n, h0, w;
  unsigned off0, rstride;   // byte offset of the strip's first pixel / of one image row, in any [N,H,W,C] tensor
  int H;
  __device__ __forceinline__ unsigned poff(int j) const { return (act && h0 + j < H) ? off0 + (unsigned)j * rstride : kSmOob; }
  // the same pixel in the GROUP-BLOCKED layout [C / V][N H W][V] (a workgroup's channel group is contiguous: its 1568 pixels are 196
  // lines instead of 16 bytes of 1568 lines): byte offset = boff0 + j * brow
  unsigned boff0, brow;
  __device__ __forceinline__ unsigned bpoff(int j) const { return (act && h0 + j < H) ? boff0 + (unsigned)j * brow : kSmOob; }
  __device__ __forceinline__ bool ok(int j) const { return act && h0 + j < H; }
};
__device__ __forceinline__ SmStrip sm_strip(const SmallGeom& g, int c, int esize = 4, int V = 4) {
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
  s.off0 = (unsigned)((((n * g.H + s.h0) * g.W + w_) * g.C + c) * esize);
  s.rstride = (unsigned)(g.W * g.C * esize);
  s.boff0 = (unsigned)((((c / V) * g.npix + (n * g.H + s.h0) * g.W + w_) * V) * esize);
  s.brow = (unsigned)(g.W * V * esize);
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
  float* z0b;           // nullable out: a copy of z0 in the group-blocked layout (for the backward kernel's contiguous re-read)
  int z1_blocked;       // != 0: z1 is written in the group-blocked layout (only the backward kernel of the same layer reads it)
  SmallGeom g;
  float eps, one_minus_momentum;
#ifdef SM_DBG
  unsigned long long* stamps;
#endif
};

// taps in registers (fully unrolled window loop) where K*K*V floats fit beside the window: every form but the 5x5 quads
template <int K, int V> struct SmTapsInRegs { static constexpr bool value = K == 3 || V <= 2; };

template <int K, int V, typename T = float>   // T: storage type of z0, z1, a1 (and a0)
__global__ __launch_bounds__(kSmThreads) void mbconv_dw_fwd_small_k(SmallFwd p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  typedef SmV<V> VT;
  constexpr int F = SmSlot<V>::F;
  const SmallGeom g = p.g;
  float* tileA = smem;                                                   // [npix] slots: a0
  float* wl = tileA + (size_t)g.npix * F;                                // [K*K] slots: filter taps
  float* red = wl + K * K * F;                                           // [kSmWaves + 1][2][4]
  float* pool = red + (kSmWaves + 1) * 2 * 4;                               // [512] slots: per-strip sums of a1
  double* fold = reinterpret_cast<double*>(pool + kSmThreads * F);       // [2 V columns]
  float* stat = reinterpret_cast<float*>(fold + 8);                      // mean0[4] rstd0[4] var0[4]
  const int cq = sm_channel_group<V>(g.C);
  if (cq < 0) return;
  SM_STAMP_DECL;
  SM_STAMP(0);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int c = cq * V;
  const SmStrip st = sm_strip(g, c, (int)sizeof(T), V);
  // own pixels' z0 first (does not depend on the statistics fold)
  const __amdgpu_buffer_rsrc_t rz0 = sm_rsrc(p.z0);
  VT zin[4];
  // z0b == z0: the expand conv has written z0 in the group-blocked layout already (mliis_conv2d_fwd, MLIIS_DT_BLOCKED): contiguous
  // loads here, no copy to make
  const bool z0_in_b = p.z0b != nullptr && static_cast<const void*>(p.z0b) == static_cast<const void*>(p.z0);
#pragma unroll
  for (int j = 0; j < 4; ++j) zin[j] = sm_ld<V, T>(rz0, z0_in_b ? st.bpoff(j) : st.poff(j));
  if (p.z0b != nullptr && !z0_in_b) {   // (contiguous stores: lanes run along W)
    const __amdgpu_buffer_rsrc_t rzb = sm_rsrc(p.z0b);
#pragma unroll
    for (int j = 0; j < 4; ++j) sm_st<V, T>(rzb, st.bpoff(j), zin[j]);
  }
  if (t < K * K) lds_st<V>(wl, t, vld<V>(p.w + (long long)t * g.C + c));
  // ---- fold the expand conv's stage-1 statistics of this group (double precision, fixed order): wave w owns column w of the 2 V
  //      {sum, sum of squares} x V channels; its lanes take the partial blocks (ONE memory round trip for up to 64 of them)
  if (wave < 2 * V) {
    const int v = wave / V, ch = wave - v * V;
    double acc = 0.0;
    for (int b = lane; b < p.nblk0; b += 64) acc += (double)p.part0[((long long)b * 2 + v) * g.C + c + ch];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) fold[wave] = acc;
  }
  // the moving averages are read now (their latency hides behind everything below) and written at the very end
  float mmv0 = 0.f, mvv0 = 0.f, mmv1 = 0.f, mvv1 = 0.f;
  if (t < V && p.mm0 != nullptr) {
    mmv0 = p.mm0[c + t];
    mvv0 = p.mv0[c + t];
  }
  if (t < V && p.mm1 != nullptr) {
    mmv1 = p.mm1[c + t];
    mvv1 = p.mv1[c + t];
  }
  SM_STAMP(1);
  __syncthreads();
  SM_STAMP(2);
  if (t < V) {
    const double s1 = fold[t], s2 = fold[V + t];
    const double inv_n = 1.0 / (double)g.npix;
    const double m = s1 * inv_n;
    double var = s2 * inv_n - m * m;
    if (var < 0.0) var = 0.0;
    stat[t] = (float)m;
    stat[4 + t] = (float)(1.0 / sqrt(var + (double)p.eps));
    stat[8 + t] = (float)var;
  }
  __syncthreads();
  SM_STAMP(3);
  const VT m0 = vld<V>(stat), r0 = vld<V>(stat + 4);
  {
    const VT ga = vld<V>(p.gamma0 + c), be = vld<V>(p.beta0 + c);
    const __amdgpu_buffer_rsrc_t ra0 = sm_rsrc(p.a0 != nullptr ? p.a0 : p.z0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (st.ok(j)) {
        const VT a = vswish(vaffine(vxhat(zin[j], m0, r0), ga, be));
        lds_st<V>(tileA, (st.n * g.H + st.h0 + j) * g.W + st.w, a);
        if (p.a0 != nullptr) sm_st<V, T>(ra0, st.poff(j), a);
      }
    }
  }
  __syncthreads();
  SM_STAMP(4);
  // ---- depthwise stencil out of LDS: outputs (h0 + j, w), j < 4; TF-SAME, stride 1.  Window row r (image row h0 - P + r) is read once
  //      and feeds the (output row j, filter row ky = r - j) pairs.
  constexpr int P = (K - 1) / 2;
  VT acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = vzero<V>();
  if (st.act) {
    const float* imgA = tileA + (size_t)st.n * g.H * g.W * F;
    if constexpr (SmTapsInRegs<K, V>::value) {   // all taps and the window rows in registers, fully unrolled
      VT wreg[K * K];
#pragma unroll
      for (int i = 0; i < K * K; ++i) wreg[i] = lds_ld<V>(wl, i);
#pragma unroll
      for (int r = 0; r < K + 3; ++r) {
        const int hi = st.h0 - P + r;
        const bool rok = (unsigned)hi < (unsigned)g.H;
        VT row[K];
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const int wi = st.w - P + kx;
          row[kx] = (rok && (unsigned)wi < (unsigned)g.W) ? lds_ld<V>(imgA, hi * g.W + wi) : vzero<V>();
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int ky = r - j;
          if (ky >= 0 && ky < K) {
#pragma unroll
            for (int kx = 0; kx < K; ++kx) acc[j] = vfma(row[kx], wreg[ky * K + kx], acc[j]);
          }
        }
        if (K == 5) __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      // 5x5 quads: a ROLLED loop over the window rows; the taps of the (output row j, filter row r - j) pairs are read from LDS where
      // they are used (one address for the whole wave: a broadcast).  Unrolled, the compiler hoists the 25 taps (100 VGPRs) and the
      // window rows on top of each other and spills 65-320 VGPRs -- and every spill is a memory round trip the next barrier waits for.
#pragma unroll 1
      for (int r = 0; r < K + 3; ++r) {
        const int hi = st.h0 - P + r;
        const bool rok = (unsigned)hi < (unsigned)g.H;
        VT row[K];
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const int wi = st.w - P + kx;
          row[kx] = (rok && (unsigned)wi < (unsigned)g.W) ? lds_ld<V>(imgA, hi * g.W + wi) : vzero<V>();
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int ky = r - j;
          if (ky >= 0 && ky < K) {
#pragma unroll
            for (int kx = 0; kx < K; ++kx) acc[j] = vfma(row[kx], lds_ld<V>(wl, ky * K + kx), acc[j]);
          }
        }
      }
    }
  }
  // ---- z1 leaves as soon as it exists: every barrier from here on orders LDS traffic only (sm_lds_barrier), so the stores drain
  //      under the statistics instead of at the end of the kernel (rows beyond H / idle lanes: out-of-range offset, store dropped)
  {
    const __amdgpu_buffer_rsrc_t rz1 = sm_rsrc(p.z1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[j] = sm_stored<V, T>(acc[j]);   // (bf16 storage: the statistics see what the backward pass will read)
      sm_st<V, T>(rz1, p.z1_blocked ? st.bpoff(j) : st.poff(j), acc[j]);
    }
  }
  // ---- statistics of z1: exact two-pass in the workgroup (mean, then centred second moment)
  VT s1 = vzero<V>(), s2 = vzero<V>();
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (st.ok(j)) s1 = vadd(s1, acc[j]);
  SM_STAMP(5);
  sm_block_sum2<V>(s1, s2, red);
  SM_STAMP(6);
  const float inv_n = 1.0f / (float)g.npix;
  const VT m1 = vscale(s1, inv_n);
  VT d2 = vzero<V>(), dummy = vzero<V>();
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (st.ok(j)) {
      const VT d = vsub(acc[j], m1);
      d2 = vfma(d, d, d2);
    }
  sm_block_sum2<V>(d2, dummy, red);
  SM_STAMP(7);
  const VT var1 = vscale(d2, inv_n);
  VT r1;
  SMV_FOR r1.v[i] = 1.0f / sqrtf(var1.v[i] + p.eps);
  // ---- a1 = swish(bn1(z1)), per-image pooled mean
  VT a1v[4];
  {
    const VT ga = vld<V>(p.gamma1 + c), be = vld<V>(p.beta1 + c);
    VT ps = vzero<V>();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a1v[j] = sm_stored<V, T>(vswish(vaffine(vxhat(acc[j], m1, r1), ga, be)));
      if (st.ok(j)) ps = vadd(ps, a1v[j]);
    }
    lds_st<V>(pool, t, ps);
    const __amdgpu_buffer_rsrc_t ra1 = sm_rsrc(p.a1);
#pragma unroll
    for (int j = 0; j < 4; ++j) sm_st<V, T>(ra1, st.poff(j), a1v[j]);
  }
  sm_lds_barrier();
  SM_STAMP(8);
  // image n owns strips [n * HS * W, (n + 1) * HS * W): one wave per image (fixed order), DPP wave sum
  {
    const int per = g.HS * g.W;
    for (int img = wave; img < g.N; img += kSmWaves) {
      VT a = vzero<V>();
      for (int i = lane; i < per; i += 64) a = vadd(a, lds_ld<V>(pool, img * per + i));
      a = sm_wave_sum63(a);
      if (lane == 63) vst<V>(p.s + (long long)img * g.C + c, vscale(a, 1.0f / (float)(g.H * g.W)));
    }
  }
  SM_STAMP(9);
  // ---- batch statistics for the backward pass, moving averages (biased variance: non-fused TpuBatchNormalization, utils.py:87-134)
  if (t < V) {
    const float mf = stat[t], vf = stat[8 + t];
    p.mean0[c + t] = mf;
    p.rstd0[c + t] = stat[4 + t];
    if (p.mm0 != nullptr) {
      p.mm0[c + t] = mmv0 - (mmv0 - mf) * p.one_minus_momentum;
      p.mv0[c + t] = mvv0 - (mvv0 - vf) * p.one_minus_momentum;
    }
    float m1t = 0.f, r1t = 0.f, v1t = 0.f;
    SMV_FOR if (i == t) {
      m1t = m1.v[i];
      r1t = r1.v[i];
      v1t = var1.v[i];
    }
    p.mean1[c + t] = m1t;
    p.rstd1[c + t] = r1t;
    if (p.mm1 != nullptr) {
      p.mm1[c + t] = mmv1 - (mmv1 - m1t) * p.one_minus_momentum;
      p.mv1[c + t] = mvv1 - (mvv1 - v1t) * p.one_minus_momentum;
    }
  }
  SM_STAMP(10);
  SM_STAMP_FLUSH(p.stamps);
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
  const float* z0b;        // nullable: z0 in the group-blocked layout (mbconv_dw_fwd_small's z0b), read instead of z0
  int z1_blocked;          // bit 0: z1 is in the group-blocked layout; bit 1: so is da2 (mliis_conv2d_bwd_data_gate, MLIIS_DT_BLOCKED)
  SmallGeom g;
#ifdef SM_DBG
  unsigned long long* stamps;
#endif
};

template <int K, int V, typename T = float>   // T: storage type of da2, z1, z0 and dz0
__global__ __launch_bounds__(kSmThreads) void mbconv_dw_bwd_small_k(SmallBwd p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  typedef SmV<V> VT;
  constexpr int F = SmSlot<V>::F;
  const SmallGeom g = p.g;
  float* tileA = smem;                                     // [npix] slots: a0 = swish(bn0(z0))
  float* tileD = tileA + (size_t)g.npix * F;               // [npix] slots: dz1
  float* wl = tileD + (size_t)g.npix * F;                  // [K*K] slots
  float* red = wl + K * K * F;                             // [kSmWaves + 1][2][4]
  float* wred = red + (kSmWaves + 1) * 2 * 4;
  // 5x5 quads: 25 float4 filter-gradient accumulators leave no room for values that are only needed again after the two window loops --
  // xhat0 of the own strip is parked in a third LDS tile (read back by the thread that wrote it: no barrier) and the bn0 parameters
  // are loaded again behind the loops (26 spilled registers otherwise, each a memory round trip the next barrier waits for)
  constexpr bool PARK = K == 5 && V == 4;
  float* tileX = wred + kSmWaves * K * K * F;              // [npix] slots (PARK only): xhat0                 // [kSmWaves][K*K] slots: filter-gradient partials
  const int cq = sm_channel_group<V>(g.C);
  if (cq < 0) return;
  SM_STAMP_DECL;
  SM_STAMP(0);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int c = cq * V;
  const SmStrip st = sm_strip(g, c, (int)sizeof(T), V);
  const float inv_n = 1.0f / (float)g.npix;
  // all global loads of the own strip up front: one memory round trip
  VT zv[4], dv[4], z0v[4];
  {
    const bool zb = p.z0b != nullptr;
    const __amdgpu_buffer_rsrc_t rz1 = sm_rsrc(p.z1), rd = sm_rsrc(p.da2), rz0 = sm_rsrc(zb ? p.z0b : p.z0);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      zv[j] = sm_ld<V, T>(rz1, (p.z1_blocked & 1) ? st.bpoff(j) : st.poff(j));
      dv[j] = sm_ld<V, T>(rd, (p.z1_blocked & 2) ? st.bpoff(j) : st.poff(j));
      z0v[j] = sm_ld<V, T>(rz0, zb ? st.bpoff(j) : st.poff(j));
    }
  }
  if (t < K * K) lds_st<V>(wl, t, vld<V>(p.w + (long long)t * g.C + c));
  VT gt = vfill<V>(1.f), ca = vzero<V>();
  if (st.act && p.gate != nullptr) gt = vld<V>(p.gate + (long long)st.n * g.C + c);
  if (st.act && p.chan_add != nullptr) ca = vld<V>(p.chan_add + (long long)st.n * g.C + c);
  const VT m0 = vld<V>(p.mean0 + c), r0_ = vld<V>(p.rstd0 + c), ga0_ = vld<V>(p.gamma0 + c), be0_ = vld<V>(p.beta0 + c);
  // ---- bn1 backward: g = (da2 * gate + chan_add) * swish'(gamma1 * xhat + beta1); sums; dz1 -> LDS
  {
    const VT m1 = vld<V>(p.mean1 + c), r1 = vld<V>(p.rstd1 + c), ga = vld<V>(p.gamma1 + c), be = vld<V>(p.beta1 + c);
    VT s1 = vzero<V>(), s2 = vzero<V>();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const VT xh = vxhat(zv[j], m1, r1);
      VT gg = vmul(vfma(dv[j], gt, ca), vswish_grad(vaffine(xh, ga, be)));
      if (!st.ok(j)) gg = vzero<V>();
      zv[j] = xh;      // (reuse: xhat)
      dv[j] = gg;      // (reuse: g)
      s1 = vadd(s1, gg);
      s2 = vfma(gg, xh, s2);
    }
    SM_STAMP(1);
    sm_block_sum2<V>(s1, s2, red);
    SM_STAMP(2);
    if (t == 0) {
      vst<V>(p.dbeta1 + c, s1);
      vst<V>(p.dgamma1 + c, s2);
    }
    const VT a = vscale(s1, inv_n), b = vscale(s2, inv_n);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const VT d = vbn_dx(dv[j], zv[j], a, b, ga, r1);
      dv[j] = st.ok(j) ? d : vzero<V>();   // dz1 of the own strip stays in registers for the filter gradient (zero on rows beyond H)
      if (st.ok(j)) lds_st<V>(tileD, (st.n * g.H + st.h0 + j) * g.W + st.w, d);
    }
  }
  // ---- a0 = swish(bn0(z0)) -> LDS; xhat0 kept
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const VT xh = vxhat(z0v[j], m0, r0_);
    z0v[j] = xh;
    if (st.ok(j)) lds_st<V>(tileA, (st.n * g.H + st.h0 + j) * g.W + st.w, vswish(vaffine(xh, ga0_, be0_)));
    if (PARK && st.ok(j)) lds_st<V>(tileX, (st.n * g.H + st.h0 + j) * g.W + st.w, xh);
  }
  __syncthreads();
  SM_STAMP(3);
  constexpr int P = (K - 1) / 2;
  // ---- depthwise filter gradient: dw[ky][kx] = sum_pixels a0[h + ky - P][w + kx - P] * dz1[h][w]; window row r (image row h0 - P + r)
  //      read once for its (j, ky = r - j) pairs; K*K accumulators, DPP wave sums, the 8 waves through LDS
  {
    VT wacc[K * K];
#pragma unroll
    for (int i = 0; i < K * K; ++i) wacc[i] = vzero<V>();
    if (st.act) {
      const float* imgA = tileA + (size_t)st.n * g.H * g.W * F;
#pragma unroll
      for (int r = 0; r < K + 3; ++r) {
        const int hi = st.h0 - P + r;
        const bool rok = (unsigned)hi < (unsigned)g.H;
        VT row[K];
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
          const int wi = st.w - P + kx;
          row[kx] = (rok && (unsigned)wi < (unsigned)g.W) ? lds_ld<V>(imgA, hi * g.W + wi) : vzero<V>();
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int ky = r - j;
          if (ky >= 0 && ky < K) {
#pragma unroll
            for (int kx = 0; kx < K; ++kx) wacc[ky * K + kx] = vfma(row[kx], dv[j], wacc[ky * K + kx]);
          }
        }
        if (K == 5) __builtin_amdgcn_sched_barrier(0);
      }
    }
    SM_STAMP(4);
#pragma unroll
    for (int i = 0; i < K * K; ++i) {
      const VT v = sm_wave_sum63(wacc[i]);
      if (lane == 63) lds_st<V>(wred, wave * (K * K) + i, v);
    }
  }
  __builtin_amdgcn_sched_barrier(0);   // (the backward-data phase's taps / window loads must not be hoisted above the K*K accumulators' end)
  SM_STAMP(5);
  // ---- depthwise backward-data out of LDS: da0[h][w] = sum dz1[h + P - ky][w + P - kx] * w[ky][kx]  (stride 1, SAME: P = (K-1)/2)
  VT acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = vzero<V>();
  if (st.act) {
    const float* imgD = tileD + (size_t)st.n * g.H * g.W * F;
    // window row r = image row h0 - P + r; pairs with (j, ky): r - j = (K - 1) - ky.  Taps in registers and unrolled, or (5x5 quads)
    // rolled with the taps read from LDS where they are used (see the forward kernel)
    constexpr bool REGS = SmTapsInRegs<K, V>::value;
    VT wreg[REGS ? K * K : 1];
    if constexpr (REGS) {
#pragma unroll
      for (int i = 0; i < K * K; ++i) wreg[i] = lds_ld<V>(wl, i);
    }
    auto body = [&](int r) {
      const int hi = st.h0 - P + r;
      const bool rok = (unsigned)hi < (unsigned)g.H;
      VT row[K];
#pragma unroll
      for (int x = 0; x < K; ++x) {
        const int wi = st.w - P + x;
        row[x] = (rok && (unsigned)wi < (unsigned)g.W) ? lds_ld<V>(imgD, hi * g.W + wi) : vzero<V>();
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ky = (K - 1) - (r - j);
        if (ky >= 0 && ky < K) {
#pragma unroll
          for (int x = 0; x < K; ++x) {
            const VT wv = REGS ? wreg[REGS ? ky * K + (K - 1 - x) : 0] : lds_ld<V>(wl, ky * K + (K - 1 - x));
            acc[j] = vfma(row[x], wv, acc[j]);
          }
        }
      }
    };
    if constexpr (REGS) {
#pragma unroll
      for (int r = 0; r < K + 3; ++r) {
        body(r);
        if (K == 5) __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll 1
      for (int r = 0; r < K + 3; ++r) body(r);
    }
  }
  SM_STAMP(6);
  __syncthreads();   // (the filter-gradient partials of all waves are in LDS)
  SM_STAMP(7);
  if (t < K * K) {
    VT v = lds_ld<V>(wred, t);
#pragma unroll
    for (int wv = 1; wv < kSmWaves; ++wv) v = vadd(v, lds_ld<V>(wred, wv * (K * K) + t));
    vst<V>(p.dw + (long long)t * g.C + c, v);
  }
  // ---- bn0 backward: g0 = da0 * swish'(gamma0 * xhat0 + beta0); sums; dz0
  {
    VT ga0 = ga0_, be0 = be0_, r0 = r0_;
    if constexpr (PARK) {
      const float* pg = p.gamma0 + c;
      const float* pb = p.beta0 + c;
      const float* pr = p.rstd0 + c;
      asm volatile("" : "+v"(pg), "+v"(pb), "+v"(pr));   // (opaque: a fresh load, not the value kept live across the loops)
      ga0 = vld<V>(pg);
      be0 = vld<V>(pb);
      r0 = vld<V>(pr);
#pragma unroll
      for (int j = 0; j < 4; ++j) z0v[j] = st.ok(j) ? lds_ld<V>(tileX, (st.n * g.H + st.h0 + j) * g.W + st.w) : vzero<V>();
    }
    VT s1 = vzero<V>(), s2 = vzero<V>();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      VT gg = vmul(acc[j], vswish_grad(vaffine(z0v[j], ga0, be0)));
      if (!st.ok(j)) gg = vzero<V>();
      acc[j] = gg;
      s1 = vadd(s1, gg);
      s2 = vfma(gg, z0v[j], s2);
    }
    SM_STAMP(8);
    sm_block_sum2<V>(s1, s2, red);
    SM_STAMP(9);
    if (t == 0) {
      vst<V>(p.dbeta0 + c, s1);
      vst<V>(p.dgamma0 + c, s2);
    }
    const VT a = vscale(s1, inv_n), b = vscale(s2, inv_n);
    const __amdgpu_buffer_rsrc_t rdz = sm_rsrc(p.dz0);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (st.ok(j)) sm_st<V, T>(rdz, st.poff(j), vbn_dx(acc[j], z0v[j], a, b, ga0, r0));
  }
  SM_STAMP(10);
  SM_STAMP_FLUSH(p.stamps);
}
#undef SMV_FOR

static bool small_geom(int N, int H, int W, int C, int k, int stride, SmallGeom* g) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || stride != 1 || (k != 3 && k != 5)) return false;
  const long long npix = (long long)N * H * W;
  const int HS = (H + 3) / 4;
  const long long nitems = (long long)N * HS * W;
  if (npix > kSmMaxPix || nitems > kSmThreads || npix * C * 4 >= (1LL << 31)) return false;
  *g = SmallGeom{N, H, W, C, HS, (int)nitems, (int)npix};
  return true;
}
// (sized for 16-byte slots whatever V is: the attribute below is raised once per instantiation to the largest eligible request)
static size_t small_fwd_lds(const SmallGeom& g, int k) {
  return ((size_t)g.npix + (size_t)k * k + (kSmWaves + 1) * 2 + kSmThreads) * 16 + 8 * 8 + 16 * 4;
}
static size_t small_bwd_lds(const SmallGeom& g, int k) { return ((size_t)g.npix * (k == 5 ? 3 : 2) + (size_t)k * k + (kSmWaves + 1) * 2 + (size_t)kSmWaves * k * k) * 16; }

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
static int sm_num_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
              ? prop.multiProcessorCount : 256;
  }
  return cus;
}
static SmallGeom small_max_geom() { return SmallGeom{1, 1, 1, 4, 1, kSmThreads, kSmMaxPix}; }

template <int K, int V, typename T>
static int small_launch_fwd_t(const SmallFwd& p, int C, size_t lds, hipStream_t stream) {
  static int attr = 0;
  int rc;
  if ((rc = small_attr(mbconv_dw_fwd_small_k<K, V, T>, small_fwd_lds(small_max_geom(), K), &attr))) return rc;
  hipLaunchKernelGGL((mbconv_dw_fwd_small_k<K, V, T>), dim3(sm_grid(C, V)), dim3(kSmThreads), lds, stream, p);
  return MLIIS_OK;
}
template <int K, int V>
static int small_launch_fwd(const SmallFwd& p, int C, size_t lds, int dt, hipStream_t stream) {
  return dt == MLIIS_DT_BF16 ? small_launch_fwd_t<K, V, bf16s>(p, C, lds, stream) : small_launch_fwd_t<K, V, float>(p, C, lds, stream);
}
template <int K, int V, typename T>
static int small_launch_bwd_t(const SmallBwd& p, int C, size_t lds, hipStream_t stream) {
  static int attr = 0;
  int rc;
  if ((rc = small_attr(mbconv_dw_bwd_small_k<K, V, T>, small_bwd_lds(small_max_geom(), K), &attr))) return rc;
  hipLaunchKernelGGL((mbconv_dw_bwd_small_k<K, V, T>), dim3(sm_grid(C, V)), dim3(kSmThreads), lds, stream, p);
  return MLIIS_OK;
}
template <int K, int V>
static int small_launch_bwd(const SmallBwd& p, int C, size_t lds, int dt, hipStream_t stream) {
  return dt == MLIIS_DT_BF16 ? small_launch_bwd_t<K, V, bf16s>(p, C, lds, stream) : small_launch_bwd_t<K, V, float>(p, C, lds, stream);
}

}  // namespace mliis

using namespace mliis;

extern "C" {

int mliis_mbconv_dw_small_supported(int N, int H, int W, int C, int k, int stride) {
  SmallGeom g;
  return small_geom(N, H, W, C, k, stride, &g) ? 1 : 0;
}

// channels per workgroup the small-map launches use for C channels and a k x k filter (2 or 4: mbconv_small.hip header); grid = C / that,
// rounded up to the XCD-grouped form.  group_width = 0: the planner's choice; 2 | 4 force a form (tests, A/B runs).
int mliis_mbconv_dw_small_group_width(int C, int k) { return C > 0 ? sm_group_width(C, k, sm_num_cus()) : 0; }

int mliis_mbconv_dw_fwd_small(const float* z0, const float* part0, int nblk0, const float* gamma0, const float* beta0, float* mean0,
                              float* rstd0, float* moving_mean0, float* moving_var0, const float* w, const float* gamma1, const float* beta1,
                              float* mean1, float* rstd1, float* moving_mean1, float* moving_var1, float* a0, float* z1, float* a1, float* s,
                              int N, int H, int W, int C, int k, float eps, float momentum, int group_width, int act_dtype, float* z0_blocked,
                              int z1_blocked, hipStream_t stream) {
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
  const int V = group_width ? group_width : sm_group_width(C, k, sm_num_cus());
  MLIIS_REQUIRE((V == 2 || V == 4) && C % V == 0, MLIIS_ERR_ARG, "mbconv_dw_fwd_small: group_width %d must be 0, 2 or 4 and divide C = %d", V, C);
  MLIIS_REQUIRE(aligned16(z0_blocked), MLIIS_ERR_ALIGN, "mbconv_dw_fwd_small: z0_blocked must be 16-byte aligned");
  SmallFwd p{z0, part0, nblk0, gamma0, beta0, mean0, rstd0, moving_mean0, moving_var0, w, gamma1, beta1, mean1, rstd1, moving_mean1,
             moving_var1, a0, z1, a1, s, z0_blocked, z1_blocked, g, eps, 1.0f - momentum};
#ifdef SM_DBG
  p.stamps = getenv("MLIIS_SM_STAMPS") ? reinterpret_cast<unsigned long long*>(strtoull(getenv("MLIIS_SM_STAMPS"), nullptr, 10)) : nullptr;
#endif
  const size_t lds = small_fwd_lds(g, k);
  int rc;
  MLIIS_REQUIRE(act_dtype == MLIIS_DT_F32 || act_dtype == MLIIS_DT_BF16, MLIIS_ERR_ARG, "mbconv_dw_fwd_small: bad act_dtype");
  if (k == 3) rc = V == 2 ? small_launch_fwd<3, 2>(p, C, lds, act_dtype, stream) : small_launch_fwd<3, 4>(p, C, lds, act_dtype, stream);
  else rc = V == 2 ? small_launch_fwd<5, 2>(p, C, lds, act_dtype, stream) : small_launch_fwd<5, 4>(p, C, lds, act_dtype, stream);
  if (rc) return rc;
  MLIIS_CHECK_LAUNCH("mbconv_dw_fwd_small");
  return MLIIS_OK;
}

int mliis_mbconv_dw_bwd_small(const float* da2, const float* gate, const float* chan_add, const float* z1, const float* mean1,
                              const float* rstd1, const float* gamma1, const float* beta1, const float* w, const float* z0, const float* mean0,
                              const float* rstd0, const float* gamma0, const float* beta0, float* dgamma1, float* dbeta1, float* dw,
                              float* dgamma0, float* dbeta0, float* dz0, int N, int H, int W, int C, int k, int group_width, int act_dtype,
                              const float* z0_blocked, int z1_blocked, hipStream_t stream) {
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
  const int V = group_width ? group_width : sm_group_width(C, k, sm_num_cus());
  MLIIS_REQUIRE((V == 2 || V == 4) && C % V == 0, MLIIS_ERR_ARG, "mbconv_dw_bwd_small: group_width %d must be 0, 2 or 4 and divide C = %d", V, C);
  MLIIS_REQUIRE(aligned16(z0_blocked), MLIIS_ERR_ALIGN, "mbconv_dw_bwd_small: z0_blocked must be 16-byte aligned");
  SmallBwd p{da2, gate, chan_add, z1, mean1, rstd1, gamma1, beta1, w, z0, mean0, rstd0, gamma0, beta0, dgamma1, dbeta1, dw, dgamma0, dbeta0,
             dz0, z0_blocked, z1_blocked, g};
#ifdef SM_DBG
  p.stamps = getenv("MLIIS_SM_STAMPS") ? reinterpret_cast<unsigned long long*>(strtoull(getenv("MLIIS_SM_STAMPS"), nullptr, 10)) : nullptr;
#endif
  const size_t lds = small_bwd_lds(g, k);
  int rc;
  MLIIS_REQUIRE(act_dtype == MLIIS_DT_F32 || act_dtype == MLIIS_DT_BF16, MLIIS_ERR_ARG, "mbconv_dw_bwd_small: bad act_dtype");
  if (k == 3) rc = V == 2 ? small_launch_bwd<3, 2>(p, C, lds, act_dtype, stream) : small_launch_bwd<3, 4>(p, C, lds, act_dtype, stream);
  else rc = V == 2 ? small_launch_bwd<5, 2>(p, C, lds, act_dtype, stream) : small_launch_bwd<5, 4>(p, C, lds, act_dtype, stream);
  if (rc) return rc;
  MLIIS_CHECK_LAUNCH("mbconv_dw_bwd_small");
  return MLIIS_OK;
}
}
