// Row-marching depthwise k x k convolution (k in {3,5}, stride in {1,2}, TF-SAME padding), NHWC fp32, with the PRECEDING batch norm
// + swish applied while the input is staged -- the depthwise half of an MBConv block on the large maps:
//
//     forward    y  = dwconv(a),  a = swish(bn(z))                 z read once, a never written, + the NEXT batch norm's stage-1 sums
//     backward   da = dwconv^T(dy),  dW = a (*) dy,  {sum g, sum g xhat} of the batch norm's backward  -- ONE pass: dy and z read once
//
// Reference call sites: models/efficientnet/efficientnet_model.py:185-196,266-271 (expand conv -> BN -> swish -> depthwise conv),
// models/efficientnet/utils.py:87-134,219-222; stem: efficientnet_model.py:409-414 (its BN + swish feed block 0's depthwise conv).
//
// Why this shape.  The kernels of dwconv.hip give a thread one output row x 4 columns and fetch every input row K times through
// L1 / L2; the tile kernel stages a halo'd tile per workgroup and exits (1.45-1.96x staged bytes for 3x3 / 5x5, one round trip of
// latency per 14 KB of output).  Here a workgroup owns (image, 32 channels, a column band, a range of output rows) and MARCHES down
// the rows: the input rows under the current RS output rows sit in an LDS ring, the rows of the next two steps are in flight in
// registers (raw buffer loads: out-of-image pixels come back as zeros, no branches), and every input byte crosses HBM once per
// workgroup -- only the K - S rows at the seam between two row ranges and the K - S columns between two bands are fetched twice.
// Because every element is STAGED once, the batch norm + swish in front of the convolution costs one evaluation per element (a
// sliding-window kernel would recompute it K^2 / S^2 times), which deletes the BN-apply pass over the expanded tensor and its
// write + read.  HBM-bound: algorithmic bytes fwd 4 (in + out + k^2 C), bwd 4 (2 in + out + 2 k^2 C).
//
// Layout.  256 threads = 8 channel quads (q, one 128-byte line per pixel) x 32 pixel lanes (p).  Ring row = IBWP pixels x 8 quads
// of float4; pixel columns are exchanged in pairs by bit 2 of the column (col ^ ((col >> 2) & 1)) so that the four strips a
// ds_read_b128 lane group touches fall on four different 64-byte bank quarters (strips are 4 input pixels = 512 bytes apart:
// unswizzled, strips s and s + 2 collide).  Compute: pixel lane -> (output row ro = p >> 3 of the step, strip sx = p & 7 of TS
// outputs); the K x ((TS - 1) S + K) window of a strip is read from the ring once per filter row.
// Every reduction is deterministic (lane butterflies, then the four waves through LDS in a fixed order; no atomics).
#include "bn_fold.hpp"
#include "common.hpp"

namespace mliis {

#ifndef DWM_STORE_AUX
#define DWM_STORE_AUX 2   // nt: measured cold at N = 8 (tools/bench_dwmarch.py): 112x112x32 forward 10.8 us plain, 8.5 nt, 9.7 sc1, 9.6 sc0 sc1
#endif
typedef unsigned dwm_u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kDwmOob = 0xFFFFFFF0u;   // beyond num_records of the buffer resource: the load returns zeros and moves no data

// Buffer resource over a kernel-argument pointer.  The halves go through readfirstlane so that the descriptor is PROVABLY wave-uniform:
// otherwise hipcc wraps every buffer load / store in a waterfall loop (readfirstlane x 4, compare, saveexec), which serialises the
// loads of a batch.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dwm_rsrc(const void* p) {
  const unsigned long long u = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, 0x80000000u, 0x00020000);
}

#ifdef DWM_DBG
#define DWM_DBG_ON(a, bit) (((a).dbg & (bit)) != 0)
#define DWM_STAMP(k) do { if (k < 16) stamp[k] = wall_clock64(); } while (0)
#else
#define DWM_DBG_ON(a, bit) false
#define DWM_STAMP(k) do { } while (0)
#endif

// T = storage type of the tensor behind the resource (float: 16-byte quads; bf16s: 8-byte quads, widened exactly); `off` in BYTES
typedef unsigned dwm_u32x2 __attribute__((ext_vector_type(2)));
template <typename T = float>
__device__ __forceinline__ float4 dwm_load(__amdgpu_buffer_rsrc_t r, int off, bool ok) {
  if constexpr (sizeof(T) == 4) {
    const dwm_u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r, ok ? off : (int)kDwmOob, 0, 0);
    return make_float4(__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w));
  } else {
    const dwm_u32x2 u = __builtin_amdgcn_raw_buffer_load_b64(r, ok ? off : (int)kDwmOob, 0, 0);
    return unpack_bf16x4(make_uint2(u.x, u.y));
  }
}

template <typename T = float>
__device__ __forceinline__ void dwm_store(__amdgpu_buffer_rsrc_t r, int off, bool ok, const float4 v) {
  if constexpr (sizeof(T) == 2) {   // bf16 storage: round to nearest even, 8-byte quad
    const uint2 pk = pack_bf16x4(v);
    dwm_u32x2 u2;
    u2.x = pk.x; u2.y = pk.y;
    __builtin_amdgcn_raw_buffer_store_b64(u2, r, ok ? off : (int)kDwmOob, 0, DWM_STORE_AUX);
    return;
  }
  dwm_u32x4 u;
  u.x = __float_as_uint(v.x); u.y = __float_as_uint(v.y); u.z = __float_as_uint(v.z); u.w = __float_as_uint(v.w);
  // (out of range: dropped by the range check, no branch.)  DWM_STORE_AUX: cache policy of the output stores (gfx950 aux bits:
  // 1 = sc0, 2 = nt, 16 = sc1).  Plain stores stay dirty in the XCD's L2 until the end-of-kernel release writes them back in one
  // burst behind the last workgroup (output bytes / ~6 TB/s added to every launch); write-through stores leave while the march runs.
  __builtin_amdgcn_raw_buffer_store_b128(u, r, ok ? off : (int)kDwmOob, 0, DWM_STORE_AUX);
}
__device__ __forceinline__ float4 f4sel(bool ok, const float4 v) { return ok ? v : make_float4(0.f, 0.f, 0.f, 0.f); }
template <typename T> __device__ __forceinline__ float4 dwm_stored(const float4 v) { return stored_value(static_cast<const T*>(nullptr), v); }

// -DDWM_K5_TS2 (experiment of round 6): the 5x5 stride-1 layer with TWO outputs per thread along W instead of four --
// bands of 14 columns (twice the workgroups), a 5 x 6 window per strip, fewer live registers in the window phase
#ifndef DWM_K5_TS2
#define DWM_K5_TS2 0
#endif
template <int K, int S>
struct MarchCfg {
  static constexpr int TS = (S == 1 && !(K == 5 && DWM_K5_TS2)) ? 4 : 2;   // outputs per thread along W (TS * S input pixels between strips)
  static constexpr int BW = 7 * TS;                    // produced columns of a band: 7 strips (28 | 14), so that ...
  static constexpr int SPR = BW / TS;
  static constexpr int RS = 4;                         // output rows per step: 4 rows x 8 strip slots = the 32 pixel lanes
  static constexpr int IBW = (BW - 1) * S + K;         // ... the input columns under a band (30 | 32 | 29 | 31) fit the 32 pixel lanes
  static constexpr int IBWP = (IBW + 1) & ~1;          // ring row pitch in pixels (even: the bank swizzle exchanges column pairs)
  static constexpr int WIN = (RS - 1) * S + K;         // input rows under one step
  static constexpr int NEW = RS * S;                   // new input rows per step = loads per thread and batch (one pixel lane per column)
  static constexpr int NR = WIN + NEW;                 // ring rows: the window being read + the rows of the next step being written
  static constexpr int WW = (TS - 1) * S + K;          // window columns of a strip
  static_assert(IBW <= 32, "one pixel lane per ring column");
  static_assert(WIN <= 2 * NEW, "head batch + batch 0 cover the window of step 0");
};

// The batch norm (+ swish) in front of the convolution.  gamma == nullptr: none (the input is used as it is).
struct DwmBn {
  BnFold f;            // f.nblk > 0: fold the producer's stage-1 sums, publish mean / rstd, update the moving averages (training);
                       // f.nblk == 0: f.mean / f.rstd are inputs (inference, or the backward pass)
  const float* gamma;
  const float* beta;
};

// Backward kernels with DYBN: the gradient that enters the ring is not read but FORMED while it is staged -- the input gradient of the
// depthwise batch norm (bn1) from the project conv's backward-data output da2 and bn1's input z1 (efficientnet_model.py:271, 238-251):
//     g = (da2 * gate[n] + chan_add[n]) * swish'(u),  u = gamma xhat + beta,     dz1 = gamma rstd (g - mean(g) - xhat mean(g xhat))
// with the two means from the per-image stage-1 sums of mliis_se_mlp_bwd_bn.  The batch norm's apply pass, its write of dz1 and the
// re-read here disappear.  Per channel: u = z sc + sh, dz1 = sc g + z k1 + k0.
struct DwmDyBn {
  const float* z;          // z1 [N,Hd,Wd,C], same shape as the ring-side tensor
  const float *mean, *rstd, *gamma, *beta;
  const float *gate, *chan_add;   // [N][C]
  const float* stage1;     // [nimg][2][C] {sum g, sum g xhat} per image
  int nimg;
  double inv_rows;         // 1 / (N Hd Wd)
  float *dgamma, *dbeta;   // written by workgroup 0 of every channel group
};
struct DwmDyQuad {
  float4 sc, sh, gt, ca, k1, k0;
};
__device__ __forceinline__ DwmDyQuad dwm_dybn_setup(const DwmDyBn& d, int C, int c0, int q, bool cok, int n, bool lead) {
  const int c = cok ? c0 + q * 4 : 0;
  const float4 m = ld4(d.mean + c), rs = ld4(d.rstd + c), ga = ld4(d.gamma + c), be = ld4(d.beta + c);
  DwmDyQuad k;
  k.gt = ld4(d.gate + (long long)n * C + c);
  k.ca = ld4(d.chan_add + (long long)n * C + c);
  double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
  for (int i0 = 0; i0 < d.nimg; i0 += 8) {   // image order (deterministic), eight images' loads together
    float4 u[8], v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int i = i0 + j < d.nimg ? i0 + j : d.nimg - 1;
      u[j] = ld4(d.stage1 + ((long long)i * 2 + 0) * C + c);
      v[j] = ld4(d.stage1 + ((long long)i * 2 + 1) * C + c);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (i0 + j < d.nimg) {
        s1[0] += u[j].x; s1[1] += u[j].y; s1[2] += u[j].z; s1[3] += u[j].w;
        s2[0] += v[j].x; s2[1] += v[j].y; s2[2] += v[j].z; s2[3] += v[j].w;
      }
  }
  if (lead && cok && (threadIdx.x >> 3) == 0) {   // one thread per quad publishes the parameter gradients
    st4(d.dbeta + c, make_float4((float)s1[0], (float)s1[1], (float)s1[2], (float)s1[3]));
    st4(d.dgamma + c, make_float4((float)s2[0], (float)s2[1], (float)s2[2], (float)s2[3]));
  }
  const float4 a = make_float4((float)(s1[0] * d.inv_rows), (float)(s1[1] * d.inv_rows), (float)(s1[2] * d.inv_rows), (float)(s1[3] * d.inv_rows));
  const float4 b = make_float4((float)(s2[0] * d.inv_rows), (float)(s2[1] * d.inv_rows), (float)(s2[2] * d.inv_rows), (float)(s2[3] * d.inv_rows));
  k.sc = f4mul(ga, rs);
  k.sh = make_float4(fmaf(-m.x, k.sc.x, be.x), fmaf(-m.y, k.sc.y, be.y), fmaf(-m.z, k.sc.z, be.z), fmaf(-m.w, k.sc.w, be.w));
  // dz1 = sc (g - a - xhat b),  xhat = (z - m) rs   ->   sc g + z k1 + k0
  k.k1 = make_float4(-k.sc.x * b.x * rs.x, -k.sc.y * b.y * rs.y, -k.sc.z * b.z * rs.z, -k.sc.w * b.w * rs.w);
  k.k0 = make_float4(-k.sc.x * a.x - k.k1.x * m.x, -k.sc.y * a.y - k.k1.y * m.y, -k.sc.z * a.z - k.k1.z * m.z, -k.sc.w * a.w - k.k1.w * m.w);
  return k;
}
__device__ __forceinline__ float dwm_dz1(float da2, float z, float sc, float sh, float gt, float ca, float k1, float k0) {
  const float u = fmaf(z, sc, sh);
  const float sg = sigmoid_f(u);
  const float g = fmaf(da2, gt, ca) * (sg * (1.0f + u * (1.0f - sg)));
  return fmaf(sc, g, fmaf(z, k1, k0));
}
__device__ __forceinline__ float4 dwm_dz1(const float4 da2, const float4 z, const DwmDyQuad& k) {
  return make_float4(dwm_dz1(da2.x, z.x, k.sc.x, k.sh.x, k.gt.x, k.ca.x, k.k1.x, k.k0.x), dwm_dz1(da2.y, z.y, k.sc.y, k.sh.y, k.gt.y, k.ca.y, k.k1.y, k.k0.y),
                     dwm_dz1(da2.z, z.z, k.sc.z, k.sh.z, k.gt.z, k.ca.z, k.k1.z, k.k0.z), dwm_dz1(da2.w, z.w, k.sc.w, k.sh.w, k.gt.w, k.ca.w, k.k1.w, k.k0.w));
}

// one batch of ring rows in registers (DYBN: two tensors)
template <int NEW, bool TWO>
struct DwmBatch {
  float4 a[NEW];
  float4 b[TWO ? NEW : 1];
};

struct DwmArgs {
  const float* x;      // the tensor the ring is fed from: forward [N,Hi,Wi,C] z (or a plain input); backward dy [N,Ho,Wo,C]
  const float* w;      // [K,K,C]
  float* y;            // forward [N,Ho,Wo,C]; backward dx [N,Hi,Wi,C]
  int Hi, Wi, Ho, Wo;  // sizes of the ring-side tensor (Hi, Wi) and of the produced tensor (Ho, Wo)
  int C, pt, pl;       // padding of the correlation this launch computes (backward: K - 1 - pad)
  int bands, bw, chunks, rpc;   // column bands of bw produced columns, row chunks of rpc produced rows
  float* stats_part;   // forward: [gridDim.x][2][C] {sum y, sum y^2} or null; backward: [gridDim.x][2][C] {sum g, sum g xhat} or null
  DwmBn bn;
#ifdef DWM_DBG
  int dbg;             // diagnosis: 1 = no window FMAs, 2 = no global loads, 4 = no stores, 8 = no march (prologue only)
  unsigned long long* stamps;   // [gridDim.x * gridDim.y][16] wall-clock stamps (100 MHz) of wave 0, or null
#endif
  DwmDyBn dyb;         // backward with DYBN: x = da2, dyb.z = z1 -- the ring receives dz1, the depthwise batch norm's input gradient
  const float* z;      // backward: the batch norm's input at the produced positions [N,Ho,Wo,C] (plain input when bn.gamma == nullptr)
  float* dw_part;      // backward: [gridDim.x][K*K][C]
};

// per-channel constants of a thread's quad: a = swish(z * sc + sh), xhat = (z - m) * rs
struct DwmQuad {
  float4 sc, sh, m, rs;
};

__device__ __forceinline__ float4 dwm_act(const float4 z, const DwmQuad& k) {
  float4 u = make_float4(fmaf(z.x, k.sc.x, k.sh.x), fmaf(z.y, k.sc.y, k.sh.y), fmaf(z.z, k.sc.z, k.sh.z), fmaf(z.w, k.sc.w, k.sh.w));
  return make_float4(swish_f(u.x), swish_f(u.y), swish_f(u.z), swish_f(u.w));
}

// The statistics of this workgroup's 32 channels, in two halves around the workgroup's burst of input loads.  Vector-memory
// operations return in issue order: the small loads (gamma, beta, the producer's partial sums or the given mean / rstd) go FIRST, the
// burst behind them, and the fold's arithmetic, LDS exchange and barriers run while the burst is in flight (folded in front of the
// burst they add 2-6 us to every workgroup; issued behind it the parameter loads wait for all of it).
constexpr int kDwmFoldBatch = 8;    // partial blocks per lane and round trip: up to 256 in one (16 would spill: the burst's registers are live too)
template <bool FOLD>    // FOLD = false: the statistics are always given (backward kernels): no fold state in registers
struct DwmBnState {
  float4 g, b, m, rs;
  FoldAcc acc;
  float4 u[FOLD ? kDwmFoldBatch : 1], v[FOLD ? kDwmFoldBatch : 1];
  int k;
};
template <bool FOLD>
__device__ __forceinline__ void dwm_bn_begin(const DwmBn& bn, int C, int c0, int q, bool cok, DwmBnState<FOLD>& st) {
  const int c = cok ? c0 + q * 4 : 0;
  st.g = ld4(bn.gamma + c);
  st.b = ld4(bn.beta + c);
  if (FOLD && bn.f.nblk > 0) {   // (uniform)
    const int bl = threadIdx.x >> 3;
    st.k = bl;
    // all rounds but the last are finished here; the last one's loads stay in flight
    for (; st.k + 32 * kDwmFoldBatch < bn.f.nblk; st.k += 32 * kDwmFoldBatch) {
      fold_issue<FOLD ? kDwmFoldBatch : 1>(bn.f.part, bn.f.nblk, C, c, st.k, bl, st.u, st.v);
      if (cok) fold_add<FOLD ? kDwmFoldBatch : 1>(st.acc, bn.f.nblk, st.k, st.u, st.v);
    }
    fold_issue<FOLD ? kDwmFoldBatch : 1>(bn.f.part, bn.f.nblk, C, c, st.k, st.k < bn.f.nblk ? st.k : 0, st.u, st.v);
  } else {
    st.m = ld4(bn.f.mean + c);
    st.rs = ld4(bn.f.rstd + c);
  }
}
// Workgroup 0 of each channel group (lead) publishes mean / rstd and applies the moving-average update.  smd: >= 16 KB of LDS
// scratch (aliased with the ring, which is not live yet).  Statistics given (f.nblk == 0: inference, backward): no LDS, no barrier.
template <bool FOLD>
__device__ __forceinline__ DwmQuad dwm_bn_end(const DwmBn& bn, int C, int c0, int q, bool cok, bool lead, double* smd, float* s_mr /*[2][32]*/,
                                              DwmBnState<FOLD>& st) {
  const int t = threadIdx.x;
  DwmQuad k;
  if (FOLD && bn.f.nblk > 0) {   // (uniform)
    if (cok) fold_add<FOLD ? kDwmFoldBatch : 1>(st.acc, bn.f.nblk, st.k, st.u, st.v);
    double s, ss;
    fold_finish(st.acc, smd, s, ss);
    if (t < 32) {
      const double m = s * bn.f.inv_n;
      double var = ss * bn.f.inv_n - m * m;
      if (var < 0.0) var = 0.0;
      const float mf = (float)m, rf = (float)(1.0 / sqrt(var + (double)bn.f.eps));
      s_mr[t] = mf;
      s_mr[32 + t] = rf;
      const int cc = c0 + t;
      if (lead && cc < C) {
        bn.f.mean[cc] = mf;
        bn.f.rstd[cc] = rf;
        if (bn.f.moving_mean != nullptr) {
          const float mm = bn.f.moving_mean[cc], mv = bn.f.moving_var[cc];
          bn.f.moving_mean[cc] = mm - (mm - mf) * bn.f.one_minus_momentum;
          bn.f.moving_var[cc] = mv - (mv - (float)(var * (double)bn.f.ema_var_factor)) * bn.f.one_minus_momentum;
        }
      }
    }
    __syncthreads();
    k.m = ld4(s_mr + q * 4);
    k.rs = ld4(s_mr + 32 + q * 4);
    __syncthreads();   // (smd / s_mr reads done before the ring is written)
  } else {
    k.m = st.m;
    k.rs = st.rs;
  }
  k.sc = f4mul(st.g, k.rs);
  k.sh = make_float4(fmaf(-k.m.x, k.sc.x, st.b.x), fmaf(-k.m.y, k.sc.y, st.b.y), fmaf(-k.m.z, k.sc.z, st.b.z), fmaf(-k.m.w, k.sc.w, st.b.w));
  return k;
}

// {s1, s2} per channel quad summed over the 32 pixel lanes of the workgroup -> part[blk][2][C].  Through LDS in two fixed-order
// levels (8 + 4 values): at one wave per SIMD the three rounds of ds_bpermute butterflies cost more than two barriers.
// buf: >= 8 KB of LDS that nothing else uses any more (the ring, behind the loop's last barrier).
__device__ __forceinline__ void dwm_emit_pair(const float4 s1, const float4 s2, float4* buf, float* __restrict__ part, unsigned blk, int C,
                                              int c0) {
  const int t = threadIdx.x;
  buf[t] = s1;            // [pixel lane][quad]
  buf[256 + t] = s2;
  __syncthreads();
  float4 v = f4zero();
  if (t < 64) {           // (value v, group of 8 pixel lanes g, quad qq)
    const int qq = t & 7, g = (t >> 3) & 3, vv = t >> 5;
#pragma unroll
    for (int i = 0; i < 8; ++i) v = f4add(v, buf[vv * 256 + (g * 8 + i) * 8 + qq]);
  }
  __syncthreads();
  if (t < 64) buf[t] = v;
  __syncthreads();
  if (t < 16) {
    const int qq = t & 7, vv = t >> 3;
    const int cc = c0 + qq * 4;
    const float4* b4 = buf + vv * 32 + qq;
    if (cc < C) st4(part + ((long long)blk * 2 + vv) * C + cc, f4add(f4add(b4[0], b4[8]), f4add(b4[16], b4[24])));
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Ring-marching correlation.  BWD (stride 1 only): x = dy, taps reversed, y = the gradient w.r.t. a = swish(bn(z)); each thread also
// loads z at its own produced pixels and accumulates the filter gradient (K*K float4 accumulators) and the batch norm's backward
// sums.  PRE: the batch norm + swish (forward: applied to the staged input; backward: recomputed at the produced pixels).
// ---------------------------------------------------------------------------------------------------------------------------------
// TA / TB: storage types.  Forward: TA = the ring-side input (z, or a plain fp32 input), TB = the output y.  Backward: TA = dy (and, DYBN,
// z1), TB = z at the produced pixels and dx.  float everywhere, or the bf16-storage combinations of `--precision bf16-storage`.
template <int K, int S, bool PRE, bool BWD, bool DYBN = false, typename TA = float, typename TB = float>
__global__ __launch_bounds__(256, BWD ? 1 : 2) void dwm_conv_k(const DwmArgs a) {
  constexpr int EA = (int)sizeof(TA), EB = (int)sizeof(TB);
  static_assert(!DYBN || BWD, "DYBN is a backward mode");
  typedef MarchCfg<K, S> G;
  constexpr int TS = G::TS, SPR = G::SPR, RS = G::RS, IBW = G::IBW, IBWP = G::IBWP, WIN = G::WIN, NEW = G::NEW, NR = G::NR, WW = G::WW;
  static_assert(!BWD || S == 1, "the stride-2 backward is dwm_bwd_s2_k");
  constexpr int kRingF4 = (NR * IBWP + 1) * 8;                     // + one pixel that swallows the writes of surplus lanes
  constexpr int kTrash = NR * IBWP * 8;
  constexpr int kScratchF4 = BWD ? (K * K * 4 * 8) : 0;            // filter-gradient fold: [tap][wave][quad]
  constexpr int kSharedF4 = kRingF4 > kScratchF4 ? kRingF4 : kScratchF4;
  static_assert(kSharedF4 * 16 >= 2 * 32 * 32 * 8, "the statistics fold borrows the ring");
  __shared__ float4 ring[kSharedF4];
  __shared__ float4 wl[K * K * 8];
  __shared__ __attribute__((aligned(16))) float s_mr[64];

#ifdef DWM_DBG
  unsigned long long stamp[16];
  for (int k = 0; k < 16; ++k) stamp[k] = 0;
#endif
  DWM_STAMP(0);
  const int t = threadIdx.x, q = t & 7, p = t >> 3;
  const unsigned bx = xcd_remap(blockIdx.x, gridDim.x);
  const int chunk = (int)(bx % (unsigned)a.chunks);
  const unsigned r_ = bx / (unsigned)a.chunks;
  const int band = (int)(r_ % (unsigned)a.bands), n = (int)(r_ / (unsigned)a.bands);
  const int c0 = blockIdx.y * 32, c = c0 + q * 4;
  const bool cok = c < a.C;
  const int Hi = a.Hi, Wi = a.Wi, Ho = a.Ho, Wo = a.Wo, C = a.C;
  const int oy0 = chunk * a.rpc;
  const int oy1 = oy0 + a.rpc < Ho ? oy0 + a.rpc : Ho;
  const int ox0 = band * a.bw;
  const int bwa = a.bw < Wo - ox0 ? a.bw : Wo - ox0;     // produced columns of this band
  const int iy0 = oy0 * S - a.pt, ix0 = ox0 * S - a.pl;
  const int ibw = (bwa - 1) * S + K;                     // ring columns that are needed
  const int rows_needed = (oy1 - oy0 - 1) * S + K;
  const int nsteps = (oy1 - oy0 + RS - 1) / RS;
  const __amdgpu_buffer_rsrc_t rX = dwm_rsrc(a.x);
  const __amdgpu_buffer_rsrc_t rZ = dwm_rsrc((BWD ? a.z : a.x));
  const __amdgpu_buffer_rsrc_t rY = dwm_rsrc(a.y);

  // ---- staging map: pixel lane p <-> ring column p, load j of a batch <-> its row j.  Everything that depends on the row is
  //      wave-uniform (a scalar offset, a scalar predicate), everything per lane is computed once: three instructions per load
  const bool colok = cok && p < ibw && (unsigned)(ix0 + p) < (unsigned)Wi && !DWM_DBG_ON(a, 2);
  const int tbase = (((n * Hi + iy0) * Wi + ix0 + p) * C + c) * EA;   // byte offset of ring row 0 at this lane's column ("negative": masked)
  const int rowbytes = Wi * C * EA;
  const int lcol = p < IBWP ? (p ^ ((p >> 2) & 1)) * 8 + q : kTrash + q;   // (lanes beyond the pitch write the spare pixel)
  auto row_in = [&](int rr) { return rr >= 0 && rr < rows_needed && (unsigned)(iy0 + rr) < (unsigned)Hi; };
  typedef DwmBatch<NEW, DYBN> Batch;
  const __amdgpu_buffer_rsrc_t rX2 = dwm_rsrc(DYBN ? a.dyb.z : a.x);
  auto issue = [&](Batch& r, int rb) {
#pragma unroll
    for (int j = 0; j < NEW; ++j) {
      r.a[j] = dwm_load<TA>(rX, tbase + (rb + j) * rowbytes, colok && row_in(rb + j));
      if (DYBN) r.b[j] = dwm_load<TA>(rX2, tbase + (rb + j) * rowbytes, colok && row_in(rb + j));
    }
  };
  DwmQuad kq;
  DwmDyQuad dq;
  auto commit = [&](const Batch& r, int rb, int sb) {
#pragma unroll
    for (int j = 0; j < NEW; ++j) {
      float4 v = r.a[j];
      if (PRE && !BWD) v = f4sel(colok && row_in(rb + j), dwm_act(v, kq));   // zero padding applies to the ACTIVATION (swish(bn(0)) != 0)
      if (DYBN) v = f4sel(colok && row_in(rb + j), dwm_dz1(v, r.b[j], dq));   // (dz1 of a pixel outside the map is zero, not k0)
      int slot = sb + j;
      if (slot >= NR) slot -= NR;
      if (rb + j >= 0) ring[(p < IBWP ? slot * (IBWP * 8) : 0) + lcol] = v;   // (uniform branch: only the head batch has rows < 0)
    }
  };

  // ---- produced pixels of this thread
  const int ro = p >> 3, sx = p & 7;
  const bool item_ok = sx < SPR && cok;
  const int colbase = sx < SPR ? sx * (TS * S) : 0;
  const int sxpar = sx & 1;
  const int zbase = BWD ? (((n * Ho + oy0 + ro) * Wo + ox0 + sx * TS) * C + c) * EB : 0;
  auto issue_z = [&](float4 (&zr)[TS], int step) {
    if (BWD) {
      const int oy = oy0 + step * RS + ro;
      const int base = zbase + step * RS * Wo * C * EB;
#pragma unroll
      for (int u = 0; u < TS; ++u) zr[u] = dwm_load<TB>(rZ, base + u * C * EB, item_ok && oy < oy1 && sx * TS + u < bwa);
    }
  };

  // ---- prologue: the window of step 0 = head rows + batch 0.  All four batches are issued before anything is waited for (ONE memory
  //      round trip in front of step 0; the two prologue-only register sets are dead before the compute phase needs its registers)
  constexpr int HEAD = WIN - 2 * NEW;   // ring row of the first row of the head batch (<= 0: its negative rows do not exist)
  // the small loads first (filter taps, batch-norm parameters / statistics): loads return in issue order
  float4 wreg = f4zero();
  if (t < K * K * 8) {
    const int tap = t >> 3, qq = t & 7;
    const int cc = c0 + qq * 4;
    // backward: tap (ky, kx) of the correlation with dy is w[K-1-ky][K-1-kx]
    if (cc < C) wreg = ld4(a.w + (long long)(BWD ? K * K - 1 - tap : tap) * C + cc);
  }
  DwmBnState<!BWD> bst;
  if (PRE) dwm_bn_begin(a.bn, C, c0, q, cok, bst);
  if (DYBN) dq = dwm_dybn_setup(a.dyb, C, c0, q, cok, n, bx == 0);
  DWM_STAMP(1);
  Batch ra, rb_;
  float4 za[TS], zb[TS];
  {
    // the window of step 0 (head rows + batch 0) first; the fold's arithmetic runs under their transfer; then the batches of steps
    // 1 and 2 (needed at the end of steps 0 and 1)
    Batch h0, h1;
    issue(h0, HEAD);
    issue(h1, WIN - NEW);
    issue_z(za, 0);
    if (PRE) kq = dwm_bn_end(a.bn, C, c0, q, cok, bx == 0, reinterpret_cast<double*>(ring), s_mr, bst);
    issue(ra, WIN);             // step 1
    issue(rb_, WIN + NEW);      // step 2
    issue_z(zb, 1);
    DWM_STAMP(2);
    if (t < K * K * 8) wl[t] = wreg;
    commit(h0, HEAD, HEAD < 0 ? HEAD + NR : HEAD);
    commit(h1, WIN - NEW, (WIN - NEW) % NR);
  }
  DWM_STAMP(3);
  __syncthreads();            // (also publishes wl)
  DWM_STAMP(4);

  float4 s1 = f4zero(), s2 = f4zero();
  float4 dwacc[BWD ? K * K : 1];
  if (BWD) {
#pragma unroll
    for (int k = 0; k < K * K; ++k) dwacc[k] = f4zero();
  }

  int sw = 0;   // ring slot of the first window row of the current step
  auto compute = [&](int step, const float4 (&zr)[TS]) {
    const int oy = oy0 + step * RS + ro;
    const bool row_ok = item_ok && oy < oy1;
    int s0 = sw + ro * S;
    if (s0 >= NR) s0 -= NR;
    float4 acc[TS];
#pragma unroll
    for (int u = 0; u < TS; ++u) acc[u] = f4zero();
    // 5x5: the taps are re-read from LDS every step (an opaque base keeps the compiler from holding all 25 float4 in registers across
    // the march); 3x3: loop-invariant, they stay in registers
    int wofs = q;
    if (K == 5) asm volatile("" : "+v"(wofs));
    const float4* wq = wl + wofs;
    // backward: the activation at this thread's pixels
    // (only the activation and the sigmoid stay live under the window loop; the derivative and xhat are re-derived from z after it)
    float4 av[TS], sgm[TS];
    if (BWD) {
#pragma unroll
      for (int u = 0; u < TS; ++u) {
        const bool ok = row_ok && sx * TS + u < bwa;
        if (PRE) {
          const float4 z = zr[u];
          const float4 uu = make_float4(fmaf(z.x, kq.sc.x, kq.sh.x), fmaf(z.y, kq.sc.y, kq.sh.y), fmaf(z.z, kq.sc.z, kq.sh.z), fmaf(z.w, kq.sc.w, kq.sh.w));
          sgm[u] = f4sel(ok, make_float4(sigmoid_f(uu.x), sigmoid_f(uu.y), sigmoid_f(uu.z), sigmoid_f(uu.w)));
          av[u] = f4mul(uu, sgm[u]);
        } else {
          av[u] = f4sel(ok, zr[u]);
        }
      }
    }
    // window row ky of this strip -> registers (K * WW ds_read_b128 per step and thread)
    auto read_row = [&](int ky, float4 (&in)[WW]) {
      int slot = s0 + ky;
      if (slot >= NR) slot -= NR;
      if constexpr ((TS * S) % 4 == 0) {
        const float4* rowp = ring + slot * (IBWP * 8) + colbase * 8 + q;
        const float4* A0 = rowp + sxpar * 8;   // columns whose swizzled index is even: je ^ sxpar = je + sxpar
        const float4* A1 = rowp - sxpar * 8;   // odd: je - sxpar
#pragma unroll
        for (int j = 0; j < WW; ++j) {
          const int je = j ^ ((j >> 2) & 1);
          in[j] = ((je & 1) ? A1 : A0)[je * 8];
        }
      } else {   // strips that do not start on a multiple of four columns: the swizzled column of every window element, computed per thread
        const float4* rowp = ring + slot * (IBWP * 8) + q;
#pragma unroll
        for (int j = 0; j < WW; ++j) {
          const int col = colbase + j;
          in[j] = rowp[(col ^ ((col >> 2) & 1)) * 8];
        }
      }
    };
    auto fma_row = [&](int ky, const float4 (&in)[WW]) {
      float4 wk[K];
#pragma unroll
      for (int kx = 0; kx < K; ++kx) wk[kx] = wq[(ky * K + kx) * 8];
#pragma unroll
      for (int kx = 0; kx < K; ++kx)
#pragma unroll
        for (int u = 0; u < TS; ++u) {
          acc[u] = f4fma(in[u * S + kx], wk[kx], acc[u]);
          if (BWD) dwacc[ky * K + kx] = f4fma(in[u + kx], av[u], dwacc[ky * K + kx]);
        }
    };
    if (!DWM_DBG_ON(a, 1)) {
      if (K == 3) {
        // 3x3: the scheduler may hoist all three rows' reads above the FMAs (18-27 float4: one LDS latency instead of three)
        float4 in[WW];
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
          read_row(ky, in);
          fma_row(ky, in);
        }
      } else {
        // 5x5: two rows in registers -- the reads of row ky + 1 are in flight under the FMAs of row ky; a scheduling fence per row
        // keeps the compiler from hoisting all K rows (K * WW float4 = 160-220 registers)
        float4 inA[WW], inB[WW];
        read_row(0, inA);
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
          if (ky + 1 < K) read_row(ky + 1, (ky & 1) ? inA : inB);
          fma_row(ky, (ky & 1) ? inB : inA);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    // stores and sums without a branch (a conditional block here makes the compiler sink the FMAs of each output into it and keep the
    // whole K x WW window live): masked buffer stores, masked adds
    const int ybase = (((n * Ho + oy) * Wo + ox0 + sx * TS) * C + c) * EB;
    if (K == 5 && EB == 2) __builtin_amdgcn_sched_barrier(0);   // (the bf16 conversions must not be interleaved with the last window rows: spills)
#pragma unroll
    for (int u = 0; u < TS; ++u) {
      const bool ok = row_ok && sx * TS + u < bwa;
      if (!(K == 5 && !BWD)) acc[u] = dwm_stored<TB>(acc[u]);   // (bf16 storage: the sums below see what the consumers will read back)
      dwm_store<TB>(rY, ybase + u * C * EB, ok && !DWM_DBG_ON(a, 4), acc[u]);
      if (!BWD) {
        const float4 v = f4sel(ok, acc[u]);
        s1 = f4add(s1, v);
        s2 = f4fma(v, v, s2);
      } else if (PRE) {
        const float4 z = zr[u], sg_ = sgm[u];    // (the sigmoid is zero outside the band: so is g)
        const float4 uu = make_float4(fmaf(z.x, kq.sc.x, kq.sh.x), fmaf(z.y, kq.sc.y, kq.sh.y), fmaf(z.z, kq.sc.z, kq.sh.z), fmaf(z.w, kq.sc.w, kq.sh.w));
        const float4 g = make_float4(acc[u].x * sg_.x * (1.0f + uu.x * (1.0f - sg_.x)), acc[u].y * sg_.y * (1.0f + uu.y * (1.0f - sg_.y)),
                                     acc[u].z * sg_.z * (1.0f + uu.z * (1.0f - sg_.z)), acc[u].w * sg_.w * (1.0f + uu.w * (1.0f - sg_.w)));
        const float4 xh = make_float4((z.x - kq.m.x) * kq.rs.x, (z.y - kq.m.y) * kq.rs.y, (z.z - kq.m.z) * kq.rs.z, (z.w - kq.m.w) * kq.rs.w);
        s1 = f4add(s1, g);
        s2 = f4fma(g, xh, s2);
      }
    }
  };

  // ---- march: step i reads its window, then the rows of step i + 1 (in flight since step i - 1) go into the ring and the loads of
  //      step i + 3 are issued; two register sets alternate, so one batch is always in flight under the compute phase
  int rbn = WIN;          // ring row of the first row of the batch for step i + 1
  int sbn = WIN % NR;     // its ring slot
  for (int i = 0; i < (DWM_DBG_ON(a, 8) ? 0 : nsteps); i += 2) {
    compute(i, za);
    if (i == 0) DWM_STAMP(5);
    commit(ra, rbn, sbn);
    if (i == 0) DWM_STAMP(6);
    issue(ra, rbn + 2 * NEW);
    issue_z(za, i + 2);
    rbn += NEW; sbn += NEW; if (sbn >= NR) sbn -= NR;
    sw += NEW; if (sw >= NR) sw -= NR;
    __syncthreads();
    if (i == 0) DWM_STAMP(7);
    if (i + 1 < nsteps) compute(i + 1, zb);
    if (i == 0) DWM_STAMP(8);
    commit(rb_, rbn, sbn);
    issue(rb_, rbn + 2 * NEW);
    issue_z(zb, i + 3);
    rbn += NEW; sbn += NEW; if (sbn >= NR) sbn -= NR;
    sw += NEW; if (sw >= NR) sw -= NR;
    __syncthreads();
  }

  DWM_STAMP(9);
  if (a.stats_part != nullptr && (!BWD || PRE)) dwm_emit_pair(s1, s2, ring, a.stats_part, bx, C, c0);
  DWM_STAMP(10);
#ifdef DWM_DBG
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  DWM_STAMP(11);
  if (a.stamps != nullptr && t == 0)
    for (int k = 0; k < 16; ++k) a.stamps[((long long)blockIdx.y * gridDim.x + blockIdx.x) * 16 + k] = stamp[k];
#endif
  if (BWD) {
    // filter gradient of this workgroup: butterfly over the 8 pixel lanes of a wave that share a quad, the 4 waves through LDS
    __syncthreads();       // (the statistics fold above used the same LDS)
    float4* fold = ring;
#pragma unroll
    for (int k = 0; k < K * K; ++k) {
      float4 v = dwacc[k];
#pragma unroll
      for (int off = 8; off < 64; off <<= 1) {
        v.x += __shfl_xor(v.x, off); v.y += __shfl_xor(v.y, off); v.z += __shfl_xor(v.z, off); v.w += __shfl_xor(v.w, off);
      }
      if ((t & 63) < 8) fold[(k * 4 + (t >> 6)) * 8 + q] = v;
    }
    __syncthreads();
    for (int e = t; e < K * K * 8; e += 256) {
      const int k = e >> 3, qq = e & 7;
      const int cc = c0 + qq * 4;
      // accumulator (ky, kx) of the flipped correlation is the gradient of tap (K-1-ky, K-1-kx)
      if (cc < C)
        st4(a.dw_part + ((long long)bx * (K * K) + (K * K - 1 - k)) * C + cc,
            f4add(f4add(fold[(k * 4 + 0) * 8 + qq], fold[(k * 4 + 1) * 8 + qq]), f4add(fold[(k * 4 + 2) * 8 + qq], fold[(k * 4 + 3) * 8 + qq])));
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Backward of a stride-2 layer.  Work is tiled over the INPUT in 2 x 2 patches aligned in padded coordinates (hy = hi + pt,
// wx = wi + pl): pixel (a, b) of patch (m, mx) takes exactly the taps ky = a (mod 2), kx = b (mod 2), from dy rows m - (ky - a) / 2 --
// every thread does the same 9 (3x3) / 25 (5x5) multiply-adds per patch, nothing is multiplied by an interleaved zero.  The ring
// holds dy rows (1/4 of the input's size); z is read and dx written by the owning thread directly.
// Pixel lane -> (pair row rp = p >> 4 of the step, patch px = p & 15): a step = 2 pair rows = 4 input rows.
// ---------------------------------------------------------------------------------------------------------------------------------
template <int K, bool PRE, bool DYBN = false, typename TA = float, typename TB = float>   // TA: dy (and z1), TB: z and dx (see dwm_conv_k)
__global__ __launch_bounds__(256, (K == 5 || DYBN) ? 1 : 2) void dwm_bwd_s2_k(const DwmArgs a) {
  constexpr int EA = (int)sizeof(TA), EB = (int)sizeof(TB);
  constexpr int BP = 14;                    // patches per band (28 input columns, as the stride-1 kernels)
  constexpr int D = (K - 1) / 2;            // dy rows / columns behind a patch
  constexpr int NEWD = 2, WIND = 2 + D, NR = WIND + NEWD, IBW = 16;   // ring row = 16 dy pixels (BP + D needed): one batch = 2 rows = the 32 pixel lanes
  static_assert(BP + D <= IBW, "one pixel lane per ring pixel of a batch");
  constexpr int kRingF4 = NR * IBW * 8;
  constexpr int kScratchF4 = K * K * 4 * 8;
  constexpr int kSharedF4 = (kRingF4 > kScratchF4 ? kRingF4 : kScratchF4) > 1024 ? (kRingF4 > kScratchF4 ? kRingF4 : kScratchF4) : 1024;
  __shared__ float4 ring[kSharedF4];
  __shared__ float4 wl[K * K * 8];
  __shared__ __attribute__((aligned(16))) float s_mr[64];

  const int t = threadIdx.x, q = t & 7, p = t >> 3;
  const unsigned bx = xcd_remap(blockIdx.x, gridDim.x);
  const int chunk = (int)(bx % (unsigned)a.chunks);
  const unsigned r_ = bx / (unsigned)a.chunks;
  const int band = (int)(r_ % (unsigned)a.bands), n = (int)(r_ / (unsigned)a.bands);
  const int c0 = blockIdx.y * 32, c = c0 + q * 4;
  const bool cok = c < a.C;
  // here: (Hi, Wi) = dy (ring side), (Ho, Wo) = the layer's input (produced side: dx, z); pt / pl = the layer's forward padding
  const int Hd = a.Hi, Wd = a.Wi, H = a.Ho, W = a.Wo, C = a.C, pt = a.pt, pl = a.pl;
  const int my_lo = pt >> 1, my_hi = (H + pt - 1) >> 1;       // pair rows that hold an input row
  const int mx_lo = pl >> 1, mx_hi = (W + pl - 1) >> 1;
  const int m0 = my_lo + chunk * a.rpc;                        // (rpc, bw in pairs)
  const int m1 = m0 + a.rpc <= my_hi + 1 ? m0 + a.rpc : my_hi + 1;
  const int x0 = mx_lo + band * a.bw;
  const int bpa = a.bw < mx_hi + 1 - x0 ? a.bw : mx_hi + 1 - x0;   // patches of this band
  const int nsteps = (m1 - m0 + 1) / 2;
  const int rows_needed = m1 - m0 + D;                         // dy rows m0 - D .. m1 - 1
  const __amdgpu_buffer_rsrc_t rX = dwm_rsrc(a.x);
  const __amdgpu_buffer_rsrc_t rZ = dwm_rsrc(a.z);
  const __amdgpu_buffer_rsrc_t rY = dwm_rsrc(a.y);

  // ---- staging map: a batch = 2 dy rows x 16 columns = one pixel per lane (row p >> 4, column p & 15); ring row r <-> dy row
  //      m0 - D + r, column col <-> dy column x0 - D + col
  const int sl = p >> 4, scol = p & 15;
  const bool colok = cok && scol < bpa + D && (unsigned)(x0 - D + scol) < (unsigned)Wd;
  const int tbase = (((n * Hd + m0 - D + sl) * Wd + x0 - D + scol) * C + c) * EA;
  const int rowbytes = Wd * C * EA;
  typedef DwmBatch<1, DYBN> Batch;
  const __amdgpu_buffer_rsrc_t rX2 = dwm_rsrc(DYBN ? a.dyb.z : a.x);
  auto row_in = [&](int rr) { return rr >= 0 && rr < rows_needed && (unsigned)(m0 - D + rr) < (unsigned)Hd; };
  auto issue = [&](Batch& r, int rb) {
    r.a[0] = dwm_load<TA>(rX, tbase + rb * rowbytes, colok && row_in(rb + sl));
    if (DYBN) r.b[0] = dwm_load<TA>(rX2, tbase + rb * rowbytes, colok && row_in(rb + sl));
  };
  DwmDyQuad dq;
  auto commit = [&](const Batch& r, int rb, int sb) {
    int slot = sb + sl;
    if (slot >= NR) slot -= NR;
    float4 v = r.a[0];
    if (DYBN) v = f4sel(colok && row_in(rb + sl), dwm_dz1(v, r.b[0], dq));
    if (rb + sl >= 0) ring[slot * (IBW * 8) + scol * 8 + q] = v;
  };

  // ---- this thread's patch: pair row m0 + 2 step + rp, pair column x0 + px; pixel (u >> 1, u & 1)
  const int rp = p >> 4, px = p & 15;
  const bool item_ok = px < bpa && cok;
  const int pxc = px < BP ? px : 0;
  const int wi0 = 2 * (x0 + px) - pl;
  auto pix_ok = [&](int step, int u) {
    const int hi = 2 * (m0 + 2 * step + rp) + (u >> 1) - pt, wi = wi0 + (u & 1);
    return item_ok && m0 + 2 * step + rp < m1 && (unsigned)hi < (unsigned)H && (unsigned)wi < (unsigned)W;
  };
  auto pix_off = [&](int step, int u) {
    const int hi = 2 * (m0 + 2 * step + rp) + (u >> 1) - pt, wi = wi0 + (u & 1);
    return (((n * H + hi) * W + wi) * C + c) * EB;
  };
  auto issue_z = [&](float4 (&zr)[4], int step) {
#pragma unroll
    for (int u = 0; u < 4; ++u) zr[u] = dwm_load<TB>(rZ, pix_off(step, u), pix_ok(step, u));
  };

  constexpr int HEAD = WIND - 2 * NEWD;   // = D - 2 <= 0
  float4 wreg = f4zero();
  if (t < K * K * 8) {
    const int tap = t >> 3, qq = t & 7;
    const int cc = c0 + qq * 4;
    if (cc < C) wreg = ld4(a.w + (long long)tap * C + cc);
  }
  DwmQuad kq;
  DwmBnState<false> bst;
  if (PRE) dwm_bn_begin(a.bn, C, c0, q, cok, bst);
  if (PRE) kq = dwm_bn_end(a.bn, C, c0, q, cok, false, reinterpret_cast<double*>(ring), s_mr, bst);   // (statistics are given here: no fold)
  if (DYBN) dq = dwm_dybn_setup(a.dyb, C, c0, q, cok, n, bx == 0);
  Batch ra, rb_, h0, h1;
  float4 za[4], zb[4];
  issue(h0, HEAD);
  issue(h1, WIND - NEWD);
  issue_z(za, 0);
  issue(ra, WIND);
  issue(rb_, WIND + NEWD);
  issue_z(zb, 1);
  if (t < K * K * 8) wl[t] = wreg;
  commit(h0, HEAD, HEAD < 0 ? HEAD + NR : HEAD);
  commit(h1, WIND - NEWD, (WIND - NEWD) % NR);
  __syncthreads();

  float4 s1 = f4zero(), s2 = f4zero();
  float4 dwacc[K * K];
#pragma unroll
  for (int k = 0; k < K * K; ++k) dwacc[k] = f4zero();

  int sw = 0;
  auto compute = [&](int step, const float4 (&zr)[4]) {
    float4 av[4], sgm[4], acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const bool ok = pix_ok(step, u);
      acc[u] = f4zero();
      if (PRE) {
        const float4 z = zr[u];
        const float4 uu = make_float4(fmaf(z.x, kq.sc.x, kq.sh.x), fmaf(z.y, kq.sc.y, kq.sh.y), fmaf(z.z, kq.sc.z, kq.sh.z), fmaf(z.w, kq.sc.w, kq.sh.w));
        sgm[u] = f4sel(ok, make_float4(sigmoid_f(uu.x), sigmoid_f(uu.y), sigmoid_f(uu.z), sigmoid_f(uu.w)));
        av[u] = f4mul(uu, sgm[u]);
      } else {
        av[u] = f4sel(ok, zr[u]);
      }
    }
    int s0 = sw + rp;
    if (s0 >= NR) s0 -= NR;
    int wofs = q;
    if (K == 5) asm volatile("" : "+v"(wofs));   // (5x5: taps re-read from LDS every step, see dwm_conv_k)
    const float4* wq = wl + wofs;
    // window row wr <-> dy row m - (D - wr): taps ky = pa + 2 (D - wr); window column wc likewise
#pragma unroll
    for (int wr = 0; wr <= D; ++wr) {
      int slot = s0 + wr;
      if (slot >= NR) slot -= NR;
      const float4* rowp = ring + slot * (IBW * 8) + pxc * 8 + q;
      float4 dv[D + 1];
#pragma unroll
      for (int wc = 0; wc <= D; ++wc) dv[wc] = rowp[wc * 8];
#pragma unroll
      for (int pa = 0; pa < 2; ++pa) {
        const int ky = pa + 2 * (D - wr);
        if (ky > K - 1) continue;
#pragma unroll
        for (int wc = 0; wc <= D; ++wc)
#pragma unroll
          for (int pb = 0; pb < 2; ++pb) {
            const int kx = pb + 2 * (D - wc);
            if (kx > K - 1) continue;
            const int u = pa * 2 + pb, tap = ky * K + kx;
            acc[u] = f4fma(dv[wc], wq[tap * 8], acc[u]);
            dwacc[tap] = f4fma(dv[wc], av[u], dwacc[tap]);
          }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc[u] = dwm_stored<TB>(acc[u]);
      dwm_store<TB>(rY, pix_off(step, u), pix_ok(step, u), acc[u]);
      if (PRE) {
        const float4 z = zr[u], sg_ = sgm[u];    // (the sigmoid is zero outside the image: so is g)
        const float4 uu = make_float4(fmaf(z.x, kq.sc.x, kq.sh.x), fmaf(z.y, kq.sc.y, kq.sh.y), fmaf(z.z, kq.sc.z, kq.sh.z), fmaf(z.w, kq.sc.w, kq.sh.w));
        const float4 g = make_float4(acc[u].x * sg_.x * (1.0f + uu.x * (1.0f - sg_.x)), acc[u].y * sg_.y * (1.0f + uu.y * (1.0f - sg_.y)),
                                     acc[u].z * sg_.z * (1.0f + uu.z * (1.0f - sg_.z)), acc[u].w * sg_.w * (1.0f + uu.w * (1.0f - sg_.w)));
        const float4 xh = make_float4((z.x - kq.m.x) * kq.rs.x, (z.y - kq.m.y) * kq.rs.y, (z.z - kq.m.z) * kq.rs.z, (z.w - kq.m.w) * kq.rs.w);
        s1 = f4add(s1, g);
        s2 = f4fma(g, xh, s2);
      }
    }
  };

  int rbn = WIND, sbn = WIND % NR;
  for (int i = 0; i < nsteps; i += 2) {
    compute(i, za);
    commit(ra, rbn, sbn);
    issue(ra, rbn + 2 * NEWD);
    issue_z(za, i + 2);
    rbn += NEWD; sbn += NEWD; if (sbn >= NR) sbn -= NR;
    sw += NEWD; if (sw >= NR) sw -= NR;
    __syncthreads();
    if (i + 1 < nsteps) compute(i + 1, zb);
    commit(rb_, rbn, sbn);
    issue(rb_, rbn + 2 * NEWD);
    issue_z(zb, i + 3);
    rbn += NEWD; sbn += NEWD; if (sbn >= NR) sbn -= NR;
    sw += NEWD; if (sw >= NR) sw -= NR;
    __syncthreads();
  }

  if (a.stats_part != nullptr && PRE) dwm_emit_pair(s1, s2, ring, a.stats_part, bx, C, c0);
  __syncthreads();       // (the statistics fold used the same LDS)
  float4* fold = ring;
#pragma unroll
  for (int k = 0; k < K * K; ++k) {
    float4 v = dwacc[k];
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) {
      v.x += __shfl_xor(v.x, off); v.y += __shfl_xor(v.y, off); v.z += __shfl_xor(v.z, off); v.w += __shfl_xor(v.w, off);
    }
    if ((t & 63) < 8) fold[(k * 4 + (t >> 6)) * 8 + q] = v;
  }
  __syncthreads();
  for (int e = t; e < K * K * 8; e += 256) {
    const int k = e >> 3, qq = e & 7;
    const int cc = c0 + qq * 4;
    if (cc < C)
      st4(a.dw_part + ((long long)bx * (K * K) + k) * C + cc,
          f4add(f4add(fold[(k * 4 + 0) * 8 + qq], fold[(k * 4 + 1) * 8 + qq]), f4add(fold[(k * 4 + 2) * 8 + qq], fold[(k * 4 + 3) * 8 + qq])));
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------------------
struct MarchGeom {
  int bands, bw, chunks, rpc;
  int Ho, Wo, pt, pl;
  long long gx;
  int gy;
};

static inline void same_pad(int H, int K, int S, int* Ho, int* pt) {
  *Ho = (H + S - 1) / S;
  int th = (*Ho - 1) * S + K - H;
  if (th < 0) th = 0;
  *pt = th / 2;
}

// Workgroups per launch: the maps of one inner step are small (7-48 MB), so the row ranges are cut until about two workgroups per
// CU exist in the forward (two waves per SIMD) and one in the backward (its kernels hold one wave per SIMD) -- shorter ranges
// re-fetch the K - S seam rows more often (rows_per_chunk + K - S over rows_per_chunk) and pay the prologue more often; measured
// cold at N = 8 and N = 64 (tools/bench_dwmarch.py, MLIIS_DWM_TARGET): forward 256 / 512 equal, 1024+ slower; backward 256 best.
static int g_dwm_target = -1;
static inline int dwm_target(bool bwd) {
  if (g_dwm_target < 0) {
    const char* e = getenv("MLIIS_DWM_TARGET");
    g_dwm_target = e ? atoi(e) : 0;
    if (g_dwm_target < 0) g_dwm_target = 0;
  }
  // (forward, cold, op by op: 256 and 512 time the same.  In the step -- round 6, same box, three alternations, the consumer's fold of
  //  the partial blocks included: 192: 3528, 256: 3537-3540, 320: 3534, 384: 3538, 448: 3545-3547, 512: 3532-3543, 640+: 3530-3535
  //  images/s; backward 192 / 224 / 256 / 320 / 512: 3507 / 3511 / 3537 / 3497 / 3506.)
  return g_dwm_target ? g_dwm_target : (bwd ? 256 : 448);
}

// produced extent P x Q (rows x columns); unit = produced rows per step; bmax = produced columns of a band (multiple of ts)
static inline void march_split(int N, int cgs, int P, int Q, int unit, int bmax, int ts, bool bwd, MarchGeom* g) {
  g->bands = (Q + bmax - 1) / bmax;
  int bw = (Q + g->bands - 1) / g->bands;   // balanced bands
  bw = (bw + ts - 1) / ts * ts;
  g->bw = bw;
  g->bands = (Q + bw - 1) / bw;
  const long long base = (long long)N * cgs * g->bands;
  long long chunks = dwm_target(bwd) / base;
  const int max_chunks = (P + 2 * unit - 1) / (2 * unit);   // at least two steps per chunk
  if (chunks > max_chunks) chunks = max_chunks;
  if (chunks < 1) chunks = 1;
  int rpc = (int)((P + chunks - 1) / chunks);
  rpc = (rpc + unit - 1) / unit * unit;
  g->rpc = rpc;
  g->chunks = (P + rpc - 1) / rpc;
  g->gx = (long long)N * g->bands * g->chunks;
  g->gy = cgs;
}

static inline MarchGeom march_geom_fwd(int N, int H, int W, int C, int K, int S) {
  MarchGeom g;
  same_pad(H, K, S, &g.Ho, &g.pt);
  same_pad(W, K, S, &g.Wo, &g.pl);
  if (S == 1 && !(K == 5 && DWM_K5_TS2)) march_split(N, ceil_div(C, 32), g.Ho, g.Wo, 4, 28, 4, false, &g);
  else march_split(N, ceil_div(C, 32), g.Ho, g.Wo, 4, 14, 2, false, &g);
  return g;
}

// backward: tiles over the layer's INPUT (H x W); stride 2 in 2 x 2 patches of padded coordinates
static inline MarchGeom march_geom_bwd(int N, int H, int W, int C, int K, int S) {
  MarchGeom g;
  same_pad(H, K, S, &g.Ho, &g.pt);
  same_pad(W, K, S, &g.Wo, &g.pl);
  if (S == 1) {
    if (K == 5 && DWM_K5_TS2) march_split(N, ceil_div(C, 32), H, W, 4, 14, 2, true, &g);
    else march_split(N, ceil_div(C, 32), H, W, 4, 28, 4, true, &g);
  } else {
    const int py = ((H + g.pt - 1) >> 1) - (g.pt >> 1) + 1, pxn = ((W + g.pl - 1) >> 1) - (g.pl >> 1) + 1;
    march_split(N, ceil_div(C, 32), py, pxn, 2, 14, 1, true, &g);
  }
  return g;
}

static inline DwmBn make_bn(const float* part, int nblk, long long count, const float* gamma, const float* beta, float* mean, float* rstd,
                            float* moving_mean, float* moving_var, float eps, float momentum, int unbiased) {
  DwmBn b;
  b.f.part = part;
  b.f.nblk = nblk;
  b.f.inv_n = 1.0 / (double)count;
  b.f.eps = eps;
  b.f.one_minus_momentum = 1.0f - momentum;
  b.f.ema_var_factor = (unbiased && count > 1) ? (float)((double)count / (double)(count - 1)) : 1.0f;
  b.f.mean = mean;
  b.f.rstd = rstd;
  b.f.moving_mean = moving_mean;
  b.f.moving_var = moving_var;
  b.gamma = gamma;
  b.beta = beta;
  return b;
}

// storage-type combination (ta, tb) of a launch (MLIIS_DT_*): forward (ring input, output) in {(f32, f32), (f32, bf16), (bf16, bf16)};
// backward (dy [and z1], z and dx) in {(f32, f32), (bf16, f32), (bf16, bf16)} -- the fp32 sides are block 0 behind the stem and the
// no-expand blocks, whose input / input gradient are fp32 block tensors
static inline bool dwm_types_ok(bool bwd, int ta, int tb) {
  if (ta == MLIIS_DT_F32 && tb == MLIIS_DT_F32) return true;
  if (ta == MLIIS_DT_BF16 && tb == MLIIS_DT_BF16) return true;
  return bwd ? (ta == MLIIS_DT_BF16 && tb == MLIIS_DT_F32) : (ta == MLIIS_DT_F32 && tb == MLIIS_DT_BF16);
}
template <int K, int S, bool PRE, bool BWD>
static void launch_conv(const MarchGeom& g, const DwmArgs& a, int ta, int tb, hipStream_t stream) {
  const dim3 grid((unsigned)g.gx, g.gy);
  if (ta == MLIIS_DT_F32 && tb == MLIIS_DT_F32) hipLaunchKernelGGL((dwm_conv_k<K, S, PRE, BWD, false, float, float>), grid, dim3(256), 0, stream, a);
  else if (ta == MLIIS_DT_BF16 && tb == MLIIS_DT_BF16) hipLaunchKernelGGL((dwm_conv_k<K, S, PRE, BWD, false, bf16s, bf16s>), grid, dim3(256), 0, stream, a);
  else if constexpr (BWD) hipLaunchKernelGGL((dwm_conv_k<K, S, PRE, BWD, false, bf16s, float>), grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((dwm_conv_k<K, S, PRE, BWD, false, float, bf16s>), grid, dim3(256), 0, stream, a);
}
template <int K, bool PRE>
static void launch_bwd_s2(const MarchGeom& g, const DwmArgs& a, int ta, int tb, hipStream_t stream) {
  const dim3 grid((unsigned)g.gx, g.gy);
  if (ta == MLIIS_DT_F32) hipLaunchKernelGGL((dwm_bwd_s2_k<K, PRE, false, float, float>), grid, dim3(256), 0, stream, a);
  else if (tb == MLIIS_DT_BF16) hipLaunchKernelGGL((dwm_bwd_s2_k<K, PRE, false, bf16s, bf16s>), grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((dwm_bwd_s2_k<K, PRE, false, bf16s, float>), grid, dim3(256), 0, stream, a);
}
// (DYBN: always with the batch norm in front -- the MBConv blocks with an expand conv, or block 0 behind the stem)
template <int K, typename TA, typename TB>
static void launch_bwd_dybn_t(const MarchGeom& g, const DwmArgs& a, int stride, hipStream_t stream) {
  const dim3 grid((unsigned)g.gx, g.gy);
  if (stride == 1) hipLaunchKernelGGL((dwm_conv_k<K, 1, true, true, true, TA, TB>), grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((dwm_bwd_s2_k<K, true, true, TA, TB>), grid, dim3(256), 0, stream, a);
}
template <int K>
static void launch_bwd_dybn(const MarchGeom& g, const DwmArgs& a, int stride, int ta, int tb, hipStream_t stream) {
  if (ta == MLIIS_DT_F32) launch_bwd_dybn_t<K, float, float>(g, a, stride, stream);
  else if (tb == MLIIS_DT_BF16) launch_bwd_dybn_t<K, bf16s, bf16s>(g, a, stride, stream);
  else launch_bwd_dybn_t<K, bf16s, float>(g, a, stride, stream);
}

}  // namespace mliis

using namespace mliis;

static int dwm_check(const char* name, int N, int H, int W, int C, int k, int stride) {
  MLIIS_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0, MLIIS_ERR_ARG, "%s: bad shape N=%d H=%d W=%d C=%d (C %% 4 must be 0)", name, N, H,
                W, C);
  MLIIS_REQUIRE((k == 3 || k == 5) && (stride == 1 || stride == 2), MLIIS_ERR_UNSUPPORTED, "%s: unsupported k=%d stride=%d", name, k, stride);
  MLIIS_REQUIRE((long long)N * H * W * C * 4 < (1LL << 31), MLIIS_ERR_UNSUPPORTED, "%s: tensor larger than 2 GiB (32-bit buffer offsets)", name);
  return MLIIS_OK;
}

extern "C" {

// 1 when the row-marching entry points take this layer (the planner's query: unsupported shapes run op by op through dwconv.hip).
int mliis_dwconv_bn_supported(int N, int H, int W, int C, int k, int stride) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || (k != 3 && k != 5) || (stride != 1 && stride != 2)) return 0;
  return ((long long)N * H * W * C * 4 < (1LL << 31)) ? 1 : 0;
}

// Number of workgroups along x of mliis_dwconv_bn_fwd (= blocks of its stats_part output).
int mliis_dwconv_bn_fwd_blocks(int N, int H, int W, int C, int k, int stride) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || (k != 3 && k != 5) || (stride != 1 && stride != 2)) return 0;
  return (int)march_geom_fwd(N, H, W, C, k, stride).gx;
}

// Number of workgroups along x of mliis_dwconv_bn_bwd (= slabs of dw_part, blocks of bn_part).
int mliis_dwconv_bn_bwd_blocks(int N, int H, int W, int C, int k, int stride) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || (k != 3 && k != 5) || (stride != 1 && stride != 2)) return 0;
  return (int)march_geom_bwd(N, H, W, C, k, stride).gx;
}

int mliis_dwconv_bn_fwd(const float* z, const float* bn_part, int bn_nblk, const float* bn_gamma, const float* bn_beta, float* bn_mean,
                        float* bn_rstd, float* bn_moving_mean, float* bn_moving_var, float eps, float momentum, const float* w, float* y,
                        int N, int H, int W, int C, int k, int stride, float* stats_part, size_t stats_floats, int* stats_nblk,
                        int in_dtype, int out_dtype, hipStream_t stream) {
  int rc = dwm_check("dwconv_bn_fwd", N, H, W, C, k, stride);
  if (rc) return rc;
  MLIIS_REQUIRE(dwm_types_ok(false, in_dtype, out_dtype), MLIIS_ERR_ARG, "dwconv_bn_fwd: unsupported storage types (in %d, out %d)", in_dtype, out_dtype);
  MLIIS_REQUIRE(z && w && y && aligned16(z) && aligned16(w) && aligned16(y), MLIIS_ERR_ARG, "dwconv_bn_fwd: null or unaligned pointer");
  const bool pre = bn_gamma != nullptr;
  if (pre) {
    MLIIS_REQUIRE(bn_beta && bn_mean && bn_rstd && aligned16(bn_gamma) && aligned16(bn_beta), MLIIS_ERR_ARG,
                  "dwconv_bn_fwd: the batch norm needs gamma, beta (16-byte aligned) and mean / rstd vectors");
    MLIIS_REQUIRE(bn_nblk >= 0 && (bn_nblk == 0 || (bn_part && aligned16(bn_part))), MLIIS_ERR_ARG, "dwconv_bn_fwd: bad statistics partials");
    MLIIS_REQUIRE((bn_moving_mean == nullptr) == (bn_moving_var == nullptr), MLIIS_ERR_ARG, "dwconv_bn_fwd: moving mean / variance come as a pair");
  }
  const MarchGeom g = march_geom_fwd(N, H, W, C, k, stride);
  MLIIS_REQUIRE(g.gx < (1LL << 31), MLIIS_ERR_UNSUPPORTED, "dwconv_bn_fwd: too many workgroups");
  if (stats_nblk) *stats_nblk = 0;
  if (stats_part != nullptr) {
    MLIIS_REQUIRE(stats_nblk && aligned16(stats_part), MLIIS_ERR_ARG, "dwconv_bn_fwd: statistics need a 16-byte aligned buffer and a stats_nblk output");
    MLIIS_REQUIRE((size_t)g.gx * 2 * C <= stats_floats, MLIIS_ERR_WORKSPACE, "dwconv_bn_fwd: statistics buffer too small (%zu floats needed, %zu given)",
                  (size_t)g.gx * 2 * C, stats_floats);
    *stats_nblk = (int)g.gx;
  }
  DwmArgs a{};
  a.x = z; a.w = w; a.y = y;
  a.Hi = H; a.Wi = W; a.Ho = g.Ho; a.Wo = g.Wo; a.C = C; a.pt = g.pt; a.pl = g.pl;
  a.bands = g.bands; a.bw = g.bw; a.chunks = g.chunks; a.rpc = g.rpc;
  a.stats_part = stats_part;
#ifdef DWM_DBG
  a.dbg = getenv("MLIIS_DWM_DBG") ? atoi(getenv("MLIIS_DWM_DBG")) : 0;
  a.stamps = getenv("MLIIS_DWM_STAMPS") ? (unsigned long long*)strtoull(getenv("MLIIS_DWM_STAMPS"), nullptr, 0) : nullptr;
#endif
  if (pre) a.bn = make_bn(bn_part, bn_nblk, (long long)N * H * W, bn_gamma, bn_beta, bn_mean, bn_rstd, bn_moving_mean, bn_moving_var, eps, momentum, 0);
#define DWM_FWD(K_, S_)                                             \
  do {                                                              \
    if (pre) launch_conv<K_, S_, true, false>(g, a, in_dtype, out_dtype, stream);  \
    else launch_conv<K_, S_, false, false>(g, a, in_dtype, out_dtype, stream);     \
  } while (0)
  if (k == 3 && stride == 1) DWM_FWD(3, 1);
  else if (k == 3) DWM_FWD(3, 2);
  else if (stride == 1) DWM_FWD(5, 1);
  else DWM_FWD(5, 2);
#undef DWM_FWD
  MLIIS_CHECK_LAUNCH("dwconv_bn_fwd");
  return MLIIS_OK;
}

// dx = gradient w.r.t. a = swish(bn(z)) (bn_gamma == NULL: a = z, the plain input); dw_part [blocks][k*k][C] slabs of the filter
// gradient (folded into dw when dw != NULL); bn_part [blocks][2][C] = stage 1 of the batch norm's backward for mliis_bn_bwd.
int mliis_dwconv_bn_bwd(const float* dy, const float* z, const float* bn_mean, const float* bn_rstd, const float* bn_gamma,
                        const float* bn_beta, const float* w, float* dx, float* dw, int N, int H, int W, int C, int k, int stride,
                        float* dw_part, size_t dw_part_floats, float* bn_part, size_t bn_part_floats, int* nblk, int dy_dtype, int zx_dtype,
                        hipStream_t stream) {
  int rc = dwm_check("dwconv_bn_bwd", N, H, W, C, k, stride);
  if (rc) return rc;
  MLIIS_REQUIRE(dwm_types_ok(true, dy_dtype, zx_dtype), MLIIS_ERR_ARG, "dwconv_bn_bwd: unsupported storage types (dy %d, z / dx %d)", dy_dtype, zx_dtype);
  MLIIS_REQUIRE(dy && z && w && dx && dw_part && nblk && aligned16(dy) && aligned16(z) && aligned16(w) && aligned16(dx) && aligned16(dw_part),
                MLIIS_ERR_ARG, "dwconv_bn_bwd: null or unaligned pointer");
  const bool pre = bn_gamma != nullptr;
  if (pre)
    MLIIS_REQUIRE(bn_beta && bn_mean && bn_rstd && aligned16(bn_gamma) && aligned16(bn_beta) && (bn_part == nullptr || aligned16(bn_part)),
                  MLIIS_ERR_ARG, "dwconv_bn_bwd: the batch norm needs gamma, beta (16-byte aligned), mean and rstd");
  const MarchGeom g = march_geom_bwd(N, H, W, C, k, stride);
  MLIIS_REQUIRE(g.gx < (1LL << 31), MLIIS_ERR_UNSUPPORTED, "dwconv_bn_bwd: too many workgroups");
  MLIIS_REQUIRE((size_t)g.gx * k * k * C <= dw_part_floats, MLIIS_ERR_WORKSPACE, "dwconv_bn_bwd: filter-gradient slabs need %zu floats, %zu given",
                (size_t)g.gx * k * k * C, dw_part_floats);
  MLIIS_REQUIRE(!pre || bn_part == nullptr || (size_t)g.gx * 2 * C <= bn_part_floats, MLIIS_ERR_WORKSPACE,
                "dwconv_bn_bwd: batch-norm partials need %zu floats, %zu given", (size_t)g.gx * 2 * C, bn_part_floats);
  *nblk = (int)g.gx;
  DwmArgs a{};
  a.x = dy; a.w = w; a.y = dx; a.z = z;
  a.C = C;
  a.bands = g.bands; a.bw = g.bw; a.chunks = g.chunks; a.rpc = g.rpc;
  a.stats_part = pre ? bn_part : nullptr;
  a.dw_part = dw_part;
  if (pre) a.bn = make_bn(nullptr, 0, 1, bn_gamma, bn_beta, const_cast<float*>(bn_mean), const_cast<float*>(bn_rstd), nullptr, nullptr, 0.f, 0.f, 0);
  if (stride == 1) {
    // dx = dy correlated with the reversed filter, padding K - 1 - pad; ring side and produced side have the same size
    a.Hi = H; a.Wi = W; a.Ho = H; a.Wo = W; a.pt = k - 1 - g.pt; a.pl = k - 1 - g.pl;
    if (k == 3) { if (pre) launch_conv<3, 1, true, true>(g, a, dy_dtype, zx_dtype, stream); else launch_conv<3, 1, false, true>(g, a, dy_dtype, zx_dtype, stream); }
    else { if (pre) launch_conv<5, 1, true, true>(g, a, dy_dtype, zx_dtype, stream); else launch_conv<5, 1, false, true>(g, a, dy_dtype, zx_dtype, stream); }
  } else {
    a.Hi = g.Ho; a.Wi = g.Wo; a.Ho = H; a.Wo = W; a.pt = g.pt; a.pl = g.pl;
    if (k == 3) { if (pre) launch_bwd_s2<3, true>(g, a, dy_dtype, zx_dtype, stream); else launch_bwd_s2<3, false>(g, a, dy_dtype, zx_dtype, stream); }
    else { if (pre) launch_bwd_s2<5, true>(g, a, dy_dtype, zx_dtype, stream); else launch_bwd_s2<5, false>(g, a, dy_dtype, zx_dtype, stream); }
  }
  MLIIS_CHECK_LAUNCH("dwconv_bn_bwd");
  if (dw != nullptr) {
    MLIIS_REQUIRE(aligned16(dw), MLIIS_ERR_ALIGN, "dwconv_bn_bwd: dw must be 16-byte aligned");
    hipLaunchKernelGGL(fold_flat_k, dim3(ceil_div(k * k * C, kFoldX)), dim3(kFoldX, kFoldY), 0, stream, dw_part, (int)g.gx, (long long)k * k * C, 1.0f,
                       dw, 0, (long long)k * k * C, 0LL, 0LL);
    MLIIS_CHECK_LAUNCH("dwconv_bn_bwd_fold");
  }
  return MLIIS_OK;
}

// The backward of the depthwise half of an MBConv block on the large maps in ONE launch: the depthwise batch norm's (bn1) backward
// apply formed while da2 / z1 are staged (see DwmDyBn), the depthwise conv's backward-data and filter-gradient slabs, stage 1 of the
// batch norm in front (bn0).  H, W = the depthwise conv's INPUT size; da2 / z1 [N,Ho,Wo,C]; z0 / dx [N,H,W,C].
int mliis_mbconv_dw_bwd_march(const float* da2, const float* z1, const float* mean1, const float* rstd1, const float* gamma1, const float* beta1,
                              const float* gate, const float* chan_add, const float* stage1, int stage1_nimg, float* dgamma1, float* dbeta1,
                              const float* z0, const float* mean0, const float* rstd0, const float* gamma0, const float* beta0, const float* w,
                              float* dx, int N, int H, int W, int C, int k, int stride, float* dw_part, size_t dw_part_floats, float* bn_part,
                              size_t bn_part_floats, int* nblk, int dy_dtype, int zx_dtype, hipStream_t stream) {
  int rc = dwm_check("mbconv_dw_bwd_march", N, H, W, C, k, stride);
  if (rc) return rc;
  MLIIS_REQUIRE(dwm_types_ok(true, dy_dtype, zx_dtype), MLIIS_ERR_ARG, "mbconv_dw_bwd_march: unsupported storage types (da2 / z1 %d, z0 / dx %d)", dy_dtype, zx_dtype);
  MLIIS_REQUIRE(da2 && z1 && mean1 && rstd1 && gamma1 && beta1 && gate && chan_add && stage1 && stage1_nimg > 0 && dgamma1 && dbeta1 && z0 && mean0 &&
                    rstd0 && gamma0 && beta0 && w && dx && dw_part && bn_part && nblk,
                MLIIS_ERR_ARG, "mbconv_dw_bwd_march: null pointer");
  MLIIS_REQUIRE(aligned16(da2) && aligned16(z1) && aligned16(mean1) && aligned16(rstd1) && aligned16(gamma1) && aligned16(beta1) && aligned16(gate) &&
                    aligned16(chan_add) && aligned16(stage1) && aligned16(dgamma1) && aligned16(dbeta1) && aligned16(z0) && aligned16(mean0) &&
                    aligned16(rstd0) && aligned16(gamma0) && aligned16(beta0) && aligned16(w) && aligned16(dx) && aligned16(dw_part) && aligned16(bn_part),
                MLIIS_ERR_ALIGN, "mbconv_dw_bwd_march: pointers must be 16-byte aligned");
  const MarchGeom g = march_geom_bwd(N, H, W, C, k, stride);
  MLIIS_REQUIRE(g.gx < (1LL << 31), MLIIS_ERR_UNSUPPORTED, "mbconv_dw_bwd_march: too many workgroups");
  MLIIS_REQUIRE((size_t)g.gx * k * k * C <= dw_part_floats && (size_t)g.gx * 2 * C <= bn_part_floats, MLIIS_ERR_WORKSPACE,
                "mbconv_dw_bwd_march: slab buffers too small (%zu / %zu floats needed)", (size_t)g.gx * k * k * C, (size_t)g.gx * 2 * C);
  *nblk = (int)g.gx;
  DwmArgs a{};
  a.x = da2; a.w = w; a.y = dx; a.z = z0;
  a.C = C;
  a.bands = g.bands; a.bw = g.bw; a.chunks = g.chunks; a.rpc = g.rpc;
  a.stats_part = bn_part;
  a.dw_part = dw_part;
  a.bn = make_bn(nullptr, 0, 1, gamma0, beta0, const_cast<float*>(mean0), const_cast<float*>(rstd0), nullptr, nullptr, 0.f, 0.f, 0);
  a.dyb = DwmDyBn{z1, mean1, rstd1, gamma1, beta1, gate, chan_add, stage1, stage1_nimg, 1.0 / ((double)N * g.Ho * g.Wo), dgamma1, dbeta1};
  if (stride == 1) {
    a.Hi = H; a.Wi = W; a.Ho = H; a.Wo = W; a.pt = k - 1 - g.pt; a.pl = k - 1 - g.pl;
  } else {
    a.Hi = g.Ho; a.Wi = g.Wo; a.Ho = H; a.Wo = W; a.pt = g.pt; a.pl = g.pl;
  }
  if (k == 3) launch_bwd_dybn<3>(g, a, stride, dy_dtype, zx_dtype, stream);
  else launch_bwd_dybn<5>(g, a, stride, dy_dtype, zx_dtype, stream);
  MLIIS_CHECK_LAUNCH("mbconv_dw_bwd_march");
  return MLIIS_OK;
}
}
