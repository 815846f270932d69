// Batch-norm (training) statistics / apply / backward, and the generic per-channel column reductions that the
// squeeze-excite, bias-gradient and pooled-branch ops reuse.  All tensors are [rows, C] fp32 views of NHWC
// activations with a row stride `ld` (floats), C % 4 == 0, 16-byte aligned; lanes run along C in float4 so every
// wave-instruction touches whole contiguous spans (HBM-bound kernels: SURVEY 2.1 K4/K5/K7/K8).
//
// Reductions are two-stage and deterministic: stage 1 writes per-block partial sums (from the producing GEMM's epilogue when there
// is one), stage 2 folds them in double precision in a fixed order INSIDE the consumer kernel (bn_apply_fused_k /
// bn_bwd_apply_fused_k: every block folds the partials of its own 32 channels), so training-mode BN costs one launch forward
// and two backward.  No float atomics anywhere.
//
// Reference semantics: models/efficientnet/utils.py:87-134 (non-fused BN, biased variance),
// models/efficientlab.py:185-190 (decoder: conv -> swish -> BN, fused BN => unbiased variance into the moving average),
// models/efficientnet/efficientnet_model.py:266,271,280-288 (BN -> swish, drop-connect + residual).
#include "common.hpp"
#include "se_wgrad.hpp"
#include "bn_fold.hpp"

namespace mliis {

constexpr int kColThreads = 256;
constexpr int kChanBlock = 32;   // channels per block: 8 float4 lanes = one 128-byte line per row
constexpr int kRowLanes = 32;    // rows walked in parallel by one block
constexpr int kBatch = 4;        // rows per thread whose loads are issued before any of them is consumed
constexpr int kPoolRows = 128;   // rows per pooling chunk of bn_apply_fused (one batch per thread)
constexpr int kStatsPartialTarget = 256;   // workgroups of mliis_bn_stats_partial (= partial blocks its consumer folds)

// Streaming geometry shared by the column reductions and the BN apply kernels: grid = (row chunks, ceil(C/32)[, segments]);
// a block owns 32 channels x rows_per_block rows, its 256 threads = 8 channel quads x 32 row lanes.  Small maps are latency-bound
// (a handful of waves per CU), so every thread issues the loads of kBatch rows before it touches any of them: the per-launch
// critical path is one or two memory round trips instead of one per row.
struct ColGeom {
  int gx;              // channel groups
  int rows_per_block;  // multiple of kRowLanes * kBatch
  int nblk;            // blocks along rows (per segment)
};

static inline ColGeom col_geom(long long rows_per_seg, int C, int nseg, int target_blocks = 2048) {
  ColGeom g;
  g.gx = ceil_div(C, kChanBlock);
  // (measured: 512 .. 8192 target blocks are within noise of each other on the whole step)
  long long want = target_blocks / ((long long)g.gx * (nseg > 0 ? nseg : 1));
  if (want < 1) want = 1;
  long long rpb = (rows_per_seg + want - 1) / want;
  const long long unit = kRowLanes * kBatch;   // (measured: 32 / 64 / 256 rows are 3-4 % slower on the whole step)
  rpb = (rpb + unit - 1) / unit * unit;
  g.rows_per_block = (int)rpb;
  g.nblk = ceil_div(rows_per_seg, rpb);
  return g;
}

// ---------------------------------------------------------------------------------------------------------------
// Generic partial column reduction.  Op: Raw load(seg,row,c) then eval(seg,row,c,raw,out[NV]).
// part layout: [seg][blk][v][C]
// ---------------------------------------------------------------------------------------------------------------
template <class Op>
__global__ __launch_bounds__(kColThreads) void colreduce_partial_k(Op op, int rows_per_seg, int C, int rows_per_block, int nblk,
                                                                    float* __restrict__ part) {
  constexpr int NV = Op::NV;
  __shared__ float4 sm[NV][4][8];
  const int t = threadIdx.x;
  const int q = t & 7, rl = t >> 3;
  const int c = blockIdx.y * kChanBlock + q * 4;
  const int seg = blockIdx.z;
  const bool active = c < C;
  float4 acc[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) acc[v] = f4zero();
  if (active) {
    const int r0 = blockIdx.x * rows_per_block;
    int r1 = r0 + rows_per_block;
    if (r1 > rows_per_seg) r1 = rows_per_seg;
    const long long base = (long long)seg * rows_per_seg;
    int r = r0 + rl;
    // first batch: its loads are issued before the per-channel constants are fetched (one memory round trip on small maps)
    typename Op::Raw first[kBatch];
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      const int ru = r + u * kRowLanes;
      op.load(seg, base + (ru < r1 ? ru : r0), c, first[u]);
    }
    const typename Op::Ctx ctx = op.ctx(c);   // per-channel constants, loaded once per thread
#pragma unroll
    for (int u = 0; u < kBatch; ++u)
      if (r + u * kRowLanes < r1) {
        float4 vals[NV];
        op.eval(ctx, first[u], vals);
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] = f4add(acc[v], vals[v]);
      }
    r += kBatch * kRowLanes;
    for (; r + (kBatch - 1) * kRowLanes < r1; r += kBatch * kRowLanes) {
      typename Op::Raw raw[kBatch];
#pragma unroll
      for (int u = 0; u < kBatch; ++u) op.load(seg, base + r + u * kRowLanes, c, raw[u]);
#pragma unroll
      for (int u = 0; u < kBatch; ++u) {
        float4 vals[NV];
        op.eval(ctx, raw[u], vals);
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] = f4add(acc[v], vals[v]);
      }
    }
    for (; r < r1; r += kRowLanes) {
      typename Op::Raw raw;
      op.load(seg, base + r, c, raw);
      float4 vals[NV];
      op.eval(ctx, raw, vals);
#pragma unroll
      for (int v = 0; v < NV; ++v) acc[v] = f4add(acc[v], vals[v]);
    }
  }
  // 8 row lanes of a wave share a channel quad: butterfly over lane bits 3..5, then the 4 waves through LDS (fixed order)
#pragma unroll
  for (int v = 0; v < NV; ++v) {
#pragma unroll
    for (int off = 8; off < 64; off <<= 1) {
      acc[v].x += __shfl_xor(acc[v].x, off);
      acc[v].y += __shfl_xor(acc[v].y, off);
      acc[v].z += __shfl_xor(acc[v].z, off);
      acc[v].w += __shfl_xor(acc[v].w, off);
    }
    if ((t & 63) < 8) sm[v][t >> 6][q] = acc[v];
  }
  __syncthreads();
  if (t < 8 * NV) {
    const int v = t >> 3, qq = t & 7;
    const int cc = blockIdx.y * kChanBlock + qq * 4;
    if (cc < C) {
      float4 s4 = f4add(f4add(sm[v][0][qq], sm[v][1][qq]), f4add(sm[v][2][qq], sm[v][3][qq]));
      st4(part + (((long long)seg * nblk + blockIdx.x) * NV + v) * C + cc, s4);
    }
  }
}

template <class Op>
static int launch_colreduce(Op op, long long rows_per_seg, int C, int nseg, float* part, size_t part_floats,
                            hipStream_t stream, ColGeom* out_g, const char* name, int target_blocks = 2048) {
  ColGeom g = col_geom(rows_per_seg, C, nseg, target_blocks);
  size_t need = (size_t)nseg * g.nblk * Op::NV * C;
  MLIIS_REQUIRE(need <= part_floats, MLIIS_ERR_WORKSPACE, "%s: workspace too small (%zu floats needed, %zu given)", name,
                need, part_floats);
  dim3 grid(g.nblk, g.gx, nseg);
  hipLaunchKernelGGL((colreduce_partial_k<Op>), grid, dim3(kColThreads), 0, stream, op, (int)rows_per_seg, C, g.rows_per_block, g.nblk,
                     part);
  MLIIS_CHECK_LAUNCH(name);
  *out_g = g;
  return MLIIS_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// BN statistics
// ---------------------------------------------------------------------------------------------------------------
struct StatsOp {
  static constexpr int NV = 2;
  const float* x;
  int ld;
  int pre_swish;
  typedef float4 Raw;
  struct Ctx {};
  __device__ __forceinline__ Ctx ctx(int) const { return Ctx(); }
  __device__ __forceinline__ void load(int, long long row, int c, Raw& r) const { r = ld4(x + row * ld + c); }
  __device__ __forceinline__ void eval(const Ctx&, const Raw& r, float4* o) const {
    float4 v = r;
    if (pre_swish) v = make_float4(swish_f(v.x), swish_f(v.y), swish_f(v.z), swish_f(v.w));
    o[0] = v;
    o[1] = f4mul(v, v);
  }
};

__global__ __launch_bounds__(256) void bn_stats_finalize_k(const float* __restrict__ part, int nblk, int C, double inv_n, float eps,
                                                           float one_minus_momentum, float ema_var_factor,
                                                           float* __restrict__ mean, float* __restrict__ rstd,
                                                           float* __restrict__ moving_mean, float* __restrict__ moving_var) {
  __shared__ double sm[kFoldY * (kFoldX + 1)];
  const int c = blockIdx.x * kFoldX + threadIdx.x;
  const bool ok = c < C;
  const double s = fold_partials(part, nblk, 2LL * C, c, ok, sm);
  const double ss = fold_partials(part, nblk, 2LL * C, (long long)C + c, ok, sm);
  if (!ok || threadIdx.y != 0) return;
  double m = s * inv_n;
  double var = ss * inv_n - m * m;
  if (var < 0.0) var = 0.0;
  mean[c] = (float)m;
  rstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (moving_mean != nullptr) {
    float mm = moving_mean[c], mv = moving_var[c];
    moving_mean[c] = mm - (mm - (float)m) * one_minus_momentum;
    moving_var[c] = mv - (mv - (float)(var * (double)ema_var_factor)) * one_minus_momentum;
  }
}

// y = [post_swish]( gamma * ([pre_swish](x) - mean) * rstd + beta ) * img_scale[n] + res
__global__ __launch_bounds__(256) void bn_apply_k(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy,
                                                  long long rows, int C, int rows_per_img,
                                                  const float* __restrict__ mean, const float* __restrict__ rstd,
                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  int pre_swish, int post_swish, const float* __restrict__ img_scale,
                                                  const float* __restrict__ res, int ldr) {
  const int Q = C >> 2;
  const long long total = rows * Q;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / Q;
    const int c = (int)(i - r * Q) << 2;
    float4 v = ld4(x + r * ldx + c);
    if (pre_swish) v = make_float4(swish_f(v.x), swish_f(v.y), swish_f(v.z), swish_f(v.w));
    const float4 m = ld4(mean + c), rs = ld4(rstd + c), g = ld4(gamma + c), b = ld4(beta + c);
    float4 o;
    o.x = fmaf((v.x - m.x) * rs.x, g.x, b.x);
    o.y = fmaf((v.y - m.y) * rs.y, g.y, b.y);
    o.z = fmaf((v.z - m.z) * rs.z, g.z, b.z);
    o.w = fmaf((v.w - m.w) * rs.w, g.w, b.w);
    if (post_swish) o = make_float4(swish_f(o.x), swish_f(o.y), swish_f(o.z), swish_f(o.w));
    if (img_scale != nullptr) o = f4scale(o, img_scale[r / rows_per_img]);
    if (res != nullptr) o = f4add(o, ld4(res + r * ldr + c));
    st4(y + r * ldy + c, o);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Fused "fold statistics + normalise": grid = (ceil(C/32), row chunks).  Every block folds the stage-1 partial sums of
// ITS 32 channels (32 channels x 8 fold lanes, double precision, fixed order), keeps mean / rstd in LDS and streams its rows
// (8 float4 lanes = one 128-byte line per row).  Blocks with blockIdx.y == 0 publish mean / rstd for the backward pass and
// apply the moving-average update -- so no separate finalize launch exists.
// ---------------------------------------------------------------------------------------------------------------
// A second batch norm of the same shape and flags in the same launch (gridDim.z == 2: the two independent branches of an RSD module,
// models/efficientlab.py:185-197): workgroups with blockIdx.z == 1 take their tensors and parameters from `alt`.
struct BnApplyAlt {
  const float* x;
  float* y;
  BnFold f;
  const float* gamma;
  const float* beta;
};
// (<= 168 VGPRs: three workgroups per CU.  With the 16-block fold batches of an earlier form the kernel needed 216 and two fitted -- the
//  600-1568-workgroup launches of the large maps ran in two or three rounds with few waves in flight; stamps build, round 3)
// TX / TY: storage types of x and y (float | bf16s: the expanded tensors of `--precision bf16-storage`; the pair form is float only)
// RES / ISC: a residual operand / a per-image scale exist.  Compile-time: without them a row is ONE load instruction instead of three
// (the shadow loads that used to stand in for absent operands were L1 hits, but these launches are bound by the number of
// vector-memory instructions, not by bytes -- profiles/r04_notes.md).
template <typename TX, typename TY, bool RES = true, bool ISC = true>
__global__ __launch_bounds__(256, 3) void bn_apply_fused_k(const TX* __restrict__ x, int ldx, TY* __restrict__ y, int ldy,
                                                        long long rows, int C, int rows_per_img, BnFold f,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int pre_swish, int post_swish, const float* __restrict__ img_scale,
                                                        const float* __restrict__ res, int ldr, int rows_per_block,
                                                        float* __restrict__ pool_part, int pool_chunks, BnApplyAlt alt) {
  if (blockIdx.z == 1) {   // (uniform)
    x = reinterpret_cast<const TX*>(alt.x);
    y = reinterpret_cast<TY*>(alt.y);
    f = alt.f;
    gamma = alt.gamma;
    beta = alt.beta;
  }
  __shared__ double smd[2 * 32 * 32];
  __shared__ __attribute__((aligned(16))) float s_mean[kChanBlock], s_rstd[kChanBlock];
  const int t = threadIdx.x;
  const int c0 = blockIdx.x * kChanBlock;
  const int q = t & 7, rl = t >> 3;
  const bool cok = c0 + q * 4 < C;
  const int c = cok ? c0 + q * 4 : 0;   // surplus lanes shadow channel 0 with an empty row range (they stay for the reductions)
  // row range: plain chunks of rows_per_block, or (pooling) chunk (blockIdx.y % pool_chunks) of image (blockIdx.y / pool_chunks)
  long long r0 = (long long)blockIdx.y * rows_per_block, r1 = r0 + rows_per_block;
  if (pool_part != nullptr) {
    const int img = blockIdx.y / pool_chunks, ch = blockIdx.y - img * pool_chunks;
    r0 = (long long)img * rows_per_img + (long long)ch * rows_per_block;
    r1 = r0 + rows_per_block;
    if (r1 > (long long)(img + 1) * rows_per_img) r1 = (long long)(img + 1) * rows_per_img;
  }
  if (r1 > rows) r1 = rows;
  if (!cok) r1 = r0;
  // The first batch of rows does not depend on the statistics: fetch it BEFORE folding them, so the small-map launches (one batch
  // per thread) pay one memory round trip instead of two.
  long long r = r0 + rl;
  const float* rbase = res != nullptr ? res : gamma;   // no residual: the loads shadow gamma (ignored later) -- no branch around them
  const int rld = res != nullptr ? ldr : 0;
  const int rcol = res != nullptr ? c : 0;
  const float* isb = img_scale != nullptr ? img_scale : gamma;   // (absent: shadows gamma, ignored)
  float4 v0[kBatch], rv0[kBatch];
  float is0[kBatch];
#pragma unroll
  for (int u = 0; u < kBatch; ++u) {
    const long long ru = r + u * kRowLanes;
    const long long rr = ru < r1 ? ru : 0;
    v0[u] = ldq(x + rr * ldx + c);
    rv0[u] = RES ? ld4(rbase + rr * rld + rcol) : f4zero();
    is0[u] = ISC ? isb[img_scale != nullptr ? (int)rr / rows_per_img : 0] : 1.f;
  }
  const float4 g = ld4(gamma + c), b = ld4(beta + c);   // (before the fold: they do not depend on it)
  double s, ss;
  // (16 partial blocks per lane and round trip when a producer left many: one round per 512 instead of per 256)
  fold32<8>(f.part, f.nblk, C, c0, smd, s, ss);
  if (t < 32) {
    double m = s * f.inv_n;
    double var = ss * f.inv_n - m * m;
    if (var < 0.0) var = 0.0;
    const float mf = (float)m, rf = (float)(1.0 / sqrt(var + (double)f.eps));
    s_mean[t] = mf;
    s_rstd[t] = rf;
    const int cc = c0 + t;
    if (blockIdx.y == 0 && cc < C) {
      f.mean[cc] = mf;
      f.rstd[cc] = rf;
      if (f.moving_mean != nullptr) {
        const float mm = f.moving_mean[cc], mv = f.moving_var[cc];
        f.moving_mean[cc] = mm - (mm - mf) * f.one_minus_momentum;
        f.moving_var[cc] = mv - (mv - (float)(var * (double)f.ema_var_factor)) * f.one_minus_momentum;
      }
    }
  }
  __syncthreads();
  const float4 m = ld4(s_mean + q * 4), rs = ld4(s_rstd + q * 4);
  float4 pool = f4zero();   // sum of this thread's outputs (squeeze-excite pooling, training)
  auto finish = [&](long long rw, float4 v, float4 rv, float isc) {
    if (pre_swish) v = make_float4(swish_f(v.x), swish_f(v.y), swish_f(v.z), swish_f(v.w));
    float4 o;
    o.x = fmaf((v.x - m.x) * rs.x, g.x, b.x);
    o.y = fmaf((v.y - m.y) * rs.y, g.y, b.y);
    o.z = fmaf((v.z - m.z) * rs.z, g.z, b.z);
    o.w = fmaf((v.w - m.w) * rs.w, g.w, b.w);
    if (post_swish) o = make_float4(swish_f(o.x), swish_f(o.y), swish_f(o.z), swish_f(o.w));
    if (img_scale != nullptr) o = f4scale(o, isc);
    if (res != nullptr) o = f4add(o, rv);
    o = stored_value(y, o);   // (bf16 storage: the pooled means see what the consumers will read)
    pool = f4add(pool, o);
    stq(y + rw * ldy + c, o);
  };
#pragma unroll
  for (int u = 0; u < kBatch; ++u)
    if (r + u * kRowLanes < r1) finish(r + u * kRowLanes, v0[u], rv0[u], is0[u]);
  r += kBatch * kRowLanes;
  for (; r + (kBatch - 1) * kRowLanes < r1; r += kBatch * kRowLanes) {
    float4 v[kBatch], rv[kBatch];
    float is[kBatch];
#pragma unroll
    for (int u = 0; u < kBatch; ++u) {
      v[u] = ldq(x + (r + u * kRowLanes) * ldx + c);
      rv[u] = RES ? ld4(rbase + (r + u * kRowLanes) * rld + rcol) : f4zero();
      is[u] = ISC ? isb[img_scale != nullptr ? (int)(r + u * kRowLanes) / rows_per_img : 0] : 1.f;
    }
#pragma unroll
    for (int u = 0; u < kBatch; ++u) finish(r + u * kRowLanes, v[u], rv[u], is[u]);
  }
  for (; r < r1; r += kRowLanes)
    finish(r, ldq(x + r * ldx + c), RES ? ld4(rbase + r * rld + rcol) : f4zero(), ISC ? isb[img_scale != nullptr ? (int)r / rows_per_img : 0] : 1.f);
  if (pool_part == nullptr) return;   // (uniform)
  // pooled partial of this block: butterfly over the 8 row lanes of a wave that share a quad, then the 4 waves through LDS
#pragma unroll
  for (int off = 8; off < 64; off <<= 1) {
    pool.x += __shfl_xor(pool.x, off);
    pool.y += __shfl_xor(pool.y, off);
    pool.z += __shfl_xor(pool.z, off);
    pool.w += __shfl_xor(pool.w, off);
  }
  float4* smp = reinterpret_cast<float4*>(smd);   // the statistics fold is done with it
  __syncthreads();
  if ((t & 63) < 8) smp[(t >> 6) * 8 + q] = pool;
  __syncthreads();
  if (t < 8 && c0 + t * 4 < C)
    st4(pool_part + (long long)blockIdx.y * C + c0 + t * 4, f4add(f4add(smp[t], smp[8 + t]), f4add(smp[16 + t], smp[24 + t])));
}

// The apply kernels keep three workgroups per CU resident (<= 168 VGPRs): a grid in (768, 1536] would run a second, mostly empty round.
// Taller row chunks until the launch fits one round of 256 CUs.
static inline int bn_max_blocks() { return 768; }   // (whole step: no cap 3167-3169, 1024: 3170, 768: 3173-3178, 512: 3170-3174 images/s)
static inline void fit_one_round(long long rows, int gx, int z, int* gy, int* rpb) {
  while ((long long)gx * *gy * z > bn_max_blocks() && *gy > 1) {
    *rpb *= 2;
    *gy = (int)((rows + *rpb - 1) / *rpb);
  }
}
static inline void chan_grid(long long rows, int C, int* gx, int* gy, int* rows_per_block) {
  const ColGeom g = col_geom(rows, C, 1);
  *gx = g.gx;
  *gy = g.nblk;
  *rows_per_block = g.rows_per_block;
}

// Geometry of the BN backward pair (reduce pass + apply pass share it: the apply blocks fold the reduce pass's partials).  On large
// tensors the default 128-row blocks mean 400-800 partials re-folded by as many apply blocks -- more L2 traffic than payload -- so the
// pair then runs on four times fewer, taller blocks (same rule as mliis_bn_apply_fused).
constexpr int kManyPartials = 128;   // (whole step, one box: off 2290, 256 -> 2322, 128 -> 2330, 64 -> 2323 images/s)
static inline int bn_bwd_target(long long rows, int C) { return col_geom(rows, C, 1).nblk >= kManyPartials ? 512 : 2048; }
static inline void bn_bwd_grid(long long rows, int C, int* gx, int* gy, int* rows_per_block) {
  const ColGeom g = col_geom(rows, C, 1, bn_bwd_target(rows, C));
  *gx = g.gx;
  *gy = g.nblk;
  *rows_per_block = g.rows_per_block;
  fit_one_round(rows, *gx, 1, gy, rows_per_block);
}

// upstream gradient seen by the BN output: dy * img_scale[n] * chan_scale[n,c] + chan_add[n,c]
template <bool SE, typename T = float>   // SE: per-image vectors (drop-connect scale, squeeze-excite gate / pooled gradient) are present;
struct BnBwdCommon {                       // T: storage type of x and dy (and of the dx the apply kernel writes)
  const T* x;   // conv output saved in forward (pre-BN, pre-swish if pre_swish)
  int ldx;
  const T* dy;
  int lddy;
  int rows_per_img;
  int C;
  const float* mean;
  const float* rstd;
  const float* gamma;
  const float* beta;
  int pre_swish, post_swish;
  const float* img_scale;   // [N] or null        (drop-connect)
  const float* chan_scale;  // [N,C] or null      (squeeze-excite gate)
  const float* chan_add;    // [N,C] or null      (squeeze-excite pooled-gradient / HW)
  // everything a row needs from memory, fetched together (no load is left inside finish(): the per-image vectors used to cost
  // one dependent round trip per row)
  struct Raw { float4 x, g, cs, ca; float is; };
  __device__ __forceinline__ void load_raw(long long row, int c, Raw& r) const {
    r.x = ldq(x + row * ldx + c);
    r.g = ldq(dy + row * lddy + c);
    if (SE) {
      const long long n = (int)row / rows_per_img;     // rows < 2^31 (checked on the host)
      // absent vectors read a valid dummy (the mean vector) and are ignored in finish(): no branch around the loads
      r.cs = ld4((chan_scale != nullptr ? chan_scale + n * C : mean) + c);
      r.ca = ld4((chan_add != nullptr ? chan_add + n * C : mean) + c);
      r.is = (img_scale != nullptr ? img_scale : mean)[img_scale != nullptr ? n : 0];
    }
  }
  struct Ctx { float4 m, rs, ga, be; };   // per-channel constants of this thread's quad, loaded once
  __device__ __forceinline__ Ctx ctx(int c) const { return Ctx{ld4(mean + c), ld4(rstd + c), ld4(gamma + c), ld4(beta + c)}; }
  __device__ __forceinline__ void finish(const Ctx& k, const Raw& r, float4& xin, float4& xhat, float4& g) const {
    xin = r.x;
    float4 v = xin;
    if (pre_swish) v = make_float4(swish_f(v.x), swish_f(v.y), swish_f(v.z), swish_f(v.w));
    const float4 m = k.m, rs = k.rs;
    xhat = make_float4((v.x - m.x) * rs.x, (v.y - m.y) * rs.y, (v.z - m.z) * rs.z, (v.w - m.w) * rs.w);
    g = r.g;
    if (SE) {
      if (img_scale != nullptr) g = f4scale(g, r.is);
      if (chan_scale != nullptr) g = f4mul(g, r.cs);
      if (chan_add != nullptr) g = f4add(g, r.ca);
    }
    if (post_swish) {
      const float4 ga = k.ga, be = k.be;
      g.x *= swish_grad_f(fmaf(xhat.x, ga.x, be.x));
      g.y *= swish_grad_f(fmaf(xhat.y, ga.y, be.y));
      g.z *= swish_grad_f(fmaf(xhat.z, ga.z, be.z));
      g.w *= swish_grad_f(fmaf(xhat.w, ga.w, be.w));
    }
  }
};

template <bool SE, typename T = float>
struct BnBwdOp {
  static constexpr int NV = 2;
  BnBwdCommon<SE, T> p;
  typedef typename BnBwdCommon<SE, T>::Raw Raw;
  typedef typename BnBwdCommon<SE, T>::Ctx Ctx;
  __device__ __forceinline__ Ctx ctx(int c) const { return p.ctx(c); }
  __device__ __forceinline__ void load(int, long long row, int c, Raw& r) const { p.load_raw(row, c, r); }
  __device__ __forceinline__ void eval(const Ctx& k, const Raw& r, float4* o) const {
    float4 xin, xhat, g;
    p.finish(k, r, xin, xhat, g);
    o[0] = g;
    o[1] = f4mul(g, xhat);
  }
};

// two plain batch-norm backward reduce passes of the same shape in one launch: segment (blockIdx.z) = problem
struct BnBwdPairOp {
  static constexpr int NV = 2;
  BnBwdCommon<false> p[2];
  long long rows;
  typedef BnBwdCommon<false>::Raw Raw;
  typedef BnBwdCommon<false>::Ctx Ctx;
  __device__ __forceinline__ Ctx ctx(int c) const { return p[blockIdx.z].ctx(c); }
  __device__ __forceinline__ void load(int seg, long long row, int c, Raw& r) const { p[seg].load_raw(row - seg * rows, c, r); }
  __device__ __forceinline__ void eval(const Ctx& k, const Raw& r, float4* o) const {
    float4 xin, xhat, g;
    p[blockIdx.z].finish(k, r, xin, xhat, g);
    o[0] = g;
    o[1] = f4mul(g, xhat);
  }
};

// finalize (fold the two gradient sums of this block's 32 channels) + input-gradient pass in one launch
// Identity-skip gradient written by the same pass: dskip[r, c] (+)= dy[r, c] (the raw upstream gradient, before any scale).
struct SkipOut {
  float* d;   // nullable
  int ld;
  int accumulate;
  __device__ __forceinline__ void put(long long r, int c, const float4& g) const {
    if (d == nullptr) return;
    float* dst = d + r * ld + c;
    st4(dst, accumulate ? f4add(ld4(dst), g) : g);
  }
};

// (second problem of the same shape in the same launch, as BnApplyAlt)
struct BnBwdAlt {
  const float *x, *dy, *mean, *rstd, *gamma, *beta, *part;
  float *dgamma, *dbeta, *dx, *dxsum_part;
};
template <bool SE, typename T = float>
__global__ __launch_bounds__(256, SE ? 2 : 3) void bn_bwd_apply_fused_k(BnBwdCommon<SE, T> p, long long rows, const float* __restrict__ part, int nblk,
                                                            double inv_n, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                            T* __restrict__ dx, int lddx, int rows_per_block, SkipOut skip,
                                                            float* __restrict__ dxsum_part, BnBwdAlt alt) {
  if (blockIdx.z == 1) {   // (uniform; the pair form is float only)
    p.x = reinterpret_cast<const T*>(alt.x); p.dy = reinterpret_cast<const T*>(alt.dy); p.mean = alt.mean; p.rstd = alt.rstd; p.gamma = alt.gamma;
    p.beta = alt.beta;
    part = alt.part; dgamma = alt.dgamma; dbeta = alt.dbeta; dx = reinterpret_cast<T*>(alt.dx); dxsum_part = alt.dxsum_part;
  }
  __shared__ double smd[2 * 32 * 32];
  __shared__ __attribute__((aligned(16))) float s_c1[kChanBlock], s_c2[kChanBlock];
  const int t = threadIdx.x;
  const int c0 = blockIdx.x * kChanBlock;
  const int q = t & 7, rl = t >> 3;
  const bool cok = c0 + q * 4 < p.C;
  const int c = cok ? c0 + q * 4 : 0;
  long long r1 = (long long)(blockIdx.y + 1) * rows_per_block;
  if (r1 > rows) r1 = rows;
  long long r = (long long)blockIdx.y * rows_per_block + rl;
  if (!cok) r1 = r;
  // first batch of rows fetched BEFORE the fold of the gradient sums (it does not depend on them): one round trip on small maps
  typename BnBwdCommon<SE, T>::Raw raw0[kBatch];
#pragma unroll
  for (int u = 0; u < kBatch; ++u) {
    const long long ru = r + u * kRowLanes;
    p.load_raw(ru < r1 ? ru : 0, c, raw0[u]);
  }
  const typename BnBwdCommon<SE, T>::Ctx kc = p.ctx(c);
  double s, sx;
  fold32<8>(part, nblk, p.C, c0, smd, s, sx);
  if (t < 32) {
    s_c1[t] = (float)(s * inv_n);
    s_c2[t] = (float)(sx * inv_n);
    const int cc = c0 + t;
    if (blockIdx.y == 0 && cc < p.C) {
      dbeta[cc] = (float)s;
      dgamma[cc] = (float)sx;
    }
  }
  __syncthreads();
  if (!cok && dxsum_part == nullptr) return;   // (surplus lanes stay for the column-sum reduction, with an empty row range)
  const float4 a = ld4(s_c1 + q * 4), b = ld4(s_c2 + q * 4), ga = kc.ga, rs = kc.rs;
  float4 dsum = f4zero();   // column sum of this thread's dx rows (bias gradient of a conv -> swish -> BN stack)
  auto finish = [&](long long rw, const typename BnBwdCommon<SE, T>::Raw& raw) {
    float4 xin, xhat, g;
    skip.put(rw, c, raw.g);
    p.finish(kc, raw, xin, xhat, g);
    float4 d;
    d.x = ga.x * rs.x * (g.x - a.x - xhat.x * b.x);
    d.y = ga.y * rs.y * (g.y - a.y - xhat.y * b.y);
    d.z = ga.z * rs.z * (g.z - a.z - xhat.z * b.z);
    d.w = ga.w * rs.w * (g.w - a.w - xhat.w * b.w);
    if (p.pre_swish) {
      d.x *= swish_grad_f(xin.x);
      d.y *= swish_grad_f(xin.y);
      d.z *= swish_grad_f(xin.z);
      d.w *= swish_grad_f(xin.w);
    }
    dsum = f4add(dsum, d);
    stq(dx + rw * lddx + c, d);
  };
#pragma unroll
  for (int u = 0; u < kBatch; ++u)
    if (r + u * kRowLanes < r1) finish(r + u * kRowLanes, raw0[u]);
  r += kBatch * kRowLanes;
  for (; r + (kBatch - 1) * kRowLanes < r1; r += kBatch * kRowLanes) {
    typename BnBwdCommon<SE, T>::Raw raw[kBatch];
#pragma unroll
    for (int u = 0; u < kBatch; ++u) p.load_raw(r + u * kRowLanes, c, raw[u]);
#pragma unroll
    for (int u = 0; u < kBatch; ++u) finish(r + u * kRowLanes, raw[u]);
  }
  for (; r < r1; r += kRowLanes) {
    typename BnBwdCommon<SE, T>::Raw raw;
    p.load_raw(r, c, raw);
    finish(r, raw);
  }
  if (dxsum_part == nullptr) return;   // (uniform)
#pragma unroll
  for (int off = 8; off < 64; off <<= 1) {
    dsum.x += __shfl_xor(dsum.x, off);
    dsum.y += __shfl_xor(dsum.y, off);
    dsum.z += __shfl_xor(dsum.z, off);
    dsum.w += __shfl_xor(dsum.w, off);
  }
  float4* smp = reinterpret_cast<float4*>(smd);   // the fold is done with it
  __syncthreads();
  if ((t & 63) < 8) smp[(t >> 6) * 8 + q] = dsum;
  __syncthreads();
  if (t < 8 && c0 + t * 4 < p.C)
    st4(dxsum_part + (long long)blockIdx.y * p.C + c0 + t * 4, f4add(f4add(smp[t], smp[8 + t]), f4add(smp[16 + t], smp[24 + t])));
}

// ---------------------------------------------------------------------------------------------------------------
// Generic sums: out[seg][v][c] = scale * sum_rows f(...)
// ---------------------------------------------------------------------------------------------------------------
template <typename T = float>   // T: storage type of a and b
struct SumOp {  // column sum of a (optionally times b)
  static constexpr int NV = 1;
  const T* a;
  int lda;
  const T* b;  // nullable
  int ldb;
  struct Raw { float4 a, b; };
  struct Ctx {};
  __device__ __forceinline__ Ctx ctx(int) const { return Ctx(); }
  __device__ __forceinline__ void load(int, long long row, int c, Raw& r) const {
    r.a = ldq(a + row * lda + c);
    r.b = b != nullptr ? ldq(b + row * ldb + c) : make_float4(1.f, 1.f, 1.f, 1.f);
  }
  __device__ __forceinline__ void eval(const Ctx&, const Raw& r, float4* o) const { o[0] = f4mul(r.a, r.b); }
};

// ONE pass over (da2, z1) of an MBConv block for everything the squeeze-excite backward and the depthwise batch norm's backward need
// from those two tensors (efficientnet_model.py:238-251,271): with u = gamma * xhat + beta, a1 = swish(u), the upstream gradient of
// a1 is g1 = da2 * gate[n] + chan_add[n] (per-image vectors that only exist after the squeeze-excite backward), so per image n
//     v0 = sum da2 * a1                  (the gate's gradient)
//     v1 = sum da2 * swish'(u)           v2 = sum da2 * swish'(u) * xhat
//     v3 = sum swish'(u)                 v4 = sum swish'(u) * xhat
// and the batch norm's stage-1 sums follow without touching the tensors again:
//     sum g = sum_n gate[n] v1[n] + chan_add[n] v3[n],   sum g xhat = sum_n gate[n] v2[n] + chan_add[n] v4[n]
// (mliis_se_mlp_bwd_bn forms them).  Replaces the gate-gradient column sum (+ its finalize) and the batch norm's reduce pass.
template <typename T = float>   // T: storage type of z1 and da2
struct SeBnOp {
  static constexpr int NV = 5;
  const T* x;    // z1: the batch norm's input
  int ldx;
  const T* dy;   // da2
  int lddy;
  const float *mean, *rstd, *gamma, *beta;
  struct Raw { float4 x, g; };
  struct Ctx { float4 m, rs, ga, be; };
  __device__ __forceinline__ Ctx ctx(int c) const { return Ctx{ld4(mean + c), ld4(rstd + c), ld4(gamma + c), ld4(beta + c)}; }
  __device__ __forceinline__ void load(int, long long row, int c, Raw& r) const {
    r.x = ldq(x + row * ldx + c);
    r.g = ldq(dy + row * lddy + c);
  }
  __device__ __forceinline__ void one(float x, float g, float m, float rs, float ga, float be, float* o) const {
    const float xh = (x - m) * rs;
    const float u = fmaf(xh, ga, be);
    const float sg = sigmoid_f(u);
    const float d = sg * (1.0f + u * (1.0f - sg));   // swish'(u)
    o[0] = g * u * sg;
    o[1] = g * d;
    o[2] = g * d * xh;
    o[3] = d;
    o[4] = d * xh;
  }
  __device__ __forceinline__ void eval(const Ctx& k, const Raw& r, float4* o) const {
    float a[5], b[5], c[5], d[5];
    one(r.x.x, r.g.x, k.m.x, k.rs.x, k.ga.x, k.be.x, a);
    one(r.x.y, r.g.y, k.m.y, k.rs.y, k.ga.y, k.be.y, b);
    one(r.x.z, r.g.z, k.m.z, k.rs.z, k.ga.z, k.be.z, c);
    one(r.x.w, r.g.w, k.m.w, k.rs.w, k.ga.w, k.be.w, d);
#pragma unroll
    for (int v = 0; v < 5; ++v) o[v] = make_float4(a[v], b[v], c[v], d[v]);
  }
};

// sum_rows x[row, c] * (mask[row,c]) * dy[row, j], j = 0,1   (final 1x1 conv, Cout = 2: weight gradient)
struct Outer2Op {
  static constexpr int NV = 2;
  const float* x;
  int ldx;
  const float* mask;  // nullable, same layout as x
  const float* dy;    // [rows, 2]
  struct Raw { float4 x, m; float2 d; };
  struct Ctx {};
  __device__ __forceinline__ Ctx ctx(int) const { return Ctx(); }
  __device__ __forceinline__ void load(int, long long row, int c, Raw& r) const {
    r.x = ld4(x + row * ldx + c);
    r.m = mask != nullptr ? ld4(mask + row * ldx + c) : make_float4(1.f, 1.f, 1.f, 1.f);
    r.d = *reinterpret_cast<const float2*>(dy + row * 2);
  }
  __device__ __forceinline__ void eval(const Ctx&, const Raw& r, float4* o) const {
    const float4 v = f4mul(r.x, r.m);
    o[0] = f4scale(v, lone(r.d.x));   // (the two halves of an 8-byte load, each broadcast over four channels: common.hpp, lone())
    o[1] = f4scale(v, lone(r.d.y));
  }
};

// out layout [seg][v][C]; part layout [seg][blk][v][C]; one block column = (seg, v, 16 channels)
__global__ __launch_bounds__(256) void sum_finalize_k(const float* __restrict__ part, int nblk, int NV, int C, int nseg, float scale,
                                                      float* __restrict__ out, int accumulate) {
  __shared__ double sm[kFoldY * (kFoldX + 1)];
  const int c = blockIdx.x * kFoldX + threadIdx.x;
  const int v = blockIdx.y, seg = blockIdx.z;
  const bool ok = c < C;
  const double s = fold_partials(part + (long long)seg * nblk * NV * C, nblk, (long long)NV * C, (long long)v * C + c, ok, sm);
  if (!ok || threadIdx.y != 0) return;
  const long long i = ((long long)seg * NV + v) * C + c;
  const float r = (float)(s * (double)scale);
  out[i] = accumulate ? out[i] + r : r;
}

// All weight-gradient slab folds of a backward pass in ONE launch.  desc (device int64 [ndesc][8]) = {part_off, out_off, total,
// seg_len, seg_stride, seg_off, nblk, tile_begin}; block b handles kFoldTile consecutive outputs of the descriptor whose tile range
// contains b (binary search over tile_begin).  256 threads = 64 output quads (1 KB contiguous per slab) x 4 slab lanes; each lane
// strides over the slabs with float4 loads issued four at a time, accumulates in double, and the 4 lanes are combined through LDS
// in a fixed order.
constexpr int kFoldTile = 256;

__global__ __launch_bounds__(256) void fold_batched_k(const float* __restrict__ part_base, float* __restrict__ out_base,
                                                      const long long* __restrict__ desc, int ndesc, long long fold_tiles,
                                                      const long long* __restrict__ se_desc, int n_se) {
  __shared__ double sm[3][64][4];
  const long long tile = blockIdx.x;
  if (tile >= fold_tiles) {   // (uniform) the squeeze-excite weight gradients of the pass ride behind the fold's tiles (se_wgrad.hpp)
    se_wgrad_tile(se_desc, n_se, tile - fold_tiles);
    return;
  }
  int lo = 0, hi = ndesc - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (desc[(long long)mid * 8 + 7] <= tile) lo = mid; else hi = mid - 1;
  }
  const long long* d = desc + (long long)lo * 8;
  const long long total = d[2];
  const int nblk = (int)d[6];
  const int t = threadIdx.x, ql = t & 63, sl = t >> 6;
  const long long i = (tile - d[7]) * kFoldTile + ql * 4;
  const bool vec = ((total | d[0]) & 3) == 0;
  const float* p = part_base + d[0] + i;
  double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  if (i < total) {
    if (vec) {
      int k = sl;
      // many slabs, few outputs (stem: 448 slabs of 864 floats; 32-channel depthwise filters: 512 slabs of 288): sixteen loads in
      // flight per lane, or the handful of blocks that fold them are a 30-round-trip critical path of the whole launch
      for (; k + 60 < nblk; k += 64) {
        float4 v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = ld4(p + (long long)(k + 4 * u) * total);
#pragma unroll
        for (int u = 0; u < 16; ++u) { a0 += v[u].x; a1 += v[u].y; a2 += v[u].z; a3 += v[u].w; }
      }
      for (; k + 12 < nblk; k += 16) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = ld4(p + (long long)(k + 4 * u) * total);
#pragma unroll
        for (int u = 0; u < 4; ++u) { a0 += v[u].x; a1 += v[u].y; a2 += v[u].z; a3 += v[u].w; }
      }
      for (; k < nblk; k += 4) {
        const float4 v = ld4(p + (long long)k * total);
        a0 += v.x; a1 += v.y; a2 += v.z; a3 += v.w;
      }
    } else {
      for (int k = sl; k < nblk; k += 4) {
        const float* pk = p + (long long)k * total;
        a0 += pk[0];
        if (i + 1 < total) a1 += pk[1];
        if (i + 2 < total) a2 += pk[2];
        if (i + 3 < total) a3 += pk[3];
      }
    }
  }
  if (sl > 0) {
    double* w = sm[sl - 1][ql];
    w[0] = a0; w[1] = a1; w[2] = a2; w[3] = a3;
  }
  __syncthreads();
  if (sl != 0 || i >= total) return;
#pragma unroll
  for (int j = 0; j < 3; ++j) { a0 += sm[j][ql][0]; a1 += sm[j][ql][1]; a2 += sm[j][ql][2]; a3 += sm[j][ql][3]; }
  const long long seg_len = d[3], seg_stride = d[4], seg_off = d[5];
  float* out = out_base + d[1];
  if (vec && ((seg_len | seg_stride | seg_off | d[1]) & 3) == 0) {
    st4(out + (i / seg_len) * seg_stride + seg_off + i % seg_len, make_float4((float)a0, (float)a1, (float)a2, (float)a3));
  } else {
    const double a[4] = {a0, a1, a2, a3};
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (i + e < total) out[((i + e) / seg_len) * seg_stride + seg_off + (i + e) % seg_len] = (float)a[e];
  }
}

// Short segments (<= 1024 rows, e.g. squeeze-excite pools on 28x28 / 14x14 maps): one block per (segment, 32 channels) walks all
// rows of the segment and finishes the sum itself -- no partials, no second launch.
template <typename T>
__global__ __launch_bounds__(256) void colsum_small_k(const T* __restrict__ a, int lda, const T* __restrict__ b, int ldb,
                                                      int rows_per_seg, int C, float scale, float* __restrict__ out, int accumulate) {
  __shared__ float4 sm[256];
  const int t = threadIdx.x, q = t & 7, rl = t >> 3;
  const int c = blockIdx.x * 32 + q * 4;
  const int seg = blockIdx.y;
  float4 acc = f4zero();
  if (c < C) {
    const long long r0 = (long long)seg * rows_per_seg;
    constexpr int U = 8;
    int r = rl;
    for (; r + (U - 1) * 32 < rows_per_seg; r += U * 32) {
      float4 va[U], vb[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        va[u] = ldq(a + (r0 + r + u * 32) * lda + c);
        vb[u] = b != nullptr ? ldq(b + (r0 + r + u * 32) * ldb + c) : make_float4(1.f, 1.f, 1.f, 1.f);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) acc = f4add(acc, f4mul(va[u], vb[u]));
    }
    for (; r < rows_per_seg; r += 32) {
      float4 v = ldq(a + (r0 + r) * lda + c);
      if (b != nullptr) v = f4mul(v, ldq(b + (r0 + r) * ldb + c));
      acc = f4add(acc, v);
    }
  }
  sm[t] = acc;
  __syncthreads();
  if (rl == 0 && c < C) {
    float4 s4 = sm[q];
    for (int j = 1; j < 32; ++j) s4 = f4add(s4, sm[j * 8 + q]);
    s4 = f4scale(s4, scale);
    float* dst = out + (long long)seg * C + c;
    if (accumulate) s4 = f4add(s4, ld4(dst));
    st4(dst, s4);
  }
}

__global__ __launch_bounds__(256) void fold_flat_k(const float* __restrict__ part, int nblk, long long total, float scale,
                                                   float* __restrict__ out, int accumulate, long long seg_len, long long seg_stride,
                                                   long long seg_off) {
  __shared__ double sm[kFoldY * (kFoldX + 1)];
  const long long i = (long long)blockIdx.x * kFoldX + threadIdx.x;
  const bool ok = i < total;
  const double s = fold_partials(part, nblk, total, i, ok, sm);
  if (!ok || threadIdx.y != 0) return;
  const float r = (float)(s * (double)scale);
  const long long o = (i / seg_len) * seg_stride + seg_off + i % seg_len;
  out[o] = accumulate ? out[o] + r : r;
}

static inline int ew_grid(long long total_quads) {
  long long b = (total_quads + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace mliis

using namespace mliis;

extern "C" {

size_t mliis_colreduce_workspace_floats(long long rows_per_seg, int C, int nseg, int nv) {
  if (rows_per_seg <= 0 || C <= 0 || (C & 3) || nseg <= 0 || nv <= 0) return 0;
  ColGeom g = col_geom(rows_per_seg, C, nseg);
  return (size_t)nseg * g.nblk * nv * C;
}

int mliis_bn_stats(const float* x, int ldx, long long rows, int C, int pre_swish, float eps, float momentum,
                   int unbiased_moving_var, float* mean, float* rstd, float* moving_mean, float* moving_var, float* ws,
                   size_t ws_floats, hipStream_t stream) {
  MLIIS_REQUIRE(x && mean && rstd && ws, MLIIS_ERR_ARG, "bn_stats: null pointer");
  MLIIS_REQUIRE(rows > 1 && C > 0 && (C & 3) == 0 && (ldx & 3) == 0 && ldx >= C, MLIIS_ERR_ARG,
                "bn_stats: bad shape rows=%lld C=%d ld=%d", rows, C, ldx);
  MLIIS_REQUIRE(aligned16(x) && aligned16(ws), MLIIS_ERR_ALIGN, "bn_stats: pointers must be 16-byte aligned");
  MLIIS_REQUIRE((moving_mean == nullptr) == (moving_var == nullptr), MLIIS_ERR_ARG, "bn_stats: moving stats must come as a pair");
  StatsOp op{x, ldx, pre_swish};
  ColGeom g;
  int rc = launch_colreduce(op, rows, C, 1, ws, ws_floats, stream, &g, "bn_stats");
  if (rc) return rc;
  double n = (double)rows;
  float factor = unbiased_moving_var ? (float)(n / (n - 1.0)) : 1.0f;
  hipLaunchKernelGGL(bn_stats_finalize_k, dim3(ceil_div(C, kFoldX)), dim3(kFoldX, kFoldY), 0, stream, ws, g.nblk, C, 1.0 / n, eps,
                     (float)(1.0 - (double)momentum), factor, mean, rstd, moving_mean, moving_var);
  MLIIS_CHECK_LAUNCH("bn_stats_finalize");
  return MLIIS_OK;
}

int mliis_bn_apply(const float* x, int ldx, float* y, int ldy, long long rows, int C, int rows_per_img, const float* mean,
                   const float* rstd, const float* gamma, const float* beta, int pre_swish, int post_swish,
                   const float* img_scale, const float* res, int ldr, hipStream_t stream) {
  MLIIS_REQUIRE(x && y && mean && rstd && gamma && beta, MLIIS_ERR_ARG, "bn_apply: null pointer");
  MLIIS_REQUIRE(rows > 0 && C > 0 && (C & 3) == 0 && (ldx & 3) == 0 && (ldy & 3) == 0 && ldx >= C && ldy >= C &&
                    rows_per_img > 0 && (res == nullptr || ((ldr & 3) == 0 && ldr >= C)),
                MLIIS_ERR_ARG, "bn_apply: bad shape");
  MLIIS_REQUIRE(aligned16(x) && aligned16(y) && aligned16(mean) && aligned16(rstd) && aligned16(gamma) && aligned16(beta) &&
                    aligned16(res),
                MLIIS_ERR_ALIGN, "bn_apply: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(bn_apply_k, dim3(ew_grid(rows * (C / 4))), dim3(256), 0, stream, x, ldx, y, ldy, rows, C, rows_per_img,
                     mean, rstd, gamma, beta, pre_swish, post_swish, img_scale, res, ldr);
  MLIIS_CHECK_LAUNCH("bn_apply");
  return MLIIS_OK;
}

int mliis_bn_bwd(const float* x, int ldx, const float* dy, int lddy, float* dx, int lddx, long long rows, int C,
                 int rows_per_img, const float* mean, const float* rstd, const float* gamma, const float* beta, int pre_swish,
                 int post_swish, const float* img_scale, const float* chan_scale, const float* chan_add, float* dgamma,
                 float* dbeta, float* dskip, int lddskip, int dskip_accumulate, float* dxsum_part, size_t dxsum_floats, float* ws,
                 size_t ws_floats, const float* stage1_part, int stage1_nblk, int act_dtype, hipStream_t stream) {
  MLIIS_REQUIRE(x && dy && dx && mean && rstd && gamma && beta && dgamma && dbeta && ws, MLIIS_ERR_ARG, "bn_bwd: null pointer");
  MLIIS_REQUIRE(act_dtype == MLIIS_DT_F32 || act_dtype == MLIIS_DT_BF16, MLIIS_ERR_ARG, "bn_bwd: act_dtype must be MLIIS_DT_F32 or MLIIS_DT_BF16");
  if (act_dtype == MLIIS_DT_BF16) {   // x, dy, dx are bf16 tensors (an expanded MBConv tensor and its gradient)
    MLIIS_REQUIRE(dskip == nullptr && dxsum_part == nullptr, MLIIS_ERR_UNSUPPORTED, "bn_bwd: bf16 tensors take no skip / column-sum outputs");
    MLIIS_REQUIRE(rows > 1 && rows < (1LL << 31) && C > 0 && (C & 3) == 0 && (ldx & 3) == 0 && (lddy & 3) == 0 && (lddx & 3) == 0 && ldx >= C &&
                      lddy >= C && lddx >= C && rows_per_img > 0 && aligned16(x) && aligned16(dy) && aligned16(dx) && aligned16(stage1_part) &&
                      aligned16(ws) && (stage1_part == nullptr || stage1_nblk > 0),
                  MLIIS_ERR_ARG, "bn_bwd: bad shape / alignment");
    int gx_, gy_, rpb_;
    bn_bwd_grid(rows, C, &gx_, &gy_, &rpb_);
    const bf16s *xb = reinterpret_cast<const bf16s*>(x), *gb = reinterpret_cast<const bf16s*>(dy);
    bf16s* db = reinterpret_cast<bf16s*>(dx);
    const bool se_ = img_scale != nullptr || chan_scale != nullptr || chan_add != nullptr;
    if (stage1_part == nullptr) {   // no producer left stage 1 (the stride-2 layer that lands on a small map): the reduce pass, on bf16 tensors
      ColGeom g_;
      int rc_;
      if (se_) {
        BnBwdCommon<true, bf16s> p{xb, ldx, gb, lddy, rows_per_img, C, mean, rstd, gamma, beta, pre_swish, post_swish, img_scale, chan_scale, chan_add};
        rc_ = launch_colreduce(BnBwdOp<true, bf16s>{p}, rows, C, 1, ws, ws_floats, stream, &g_, "bn_bwd", bn_bwd_target(rows, C));
      } else {
        BnBwdCommon<false, bf16s> p{xb, ldx, gb, lddy, rows_per_img, C, mean, rstd, gamma, beta, pre_swish, post_swish, nullptr, nullptr, nullptr};
        rc_ = launch_colreduce(BnBwdOp<false, bf16s>{p}, rows, C, 1, ws, ws_floats, stream, &g_, "bn_bwd", bn_bwd_target(rows, C));
      }
      if (rc_) return rc_;
      stage1_part = ws;
      stage1_nblk = g_.nblk;
    }
    if (se_) {
      BnBwdCommon<true, bf16s> p{xb, ldx, gb, lddy, rows_per_img, C, mean, rstd, gamma, beta, pre_swish, post_swish, img_scale, chan_scale, chan_add};
      hipLaunchKernelGGL((bn_bwd_apply_fused_k<true, bf16s>), dim3(gx_, gy_), dim3(256), 0, stream, p, rows, stage1_part, stage1_nblk,
                         1.0 / (double)rows, dgamma, dbeta, db, lddx, rpb_, SkipOut{nullptr, 0, 0}, (float*)nullptr, BnBwdAlt{});
    } else {
      BnBwdCommon<false, bf16s> p{xb, ldx, gb, lddy, rows_per_img, C, mean, rstd, gamma, beta, pre_swish, post_swish, nullptr, nullptr, nullptr};
      hipLaunchKernelGGL((bn_bwd_apply_fused_k<false, bf16s>), dim3(gx_, gy_), dim3(256), 0, stream, p, rows, stage1_part, stage1_nblk,
                         1.0 / (double)rows, dgamma, dbeta, db, lddx, rpb_, SkipOut{nullptr, 0, 0}, (float*)nullptr, BnBwdAlt{});
    }
    MLIIS_CHECK_LAUNCH("bn_bwd_apply_fused (bf16)");
    return MLIIS_OK;
  }
  MLIIS_REQUIRE(dskip == nullptr || (aligned16(dskip) && (lddskip & 3) == 0 && lddskip >= C && dskip != dx), MLIIS_ERR_ARG,
                "bn_bwd: bad skip-gradient output");
  MLIIS_REQUIRE(aligned16(dgamma) && aligned16(dbeta), MLIIS_ERR_ALIGN, "bn_bwd: dgamma / dbeta must be 16-byte aligned");
  MLIIS_REQUIRE(rows > 1 && rows < (1LL << 31) && C > 0 && (C & 3) == 0 && (ldx & 3) == 0 && (lddy & 3) == 0 && (lddx & 3) == 0 &&
                    ldx >= C && lddy >= C && lddx >= C && rows_per_img > 0,
                MLIIS_ERR_ARG, "bn_bwd: bad shape");
  MLIIS_REQUIRE(aligned16(x) && aligned16(dy) && aligned16(dx) && aligned16(ws) && aligned16(chan_scale) && aligned16(chan_add) &&
                    aligned16(mean) && aligned16(rstd) && aligned16(gamma) && aligned16(beta),
                MLIIS_ERR_ALIGN, "bn_bwd: pointers must be 16-byte aligned");
  SkipOut skip{dskip, lddskip, dskip_accumulate};
  int gx, gy, rpb;
  bn_bwd_grid(rows, C, &gx, &gy, &rpb);
  const int target = bn_bwd_target(rows, C);
  MLIIS_REQUIRE(dxsum_part == nullptr || (aligned16(dxsum_part) && (size_t)gy * C <= dxsum_floats), MLIIS_ERR_WORKSPACE,
                "bn_bwd: column-sum buffer unaligned or too small (%zu floats needed)", (size_t)gy * C);
  ColGeom g;
  int rc;
  if (stage1_part != nullptr) {   // stage 1 ({sum g, sum g * xhat} partials [stage1_nblk][2][C]) came from the producer of dy
    // (the producer computed its sums with the same g: the drop-connect scale -- mliis_conv2d_bwd_data_bn -- or the squeeze-excite
    // vectors -- mliis_se_bn_bwd_sums + mliis_se_mlp_bwd_bn -- are part of it)
    MLIIS_REQUIRE(stage1_nblk > 0 && aligned16(stage1_part), MLIIS_ERR_ARG, "bn_bwd: bad external stage-1 partials");
    if (img_scale != nullptr || chan_scale != nullptr || chan_add != nullptr) {
      BnBwdCommon<true> p{x, ldx, dy, lddy, rows_per_img, C, mean, rstd, gamma, beta, pre_swish, post_swish, img_scale, chan_scale, chan_add};
      hipLaunchKernelGGL(bn_bwd_apply_fused_k<true>, dim3(gx, gy), dim3(256), 0, stream, p, rows, stage1_part, stage1_nblk, 1.0 / (double)rows,
                         dgamma, dbeta, dx, lddx, rpb, skip, dxsum_part, BnBwdAlt{});
    } else {
      BnBwdCommon<false> p{x, ldx, dy, lddy, rows_per_img, C, mean, rstd, gamma, beta, pre_swish, post_swish, nullptr, nullptr, nullptr};
      hipLaunchKernelGGL(bn_bwd_apply_fused_k<false>, dim3(gx, gy), dim3(256), 0, stream, p, rows, stage1_part, stage1_nblk, 1.0 / (double)rows,
                         dgamma, dbeta, dx, lddx, rpb, skip, dxsum_part, BnBwdAlt{});
    }
    MLIIS_CHECK_LAUNCH("bn_bwd_apply_fused");
    return MLIIS_OK;
  }
  if (img_scale != nullptr || chan_scale != nullptr || chan_add != nullptr) {
    BnBwdCommon<true> p{x, ldx, dy, lddy, rows_per_img, C, mean, rstd, gamma, beta, pre_swish, post_swish, img_scale, chan_scale, chan_add};
    rc = launch_colreduce(BnBwdOp<true>{p}, rows, C, 1, ws, ws_floats, stream, &g, "bn_bwd", target);
    if (rc) return rc;
    hipLaunchKernelGGL(bn_bwd_apply_fused_k<true>, dim3(gx, gy), dim3(256), 0, stream, p, rows, ws, g.nblk, 1.0 / (double)rows, dgamma, dbeta,
                       dx, lddx, rpb, skip, dxsum_part, BnBwdAlt{});
  } else {
    BnBwdCommon<false> p{x, ldx, dy, lddy, rows_per_img, C, mean, rstd, gamma, beta, pre_swish, post_swish, nullptr, nullptr, nullptr};
    rc = launch_colreduce(BnBwdOp<false>{p}, rows, C, 1, ws, ws_floats, stream, &g, "bn_bwd", target);
    if (rc) return rc;
    hipLaunchKernelGGL(bn_bwd_apply_fused_k<false>, dim3(gx, gy), dim3(256), 0, stream, p, rows, ws, g.nblk, 1.0 / (double)rows, dgamma, dbeta,
                       dx, lddx, rpb, skip, dxsum_part, BnBwdAlt{});
  }
  MLIIS_CHECK_LAUNCH("bn_bwd_apply_fused");
  return MLIIS_OK;
}

// part [N][*nblk][5][C]: per image and row chunk, the five sums of SeBnOp over (dy = da2, x = z1).
int mliis_se_bn_bwd_sums(const float* x, int ldx, const float* dy, int lddy, int N, int rows_per_img, int C, const float* mean,
                         const float* rstd, const float* gamma, const float* beta, float* part, size_t part_floats, int* nblk,
                         int act_dtype, hipStream_t stream) {
  MLIIS_REQUIRE(x && dy && mean && rstd && gamma && beta && part && nblk, MLIIS_ERR_ARG, "se_bn_bwd_sums: null pointer");
  MLIIS_REQUIRE(act_dtype == MLIIS_DT_F32 || act_dtype == MLIIS_DT_BF16, MLIIS_ERR_ARG, "se_bn_bwd_sums: bad act_dtype");
  MLIIS_REQUIRE(N > 0 && rows_per_img > 0 && C > 0 && (C & 3) == 0 && (ldx & 3) == 0 && (lddy & 3) == 0 && ldx >= C && lddy >= C &&
                    (long long)N * rows_per_img < (1LL << 31),
                MLIIS_ERR_ARG, "se_bn_bwd_sums: bad shape");
  MLIIS_REQUIRE(aligned16(x) && aligned16(dy) && aligned16(mean) && aligned16(rstd) && aligned16(gamma) && aligned16(beta) && aligned16(part),
                MLIIS_ERR_ALIGN, "se_bn_bwd_sums: pointers must be 16-byte aligned");
  ColGeom g;
  // (about two workgroups per CU: the squeeze-excite backward folds an image's chunks itself)
  int rc;
  if (act_dtype == MLIIS_DT_BF16) {
    SeBnOp<bf16s> op{reinterpret_cast<const bf16s*>(x), ldx, reinterpret_cast<const bf16s*>(dy), lddy, mean, rstd, gamma, beta};
    rc = launch_colreduce(op, rows_per_img, C, N, part, part_floats, stream, &g, "se_bn_bwd_sums", 512);
  } else {
    SeBnOp<float> op{x, ldx, dy, lddy, mean, rstd, gamma, beta};
    rc = launch_colreduce(op, rows_per_img, C, N, part, part_floats, stream, &g, "se_bn_bwd_sums", 512);
  }
  if (rc) return rc;
  *nblk = g.nblk;
  return MLIIS_OK;
}

size_t mliis_se_bn_bwd_sums_floats(int N, int rows_per_img, int C) {
  if (N <= 0 || rows_per_img <= 0 || C <= 0 || (C & 3)) return 0;
  const ColGeom g = col_geom(rows_per_img, C, N, 512);
  return (size_t)N * g.nblk * 5 * C;
}

// Stage 1 only: per-block column sums {sum x, sum x^2} (of swish(x) when pre_swish) into part [nblk][2][C]; *nblk_out = nblk.
// Used when the producer of x could not emit the statistics itself (depthwise / stem outputs).
int mliis_bn_stats_partial(const float* x, int ldx, long long rows, int C, int pre_swish, float* part, size_t part_floats, int* nblk_out,
                           hipStream_t stream) {
  MLIIS_REQUIRE(x && part && nblk_out, MLIIS_ERR_ARG, "bn_stats_partial: null pointer");
  MLIIS_REQUIRE(rows > 1 && C > 0 && (C & 3) == 0 && (ldx & 3) == 0 && ldx >= C, MLIIS_ERR_ARG, "bn_stats_partial: bad shape");
  MLIIS_REQUIRE(aligned16(x) && aligned16(part), MLIIS_ERR_ALIGN, "bn_stats_partial: pointers must be 16-byte aligned");
  StatsOp op{x, ldx, pre_swish};
  ColGeom g;
  // (fewer, taller blocks than the other column reductions: every workgroup of the consumer folds ALL these partials)
  int rc = launch_colreduce(op, rows, C, 1, part, part_floats, stream, &g, "bn_stats_partial", kStatsPartialTarget);
  if (rc) return rc;
  *nblk_out = g.nblk;
  return MLIIS_OK;
}

// Folds partial statistics (part [nblk][2][C], from mliis_bn_stats_partial or from mliis_conv2d_fwd's fused epilogue), writes
// mean / rstd (kept for the backward pass), updates the moving averages and applies
// y = [swish](gamma * ([swish](x) - mean) * rstd + beta) * img_scale[n] + res      -- ONE launch.
int mliis_bn_apply_fused(const float* x, int ldx, float* y, int ldy, long long rows, int C, int rows_per_img, const float* part, int nblk,
                         float eps, float momentum, int unbiased_moving_var, float* mean, float* rstd, float* moving_mean,
                         float* moving_var, const float* gamma, const float* beta, int pre_swish, int post_swish, const float* img_scale,
                         const float* res, int ldr, float* pool_part, size_t pool_floats, int* pool_chunks, int act_dtype, hipStream_t stream) {
  MLIIS_REQUIRE(x && y && part && mean && rstd && gamma && beta, MLIIS_ERR_ARG, "bn_apply_fused: null pointer");
  MLIIS_REQUIRE(act_dtype == MLIIS_DT_F32 || act_dtype == MLIIS_DT_BF16, MLIIS_ERR_ARG, "bn_apply_fused: bad act_dtype");
  MLIIS_REQUIRE(rows > 1 && rows < (1LL << 31) && nblk > 0 && C > 0 && (C & 3) == 0 && (ldx & 3) == 0 && (ldy & 3) == 0 && ldx >= C && ldy >= C &&
                    rows_per_img > 0 && (res == nullptr || ((ldr & 3) == 0 && ldr >= C)),
                MLIIS_ERR_ARG, "bn_apply_fused: bad shape");
  MLIIS_REQUIRE(aligned16(x) && aligned16(y) && aligned16(gamma) && aligned16(beta) && aligned16(res), MLIIS_ERR_ALIGN,
                "bn_apply_fused: pointers must be 16-byte aligned");
  MLIIS_REQUIRE((moving_mean == nullptr) == (moving_var == nullptr), MLIIS_ERR_ARG, "bn_apply_fused: moving stats must come as a pair");
  const double n = (double)rows;
  BnFold f{part, nblk, 1.0 / n, eps, (float)(1.0 - (double)momentum), unbiased_moving_var ? (float)(n / (n - 1.0)) : 1.0f, mean, rstd,
           moving_mean, moving_var};
  int gx, gy, rpb, cpi = 0;
  chan_grid(rows, C, &gx, &gy, &rpb);
  // Every block folds all nblk statistics partials of its channels (the price of one launch instead of three).  With many partials
  // (the 112x112 / 56x56 producers hand over 196-1568) that redundant L2 traffic outweighs the payload at 128 rows per block:
  // four times taller row chunks then (tools/bn_probe.py, 784 partials: 28.9 -> 23.7 us on 77 MB, 21.8 -> 17.0 us on 28 MB).
  const bool many_partials = nblk >= kManyPartials;
  if (many_partials) {
    const ColGeom g2 = col_geom(rows, C, 1, 512);
    gx = g2.gx;
    gy = g2.nblk;
    rpb = g2.rows_per_block;
  }
  fit_one_round(rows, gx, 1, &gy, &rpb);
  if (pool_part != nullptr) {   // per-image pooling of the output: row chunks are cut per image (about 128 rows each)
    MLIIS_REQUIRE(pool_chunks && aligned16(pool_part) && rows % rows_per_img == 0, MLIIS_ERR_ARG,
                  "bn_apply_fused: pooling needs an aligned buffer, a pool_chunks output and whole images");
    const long long nimg = rows / rows_per_img;
    int pool_rows = many_partials ? 4 * kPoolRows : kPoolRows;
    while ((long long)gx * nimg * ((rows_per_img + pool_rows - 1) / pool_rows) > bn_max_blocks() && pool_rows < rows_per_img) pool_rows *= 2;
    cpi = (rows_per_img + pool_rows - 1) / pool_rows;
    rpb = (rows_per_img + cpi - 1) / cpi;
    gy = (int)(nimg * cpi);
    MLIIS_REQUIRE((size_t)gy * C <= pool_floats, MLIIS_ERR_WORKSPACE, "bn_apply_fused: pool buffer too small (%zu floats needed, %zu given)",
                  (size_t)gy * C, pool_floats);
    *pool_chunks = cpi;
  }
#define BN_APPLY(TX_, TY_, RES_, ISC_)                                                                                                        \
  hipLaunchKernelGGL((bn_apply_fused_k<TX_, TY_, RES_, ISC_>), dim3(gx, gy), dim3(256), 0, stream, reinterpret_cast<const TX_*>(x), ldx,         \
                     reinterpret_cast<TY_*>(y), ldy, rows, C, rows_per_img, f, gamma, beta, pre_swish, post_swish, img_scale, res, ldr, rpb,   \
                     pool_part, cpi, BnApplyAlt{})
  const bool has_res = res != nullptr, has_isc = img_scale != nullptr;
  if (act_dtype == MLIIS_DT_BF16) {   // x and y are bf16 tensors (z1 -> a1 of an MBConv block)
    if (!has_res && !has_isc) BN_APPLY(bf16s, bf16s, false, false);
    else BN_APPLY(bf16s, bf16s, true, true);
  } else if (!has_res && !has_isc) {
    BN_APPLY(float, float, false, false);
  } else if (has_res && !has_isc) {
    BN_APPLY(float, float, true, false);
  } else {
    BN_APPLY(float, float, true, true);
  }
#undef BN_APPLY
  MLIIS_CHECK_LAUNCH("bn_apply_fused");
  return MLIIS_OK;
}

// Two batch norms of the same shape, flags and leading dimensions in ONE launch (the 1x1 and the 3x3-dilated branch of an RSD module
// are independent: efficientlab.py:185-197).  Arguments as mliis_bn_apply_fused, per problem.
int mliis_bn_apply_fused_pair(const float* x0, float* y0, const float* part0, int nblk0, float* mean0, float* rstd0, float* moving_mean0,
                              float* moving_var0, const float* gamma0, const float* beta0, const float* x1, float* y1, const float* part1,
                              int nblk1, float* mean1, float* rstd1, float* moving_mean1, float* moving_var1, const float* gamma1,
                              const float* beta1, int ldx, int ldy, long long rows, int C, float eps, float momentum,
                              int unbiased_moving_var, int pre_swish, int post_swish, hipStream_t stream) {
  MLIIS_REQUIRE(x0 && y0 && part0 && mean0 && rstd0 && gamma0 && beta0 && x1 && y1 && part1 && mean1 && rstd1 && gamma1 && beta1, MLIIS_ERR_ARG,
                "bn_apply_fused_pair: null pointer");
  MLIIS_REQUIRE(rows > 1 && rows < (1LL << 31) && nblk0 > 0 && nblk1 > 0 && C > 0 && (C & 3) == 0 && (ldx & 3) == 0 && (ldy & 3) == 0 && ldx >= C &&
                    ldy >= C,
                MLIIS_ERR_ARG, "bn_apply_fused_pair: bad shape");
  MLIIS_REQUIRE(aligned16(x0) && aligned16(y0) && aligned16(gamma0) && aligned16(beta0) && aligned16(x1) && aligned16(y1) && aligned16(gamma1) &&
                    aligned16(beta1) && aligned16(part0) && aligned16(part1),
                MLIIS_ERR_ALIGN, "bn_apply_fused_pair: pointers must be 16-byte aligned");
  MLIIS_REQUIRE((moving_mean0 == nullptr) == (moving_var0 == nullptr) && (moving_mean1 == nullptr) == (moving_var1 == nullptr), MLIIS_ERR_ARG,
                "bn_apply_fused_pair: moving stats must come as a pair");
  const double n = (double)rows;
  const float omm = (float)(1.0 - (double)momentum), evf = unbiased_moving_var ? (float)(n / (n - 1.0)) : 1.0f;
  BnFold f0{part0, nblk0, 1.0 / n, eps, omm, evf, mean0, rstd0, moving_mean0, moving_var0};
  BnFold f1{part1, nblk1, 1.0 / n, eps, omm, evf, mean1, rstd1, moving_mean1, moving_var1};
  int gx, gy, rpb;
  chan_grid(rows, C, &gx, &gy, &rpb);
  if ((nblk0 > nblk1 ? nblk0 : nblk1) >= kManyPartials) {
    const ColGeom g2 = col_geom(rows, C, 1, 512);
    gx = g2.gx;
    gy = g2.nblk;
    rpb = g2.rows_per_block;
  }
  fit_one_round(rows, gx, 2, &gy, &rpb);
  hipLaunchKernelGGL((bn_apply_fused_k<float, float, false, false>), dim3(gx, gy, 2), dim3(256), 0, stream, x0, ldx, y0, ldy, rows, C, (int)rows, f0, gamma0, beta0, pre_swish,
                     post_swish, nullptr, nullptr, 0, rpb, nullptr, 0, BnApplyAlt{x1, y1, f1, gamma1, beta1});
  MLIIS_CHECK_LAUNCH("bn_apply_fused_pair");
  return MLIIS_OK;
}

// The backward of two such batch norms (no per-image vectors, no skip output): ONE reduce launch and ONE apply launch for both.
// dxsum0 / dxsum1 (nullable pair): per-row-chunk column sums of dx, as mliis_bn_bwd's dxsum_part.  ws: 2 x
// mliis_colreduce_workspace_floats(rows, C, 1, 2) floats.
int mliis_bn_bwd_pair(const float* x0, const float* dy0, float* dx0, const float* mean0, const float* rstd0, const float* gamma0,
                      const float* beta0, float* dgamma0, float* dbeta0, float* dxsum0, const float* x1, const float* dy1, float* dx1,
                      const float* mean1, const float* rstd1, const float* gamma1, const float* beta1, float* dgamma1, float* dbeta1,
                      float* dxsum1, int ldx, int lddy, int lddx, long long rows, int C, int pre_swish, int post_swish, size_t dxsum_floats,
                      float* ws, size_t ws_floats, hipStream_t stream) {
  MLIIS_REQUIRE(x0 && dy0 && dx0 && mean0 && rstd0 && gamma0 && beta0 && dgamma0 && dbeta0 && x1 && dy1 && dx1 && mean1 && rstd1 && gamma1 && beta1 &&
                    dgamma1 && dbeta1 && ws,
                MLIIS_ERR_ARG, "bn_bwd_pair: null pointer");
  MLIIS_REQUIRE(rows > 1 && rows < (1LL << 30) && C > 0 && (C & 3) == 0 && (ldx & 3) == 0 && (lddy & 3) == 0 && (lddx & 3) == 0 && ldx >= C &&
                    lddy >= C && lddx >= C && (dxsum0 == nullptr) == (dxsum1 == nullptr),
                MLIIS_ERR_ARG, "bn_bwd_pair: bad shape");
  MLIIS_REQUIRE(aligned16(x0) && aligned16(dy0) && aligned16(dx0) && aligned16(x1) && aligned16(dy1) && aligned16(dx1) && aligned16(ws) &&
                    aligned16(mean0) && aligned16(rstd0) && aligned16(gamma0) && aligned16(beta0) && aligned16(mean1) && aligned16(rstd1) &&
                    aligned16(gamma1) && aligned16(beta1) && aligned16(dgamma0) && aligned16(dbeta0) && aligned16(dgamma1) && aligned16(dbeta1) &&
                    aligned16(dxsum0) && aligned16(dxsum1),
                MLIIS_ERR_ALIGN, "bn_bwd_pair: pointers must be 16-byte aligned");
  int gx, gy, rpb;
  bn_bwd_grid(rows, C, &gx, &gy, &rpb);
  MLIIS_REQUIRE(dxsum0 == nullptr || (size_t)gy * C <= dxsum_floats, MLIIS_ERR_WORKSPACE, "bn_bwd_pair: column-sum buffers too small (%zu floats each)",
                (size_t)gy * C);
  BnBwdCommon<false> p0{x0, ldx, dy0, lddy, (int)rows, C, mean0, rstd0, gamma0, beta0, pre_swish, post_swish, nullptr, nullptr, nullptr};
  BnBwdCommon<false> p1{x1, ldx, dy1, lddy, (int)rows, C, mean1, rstd1, gamma1, beta1, pre_swish, post_swish, nullptr, nullptr, nullptr};
  ColGeom g;
  int rc = launch_colreduce(BnBwdPairOp{{p0, p1}, rows}, rows, C, 2, ws, ws_floats, stream, &g, "bn_bwd_pair", bn_bwd_target(rows, C));
  if (rc) return rc;
  const float* part1 = ws + (size_t)g.nblk * 2 * C;   // part layout [seg][blk][2][C]
  hipLaunchKernelGGL(bn_bwd_apply_fused_k<false>, dim3(gx, gy, 2), dim3(256), 0, stream, p0, rows, ws, g.nblk, 1.0 / (double)rows, dgamma0, dbeta0,
                     dx0, lddx, rpb, SkipOut{nullptr, 0, 0}, dxsum0,
                     BnBwdAlt{x1, dy1, mean1, rstd1, gamma1, beta1, part1, dgamma1, dbeta1, dx1, dxsum1});
  MLIIS_CHECK_LAUNCH("bn_bwd_pair");
  return MLIIS_OK;
}

int mliis_fold_tile_outputs(void) { return kFoldTile; }

// floats of the dxsum_part output of mliis_bn_bwd: [row chunks][C] (the number of row chunks is its slab count for mliis_fold_batched)
size_t mliis_bn_bwd_dxsum_floats(long long rows, int C) {
  if (rows <= 0 || C <= 0 || (C & 3)) return 0;
  int gx, gy, rpb;
  bn_bwd_grid(rows, C, &gx, &gy, &rpb);
  return (size_t)gy * C;
}

int mliis_fold_batched(const float* part_base, float* out_base, const long long* desc, int ndesc, long long total_tiles,
                       const long long* se_desc, int n_se, long long se_tiles, hipStream_t stream) {
  MLIIS_REQUIRE(part_base && out_base && desc && ndesc > 0 && total_tiles > 0, MLIIS_ERR_ARG, "fold_batched: bad arguments");
  MLIIS_REQUIRE(aligned16(part_base) && aligned16(out_base), MLIIS_ERR_ALIGN, "fold_batched: bases must be 16-byte aligned");
  MLIIS_REQUIRE((se_desc == nullptr && n_se == 0 && se_tiles == 0) || (se_desc != nullptr && n_se > 0 && se_tiles > 0), MLIIS_ERR_ARG,
                "fold_batched: the squeeze-excite table comes with its row and tile counts (or NULL, 0, 0)");
  hipLaunchKernelGGL(fold_batched_k, dim3((unsigned)(total_tiles + se_tiles)), dim3(256), 0, stream, part_base, out_base, desc, ndesc, total_tiles,
                     se_desc, n_se);
  MLIIS_CHECK_LAUNCH("fold_batched");
  return MLIIS_OK;
}

// out[seg, c] (+)= scale * sum_{rows of seg} a[row,c] * (b[row,c] if b)
int mliis_colsum(const float* a, int lda, const float* b, int ldb, long long rows_per_seg, int nseg, int C, float scale,
                 float* out, int accumulate, float* ws, size_t ws_floats, int ab_dtype, hipStream_t stream) {
  MLIIS_REQUIRE(a && out && ws, MLIIS_ERR_ARG, "colsum: null pointer");
  MLIIS_REQUIRE(ab_dtype == MLIIS_DT_F32 || ab_dtype == MLIIS_DT_BF16, MLIIS_ERR_ARG, "colsum: bad ab_dtype");
  const bool bf = ab_dtype == MLIIS_DT_BF16;   // a and b are bf16 tensors (da2 and a1 of an MBConv block)
  const bf16s *ab_ = reinterpret_cast<const bf16s*>(a), *bb_ = reinterpret_cast<const bf16s*>(b);
  MLIIS_REQUIRE(rows_per_seg > 0 && nseg > 0 && C > 0 && (C & 3) == 0 && (lda & 3) == 0 && lda >= C &&
                    (b == nullptr || ((ldb & 3) == 0 && ldb >= C)),
                MLIIS_ERR_ARG, "colsum: bad shape");
  MLIIS_REQUIRE(aligned16(a) && aligned16(b) && aligned16(ws), MLIIS_ERR_ALIGN, "colsum: pointers must be 16-byte aligned");
  if (rows_per_seg <= 1024 && aligned16(out)) {
    if (bf) hipLaunchKernelGGL(colsum_small_k<bf16s>, dim3(ceil_div(C, 32), nseg), dim3(256), 0, stream, ab_, lda, bb_, ldb, (int)rows_per_seg, C, scale, out, accumulate);
    else hipLaunchKernelGGL(colsum_small_k<float>, dim3(ceil_div(C, 32), nseg), dim3(256), 0, stream, a, lda, b, ldb, (int)rows_per_seg, C, scale, out, accumulate);
    MLIIS_CHECK_LAUNCH("colsum_small");
    return MLIIS_OK;
  }
  ColGeom g;
  int rc = bf ? launch_colreduce(SumOp<bf16s>{ab_, lda, bb_, ldb}, rows_per_seg, C, nseg, ws, ws_floats, stream, &g, "colsum")
              : launch_colreduce(SumOp<float>{a, lda, b, ldb}, rows_per_seg, C, nseg, ws, ws_floats, stream, &g, "colsum");
  if (rc) return rc;
  hipLaunchKernelGGL(sum_finalize_k, dim3(ceil_div(C, kFoldX), 1, nseg), dim3(kFoldX, kFoldY), 0, stream, ws, g.nblk, 1, C, nseg, scale, out,
                     accumulate);
  MLIIS_CHECK_LAUNCH("colsum_finalize");
  return MLIIS_OK;
}

// dw[c, j] = sum_rows x[row,c]*mask[row,c]*dy[row,j]  (j in {0,1});  dw is [C,2] row-major (TF HWIO of a 1x1 conv)
int mliis_final_conv_bwd_filter(const float* x, int ldx, const float* mask, const float* dy, long long rows, int C, float* dw,
                                float* db, float* ws, size_t ws_floats, hipStream_t stream);
}

namespace mliis {
// blockIdx.y < 2: weight-gradient column j of 16 channels (fold of the Outer2Op partials); blockIdx.y == 2 (one workgroup): the bias
// gradient db[j] = sum_rows dy[row, j] -- independent of the fold, so it rides in the same launch (it used to be one of its own).
__global__ __launch_bounds__(256) void final_dw_finalize_k(const float* __restrict__ part, int nblk, int C, float* __restrict__ dw,
                                                           const float* __restrict__ dy, long long rows, float* __restrict__ db) {
  __shared__ double sm[kFoldY * (kFoldX + 1)];
  if (blockIdx.y == 2) {   // (uniform)
    if (blockIdx.x != 0) return;
    __shared__ double sb[2][256];
    const int t = threadIdx.y * kFoldX + threadIdx.x;
    double a0 = 0.0, a1 = 0.0;
    const long long pairs = rows >> 1;
    for (long long r = t; r < pairs; r += 16 * 256) {   // two rows per float4 load, sixteen loads in flight (one workgroup walks the whole
                                                          // tensor: at four per trip the 12 dependent round trips were the launch's 8.7 us)
      float4 d[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) d[u] = r + u * 256 < pairs ? ld4(dy + (r + u * 256) * 4) : f4zero();
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        a0 += (double)d[u].x + (double)d[u].z;
        a1 += (double)d[u].y + (double)d[u].w;
      }
    }
    if ((rows & 1) && t == 0) {
      a0 += dy[(rows - 1) * 2];
      a1 += dy[(rows - 1) * 2 + 1];
    }
    sb[0][t] = a0;
    sb[1][t] = a1;
    __syncthreads();
    for (int s_ = 128; s_ > 0; s_ >>= 1) {
      if (t < s_) {
        sb[0][t] += sb[0][t + s_];
        sb[1][t] += sb[1][t + s_];
      }
      __syncthreads();
    }
    if (t == 0) {
      db[0] = (float)sb[0][0];
      db[1] = (float)sb[1][0];
    }
    return;
  }
  const int c = blockIdx.x * kFoldX + threadIdx.x;
  const int j = blockIdx.y;
  const bool ok = c < C;
  const double s = fold_partials(part, nblk, 2LL * C, (long long)j * C + c, ok, sm);
  if (ok && threadIdx.y == 0) dw[c * 2 + j] = (float)s;
}
}  // namespace mliis

extern "C" int mliis_final_conv_bwd_filter(const float* x, int ldx, const float* mask, const float* dy, long long rows, int C,
                                           float* dw, float* db, float* ws, size_t ws_floats, hipStream_t stream) {
  MLIIS_REQUIRE(x && dy && dw && db && ws, MLIIS_ERR_ARG, "final_conv_bwd_filter: null pointer");
  MLIIS_REQUIRE(rows > 0 && C > 0 && (C & 3) == 0 && (ldx & 3) == 0 && ldx >= C, MLIIS_ERR_ARG, "final_conv_bwd_filter: bad shape");
  MLIIS_REQUIRE(aligned16(x) && aligned16(mask) && aligned16(ws) && aligned16(dy), MLIIS_ERR_ALIGN, "final_conv_bwd_filter: alignment");
  Outer2Op op{x, ldx, mask, dy};
  ColGeom g;
  int rc = launch_colreduce(op, rows, C, 1, ws, ws_floats, stream, &g, "final_conv_bwd_filter");
  if (rc) return rc;
  hipLaunchKernelGGL(final_dw_finalize_k, dim3(ceil_div(C, kFoldX), 3), dim3(kFoldX, kFoldY), 0, stream, ws, g.nblk, C, dw, dy, rows, db);
  MLIIS_CHECK_LAUNCH("final_dw_finalize");
  return MLIIS_OK;
}
