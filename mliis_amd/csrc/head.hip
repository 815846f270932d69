// Decoder tail: bilinear resize (align_corners=True) forward / backward, final 1x1 conv (Cout = 2, + bias, optional
// dropout mask) forward / backward-data, and the fused per-pixel softmax cross-entropy (+ label smoothing, + soft-IoU "dice"
// term) forward/backward with the thresholded prediction mask.
// Reference: models/efficientlab.py:161-177 (dropout -> 1x1 -> resize -> softmax -> >0.5), :205-206 (RSD upsample),
// :294-303,319-327,329-396 (loss).  All HBM-bound, all deterministic (gather-form backward, two-stage reductions).
#include "common.hpp"

namespace mliis {

template <int V>
struct Vec;
template <>
struct Vec<4> {
  typedef float4 T;
};
template <>
struct Vec<2> {
  typedef float2 T;
};
__device__ __forceinline__ float4 vfma(float s, float4 a, float4 c) {
  return make_float4(fmaf(s, a.x, c.x), fmaf(s, a.y, c.y), fmaf(s, a.z, c.z), fmaf(s, a.w, c.w));
}
__device__ __forceinline__ float2 vfma(float s, float2 a, float2 c) { return make_float2(fmaf(s, a.x, c.x), fmaf(s, a.y, c.y)); }
template <class T>
__device__ __forceinline__ T vzero();
template <>
__device__ __forceinline__ float4 vzero<float4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }
template <>
__device__ __forceinline__ float2 vzero<float2>() { return make_float2(0.f, 0.f); }

template <int V>
__global__ __launch_bounds__(256) void resize_fwd_k(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int N,
                                                    int Hi, int Wi, int Ho, int Wo, int C, float sh, float sw) {
  typedef typename Vec<V>::T T;
  const int Q = C / V;
  const long long total = (long long)N * Ho * Wo * Q;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Q) * V;
    long long p = i / Q;
    const int wo = (int)(p % Wo);
    long long r = p / Wo;
    const int ho = (int)(r % Ho);
    const int n = (int)(r / Ho);
    int y0, y1, x0, x1;
    float ly, lx;
    src_coord(ho, sh, Hi, y0, y1, ly);
    src_coord(wo, sw, Wi, x0, x1, lx);
    const float* base = x + (long long)n * Hi * Wi * ldx + c;
    const T tl = *reinterpret_cast<const T*>(base + ((long long)y0 * Wi + x0) * ldx);
    const T tr = *reinterpret_cast<const T*>(base + ((long long)y0 * Wi + x1) * ldx);
    const T bl = *reinterpret_cast<const T*>(base + ((long long)y1 * Wi + x0) * ldx);
    const T br = *reinterpret_cast<const T*>(base + ((long long)y1 * Wi + x1) * ldx);
    // top = tl + (tr - tl) * lx ; bottom likewise ; out = top + (bottom - top) * ly   (TF ResizeBilinear form)
    T o = vzero<T>();
    float wtl, wtr, wbl, wbr;
    bilinear_weights(ly, lx, wtl, wtr, wbl, wbr);
    o = vfma(wtl, tl, o);
    o = vfma(wtr, tr, o);
    o = vfma(wbl, bl, o);
    o = vfma(wbr, br, o);
    *reinterpret_cast<T*>(y + p * ldy + c) = o;
  }
}

// gather-form transpose: dx[n,hi,wi] = sum_{ho,wo} wy(ho,hi) * wx(wo,wi) * dy[n,ho,wo]   (atomic-free, deterministic)
// 8 lanes per output element: lane l takes the candidate output rows ho_lo + l, ho_lo + l + 8, ... and walks their candidate columns
// four loads at a time; the lanes are combined with a fixed xor butterfly.  (A 4x upsample touches ~11 x 11 candidates per element:
// one thread walking them serially made this kernel pure latency.)
template <int V>
__global__ __launch_bounds__(256) void resize_bwd_k(const float* __restrict__ dy, int lddy, float* __restrict__ dx, int lddx, int N,
                                                    int Hi, int Wi, int Ho, int Wo, int C, float sh, float sw, int accumulate,
                                                    long long total) {
  typedef typename Vec<V>::T T;
  const int Q = C / V;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int rl = (int)(gid & 7);
  const long long i_ = gid >> 3;
  const bool live = i_ < total;
  const long long i = live ? i_ : 0;   // surplus lanes shadow element 0 (they take part in the shuffles, never store)
  const int c = (int)(i % Q) * V;
  const long long p = i / Q;
  const int wi = (int)(p % Wi);
  const long long r = p / Wi;
  const int hi = (int)(r % Hi);
  const int n = (int)(r / Hi);
  // conservative candidate ranges of output rows/cols that can touch (hi, wi)
  int ho_lo = (int)floorf((float)(hi - 1) / sh) - 1, ho_hi = (int)ceilf((float)(hi + 1) / sh) + 1;
  int wo_lo = (int)floorf((float)(wi - 1) / sw) - 1, wo_hi = (int)ceilf((float)(wi + 1) / sw) + 1;
  if (ho_lo < 0) ho_lo = 0;
  if (wo_lo < 0) wo_lo = 0;
  if (ho_hi > Ho - 1) ho_hi = Ho - 1;
  if (wo_hi > Wo - 1) wo_hi = Wo - 1;
  T acc = vzero<T>();
  const float* base = dy + (long long)n * Ho * Wo * lddy + c;
  for (int ho = ho_lo + rl; ho <= ho_hi; ho += 8) {
    int y0, y1;
    float ly;
    src_coord(ho, sh, Hi, y0, y1, ly);
    const float wy = (y0 == hi ? 1.f - ly : 0.f) + (y1 == hi ? ly : 0.f);
    if (wy == 0.f) continue;
    const float* row = base + (long long)ho * Wo * lddy;
    for (int wo = wo_lo; wo <= wo_hi; wo += 4) {
      T v[4];
      float wgt[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int w_ = wo + u <= wo_hi ? wo + u : wo_hi;
        int x0, x1;
        float lx;
        src_coord(w_, sw, Wi, x0, x1, lx);
        const float wx = (x0 == wi ? 1.f - lx : 0.f) + (x1 == wi ? lx : 0.f);
        wgt[u] = wo + u <= wo_hi ? wy * wx : 0.f;
        v[u] = *reinterpret_cast<const T*>(row + (long long)w_ * lddy);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = vfma(wgt[u], v[u], acc);
    }
  }
  float* a = reinterpret_cast<float*>(&acc);
#pragma unroll
  for (int off = 1; off < 8; off <<= 1)
#pragma unroll
    for (int e = 0; e < V; ++e) a[e] += __shfl_xor(a[e], off);
  if (!live || rl != 0) return;
  T* dst = reinterpret_cast<T*>(dx + p * lddx + c);
  if (accumulate) {
    T old = *dst;
    acc = vfma(1.f, old, acc);
  }
  *dst = acc;
}

// Separable form of the same transpose for wide tensors (C a multiple of 4): dx = Wy^T (dy Wx^T) per (image, input row hi, group of
// 32 channels).  Phase 1 folds the candidate output rows of hi into ONE row t [Wo][32 channels] in LDS -- a thread owns (column, channel
// quad), its candidate rows are loaded RB at a time (128-byte lines, all in flight); phase 2 folds the candidate columns of every
// input column out of LDS.  An output row feeds two input rows, so dy crosses L2 ~2x instead of the ~7.5x of the gather form (a 4x
// upsample), and the chain is one round of loads + one barrier.  Deterministic (fixed loop orders, no atomics).
#ifndef RESIZE_BWD_ROWS
#define RESIZE_BWD_ROWS 1   // 0: the gather form for every shape (A/B: tools/build_variants.sh)
#endif
constexpr int kResizeRowsThreads = 512;
template <int RB>
__global__ __launch_bounds__(kResizeRowsThreads) void resize_bwd_rows_k(const float* __restrict__ dy, int lddy, float* __restrict__ dx,
                                                                         int lddx, int Hi, int Wi, int Ho, int Wo, int C, float sh,
                                                                         float sw, int accumulate) {
  extern __shared__ float4 rsz_t[];   // [Wo][8 quads]
  const int cg = blockIdx.x, hi = blockIdx.y, n = blockIdx.z;
  const int q = threadIdx.x & 7, lane_px = threadIdx.x >> 3;
  const int c = cg * 32 + q * 4;
  const bool cok = c < C;
  int ho_lo = (int)floorf((float)(hi - 1) / sh) - 1, ho_hi = (int)ceilf((float)(hi + 1) / sh) + 1;
  if (ho_lo < 0) ho_lo = 0;
  if (ho_hi > Ho - 1) ho_hi = Ho - 1;
  const float* base = dy + (long long)n * Ho * Wo * lddy + (cok ? c : 0);
  for (int wo = lane_px; wo < Wo; wo += kResizeRowsThreads / 8) {
    float4 acc = vzero<float4>();
    for (int h0 = ho_lo; h0 <= ho_hi; h0 += RB) {
      float4 v[RB];
      float wy[RB];
#pragma unroll
      for (int u = 0; u < RB; ++u) {
        const int ho = h0 + u <= ho_hi ? h0 + u : ho_hi;
        int y0, y1;
        float ly;
        src_coord(ho, sh, Hi, y0, y1, ly);
        const float w_ = (y0 == hi ? 1.f - ly : 0.f) + (y1 == hi ? ly : 0.f);
        wy[u] = h0 + u <= ho_hi ? w_ : 0.f;
        v[u] = ld4(base + ((long long)ho * Wo + wo) * lddy);
      }
#pragma unroll
      for (int u = 0; u < RB; ++u) acc = vfma(wy[u], v[u], acc);
    }
    rsz_t[wo * 8 + q] = acc;
  }
  __syncthreads();
  for (int wi = lane_px; wi < Wi; wi += kResizeRowsThreads / 8) {
    int wo_lo = (int)floorf((float)(wi - 1) / sw) - 1, wo_hi = (int)ceilf((float)(wi + 1) / sw) + 1;
    if (wo_lo < 0) wo_lo = 0;
    if (wo_hi > Wo - 1) wo_hi = Wo - 1;
    float4 acc = vzero<float4>();
    for (int wo = wo_lo; wo <= wo_hi; ++wo) {
      int x0, x1;
      float lx;
      src_coord(wo, sw, Wi, x0, x1, lx);
      const float wx = (x0 == wi ? 1.f - lx : 0.f) + (x1 == wi ? lx : 0.f);
      acc = vfma(wx, rsz_t[wo * 8 + q], acc);
    }
    if (cok) {
      float* dst = dx + (((long long)n * Hi + hi) * Wi + wi) * lddx + c;
      if (accumulate) acc = f4add(acc, ld4(dst));
      st4(dst, acc);
    }
  }
}

// ------------------------------------------------------------------------------------------------ final 1x1 conv, Cout = 2
// 4 lanes per pixel, lanes interleave the channel quads (64 contiguous bytes per pixel per step), xor-shuffle reduce.
__global__ __launch_bounds__(256) void final_conv_fwd_k(const float* __restrict__ x, int ldx, const float* __restrict__ mask,
                                                        const float* __restrict__ w, const float* __restrict__ b,
                                                        float* __restrict__ y, long long rows, int C) {
  const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long row = gid >> 2;
  const int part = (int)(gid & 3);
  float a0 = 0.f, a1 = 0.f;
  if (row < rows) {
    const int Q = C >> 2;
    for (int q = part; q < Q; q += 4) {
      float4 v = ld4(x + row * ldx + q * 4);
      if (mask) v = f4mul(v, ld4(mask + row * ldx + q * 4));
      const float4 w0 = ld4(w + q * 8), w1 = ld4(w + q * 8 + 4);  // w[c][0..1] for c = 4q..4q+3
      a0 += v.x * w0.x + v.y * w0.z + v.z * w1.x + v.w * w1.z;
      a1 += v.x * w0.y + v.y * w0.w + v.z * w1.y + v.w * w1.w;
    }
  }
  a0 += __shfl_xor(a0, 1, 64);
  a1 += __shfl_xor(a1, 1, 64);
  a0 += __shfl_xor(a0, 2, 64);
  a1 += __shfl_xor(a1, 2, 64);
  if (row < rows && part == 0) *reinterpret_cast<float2*>(y + row * 2) = make_float2(a0 + b[0], a1 + b[1]);
}

// fin_part != nullptr (mliis_final_conv_bwd_data_fin): workgroup 0 first folds the loss partials the fused head launch left behind
// (ce_finalize: out[0..2] = {loss, ce, iou}) -- the fold needs nothing of this kernel and nothing of the step waits for it, so it
// rides here instead of in a launch of its own between the head and the backward pass.
struct CeFin {
  const float* part;
  int nblk, N, HW;
  float extra_loss;
  float* out;
  float* coef;
};
__device__ __forceinline__ void ce_finalize(const float* __restrict__ part, int nblk, int N, int HW, int dice, float extra_loss, float* __restrict__ out,
                                            float* __restrict__ coef);
__global__ __launch_bounds__(256) void final_conv_bwd_data_k(const float* __restrict__ dy, const float* __restrict__ w,
                                                             const float* __restrict__ mask, float* __restrict__ dx, int lddx,
                                                             long long rows, int C, CeFin fin) {
  if (fin.part != nullptr && blockIdx.x == 0) ce_finalize(fin.part, fin.nblk, fin.N, fin.HW, 0, fin.extra_loss, fin.out, fin.coef);   // (uniform)
  const int Q = C >> 2;
  const long long total = rows * Q;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / Q;
    const int q = (int)(i - r * Q);
    const float2 d = *reinterpret_cast<const float2*>(dy + r * 2);
    const float4 w0 = ld4(w + q * 8), w1 = ld4(w + q * 8 + 4);
    float4 o = make_float4(d.x * w0.x + d.y * w0.y, d.x * w0.z + d.y * w0.w, d.x * w1.x + d.y * w1.y, d.x * w1.z + d.y * w1.w);
    if (mask) o = f4mul(o, ld4(mask + r * lddx + q * 4));
    st4(dx + r * lddx + q * 4, o);
  }
}

// ------------------------------------------------------------------------------------------------ softmax CE (+dice)
// part layout [n][blk][4] = {sum CE, sum p1*t1, sum p1, sum t1}
__global__ __launch_bounds__(256) void ce_partial_k(const float* __restrict__ logits, const float* __restrict__ labels,
                                                    const int* __restrict__ idx, int HW, float ls, float* __restrict__ part) {
  __shared__ float sm[4][4];
  const int n = blockIdx.y;
  const int src = idx ? idx[n] : n;
  const float* z = logits + (long long)n * HW * 2;
  const float* t = labels + (long long)src * HW * 2;
  float ce = 0.f, I = 0.f, Sp = 0.f, St = 0.f;
  for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
    const float2 zz = *reinterpret_cast<const float2*>(z + (long long)p * 2);
    const float2 tt = *reinterpret_cast<const float2*>(t + (long long)p * 2);
    const float m = fmaxf(zz.x, zz.y);
    const float e0 = expf(zz.x - m), e1 = expf(zz.y - m);
    const float lse = m + logf(e0 + e1);
    const float t0 = tt.x * (1.f - ls) + 0.5f * ls, t1 = tt.y * (1.f - ls) + 0.5f * ls;
    ce -= t0 * (zz.x - lse) + t1 * (zz.y - lse);
    const float p1 = e1 / (e0 + e1);
    I += p1 * tt.y;
    Sp += p1;
    St += tt.y;
  }
  ce = wave_sum(ce);
  I = wave_sum(I);
  Sp = wave_sum(Sp);
  St = wave_sum(St);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    sm[wave][0] = ce;
    sm[wave][1] = I;
    sm[wave][2] = Sp;
    sm[wave][3] = St;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const float s = sm[0][threadIdx.x] + sm[1][threadIdx.x] + sm[2][threadIdx.x] + sm[3][threadIdx.x];
    part[((long long)n * gridDim.x + blockIdx.x) * 4 + threadIdx.x] = s;
  }
}

// out: [0] loss, [1] ce, [2] iou ; coef[n] = {A_n, B_n} for the dice gradient (0 when dice off).  One block: 8 images at a time,
// 32 lanes per image fold that image's partial blocks (double precision), a lane butterfly finishes the image, thread 0 adds the
// images up in index order (fixed order: deterministic).
__device__ __forceinline__ void ce_finalize(const float* __restrict__ part, int nblk, int N, int HW, int dice, float extra_loss,
                                            float* __restrict__ out, float* __restrict__ coef) {
  __shared__ double s_ce[8], s_iou[8];
  double ce = 0.0, iou = 0.0;
  const double eps = 1e-7;
  const int g = threadIdx.x >> 5, lane = threadIdx.x & 31;
  for (int n0 = 0; n0 < N; n0 += 8) {
    const int n = n0 + g;
    double v[4] = {0, 0, 0, 0};
    if (n < N)
      for (int b = lane; b < nblk; b += 32) {
        const float4 q = ld4(part + ((long long)n * nblk + b) * 4);
        v[0] += q.x;
        v[1] += q.y;
        v[2] += q.z;
        v[3] += q.w;
      }
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int off = 1; off < 32; off <<= 1) v[k] += __shfl_xor(v[k], off);
    if (lane == 0 && n < N) {
      const double I = v[1], Sp = v[2], St = v[3];
      const double U = Sp + St - I;
      s_ce[g] = v[0];
      s_iou[g] = (I + eps) / (U + eps);
      coef[2 * n + 0] = (float)(1.0 / (U + eps));
      coef[2 * n + 1] = (float)((I + eps) / ((U + eps) * (U + eps)));
    }
    __syncthreads();
    if (threadIdx.x == 0)
      for (int j = 0; j < 8 && n0 + j < N; ++j) {
        ce += s_ce[j];
        iou += s_iou[j];
      }
    __syncthreads();
  }
  if (threadIdx.x != 0) return;
  ce /= (double)N * HW;
  iou /= N;
  double loss = ce + extra_loss;
  double dLdiou = 0.0;
  if (dice) {
    loss -= log(2.0 * iou / (iou + 1.0));
    dLdiou = -1.0 / (iou * (iou + 1.0));
  }
  for (int n = 0; n < N; ++n) {
    coef[2 * n + 0] = (float)(coef[2 * n + 0] * dLdiou / N);
    coef[2 * n + 1] = (float)(coef[2 * n + 1] * dLdiou / N);
  }
  out[0] = (float)loss;
  out[1] = (float)ce;
  out[2] = (float)iou;
}

__global__ __launch_bounds__(256) void ce_finalize_k(const float* __restrict__ part, int nblk, int N, int HW, int dice, float extra_loss,
                                                     float* __restrict__ out, float* __restrict__ coef) {
  ce_finalize(part, nblk, N, HW, dice, extra_loss, out, coef);
}

// fin_part != nullptr (no dice term: the gradient needs nothing from the finalize step): workgroup (0, 0) also folds the loss partials
// into out[] before its share of the gradient -- one launch fewer than ce_partial -> ce_finalize -> ce_grad.
__global__ __launch_bounds__(256) void ce_grad_k(const float* __restrict__ logits, const float* __restrict__ labels,
                                                 const int* __restrict__ idx, int HW, float ls, float inv_rows,
                                                 const float* __restrict__ coef, float* __restrict__ dlogits,
                                                 float* __restrict__ pred, const float* __restrict__ fin_part, int fin_nblk, int N,
                                                 float extra_loss, float* __restrict__ fin_out, float* __restrict__ fin_coef) {
  if (fin_part != nullptr && blockIdx.x == 0 && blockIdx.y == 0) ce_finalize(fin_part, fin_nblk, N, HW, 0, extra_loss, fin_out, fin_coef);
  const int n = blockIdx.y;
  const int src = idx ? idx[n] : n;
  const float* z = logits + (long long)n * HW * 2;
  const float* t = labels + (long long)src * HW * 2;
  const float An = fin_part != nullptr ? 0.f : coef[2 * n], Bn = fin_part != nullptr ? 0.f : coef[2 * n + 1];
  for (int p = blockIdx.x * 256 + threadIdx.x; p < HW; p += gridDim.x * 256) {
    const float2 zz = *reinterpret_cast<const float2*>(z + (long long)p * 2);
    const float2 tt = *reinterpret_cast<const float2*>(t + (long long)p * 2);
    const float m = fmaxf(zz.x, zz.y);
    const float e0 = expf(zz.x - m), e1 = expf(zz.y - m);
    const float inv = 1.f / (e0 + e1);
    const float p0 = e0 * inv, p1 = e1 * inv;
    const float t0 = tt.x * (1.f - ls) + 0.5f * ls, t1 = tt.y * (1.f - ls) + 0.5f * ls;
    const float ts = t0 + t1;
    float d0 = (p0 * ts - t0) * inv_rows, d1 = (p1 * ts - t1) * inv_rows;
    const float dp1 = (An * tt.y - Bn * (1.f - tt.y)) * p1 * (1.f - p1);
    d1 += dp1;
    d0 -= dp1;
    if (dlogits) *reinterpret_cast<float2*>(dlogits + ((long long)n * HW + p) * 2) = make_float2(d0, d1);
    if (pred) *reinterpret_cast<float2*>(pred + ((long long)n * HW + p) * 2) = make_float2(p0 > 0.5f ? 1.f : 0.f, p1 > 0.5f ? 1.f : 0.f);
  }
}

// ------------------------------------------------------------------------------------------------ resize -> softmax CE -> resize^T, one pass
// The tail of a training step without the dice term: logits = resize(small) (models/efficientlab.py:166-173), per-pixel softmax
// cross-entropy (:294-303), its gradient (p - t) / (N H W) -- which needs no global sum -- and the transpose of the resize back to
// the decoder's map.  A workgroup owns an 8 x 8 tile of DECODER pixels:
//   phase 1: every image pixel that a pixel of the tile reaches (the tile's footprint, <= kHeadFoot^2 pixels) gets its logits rebuilt
//            from the four decoder pixels around it (resize_fwd_k's arithmetic) and its gradient formed (ce_grad_k's arithmetic) into
//            LDS; the loss terms of an image pixel are added by the tile that holds its top-left source pixel (one owner each);
//   phase 2: the 8 lanes of a decoder pixel walk its candidates exactly as resize_bwd_k does, reading the gradients from LDS.
// The full-resolution logits / gradient tensors are never written (four launches and 4 x 3.2 MB of traffic at 224x224, N = 8, become
// two).  Loss partials per workgroup [n][tile][4] as ce_partial_k's, folded by ce_finalize_k -- a launch of its own: a last-arriver
// fold inside this kernel needs device-scope fences, which on this eight-L2 part write back and invalidate a whole L2 each
// (measured: 39 us for the launch, more than the four it replaces).  (A first form without the LDS phase -- every decoder pixel
// rebuilding its ~100 candidates itself -- took 69 us: one dependent load chain per candidate.)
constexpr int kHeadTile = 8, kHeadFoot = 48;
__device__ __forceinline__ int head_lo(int i, float s) {
  const int v = (int)floorf((float)(i - 1) / s) - 1;
  return v < 0 ? 0 : v;
}
__device__ __forceinline__ int head_hi(int i, float s, int n) {
  const int v = (int)ceilf((float)(i + 1) / s) + 1;
  return v > n - 1 ? n - 1 : v;
}
__global__ __launch_bounds__(256) void head_ce_fused_k(const float* __restrict__ small, const float* __restrict__ labels,
                                                       const int* __restrict__ idx, int N, int Hi, int Wi, int Ho, int Wo, float sh, float sw,
                                                       float ls, float inv_rows, float* __restrict__ dsmall, float* __restrict__ part,
                                                       int tiles_x) {
  __shared__ float2 gl[kHeadFoot * kHeadFoot];
  __shared__ float sm[4][4];
  const int t = threadIdx.x;
  const int n = blockIdx.y;
  const int src = idx ? idx[n] : n;
  const int ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const int hi0 = ty * kHeadTile, wi0 = tx * kHeadTile;
  const int hi1 = hi0 + kHeadTile - 1 < Hi - 1 ? hi0 + kHeadTile - 1 : Hi - 1;
  const int wi1 = wi0 + kHeadTile - 1 < Wi - 1 ? wi0 + kHeadTile - 1 : Wi - 1;
  const int fo0 = head_lo(hi0, sh), fo1 = head_hi(hi1, sh, Ho), fw0 = head_lo(wi0, sw), fw1 = head_hi(wi1, sw, Wo);
  const int FH = fo1 - fo0 + 1, FW = fw1 - fw0 + 1;   // (host: <= kHeadFoot)
  const float* zb = small + (long long)n * Hi * Wi * 2;
  const float* tb = labels + (long long)src * Ho * Wo * 2;
  float ce = 0.f, I = 0.f, Sp = 0.f, St = 0.f;
  for (int p = t; p < FH * FW; p += 256) {
    const int fr = p / FW, fc = p - fr * FW;
    const int ho = fo0 + fr, wo = fw0 + fc;
    int y0, y1, x0, x1;
    float ly, lx;
    src_coord(ho, sh, Hi, y0, y1, ly);
    src_coord(wo, sw, Wi, x0, x1, lx);
    const float2 tl = *reinterpret_cast<const float2*>(zb + ((long long)y0 * Wi + x0) * 2);
    const float2 tr = *reinterpret_cast<const float2*>(zb + ((long long)y0 * Wi + x1) * 2);
    const float2 bl = *reinterpret_cast<const float2*>(zb + ((long long)y1 * Wi + x0) * 2);
    const float2 br = *reinterpret_cast<const float2*>(zb + ((long long)y1 * Wi + x1) * 2);
    const float2 tt = *reinterpret_cast<const float2*>(tb + ((long long)ho * Wo + wo) * 2);
    float2 zz = make_float2(0.f, 0.f);
    float wtl, wtr, wbl, wbr;
    bilinear_weights(ly, lx, wtl, wtr, wbl, wbr);
    zz = vfma(wtl, tl, zz);
    zz = vfma(wtr, tr, zz);
    zz = vfma(wbl, bl, zz);
    zz = vfma(wbr, br, zz);
    const float m = fmaxf(zz.x, zz.y);
    const float e0 = expf(zz.x - m), e1 = expf(zz.y - m);
    const float inv = 1.f / (e0 + e1);
    const float p0 = e0 * inv, p1 = e1 * inv;
    const float t0 = tt.x * (1.f - ls) + 0.5f * ls, t1 = tt.y * (1.f - ls) + 0.5f * ls;
    const float ts = lone(t0) + lone(t1);   // (the sum of the two halves of a packed pair, broadcast back over the pair: common.hpp, lone())
    gl[fr * kHeadFoot + fc] = make_float2((p0 * ts - t0) * inv_rows, (p1 * ts - t1) * inv_rows);
    if (y0 >= hi0 && y0 <= hi1 && x0 >= wi0 && x0 <= wi1) {   // this tile owns the image pixel's loss terms
      const float lse = m + logf(e0 + e1);
      ce -= t0 * (zz.x - lse) + t1 * (zz.y - lse);
      const float q1 = e1 / (e0 + e1);
      I += q1 * tt.y;
      Sp += q1;
      St += tt.y;
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int pg = (t >> 3) + 32 * it, rl = t & 7;
    const int hi = hi0 + pg / kHeadTile, wi = wi0 + pg % kHeadTile;
    const bool live = hi <= hi1 && wi <= wi1;
    const int hc = live ? hi : hi0, wc = live ? wi : wi0;
    const int ho_lo = head_lo(hc, sh), ho_hi = head_hi(hc, sh, Ho), wo_lo = head_lo(wc, sw), wo_hi = head_hi(wc, sw, Wo);
    float2 acc = make_float2(0.f, 0.f);
    for (int ho = ho_lo + rl; ho <= ho_hi; ho += 8) {
      int y0, y1;
      float ly;
      src_coord(ho, sh, Hi, y0, y1, ly);
      const float wy = (y0 == hc ? 1.f - ly : 0.f) + (y1 == hc ? ly : 0.f);
      if (wy == 0.f) continue;
      const float2* row = gl + (ho - fo0) * kHeadFoot - fw0;
      for (int wo = wo_lo; wo <= wo_hi; ++wo) {
        int x0, x1;
        float lx;
        src_coord(wo, sw, Wi, x0, x1, lx);
        const float wx = (x0 == wc ? 1.f - lx : 0.f) + (x1 == wc ? lx : 0.f);
        acc = vfma(wy * wx, row[wo], acc);
      }
    }
#pragma unroll
    for (int off = 1; off < 8; off <<= 1) {
      acc.x += __shfl_xor(acc.x, off);
      acc.y += __shfl_xor(acc.y, off);
    }
    if (live && rl == 0) *reinterpret_cast<float2*>(dsmall + ((long long)n * Hi * Wi + (long long)hi * Wi + wi) * 2) = acc;
  }
  ce = wave_sum(ce);
  I = wave_sum(I);
  Sp = wave_sum(Sp);
  St = wave_sum(St);
  const int lane = t & 63, wave = t >> 6;
  if (lane == 0) {
    sm[wave][0] = ce;
    sm[wave][1] = I;
    sm[wave][2] = Sp;
    sm[wave][3] = St;
  }
  __syncthreads();
  if (t < 4) part[((long long)n * gridDim.x + blockIdx.x) * 4 + t] = sm[0][t] + sm[1][t] + sm[2][t] + sm[3][t];
}

// ------------------------------------------------------------------------------------------------ DARC1 regulariser
// models/regularizers.py:20-22: weight * max over (h, w, c) of sum_n |logits[n,h,w,c]|.  Gradient: weight * sign(logits[n, argmax]) for
// every n at the ONE arg-max position (ties: the first index).  darc1_partial_k: per-block (max, index); darc1_apply_k (one block):
// global arg-max in block order (deterministic), then the loss term and the N gradient entries.
__global__ __launch_bounds__(256) void darc1_partial_k(const float* __restrict__ z, int N, long long per_img, float* __restrict__ pval,
                                                       int* __restrict__ pidx) {
  __shared__ float sv[256];
  __shared__ int si[256];
  float best = -1.f;
  int bidx = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < per_img; i += (long long)gridDim.x * 256) {
    float s = 0.f;
    for (int n = 0; n < N; ++n) s += fabsf(z[(long long)n * per_img + i]);
    if (s > best) {
      best = s;
      bidx = (int)i;
    }
  }
  sv[threadIdx.x] = best;
  si[threadIdx.x] = bidx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      const float a = sv[threadIdx.x], b = sv[threadIdx.x + o];
      const int ia = si[threadIdx.x], ib = si[threadIdx.x + o];
      if (b > a || (b == a && ib < ia)) {
        sv[threadIdx.x] = b;
        si[threadIdx.x] = ib;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    pval[blockIdx.x] = sv[0];
    pidx[blockIdx.x] = si[0];
  }
}

__global__ __launch_bounds__(256) void darc1_apply_k(const float* __restrict__ z, int N, long long per_img, const float* __restrict__ pval,
                                                     const int* __restrict__ pidx, int nblk, float weight, float* __restrict__ dlogits,
                                                     float* __restrict__ out) {
  __shared__ float sv[256];
  __shared__ int si[256];
  float best = -1.f;
  int bidx = 0x7fffffff;
  for (int b = threadIdx.x; b < nblk; b += 256) {
    const float v = pval[b];
    const int ix = pidx[b];
    if (v > best || (v == best && ix < bidx)) {
      best = v;
      bidx = ix;
    }
  }
  sv[threadIdx.x] = best;
  si[threadIdx.x] = bidx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) {
      const float a = sv[threadIdx.x], b = sv[threadIdx.x + o];
      const int ia = si[threadIdx.x], ib = si[threadIdx.x + o];
      if (b > a || (b == a && ib < ia)) {
        sv[threadIdx.x] = b;
        si[threadIdx.x] = ib;
      }
    }
    __syncthreads();
  }
  const int pos = si[0];
  if (threadIdx.x == 0 && out != nullptr) out[0] += weight * sv[0];
  if (dlogits != nullptr)
    for (int n = threadIdx.x; n < N; n += 256) {
      const float v = z[(long long)n * per_img + pos];
      dlogits[(long long)n * per_img + pos] += weight * (v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f));
    }
}

static inline int ew_blocks(long long q) {
  long long b = (q + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

// ASPP activation sites (models/efficientlab.py:258-286): y = swish(z) * mask (conv -> swish -> dropout; branches 0, 1 and the
// output conv) or, with pre_mask, y = swish(z * mask) (the pooled branch applies dropout BEFORE the swish).  mask: dropout scale per
// element (0 or 1/keep), NULL = inference.  backward: dz = dy * mask * swish'(z)   |   dz = dy * swish'(z * mask) * mask.
// Row-strided operands (channel slices of the concat buffer).  One memory round trip: all operands are fetched first.
__global__ __launch_bounds__(256) void swish_mask_fwd_k(const float* __restrict__ z, int ldz, const float* __restrict__ mask, int ldm,
                                                        float* __restrict__ y, int ldy, long long rows, int C, int pre_mask) {
  const unsigned Q = (unsigned)C >> 2;
  const unsigned total = (unsigned)rows * Q;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned r = i / Q;
    const int c = (int)(i - r * Q) << 2;
    const float4 vz = ld4(z + (long long)r * ldz + c);
    const float4 vm = mask ? ld4(mask + (long long)r * ldm + c) : make_float4(1.f, 1.f, 1.f, 1.f);
    float4 v;
    if (pre_mask) {
      v = make_float4(swish_f(vz.x * vm.x), swish_f(vz.y * vm.y), swish_f(vz.z * vm.z), swish_f(vz.w * vm.w));
    } else {
      v = make_float4(swish_f(vz.x) * vm.x, swish_f(vz.y) * vm.y, swish_f(vz.z) * vm.z, swish_f(vz.w) * vm.w);
    }
    st4(y + (long long)r * ldy + c, v);
  }
}

__global__ __launch_bounds__(256) void swish_mask_bwd_k(const float* __restrict__ dy, int lddy, const float* __restrict__ z, int ldz,
                                                        const float* __restrict__ mask, int ldm, float* __restrict__ dz, int lddz,
                                                        long long rows, int C, int pre_mask) {
  const unsigned Q = (unsigned)C >> 2;
  const unsigned total = (unsigned)rows * Q;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned r = i / Q;
    const int c = (int)(i - r * Q) << 2;
    const float4 vd = ld4(dy + (long long)r * lddy + c);
    const float4 vz = ld4(z + (long long)r * ldz + c);
    const float4 vm = mask ? ld4(mask + (long long)r * ldm + c) : make_float4(1.f, 1.f, 1.f, 1.f);
    float4 g;
    if (pre_mask) {
      g = make_float4(swish_grad_f(vz.x * vm.x), swish_grad_f(vz.y * vm.y), swish_grad_f(vz.z * vm.z), swish_grad_f(vz.w * vm.w));
    } else {
      g = make_float4(swish_grad_f(vz.x), swish_grad_f(vz.y), swish_grad_f(vz.z), swish_grad_f(vz.w));
    }
    st4(dz + (long long)r * lddz + c, make_float4(vd.x * vm.x * g.x, vd.y * vm.y * g.y, vd.z * vm.z * g.z, vd.w * vm.w * g.w));
  }
}

}  // namespace mliis

using namespace mliis;

extern "C" {

int mliis_resize_bilinear_fwd(const float* x, int ldx, float* y, int ldy, int N, int Hi, int Wi, int Ho, int Wo, int C,
                              hipStream_t stream) {
  MLIIS_REQUIRE(x && y, MLIIS_ERR_ARG, "resize_bilinear_fwd: null pointer");
  MLIIS_REQUIRE(N > 0 && Hi > 0 && Wi > 0 && Ho > 1 && Wo > 1 && C > 0 && (C & 1) == 0 && ldx >= C && ldy >= C, MLIIS_ERR_ARG,
                "resize_bilinear_fwd: bad shape");
  const float sh = (float)(Hi - 1) / (float)(Ho - 1), sw = (float)(Wi - 1) / (float)(Wo - 1);
  if ((C & 3) == 0 && (ldx & 3) == 0 && (ldy & 3) == 0 && aligned16(x) && aligned16(y)) {
    long long q = (long long)N * Ho * Wo * (C / 4);
    hipLaunchKernelGGL((resize_fwd_k<4>), dim3(ew_blocks(q)), dim3(256), 0, stream, x, ldx, y, ldy, N, Hi, Wi, Ho, Wo, C, sh, sw);
  } else {
    MLIIS_REQUIRE((ldx & 1) == 0 && (ldy & 1) == 0, MLIIS_ERR_ALIGN, "resize_bilinear_fwd: leading dims must be even");
    long long q = (long long)N * Ho * Wo * (C / 2);
    hipLaunchKernelGGL((resize_fwd_k<2>), dim3(ew_blocks(q)), dim3(256), 0, stream, x, ldx, y, ldy, N, Hi, Wi, Ho, Wo, C, sh, sw);
  }
  MLIIS_CHECK_LAUNCH("resize_bilinear_fwd");
  return MLIIS_OK;
}

// dx [N,Hi,Wi,C] (+)= transpose-resize of dy [N,Ho,Wo,C]  (Hi,Wi = forward INPUT size)
int mliis_resize_bilinear_bwd(const float* dy, int lddy, float* dx, int lddx, int N, int Hi, int Wi, int Ho, int Wo, int C,
                              int accumulate, hipStream_t stream) {
  MLIIS_REQUIRE(dy && dx, MLIIS_ERR_ARG, "resize_bilinear_bwd: null pointer");
  MLIIS_REQUIRE(N > 0 && Hi > 1 && Wi > 1 && Ho > 1 && Wo > 1 && C > 0 && (C & 1) == 0 && lddy >= C && lddx >= C, MLIIS_ERR_ARG,
                "resize_bilinear_bwd: bad shape (input side must be > 1)");
  const float sh = (float)(Hi - 1) / (float)(Ho - 1), sw = (float)(Wi - 1) / (float)(Wo - 1);
  if ((C & 3) == 0 && (lddx & 3) == 0 && (lddy & 3) == 0 && aligned16(dx) && aligned16(dy)) {
    if (RESIZE_BWD_ROWS && C >= 16 && Wo <= 512 && Hi <= 65535 && N <= 65535) {   // separable form: one workgroup per (32-channel group, input row, image)
      const dim3 grid((unsigned)((C + 31) / 32), (unsigned)Hi, (unsigned)N);
      const size_t lds = (size_t)Wo * 8 * sizeof(float4);
      if (2.f / sh + 3.f > 4.f)
        hipLaunchKernelGGL((resize_bwd_rows_k<12>), grid, dim3(kResizeRowsThreads), lds, stream, dy, lddy, dx, lddx, Hi, Wi, Ho, Wo, C, sh, sw, accumulate);
      else
        hipLaunchKernelGGL((resize_bwd_rows_k<4>), grid, dim3(kResizeRowsThreads), lds, stream, dy, lddy, dx, lddx, Hi, Wi, Ho, Wo, C, sh, sw, accumulate);
      MLIIS_CHECK_LAUNCH("resize_bilinear_bwd");
      return MLIIS_OK;
    }
    long long q = (long long)N * Hi * Wi * (C / 4);
    hipLaunchKernelGGL((resize_bwd_k<4>), dim3((unsigned)((q * 8 + 255) / 256)), dim3(256), 0, stream, dy, lddy, dx, lddx, N, Hi, Wi, Ho, Wo, C, sh,
                       sw, accumulate, q);
  } else {
    MLIIS_REQUIRE((lddx & 1) == 0 && (lddy & 1) == 0, MLIIS_ERR_ALIGN, "resize_bilinear_bwd: leading dims must be even");
    long long q = (long long)N * Hi * Wi * (C / 2);
    hipLaunchKernelGGL((resize_bwd_k<2>), dim3((unsigned)((q * 8 + 255) / 256)), dim3(256), 0, stream, dy, lddy, dx, lddx, N, Hi, Wi, Ho, Wo, C, sh,
                       sw, accumulate, q);
  }
  MLIIS_CHECK_LAUNCH("resize_bilinear_bwd");
  return MLIIS_OK;
}

int mliis_final_conv_fwd(const float* x, int ldx, const float* mask, const float* w, const float* b, float* y, long long rows, int C,
                         hipStream_t stream) {
  MLIIS_REQUIRE(x && w && b && y, MLIIS_ERR_ARG, "final_conv_fwd: null pointer");
  MLIIS_REQUIRE(rows > 0 && C > 0 && (C & 3) == 0 && (ldx & 3) == 0 && ldx >= C, MLIIS_ERR_ARG, "final_conv_fwd: bad shape");
  MLIIS_REQUIRE(aligned16(x) && aligned16(mask) && aligned16(w), MLIIS_ERR_ALIGN, "final_conv_fwd: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(final_conv_fwd_k, dim3(ceil_div(rows * 4, 256)), dim3(256), 0, stream, x, ldx, mask, w, b, y, rows, C);
  MLIIS_CHECK_LAUNCH("final_conv_fwd");
  return MLIIS_OK;
}

int mliis_final_conv_bwd_data(const float* dy, const float* w, const float* mask, float* dx, int lddx, long long rows, int C,
                              hipStream_t stream) {
  MLIIS_REQUIRE(dy && w && dx, MLIIS_ERR_ARG, "final_conv_bwd_data: null pointer");
  MLIIS_REQUIRE(rows > 0 && C > 0 && (C & 3) == 0 && (lddx & 3) == 0 && lddx >= C, MLIIS_ERR_ARG, "final_conv_bwd_data: bad shape");
  MLIIS_REQUIRE(aligned16(dx) && aligned16(mask) && aligned16(w), MLIIS_ERR_ALIGN, "final_conv_bwd_data: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(final_conv_bwd_data_k, dim3(ew_blocks(rows * (C / 4))), dim3(256), 0, stream, dy, w, mask, dx, lddx, rows, C, CeFin{});
  MLIIS_CHECK_LAUNCH("final_conv_bwd_data");
  return MLIIS_OK;
}

static inline int head_tiles(int n);
// mliis_final_conv_bwd_data + the loss fold of a mliis_head_ce_fused call that was given out == NULL: fin_ws = that call's workspace
// (untouched since), (N, Hd, Wd, H, W, extra_loss) as given there; out[0..2] = {loss + extra_loss, ce, iou}.
int mliis_final_conv_bwd_data_fin(const float* dy, const float* w, const float* mask, float* dx, int lddx, long long rows, int C, float* fin_ws,
                                  int N, int Hd, int Wd, int H, int W, float extra_loss, float* out, hipStream_t stream) {
  MLIIS_REQUIRE(dy && w && dx && fin_ws && out, MLIIS_ERR_ARG, "final_conv_bwd_data_fin: null pointer");
  MLIIS_REQUIRE(rows > 0 && C > 0 && (C & 3) == 0 && (lddx & 3) == 0 && lddx >= C && N > 0 && Hd > 1 && Wd > 1 && H > 1 && W > 1, MLIIS_ERR_ARG,
                "final_conv_bwd_data_fin: bad shape");
  MLIIS_REQUIRE(aligned16(dx) && aligned16(mask) && aligned16(w) && aligned16(fin_ws), MLIIS_ERR_ALIGN, "final_conv_bwd_data_fin: pointers must be 16-byte aligned");
  const int nblk = head_tiles(Hd) * head_tiles(Wd);
  const CeFin fin{fin_ws, nblk, N, H * W, extra_loss, out, fin_ws + (size_t)N * nblk * 4};
  hipLaunchKernelGGL(final_conv_bwd_data_k, dim3(ew_blocks(rows * (C / 4))), dim3(256), 0, stream, dy, w, mask, dx, lddx, rows, C, fin);
  MLIIS_CHECK_LAUNCH("final_conv_bwd_data_fin");
  return MLIIS_OK;
}

size_t mliis_softmax_ce_workspace_floats(int N, int H, int W) {
  if (N <= 0 || H <= 0 || W <= 0) return 0;
  int nblk = ceil_div((long long)H * W, 256 * 8);
  if (nblk > 256) nblk = 256;
  return (size_t)N * nblk * 4 + 2 * (size_t)N + 8;
}

// logits [N,H,W,2]; labels [S,H,W,2] addressed through img_idx (nullable).  Writes out[0..2] = {loss, ce, iou} (device),
// dlogits (nullable), pred (nullable; (softmax > 0.5) as float).  extra_loss is added to the reported loss only (L2 term).
int mliis_softmax_ce(const float* logits, const float* labels, const int* img_idx, int N, int H, int W, float label_smoothing,
                     int dice, float extra_loss, float* dlogits, float* pred, float* out, float* ws, size_t ws_floats,
                     hipStream_t stream) {
  MLIIS_REQUIRE(logits && labels && out && ws, MLIIS_ERR_ARG, "softmax_ce: null pointer");
  MLIIS_REQUIRE(N > 0 && H > 0 && W > 0, MLIIS_ERR_ARG, "softmax_ce: bad shape");
  MLIIS_REQUIRE((reinterpret_cast<uintptr_t>(logits) & 7u) == 0 && (reinterpret_cast<uintptr_t>(labels) & 7u) == 0, MLIIS_ERR_ALIGN,
                "softmax_ce: logits/labels must be 8-byte aligned");
  const int HW = H * W;
  int nblk = ceil_div((long long)HW, 256 * 8);
  if (nblk > 256) nblk = 256;
  MLIIS_REQUIRE((size_t)N * nblk * 4 + 2 * (size_t)N <= ws_floats && aligned16(ws), MLIIS_ERR_WORKSPACE,
                "softmax_ce: workspace too small or unaligned");
  float* coef = ws + (size_t)N * nblk * 4;
  hipLaunchKernelGGL(ce_partial_k, dim3(nblk, N), dim3(256), 0, stream, logits, labels, img_idx, HW, label_smoothing, ws);
  MLIIS_CHECK_LAUNCH("softmax_ce_partial");
  const bool merged = !dice && (dlogits || pred);   // (without the dice term the gradient does not depend on the finalize step)
  if (!merged) {
    hipLaunchKernelGGL(ce_finalize_k, dim3(1), dim3(256), 0, stream, ws, nblk, N, HW, dice, extra_loss, out, coef);
    MLIIS_CHECK_LAUNCH("softmax_ce_finalize");
  }
  if (dlogits || pred) {
    // one pixel per thread (the partial-sum pass above keeps its coarser grid: its block count is the finalize kernel's work)
    hipLaunchKernelGGL(ce_grad_k, dim3(ceil_div(HW, 256), N), dim3(256), 0, stream, logits, labels, img_idx, HW, label_smoothing,
                       1.0f / ((float)N * (float)HW), coef, dlogits, pred, merged ? ws : nullptr, nblk, N, extra_loss, out, coef);
    MLIIS_CHECK_LAUNCH("softmax_ce_grad");
  }
  return MLIIS_OK;
}
static inline int head_tiles(int n) { return (n + kHeadTile - 1) / kHeadTile; }
// 1 when mliis_head_ce_fused takes this pair of maps (the image-side footprint of an 8 x 8 tile of the decoder's map fits its LDS
// buffer: up-sampling factors up to ~4.5), else 0: use mliis_resize_bilinear_fwd -> mliis_softmax_ce -> mliis_resize_bilinear_bwd.
int mliis_head_ce_fused_supported(int Hd, int Wd, int H, int W) {
  if (Hd <= 1 || Wd <= 1 || H <= 1 || W <= 1) return 0;
  const float sh = (float)(Hd - 1) / (float)(H - 1), sw = (float)(Wd - 1) / (float)(W - 1);
  const int fh = (int)ceilf((float)(kHeadTile + 1) / sh) + 5, fw = (int)ceilf((float)(kHeadTile + 1) / sw) + 5;
  return fh <= kHeadFoot && fw <= kHeadFoot;
}

size_t mliis_head_ce_fused_workspace_floats(int N, int Hd, int Wd) {
  if (N <= 0 || Hd <= 0 || Wd <= 0) return 0;
  return (size_t)N * head_tiles(Hd) * head_tiles(Wd) * 4 + 2 * (size_t)N + 8;
}

// The training step's tail (no dice term) in two launches instead of five: small [N,Hd,Wd,2] = the final conv's output on the
// decoder's map; logits = bilinear resize (align_corners) to [H,W]; softmax cross-entropy against labels [S,H,W,2] (addressed through
// img_idx, nullable) with label smoothing; dsmall [N,Hd,Wd,2] = the gradient w.r.t. small (what mliis_resize_bilinear_fwd ->
// mliis_softmax_ce -> mliis_resize_bilinear_bwd compute, the same arithmetic per element); out[0..2] = {loss + extra_loss, ce, iou}
// (folded from the launch's per-workgroup partials by ce_finalize_k).  ws: mliis_head_ce_fused_workspace_floats(N, Hd, Wd) floats.
int mliis_head_ce_fused(const float* small, const float* labels, const int* img_idx, int N, int Hd, int Wd, int H, int W,
                        float label_smoothing, float extra_loss, float* dsmall, float* out, float* ws, size_t ws_floats,
                        hipStream_t stream) {
  MLIIS_REQUIRE(small && labels && dsmall && ws, MLIIS_ERR_ARG, "head_ce_fused: null pointer");
  MLIIS_REQUIRE(N > 0 && Hd > 1 && Wd > 1 && H > 1 && W > 1 && N <= 65535, MLIIS_ERR_ARG, "head_ce_fused: bad shape (both maps must be larger than 1x1)");
  MLIIS_REQUIRE(mliis_head_ce_fused_supported(Hd, Wd, H, W), MLIIS_ERR_UNSUPPORTED,
                "head_ce_fused: up-sampling factor too large for the tile footprint (mliis_head_ce_fused_supported)");
  MLIIS_REQUIRE((reinterpret_cast<uintptr_t>(small) & 7u) == 0 && (reinterpret_cast<uintptr_t>(labels) & 7u) == 0 &&
                    (reinterpret_cast<uintptr_t>(dsmall) & 7u) == 0,
                MLIIS_ERR_ALIGN, "head_ce_fused: tensors must be 8-byte aligned");
  const int tx = head_tiles(Wd), nblk = head_tiles(Hd) * tx;
  MLIIS_REQUIRE((size_t)N * nblk * 4 + 2 * (size_t)N <= ws_floats && aligned16(ws), MLIIS_ERR_WORKSPACE,
                "head_ce_fused: workspace too small or unaligned");
  float* coef = ws + (size_t)N * nblk * 4;
  const float sh = (float)(Hd - 1) / (float)(H - 1), sw = (float)(Wd - 1) / (float)(W - 1);
  hipLaunchKernelGGL(head_ce_fused_k, dim3(nblk, N), dim3(256), 0, stream, small, labels, img_idx, N, Hd, Wd, H, W, sh, sw, label_smoothing,
                     1.0f / ((float)N * (float)H * (float)W), dsmall, ws, tx);
  MLIIS_CHECK_LAUNCH("head_ce_fused");
  if (out == nullptr) return MLIIS_OK;   // (the loss fold is left to mliis_final_conv_bwd_data_fin, with ws untouched until then)
  hipLaunchKernelGGL(ce_finalize_k, dim3(1), dim3(256), 0, stream, ws, nblk, N, H * W, 0, extra_loss, out, coef);
  MLIIS_CHECK_LAUNCH("head_ce_fused_finalize");
  return MLIIS_OK;
}

static int swish_mask_check(const char* name, const void* a, int lda, const void* b, int ldb, const void* m, int ldm, const void* o, int ldo,
                            long long rows, int C) {
  MLIIS_REQUIRE(a && b && o, MLIIS_ERR_ARG, "%s: null pointer", name);
  MLIIS_REQUIRE(rows > 0 && C > 0 && (C & 3) == 0 && (lda & 3) == 0 && lda >= C && (ldb & 3) == 0 && ldb >= C && (ldo & 3) == 0 && ldo >= C &&
                    (m == nullptr || ((ldm & 3) == 0 && ldm >= C)),
                MLIIS_ERR_ARG, "%s: bad shape", name);
  MLIIS_REQUIRE(aligned16(a) && aligned16(b) && aligned16(m) && aligned16(o), MLIIS_ERR_ALIGN, "%s: pointers must be 16-byte aligned", name);
  MLIIS_REQUIRE(rows * (C / 4) < (1LL << 31), MLIIS_ERR_UNSUPPORTED, "%s: tensor too large for 32-bit indexing", name);
  return MLIIS_OK;
}

int mliis_swish_mask_fwd(const float* z, int ldz, const float* mask, int ldm, float* y, int ldy, long long rows, int C, int pre_mask,
                         hipStream_t stream) {
  int rc = swish_mask_check("swish_mask_fwd", z, ldz, z, ldz, mask, ldm, y, ldy, rows, C);
  if (rc) return rc;
  const long long q = rows * (C / 4);
  hipLaunchKernelGGL(swish_mask_fwd_k, dim3((unsigned)((q + 255) / 256 > 4096 ? 4096 : (q + 255) / 256)), dim3(256), 0, stream, z, ldz, mask, ldm, y,
                     ldy, rows, C, pre_mask);
  MLIIS_CHECK_LAUNCH("swish_mask_fwd");
  return MLIIS_OK;
}

int mliis_swish_mask_bwd(const float* dy, int lddy, const float* z, int ldz, const float* mask, int ldm, float* dz, int lddz, long long rows,
                         int C, int pre_mask, hipStream_t stream) {
  int rc = swish_mask_check("swish_mask_bwd", dy, lddy, z, ldz, mask, ldm, dz, lddz, rows, C);
  if (rc) return rc;
  const long long q = rows * (C / 4);
  hipLaunchKernelGGL(swish_mask_bwd_k, dim3((unsigned)((q + 255) / 256 > 4096 ? 4096 : (q + 255) / 256)), dim3(256), 0, stream, dy, lddy, z, ldz, mask,
                     ldm, dz, lddz, rows, C, pre_mask);
  MLIIS_CHECK_LAUNCH("swish_mask_bwd");
  return MLIIS_OK;
}

// DARC1 regulariser on the logits [N, per_img] (models/regularizers.py:20-22): adds weight * max_pos sum_n |logits| to out[0] (nullable)
// and its gradient to dlogits (nullable).  ws: 2 * 1024 floats.
int mliis_darc1(const float* logits, int N, long long per_img, float weight, float* dlogits, float* out, float* ws, size_t ws_floats,
                hipStream_t stream) {
  MLIIS_REQUIRE(logits && ws && N > 0 && per_img > 0 && per_img < (1LL << 31), MLIIS_ERR_ARG, "darc1: bad arguments");
  MLIIS_REQUIRE(ws_floats >= 2048, MLIIS_ERR_WORKSPACE, "darc1: workspace too small (2048 floats needed, %zu given)", ws_floats);
  int nblk = (int)((per_img + 255) / 256);
  if (nblk > 1024) nblk = 1024;
  float* pval = ws;
  int* pidx = reinterpret_cast<int*>(ws + 1024);
  hipLaunchKernelGGL(darc1_partial_k, dim3(nblk), dim3(256), 0, stream, logits, N, per_img, pval, pidx);
  MLIIS_CHECK_LAUNCH("darc1_partial");
  hipLaunchKernelGGL(darc1_apply_k, dim3(1), dim3(256), 0, stream, logits, N, per_img, pval, pidx, nblk, weight, dlogits, out);
  MLIIS_CHECK_LAUNCH("darc1_apply");
  return MLIIS_OK;
}
}
