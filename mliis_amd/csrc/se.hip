// Squeeze-excite gate MLP (forward and backward) and the per-image/per-channel affine elementwise op that applies the
// gate, tiles pooled vectors and distributes pooled gradients.
// Reference: models/efficientnet/efficientnet_model.py:238-251:  s = mean_hw(x); g = sigmoid(W2 . swish(W1 . s + b1) + b2);
// y = g * x.  The spatial mean itself is mliis_colsum (bn.hip); the big tensors are touched by chan_affine only.
// SE weights are TF HWIO 1x1 kernels: w1 [C][R], w2 [R][C].
#include "common.hpp"
#include "se_wgrad.hpp"

namespace mliis {

constexpr int kMaxR = 128;
constexpr int kSeThreads = 1024;  // one workgroup per image: all 16 waves of a CU work on the tiny MLP to cut its latency

// dot product with a stride, loads issued U at a time (these kernels run one workgroup per image: pure latency, so the number of
// dependent memory round trips is what matters)
template <int U>
__device__ __forceinline__ float dot_strided(const float* __restrict__ a, int sa, const float* __restrict__ b, long long sb, int n) {
  float acc = 0.f;
  for (int i = 0; i < n; i += U) {   // the last batch is padded with clamped, zero-weighted elements: no serial tail
    float x[U], y[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = i + u < n ? i + u : n - 1;
      x[u] = a[(long long)k * sa];
      y[u] = b[(long long)k * sb];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) acc = fmaf(i + u < n ? x[u] : 0.f, y[u], acc);
  }
  return acc;
}

// one workgroup per image.  The pooled vector arrives as `chunks` partial sums per image (s_part [N][chunks][C], from the epilogue
// of the batch-norm apply that produced the activation, or chunks = 1 for a finished vector); phase 0 folds them (x scale) into LDS
// and publishes s [N][C] for the backward pass.
// FAST (C <= 1024, R <= 32, at most kSeFastDot phase-1 terms per thread: every layer of EfficientLab-6-3 / B0): the kernel is one latency
// chain, so every weight a thread will need (its w1 column slice, its w2 column, the biases) is requested before phase 0 instead of
// phase by phase -- the chain is one memory round trip + the three barriers.  Same summation order as the general form (bit-identical).
constexpr int kSeFastDot = 24;
template <bool FAST>
__global__ __launch_bounds__(kSeThreads) void se_mlp_fwd_k(const float* __restrict__ s_part, int chunks, float scale,
                                                    float* __restrict__ s_out, const float* __restrict__ w1,
                                                    const float* __restrict__ b1, const float* __restrict__ w2,
                                                    const float* __restrict__ b2, float* __restrict__ hpre,
                                                    float* __restrict__ gate, int C, int R) {
  __shared__ float sh[kMaxR];
  __shared__ float red[kSeThreads];
  __shared__ float red2[8 * kMaxR];
  extern __shared__ float sn[];   // [C]
  const int n = blockIdx.x, t = threadIdx.x;
  const int CL = kSeThreads / R;
  const int j = t % R, cl = t / R;
  const int cnt = (cl < CL && cl < C) ? (C - cl + CL - 1) / CL : 0;
  float w1v[FAST ? kSeFastDot : 1], w2v[FAST ? 32 : 1], b1v = 0.f, b2v = 0.f;
  if constexpr (FAST) {
    const int cc = t < C ? t : C - 1;
#pragma unroll
    for (int i = 0; i < kSeFastDot; ++i) w1v[i] = w1[(long long)(cl + (i < cnt ? i : 0) * CL < C ? cl + (i < cnt ? i : 0) * CL : 0) * R + j];
#pragma unroll
    for (int k = 0; k < 32; ++k) w2v[k] = w2[(long long)(k < R ? k : R - 1) * C + cc];
    b1v = b1[t < R ? t : 0];
    b2v = b2[cc];
  }
  for (int c = t; c < C; c += kSeThreads) {
    const float* pp = s_part + ((long long)n * chunks) * C + c;
    float a = 0.f;
    for (int k = 0; k < chunks; k += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = pp[(long long)(k + u < chunks ? k + u : chunks - 1) * C];
#pragma unroll
      for (int u = 0; u < 8; ++u) a += k + u < chunks ? v[u] : 0.f;
    }
    a *= scale;
    sn[c] = a;
    if (s_out != nullptr) s_out[(long long)n * C + c] = a;
  }
  __syncthreads();
  // phase 1: h_j = b1[j] + sum_c s[c] * w1[c][j];  threads laid out (c-lane, j) so w1 reads are contiguous
  float part = 0.f;
  if constexpr (FAST) {
#pragma unroll
    for (int i = 0; i < kSeFastDot; ++i) part = fmaf(i < cnt ? sn[cl + i * CL] : 0.f, w1v[i], part);
  } else {
    if (cnt > 0) part = dot_strided<8>(sn + cl, CL, w1 + (long long)cl * R + j, (long long)CL * R, cnt);
  }
  red[t] = part;
  __syncthreads();
  if (t < 8 * R) {   // two-level fold of the CL c-lanes: 8 groups, then 8 values (fixed order)
    const int g = t / R;
    float h = 0.f;
    for (int k = g; k < CL; k += 8) h += red[k * R + j];
    red2[g * R + j] = h;
  }
  __syncthreads();
  if (t < R) {
    float h = FAST ? b1v : b1[t];
#pragma unroll
    for (int g = 0; g < 8; ++g) h += red2[g * R + t];
    hpre[(long long)n * R + t] = h;
    sh[t] = swish_f(h);
  }
  __syncthreads();
  if constexpr (FAST) {
    if (t < C) {
      float a = b2v;
#pragma unroll
      for (int k = 0; k < 32; ++k) a = fmaf(k < R ? sh[k] : 0.f, w2v[k], a);
      gate[(long long)n * C + t] = sigmoid_f(a);
    }
    return;
  }
  for (int c = t; c < C; c += kSeThreads) {
    float a = b2[c];
    for (int k = 0; k < R; k += 8) {
      float wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) wv[u] = w2[(long long)(k + u < R ? k + u : R - 1) * C + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) a = fmaf(k + u < R ? sh[k + u] : 0.f, wv[u], a);
    }
    gate[(long long)n * C + c] = sigmoid_f(a);
  }
}

// one workgroup per image: dgate -> dpre2 [C], dpre1 [R], chan_add = ds / HW [C]
// dgate_groups > 0: dgate holds per-16-row-group partial sums [groups][2][C] from the producer of the gradient
// (mliis_conv2d_bwd_data_gate): slot 0 = rows of the image the group starts in, slot 1 = rows of the next image; image n owns rows
// [n * HW, (n + 1) * HW).  Its groups are folded here in group order (deterministic), all loads of eight groups issued together.
// sums (nullable, [N][sums_nblk][5][C] from mliis_se_bn_bwd_sums): dgate = the image's chunks of value 0, folded here; values 1..4 and
// the finished gate / chan_add give stage 1 of the depthwise batch norm's backward for this image, stage1 [N][2][C].
// w1t (nullable): the [R][C] transpose of w1 (the learner's shadow copy, mliis_transpose_weights) -- the last phase reads a COLUMN of w1
// per channel, which straight from w1 [C][R] is 64 different rows per wave load (measured: the launch's critical path at C = 672).
// FAST (C <= 1024, R <= 32): one channel per thread; the w2 rows of this wave, the w1 column of this thread and hpre are requested
// before the gate gradient is folded (the general form pays a round trip per batch of eight and one more per row for hpre), and an
// image's chunk partials are folded by kSeThreads / C thread groups side by side (chunk b -> group b mod G, groups summed in order).
template <bool FAST>
__global__ __launch_bounds__(kSeThreads) void se_mlp_bwd_k(const float* __restrict__ dgate, int dgate_groups, int HW,
                                                    const float* __restrict__ gate,
                                                    const float* __restrict__ hpre, const float* __restrict__ w1,
                                                    const float* __restrict__ w1t,
                                                    const float* __restrict__ w2, float* __restrict__ dpre2,
                                                    float* __restrict__ dpre1, float* __restrict__ chan_add, int C, int R,
                                                    float inv_hw, const float* __restrict__ sums, int sums_nblk,
                                                    float* __restrict__ stage1) {
  __shared__ float sd1[kMaxR];
  extern __shared__ float sd2[];   // [C] (+ [4][C] with sums; FAST: + [G][5][C] behind them)
  const int n = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  float* d2 = dpre2 + (long long)n * C;
  const int g_lo = dgate_groups > 0 ? (n * HW) / 16 : 0, g_hi = dgate_groups > 0 ? ((n + 1) * HW - 1) / 16 : -1;
  if constexpr (FAST) {
    const int cc = t < C ? t : C - 1;
    const float gte = gate[(long long)n * C + cc];
    float w2v[2][16], w1v[32], hp[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int jj = wave + 16 * r < R ? wave + 16 * r : R - 1;
      hp[r] = hpre[(long long)n * R + jj];
#pragma unroll
      for (int u = 0; u < 16; ++u) w2v[r][u] = w2[(long long)jj * C + (lane + 64 * u < C ? lane + 64 * u : C - 1)];
    }
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      const int kk = k < R ? k : R - 1;
      w1v[k] = w1t != nullptr ? w1t[(long long)kk * C + cc] : w1[(long long)cc * R + kk];
    }
    float dg = 0.f;
    if (sums != nullptr) {
      const int G = kSeThreads / C;
      const int g = t / C, c = t - g * C;
      float* scr = sd2 + 5 * C;
      if (g < G) {
        float p[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        const float* base = sums + ((long long)n * sums_nblk) * 5 * C + c;
        for (int b = g; b < sums_nblk; b += 4 * G) {
          float v[4][5];
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 5; ++k) v[u][k] = base[((long long)(b + u * G < sums_nblk ? b + u * G : b) * 5 + k) * C];
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (b + u * G < sums_nblk) {
#pragma unroll
              for (int k = 0; k < 5; ++k) p[k] += v[u][k];
            }
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) scr[(g * 5 + k) * C + c] = p[k];
      }
      __syncthreads();
      if (t < C) {
        float p[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        for (int gg = 0; gg < G; ++gg) {
#pragma unroll
          for (int k = 0; k < 5; ++k) p[k] += scr[(gg * 5 + k) * C + t];
        }
        dg = p[0];
#pragma unroll
        for (int k = 1; k < 5; ++k) sd2[k * C + t] = p[k];
      }
    } else if (dgate_groups > 0) {
      for (int k = g_lo; k <= g_hi; k += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          const int rg = k + u <= g_hi ? k + u : g_hi;
          const int slot = rg * 16 >= n * HW ? 0 : 1;          // the group starts inside image n, or in the image before it
          v[u] = dgate[((long long)rg * 2 + slot) * C + cc];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) dg += k + u <= g_hi ? v[u] : 0.f;
      }
    } else {
      dg = dgate[(long long)n * C + cc];
    }
    if (t < C) {
      const float v = dg * gte * (1.f - gte);
      d2[t] = v;
      sd2[t] = v;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int jj = wave + 16 * r;
      if (jj < R) {
        float p = 0.f;
#pragma unroll
        for (int u = 0; u < 16; ++u) p = fmaf(lane + 64 * u < C ? sd2[lane + 64 * u] : 0.f, w2v[r][u], p);
        p = wave_sum(p);
        if (lane == 0) {
          const float d = p * swish_grad_f(hp[r]);
          sd1[jj] = d;
          dpre1[(long long)n * R + jj] = d;
        }
      }
    }
    __syncthreads();
    if (t < C) {
      float a = 0.f;
#pragma unroll
      for (int k = 0; k < 32; ++k) a = fmaf(k < R ? sd1[k] : 0.f, w1v[k], a);
      const float ca = a * inv_hw;
      chan_add[(long long)n * C + t] = ca;
      if (sums != nullptr) {   // {sum g, sum g xhat} of this image, g = (da2 * gate + chan_add) * swish'
        stage1[((long long)n * 2 + 0) * C + t] = fmaf(gte, sd2[1 * C + t], ca * sd2[3 * C + t]);
        stage1[((long long)n * 2 + 1) * C + t] = fmaf(gte, sd2[2 * C + t], ca * sd2[4 * C + t]);
      }
    }
    return;
  }
  for (int c = t; c < C; c += kSeThreads) {
    const float g = gate[(long long)n * C + c];
    float dg;
    if (sums != nullptr) {   // chunks of this image in chunk order (deterministic); the five values of a chunk fetched together
      float p[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
      const float* base = sums + ((long long)n * sums_nblk) * 5 * C + c;
      for (int b = 0; b < sums_nblk; b += 2) {
        float v[2][5];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int k = 0; k < 5; ++k) v[u][k] = base[((long long)(b + u < sums_nblk ? b + u : b) * 5 + k) * C];
#pragma unroll
        for (int u = 0; u < 2; ++u)
          if (b + u < sums_nblk) {
#pragma unroll
            for (int k = 0; k < 5; ++k) p[k] += v[u][k];
          }
      }
      dg = p[0];
#pragma unroll
      for (int k = 1; k < 5; ++k) sd2[k * C + c] = p[k];
    } else if (dgate_groups > 0) {
      dg = 0.f;
      for (int k = g_lo; k <= g_hi; k += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int rg = k + u <= g_hi ? k + u : g_hi;
          const int slot = rg * 16 >= n * HW ? 0 : 1;          // the group starts inside image n, or in the image before it
          v[u] = dgate[((long long)rg * 2 + slot) * C + c];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) dg += k + u <= g_hi ? v[u] : 0.f;
      }
    } else {
      dg = dgate[(long long)n * C + c];
    }
    const float v = dg * g * (1.f - g);
    d2[c] = v;
    sd2[c] = v;
  }
  __syncthreads();
  for (int jj = wave; jj < R; jj += kSeThreads / 64) {
    const float* wr = w2 + (long long)jj * C;
    float p = 0.f;
    for (int c = lane; c < C; c += 8 * 64) {
      float wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) wv[u] = wr[c + u * 64 < C ? c + u * 64 : lane];
#pragma unroll
      for (int u = 0; u < 8; ++u) p = fmaf(c + u * 64 < C ? sd2[c + u * 64] : 0.f, wv[u], p);
    }
    p = wave_sum(p);
    if (lane == 0) {
      const float d = p * swish_grad_f(hpre[(long long)n * R + jj]);
      sd1[jj] = d;
      dpre1[(long long)n * R + jj] = d;
    }
  }
  __syncthreads();
  for (int c = t; c < C; c += kSeThreads) {
    float a = 0.f;
    for (int k = 0; k < R; k += 8) {
      float wv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int kk = k + u < R ? k + u : R - 1;
        wv[u] = w1t != nullptr ? w1t[(long long)kk * C + c] : w1[(long long)c * R + kk];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) a = fmaf(k + u < R ? sd1[k + u] : 0.f, wv[u], a);
    }
    const float ca = a * inv_hw;
    chan_add[(long long)n * C + c] = ca;
    if (sums != nullptr) {   // {sum g, sum g xhat} of this image, g = (da2 * gate + chan_add) * swish'
      const float gt = gate[(long long)n * C + c];
      stage1[((long long)n * 2 + 0) * C + c] = fmaf(gt, sd2[1 * C + c], ca * sd2[3 * C + c]);
      stage1[((long long)n * 2 + 1) * C + c] = fmaf(gt, sd2[2 * C + c], ca * sd2[4 * C + c]);
    }
  }
}

__global__ __launch_bounds__(256) void se_wgrad_k(const float* __restrict__ s, const float* __restrict__ hpre,
                                                  const float* __restrict__ dpre1, const float* __restrict__ dpre2,
                                                  float* __restrict__ dw1, float* __restrict__ db1, float* __restrict__ dw2,
                                                  float* __restrict__ db2, int N, int C, int R) {
  se_wgrad_elem(blockIdx.x * 256 + threadIdx.x, s, hpre, dpre1, dpre2, dw1, db1, dw2, db2, N, C, R);
}

// The squeeze-excite weight gradients of ALL blocks of a backward pass in one launch (nothing downstream needs them before the
// optimizer).  desc: device int64 [ndesc][12] = {s, hpre, dpre1, dpre2, dw1, db1, dw2, db2 (device addresses), N, C, R, tile_begin};
// block b handles 256 consecutive elements of the descriptor whose tile range contains b.
__global__ __launch_bounds__(256) void se_wgrad_batched_k(const long long* __restrict__ desc, int ndesc) {
  se_wgrad_tile(desc, ndesc, blockIdx.x);
}

// y[m, c] (+)= x[m, c] * S[n(m), c] + A[n(m), c]      (x, S, A each optional)
__global__ __launch_bounds__(256) void chan_affine_k(const float* __restrict__ x, int ldx, const float* __restrict__ S,
                                                     const float* __restrict__ A, float* __restrict__ y, int ldy, long long rows,
                                                     int C, int rows_per_img, int accumulate) {
  const unsigned Q = (unsigned)C >> 2;
  const unsigned total = (unsigned)rows * Q;   // host guarantees < 2^31 (32-bit index math)
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned ru = i / Q;
    const int c = (int)(i - ru * Q) << 2;
    const long long r = ru;
    const long long n = ru / (unsigned)rows_per_img;
    // every operand is fetched before any is used (absent ones shadow the output row: valid, ignored) -- one memory round trip
    float* dst = y + r * ldy + c;
    const float4 vx = ld4(x ? x + r * ldx + c : dst);
    const float4 vs = ld4(S ? S + n * C + c : dst);
    const float4 va = ld4(A ? A + n * C + c : dst);
    const float4 vo = ld4(dst);
    float4 v = x ? vx : f4zero();
    if (S) v = f4mul(v, vs);
    if (A) v = f4add(v, va);
    if (accumulate) v = f4add(v, vo);
    st4(dst, v);
  }
}

// One pass over x [rows][C] (+ A[n(m), c]) routed to two destinations by channel: y0[m, c] for c < c0, y1[m, c - c0] otherwise,
// each stored or accumulated -- the tail of the RSD module's backward (gradient of the concat: + the pooled branch's per-image term,
// deep half added to the residual gradient, skip half to the endpoint's gradient) in one launch instead of three.
__global__ __launch_bounds__(256) void chan_split_k(const float* __restrict__ x, int ldx, const float* __restrict__ A, float* __restrict__ y0,
                                                    int ld0, int c0, int acc0, float* __restrict__ y1, int ld1, int acc1, long long rows, int C,
                                                    int rows_per_img) {
  const unsigned Q = (unsigned)C >> 2;
  const unsigned total = (unsigned)rows * Q;   // host guarantees < 2^31
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const unsigned ru = i / Q;
    const int c = (int)(i - ru * Q) << 2;
    const long long r = ru;
    const long long n = ru / (unsigned)rows_per_img;
    const bool first = c < c0;
    float* dst = first ? y0 + r * ld0 + c : y1 + r * ld1 + (c - c0);
    const bool acc = first ? acc0 != 0 : acc1 != 0;
    const float4 vx = ld4(x + r * ldx + c);
    const float4 va = ld4(A ? A + n * C + c : x + r * ldx + c);
    const float4 vo = ld4(dst);          // (read even when it is overwritten: one round trip, no branch around a load)
    float4 v = vx;
    if (A) v = f4add(v, va);
    if (acc) v = f4add(v, vo);
    st4(dst, v);
  }
}

}  // namespace mliis

using namespace mliis;

extern "C" {

int mliis_se_mlp_fwd(const float* s_part, int chunks, float scale, float* s_out, const float* w1, const float* b1, const float* w2,
                     const float* b2, float* hpre, float* gate, int N, int C, int R, hipStream_t stream) {
  MLIIS_REQUIRE(s_part && w1 && b1 && w2 && b2 && hpre && gate, MLIIS_ERR_ARG, "se_mlp_fwd: null pointer");
  MLIIS_REQUIRE(N > 0 && C > 0 && C <= 8192 && R > 0 && R <= kMaxR && chunks > 0, MLIIS_ERR_ARG, "se_mlp_fwd: bad shape (R <= %d, C <= 8192)",
                kMaxR);
  const int CL = kSeThreads / R;
  if (C <= kSeThreads && R <= 32 && (C + CL - 1) / CL <= kSeFastDot)
    hipLaunchKernelGGL(se_mlp_fwd_k<true>, dim3(N), dim3(kSeThreads), (size_t)C * sizeof(float), stream, s_part, chunks, scale, s_out, w1, b1, w2,
                       b2, hpre, gate, C, R);
  else
    hipLaunchKernelGGL(se_mlp_fwd_k<false>, dim3(N), dim3(kSeThreads), (size_t)C * sizeof(float), stream, s_part, chunks, scale, s_out, w1, b1, w2,
                       b2, hpre, gate, C, R);
  MLIIS_CHECK_LAUNCH("se_mlp_fwd");
  return MLIIS_OK;
}

// dgate[N,C] = sum_hw dy * x (mliis_colsum).  Produces dpre1 [N,R], dpre2 [N,C] (scratch kept for the weight gradients),
// chan_add [N,C] = (dL/ds)/HW, and the four weight gradients.
int mliis_se_mlp_bwd(const float* dgate, int dgate_row_groups, const float* gate, const float* s, const float* hpre, const float* w1,
                     const float* w1t, const float* w2, float* dpre1, float* dpre2, float* chan_add, float* dw1, float* db1, float* dw2, float* db2, int N,
                     int C, int R, int HW, hipStream_t stream) {
  MLIIS_REQUIRE(dgate_row_groups == 0 || (HW >= 16 && (long long)dgate_row_groups * 16 >= (long long)N * HW), MLIIS_ERR_ARG,
                "se_mlp_bwd: the row-group partials do not cover the N * HW rows (maps of at least 16 pixels)");
  MLIIS_REQUIRE(dgate && gate && s && hpre && w1 && w2 && dpre1 && dpre2 && chan_add, MLIIS_ERR_ARG, "se_mlp_bwd: null pointer (w1t alone may be NULL)");
  MLIIS_REQUIRE((dw1 && db1 && dw2 && db2) || (!dw1 && !db1 && !dw2 && !db2), MLIIS_ERR_ARG,
                "se_mlp_bwd: the four weight-gradient outputs come together (all NULL = deferred to mliis_se_wgrad_batched)");
  MLIIS_REQUIRE(N > 0 && C > 0 && R > 0 && R <= kMaxR && HW > 0, MLIIS_ERR_ARG, "se_mlp_bwd: bad shape");
  MLIIS_REQUIRE(C <= 8192, MLIIS_ERR_UNSUPPORTED, "se_mlp_bwd: C > 8192");
  if (C <= kSeThreads && R <= 32)
    hipLaunchKernelGGL(se_mlp_bwd_k<true>, dim3(N), dim3(kSeThreads), (size_t)C * sizeof(float), stream, dgate, dgate_row_groups, HW, gate, hpre, w1,
                       w1t, w2, dpre2, dpre1, chan_add, C, R, 1.0f / (float)HW, nullptr, 0, nullptr);
  else
    hipLaunchKernelGGL(se_mlp_bwd_k<false>, dim3(N), dim3(kSeThreads), (size_t)C * sizeof(float), stream, dgate, dgate_row_groups, HW, gate, hpre, w1,
                       w1t, w2, dpre2, dpre1, chan_add, C, R, 1.0f / (float)HW, nullptr, 0, nullptr);
  MLIIS_CHECK_LAUNCH("se_mlp_bwd");
  if (dw1 == nullptr) return MLIIS_OK;
  int total = 2 * C * R + C + R;
  hipLaunchKernelGGL(se_wgrad_k, dim3(ceil_div(total, 256)), dim3(256), 0, stream, s, hpre, dpre1, dpre2, dw1, db1, dw2, db2, N, C, R);
  MLIIS_CHECK_LAUNCH("se_wgrad");
  return MLIIS_OK;
}

// The squeeze-excite backward fed by mliis_se_bn_bwd_sums (sums [N][sums_nblk][5][C]): the gate's gradient is folded from value 0, and
// stage 1 of the depthwise batch norm's backward leaves as stage1 [N][2][C] for mliis_bn_bwd(chan_scale = gate, chan_add,
// stage1_part = stage1, stage1_nblk = N).  The weight gradients are left to mliis_se_wgrad_batched.
int mliis_se_mlp_bwd_bn(const float* sums, int sums_nblk, const float* gate, const float* hpre, const float* w1, const float* w1t, const float* w2,
                        float* dpre1, float* dpre2, float* chan_add, float* stage1, int N, int C, int R, int HW, hipStream_t stream) {
  MLIIS_REQUIRE(sums && sums_nblk > 0 && gate && hpre && w1 && w2 && dpre1 && dpre2 && chan_add && stage1, MLIIS_ERR_ARG, "se_mlp_bwd_bn: null pointer");
  MLIIS_REQUIRE(N > 0 && C > 0 && C <= 8192 && R > 0 && R <= kMaxR && HW > 0, MLIIS_ERR_ARG, "se_mlp_bwd_bn: bad shape (R <= %d, C <= 8192)", kMaxR);
  if (C <= kSeThreads && R <= 32)
    hipLaunchKernelGGL(se_mlp_bwd_k<true>, dim3(N), dim3(kSeThreads), (size_t)(5 * C + 5 * kSeThreads) * sizeof(float), stream, nullptr, 0, HW, gate, hpre, w1,
                       w1t, w2, dpre2,
                       dpre1, chan_add, C, R, 1.0f / (float)HW, sums, sums_nblk, stage1);
  else
    hipLaunchKernelGGL(se_mlp_bwd_k<false>, dim3(N), dim3(kSeThreads), (size_t)5 * C * sizeof(float), stream, nullptr, 0, HW, gate, hpre, w1, w1t, w2, dpre2,
                       dpre1, chan_add, C, R, 1.0f / (float)HW, sums, sums_nblk, stage1);
  MLIIS_CHECK_LAUNCH("se_mlp_bwd_bn");
  return MLIIS_OK;
}

int mliis_se_wgrad_batched(const long long* desc, int ndesc, long long total_tiles, hipStream_t stream) {
  MLIIS_REQUIRE(desc && ndesc > 0 && total_tiles > 0, MLIIS_ERR_ARG, "se_wgrad_batched: bad arguments");
  hipLaunchKernelGGL(se_wgrad_batched_k, dim3((unsigned)total_tiles), dim3(256), 0, stream, desc, ndesc);
  MLIIS_CHECK_LAUNCH("se_wgrad_batched");
  return MLIIS_OK;
}

int mliis_chan_split(const float* x, int ldx, const float* A, float* y0, int ld0, int c0, int accumulate0, float* y1, int ld1, int accumulate1,
                     long long rows, int C, int rows_per_img, hipStream_t stream) {
  MLIIS_REQUIRE(x && y0 && y1, MLIIS_ERR_ARG, "chan_split: null pointer");
  MLIIS_REQUIRE(rows > 0 && C > 0 && (C & 3) == 0 && c0 > 0 && c0 < C && (c0 & 3) == 0 && (ldx & 3) == 0 && ldx >= C && (ld0 & 3) == 0 &&
                    ld0 >= c0 && (ld1 & 3) == 0 && ld1 >= C - c0 && rows_per_img > 0,
                MLIIS_ERR_ARG, "chan_split: bad shape");
  MLIIS_REQUIRE(aligned16(x) && aligned16(A) && aligned16(y0) && aligned16(y1), MLIIS_ERR_ALIGN, "chan_split: pointers must be 16-byte aligned");
  MLIIS_REQUIRE(rows * (C / 4) < (1LL << 31), MLIIS_ERR_UNSUPPORTED, "chan_split: tensor too large for 32-bit indexing");
  long long q = rows * (C / 4);
  int blocks = (int)((q + 255) / 256 > 4096 ? 4096 : (q + 255) / 256);
  hipLaunchKernelGGL(chan_split_k, dim3(blocks), dim3(256), 0, stream, x, ldx, A, y0, ld0, c0, accumulate0, y1, ld1, accumulate1, rows, C,
                     rows_per_img);
  MLIIS_CHECK_LAUNCH("chan_split");
  return MLIIS_OK;
}

int mliis_chan_affine(const float* x, int ldx, const float* S, const float* A, float* y, int ldy, long long rows, int C,
                      int rows_per_img, int accumulate, hipStream_t stream) {
  MLIIS_REQUIRE(y && (x || A), MLIIS_ERR_ARG, "chan_affine: need an output and at least one of x / A");
  MLIIS_REQUIRE(rows > 0 && C > 0 && (C & 3) == 0 && (ldy & 3) == 0 && ldy >= C && rows_per_img > 0 &&
                    (x == nullptr || ((ldx & 3) == 0 && ldx >= C)),
                MLIIS_ERR_ARG, "chan_affine: bad shape");
  MLIIS_REQUIRE(aligned16(x) && aligned16(S) && aligned16(A) && aligned16(y), MLIIS_ERR_ALIGN, "chan_affine: pointers must be 16-byte aligned");
  MLIIS_REQUIRE(rows * (C / 4) < (1LL << 31), MLIIS_ERR_UNSUPPORTED, "chan_affine: tensor too large for 32-bit indexing");
  long long q = rows * (C / 4);
  int blocks = (int)((q + 255) / 256 > 4096 ? 4096 : (q + 255) / 256);
  hipLaunchKernelGGL(chan_affine_k, dim3(blocks), dim3(256), 0, stream, x, ldx, S, A, y, ldy, rows, C, rows_per_img, accumulate);
  MLIIS_CHECK_LAUNCH("chan_affine");
  return MLIIS_OK;
}
}
