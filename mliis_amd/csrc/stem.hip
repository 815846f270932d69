// Stem: input normalisation fused into the 3x3 stride-2 TF-SAME conv 3 -> Co (no bias).
// Reference: models/efficientlab.py:113-114 ((x - MEAN_RGB) / STDDEV_RGB on 0..255 inputs, zero padding is applied to the
// NORMALISED image) and models/efficientnet/efficientnet_model.py:359-366,411-412.  K = 27: far too skinny for the matrix
// cores and HBM-bound, so this is a direct kernel; the batch is addressed through an optional index vector so the inner
// loop never re-stacks images (meta_learners/metaseg.py:285-302 wrap-around batches are index lists here).
#include "common.hpp"

namespace mliis {

struct StemGeom {
  int Ho, Wo, pt, pl;
};
static inline StemGeom stem_geom(int H, int W) {
  StemGeom g;
  g.Ho = (H + 1) / 2;
  g.Wo = (W + 1) / 2;
  int th = (g.Ho - 1) * 2 + 3 - H, tw = (g.Wo - 1) * 2 + 3 - W;
  if (th < 0) th = 0;
  if (tw < 0) tw = 0;
  g.pt = th / 2;
  g.pl = tw / 2;
  return g;
}

struct Norm3 {
  float m0, m1, m2, i0, i1, i2;  // mean (m*) and std (i*) per channel
};

// the 27 raw values of a 3x3x3 window, fetched together without branches (out-of-image taps read a clamped, valid pixel and are
// zeroed afterwards: the padding applies to the NORMALISED image), then normalised
__device__ __forceinline__ void load_window(const float* __restrict__ img, int H, int W, int hi0, int wi0, Norm3 nm, float* v) {
  bool ok[9];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int hi = hi0 + ky;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int wi = wi0 + kx;
      ok[ky * 3 + kx] = hi >= 0 && hi < H && wi >= 0 && wi < W;
      const int hc = hi < 0 ? 0 : (hi >= H ? H - 1 : hi), wc = wi < 0 ? 0 : (wi >= W ? W - 1 : wi);
      const float* px = img + ((long long)hc * W + wc) * 3;
      v[(ky * 3 + kx) * 3 + 0] = px[0];
      v[(ky * 3 + kx) * 3 + 1] = px[1];
      v[(ky * 3 + kx) * 3 + 2] = px[2];
    }
  }
  // TF divides by std; (x - m) / s is restated as a true division to stay within 1 ulp of it
#pragma unroll
  for (int tp = 0; tp < 9; ++tp) {
    v[tp * 3 + 0] = ok[tp] ? (v[tp * 3 + 0] - nm.m0) / nm.i0 : 0.f;
    v[tp * 3 + 1] = ok[tp] ? (v[tp * 3 + 1] - nm.m1) / nm.i1 : 0.f;
    v[tp * 3 + 2] = ok[tp] ? (v[tp * 3 + 2] - nm.m2) / nm.i2 : 0.f;
  }
}

// thread = (pixel, channel octet)
__global__ __launch_bounds__(256) void stem_fwd_k(const float* __restrict__ x, const int* __restrict__ idx,
                                                  const float* __restrict__ w, float* __restrict__ z, int N, int H, int W, int Ho,
                                                  int Wo, int Co, int pt, int pl, Norm3 nm) {
  extern __shared__ float sw[];  // [27][Co]
  for (int i = threadIdx.x; i < 27 * Co; i += 256) sw[i] = w[i];
  __syncthreads();
  const int OPP = Co >> 3;
  const long long total = (long long)N * Ho * Wo * OPP;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int oct = (int)(i % OPP);
  long long pix = i / OPP;
  const int wo = (int)(pix % Wo);
  long long r = pix / Wo;
  const int ho = (int)(r % Ho);
  const int n = (int)(r / Ho);
  const int src = idx ? idx[n] : n;
  float v[27];
  load_window(x + (long long)src * H * W * 3, H, W, ho * 2 - pt, wo * 2 - pl, nm, v);
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    const float4 w0 = ld4(sw + k * Co + oct * 8), w1 = ld4(sw + k * Co + oct * 8 + 4);
    acc[0] = fmaf(v[k], w0.x, acc[0]);
    acc[1] = fmaf(v[k], w0.y, acc[1]);
    acc[2] = fmaf(v[k], w0.z, acc[2]);
    acc[3] = fmaf(v[k], w0.w, acc[3]);
    acc[4] = fmaf(v[k], w1.x, acc[4]);
    acc[5] = fmaf(v[k], w1.y, acc[5]);
    acc[6] = fmaf(v[k], w1.z, acc[6]);
    acc[7] = fmaf(v[k], w1.w, acc[7]);
  }
  float* dst = z + pix * Co + oct * 8;
  st4(dst, make_float4(acc[0], acc[1], acc[2], acc[3]));
  st4(dst + 4, make_float4(acc[4], acc[5], acc[6], acc[7]));
}

// Row-strip form with the NEXT batch norm's stage-1 statistics (round 6).  stem_fwd_k gives every (pixel, channel octet) thread its own
// 27 loads and 27 divisions -- each input value is fetched and normalised ~2.25 x Co / 8 times -- and the statistics of z were a launch
// of their own over the 12.8 MB it had just written.  Here a workgroup owns (image, a run of output rows) and walks it four output rows
// at a time: the nine input rows under them are fetched once (contiguous 4-byte lanes), normalised once and staged in LDS with the
// zero padding in place; thread = (pixel lane, channel octet) as before, the window comes from LDS (same fma order: z is bit-identical
// to stem_fwd_k's), and the thread keeps sum / sum of squares of its octet, folded through LDS in lane order at the end:
// part [workgroup][2][Co], the layout of mliis_bn_stats_partial.  At most ~256 workgroups (= partial blocks the consumer folds).
constexpr int kStemRows = 4;        // output rows per staging round
constexpr int kStemRowsThreads = 512;
__global__ __launch_bounds__(kStemRowsThreads) void stem_fwd_rows_k(const float* __restrict__ x, const int* __restrict__ idx,
                                                                    const float* __restrict__ w, float* __restrict__ z, int H, int W,
                                                                    int Ho, int Wo, int Co, int pt, int pl, Norm3 nm,
                                                                    int rows_per_block, int blocks_per_img, float* __restrict__ part) {
  extern __shared__ float sdyn[];
  constexpr int NT = kStemRowsThreads;
  const int IW = 2 * Wo + 1, IR = 2 * kStemRows + 1;
  float* sx = sdyn;                        // [IR][IW][3] normalised, zero outside the image
  float* red = sx + IR * IW * 3;           // [NT][8]
  const int t = threadIdx.x;
  const int n = blockIdx.x / blocks_per_img, chunk = blockIdx.x - n * blocks_per_img;
  const int src = idx ? idx[n] : n;
  const float* img = x + (long long)src * H * W * 3;
  const int Q = Co >> 2, TPW = NT / Q;               // channel quads per pixel, pixel lanes
  const bool live = t < TPW * Q;
  const int quad = live ? t % Q : 0, lane_px = t / Q;
  float4 wr[27];                                     // the thread's 27 x 4 weights stay in registers for the whole strip
#pragma unroll
  for (int k = 0; k < 27; ++k) wr[k] = ld4(w + k * Co + quad * 4);
  const int ho_end = (chunk + 1) * rows_per_block < Ho ? (chunk + 1) * rows_per_block : Ho;
  float4 s1 = f4zero(), s2 = f4zero();
  for (int ho0 = chunk * rows_per_block; ho0 < ho_end; ho0 += kStemRows) {
    if (ho0 != chunk * rows_per_block) __syncthreads();   // the previous round's window reads are done
    const int hi0 = ho0 * 2 - pt;
    constexpr int FB = 12;   // loads of a thread in flight per round trip (a plain loop waits for every value before it asks for the next)
    for (int e0 = t; e0 < IR * IW * 3; e0 += NT * FB) {
      float raw[FB];
      int chs[FB];
      bool oks[FB];
#pragma unroll
      for (int u = 0; u < FB; ++u) {
        const int e = e0 + u * NT;
        const int r = e / (IW * 3), rem = e - r * (IW * 3);
        const int col = rem / 3, ch = rem - col * 3;
        const int hi = hi0 + r, wi = col - pl;
        oks[u] = e < IR * IW * 3 && hi >= 0 && hi < H && wi >= 0 && wi < W;
        chs[u] = ch;
        raw[u] = img[oks[u] ? ((long long)hi * W + wi) * 3 + ch : 0];
      }
#pragma unroll
      for (int u = 0; u < FB; ++u) {
        const int e = e0 + u * NT;
        const float m = chs[u] == 0 ? nm.m0 : (chs[u] == 1 ? nm.m1 : nm.m2), sd = chs[u] == 0 ? nm.i0 : (chs[u] == 1 ? nm.i1 : nm.i2);
        if (e < IR * IW * 3) sx[e] = oks[u] ? (raw[u] - m) / sd : 0.f;   // (a true division, as stem_fwd_k: within 1 ulp of TF's)
      }
    }
    __syncthreads();
    const int npix = (ho_end - ho0 < kStemRows ? ho_end - ho0 : kStemRows) * Wo;
    if (live)
      for (int p = lane_px; p < npix; p += TPW) {
        const int ro = p / Wo, wo = p - ro * Wo;
        float4 acc = f4zero();
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const float* row = sx + ((ro * 2 + ky) * IW + wo * 2) * 3;
#pragma unroll
          for (int q = 0; q < 9; ++q) {
            const float v = row[q];
            const float4 wk = wr[ky * 9 + q];
            acc.x = fmaf(v, wk.x, acc.x);
            acc.y = fmaf(v, wk.y, acc.y);
            acc.z = fmaf(v, wk.z, acc.z);
            acc.w = fmaf(v, wk.w, acc.w);
          }
        }
        st4(z + (((long long)n * Ho + ho0 + ro) * Wo + wo) * Co + quad * 4, acc);
        s1 = f4add(s1, acc);
        s2 = f4fma(acc, acc, s2);
      }
  }
  if (part == nullptr) return;   // (uniform)
  st4(red + t * 8, s1);
  st4(red + t * 8 + 4, s2);
  __syncthreads();
  if (t < Q * 8) {   // the pixel lanes in lane order (deterministic)
    const int o = t >> 3, j = t & 7;
    float a = 0.f;
    for (int l = 0; l < TPW; ++l) a += red[(l * Q + o) * 8 + j];
    part[((long long)blockIdx.x * 2 + (j >> 2)) * Co + o * 4 + (j & 3)] = a;
  }
}

// thread = (pixel lane, channel quad): 27 x float4 accumulators.  The block walks its pixels PL at a time: the 27 normalised window
// values of those PL pixels are fetched ONCE, cooperatively, into LDS (instead of once per channel-quad thread) and every thread
// reads its pixel's window from there (8-way broadcast, conflict-free stride 27).  Block reduction: butterfly over the pixel lanes
// that share a wave, then the waves through LDS in a fixed order.  part layout [blk][27][Co].
// The quad index is padded to a power of two (QP >= Co/4; surplus lanes idle) so the lanes of one quad are a fixed xor pattern.
__global__ __launch_bounds__(256) void stem_bwd_filter_k(const float* __restrict__ x, const int* __restrict__ idx,
                                                         const float* __restrict__ dz, float* __restrict__ part, int N, int H,
                                                         int W, int Ho, int Wo, int Co, int pt, int pl, Norm3 nm,
                                                         int pix_per_block, int QP) {
  extern __shared__ float4 sred[];   // [4 waves][27][QC], then the window stage [PL][27] floats
  const int QC = Co >> 2;            // channel quads
  const int PL = 256 / QP;           // pixel lanes
  float* swin = reinterpret_cast<float*>(sred + 4 * 27 * QC);
  const int q = threadIdx.x & (QP - 1), pl_ = threadIdx.x / QP;
  const bool active = q < QC;
  const long long P = (long long)N * Ho * Wo;
  float4 acc[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) acc[k] = f4zero();
  const long long p0 = (long long)blockIdx.x * pix_per_block;
  long long p1 = p0 + pix_per_block;
  if (p1 > P) p1 = P;
  const float mean[3] = {nm.m0, nm.m1, nm.m2}, sd[3] = {nm.i0, nm.i1, nm.i2};
  for (long long pg = p0; pg < p1; pg += PL) {
    for (int e = threadIdx.x; e < PL * 27; e += 256) {
      const int pp = e / 27, tap = e - pp * 27;
      const long long pix = pg + pp;
      float v = 0.f;
      if (pix < p1) {
        const int wo = (int)(pix % Wo);
        const long long r = pix / Wo;
        const int ho = (int)(r % Ho);
        const int n = (int)(r / Ho);
        const int src = idx ? idx[n] : n;
        const int ky = tap / 9, rem = tap - ky * 9, kx = rem / 3, ci = rem - kx * 3;
        const int hi = ho * 2 - pt + ky, wi = wo * 2 - pl + kx;
        // TF divides by std; (x - m) / s is restated as a true division to stay within 1 ulp of it
        if (hi >= 0 && hi < H && wi >= 0 && wi < W) v = (x[(((long long)src * H + hi) * W + wi) * 3 + ci] - mean[ci]) / sd[ci];
      }
      swin[e] = v;
    }
    __syncthreads();
    const long long pix = pg + pl_;
    if (active && pix < p1) {
      const float4 d = ld4(dz + pix * Co + q * 4);
      const float* wv = swin + pl_ * 27;
#pragma unroll
      for (int k = 0; k < 27; ++k) {
        const float v = wv[k];
        acc[k] = f4fma(make_float4(v, v, v, v), d, acc[k]);
      }
    }
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 27; ++k) {
    for (int off = QP; off < 64; off <<= 1) {
      acc[k].x += __shfl_xor(acc[k].x, off);
      acc[k].y += __shfl_xor(acc[k].y, off);
      acc[k].z += __shfl_xor(acc[k].z, off);
      acc[k].w += __shfl_xor(acc[k].w, off);
    }
    if (lane < QP && active) sred[(wave * 27 + k) * QC + q] = acc[k];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 27 * QC; i += 256) {
    const float4 s4 = f4add(f4add(sred[i], sred[27 * QC + i]), f4add(sred[2 * 27 * QC + i], sred[3 * 27 * QC + i]));
    st4(part + (long long)blockIdx.x * 27 * Co + (long long)i * 4, s4);
  }
}

// Matrix-core version for Co <= 64 (EfficientNet-B0 / B3 stems: 32 / 40 channels): dW[tap][co] = sum_pixels win[pixel][tap] * dz[pixel][co]
// is a [32 x pixels] x [pixels x Co] product (27 taps padded to 32).  A block stages the normalised windows of ALL its pixels in LDS in
// one round (pixel coordinates first, so the 27 loads per pixel need no integer division), then every wave multiplies a quarter of the
// pixels with v_mfma_f32_16x16x4_f32 -- lane (l15, g) feeds tap l15 (+16) of pixel p + g and channel l15 (+16 j) of the same pixel,
// dz straight from memory -- and the four waves' tiles are added through LDS in a fixed order.  part layout [blk][27][Co].
constexpr int kStemPix = 256;   // pixels per block (upper bound of pix_per_block)
typedef float stem_f32x4 __attribute__((ext_vector_type(4)));

template <int NTC>
__global__ __launch_bounds__(256) void stem_bwd_filter_mfma_k(const float* __restrict__ x, const int* __restrict__ idx,
                                                              const float* __restrict__ dz, float* __restrict__ part, int N, int H, int W,
                                                              int Ho, int Wo, int Co, int pt, int pl, Norm3 nm, int pix_per_block) {
  __shared__ int s_base[kStemPix], s_h0[kStemPix], s_w0[kStemPix];
  __shared__ float swin[kStemPix * 32];
  __shared__ float red[4][32][16 * NTC + 1];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l15 = lane & 15, g = lane >> 4;
  const long long P = (long long)N * Ho * Wo;
  const long long p0 = (long long)blockIdx.x * pix_per_block;
  const int npix = (int)((p0 + pix_per_block > P ? P : p0 + pix_per_block) - p0);
  if (t < npix) {
    const long long pix = p0 + t;
    const int wo = (int)(pix % Wo);
    const long long r = pix / Wo;
    const int ho = (int)(r % Ho), n = (int)(r / Ho);
    s_base[t] = (idx ? idx[n] : n) * H * W * 3;     // host guarantees the image tensor is below 2^31 floats
    s_h0[t] = ho * 2 - pt;
    s_w0[t] = wo * 2 - pl;
  }
  __syncthreads();
  // window staging: thread = (tap = t & 31, pixel row t >> 5); eight pixels per thread and round trip, loads issued together
  {
    const int tap = t & 31, prow = t >> 5;
    const int ky = tap / 9, rem = tap - ky * 9, kx = rem / 3, ci = rem - kx * 3;
    const float mu = ci == 0 ? nm.m0 : (ci == 1 ? nm.m1 : nm.m2), sg = ci == 0 ? nm.i0 : (ci == 1 ? nm.i1 : nm.i2);
    constexpr int U = 8;
    for (int i0 = 0; i0 < npix; i0 += 8 * U) {
      float v[U];
      bool ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int pp = i0 + prow + 8 * u;
        const int ps = pp < npix ? pp : 0;
        const int hi = s_h0[ps] + ky, wi = s_w0[ps] + kx;
        ok[u] = pp < npix && tap < 27 && hi >= 0 && hi < H && wi >= 0 && wi < W;
        v[u] = x[ok[u] ? (long long)s_base[ps] + ((long long)hi * W + wi) * 3 + ci : 0];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int pp = i0 + prow + 8 * u;
        // TF divides by std; (x - m) / s is restated as a true division to stay within 1 ulp of it
        if (pp < npix) swin[pp * 32 + tap] = ok[u] ? (v[u] - mu) / sg : 0.f;
      }
    }
  }
  __syncthreads();
  stem_f32x4 acc[2][NTC];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NTC; ++j) acc[i][j] = (stem_f32x4){0.f, 0.f, 0.f, 0.f};
  const int per_wave = (npix + 3) / 4;
  const int w0 = wave * per_wave, w1 = w0 + per_wave < npix ? w0 + per_wave : npix;
  for (int pb = w0; pb < w1; pb += 4) {
    const int pp = pb + g;
    const bool ok = pp < w1;
    const int ps = ok ? pp : w0;
    const float a0 = ok ? swin[ps * 32 + l15] : 0.f, a1 = ok ? swin[ps * 32 + 16 + l15] : 0.f;
    float b[NTC];
#pragma unroll
    for (int j = 0; j < NTC; ++j) {
      const int co = j * 16 + l15;
      b[j] = (ok && co < Co) ? dz[(p0 + ps) * Co + co] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < NTC; ++j) {
      acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b[j], acc[0][j], 0, 0, 0);
      acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b[j], acc[1][j], 0, 0, 0);
    }
  }
  // C/D layout: column l15, rows 4 g + r
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NTC; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wave][i * 16 + g * 4 + r][j * 16 + l15] = acc[i][j][r];
  __syncthreads();
  for (int e = t; e < 27 * Co; e += 256) {
    const int tap = e / Co, co = e - tap * Co;
    part[(long long)blockIdx.x * 27 * Co + e] = (red[0][tap][co] + red[1][tap][co]) + (red[2][tap][co] + red[3][tap][co]);
  }
}

static inline int stem_quad_pad(int Co) {
  int qp = 1;
  while (qp < Co / 4) qp <<= 1;
  return qp;
}

static inline void stem_filter_geom(int N, int Ho, int Wo, int Co, int* pix_per_block, int* nblk) {
  long long P = (long long)N * Ho * Wo;
  int PL = 256 / stem_quad_pad(Co);
  const int nb = 512;   // measured (tools/stem_probe.py): 128 -> 72 us, 256 -> 49, 392 -> 35, 512 -> 32, 784+ -> 39 (block reduction cost)
  long long ppb = (P + nb - 1) / nb;
  if (ppb < PL * 4) ppb = PL * 4;
  ppb = (ppb + PL - 1) / PL * PL;
  *pix_per_block = (int)ppb;
  *nblk = (int)((P + ppb - 1) / ppb);
}

}  // namespace mliis

using namespace mliis;

extern "C" {

int mliis_stem_conv_fwd(const float* x, const int* img_idx, const float* w, float* z, int N, int H, int W, int Co, const float* mean3,
                        const float* std3, hipStream_t stream) {
  MLIIS_REQUIRE(x && w && z && mean3 && std3, MLIIS_ERR_ARG, "stem_conv_fwd: null pointer");
  MLIIS_REQUIRE(N > 0 && H > 1 && W > 1 && Co > 0 && (Co & 7) == 0 && Co <= 256, MLIIS_ERR_ARG, "stem_conv_fwd: bad shape (Co %% 8 == 0 required)");
  MLIIS_REQUIRE(aligned16(z), MLIIS_ERR_ALIGN, "stem_conv_fwd: output must be 16-byte aligned");
  StemGeom g = stem_geom(H, W);
  Norm3 nm{mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]};
  long long total = (long long)N * g.Ho * g.Wo * (Co / 8);
  hipLaunchKernelGGL(stem_fwd_k, dim3(ceil_div(total, 256)), dim3(256), 27 * Co * sizeof(float), stream, x, img_idx, w, z, N, H, W,
                     g.Ho, g.Wo, Co, g.pt, g.pl, nm);
  MLIIS_CHECK_LAUNCH("stem_conv_fwd");
  return MLIIS_OK;
}

// The same conv with the stage-1 statistics of z ({sum, sum of squares} per workgroup, [*nblk][2][Co]: what mliis_bn_stats_partial(z)
// would produce in a second launch) for the batch norm that follows.  stats_part == NULL: no statistics (z only, *nblk = 0).
// MLIIS_ERR_UNSUPPORTED (nothing launched) when the staging window of a row does not fit LDS (W > ~400): the caller then takes
// mliis_stem_conv_fwd + mliis_bn_stats_partial.
static void stem_rows_geom(int N, int Ho, int* rows_per_block, int* blocks_per_img) {
  const int groups = (Ho + kStemRows - 1) / kStemRows;
  long long g = ((long long)N * groups + 255) / 256;   // staging rounds per workgroup so that the grid stays near 256 workgroups
  if (g < 1) g = 1;
  *rows_per_block = (int)g * kStemRows;
  *blocks_per_img = (Ho + *rows_per_block - 1) / *rows_per_block;
}
static size_t stem_rows_lds(int Wo, int Co) { return ((size_t)(2 * kStemRows + 1) * (2 * Wo + 1) * 3 + kStemRowsThreads * 8) * sizeof(float); }

size_t mliis_stem_conv_fwd_stats_floats(int N, int H, int W, int Co) {
  if (N <= 0 || H <= 1 || W <= 1 || Co <= 0) return 0;
  int rpb, bpi;
  stem_rows_geom(N, (H + 1) / 2, &rpb, &bpi);
  return (size_t)N * bpi * 2 * Co;
}

int mliis_stem_conv_fwd_stats(const float* x, const int* img_idx, const float* w, float* z, int N, int H, int W, int Co, const float* mean3,
                              const float* std3, float* stats_part, size_t stats_floats, int* nblk, hipStream_t stream) {
  MLIIS_REQUIRE(x && w && z && mean3 && std3, MLIIS_ERR_ARG, "stem_conv_fwd_stats: null pointer");
  MLIIS_REQUIRE(N > 0 && H > 1 && W > 1 && Co > 0 && (Co & 7) == 0 && Co <= 256, MLIIS_ERR_ARG, "stem_conv_fwd_stats: bad shape (Co %% 8 == 0 required)");
  MLIIS_REQUIRE(aligned16(w), MLIIS_ERR_ALIGN, "stem_conv_fwd_stats: weights must be 16-byte aligned");
  MLIIS_REQUIRE(aligned16(z), MLIIS_ERR_ALIGN, "stem_conv_fwd_stats: output must be 16-byte aligned");
  StemGeom g = stem_geom(H, W);
  const size_t lds = stem_rows_lds(g.Wo, Co);
  MLIIS_REQUIRE(lds <= 64 * 1024, MLIIS_ERR_UNSUPPORTED, "stem_conv_fwd_stats: the staging window of a %d-pixel row does not fit LDS", W);
  int rpb, bpi;
  stem_rows_geom(N, g.Ho, &rpb, &bpi);
  MLIIS_REQUIRE(stats_part == nullptr || (size_t)N * bpi * 2 * Co <= stats_floats, MLIIS_ERR_WORKSPACE, "stem_conv_fwd_stats: statistics buffer too small");
  Norm3 nm{mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]};
  hipLaunchKernelGGL(stem_fwd_rows_k, dim3((unsigned)(N * bpi)), dim3(kStemRowsThreads), lds, stream, x, img_idx, w, z, H, W, g.Ho, g.Wo, Co, g.pt, g.pl, nm, rpb,
                     bpi, stats_part);
  MLIIS_CHECK_LAUNCH("stem_conv_fwd_stats");
  if (nblk) *nblk = stats_part != nullptr ? N * bpi : 0;
  return MLIIS_OK;
}

size_t mliis_stem_conv_bwd_filter_workspace_floats(int N, int H, int W, int Co) {
  if (N <= 0 || H <= 1 || W <= 1 || Co <= 0 || Co > 256) return 0;
  StemGeom g = stem_geom(H, W);
  int ppb, nblk;
  stem_filter_geom(N, g.Ho, g.Wo, Co, &ppb, &nblk);
  return (size_t)nblk * 27 * Co;
}

int mliis_stem_conv_bwd_filter(const float* x, const int* img_idx, const float* dz, float* dw, int N, int H, int W, int Co,
                               const float* mean3, const float* std3, float* ws, size_t ws_floats, hipStream_t stream) {
  MLIIS_REQUIRE(x && dz && ws && mean3 && std3, MLIIS_ERR_ARG, "stem_conv_bwd_filter: null pointer");   // dw == NULL: slabs stay in ws
  MLIIS_REQUIRE(N > 0 && H > 1 && W > 1 && Co > 0 && (Co & 3) == 0 && Co <= 256, MLIIS_ERR_ARG, "stem_conv_bwd_filter: bad shape");
  MLIIS_REQUIRE(aligned16(dz) && aligned16(ws), MLIIS_ERR_ALIGN, "stem_conv_bwd_filter: dz / workspace must be 16-byte aligned");
  StemGeom g = stem_geom(H, W);
  int ppb, nblk;
  stem_filter_geom(N, g.Ho, g.Wo, Co, &ppb, &nblk);
  MLIIS_REQUIRE((size_t)nblk * 27 * Co <= ws_floats, MLIIS_ERR_WORKSPACE, "stem_conv_bwd_filter: workspace too small");
  Norm3 nm{mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]};
  if (Co <= 64 && ppb <= kStemPix && (long long)N * H * W * 3 < (1LL << 31)) {   // matrix-core kernel (same slab layout)
    if (Co <= 32) hipLaunchKernelGGL(stem_bwd_filter_mfma_k<2>, dim3(nblk), dim3(256), 0, stream, x, img_idx, dz, ws, N, H, W, g.Ho, g.Wo, Co, g.pt, g.pl, nm, ppb);
    else if (Co <= 48) hipLaunchKernelGGL(stem_bwd_filter_mfma_k<3>, dim3(nblk), dim3(256), 0, stream, x, img_idx, dz, ws, N, H, W, g.Ho, g.Wo, Co, g.pt, g.pl, nm, ppb);
    else hipLaunchKernelGGL(stem_bwd_filter_mfma_k<4>, dim3(nblk), dim3(256), 0, stream, x, img_idx, dz, ws, N, H, W, g.Ho, g.Wo, Co, g.pt, g.pl, nm, ppb);
    MLIIS_CHECK_LAUNCH("stem_conv_bwd_filter_mfma");
    if (dw == nullptr) return MLIIS_OK;
    hipLaunchKernelGGL(fold_flat_k, dim3(ceil_div(27 * Co, kFoldX)), dim3(kFoldX, kFoldY), 0, stream, ws, nblk, (long long)27 * Co, 1.0f, dw, 0, (long long)27 * Co, 0LL, 0LL);
    MLIIS_CHECK_LAUNCH("stem_conv_bwd_filter_finalize");
    return MLIIS_OK;
  }
  const size_t lds = (size_t)4 * 27 * (Co / 4) * sizeof(float4) + (size_t)(256 / stem_quad_pad(Co)) * 27 * sizeof(float);
  hipLaunchKernelGGL(stem_bwd_filter_k, dim3(nblk), dim3(256), lds, stream, x, img_idx, dz, ws, N, H, W, g.Ho, g.Wo, Co,
                     g.pt, g.pl, nm, ppb, stem_quad_pad(Co));
  MLIIS_CHECK_LAUNCH("stem_conv_bwd_filter");
  if (dw == nullptr) return MLIIS_OK;
  hipLaunchKernelGGL(fold_flat_k, dim3(ceil_div(27 * Co, kFoldX)), dim3(kFoldX, kFoldY), 0, stream, ws, nblk, (long long)27 * Co, 1.0f, dw, 0, (long long)27 * Co, 0LL, 0LL);
  MLIIS_CHECK_LAUNCH("stem_conv_bwd_filter_finalize");
  return MLIIS_OK;
}
}
