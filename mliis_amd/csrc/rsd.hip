// Residual-skip-decoder pooled branch without convolving it.
// Reference: models/efficientlab.py:192-197,220-224 -- branch_2 = spatial mean of the concatenated features, tiled to HxW and
// concatenated into the "pyramid" that the 3x3 fuse conv reads.  Those Cp input channels are spatially CONSTANT, so their
// contribution to output pixel (h,w) is  sum_{taps valid at (h,w)} pool[n,:] . W[tap, c_begin:c_begin+Cp, :]  -- a per-image
// vector that depends only on the pixel's border class (3 row classes x 3 column classes; zero padding removes taps at the
// borders).  Forward: E[n][class][co] goes into the GEMM epilogue (conv_gemm.hip, border_bias).  Backward: the gradients of
// the pooled channels and of their weight rows only need per-image sums of dZ over the whole map, its first/last row and
// column and its four corners.  This removes Cp of the Cin_total input channels (136 of 360 at 56x56: 38 %) from the fuse
// conv's forward, backward-data and backward-filter GEMMs; results are algebraically identical.
#include "common.hpp"

namespace mliis {

__device__ __forceinline__ bool tap_valid(int t3, int cls3) {  // t3: 0,1,2 <-> offset -1,0,+1 ; cls3: 0 first, 1 interior, 2 last
  return !((t3 == 0 && cls3 == 0) || (t3 == 2 && cls3 == 2));
}

constexpr int kRsdThreads = 1024;
constexpr int kRsdMaxN = 16;

__device__ __forceinline__ float4 f4sfma(float s, float4 a, float4 c) {
  return make_float4(fmaf(s, a.x, c.x), fmaf(s, a.y, c.y), fmaf(s, a.z, c.z), fmaf(s, a.w, c.w));
}

// The RSD module's input: cat[n,h,w,:] = [deep map (copied, or bilinearly resized to HxW: efficientlab.py:205-206) | skip feature]
// (tf.concat, efficientlab.py:208) written in ONE pass that also leaves the per-image column sums of what it wrote -- the pooled
// branch's spatial mean (efficientlab.py:192-197) -- as `chunks` partials per image, part [N][chunks][Cd + Cs].
// grid (chunks, N); thread = (channel quad, pixel lane) of a 1024-thread workgroup; four pixels in flight per thread.
constexpr int kCatThreads = 1024;
template <bool RESIZE>
__global__ __launch_bounds__(kCatThreads) void rsd_concat_pool_k(const float* __restrict__ deep, int ld_deep, int Hi, int Wi, int Cd,
                                                         const float* __restrict__ skip, int ld_skip, int Cs, float* __restrict__ cat, int ldcat,
                                                         int H, int W, float sh, float sw, int chunks, float* __restrict__ part) {
  __shared__ float4 red[kCatThreads];
  const int C = Cd + Cs, Q = C >> 2;
  const int RL = kCatThreads / Q;
  const int t = threadIdx.x, rl = t / Q, q = t - rl * Q;
  const int n = blockIdx.y, HW = H * W;
  const int ppc = (HW + chunks - 1) / chunks;
  const int p0 = blockIdx.x * ppc, p1 = p0 + ppc < HW ? p0 + ppc : HW;
  const int c = q * 4;
  float4 acc = f4zero();
  if (rl < RL) {
    const bool from_deep = c < Cd;
    const float* dn = deep + (long long)n * Hi * Wi * ld_deep + c;
    const float* sn = skip + (long long)n * HW * ld_skip + (c - Cd);
    float* cn = cat + (long long)n * HW * ldcat + c;
    for (int p = p0 + rl; p < p1; p += 4 * RL) {
      float4 v[4];
      if (from_deep && RESIZE) {
        float4 tl[4], tr[4], bl[4], br[4];
        float ly[4], lx[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int pu = p + u * RL < p1 ? p + u * RL : p;
          const int ho = pu / W, wo = pu - ho * W;
          int y0, y1, x0, x1;
          src_coord(ho, sh, Hi, y0, y1, ly[u]);
          src_coord(wo, sw, Wi, x0, x1, lx[u]);
          tl[u] = ld4(dn + ((long long)y0 * Wi + x0) * ld_deep);
          tr[u] = ld4(dn + ((long long)y0 * Wi + x1) * ld_deep);
          bl[u] = ld4(dn + ((long long)y1 * Wi + x0) * ld_deep);
          br[u] = ld4(dn + ((long long)y1 * Wi + x1) * ld_deep);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {   // (the weights and the order of mliis_resize_bilinear_fwd: bit-identical)
          float4 o = f4zero();
          float wtl, wtr, wbl, wbr;
          bilinear_weights(ly[u], lx[u], wtl, wtr, wbl, wbr);
          o = f4sfma(wtl, tl[u], o);
          o = f4sfma(wtr, tr[u], o);
          o = f4sfma(wbl, bl[u], o);
          o = f4sfma(wbr, br[u], o);
          v[u] = o;
        }
      } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int pu = p + u * RL < p1 ? p + u * RL : p;
          v[u] = from_deep ? ld4(dn + (long long)pu * ld_deep) : ld4(sn + (long long)pu * ld_skip);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (p + u * RL < p1) {
          st4(cn + (long long)(p + u * RL) * ldcat, v[u]);
          acc = f4add(acc, v[u]);
        }
    }
  }
  red[t] = acc;
  __syncthreads();
  if (t < Q) {   // the pixel lanes in lane order (deterministic)
    float4 s4 = red[t];
    for (int k = 1; k < RL; ++k) s4 = f4add(s4, red[k * Q + t]);
    st4(part + ((long long)n * chunks + blockIdx.x) * C + t * 4, s4);
  }
}

// grid (9 border classes, column groups of 16); threads = (16 columns, 64 channel lanes); all images (<= kRsdMaxN per launch) inside
// the block, so a weight is read once per block and nothing is left to fold:
// E[n][cls][co] = sum_c pool[n][c] * sum_{taps valid in cls} W[tap][c_begin + c][co],  pool[n][c] = scale * sum_chunks part[n][k][c]
// (the partial sums of mliis_rsd_concat_pool, or chunks = 1 for a finished vector); block (0, 0) also publishes pool for the backward pass.
__global__ __launch_bounds__(kRsdThreads) void rsd_pool_fwd_k(const float* __restrict__ part, int chunks, float scale, float* __restrict__ pool_out,
                                                              const float* __restrict__ w, float* __restrict__ E, int N, int Cp, int Cin_total,
                                                              int c_begin, int Co) {
  extern __shared__ float dyn[];          // [N][Cp] pooled vectors, [16 waves][N][16] partial results
  float* sp = dyn;
  float* red = dyn + (size_t)N * Cp;
  const int cls = blockIdx.x, n0 = blockIdx.y * 16;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int col = t & 15, cl = t >> 4;
  const int co = n0 + col < Co ? n0 + col : Co - 1;
  const int rc = cls / 3, cc = cls - rc * 3;
  // the class-summed weights of this thread's first four channels (Cp <= 256: all of them) are requested before the pooled sums are
  // folded: the launch is one latency chain, and the weights do not depend on the data
  float wsum[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = cl + 64 * i < Cp ? cl + 64 * i : Cp - 1;
    float wt[9], ws = 0.f;   // all nine taps fetched together (valid addresses either way), the excluded ones weighted by zero
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) wt[tp] = w[((long long)tp * Cin_total + c_begin + c) * Co + co];
#pragma unroll
    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) ws += (tap_valid(ty, rc) && tap_valid(tx, cc)) ? wt[ty * 3 + tx] : 0.f;
    wsum[i] = ws;
  }
  for (int i = t; i < N * Cp; i += kRsdThreads) {
    const int n = i / Cp, c = i - n * Cp;
    const float* pp = part + ((long long)n * chunks) * Cp + c;
    float a = 0.f;
    for (int k = 0; k < chunks; k += 8) {   // (eight chunks per round trip, chunk order: deterministic)
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = pp[(long long)(k + u < chunks ? k + u : chunks - 1) * Cp];
#pragma unroll
      for (int u = 0; u < 8; ++u) a += k + u < chunks ? v[u] : 0.f;
    }
    a *= scale;
    sp[i] = a;
    if (pool_out != nullptr && blockIdx.x == 0 && blockIdx.y == 0) pool_out[i] = a;
  }
  __syncthreads();
  float a[kRsdMaxN];
#pragma unroll
  for (int n = 0; n < kRsdMaxN; ++n) a[n] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = cl + 64 * i;
    if (c < Cp) {
#pragma unroll
      for (int n = 0; n < kRsdMaxN; ++n)
        if (n < N) a[n] = fmaf(sp[n * Cp + c], wsum[i], a[n]);
    }
  }
  for (int c = cl + 256; c < Cp; c += 64) {
    float wt[9], ws = 0.f;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) wt[tp] = w[((long long)tp * Cin_total + c_begin + c) * Co + co];
#pragma unroll
    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) ws += (tap_valid(ty, rc) && tap_valid(tx, cc)) ? wt[ty * 3 + tx] : 0.f;
#pragma unroll
    for (int n = 0; n < kRsdMaxN; ++n)
      if (n < N) a[n] = fmaf(sp[n * Cp + c], ws, a[n]);
  }
#pragma unroll
  for (int n = 0; n < kRsdMaxN; ++n) {   // the four channel lanes of a wave (fixed butterfly), then the 16 waves in order
    if (n >= N) break;
    float v = a[n];
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (lane < 16) red[(wave * N + n) * 16 + lane] = v;
  }
  __syncthreads();
  if (t < N * 16) {
    const int n = t >> 4, c2 = t & 15;
    float r = 0.f;
#pragma unroll
    for (int wv = 0; wv < kRsdThreads / 64; ++wv) r += red[(wv * N + n) * 16 + c2];
    if (n0 + c2 < Co) E[((long long)n * 9 + cls) * Co + n0 + c2] = r;
  }
}

// grid (N, 4 + column groups of 16): parts 0..3 = border sums of dz -- first row, last row, first column, last column -- B[n][part][co];
// parts 4.. = the whole-map sums of 16 columns, tot[n][co] (64 pixel lanes, sixteen pixels in flight per thread).
__global__ __launch_bounds__(kRsdThreads) void rsd_border_sums_k(const float* __restrict__ dz, int ld, float* __restrict__ B, float* __restrict__ tot,
                                                                 int H, int W, int Co) {
  __shared__ float red[kRsdThreads];
  const int n = blockIdx.x, part = blockIdx.y;
  const float* d = dz + (long long)n * H * W * ld;
  if (part >= 4) {
    const int col = threadIdx.x & 15, rl = threadIdx.x >> 4, HW = H * W;
    const int c0 = (part - 4) * 16 + col;
    const int co = c0 < Co ? c0 : Co - 1;
    float a = 0.f;
    for (int i = rl; i < HW; i += 16 * 64) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = d[(long long)(i + u * 64 < HW ? i + u * 64 : rl) * ld + co];
#pragma unroll
      for (int u = 0; u < 16; ++u) a += i + u * 64 < HW ? v[u] : 0.f;
    }
    red[threadIdx.x] = a;
    __syncthreads();
    if (threadIdx.x < 16 && c0 < Co) {
      float r = 0.f;
      for (int k = 0; k < 64; ++k) r += red[k * 16 + threadIdx.x];
      tot[(long long)n * Co + c0] = r;
    }
    return;
  }
  const int CL = kRsdThreads / Co;
  const int co = threadIdx.x % Co, cl = threadIdx.x / Co;
  const int len = part < 2 ? W : H;
  float a = 0.f;
  if (cl < CL) {
    const long long p0 = part == 0 ? 0 : part == 1 ? (long long)(H - 1) * W : part == 2 ? 0 : W - 1;
    const long long ps = part < 2 ? 1 : W;   // pixel index = p0 + i * ps
    for (int i = cl; i < len; i += 8 * CL) {   // eight loads per trip (clamped, zero-weighted past the end)
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int k = i + u * CL < len ? i + u * CL : cl;
        v[u] = d[(p0 + k * ps) * ld + co];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) a += i + u * CL < len ? v[u] : 0.f;
    }
  }
  red[threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.x < Co) {
    float r = 0.f;
    for (int k = 0; k < CL; ++k) r += red[k * Co + threadIdx.x];
    B[((long long)n * 4 + part) * Co + threadIdx.x] = r;
  }
}

// G[n][tap][co] = Tot - excluded row - excluded column + corner: the sum of dZ over the pixels where tap (ty, tx) of the 3x3 window
// falls inside the image.  Evaluated where it is used (four loads), not stored: one launch and one round trip fewer.
struct RsdG {
  const float* dz;   // [N,H,W,ld]
  int ld;
  const float* tot;  // [N][Co]
  const float* B;    // [N][4][Co]: first row, last row, first column, last column
  int H, W, Co;
  __device__ __forceinline__ float at(int n, int tap, int co) const {
    const int ty = tap / 3, tx = tap - ty * 3;
    const float* Bn = B + (long long)n * 4 * Co;
    // (all four loads issued; the excluded ones weighted by zero -- no dependent branches around memory)
    const float r0 = Bn[(ty == 2 ? 1 : 0) * Co + co], c0 = Bn[(tx == 2 ? 3 : 2) * Co + co];
    const long long py = ty == 0 ? 0 : H - 1, px = tx == 0 ? 0 : W - 1;
    const float cr = dz[(((long long)n * H + py) * W + px) * ld + co];
    const float t = tot[(long long)n * Co + co];
    return t - (ty != 1 ? r0 : 0.f) - (tx != 1 ? c0 : 0.f) + ((ty != 1 && tx != 1) ? cr : 0.f);
  }
};

// ONE launch for the two independent consumers of G: workgroups [0, dw_blocks) = the weight-gradient rows of the constant channels
// (one thread per element) and the bias gradient; the rest = dpool[n][c] = inv_hw * sum_tap sum_co G[n][tap][co] * W[tap][c_begin+c][co]
// (one wave per (n, c)).
__global__ __launch_bounds__(256) void rsd_pool_bwd_k(RsdG g, const float* __restrict__ pool, const float* __restrict__ w,
                                                      float* __restrict__ dw, float* __restrict__ dbias, float* __restrict__ dpool, int N,
                                                      int Cp, int Cin_total, int c_begin, int Co, float inv_hw, int dw_blocks) {
  if ((int)blockIdx.x < dw_blocks) {
    const long long nW = 9LL * Cp * Co;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i < nW) {
      const int co = (int)(i % Co);
      const long long r = i / Co;
      const int c = (int)(r % Cp), tap = (int)(r / Cp);
      float a = 0.f;
      for (int n0 = 0; n0 < N; n0 += 8) {   // all loads of eight images first
        float pv[8], gv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int n = n0 + u < N ? n0 + u : N - 1;
          pv[u] = pool[(long long)n * Cp + c];
          gv[u] = g.at(n, tap, co);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) a = fmaf(n0 + u < N ? pv[u] : 0.f, gv[u], a);
      }
      dw[((long long)tap * Cin_total + c_begin + c) * Co + co] = a;
    } else if (dbias != nullptr && i < nW + Co) {
      const int co = (int)(i - nW);
      float a = 0.f;
      for (int n = 0; n < N; ++n) a += g.tot[(long long)n * Co + co];
      dbias[co] = a;
    }
    return;
  }
  const int wv = (((int)blockIdx.x - dw_blocks) * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (wv >= N * Cp) return;
  const int c = wv % Cp, n = wv / Cp;
  float a = 0.f;
  for (int co = lane; co < Co; co += 64) {   // the nine taps' loads of a column first (one round trip per 64 columns)
    float gv[9], wt[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      wt[tap] = w[((long long)tap * Cin_total + c_begin + c) * Co + co];
      gv[tap] = g.at(n, tap, co);
    }
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) a = fmaf(gv[tap], wt[tap], a);
  }
  a = wave_sum(a);
  if (lane == 0) dpool[wv] = a * inv_hw;
}

}  // namespace mliis

using namespace mliis;

extern "C" {

// cat [N,H,W,Cd+Cs] = [deep [N,Hi,Wi,Cd] copied (Hi x Wi == H x W) or bilinearly resized | skip [N,H,W,Cs]] and the per-image column
// sums of cat in *chunks partials per image, pool_part [N][*chunks][Cd+Cs] (feed both to mliis_rsd_pool_fwd with scale = 1 / (H W)).
static inline int rsd_concat_chunks(int H, int W) {
  const int c = (H * W + 48) / 49;
  return c < 1 ? 1 : (c > 16 ? 16 : c);
}
size_t mliis_rsd_concat_pool_floats(int N, int H, int W, int C) {
  return (N > 0 && H > 0 && W > 0 && C > 0) ? (size_t)N * rsd_concat_chunks(H, W) * C : 0;
}
int mliis_rsd_concat_pool(const float* deep, int ld_deep, int Hi, int Wi, int Cd, const float* skip, int ld_skip, int Cs, float* cat, int ldcat,
                          int N, int H, int W, float* pool_part, size_t pool_part_floats, int* chunks, hipStream_t stream) {
  MLIIS_REQUIRE(deep && skip && cat && pool_part && chunks, MLIIS_ERR_ARG, "rsd_concat_pool: null pointer");
  MLIIS_REQUIRE(N > 0 && N < 65536 && H > 1 && W > 1 && Hi > 0 && Wi > 0 && Cd > 0 && Cs > 0 && (Cd & 3) == 0 && (Cs & 3) == 0 && Cd + Cs <= 4 * kCatThreads &&
                    ld_deep >= Cd && ld_skip >= Cs && ldcat >= Cd + Cs && ((ld_deep | ld_skip | ldcat) & 3) == 0,
                MLIIS_ERR_ARG, "rsd_concat_pool: bad shape (channel counts and leading dimensions multiples of 4, Cd + Cs <= 4096)");
  MLIIS_REQUIRE(aligned16(deep) && aligned16(skip) && aligned16(cat) && aligned16(pool_part), MLIIS_ERR_ALIGN,
                "rsd_concat_pool: pointers must be 16-byte aligned");
  const int ch = rsd_concat_chunks(H, W);
  MLIIS_REQUIRE((size_t)N * ch * (Cd + Cs) <= pool_part_floats, MLIIS_ERR_WORKSPACE, "rsd_concat_pool: pool_part too small");
  const float sh = (float)(Hi - 1) / (float)(H - 1), sw = (float)(Wi - 1) / (float)(W - 1);
  if (Hi == H && Wi == W)
    hipLaunchKernelGGL(rsd_concat_pool_k<false>, dim3(ch, N), dim3(kCatThreads), 0, stream, deep, ld_deep, Hi, Wi, Cd, skip, ld_skip, Cs, cat, ldcat, H, W, sh,
                       sw, ch, pool_part);
  else
    hipLaunchKernelGGL(rsd_concat_pool_k<true>, dim3(ch, N), dim3(kCatThreads), 0, stream, deep, ld_deep, Hi, Wi, Cd, skip, ld_skip, Cs, cat, ldcat, H, W, sh,
                       sw, ch, pool_part);
  MLIIS_CHECK_LAUNCH("rsd_concat_pool");
  *chunks = ch;
  return MLIIS_OK;
}

// border_bias[n][class][co] for mliis_conv2d_fwd: contribution of the constant input channels [c_begin, c_begin+Cp) whose
// per-image values are pool[n][:] = scale * sum_k pool_part[n][k][:] (chunks = 1, scale = 1: a finished vector).  w: HWIO
// [3,3,Cin_total,Co].  pool_out (nullable) [N][Cp]: the folded vectors, kept for mliis_rsd_pool_bwd.
int mliis_rsd_pool_fwd(const float* pool_part, int chunks, float scale, float* pool_out, const float* w, float* border_bias, int N, int Cp,
                       int Cin_total, int c_begin, int Co, hipStream_t stream) {
  MLIIS_REQUIRE(pool_part && w && border_bias, MLIIS_ERR_ARG, "rsd_pool_fwd: null pointer");
  MLIIS_REQUIRE(N > 0 && chunks > 0 && Cp > 0 && Co > 0 && c_begin >= 0 && c_begin + Cp <= Cin_total &&
                    (size_t)(N < kRsdMaxN ? N : kRsdMaxN) * (Cp + 256) <= 15360,
                MLIIS_ERR_ARG, "rsd_pool_fwd: bad shape");
  // the kernel keeps one accumulator per image in registers: batches beyond kRsdMaxN images go through in groups
  for (int n0 = 0; n0 < N; n0 += kRsdMaxN) {
    const int nc = N - n0 < kRsdMaxN ? N - n0 : kRsdMaxN;
    hipLaunchKernelGGL(rsd_pool_fwd_k, dim3(9, ceil_div(Co, 16)), dim3(kRsdThreads), ((size_t)nc * Cp + (size_t)16 * nc * 16) * sizeof(float), stream,
                       pool_part + (size_t)n0 * chunks * Cp, chunks, scale, pool_out ? pool_out + (size_t)n0 * Cp : nullptr, w,
                       border_bias + (size_t)n0 * 9 * Co, nc, Cp, Cin_total, c_begin, Co);
    MLIIS_CHECK_LAUNCH("rsd_pool_fwd");
  }
  return MLIIS_OK;
}

size_t mliis_rsd_pool_bwd_workspace_floats(int N, int Co) { return (N > 0 && Co > 0) ? (size_t)N * 13 * Co : 0; }

// dz: gradient of the fuse conv output [N,H,W,Co] (row stride lddz); tot [N][Co] (OUTPUT, kept by the caller): per-image column sums of
// dz, formed by the same launch as the border sums.  Writes the
// weight-gradient rows of the constant channels into dw (full HWIO gradient tensor), dbias (nullable) = sum_n tot, and
// dpool[n][c] = (dL/dpool[n][c]) / (H*W)  (what has to be added to every pixel of the tensor the pool was taken from).
int mliis_rsd_pool_bwd(const float* dz, int lddz, float* tot, const float* pool, const float* w, float* dw, float* dbias,
                       float* dpool, int N, int H, int W, int Cp, int Cin_total, int c_begin, int Co, float* ws, size_t ws_floats,
                       hipStream_t stream) {
  MLIIS_REQUIRE(dz && tot && pool && w && dw && dpool && ws, MLIIS_ERR_ARG, "rsd_pool_bwd: null pointer");
  MLIIS_REQUIRE(N > 0 && H >= 2 && W >= 2 && Cp > 0 && Co > 0 && lddz >= Co && c_begin >= 0 && c_begin + Cp <= Cin_total, MLIIS_ERR_ARG,
                "rsd_pool_bwd: bad shape");
  MLIIS_REQUIRE((size_t)N * 13 * Co <= ws_floats && Co <= 1024, MLIIS_ERR_WORKSPACE, "rsd_pool_bwd: workspace too small");
  float* B = ws;     // [N][4][Co] border sums (the per-tap sums G are formed from them where they are used)
  hipLaunchKernelGGL(rsd_border_sums_k, dim3(N, 4 + ceil_div(Co, 16)), dim3(kRsdThreads), 0, stream, dz, lddz, B, tot, H, W, Co);
  MLIIS_CHECK_LAUNCH("rsd_border_sums");
  const int dw_blocks = ceil_div(9LL * Cp * Co + Co, 256), dp_blocks = ceil_div((long long)N * Cp * 64, 256);
  const RsdG g{dz, lddz, tot, B, H, W, Co};
  hipLaunchKernelGGL(rsd_pool_bwd_k, dim3(dw_blocks + dp_blocks), dim3(256), 0, stream, g, pool, w, dw, dbias, dpool, N, Cp, Cin_total, c_begin,
                     Co, 1.0f / ((float)H * (float)W), dw_blocks);
  MLIIS_CHECK_LAUNCH("rsd_pool_bwd");
  return MLIIS_OK;
}
}
