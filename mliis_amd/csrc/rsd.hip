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

// grid (9 border classes); threads = (co, c-lane); all images handled inside the block so every weight is read once:
// E[n][cls][co] = sum_c pool[n][c] * sum_{taps valid in cls} W[tap][c][co]
constexpr int kRsdMaxN = 16;
__global__ __launch_bounds__(kRsdThreads) void rsd_pool_fwd_k(const float* __restrict__ pool, const float* __restrict__ w,
                                                              float* __restrict__ E, int N, int Cp, int Cin_total, int c_begin, int Co) {
  extern __shared__ float dyn[];          // [N][Cp] pooled vectors, then reused as the [CL][Co] reduction buffer per image
  float* sp = dyn;
  float* red = dyn + (size_t)N * Cp;
  const int cls = blockIdx.x;
  // blockIdx.y selects a chunk of the constant channels; the chunks' partial results are folded by a second (tiny) launch
  const int cchunk = (Cp + gridDim.y - 1) / gridDim.y;
  const int cbeg = blockIdx.y * cchunk;
  const int cend = (cbeg + cchunk < Cp) ? cbeg + cchunk : Cp;
  E += (long long)blockIdx.y * N * 9 * Co;
  const int rc = cls / 3, cc = cls - rc * 3;
  const int CL = kRsdThreads / Co;
  const int co = threadIdx.x % Co, cl = threadIdx.x / Co;
  for (int i = threadIdx.x; i < N * Cp; i += kRsdThreads) sp[i] = pool[i];
  __syncthreads();
  float a[kRsdMaxN];
#pragma unroll
  for (int n = 0; n < kRsdMaxN; ++n) a[n] = 0.f;
  if (cl < CL)
    for (int c = cbeg + cl; c < cend; c += CL) {
      float wt[9], ws = 0.f;   // all nine taps fetched together (valid addresses either way), the excluded ones weighted by zero
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
  for (int n = 0; n < kRsdMaxN; ++n) {
    if (n >= N) break;
    __syncthreads();
    red[threadIdx.x] = a[n];
    __syncthreads();
    if (threadIdx.x < Co) {
      float r = 0.f;
      for (int k = 0; k < CL; ++k) r += red[k * Co + threadIdx.x];
      E[((long long)n * 9 + cls) * Co + threadIdx.x] = r;
    }
  }
}

// grid (N, 4): border sums of dz: part 0 = first row, 1 = last row, 2 = first column, 3 = last column.  B[n][part][co]
__global__ __launch_bounds__(kRsdThreads) void rsd_border_sums_k(const float* __restrict__ dz, int ld, float* __restrict__ B, int H, int W,
                                                                 int Co) {
  __shared__ float red[kRsdThreads];
  const int n = blockIdx.x, part = blockIdx.y;
  const int CL = kRsdThreads / Co;
  const int co = threadIdx.x % Co, cl = threadIdx.x / Co;
  const float* d = dz + (long long)n * H * W * ld;
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

// border_bias[n][class][co] for mliis_conv2d_fwd: contribution of the constant input channels [c_begin, c_begin+Cp) whose
// per-image values are pool[n][:].  w: HWIO [3,3,Cin_total,Co].
size_t mliis_rsd_pool_fwd_workspace_floats(int N, int Co) { return (N > 0 && Co > 0) ? (size_t)8 * N * 9 * Co : 0; }

int mliis_rsd_pool_fwd(const float* pool, const float* w, float* border_bias, int N, int Cp, int Cin_total, int c_begin, int Co, float* ws,
                       size_t ws_floats, hipStream_t stream) {
  MLIIS_REQUIRE(pool && w && border_bias, MLIIS_ERR_ARG, "rsd_pool_fwd: null pointer");
  MLIIS_REQUIRE(N > 0 && Cp > 0 && Co > 0 && Co <= 1024 && c_begin >= 0 && c_begin + Cp <= Cin_total &&
                    (size_t)(N < kRsdMaxN ? N : kRsdMaxN) * Cp + kRsdThreads <= 15360,
                MLIIS_ERR_ARG, "rsd_pool_fwd: bad shape");
  const int S = 8;   // channel chunks: 72 workgroups instead of 9, each streaming 1/8 of the weights
  MLIIS_REQUIRE(ws && (size_t)S * N * 9 * Co <= ws_floats, MLIIS_ERR_WORKSPACE, "rsd_pool_fwd: workspace too small");
  // the kernel keeps one accumulator per image in registers: batches beyond kRsdMaxN images go through in groups (the workspace
  // is reused, stream-ordered)
  for (int n0 = 0; n0 < N; n0 += kRsdMaxN) {
    const int nc = N - n0 < kRsdMaxN ? N - n0 : kRsdMaxN;
    hipLaunchKernelGGL(rsd_pool_fwd_k, dim3(9, S), dim3(kRsdThreads), ((size_t)nc * Cp + kRsdThreads) * sizeof(float), stream,
                       pool + (size_t)n0 * Cp, w, ws, nc, Cp, Cin_total, c_begin, Co);
    MLIIS_CHECK_LAUNCH("rsd_pool_fwd");
    const long long total = (long long)nc * 9 * Co;
    hipLaunchKernelGGL(fold_flat_k, dim3(ceil_div(total, kFoldX)), dim3(kFoldX, kFoldY), 0, stream, ws, S, total, 1.0f,
                       border_bias + (size_t)n0 * 9 * Co, 0, total, 0LL, 0LL);
    MLIIS_CHECK_LAUNCH("rsd_pool_fwd_fold");
  }
  return MLIIS_OK;
}

size_t mliis_rsd_pool_bwd_workspace_floats(int N, int Co) { return (N > 0 && Co > 0) ? (size_t)N * 13 * Co : 0; }

// dz: gradient of the fuse conv output [N,H,W,Co] (row stride lddz); tot[n][co] = per-image column sums of dz.  Writes the
// weight-gradient rows of the constant channels into dw (full HWIO gradient tensor), dbias (nullable) = sum_n tot, and
// dpool[n][c] = (dL/dpool[n][c]) / (H*W)  (what has to be added to every pixel of the tensor the pool was taken from).
int mliis_rsd_pool_bwd(const float* dz, int lddz, const float* tot, const float* pool, const float* w, float* dw, float* dbias,
                       float* dpool, int N, int H, int W, int Cp, int Cin_total, int c_begin, int Co, float* ws, size_t ws_floats,
                       hipStream_t stream) {
  MLIIS_REQUIRE(dz && tot && pool && w && dw && dpool && ws, MLIIS_ERR_ARG, "rsd_pool_bwd: null pointer");
  MLIIS_REQUIRE(N > 0 && H >= 2 && W >= 2 && Cp > 0 && Co > 0 && lddz >= Co && c_begin >= 0 && c_begin + Cp <= Cin_total, MLIIS_ERR_ARG,
                "rsd_pool_bwd: bad shape");
  MLIIS_REQUIRE((size_t)N * 13 * Co <= ws_floats && Co <= 1024, MLIIS_ERR_WORKSPACE, "rsd_pool_bwd: workspace too small");
  float* B = ws;     // [N][4][Co] border sums (the per-tap sums G are formed from them where they are used)
  hipLaunchKernelGGL(rsd_border_sums_k, dim3(N, 4), dim3(kRsdThreads), 0, stream, dz, lddz, B, H, W, Co);
  MLIIS_CHECK_LAUNCH("rsd_border_sums");
  const int dw_blocks = ceil_div(9LL * Cp * Co + Co, 256), dp_blocks = ceil_div((long long)N * Cp * 64, 256);
  const RsdG g{dz, lddz, tot, B, H, W, Co};
  hipLaunchKernelGGL(rsd_pool_bwd_k, dim3(dw_blocks + dp_blocks), dim3(256), 0, stream, g, pool, w, dw, dbias, dpool, N, Cp, Cin_total, c_begin,
                     Co, 1.0f / ((float)H * (float)W), dw_blocks);
  MLIIS_CHECK_LAUNCH("rsd_pool_bwd");
  return MLIIS_OK;
}
}
