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

// one workgroup per image
__global__ __launch_bounds__(256) void rsd_pool_fwd_k(const float* __restrict__ pool, const float* __restrict__ w,
                                                      float* __restrict__ E, int Cp, int Cin_total, int c_begin, int Co) {
  extern __shared__ float T[];  // [9][Co]
  const int n = blockIdx.x;
  const float* pn = pool + (long long)n * Cp;
  for (int idx = threadIdx.x; idx < 9 * Co; idx += 256) {
    const int tap = idx / Co, co = idx - tap * Co;
    const float* wp = w + ((long long)tap * Cin_total + c_begin) * Co + co;
    float a = 0.f;
    for (int c = 0; c < Cp; ++c) a = fmaf(pn[c], wp[(long long)c * Co], a);
    T[idx] = a;
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < 9 * Co; idx += 256) {
    const int cls = idx / Co, co = idx - cls * Co;
    const int rc = cls / 3, cc = cls - rc * 3;
    float a = 0.f;
#pragma unroll
    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
      for (int tx = 0; tx < 3; ++tx)
        if (tap_valid(ty, rc) && tap_valid(tx, cc)) a += T[(ty * 3 + tx) * Co + co];
    E[((long long)n * 9 + cls) * Co + co] = a;
  }
}

// G[n][tap][co] = sum of dz over the pixels at which `tap` reads inside the map = Tot - excluded row - excluded column + corner
__global__ __launch_bounds__(256) void rsd_pool_gsum_k(const float* __restrict__ dz, int ld, const float* __restrict__ tot,
                                                       float* __restrict__ G, int H, int W, int Co) {
  const int n = blockIdx.x;
  const float* d = dz + (long long)n * H * W * ld;
  for (int co = threadIdx.x; co < Co; co += 256) {
    float r0 = 0.f, rh = 0.f, c0 = 0.f, cw = 0.f;
    for (int x = 0; x < W; ++x) {
      r0 += d[(long long)x * ld + co];
      rh += d[((long long)(H - 1) * W + x) * ld + co];
    }
    for (int y = 0; y < H; ++y) {
      c0 += d[((long long)y * W) * ld + co];
      cw += d[((long long)y * W + W - 1) * ld + co];
    }
    const float k00 = d[co], k0w = d[(long long)(W - 1) * ld + co], kh0 = d[((long long)(H - 1) * W) * ld + co],
                khw = d[((long long)(H - 1) * W + W - 1) * ld + co];
    const float t = tot[(long long)n * Co + co];
#pragma unroll
    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        // tap offset -1 reads above/left: invalid on the first row/col; offset +1 invalid on the last row/col
        const float rex = ty == 0 ? r0 : (ty == 2 ? rh : 0.f);
        const float cex = tx == 0 ? c0 : (tx == 2 ? cw : 0.f);
        float cor = 0.f;
        if (ty == 0 && tx == 0) cor = k00;
        if (ty == 0 && tx == 2) cor = k0w;
        if (ty == 2 && tx == 0) cor = kh0;
        if (ty == 2 && tx == 2) cor = khw;
        G[((long long)n * 9 + ty * 3 + tx) * Co + co] = t - rex - cex + cor;
      }
  }
}

// element-parallel: [0, 9*Cp*Co) weight-gradient rows, then N*Cp pooled-input gradients, then Co bias gradients
__global__ __launch_bounds__(256) void rsd_pool_bwd_k(const float* __restrict__ G, const float* __restrict__ tot,
                                                      const float* __restrict__ pool, const float* __restrict__ w,
                                                      float* __restrict__ dw, float* __restrict__ dbias, float* __restrict__ dpool,
                                                      int N, int Cp, int Cin_total, int c_begin, int Co, float inv_hw) {
  const long long nW = 9LL * Cp * Co;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < nW) {
    const int co = (int)(i % Co);
    const long long r = i / Co;
    const int c = (int)(r % Cp), tap = (int)(r / Cp);
    float a = 0.f;
    for (int n = 0; n < N; ++n) a = fmaf(pool[(long long)n * Cp + c], G[((long long)n * 9 + tap) * Co + co], a);
    dw[((long long)tap * Cin_total + c_begin + c) * Co + co] = a;
  } else if (i < nW + (long long)N * Cp) {
    const long long k = i - nW;
    const int c = (int)(k % Cp), n = (int)(k / Cp);
    float a = 0.f;
    for (int tap = 0; tap < 9; ++tap) {
      const float* wp = w + ((long long)tap * Cin_total + c_begin + c) * Co;
      const float* gp = G + ((long long)n * 9 + tap) * Co;
      for (int co = 0; co < Co; ++co) a = fmaf(gp[co], wp[co], a);
    }
    dpool[k] = a * inv_hw;
  } else if (dbias != nullptr && i < nW + (long long)N * Cp + Co) {
    const int co = (int)(i - nW - (long long)N * Cp);
    float a = 0.f;
    for (int n = 0; n < N; ++n) a += tot[(long long)n * Co + co];
    dbias[co] = a;
  }
}

}  // namespace mliis

using namespace mliis;

extern "C" {

// border_bias[n][class][co] for mliis_conv2d_fwd: contribution of the constant input channels [c_begin, c_begin+Cp) whose
// per-image values are pool[n][:].  w: HWIO [3,3,Cin_total,Co].
int mliis_rsd_pool_fwd(const float* pool, const float* w, float* border_bias, int N, int Cp, int Cin_total, int c_begin, int Co,
                       hipStream_t stream) {
  MLIIS_REQUIRE(pool && w && border_bias, MLIIS_ERR_ARG, "rsd_pool_fwd: null pointer");
  MLIIS_REQUIRE(N > 0 && Cp > 0 && Co > 0 && Co <= 1024 && c_begin >= 0 && c_begin + Cp <= Cin_total, MLIIS_ERR_ARG, "rsd_pool_fwd: bad shape");
  hipLaunchKernelGGL(rsd_pool_fwd_k, dim3(N), dim3(256), 9 * Co * sizeof(float), stream, pool, w, border_bias, Cp, Cin_total, c_begin, Co);
  MLIIS_CHECK_LAUNCH("rsd_pool_fwd");
  return MLIIS_OK;
}

size_t mliis_rsd_pool_bwd_workspace_floats(int N, int Co) { return (N > 0 && Co > 0) ? (size_t)N * 9 * Co : 0; }

// dz: gradient of the fuse conv output [N,H,W,Co] (row stride lddz); tot[n][co] = per-image column sums of dz.  Writes the
// weight-gradient rows of the constant channels into dw (full HWIO gradient tensor), dbias (nullable) = sum_n tot, and
// dpool[n][c] = (dL/dpool[n][c]) / (H*W)  (what has to be added to every pixel of the tensor the pool was taken from).
int mliis_rsd_pool_bwd(const float* dz, int lddz, const float* tot, const float* pool, const float* w, float* dw, float* dbias,
                       float* dpool, int N, int H, int W, int Cp, int Cin_total, int c_begin, int Co, float* ws, size_t ws_floats,
                       hipStream_t stream) {
  MLIIS_REQUIRE(dz && tot && pool && w && dw && dpool && ws, MLIIS_ERR_ARG, "rsd_pool_bwd: null pointer");
  MLIIS_REQUIRE(N > 0 && H >= 2 && W >= 2 && Cp > 0 && Co > 0 && lddz >= Co && c_begin >= 0 && c_begin + Cp <= Cin_total, MLIIS_ERR_ARG,
                "rsd_pool_bwd: bad shape");
  MLIIS_REQUIRE((size_t)N * 9 * Co <= ws_floats, MLIIS_ERR_WORKSPACE, "rsd_pool_bwd: workspace too small");
  hipLaunchKernelGGL(rsd_pool_gsum_k, dim3(N), dim3(256), 0, stream, dz, lddz, tot, ws, H, W, Co);
  MLIIS_CHECK_LAUNCH("rsd_pool_gsum");
  const long long total = 9LL * Cp * Co + (long long)N * Cp + Co;
  hipLaunchKernelGGL(rsd_pool_bwd_k, dim3(ceil_div(total, 256)), dim3(256), 0, stream, ws, tot, pool, w, dw, dbias, dpool, N, Cp, Cin_total,
                     c_begin, Co, 1.0f / ((float)H * (float)W));
  MLIIS_CHECK_LAUNCH("rsd_pool_bwd");
  return MLIIS_OK;
}
}
