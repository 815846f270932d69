// Kernel templates of the dense-conv implicit GEMMs (forward / backward-data: conv_gemm_nk_k, backward-filter: conv_filter_grad2_k)
// and their instantiation switches.  Included by conv_gemm.hip (fp32 operands: v_mfma_f32_16x16x4_f32) and conv_gemm_bf16.hip (bf16
// operands, fp32 accumulation: v_mfma_f32_16x16x32_bf16) so the two instantiation sets compile in parallel.  See conv_gemm.hip for the
// tiling and the host side.
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "bn_fold.hpp"
#include "common.hpp"

namespace mliis {

typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvGemmParams {
  const float* A;
  int lda;
  int Nimg, H, W;
  int C;
  int ntaps, dil, sign;
  const float* B;
  long long b_tap_stride;
  int ldb;
  int Nout;
  float* Cmat;
  int ldc;
  const float* bias;
  int accumulate;
  float* partial;        // non-null => split-K partial output [z][M][Nout]
  int chunks_per_split;  // K chunks per blockIdx.z
  float* stats_part;     // non-null => epilogue also writes per-row-block column sums [gridDim.x][2][Nout] of the
  int stats_swish;       //             stored values (of swish(value) when stats_swish) for the following batch norm
  const float* a_scale;  // non-null => A[m][c] is multiplied by a_scale[image(m)][c] while it is staged (squeeze-excite gate
                         //             applied on the fly: the gated activation tensor is never materialised)
  const float* border_bias;  // non-null => [Nimg][9][Nout] added per output pixel by its border class 3*rowclass + colclass
                             //             (0 = first, 1 = interior, 2 = last): the contribution of spatially CONSTANT input
                             //             channels of a 3x3 SAME conv (the RSD pooled branch) without convolving them
  // fp8 (OCP e4m3) operands, PREC == 2 only: A is multiplied by a_qscale and B by 2^floor(log2(224 / *b_amax)) before the conversion
  // (saturating at +-448); the accumulators are divided by the product of the two scales before the epilogue.
  float a_qscale;
  const float* b_amax;       // device scalar: max |B| of the weight tensor (mliis_transpose_weights)
  // conv1x1_ksplit_k only, with stats_part: the output is the gradient w.r.t. the output of a plain batch norm y = gamma * xhat + beta
  // (an MBConv project BN: no activation) whose input was bnb_x -- the statistics emitted are then stage 1 of THAT batch norm's
  // backward, {sum g, sum g * xhat} with g = value * bnb_scale[image] (drop-connect) and xhat = (bnb_x - bnb_mean) * bnb_rstd,
  // instead of {sum v, sum v^2}: mliis_bn_bwd skips its reduce pass (mliis_conv2d_bwd_data_bn)
  const float* bnb_x = nullptr;
  int bnb_ldx = 0;
  const float* bnb_mean = nullptr;
  const float* bnb_rstd = nullptr;
  const float* bnb_scale = nullptr;   // nullable [Nimg]
  // conv1x1_stream_k only: the output is the gradient w.r.t. gp_x * gate (the squeeze-excite product, efficientnet_model.py:251) --
  // the launch also leaves, per 16-row group, the column sums of output * gp_x split by image (a group spans at most two images
  // when H * W >= 16) in gp_part [row groups][2][Nout]: the gate's gradient without a pass over the two tensors
  // (mliis_conv2d_bwd_data_gate; mliis_se_mlp_bwd folds the groups of an image)
  const float* gp_x = nullptr;
  int gp_ldx = 0;
  float* gp_part = nullptr;
  // bf16 STORAGE of the expanded MBConv tensors (`--precision bf16-storage`; honoured by the PREC == 1 instances only, as uniform
  // run-time branches -- the fp32 instances are untouched): A / Cmat / gp_x address 2-byte elements (leading dimensions in elements);
  // loads widen exactly, the output is rounded to nearest even BEFORE the fused statistics are formed (they see what consumers read).
  int a_bf16 = 0, out_bf16 = 0, side_bf16 = 0;
  // conv1x1_stream_k only: c_block = v (2 | 4): Cmat is written in the group-blocked layout [Nout / v][M][v] (fp32) that the small-map
  // fused MBConv kernels read contiguously (mbconv_small.hip: a workgroup owns v channels over all pixels); ldc is not used then
  int c_block = 0;
  // conv1x1_stream_k<..., AIN = true> only (mliis_conv2d_fwd_bnin): A is the INPUT z of a plain batch norm (an MBConv project BN,
  // efficientnet_model.py:283-288: no activation) whose output -- x drop-connect scale per image, + the block's residual -- is this
  // conv's operand.  The launch folds that batch norm's stage-1 partials (ain.part [nblk][2][C], left by the conv that produced z) in
  // its prologue, forms  a = ((z - mean) * (rstd * gamma) + beta) * ain_scale[image] + ain_res  while the rows are loaded, and the
  // workgroups of column tile 0 write the finished tensor to ain_out (the block output: residual of the next block, endpoint, X
  // operand of the filter gradient); workgroup (0, 0) publishes mean / rstd and advances the moving averages.  The stand-alone
  // apply launch between the two convs (mliis_bn_apply_fused) is gone.
  BnFold ain = {};
  const float* ain_gamma = nullptr;
  const float* ain_beta = nullptr;
  const float* ain_res = nullptr;
  int ain_ldr = 0;
  const float* ain_scale = nullptr;   // nullable [Nimg]
  float* ain_out = nullptr;
  int ain_ldo = 0;
#ifdef KS_DBG
  unsigned long long* dbg_stamps = nullptr;   // [workgroups][8] wall-clock stamps (100 MHz) of thread 0
#endif
};

__device__ __forceinline__ float4 buf_ld4_bf16(__amdgpu_buffer_rsrc_t r, unsigned byte_off);

// operand precision of an instance: 0 = fp32 (v_mfma_f32_16x16x4_f32), 1 = bf16 (v_mfma_f32_16x16x32_bf16), 2 = fp8 e4m3
// (v_mfma_f32_16x16x32_fp8_fp8); operands are converted in registers from the fp32 tensors, accumulation is fp32
constexpr float kFp8Max = 448.0f;
__device__ __forceinline__ float fp8_weight_scale(const float* amax_ptr) {
  const float amax = amax_ptr != nullptr ? *amax_ptr : 0.0f;
  return (amax > 0.0f && amax < 3.0e38f) ? exp2f(floorf(log2f(224.0f / amax))) : 1.0f;
}
__device__ __forceinline__ long pack_fp8x8(float4 a, float4 b, float s) {
  auto q = [s](float v) { return fminf(fmaxf(v * s, -kFp8Max), kFp8Max); };
  int lo = __builtin_amdgcn_cvt_pk_fp8_f32(q(a.x), q(a.y), 0, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(q(a.z), q(a.w), lo, true);
  int hi = __builtin_amdgcn_cvt_pk_fp8_f32(q(b.x), q(b.y), 0, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(q(b.z), q(b.w), hi, true);
  return (long)(((unsigned long)(unsigned)hi << 32) | (unsigned long)(unsigned)lo);
}

// ------------------------------------------------------------------------------------------------ forward / backward-data
// Built so that the matrix pipe is not starved:
//  * global loads are raw buffer loads whose offset is forced out of range for padded / out-of-image / out-of-K elements (the
//    hardware returns zeros) -- no branches, so the address arithmetic of chunk k+1 is scheduled between the MFMAs of chunk k;
//  * every thread advances the (tap, channel) position of its own k quad incrementally, per-row offsets are computed once;
//  * LDS is double buffered: one barrier per K chunk, the LDS writes of chunk k+1 overlap the MFMAs of chunk k of the other waves;
//  * all fragments of a chunk are read from LDS before its first MFMA.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8 pack_bf16x8(float4 a, float4 b) {
  bf16x8 r;
  r[0] = (__bf16)a.x; r[1] = (__bf16)a.y; r[2] = (__bf16)a.z; r[3] = (__bf16)a.w;
  r[4] = (__bf16)b.x; r[5] = (__bf16)b.y; r[6] = (__bf16)b.z; r[7] = (__bf16)b.w;
  return r;
}
// x = hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid) (round to nearest even, v_cvt_pk_bf16_f32): the
// remainders are exact in fp32 and three 8-bit significands cover the 24 of an fp32 value, so the sum of the three terms IS x (finite
// x above the bf16 underflow range; an infinity gives NaN terms).  Four values -> 4 bf16 of each term, k order kept.  (conv_x3.hip)
__device__ __forceinline__ unsigned pk_bf16x2(float a, float b) {
  typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
  bf16x2_ r;
  r[0] = (__bf16)a;
  r[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ void split3(float4 v, uint2& h, uint2& m, uint2& l) {
  h.x = pk_bf16x2(v.x, v.y);
  h.y = pk_bf16x2(v.z, v.w);
  const float rx = v.x - __uint_as_float(h.x << 16), ry = v.y - __uint_as_float(h.x & 0xffff0000u);
  const float rz = v.z - __uint_as_float(h.y << 16), rw = v.w - __uint_as_float(h.y & 0xffff0000u);
  m.x = pk_bf16x2(rx, ry);
  m.y = pk_bf16x2(rz, rw);
  l.x = pk_bf16x2(rx - __uint_as_float(m.x << 16), ry - __uint_as_float(m.x & 0xffff0000u));
  l.y = pk_bf16x2(rz - __uint_as_float(m.y << 16), rw - __uint_as_float(m.y & 0xffff0000u));
}
// Workgroup barrier that orders LDS traffic only: __syncthreads() carries a fence that also drains every outstanding global load of the
// wave (s_waitcnt vmcnt(0)), pinning register-prefetched chunks to the chunk they were requested in; gfx950 backs a barrier off under
// pending memory operations, so waiting for the wave's own LDS operations is enough for an LDS hand-over.  (conv_x3.hip; in the fp32
// kernels below the loads of a chunk are requested at the top of the chunk before, and the change measured nothing: 3340 images/s both.)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
constexpr unsigned kOob = 0xFFFFFFF0u;          // beyond any num_records: the load returns 0
constexpr unsigned kBufRecords = 0x80000000u;   // host guarantees every legal byte offset is below 2 GiB

__device__ __forceinline__ float4 buf_ld4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

__device__ __forceinline__ float4 buf_ld4_bf16(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {   // 4 bf16 (8 bytes) -> float4
  typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
  const u32x2_ v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_off, 0, 0);
  return unpack_bf16x4(make_uint2(v.x, v.y));
}

template <int TM, int NT>
struct GemmSm {
  static constexpr int BM = 64 * TM, BN = 16 * NT;
  static constexpr int A_FLOATS = 8 * BM * 4, B_FLOATS = 8 * BN * 4, BUF_FLOATS = A_FLOATS + B_FLOATS;
  static constexpr int LDS_STAGE = BN + 4;             // epilogue staging row stride (floats)
  static constexpr int STAGE_FLOATS = 4 * 16 * LDS_STAGE;
  static constexpr int SM_FLOATS = (2 * BUF_FLOATS) > STAGE_FLOATS ? (2 * BUF_FLOATS) : STAGE_FLOATS;
};

// One output tile (row tile bx, column tile by) over the K chunks [it0, it1) out of the workgroup's LDS buffer `sm`.
// pdst == nullptr: finished tile -- bias / border bias / accumulate / fused BN statistics, staged row stores into p.Cmat.
// pdst != nullptr: raw partial tile, element (row r, column c) of the tile to pdst[r * pstride + c] (split-K slabs [z][M][Nout]:
// pstride = Nout; stream-K segment slabs [64][BN]: pstride = BN); a fold kernel finishes those tiles.
template <int TM, int NT, int PF, bool SC, bool NARROW, int PREC>
__device__ __forceinline__ void conv_gemm_tile(const ConvGemmParams& p, float* __restrict__ sm, unsigned bx, int by, int it0, int it1,
                                               float* __restrict__ pdst, int pstride) {
  constexpr int BM = 64 * TM, BN = 16 * NT, BK = 32;
  constexpr int A_FLOATS = GemmSm<TM, NT>::A_FLOATS;
  constexpr int BUF_FLOATS = GemmSm<TM, NT>::BUF_FLOATS;
  constexpr int A_PER_THREAD = 2 * TM;          // float4 per thread per chunk
  constexpr int B_TOTAL = BN * 8;               // float4 per chunk
  constexpr int B_PER_THREAD = (B_TOTAL + 255) / 256;
  constexpr int LDS_STAGE = GemmSm<TM, NT>::LDS_STAGE;
  constexpr int kDummy = GemmSm<TM, NT>::SM_FLOATS;   // 16-byte scratch slot behind the buffers

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const long long M = (long long)p.Nimg * p.H * p.W;
  const int n0 = by * BN;
  // K is the flattened (tap, channel) index cut into chunks of 32: chunks may straddle taps (every thread tracks the tap / channel of
  // ITS k quad), so only the very last chunk carries padding and every chunk is a full 2 x 4 x NT MFMA block -- no per-chunk branch.

  const bool split = pdst != nullptr;   // raw partial tile, a fold kernel finishes it
  const bool stats = (p.stats_part != nullptr) && !split;
  float s1[NT], s2[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) s1[j] = s2[j] = 0.f;

  const long long m0 = (long long)bx * BM;
  const bool abf = PREC == 1 && p.a_bf16 != 0, obf = PREC == 1 && p.out_bf16 != 0;   // (uniform; fp32 / fp8 instances: compile-time false)
  const int ab = abf ? 2 : 4;   // bytes per element of A

  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, kBufRecords, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, kBufRecords, 0x00020000);
  const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void*)(SC ? p.a_scale : p.A), 0, kBufRecords, 0x00020000);

  // ---- per-thread rows: byte offsets computed once
  const int kq = t & 7;        // this thread's k quad inside a chunk (same for its A and B elements)
  const int a_r0 = t >> 3;     // 0..31
  int a_h[A_PER_THREAD], a_w[A_PER_THREAD];
  unsigned a_off[A_PER_THREAD], s_off[A_PER_THREAD];
#pragma unroll
  for (int i = 0; i < A_PER_THREAD; ++i) {
    const long long m = m0 + a_r0 + 32 * i;
    a_h[i] = a_w[i] = -0x40000000;   // row beyond M: never in range
    a_off[i] = s_off[i] = 0;
    if (m < M) {
      a_h[i] = a_w[i] = 0;   // 1x1 convs never leave the pixel: no (h, w) needed -- and no integer divisions in the prologue
      a_off[i] = (unsigned)(m * p.lda * ab);
      if (p.ntaps > 1 || SC) {
        const int HWp = p.H * p.W;
        const int n = (int)(m / HWp);
        const int rem = (int)(m - (long long)n * HWp);
        if (p.ntaps > 1) {
          a_h[i] = rem / p.W;
          a_w[i] = rem - a_h[i] * p.W;
        }
        s_off[i] = (unsigned)((long long)n * p.C * 4);
      }
    }
  }
  unsigned b_off[B_PER_THREAD];
  bool b_ok[B_PER_THREAD];
#pragma unroll
  for (int i = 0; i < B_PER_THREAD; ++i) {
    const int idx = t + 256 * i;
    const int n = idx >> 3;
    b_ok[i] = (idx < B_TOTAL) && (n0 + n < p.Nout);
    b_off[i] = (unsigned)((long long)(n0 + n) * p.ldb * 4);
  }

  // ---- (tap, channel) of this thread's k quad in the next chunk to LOAD: channel k_c inside tap k_tap = 3 * k_th + k_tw
  int k_tap, k_c, k_th, k_tw;
  auto seek = [&](int kabs) {   // from scratch: integer divisions (start of the K range; every chunk of a NARROW instance)
    k_tap = kabs / p.C;
    k_c = kabs - k_tap * p.C;
    k_th = p.ntaps > 1 ? k_tap / 3 : 1;   // 1x1 convs sit on the centre tap (dh = dw = 0)
    k_tw = p.ntaps > 1 ? k_tap - k_th * 3 : 1;
  };
  int kabs = it0 * BK + kq * 4;
  seek(kabs);
  float4 ra[PF][A_PER_THREAD], rs[PF][A_PER_THREAD], rb[PF][B_PER_THREAD];

  // next chunk -> registers, then advance.  Branch-free (live == false turns every offset out of range: no memory traffic, zeros
  // come back) so the compiler knows exactly how many loads are in flight and waits only for the older chunk.
  auto load_chunk = [&](float4* ra_, float4* rs_, float4* rb_, bool live) {
    const int dh = (k_th - 1) * p.dil * p.sign, dw = (k_tw - 1) * p.dil * p.sign;
    const bool kok = live & (k_tap < p.ntaps);
    // (recomputed from (k_th, k_tw, k_c) every chunk: carrying the two byte offsets and (dh, dw) incrementally instead removes every
    // integer multiply from the K loop but measured 4 % SLOWER on the 3x3 decoder convs -- three more live registers per thread)
    const unsigned da = (unsigned)(((dh * p.W + dw) * p.lda + k_c) * ab);
#pragma unroll
    for (int i = 0; i < A_PER_THREAD; ++i) {
      const bool ok = kok & ((unsigned)(a_h[i] + dh) < (unsigned)p.H) & ((unsigned)(a_w[i] + dw) < (unsigned)p.W);
      if (PREC == 1 && abf) ra_[i] = buf_ld4_bf16(rA, ok ? a_off[i] + da : kOob);
      else ra_[i] = buf_ld4(rA, ok ? a_off[i] + da : kOob);
      if (SC) rs_[i] = buf_ld4(rS, ok ? s_off[i] + (unsigned)(k_c * 4) : kOob);
    }
    const unsigned db = (unsigned)(((long long)k_tap * p.b_tap_stride + k_c) * 4);
#pragma unroll
    for (int i = 0; i < B_PER_THREAD; ++i) rb_[i] = buf_ld4(rB, (kok & b_ok[i]) ? b_off[i] + db : kOob);
    if (NARROW) {   // 3x3 convs over fewer than 32 channels: a chunk spans several taps
      kabs += BK;
      seek(kabs);
    } else {        // C >= 32 (or a 1x1 conv): at most one tap boundary per chunk
      k_c += BK;
      const bool wrap = k_c >= p.C;
      k_c = wrap ? k_c - p.C : k_c;
      k_tap += wrap ? 1 : 0;
      k_tw += wrap ? 1 : 0;   // (a 1x1 conv wraps only into out-of-range taps, where the offsets are ignored)
      const bool roll = k_tw == 3;
      k_tw = roll ? 0 : k_tw;
      k_th += roll ? 1 : 0;
    }
  };
  auto store_chunk = [&](float* buf, const float4* ra_, const float4* rs_, const float4* rb_) {
    float* smA = buf;
    float* smB = buf + A_FLOATS;
#pragma unroll
    for (int i = 0; i < A_PER_THREAD; ++i) {
      float4 v = ra_[i];
      if (SC) v = f4mul(v, rs_[i]);
      st4(smA + (kq * BM + (a_r0 ^ kq)) * 4 + 32 * i * 4, v);   // row a_r0 + 32 i, swizzled: (row ^ kq) = 32 i + (a_r0 ^ kq) as kq < 8
    }
#pragma unroll
    for (int i = 0; i < B_PER_THREAD; ++i) {
      const int idx = t + 256 * i;
      if (256 * (i + 1) <= B_TOTAL) st4(smB + (kq * BN + (a_r0 ^ kq)) * 4 + 32 * i * 4, rb_[i]);   // column (t >> 3) + 32 i, swizzled as above
      else st4(idx < B_TOTAL ? smB + (kq * BN + (a_r0 ^ kq)) * 4 + 32 * i * 4 : sm + kDummy, rb_[i]);   // surplus lanes: a scratch slot, no branch
    }
  };

  f32x4 acc[TM][NT];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const float b_qscale = PREC == 2 ? fp8_weight_scale(p.b_amax) : 1.0f;

  // ---- software pipeline: chunk j is computed from LDS buffer (j - it0) & 1 while chunk j+1 waits in registers (loaded one
  // iteration earlier when PF == 2) and the loads of chunk j+PF are in flight; one barrier per chunk.
  load_chunk(ra[0], rs[0], rb[0], it0 < it1);
  store_chunk(sm, ra[0], rs[0], rb[0]);
#pragma unroll
  for (int s_ = 1; s_ < PF; ++s_) load_chunk(ra[s_], rs[s_], rb[s_], it0 + s_ < it1);
  __syncthreads();
  // One chunk: issue the loads of chunk j + PF into register set U (it held chunk j, already in LDS), multiply chunk j out of
  // LDS buffer `cur`, then move chunk j + 1 (register set (U + 1) % PF, loaded one step earlier when PF == 2) into the other buffer.
  int cur = 0;
  auto step = [&](auto UC, int j) {
    constexpr int U = decltype(UC)::value;
    constexpr int NX = (U + 1) % PF;
    load_chunk(ra[U], rs[U], rb[U], j + PF < it1);
    const float* smA = sm + cur * BUF_FLOATS;
    const float* smB = smA + A_FLOATS;
    constexpr bool kAllFirst = TM == 1;   // read the fragments of both k groups before the first MFMA (register budget permitting)
    float4 av[2][TM], bv[2][NT];
    auto read_frags = [&](int q) {
      const int fq = q * 4 + g;
#pragma unroll
      for (int i = 0; i < TM; ++i) av[q][i] = ld4(smA + (fq * BM + wave * 16 * TM + (l15 ^ fq)) * 4 + i * 64);   // (= (row ^ fq) as fq < 8: one address, immediate offsets)
#pragma unroll
      for (int jn = 0; jn < NT; ++jn) bv[q][jn] = ld4(smB + (fq * BN + (l15 ^ fq)) * 4 + jn * 64);
    };
    auto multiply = [&](int q) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const float a = s == 0 ? av[q][i].x : s == 1 ? av[q][i].y : s == 2 ? av[q][i].z : av[q][i].w;
#pragma unroll
          for (int jn = 0; jn < NT; ++jn) {
            const float b = s == 0 ? bv[q][jn].x : s == 1 ? bv[q][jn].y : s == 2 ? bv[q][jn].z : bv[q][jn].w;
            acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i][jn], 0, 0, 0);
          }
        }
      }
    };
    if constexpr (PREC == 2) {
      // fp8 (e4m3) operands: as the bf16 form below with 8 fp8 values per lane and operand (v_cvt_pk_fp8_f32 after scaling)
      read_frags(0);
      read_frags(1);
      long a8[TM], b8[NT];
#pragma unroll
      for (int i = 0; i < TM; ++i) a8[i] = pack_fp8x8(av[0][i], av[1][i], p.a_qscale);
#pragma unroll
      for (int jn = 0; jn < NT; ++jn) b8[jn] = pack_fp8x8(bv[0][jn], bv[1][jn], b_qscale);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < NT; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a8[i], b8[jn], acc[i][jn], 0, 0, 0);
    } else if constexpr (PREC == 1) {
      // bf16 operands, fp32 accumulation: ONE v_mfma_f32_16x16x32_bf16 per output tile and chunk.  Lane group g feeds the k values
      // {4g..4g+3} of both 16-wide halves of the chunk -- the same for A and B, so the fp32 fragment layout in LDS is used as it is
      // and the conversion happens in registers (v_cvt_pk_bf16_f32, round to nearest even).
      read_frags(0);
      read_frags(1);
      bf16x8 a8[TM], b8[NT];
#pragma unroll
      for (int i = 0; i < TM; ++i) a8[i] = pack_bf16x8(av[0][i], av[1][i]);
#pragma unroll
      for (int jn = 0; jn < NT; ++jn) b8[jn] = pack_bf16x8(bv[0][jn], bv[1][jn]);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < NT; ++jn) acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8[i], b8[jn], acc[i][jn], 0, 0, 0);
    } else if (kAllFirst) {
      read_frags(0);
      read_frags(1);
      multiply(0);
      multiply(1);
    } else {
      read_frags(0);
      multiply(0);
      read_frags(1);
      multiply(1);
    }
    store_chunk(sm + (cur ^ 1) * BUF_FLOATS, ra[NX], rs[NX], rb[NX]);   // zeros after the last chunk: nobody reads them
    cur ^= 1;
    __syncthreads();
  };
  typedef std::integral_constant<int, 0> U0;
  typedef std::integral_constant<int, 1 % PF> U1;
  typedef std::integral_constant<int, 2 % PF> U2;
  int it = it0;
  // Main loop: four (PF = 3: six) chunks per trip.  The compiler flushes the load counter at a loop header (it cannot prove across the
  // back edge that only the newest chunks are in flight), so the long body keeps that flush to every fourth chunk; the tail is
  // straight-line.  The register sets rotate with period PF.
  if constexpr (PF == 3) {
    for (; it + 6 <= it1; it += 6) {
      step(U0{}, it);
      step(U1{}, it + 1);
      step(U2{}, it + 2);
      step(U0{}, it + 3);
      step(U1{}, it + 4);
      step(U2{}, it + 5);
    }
    if (it < it1) step(U0{}, it);
    if (it + 1 < it1) step(U1{}, it + 1);
    if (it + 2 < it1) step(U2{}, it + 2);
    if (it + 3 < it1) step(U0{}, it + 3);
    if (it + 4 < it1) step(U1{}, it + 4);
  } else {
    for (; it + 4 <= it1; it += 4) {
      step(U0{}, it);
      step(U1{}, it + 1);
      step(U0{}, it + 2);
      step(U1{}, it + 3);
    }
    if (it < it1) step(U0{}, it);
    if (it + 1 < it1) step(U1{}, it + 1);
    if (it + 2 < it1) step(U0{}, it + 2);
  }

  if constexpr (PREC == 2) {   // undo the operand scales
    const float inv = 1.0f / (p.a_qscale * b_qscale);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] *= inv;
  }
  // ---- epilogue: C/D layout of 16x16x4: col = lane & 15, row = 4 * (lane >> 4) + reg
  if (split) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wave * 16 * TM + i * 16 + g * 4 + r;
        if (m0 + row >= M) continue;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int n = n0 + j * 16 + l15;
          if (n < p.Nout) pdst[(long long)row * pstride + j * 16 + l15] = acc[i][j][r];
        }
      }
    return;
  }
  float* stage = sm + wave * 16 * LDS_STAGE;
  float bj[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + j * 16 + l15;
    bj[j] = (p.bias != nullptr && n < p.Nout) ? p.bias[n] : 0.f;
  }
  const long long HWp = (long long)p.H * p.W;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const long long mbase = m0 + wave * 16 * TM + i * 16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* bb = nullptr;
      if (p.border_bias != nullptr) {
        const long long m = mbase + g * 4 + r;
        if (m < M) {
          const int ni = (int)(m / HWp);
          const int rem = (int)(m - (long long)ni * HWp);
          const int h = rem / p.W, w_ = rem - h * p.W;
          const int cls = (h == 0 ? 0 : (h == p.H - 1 ? 2 : 1)) * 3 + (w_ == 0 ? 0 : (w_ == p.W - 1 ? 2 : 1));
          bb = p.border_bias + ((long long)ni * 9 + cls) * p.Nout;
        }
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = n0 + j * 16 + l15;
        float v = acc[i][j][r] + bj[j];
        if (bb != nullptr && n < p.Nout) v += bb[n];
        if (PREC == 1 && obf) v = bf16_round_f(v);
        acc[i][j][r] = v;
        stage[(g * 4 + r) * LDS_STAGE + j * 16 + l15] = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < NT; ++it) {
      const int idx = it * 64 + lane;
      const int row = idx / (BN / 4), q = idx - row * (BN / 4);
      const long long m = mbase + row;
      const int n = n0 + q * 4;
      if (m < M && n < p.Nout) {
        float4 v = ld4(stage + row * LDS_STAGE + q * 4);
        if (PREC == 1 && obf) {   // (no accumulate into a bf16 tensor: checked on the host)
          stq(reinterpret_cast<bf16s*>(p.Cmat) + m * p.ldc + n, v);
        } else {
          float* dst = p.Cmat + m * p.ldc + n;
          if (p.accumulate) v = f4add(v, ld4(dst));
          st4(dst, v);
        }
      }
    }
    __syncthreads();
  }
  if (!stats) return;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const long long m = m0 + wave * 16 * TM + i * 16 + g * 4 + r;
      if (m >= M) continue;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float v = acc[i][j][r];
        const float u = p.stats_swish ? swish_f(v) : v;
        s1[j] += u;
        s2[j] = fmaf(u, u, s2[j]);
      }
    }
  float* red = sm;  // layout [wave][2][BN]
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    float a = s1[j], b = s2[j];
    a += __shfl_xor(a, 16, 64);
    b += __shfl_xor(b, 16, 64);
    a += __shfl_xor(a, 32, 64);
    b += __shfl_xor(b, 32, 64);
    if (g == 0) {
      red[(wave * 2 + 0) * BN + j * 16 + l15] = a;
      red[(wave * 2 + 1) * BN + j * 16 + l15] = b;
    }
  }
  __syncthreads();
  for (int idx = t; idx < 2 * BN; idx += 256) {
    const int v = idx / BN, col = idx - v * BN;
    const int n = n0 + col;
    if (n < p.Nout) {
      const float r0 = red[(0 * 2 + v) * BN + col] + red[(1 * 2 + v) * BN + col] + red[(2 * 2 + v) * BN + col] + red[(3 * 2 + v) * BN + col];
      p.stats_part[((long long)bx * 2 + v) * p.Nout + n] = r0;
    }
  }
}

// grid = (row tiles, column tiles, K splits).  SPLIT: raw partial tiles [z][M][Nout], the fold kernel finishes them.
template <int TM, int NT, int PF, bool SC, bool SPLIT, bool NARROW, int PREC>
__global__ __launch_bounds__(256, 2) void conv_gemm_nk_k(ConvGemmParams p) {
  __shared__ __attribute__((aligned(16))) float sm[GemmSm<TM, NT>::SM_FLOATS + 4];
  const int nchunks_total = (p.ntaps * p.C + 31) / 32;
  const int it0 = blockIdx.z * p.chunks_per_split;
  int it1 = it0 + p.chunks_per_split;
  if (it1 > nchunks_total) it1 = nchunks_total;
  const unsigned bx = xcd_remap(blockIdx.x, gridDim.x);
  float* pdst = nullptr;
  if (SPLIT) {
    const long long M = (long long)p.Nimg * p.H * p.W;
    pdst = p.partial + ((long long)blockIdx.z * M + (long long)bx * (64 * TM)) * p.Nout + blockIdx.y * (16 * NT);
  }
  conv_gemm_tile<TM, NT, PF, SC, NARROW, PREC>(p, sm, bx, blockIdx.y, it0, it1, pdst, p.Nout);
}

// ------------------------------------------------------------------------------------------------ data-parallel + stream-K remainder
// A layer whose T = (row tiles) x (column tiles) output tiles are not a multiple of the CU count leaves the chip part-idle in its last
// round (the 56x56 decoder convs: 392 tiles of 64 x 112 on 256 CUs -> the CUs holding two tiles set the time, 0.77 of the chip).
// Here the first F = floor(T / CUs) * CUs tiles run as whole tiles (workgroups [0, F)), and the K iterations of the remaining R = T - F
// tiles -- R * nchunks chunks of 32 -- are cut into `parts` equal contiguous ranges, one per workgroup [F, F + parts): a range covers
// the tail of one remainder tile and the head of the next (at most two segments since parts >= R), each written as a raw 64 x BN
// partial into its (tile, slot) slab; sk_fixup_k adds a tile's slots in a fixed order (deterministic), applies the epilogue and
// emits its BN statistics.  Every CU then carries F / CUs whole tiles plus one equal share of the rest.
struct SkPlan {
  int full;      // F: tiles computed whole
  int rem;       // R: tiles cut into parts
  int parts;     // workgroups that share the remainder
  int ipp;       // K chunks per part
  int nchunks;   // K chunks per tile
  int smax;      // slab slots per remainder tile
  int gy;        // column tiles
  float* slab;   // [rem][smax][64][BN]
};

template <int NT, int PF, bool SC, int PREC>
__global__ __launch_bounds__(256, 2) void conv_gemm_sk_k(ConvGemmParams p, SkPlan k) {
  __shared__ __attribute__((aligned(16))) float sm[GemmSm<1, NT>::SM_FLOATS + 4];
  constexpr int BN = 16 * NT;
  const int w = blockIdx.x;
  if (w < k.full) {   // tile index tau -> (row tile tau / gy, column tile tau % gy): the column tiles of a row tile are neighbours
    const unsigned tau = xcd_remap(w, k.full);
    conv_gemm_tile<1, NT, PF, SC, false, PREC>(p, sm, tau / k.gy, tau % k.gy, 0, k.nchunks, nullptr, 0);
    return;
  }
  const int part = w - k.full;
  int lo = part * k.ipp;
  const int total = k.rem * k.nchunks;
  int hi = lo + k.ipp;
  if (hi > total) hi = total;
#pragma unroll 1
  while (lo < hi) {
    const int rt = lo / k.nchunks, c0 = lo - rt * k.nchunks;
    int c1 = c0 + (hi - lo);
    if (c1 > k.nchunks) c1 = k.nchunks;
    const int first = (rt * k.nchunks) / k.ipp;          // first part that touches this tile -> slot 0
    const unsigned tau = k.full + rt;
    float* dst = k.slab + ((long long)rt * k.smax + (part - first)) * (64 * BN);
    conv_gemm_tile<1, NT, PF, SC, false, PREC>(p, sm, tau / k.gy, tau % k.gy, c0, c1, dst, BN);
    lo += c1 - c0;
    __syncthreads();   // the next segment reuses the LDS buffers
  }
}

// Finishes the remainder tiles of conv_gemm_sk_k: grid = (rem); 1024 threads = 32 column-quad slots x 32 row lanes (two rows per
// thread: all slot loads of a row pair are issued together -- the kernel is a few dependent round trips, nothing else).
template <int NT>
__global__ __launch_bounds__(1024) void sk_fixup_k(ConvGemmParams p, SkPlan k) {
  constexpr int BN = 16 * NT, QN = BN / 4;
  __shared__ float4 red[2][32][32];
  const int t = threadIdx.x, q = t & 31, rl = t >> 5;
  const int rt = blockIdx.x;
  const unsigned tau = k.full + rt;
  const int bx = tau / k.gy, by = tau % k.gy;
  const long long M = (long long)p.Nimg * p.H * p.W;
  const long long m0 = (long long)bx * 64;
  const int n = by * BN + q * 4;
  const bool cok = q < QN && n < p.Nout;
  const int first = (rt * k.nchunks) / k.ipp;
  int last = ((rt + 1) * k.nchunks - 1) / k.ipp;
  if (last > k.parts - 1) last = k.parts - 1;
  const int nslots = last - first + 1;
  const float* base = k.slab + (long long)rt * k.smax * (64 * BN) + q * 4;
  float4 s1 = f4zero(), s2 = f4zero();
  if (cok) {
    const float4 bv = p.bias != nullptr ? ld4(p.bias + n) : f4zero();
    const long long HWp = (long long)p.H * p.W;
    float4 v[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) v[i] = ld4(base + (long long)(rl + 32 * i) * BN);
    for (int z = 1; z < nslots; ++z) {
      float4 u[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) u[i] = ld4(base + ((long long)z * 64 + rl + 32 * i) * BN);
#pragma unroll
      for (int i = 0; i < 2; ++i) v[i] = f4add(v[i], u[i]);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long long m = m0 + rl + 32 * i;
      if (m >= M) continue;
      float4 o = f4add(v[i], bv);
      if (p.border_bias != nullptr) {
        const int ni = (int)(m / HWp);
        const int rem = (int)(m - (long long)ni * HWp);
        const int h = rem / p.W, w_ = rem - h * p.W;
        const int cls = (h == 0 ? 0 : (h == p.H - 1 ? 2 : 1)) * 3 + (w_ == 0 ? 0 : (w_ == p.W - 1 ? 2 : 1));
        o = f4add(o, ld4(p.border_bias + ((long long)ni * 9 + cls) * p.Nout + n));
      }
      float* dst = p.Cmat + m * p.ldc + n;
      if (p.accumulate) o = f4add(o, ld4(dst));
      st4(dst, o);
      if (p.stats_swish) o = make_float4(swish_f(o.x), swish_f(o.y), swish_f(o.z), swish_f(o.w));
      s1 = f4add(s1, o);
      s2 = f4fma(o, o, s2);
    }
  }
  if (p.stats_part == nullptr) return;
  red[0][rl][q] = s1;
  red[1][rl][q] = s2;
  __syncthreads();
  if (t < 64) {
    const int v = t >> 5, qq = t & 31;
    const int nn = by * BN + qq * 4;
    if (qq < QN && nn < p.Nout) {
      float4 a = red[v][0][qq];
#pragma unroll 8
      for (int r = 1; r < 32; ++r) a = f4add(a, red[v][r][qq]);
      st4(p.stats_part + ((long long)bx * 2 + v) * p.Nout + nn, a);
    }
  }
}

// ------------------------------------------------------------------------------------------------ short-K 1x1 convs, streamed
// The MBConv expand convs (K = 16..112 input channels) and the backward-data of the project convs (K = 16..112 output channels) are
// memory-bound GEMMs with one to four K chunks: in conv_gemm_nk_k they spend their time in the fixed chain load -> LDS -> barrier
// -> a handful of MFMAs -> staged store -> statistics (9-11 us per launch for 3-4 us of traffic).  Here nothing goes through LDS and
// no wave waits for another: every wave keeps the B fragments of its 16*NT output columns (all of K) in registers, streams 16-row
// groups of A straight from memory in MFMA operand layout (lane (row, g) loads k = 16*kg + 4g..4g+3 as one 16-byte word), multiplies,
// adds the bias, stores its accumulators and accumulates the batch-norm statistics; the next row group is in flight meanwhile.
// One barrier at the very end folds the four waves' statistics.  grid = (row-group blocks, column tiles).
// Workgroup = kStreamWaves waves (each streams its own row groups; nothing is shared but the final fold of the statistics): twice the
// four of a 256-thread block, so a launch leaves half as many partial blocks -- every workgroup of the consumer folds ALL of them.
constexpr int kStreamWaves = 8;
// AIN: the batch norm in front of the conv applied while A is loaded (ConvGemmParams::ain).  Fold scratch: K / 4 channel quads x L
// fold lanes = the 512 threads (L <= 32), double partials.
template <int KC>
struct AinFold {
  static constexpr int NQ = KC * 4;
  static constexpr int L = (64 * 8 / NQ) > 32 ? 32 : (64 * 8 / NQ);
};
template <int KC, int NT, int PREC, bool AIN = false>
__global__ __launch_bounds__(64 * kStreamWaves, 4) void conv1x1_stream_k(ConvGemmParams p, int row_groups) {   // (4 waves per SIMD: <= 128 VGPRs, two workgroups per CU)
  __shared__ float red[kStreamWaves][2][16 * NT];
  constexpr int FL = AinFold<KC>::L, FQ = AinFold<KC>::NQ;
  __shared__ double ain_scr[AIN ? 2 * FL * FQ * 4 : 1];
  __shared__ __attribute__((aligned(16))) float ain_m[AIN ? KC * 16 : 4], ain_sc[AIN ? KC * 16 : 4], ain_b[AIN ? KC * 16 : 4];
  // per-wave staging tile of a finished row group, [16 rows][16 NT + 4]: the accumulators (C/D layout: a lane holds 4 rows x NT
  // columns 64 bytes apart) leave as whole-row 16-byte stores -- a dword store of that layout is four 64-byte segments per instruction
  // and the address path, not HBM, bound the large launches (stamps build, round 3: 3.2 us per row group in the store phase)
  constexpr int SROW = 16 * NT + 4;
  __shared__ __attribute__((aligned(16))) float stage[kStreamWaves][16 * SROW];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int M = p.Nimg * p.H * p.W;                  // host guarantees < 2^31 and 32-bit byte offsets
  const int n0 = blockIdx.y * (16 * NT);
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, kBufRecords, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, kBufRecords, 0x00020000);
  // B fragments: column n0 + 16 j + l15, k = 16 kg + 4 g .. + 3 (zero beyond K / Nout)
  float4 bv[KC][NT];
  auto load_bv = [&]() {
#pragma unroll
    for (int kg = 0; kg < KC; ++kg)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = n0 + j * 16 + l15, k = kg * 16 + g * 4;
        bv[kg][j] = buf_ld4(rB, (n < p.Nout && k < p.C) ? (unsigned)((n * p.ldb + k) * 4) : kOob);
      }
  };
  // reduced-precision operands (PREC 1 = bf16, 2 = fp8 e4m3): two 16-wide K groups make one 32-deep MFMA -- lane group g supplies
  // k = 16 kg + 4g..4g+3 of both groups, for A and B alike (the K order inside an MFMA is free as long as both operands agree)
  constexpr int KC2 = (KC + 1) / 2;
  const float b_qscale = PREC == 2 ? fp8_weight_scale(p.b_amax) : 1.0f;
  const float out_scale = PREC == 2 ? 1.0f / (p.a_qscale * b_qscale) : 1.0f;
  bf16x8 bq16[PREC == 1 ? KC2 : 1][PREC == 1 ? NT : 1];
  long bq8[PREC == 2 ? KC2 : 1][PREC == 2 ? NT : 1];
  auto prep_b = [&]() {
    load_bv();
    if constexpr (PREC != 0) {
#pragma unroll
      for (int k2 = 0; k2 < KC2; ++k2)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const float4 lo = bv[2 * k2][j], hi = (2 * k2 + 1 < KC) ? bv[(2 * k2 + 1 < KC) ? 2 * k2 + 1 : 0][j] : f4zero();
          if constexpr (PREC == 1) bq16[k2][j] = pack_bf16x8(lo, hi);
          else bq8[k2][j] = pack_fp8x8(lo, hi, b_qscale);
        }
    }
  };
  if constexpr (!AIN) prep_b();   // (AIN: requested behind the fold of the statistics, whose batches of loads need the registers)
  float bj[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + j * 16 + l15;
    bj[j] = (p.bias != nullptr && n < p.Nout) ? p.bias[n] : 0.f;
  }
  float s1[NT], s2[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) s1[j] = s2[j] = 0.f;
  const bool stats = p.stats_part != nullptr;
  const bool bnb = stats && p.bnb_x != nullptr;
  const bool gated_part = p.gp_part != nullptr;
  const int HWs = p.H * p.W;
  float bmean[NT], brstd[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + j * 16 + l15;
    bmean[j] = (bnb && n < p.Nout) ? p.bnb_mean[n] : 0.f;
    brstd[j] = (bnb && n < p.Nout) ? p.bnb_rstd[n] : 0.f;
  }
  const int stride = gridDim.x * kStreamWaves;
  const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc((void*)(AIN && p.ain_res != nullptr ? p.ain_res : p.A), 0, kBufRecords, 0x00020000);
  auto load_a = [&](int rg, float4* a) {
    const int m = rg * 16 + l15;
#pragma unroll
    for (int kg = 0; kg < KC; ++kg) {
      const int k = kg * 16 + g * 4;
      a[kg] = buf_ld4(rA, (rg < row_groups && m < M && k < p.C) ? (unsigned)((m * p.lda + k) * 4) : kOob);
    }
  };
  // (AIN) the residual quads and the per-image scale of the same rows, requested with the rows themselves
  auto load_side = [&](int rg, float4* r, float& isc) {
    if constexpr (AIN) {
      const int m = rg * 16 + l15;
      const bool rok = rg < row_groups && m < M;
#pragma unroll
      for (int kg = 0; kg < KC; ++kg) {
        const int k = kg * 16 + g * 4;
        r[kg] = buf_ld4(rR, (p.ain_res != nullptr && rok && k < p.C) ? (unsigned)((m * p.ain_ldr + k) * 4) : kOob);
      }
      isc = (p.ain_scale != nullptr && rok) ? p.ain_scale[m / HWs] : 1.f;
    }
  };
  int rg = blockIdx.x * kStreamWaves + wave;
  float4 a_cur[KC], a_nxt[KC];
  float4 r_cur[AIN ? KC : 1], r_nxt[AIN ? KC : 1];
  float is_cur = 1.f, is_nxt = 1.f;
  load_a(rg, a_cur);
  if constexpr (AIN) {
    // ---- fold the stage-1 partials of the batch norm in front (every workgroup, all K channels): thread (quad fq, lane fl) adds the
    // blocks fl, fl + FL, ... of its four channels in double, eight loads in flight; the FL lanes of a channel are summed in lane order
    const int fq = t % FQ, fl = t / FQ;
    const int fc = fq * 4;
    FoldAcc f;
    if (fl < FL && fc < p.C) {
      // (a descriptor that ends with the last partial block: the surplus slots of a batch read zeros -- no clamps, no predicates.  The
      //  slot displacement goes into the VECTOR offset: the range check of a raw buffer load covers voffset + the instruction offset, the
      //  scalar offset is not promised to take part in it -- ADVICE r05)
      const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc((void*)p.ain.part, 0, p.ain.nblk * 2 * p.C * 4, 0x00020000);
      const int bstride = FL * 2 * p.C * 4;
      for (int k = fl; k < p.ain.nblk; k += FL * 8) {
        const int off = (k * 2 * p.C + fc) * 4;
        u32x4 u[8], v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          u[j] = __builtin_amdgcn_raw_buffer_load_b128(rP, off + j * bstride, 0, 0);
          v[j] = __builtin_amdgcn_raw_buffer_load_b128(rP, off + j * bstride + p.C * 4, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          f.a0 += __uint_as_float(u[j].x); f.a1 += __uint_as_float(u[j].y); f.a2 += __uint_as_float(u[j].z); f.a3 += __uint_as_float(u[j].w);
          f.b0 += __uint_as_float(v[j].x); f.b1 += __uint_as_float(v[j].y); f.b2 += __uint_as_float(v[j].z); f.b3 += __uint_as_float(v[j].w);
        }
      }
    }
    prep_b();
    load_side(rg, r_cur, is_cur);
    if (fl < FL) {
      double* d0 = ain_scr + ((0 * FL + fl) * FQ + fq) * 4;
      double* d1 = ain_scr + ((1 * FL + fl) * FQ + fq) * 4;
      d0[0] = f.a0; d0[1] = f.a1; d0[2] = f.a2; d0[3] = f.a3;
      d1[0] = f.b0; d1[1] = f.b1; d1[2] = f.b2; d1[3] = f.b3;
    }
    __syncthreads();
    if (t < KC * 16) {
      float mf = 0.f, scf = 0.f, bf = 0.f;
      if (t < p.C) {
        double s = 0.0, ss = 0.0;
#pragma unroll 8
        for (int k = 0; k < FL; ++k) {
          s += ain_scr[(0 * FL + k) * FQ * 4 + t];
          ss += ain_scr[(1 * FL + k) * FQ * 4 + t];
        }
        const double m = s * p.ain.inv_n;
        double var = ss * p.ain.inv_n - m * m;
        if (var < 0.0) var = 0.0;
        mf = (float)m;
        const float rf = (float)(1.0 / sqrt(var + (double)p.ain.eps));
        scf = rf * p.ain_gamma[t];
        bf = p.ain_beta[t];
        if (blockIdx.x == 0 && blockIdx.y == 0) {
          p.ain.mean[t] = mf;
          p.ain.rstd[t] = rf;
          if (p.ain.moving_mean != nullptr) {
            const float mm = p.ain.moving_mean[t], mv = p.ain.moving_var[t];
            p.ain.moving_mean[t] = mm - (mm - mf) * p.ain.one_minus_momentum;
            p.ain.moving_var[t] = mv - (mv - (float)(var * (double)p.ain.ema_var_factor)) * p.ain.one_minus_momentum;
          }
        }
      }
      ain_m[t] = mf;      // (channels beyond K: zeros -- their A elements become zeros, as the loader's out-of-range zeros were)
      ain_sc[t] = scf;
      ain_b[t] = bf;
    }
    __syncthreads();
  }
  // (AIN) raw rows -> the conv's operand, in place; column tile 0 also writes the finished tensor
  auto bn_in = [&](int rg, float4* a, const float4* r, float isc) {
    if constexpr (AIN) {
      const int m = rg * 16 + l15;
      // (the per-channel constants are re-read from LDS for every row group: hoisted out of the loop -- loop-invariant loads -- they
      //  would hold 12 KC registers and the kernel would spill)
      asm volatile("" ::: "memory");
#pragma unroll
      for (int kg = 0; kg < KC; ++kg) {
        const int k = kg * 16 + g * 4;
        const float4 mq = ld4(ain_m + k), sq = ld4(ain_sc + k), bq = ld4(ain_b + k);
        float4 o;
        o.x = fmaf(a[kg].x - mq.x, sq.x, bq.x);
        o.y = fmaf(a[kg].y - mq.y, sq.y, bq.y);
        o.z = fmaf(a[kg].z - mq.z, sq.z, bq.z);
        o.w = fmaf(a[kg].w - mq.w, sq.w, bq.w);
        if (p.ain_scale != nullptr) o = f4scale(o, isc);
        if (p.ain_res != nullptr) o = f4add(o, r[kg]);
        a[kg] = o;
        if (blockIdx.y == 0 && m < M && k < p.C) st4(p.ain_out + (long long)m * p.ain_ldo + k, o);
      }
    }
  };
  bn_in(rg, a_cur, r_cur, is_cur);
  // (AIN with K > 64: no register room for the next row group beside the residual's -- and these are the 14x14 layers, one or two row
  //  groups per wave: the next rows are requested when the current ones are done)
  constexpr bool PF = !(AIN && KC >= 5);
  for (; rg < row_groups; rg += stride) {
    if constexpr (PF) {
      load_a(rg + stride, a_nxt);
      load_side(rg + stride, r_nxt, is_nxt);
    }
    // the side operand of the gate-gradient / BN-backward epilogues (the element of gp_x / bnb_x beside each accumulator element, C/D
    // layout) is requested here, under the MFMAs, not where it is used: one memory round trip per row group less
    float side[4][NT];
    if (gated_part || bnb) {   // (uniform)
      const float* sx = gated_part ? p.gp_x : p.bnb_x;
      const int sld = gated_part ? p.gp_ldx : p.bnb_ldx;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = rg * 16 + g * 4 + r;
        const long long mcl = m < M ? m : M - 1;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int n = n0 + j * 16 + l15;
          if (PREC == 1 && p.side_bf16 && gated_part)
            side[r][j] = __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(sx)[mcl * sld + (n < p.Nout ? n : p.Nout - 1)] << 16);
          else
            side[r][j] = sx[mcl * sld + (n < p.Nout ? n : p.Nout - 1)];
        }
      }
    }
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (PREC == 0) {
#pragma unroll
      for (int kg = 0; kg < KC; ++kg)
#pragma unroll
        for (int sI = 0; sI < 4; ++sI) {
          const float a = sI == 0 ? a_cur[kg].x : sI == 1 ? a_cur[kg].y : sI == 2 ? a_cur[kg].z : a_cur[kg].w;
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const float b = sI == 0 ? bv[kg][j].x : sI == 1 ? bv[kg][j].y : sI == 2 ? bv[kg][j].z : bv[kg][j].w;
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
          }
        }
    } else {
#pragma unroll
      for (int k2 = 0; k2 < KC2; ++k2) {
        const float4 lo = a_cur[2 * k2], hi = (2 * k2 + 1 < KC) ? a_cur[(2 * k2 + 1 < KC) ? 2 * k2 + 1 : 0] : f4zero();
        if constexpr (PREC == 1) {
          const bf16x8 a8 = pack_bf16x8(lo, hi);
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, bq16[k2][j], acc[j], 0, 0, 0);
        } else {
          const long a8 = pack_fp8x8(lo, hi, p.a_qscale);
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a8, bq8[k2][j], acc[j], 0, 0, 0);
        }
      }
    }
    // C/D layout: column l15 of tile j, rows 4 g + r
    float gp0[NT], gp1[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) gp0[j] = gp1[j] = 0.f;
    const int gp_bound = gated_part ? ((rg * 16) / HWs + 1) * HWs : 0;   // first row of the image after the one this group starts in
    // The extras of the epilogue are picked ONCE per row group by uniform branches and each form is straight-line code with
    // predicated contributions (rows >= M / columns >= Nout add zeros; their side loads are clamped).  Testing the run-time flags per
    // element cost ~30 instructions and four scalar branches for each of the 4 NT values of a lane: the launch was bound by
    // instruction issue in its epilogue (stamps build, round 3: 1.9 us of a row group's 2.8 us).
    // MODE 0: store only; 1: BN statistics; 2: statistics of swish(value); 3: stage 1 of the consumer BN's backward; 4: gate-gradient sums
    auto epilogue = [&](auto mode_c) {
      constexpr int MODE = decltype(mode_c)::value;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = rg * 16 + g * 4 + r;
        const bool rok = m < M;
        const long long mcl = rok ? m : M - 1;
        float rscale = 1.f;
        if constexpr (MODE == 3) rscale = p.bnb_scale != nullptr ? p.bnb_scale[mcl / HWs] : 1.f;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int n = n0 + j * 16 + l15;
          const bool cok = n < p.Nout;
          const int ncl = cok ? n : p.Nout - 1;
          float v = (PREC == 2 ? acc[j][r] * out_scale : acc[j][r]) + bj[j];
          if (PREC == 1 && p.out_bf16) v = bf16_round_f(v);   // (the sums below see what the consumers will read)
          stage[wave][(g * 4 + r) * SROW + j * 16 + l15] = v;
          if constexpr (MODE == 4) {
            const float pv = (rok && cok) ? v * side[r][j] : 0.f;
            gp0[j] += m < gp_bound ? pv : 0.f;
            gp1[j] += m < gp_bound ? 0.f : pv;
          } else if constexpr (MODE == 3) {   // (mliis_conv2d_bwd_data_bn): {sum g, sum g * xhat}
            const float gq = (rok && cok) ? v * rscale : 0.f;
            const float xh = (side[r][j] - bmean[j]) * brstd[j];
            s1[j] += gq;
            s2[j] = fmaf(gq, xh, s2[j]);
          } else if constexpr (MODE == 1 || MODE == 2) {
            const float u0 = MODE == 2 ? swish_f(v) : v;
            const float u = rok ? u0 : 0.f;
            s1[j] += u;
            s2[j] = fmaf(u, u, s2[j]);
          }
        }
      }
    };
    if (bnb) epilogue(std::integral_constant<int, 3>{});
    else if (gated_part) epilogue(std::integral_constant<int, 4>{});
    else if (stats && p.stats_swish) epilogue(std::integral_constant<int, 2>{});
    else if (stats) epilogue(std::integral_constant<int, 1>{});
    else epilogue(std::integral_constant<int, 0>{});
    if (p.c_block != 0) {   // (uniform) group-blocked output: 16 consecutive lanes write the 16 rows of one channel quad (256 contiguous
                            // bytes per quad for v = 4; two 128-byte runs for v = 2)
      constexpr int QN = 4 * NT;
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const int e = lane + 64 * u;
        const int q4 = e >> 4, row = e & 15;
        const int m = rg * 16 + row, n = n0 + q4 * 4;
        const float4 v4 = ld4(&stage[wave][row * SROW + q4 * 4]);
        if (q4 < QN && m < M && n < p.Nout) {
          if (p.c_block == 4) {
            st4(p.Cmat + ((long long)(n >> 2) * M + m) * 4, v4);
          } else {
            float* d0 = p.Cmat + ((long long)(n >> 1) * M + m) * 2;
            *reinterpret_cast<float2*>(d0) = make_float2(v4.x, v4.y);
            *reinterpret_cast<float2*>(d0 + (long long)M * 2) = make_float2(v4.z, v4.w);
          }
        }
      }
    } else {   // the staged tile -> memory, a float4 per lane and trip (the wave reads what it wrote itself: LDS is in order, no barrier)
      constexpr int QN = 4 * NT;
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const int e = lane + 64 * u;
        const int row = e / QN, q4 = e - row * QN;
        const int m = rg * 16 + row, n = n0 + q4 * 4;
        const float4 v4 = ld4(&stage[wave][row * SROW + q4 * 4]);
        if (m < M && n < p.Nout) {
          if (PREC == 1 && p.out_bf16) stq(reinterpret_cast<bf16s*>(p.Cmat) + (long long)m * p.ldc + n, v4);
          else st4(p.Cmat + (long long)m * p.ldc + n, v4);
        }
      }
    }
    if (gated_part) {   // the four row quads of the group (lane groups g) folded; lanes g == 0 publish [rg][slot][column]
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        float a = gp0[j], b = gp1[j];
        a += __shfl_xor(a, 16, 64);
        b += __shfl_xor(b, 16, 64);
        a += __shfl_xor(a, 32, 64);
        b += __shfl_xor(b, 32, 64);
        const int n = n0 + j * 16 + l15;
        if (g == 0 && n < p.Nout) {
          p.gp_part[((long long)rg * 2 + 0) * p.Nout + n] = a;
          p.gp_part[((long long)rg * 2 + 1) * p.Nout + n] = b;
        }
      }
    }
    if constexpr (PF) {
#pragma unroll
      for (int kg = 0; kg < KC; ++kg) a_cur[kg] = a_nxt[kg];
      if constexpr (AIN) {
        if (rg + stride < row_groups) bn_in(rg + stride, a_cur, r_nxt, is_nxt);
      }
    } else {
      if (rg + stride < row_groups) {
        load_a(rg + stride, a_cur);
        load_side(rg + stride, r_cur, is_cur);
        bn_in(rg + stride, a_cur, r_cur, is_cur);
      }
    }
  }
  if (!stats) return;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    float a = s1[j], b = s2[j];
    a += __shfl_xor(a, 16, 64);
    b += __shfl_xor(b, 16, 64);
    a += __shfl_xor(a, 32, 64);
    b += __shfl_xor(b, 32, 64);
    if (g == 0) {
      red[wave][0][j * 16 + l15] = a;
      red[wave][1][j * 16 + l15] = b;
    }
  }
  __syncthreads();
  for (int idx = t; idx < 2 * 16 * NT; idx += 64 * kStreamWaves) {
    const int v = idx / (16 * NT), col = idx - v * (16 * NT);
    const int n = n0 + col;
    if (n < p.Nout) {
      float a = red[0][v][col];
#pragma unroll
      for (int wv = 1; wv < kStreamWaves; ++wv) a += red[wv][v][col];   // (wave order: deterministic)
      p.stats_part[((long long)blockIdx.x * 2 + v) * p.Nout + n] = a;
    }
  }
}

// ------------------------------------------------------------------------------------------------ long-K 1x1 convs on small maps
// The MBConv project convs forward (K = 240..672 expanded channels, SE gate applied on load) and the expand convs' backward-data
// (K = the same expanded channels) on the 28x28 / 14x14 maps have 1568-6272 rows: conv_gemm_nk_k needs split-K to occupy the chip
// (25-98 row tiles) and then a second launch to fold the slabs -- 10 + 5 us for < 1 us of work.  Here the K split happens INSIDE a
// workgroup: WV waves share one 16-row group, wave w multiplies the K slice [w * 16 KC, (w + 1) * 16 KC) exactly as
// conv1x1_stream_k does (B fragments of the slice in registers, A straight from memory in MFMA operand layout, the next row group in
// flight), the WV partial accumulators meet in LDS, and the first 64 NT threads finish the rows: sum in wave order (deterministic),
// bias, accumulate, coalesced float4 stores, BN statistics.  One launch, no slabs, every load of a row group issued at once.
// grid = (row-group blocks, column tiles); block = 64 WV threads.
#ifdef KS_DBG
#define KS_STAMP(k) do { ks_st[k] = wall_clock64(); } while (0)
#else
#define KS_STAMP(k) do { } while (0)
#endif
template <int KC, int NT, int WV, int PREC>
__global__ __launch_bounds__(64 * WV, (KC <= 6 && KC * NT <= 8) ? 4 : 2) void conv1x1_ksplit_k(ConvGemmParams p, int row_groups) {   // (KC <= 6, KC NT <= 8: two workgroups per CU)
#ifdef KS_DBG
  unsigned long long ks_st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  KS_STAMP(0);
  constexpr int BN = 16 * NT, RS = BN + 4, QN = BN / 4;
  __shared__ __attribute__((aligned(16))) float red[WV][16][RS];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int M = p.Nimg * p.H * p.W, HW = p.H * p.W;
  const int bx = blockIdx.x, by = blockIdx.y;
  const int n0 = by * BN;
  const int k0 = wave * 16 * KC;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, kBufRecords, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, kBufRecords, 0x00020000);
  const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void*)(p.a_scale ? p.a_scale : p.A), 0, kBufRecords, 0x00020000);
  const bool gated = p.a_scale != nullptr;
  float4 bv[KC][NT];
#pragma unroll
  for (int kg = 0; kg < KC; ++kg)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + j * 16 + l15, k = k0 + kg * 16 + g * 4;
      bv[kg][j] = buf_ld4(rB, (n < p.Nout && k < p.C) ? (unsigned)((n * p.ldb + k) * 4) : kOob);
    }
  constexpr int KC2 = (KC + 1) / 2;
  const float b_qscale = PREC == 2 ? fp8_weight_scale(p.b_amax) : 1.0f;
  const float out_scale = PREC == 2 ? 1.0f / (p.a_qscale * b_qscale) : 1.0f;
  bf16x8 bq16[PREC == 1 ? KC2 : 1][PREC == 1 ? NT : 1];
  long bq8[PREC == 2 ? KC2 : 1][PREC == 2 ? NT : 1];
  if constexpr (PREC != 0) {
#pragma unroll
    for (int k2 = 0; k2 < KC2; ++k2)
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float4 lo = bv[2 * k2][j], hi = (2 * k2 + 1 < KC) ? bv[(2 * k2 + 1 < KC) ? 2 * k2 + 1 : 0][j] : f4zero();
        if constexpr (PREC == 1) bq16[k2][j] = pack_bf16x8(lo, hi);
        else bq8[k2][j] = pack_fp8x8(lo, hi, b_qscale);
      }
  }
  // finishing threads: t < 16 * QN own (row t / QN, column quad t % QN) of every row group of this workgroup
  const bool fin = t < 16 * QN;
  const int frow = fin ? t / QN : 0, fq = fin ? t - frow * QN : 0;
  const int fn = n0 + fq * 4;
  const bool fcol = fin && fn < p.Nout;
  const float4 fbias = (fcol && p.bias != nullptr) ? ld4(p.bias + fn) : f4zero();
  float4 s1 = f4zero(), s2 = f4zero();
  const bool stats = p.stats_part != nullptr;
  const bool bnb = stats && p.bnb_x != nullptr;
  const float4 bmean = (bnb && fcol) ? ld4(p.bnb_mean + fn) : f4zero(), brstd = (bnb && fcol) ? ld4(p.bnb_rstd + fn) : f4zero();
  // The squeeze-excite gate of a row group's (at most two) images goes through LDS: one 16-byte load per thread for the whole
  // workgroup instead of KC per lane -- a third of the launch's vector-memory instructions, each of them the same 64 bytes for the
  // sixteen rows of a wave (stamps build, round 4: the operand phase of the 672-channel project convs is bound by the number of
  // vector-memory instructions, 5.5 us for 18 per lane).
  // (The host sends gated calls here only for maps of >= 16 pixels.)
  __shared__ __attribute__((aligned(16))) float sgate[2][16 * KC * WV];
  auto load_a = [&](int rg, float4* a) {
    const int m = rg * 16 + l15;
    const bool rok = rg < row_groups && m < M;
#pragma unroll
    for (int kg = 0; kg < KC; ++kg) {
      const int k = k0 + kg * 16 + g * 4;
      const bool ok = rok && k < p.C;
      if (PREC == 1 && p.a_bf16) a[kg] = buf_ld4_bf16(rA, ok ? (unsigned)((m * p.lda + k) * 2) : kOob);
      else a[kg] = buf_ld4(rA, ok ? (unsigned)((m * p.lda + k) * 4) : kOob);
    }
  };
  // the gate rows of the two images of row group rg: one float4 per thread of the first 2 GQ, requested a row group ahead
  constexpr int GQ = 4 * KC * WV;                       // float4 per image row of the gate slice the workgroup covers (K <= 16 KC WV)
  auto gate_fetch = [&](int rg) {
    float4 v = f4zero();
    if (gated && t < 2 * GQ) {
      const int im = t / GQ, q = t - im * GQ;
      const int m0 = rg * 16;
      const int img = (m0 < M ? m0 : M - 1) / HW + im;
      if (rg < row_groups && img < p.Nimg && q * 4 < p.C) v = buf_ld4(rS, (unsigned)((img * p.C + q * 4) * 4));
    }
    return v;
  };
  auto gate_apply = [&](int rg, float4* a) {
    const int m = rg * 16 + l15;
    const int im = (m < M ? m : M - 1) / HW - (rg * 16 < M ? rg * 16 : M - 1) / HW;   // 0 or 1: the image of this row among the group's
#pragma unroll
    for (int kg = 0; kg < KC; ++kg) a[kg] = f4mul(a[kg], ld4(&sgate[im][k0 + kg * 16 + g * 4]));
  };
  int rg = bx;
  float4 a_cur[KC], a_nxt[KC];
  float4 gq = gate_fetch(rg);
  load_a(rg, a_cur);
  KS_STAMP(1);
  for (; rg < row_groups; rg += gridDim.x) {
    load_a(rg + gridDim.x, a_nxt);
    if (gated) {   // (uniform)
      if (t < 2 * GQ) st4(&sgate[0][0] + t * 4, gq);
      __syncthreads();
      gate_apply(rg, a_cur);
      gq = gate_fetch(rg + gridDim.x);   // (stored to LDS at the top of the next row group: two barriers behind these reads)
    }
    // what the finishing threads add to / read beside their float4 (the accumulate target, the BN input of the stage-1 sums) is
    // requested here, under the MFMAs and the barrier, not after them: one memory round trip per row group less
    float4 pre_dst = f4zero(), pre_x = f4zero();
    {
      const int mf = rg * 16 + frow;
      const bool pok = fcol && mf < M;
      if (p.accumulate && pok) pre_dst = ld4(p.Cmat + (long long)mf * p.ldc + fn);
      if (bnb && pok) pre_x = ld4(p.bnb_x + (long long)mf * p.bnb_ldx + fn);
    }
    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (PREC == 0) {
#pragma unroll
      for (int kg = 0; kg < KC; ++kg)
#pragma unroll
        for (int sI = 0; sI < 4; ++sI) {
          const float a = sI == 0 ? a_cur[kg].x : sI == 1 ? a_cur[kg].y : sI == 2 ? a_cur[kg].z : a_cur[kg].w;
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const float b = sI == 0 ? bv[kg][j].x : sI == 1 ? bv[kg][j].y : sI == 2 ? bv[kg][j].z : bv[kg][j].w;
            acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j], 0, 0, 0);
          }
        }
    } else {
#pragma unroll
      for (int k2 = 0; k2 < KC2; ++k2) {
        const float4 lo = a_cur[2 * k2], hi = (2 * k2 + 1 < KC) ? a_cur[(2 * k2 + 1 < KC) ? 2 * k2 + 1 : 0] : f4zero();
        if constexpr (PREC == 1) {
          const bf16x8 a8 = pack_bf16x8(lo, hi);
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, bq16[k2][j], acc[j], 0, 0, 0);
        } else {
          const long a8 = pack_fp8x8(lo, hi, p.a_qscale);
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(a8, bq8[k2][j], acc[j], 0, 0, 0);
        }
      }
    }
    // C/D layout: column l15 of tile j, rows 4 g + r -> this wave's plane of the staging tile
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int j = 0; j < NT; ++j) red[wave][g * 4 + r][j * 16 + l15] = PREC == 2 ? acc[j][r] * out_scale : acc[j][r];
    if (rg == bx) KS_STAMP(2);
    __syncthreads();
    if (rg == bx) KS_STAMP(3);
    const int m = rg * 16 + frow;
    if (fcol && m < M) {
      float4 v = ld4(&red[0][frow][fq * 4]);
#pragma unroll
      for (int w = 1; w < WV; ++w) v = f4add(v, ld4(&red[w][frow][fq * 4]));
      v = f4add(v, fbias);
      float* dst = p.Cmat + (long long)m * p.ldc + fn;
      if (p.accumulate) v = f4add(v, pre_dst);
      st4(dst, v);
      if (bnb) {   // stage 1 of the consumer batch norm's backward
        const float4 xv = pre_x;
        if (p.bnb_scale != nullptr) v = f4scale(v, p.bnb_scale[m / HW]);
        const float4 xh = make_float4((xv.x - bmean.x) * brstd.x, (xv.y - bmean.y) * brstd.y, (xv.z - bmean.z) * brstd.z,
                                      (xv.w - bmean.w) * brstd.w);
        s1 = f4add(s1, v);
        s2 = f4fma(v, xh, s2);
      } else if (stats) {
        if (p.stats_swish) v = make_float4(swish_f(v.x), swish_f(v.y), swish_f(v.z), swish_f(v.w));
        s1 = f4add(s1, v);
        s2 = f4fma(v, v, s2);
      }
    }
    if (rg == bx) KS_STAMP(4);
    __syncthreads();   // the staging tile is rewritten by the next row group
#pragma unroll
    for (int kg = 0; kg < KC; ++kg) a_cur[kg] = a_nxt[kg];
  }
  KS_STAMP(5);
#ifdef KS_DBG
  auto ks_flush = [&]() {
    KS_STAMP(7);
    if (p.dbg_stamps != nullptr && threadIdx.x == 0)
      for (int k = 0; k < 8; ++k) p.dbg_stamps[((long long)blockIdx.y * gridDim.x + blockIdx.x) * 8 + k] = ks_st[k];
  };
#else
  auto ks_flush = [&]() {};
#endif
  if (!stats) { ks_flush(); return; }
  // column sums over the 16 rows of the finishing threads: through the (now free) staging tile, [v][row][quad]
  float4* fold = reinterpret_cast<float4*>(&red[0][0][0]);
  if (fin) {
    fold[(0 * 16 + frow) * QN + fq] = s1;
    fold[(1 * 16 + frow) * QN + fq] = s2;
  }
  __syncthreads();
  if (t < 2 * QN) {
    const int v = t / QN, q = t - v * QN;
    const int n = n0 + q * 4;
    if (n < p.Nout) {
      float4 a = fold[(v * 16) * QN + q];
#pragma unroll
      for (int r = 1; r < 16; ++r) a = f4add(a, fold[(v * 16 + r) * QN + q]);
      st4(p.stats_part + ((long long)bx * 2 + v) * p.Nout + n, a);
    }
  }
  KS_STAMP(6);
  ks_flush();
}

// ------------------------------------------------------------------------------------------------ backward-filter
struct FilterGradParams {
  const float* X;
  int ldx;
  int Nimg, H, W;
  int C;
  int ntaps, dil;
  const float* dY;
  int lddy;
  int Nout;
  float* partial;  // [splits][ntaps*C][Nout]
  int rows_per_split;
  const float* x_scale;  // nullable [Nimg][C]: X[m][c] *= x_scale[image(m)][c] on load
  int multitap;          // != 0: ntaps * C <= 64 * TMF -- ONE channel block holds all taps (its rows are the flattened (tap, channel)
                         //       index), so dY is read once instead of once per tap (the 8-channel sliver of the RSD concat)
  int x_bf16 = 0, dy_bf16 = 0;   // bf16 STORAGE of X / dY (an expanded MBConv tensor: a1, dz0); honoured by the bf16-operand instances only
};

// ------------------------------------------------------------------------------------------------ backward-filter kernel
// The pipeline of conv_gemm_nk_k applied to the pixel reduction: branch-free buffer loads
// (out-of-range rows and halo pixels return zeros), per-row (h, w) advanced incrementally instead of two integer divisions per row
// and chunk, double-buffered LDS (one barrier per 32-pixel chunk) and a two-chunk register prefetch.
template <int TMF, int NT>
struct FilterSm {
  static constexpr int BCI = 64 * TMF, BN = 16 * NT, BKM = 32;
  static constexpr int LDX = BCI + 16;
  static constexpr int LDD = (BN % 32 == 0) ? BN + 16 : BN;
  static constexpr int BUF_FLOATS = BKM * LDX + BKM * LDD;
};

// the block (bx, by, bz) of one filter-gradient problem: bx = (tap, channel block), by = output-column tile, bz = pixel split
template <int TMF, int NT, bool SC, bool BF>
__device__ __forceinline__ void conv_filter_grad2_body(const FilterGradParams& p, float* __restrict__ sm, int bx, int by, int bz) {
  constexpr int BCI = 64 * TMF, BN = 16 * NT, BKM = 32;
  constexpr int LDX = FilterSm<TMF, NT>::LDX;
  constexpr int LDD = FilterSm<TMF, NT>::LDD;
  constexpr int X_PER_THREAD = (BKM * (BCI / 4)) / 256;  // 2 * TMF
  constexpr int D_TOTAL = BKM * (BN / 4);
  constexpr int D_PER_THREAD = (D_TOTAL + 255) / 256;
  constexpr int X_RSTEP = 256 / (BCI / 4);
  constexpr int BUF_FLOATS = FilterSm<TMF, NT>::BUF_FLOATS;
  constexpr int PF = (TMF == 2 && (SC || NT >= 8)) ? 1 : 2;   // register budget: one staging set for the widest instances

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int M = p.Nimg * p.H * p.W;            // host guarantees < 2^31
  const bool mt = p.multitap != 0;
  const int cblocks = mt ? 1 : (p.C + BCI - 1) / BCI;
  const int tap = mt ? 0 : bx / cblocks;
  const int ci0 = mt ? 0 : (bx - tap * cblocks) * BCI;
  const int n0 = by * BN;
  const int mbeg = bz * p.rows_per_split;
  int mend = mbeg + p.rows_per_split;
  if (mend > M) mend = M;
  const int HW = p.H * p.W;
  const int adv_h = BKM / p.W, adv_w = BKM - adv_h * p.W;   // one chunk = 32 pixels further along the flattened (n, h, w) index
  const bool xbf = BF && p.x_bf16 != 0, dbf = BF && p.dy_bf16 != 0;   // (uniform; fp32 instances: compile-time false)
  const int xb = xbf ? 2 : 4, db = dbf ? 2 : 4;                       // bytes per element of X / dY

  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, kBufRecords, 0x00020000);
  const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc((void*)p.dY, 0, kBufRecords, 0x00020000);
  const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void*)(SC ? p.x_scale : p.X), 0, kBufRecords, 0x00020000);

  // ---- per-thread X rows: position of the row the NEXT load will fetch
  const int x_cq = t % (BCI / 4);
  const int x_r0 = t / (BCI / 4);
  // tap / first channel of the quad this thread loads (block-uniform tap, or per thread in multitap mode)
  int l_tap = tap, l_c = ci0 + x_cq * 4;
  if (mt) {
    l_tap = (x_cq * 4) / p.C;
    l_c = x_cq * 4 - l_tap * p.C;
  }
  const bool x_cok = mt ? l_tap < p.ntaps : l_c < p.C;
  int dh = 0, dw = 0;
  if (p.ntaps > 1) {
    dh = (l_tap / 3 - 1) * p.dil;
    dw = (l_tap % 3 - 1) * p.dil;
  }
  int x_m[X_PER_THREAD], x_h[X_PER_THREAD], x_w[X_PER_THREAD], x_n[X_PER_THREAD];
  unsigned x_off[X_PER_THREAD];
#pragma unroll
  for (int i = 0; i < X_PER_THREAD; ++i) {
    const int m = mbeg + x_r0 + X_RSTEP * i;
    x_m[i] = m;
    x_n[i] = m / HW;
    const int rem = m - x_n[i] * HW;
    x_h[i] = rem / p.W;
    x_w[i] = rem - x_h[i] * p.W;
    x_off[i] = (unsigned)((((long long)m + (long long)dh * p.W + dw) * p.ldx + l_c) * xb);
  }
  const unsigned x_step = (unsigned)(BKM * p.ldx * xb);
  // ---- per-thread dY elements
  int d_m[D_PER_THREAD];
  unsigned d_off[D_PER_THREAD];
  bool d_ok[D_PER_THREAD];
#pragma unroll
  for (int i = 0; i < D_PER_THREAD; ++i) {
    const int idx = t + 256 * i;
    const int r = idx / (BN / 4), nq = idx - r * (BN / 4);
    d_m[i] = mbeg + r;
    d_ok[i] = (idx < D_TOTAL) && (n0 + nq * 4 < p.Nout);
    d_off[i] = (unsigned)((((long long)mbeg + r) * p.lddy + n0 + nq * 4) * db);
  }
  const unsigned d_step = (unsigned)(BKM * p.lddy * db);

  float4 rx[PF][X_PER_THREAD], rs[PF][X_PER_THREAD], rd[PF][D_PER_THREAD];
  auto load_chunk = [&](float4* rx_, float4* rs_, float4* rd_) {   // next 32 pixel rows -> registers, then advance
#pragma unroll
    for (int i = 0; i < X_PER_THREAD; ++i) {
      const bool ok = x_cok & (x_m[i] < mend) & ((unsigned)(x_h[i] + dh) < (unsigned)p.H) & ((unsigned)(x_w[i] + dw) < (unsigned)p.W);
      if (BF && xbf) rx_[i] = buf_ld4_bf16(rX, ok ? x_off[i] : kOob);
      else rx_[i] = buf_ld4(rX, ok ? x_off[i] : kOob);
      if (SC) rs_[i] = buf_ld4(rS, ok ? (unsigned)((x_n[i] * p.C + l_c) * 4) : kOob);
      x_m[i] += BKM;
      x_off[i] += x_step;
      x_w[i] += adv_w;
      x_h[i] += adv_h;
      if (x_w[i] >= p.W) {
        x_w[i] -= p.W;
        ++x_h[i];
      }
      if (x_h[i] >= p.H) {   // crossed into the next image(s)
        const int k = x_h[i] / p.H;
        x_h[i] -= k * p.H;
        x_n[i] += k;
      }
    }
#pragma unroll
    for (int i = 0; i < D_PER_THREAD; ++i) {
      if (BF && dbf) rd_[i] = buf_ld4_bf16(rD, (d_ok[i] & (d_m[i] < mend)) ? d_off[i] : kOob);
      else rd_[i] = buf_ld4(rD, (d_ok[i] & (d_m[i] < mend)) ? d_off[i] : kOob);
      d_m[i] += BKM;
      d_off[i] += d_step;
    }
  };
  auto store_chunk = [&](float* buf, const float4* rx_, const float4* rs_, const float4* rd_) {
    float* smX = buf;
    float* smD = buf + BKM * LDX;
#pragma unroll
    for (int i = 0; i < X_PER_THREAD; ++i) {
      float4 v = rx_[i];
      if (SC) v = f4mul(v, rs_[i]);
      st4(smX + (x_r0 + X_RSTEP * i) * LDX + x_cq * 4, v);
    }
#pragma unroll
    for (int i = 0; i < D_PER_THREAD; ++i) {
      const int idx = t + 256 * i;
      if (idx < D_TOTAL) {
        const int r = idx / (BN / 4), nq = idx - r * (BN / 4);
        st4(smD + r * LDD + nq * 4, rd_[i]);
      }
    }
  };

  f32x4 acc[TMF][NT];
#pragma unroll
  for (int i = 0; i < TMF; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  load_chunk(rx[0], rs[0], rd[0]);
  store_chunk(sm, rx[0], rs[0], rd[0]);
  if (PF == 2) load_chunk(rx[PF - 1], rs[PF - 1], rd[PF - 1]);
  __syncthreads();
  int cur = 0;
  const bool wave_live = mt ? wave * TMF * 16 < p.ntaps * p.C : ci0 + wave * TMF * 16 < p.C;
  auto step = [&](auto UC) {
    constexpr int U = decltype(UC)::value;
    constexpr int NX = (U + 1) % PF;
    load_chunk(rx[U], rs[U], rd[U]);   // PF chunks ahead (rows beyond mend come back as zeros)
    const float* smX = sm + cur * BUF_FLOATS;
    const float* smD = smX + BKM * LDX;
    // a wave whose 16 * TMF input channels all lie beyond C (the tail block of Cin = 136 = 2 * 64 + 8 keeps one wave in four busy; the
    // 16..40-channel expand convs one or two) skips the fragment reads and MFMAs and leaves the matrix pipe to the co-resident block
    if (!wave_live) {
    } else if constexpr (BF) {   // bf16 operands: lane group g feeds pixels g, 4 + g, ..., 28 + g of the chunk to one 16x16x32 MFMA per tile
      bf16x8 a8[TMF], b8[NT];
#pragma unroll
      for (int kk = 0; kk < BKM / 4; ++kk) {
        const int mrow = kk * 4 + g;
#pragma unroll
        for (int i = 0; i < TMF; ++i) a8[i][kk] = (__bf16)smX[mrow * LDX + (wave * TMF + i) * 16 + l15];
#pragma unroll
        for (int j = 0; j < NT; ++j) b8[j][kk] = (__bf16)smD[mrow * LDD + j * 16 + l15];
      }
#pragma unroll
      for (int i = 0; i < TMF; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8[i], b8[j], acc[i][j], 0, 0, 0);
    } else
#pragma unroll
    for (int kk = 0; kk < BKM / 4; ++kk) {
      const int mrow = kk * 4 + g;
      float a[TMF], b[NT];
#pragma unroll
      for (int i = 0; i < TMF; ++i) a[i] = smX[mrow * LDX + (wave * TMF + i) * 16 + l15];
#pragma unroll
      for (int j = 0; j < NT; ++j) b[j] = smD[mrow * LDD + j * 16 + l15];
#pragma unroll
      for (int i = 0; i < TMF; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    store_chunk(sm + (cur ^ 1) * BUF_FLOATS, rx[NX], rs[NX], rd[NX]);
    cur ^= 1;
    __syncthreads();
  };
  typedef std::integral_constant<int, 0> U0;
  typedef std::integral_constant<int, 1 % PF> U1;
  const int nchunks = (mend - mbeg + BKM - 1) / BKM;
  int it = 0;
  for (; it + 4 <= nchunks; it += 4) {
    step(U0{});
    step(U1{});
    step(U0{});
    step(U1{});
  }
  if (it < nchunks) step(U0{});
  if (it + 1 < nchunks) step(U1{});
  if (it + 2 < nchunks) step(U0{});

  const long long Ktot = (long long)p.ntaps * p.C;
#pragma unroll
  for (int i = 0; i < TMF; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ci = ci0 + (wave * TMF + i) * 16 + g * 4 + r;   // multitap: the flattened (tap, channel) row, tap == 0 in the index below
      if (ci >= (mt ? p.ntaps * p.C : p.C)) continue;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = n0 + j * 16 + l15;
        if (n < p.Nout) p.partial[((long long)bz * Ktot + (long long)tap * p.C + ci) * p.Nout + n] = acc[i][j][r];
      }
    }
}

template <int TMF, int NT, bool SC, bool BF>
__global__ __launch_bounds__(256, 2) void conv_filter_grad2_k(FilterGradParams p) {
  __shared__ __attribute__((aligned(16))) float sm[2 * FilterSm<TMF, NT>::BUF_FLOATS];
  conv_filter_grad2_body<TMF, NT, SC, BF>(p, sm, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Several filter-gradient problems of ONE instantiation as one grid (the weight gradients are off the critical path of the backward
// pass: the learner defers the 1x1 MBConv ones to the end and issues them together, so the 14x14 layers' launches -- each far too
// small for the chip -- share it).  desc: device int64 [nprob][16] rows
//   {x, dy, x_scale (0: none), slabs, ldx, lddy, Nimg, H, W, Cin, Cout, ksize | x_bf16 << 8 | dy_bf16 << 9, dil, rows_per_split | multitap << 32,
//    gx | gy << 20 | gz << 40, first block of the problem in this grid}
// (the plan fields from mliis_conv2d_bwd_filter_plan); grid = sum of gx * gy * gz.
constexpr int kFilterDescWords = 16;
template <int TMF, int NT, bool SC, bool BF>
__global__ __launch_bounds__(256, 2) void conv_filter_grad2_batched_k(const long long* __restrict__ desc, int nprob) {
  __shared__ __attribute__((aligned(16))) float sm[2 * FilterSm<TMF, NT>::BUF_FLOATS];
  const int b = blockIdx.x;
  int j = 0;
  for (int k = 1; k < nprob; ++k)
    if (b >= (int)desc[k * kFilterDescWords + 15]) j = k;
  const long long* d = desc + (long long)j * kFilterDescWords;
  const int ks = (int)(d[11] & 0xff);   // (bits 8 / 9: X / dY stored as bf16)
  const FilterGradParams p{reinterpret_cast<const float*>(d[0]), (int)d[4], (int)d[6], (int)d[7], (int)d[8], (int)d[9], ks * ks, (int)d[12],
                           reinterpret_cast<const float*>(d[1]), (int)d[5], (int)d[10], reinterpret_cast<float*>(d[3]),
                           (int)(d[13] & 0xffffffffLL), reinterpret_cast<const float*>(d[2]), (int)(d[13] >> 32),
                           (int)((d[11] >> 8) & 1), (int)((d[11] >> 9) & 1)};
  const int gx = (int)(d[14] & 0xfffff), gy = (int)((d[14] >> 20) & 0xfffff);
  const int local = b - (int)d[15];
  const int bx = local % gx, r = local / gx;
  conv_filter_grad2_body<TMF, NT, SC, BF>(p, sm, bx, r % gy, r / gy);
}

// ------------------------------------------------------------------------------------------------ plans + instantiation switches
struct GemmPlan {
  int tm, nt, gx, gy, gz, chunks_per_split;
  // data-parallel + stream-K remainder (conv_gemm_sk_k): sk_parts > 0
  int sk_full, sk_rem, sk_parts, sk_ipp, sk_nchunks, sk_smax;
  size_t sk_slab_floats() const { return sk_parts > 0 ? (size_t)sk_rem * sk_smax * 64 * (16 * nt) : 0; }
};

// 3x3 convs over fewer than 32 channels: a 32-wide K chunk spans several taps (conv_gemm_nk_k<..., NARROW = true>)
static inline bool gemm_narrow(int ntaps, int C) { return ntaps > 1 && C < 32; }

template <int PREC>
static void launch_gemm_sk_t(const GemmPlan& g, const ConvGemmParams& p, float* slab, hipStream_t stream) {
  const SkPlan k{g.sk_full, g.sk_rem, g.sk_parts, g.sk_ipp, g.sk_nchunks, g.sk_smax, g.gy, slab};
  dim3 grid(g.sk_full + g.sk_parts), block(256);
#ifndef GEMM_SK_PF
#define GEMM_SK_PF 2   // chunks of global loads in flight per thread in the stream-K kernel (3 measured: see profiles/r04_notes.md)
#endif
#define SKL(NT_)                                                                                       \
  hipLaunchKernelGGL((conv_gemm_sk_k<NT_, GEMM_SK_PF, false, PREC>), grid, block, 0, stream, p, k);   \
  hipLaunchKernelGGL((sk_fixup_k<NT_>), dim3(g.sk_rem), dim3(1024), 0, stream, p, k);                  \
  break;
  switch (g.nt) {
    case 1: SKL(1)
    case 2: SKL(2)
    case 3: SKL(3)
    case 4: SKL(4)
    case 5: SKL(5)
    case 6: SKL(6)
    case 7: SKL(7)
    default: SKL(8)
  }
#undef SKL
}

template <int PREC>
static void launch_gemm_t(const GemmPlan& g, const ConvGemmParams& p, hipStream_t stream) {
  dim3 grid(g.gx, g.gy, g.gz), block(256);
  const bool sp = p.partial != nullptr, sc = p.a_scale != nullptr, narrow = gemm_narrow(p.ntaps, p.C);
#define NK(TM_, NT_, SC_, SP_) hipLaunchKernelGGL((conv_gemm_nk_k<TM_, NT_, (TM_ == 1 ? 2 : 1), SC_, SP_, false, PREC>), grid, block, 0, stream, p)
#define L(TM_, NT_)                                  \
  if (TM_ == 1 && narrow) {                          \
    hipLaunchKernelGGL((conv_gemm_nk_k<1, NT_, 2, false, false, true, PREC>), grid, block, 0, stream, p); \
  } else if (TM_ == 1 && sp) {                       \
    if (sc) NK(1, NT_, true, true);                  \
    else NK(1, NT_, false, true);                    \
  } else if (sc) NK(TM_, NT_, true, false);          \
  else NK(TM_, NT_, false, false);                   \
  break;
#define ROW(TM_)       \
  switch (g.nt) {      \
    case 1: L(TM_, 1)  \
    case 2: L(TM_, 2)  \
    case 3: L(TM_, 3)  \
    case 4: L(TM_, 4)  \
    case 5: L(TM_, 5)  \
    case 6: L(TM_, 6)  \
    case 7: L(TM_, 7)  \
    default: L(TM_, 8) \
  }
  if (g.tm == 2) {
    ROW(2)
  } else {
    ROW(1)
  }
#undef ROW
#undef L
#undef NK
}

// conv1x1_ksplit_k instances: KC 16-wide K groups per wave (8 waves: K <= 128 KC), NT column tiles, KC * NT <= 8, NT <= 7; and the
// three instances with KC * NT = 12 that the planner takes on the small maps (one workgroup per CU there: 256 registers)
template <int PREC>
static bool launch_ksplit_t(int kc, int nt, dim3 grid, const ConvGemmParams& p, int row_groups, hipStream_t stream) {
  dim3 block(512);
#define S(KC_, NT_) hipLaunchKernelGGL((conv1x1_ksplit_k<KC_, NT_, 8, PREC>), grid, block, 0, stream, p, row_groups); break;
  switch (kc) {
    case 1: switch (nt) { case 1: S(1, 1) case 2: S(1, 2) case 3: S(1, 3) case 4: S(1, 4) case 5: S(1, 5) case 6: S(1, 6) case 7: S(1, 7) default: return false; } break;
    case 2: switch (nt) { case 1: S(2, 1) case 2: S(2, 2) case 3: S(2, 3) case 4: S(2, 4) default: return false; } break;
    case 3: switch (nt) { case 1: S(3, 1) case 2: S(3, 2) default: return false; } break;
    case 4: switch (nt) { case 1: S(4, 1) case 2: S(4, 2) case 3: S(4, 3) default: return false; } break;
    case 5: switch (nt) { case 1: S(5, 1) case 2: S(5, 2) default: return false; } break;
    case 6: switch (nt) { case 1: S(6, 1) case 2: S(6, 2) default: return false; } break;
    case 7: switch (nt) { case 1: S(7, 1) default: return false; } break;
    default: return false;
  }
#undef S
  return true;
}

struct FilterPlan {
  int tmf, nt, gx, gy, gz, rows_per_split, multitap;
};

template <bool BF>
static bool launch_filter_batched_t(int tmf, int nt, bool sc, const long long* desc, int nprob, int blocks, hipStream_t stream) {
  dim3 grid(blocks), block(256);
#define L(T_, NT_)                                                                                                            \
  if (sc) hipLaunchKernelGGL((conv_filter_grad2_batched_k<T_, NT_, true, BF>), grid, block, 0, stream, desc, nprob);          \
  else hipLaunchKernelGGL((conv_filter_grad2_batched_k<T_, NT_, false, BF>), grid, block, 0, stream, desc, nprob);            \
  break;
#define ROW(T_)       \
  switch (nt) {       \
    case 1: L(T_, 1)  \
    case 2: L(T_, 2)  \
    case 3: L(T_, 3)  \
    case 4: L(T_, 4)  \
    case 5: L(T_, 5)  \
    case 6: L(T_, 6)  \
    case 7: L(T_, 7)  \
    case 8: L(T_, 8)  \
    default: return false; \
  }
  if (tmf == 2) {
    ROW(2)
  } else if (tmf == 1) {
    ROW(1)
  } else {
    return false;
  }
#undef ROW
#undef L
  return true;
}

template <bool BF>
static void launch_filter_t(const FilterPlan& f, const FilterGradParams& p, hipStream_t stream) {
  dim3 grid(f.gx, f.gy, f.gz), block(256);
#define L(T_, NT_)                                                                                                   \
  if (p.x_scale) hipLaunchKernelGGL((conv_filter_grad2_k<T_, NT_, true, BF>), grid, block, 0, stream, p);           \
  else hipLaunchKernelGGL((conv_filter_grad2_k<T_, NT_, false, BF>), grid, block, 0, stream, p);                         \
  break;
#define ROW(T_)       \
  switch (f.nt) {     \
    case 1: L(T_, 1)  \
    case 2: L(T_, 2)  \
    case 3: L(T_, 3)  \
    case 4: L(T_, 4)  \
    case 5: L(T_, 5)  \
    case 6: L(T_, 6)  \
    case 7: L(T_, 7)  \
    default: L(T_, 8) \
  }
  if (f.tmf == 2) {
    ROW(2)
  } else {
    ROW(1)
  }
#undef ROW
#undef L
}

// operand precision of the matrix cores: the fp32 instances live in conv_gemm.hip, the bf16 ones in conv_gemm_bf16.hip
// ------------------------------------------------------------------------------------------------ split-product weight images (conv_x3.hip)
// One 128-thread block of the pack: see conv_x3.hip for the image layout.  Shared with conv_gemm.hip, whose weight-shadow launch
// (mliis_weight_shadows) builds the K-contiguous weight copies and these images in ONE grid.
constexpr int kX3Block = 3072;   // bytes of one (chunk, 16-column tile) block of a weight image
constexpr int kX3DescWords = 8;
__device__ __forceinline__ void x3_pack_block(const float* __restrict__ theta, char* __restrict__ images, const long long* __restrict__ desc,
                                              int ndesc, int b, int t) {
  int j = 0;
  for (int k = 1; k < ndesc; ++k)
    if (b >= (int)(desc[k * kX3DescWords + 6] >> 8)) j = k;
  const long long* d = desc + (long long)j * kX3DescWords;
  const float* w = theta + d[0];
  const int taps = (int)d[1], cin_total = (int)d[2], cout = (int)d[3], ci_begin = (int)d[4], cin = (int)d[5], mode = (int)(d[6] & 0xff);
  const int local = b - (int)(d[6] >> 8);
  const int nn = mode == 0 ? cout : cin, kc = mode == 0 ? cin : cout;   // columns of B, K per tap
  const int ncol16 = (nn + 15) / 16;
  const int chunk = local / ncol16, c16 = local - chunk * ncol16;
  const int nl = t & 15, kq = t >> 4;   // column of the tile, k quad 0..7 of the chunk
  const int n = c16 * 16 + nl;
  // K = the flattened (tap, channel) index, tap-major, in chunks of 32 without padding between taps (kc % 4 == 0: a k quad belongs to one
  // tap; only the last chunk of the image is zero-padded) -- conv_x3_tile advances the same index per lane
  const int k = chunk * 32 + kq * 4;
  const int tap = k / kc, c = k - tap * kc;
  float4 v = f4zero();
  if (n < nn && tap < taps) {
    if (mode == 0) {
      const float* s = w + ((long long)tap * cin_total + ci_begin + c) * cout + n;
      v = make_float4(s[0], s[cout], s[2 * (long long)cout], s[3 * (long long)cout]);
    } else {
      v = ld4(w + ((long long)tap * cin_total + ci_begin + n) * cout + c);
    }
  }
  uint2 h, m, l;
  split3(v, h, m, l);
  // lane group g = kq & 3 holds k = 4g..4g+3 (elements 0..3) and 16 + 4g..16 + 4g + 3 (elements 4..7) of the chunk
  char* dst = images + d[7] + (long long)local * kX3Block + (kq & 3) * 256 + nl * 16 + (kq >> 2) * 8;
  *reinterpret_cast<uint2*>(dst) = h;
  *reinterpret_cast<uint2*>(dst + 1024) = m;
  *reinterpret_cast<uint2*>(dst + 2048) = l;
}


void launch_gemm_bf16(const GemmPlan& g, const ConvGemmParams& p, hipStream_t stream);
void launch_gemm_sk_bf16(const GemmPlan& g, const ConvGemmParams& p, float* slab, hipStream_t stream);
void launch_gemm_fp8(const GemmPlan& g, const ConvGemmParams& p, hipStream_t stream);            // conv_gemm_fp8.hip
bool launch_stream_lowp(int precision, int kc, int nt, dim3 grid, const ConvGemmParams& p, int row_groups, hipStream_t stream, bool ain = false);   // conv_gemm_fp8.hip
bool launch_ksplit_lowp(int precision, int kc, int nt, dim3 grid, const ConvGemmParams& p, int row_groups, hipStream_t stream);   // conv_gemm_fp8.hip
void launch_filter_bf16(const FilterPlan& f, const FilterGradParams& p, hipStream_t stream);
bool launch_filter_batched_bf16(int tmf, int nt, bool sc, const long long* desc, int nprob, int blocks, hipStream_t stream);
bool launch_filter_batched_x3(int nt, const long long* desc, int nprob, int blocks, int tile_ci, hipStream_t stream);   // conv_x3.hip: the (TMF = 2, NT >= 4, no x_scale) groups as split products (tile_ci: 128 | 256 input channels per workgroup tile)

}  // namespace mliis
