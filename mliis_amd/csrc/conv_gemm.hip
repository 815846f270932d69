// Dense convolutions as implicit GEMMs on the fp32 matrix cores (v_mfma_f32_16x16x4_f32), NHWC activations, TF HWIO weights.
// One kernel family serves
//   * MBConv 1x1 expand/project convs           (models/efficientnet/efficientnet_model.py:175-182,225-232)
//   * RSD decoder 1x1 / 3x3 / 3x3-dilated convs (models/efficientlab.py:185-190,218-224) and optional ASPP (:248-289)
// in forward, backward-data and backward-filter.  Both GEMM operands are always K-contiguous ("NK"): backward-data reads the
// HWIO buffer itself as B^T with the tap offset negated; forward reads the HWOI shadow copy that mliis_transpose_weights
// refreshes once per inner step (2 % of a step, and it buys the 16-byte fragment path for B).
//
// Tiling (wave = 64 lanes, 4 waves per workgroup):
//   forward / bwd-data (conv_gemm_nk_k): block tile (64*TM) x (16*NT) outputs; wave w owns rows [16*TM*w, 16*TM*(w+1)).
//     K is the flattened (tap, channel) index cut into chunks of 32.  A and B tiles sit in LDS as [kq = k/4][row ^ kq][4]:
//     16-byte fragments, conflict-free ds_write_b128 (8 lanes = 8 rows ^ kq) and conflict-free ds_read_b128 (a 16-lane read group
//     covers 16 distinct rows mod 16).  Inside every 16-wide K group the K order is permuted identically for A and B (lane group g
//     owns k = 4g..4g+3), so one b128 read feeds 4 MFMAs.
//   Planner: 64-row blocks (TM = 1) so two workgroups share a CU and hide each other's LDS/barrier latency (measured: rsd2
//     fuse bwd-data 54 -> 85 TF vs one 128-row block per CU); narrower column tiles / split-K for grids smaller than the chip.
//   bwd-filter (conv_filter_grad2_k): block tile (64*TMF ci) x (16*NT co) for one tap; streams 32 pixels per step; A = X^T,
//     B = dY, both read k-major with ds_read_b32 (row stride == 16 mod 32 banks); per-split slabs, deterministic fold.
// Global loads are raw 16-byte buffer loads along C (128-byte spans per 8 lanes) whose offset is pushed out of range for padding
// (the hardware returns zeros: no branches), software-pipelined two chunks deep through registers into double-buffered LDS.
#include <stdlib.h>

#include <type_traits>

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
};

// ------------------------------------------------------------------------------------------------ forward / backward-data
// Built so that the matrix pipe is not starved:
//  * global loads are raw buffer loads whose offset is forced out of range for padded / out-of-image / out-of-K elements (the
//    hardware returns zeros) -- no branches, so the address arithmetic of chunk k+1 is scheduled between the MFMAs of chunk k;
//  * every thread advances the (tap, channel) position of its own k quad incrementally, per-row offsets are computed once;
//  * LDS is double buffered: one barrier per K chunk, the LDS writes of chunk k+1 overlap the MFMAs of chunk k of the other waves;
//  * all fragments of a chunk are read from LDS before its first MFMA.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOob = 0xFFFFFFF0u;          // beyond any num_records: the load returns 0
constexpr unsigned kBufRecords = 0x80000000u;   // host guarantees every legal byte offset is below 2 GiB

__device__ __forceinline__ float4 buf_ld4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

template <int TM, int NT, int PF, bool SC, bool SPLIT, bool NARROW>
__global__ __launch_bounds__(256, 2) void conv_gemm_nk_k(ConvGemmParams p) {
  constexpr int BM = 64 * TM, BN = 16 * NT, BK = 32;
  constexpr int A_FLOATS = 8 * BM * 4;
  constexpr int B_FLOATS = 8 * BN * 4;
  constexpr int BUF_FLOATS = A_FLOATS + B_FLOATS;
  constexpr int A_PER_THREAD = 2 * TM;          // float4 per thread per chunk
  constexpr int B_TOTAL = BN * 8;               // float4 per chunk
  constexpr int B_PER_THREAD = (B_TOTAL + 255) / 256;
  constexpr int LDS_STAGE = BN + 4;             // epilogue staging row stride (floats)
  constexpr int STAGE_FLOATS = 4 * 16 * LDS_STAGE;
  constexpr int SM_FLOATS = (2 * BUF_FLOATS) > STAGE_FLOATS ? (2 * BUF_FLOATS) : STAGE_FLOATS;
  constexpr int kDummy = SM_FLOATS;   // 16-byte scratch slot behind the buffers
  __shared__ __attribute__((aligned(16))) float sm[SM_FLOATS + 4];

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const long long M = (long long)p.Nimg * p.H * p.W;
  const int n0 = blockIdx.y * BN;
  // K is the flattened (tap, channel) index cut into chunks of 32: chunks may straddle taps (every thread tracks the tap / channel of
  // ITS k quad), so only the very last chunk carries padding and every chunk is a full 2 x 4 x NT MFMA block -- no per-chunk branch.
  const int nchunks_total = (p.ntaps * p.C + BK - 1) / BK;
  const int it0 = blockIdx.z * p.chunks_per_split;
  int it1 = it0 + p.chunks_per_split;
  if (it1 > nchunks_total) it1 = nchunks_total;

  constexpr bool split = SPLIT;   // split-K instance: raw partial tiles [z][M][Nout], the fold kernel finishes them
  const bool stats = (p.stats_part != nullptr) && !split;
  float s1[NT], s2[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) s1[j] = s2[j] = 0.f;

  const unsigned bx = xcd_remap(blockIdx.x, gridDim.x);
  const long long m0 = (long long)bx * BM;

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
      const int HWp = p.H * p.W;
      const int n = (int)(m / HWp);
      const int rem = (int)(m - (long long)n * HWp);
      a_h[i] = rem / p.W;
      a_w[i] = rem - a_h[i] * p.W;
      a_off[i] = (unsigned)(m * p.lda * 4);
      s_off[i] = (unsigned)((long long)n * p.C * 4);
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
    const unsigned da = (unsigned)(((dh * p.W + dw) * p.lda + k_c) * 4);
#pragma unroll
    for (int i = 0; i < A_PER_THREAD; ++i) {
      const bool ok = kok & ((unsigned)(a_h[i] + dh) < (unsigned)p.H) & ((unsigned)(a_w[i] + dw) < (unsigned)p.W);
      ra_[i] = buf_ld4(rA, ok ? a_off[i] + da : kOob);
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
      const int row = a_r0 + 32 * i;
      float4 v = ra_[i];
      if (SC) v = f4mul(v, rs_[i]);
      st4(smA + (kq * BM + (row ^ kq)) * 4, v);
    }
#pragma unroll
    for (int i = 0; i < B_PER_THREAD; ++i) {
      const int idx = t + 256 * i;
      if (256 * (i + 1) <= B_TOTAL) st4(smB + (kq * BN + ((idx >> 3) ^ kq)) * 4, rb_[i]);
      else st4(idx < B_TOTAL ? smB + (kq * BN + ((idx >> 3) ^ kq)) * 4 : sm + kDummy, rb_[i]);   // surplus lanes: a scratch slot, no branch
    }
  };

  f32x4 acc[TM][NT];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

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
      for (int i = 0; i < TM; ++i) av[q][i] = ld4(smA + (fq * BM + ((wave * 16 * TM + i * 16 + l15) ^ fq)) * 4);
#pragma unroll
      for (int jn = 0; jn < NT; ++jn) bv[q][jn] = ld4(smB + (fq * BN + ((jn * 16 + l15) ^ fq)) * 4);
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
    if (kAllFirst) {
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
  int it = it0;
  // Main loop: four chunks per trip.  The compiler flushes the load counter at a loop header (it cannot prove across the back edge
  // that only the newest chunk is in flight), so the long body keeps that flush to every fourth chunk; the tail is straight-line.
  for (; it + 4 <= it1; it += 4) {
    step(U0{}, it);
    step(U1{}, it + 1);
    step(U0{}, it + 2);
    step(U1{}, it + 3);
  }
  if (it < it1) step(U0{}, it);
  if (it + 1 < it1) step(U1{}, it + 1);
  if (it + 2 < it1) step(U0{}, it + 2);

  // ---- epilogue: C/D layout of 16x16x4: col = lane & 15, row = 4 * (lane >> 4) + reg
  if (split) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const long long m = m0 + wave * 16 * TM + i * 16 + g * 4 + r;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int n = n0 + j * 16 + l15;
          if (n < p.Nout) p.partial[((long long)blockIdx.z * M + m) * p.Nout + n] = acc[i][j][r];
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
        float* dst = p.Cmat + m * p.ldc + n;
        if (p.accumulate) v = f4add(v, ld4(dst));
        st4(dst, v);
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

__global__ __launch_bounds__(256) void splitk_reduce_k(const float* __restrict__ partial, int splits, long long M, int Nout,
                                                       float* __restrict__ Cmat, int ldc, const float* __restrict__ bias,
                                                       int accumulate, const float* __restrict__ border_bias, int H, int W) {
  const int Q = Nout >> 2;
  const long long total = M * Q;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long m = i / Q;
    const int n = (int)(i - m * Q) << 2;
    float4 s = ld4(partial + m * Nout + n);
    for (int z = 1; z < splits; ++z) s = f4add(s, ld4(partial + ((long long)z * M + m) * Nout + n));
    if (bias != nullptr) s = f4add(s, ld4(bias + n));
    if (border_bias != nullptr) {
      const long long HWp = (long long)H * W;
      const int ni = (int)(m / HWp);
      const int rem = (int)(m - (long long)ni * HWp);
      const int h = rem / W, w_ = rem - h * W;
      const int cls = (h == 0 ? 0 : (h == H - 1 ? 2 : 1)) * 3 + (w_ == 0 ? 0 : (w_ == W - 1 ? 2 : 1));
      s = f4add(s, ld4(border_bias + ((long long)ni * 9 + cls) * Nout + n));
    }
    float* dst = Cmat + m * ldc + n;
    if (accumulate) s = f4add(s, ld4(dst));
    st4(dst, s);
  }
}

// Split-K fold that also emits the stage-1 BN statistics of the finished values (what the GEMM epilogue does when gz == 1), so a
// split-K layer needs no separate statistics launch.  grid = (row chunks, ceil(Nout/32)); 256 threads = 8 channel quads x 32 row
// lanes; stats_part layout [row chunk][2][Nout] ({sum v, sum v^2}, of swish(v) when stats_swish).
constexpr int kRedRows = 128;   // rows per block: 4 per thread, all slab loads of a row issued together

__global__ __launch_bounds__(256) void splitk_reduce_stats_k(const float* __restrict__ partial, int splits, int M, int Nout,
                                                             float* __restrict__ Cmat, int ldc, const float* __restrict__ bias,
                                                             const float* __restrict__ border_bias, int H, int W,
                                                             float* __restrict__ stats_part, int stats_swish) {
  __shared__ float4 sm[2][4][8];
  const int t = threadIdx.x, q = t & 7, rl = t >> 3;
  const int n = blockIdx.y * 32 + q * 4;
  float4 s1 = f4zero(), s2 = f4zero();
  if (n < Nout) {
    const float4 bv = bias != nullptr ? ld4(bias + n) : f4zero();
    const int r0 = blockIdx.x * kRedRows;
    const int r1 = r0 + kRedRows < M ? r0 + kRedRows : M;
    for (int m = r0 + rl; m < r1; m += 32) {
      float4 s = ld4(partial + (long long)m * Nout + n);
      for (int z = 1; z < splits; ++z) s = f4add(s, ld4(partial + ((long long)z * M + m) * Nout + n));
      s = f4add(s, bv);
      if (border_bias != nullptr) {
        const int HWp = H * W;
        const int ni = m / HWp;
        const int rem = m - ni * HWp;
        const int h = rem / W, w_ = rem - h * W;
        const int cls = (h == 0 ? 0 : (h == H - 1 ? 2 : 1)) * 3 + (w_ == 0 ? 0 : (w_ == W - 1 ? 2 : 1));
        s = f4add(s, ld4(border_bias + ((long long)ni * 9 + cls) * Nout + n));
      }
      st4(Cmat + (long long)m * ldc + n, s);
      if (stats_swish) s = make_float4(swish_f(s.x), swish_f(s.y), swish_f(s.z), swish_f(s.w));
      s1 = f4add(s1, s);
      s2 = f4add(s2, f4mul(s, s));
    }
  }
#pragma unroll
  for (int off = 8; off < 64; off <<= 1) {
    s1.x += __shfl_xor(s1.x, off); s1.y += __shfl_xor(s1.y, off); s1.z += __shfl_xor(s1.z, off); s1.w += __shfl_xor(s1.w, off);
    s2.x += __shfl_xor(s2.x, off); s2.y += __shfl_xor(s2.y, off); s2.z += __shfl_xor(s2.z, off); s2.w += __shfl_xor(s2.w, off);
  }
  if ((t & 63) < 8) {
    sm[0][t >> 6][q] = s1;
    sm[1][t >> 6][q] = s2;
  }
  __syncthreads();
  if (t < 16) {
    const int v = t >> 3, qq = t & 7;
    const int nn = blockIdx.y * 32 + qq * 4;
    if (nn < Nout)
      st4(stats_part + ((long long)blockIdx.x * 2 + v) * Nout + nn,
          f4add(f4add(sm[v][0][qq], sm[v][1][qq]), f4add(sm[v][2][qq], sm[v][3][qq])));
  }
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
};

// ------------------------------------------------------------------------------------------------ backward-filter kernel
// The pipeline of conv_gemm_nk_k applied to the pixel reduction: branch-free buffer loads
// (out-of-range rows and halo pixels return zeros), per-row (h, w) advanced incrementally instead of two integer divisions per row
// and chunk, double-buffered LDS (one barrier per 32-pixel chunk) and a two-chunk register prefetch.
template <int TMF, int NT, bool SC>
__global__ __launch_bounds__(256, 2) void conv_filter_grad2_k(FilterGradParams p) {
  constexpr int BCI = 64 * TMF, BN = 16 * NT, BKM = 32;
  constexpr int LDX = BCI + 16;
  constexpr int LDD = (BN % 32 == 0) ? BN + 16 : BN;
  constexpr int X_PER_THREAD = (BKM * (BCI / 4)) / 256;  // 2 * TMF
  constexpr int D_TOTAL = BKM * (BN / 4);
  constexpr int D_PER_THREAD = (D_TOTAL + 255) / 256;
  constexpr int X_RSTEP = 256 / (BCI / 4);
  constexpr int BUF_FLOATS = BKM * LDX + BKM * LDD;
  constexpr int PF = (TMF == 2 && (SC || NT >= 8)) ? 1 : 2;   // register budget: one staging set for the widest instances
  __shared__ __attribute__((aligned(16))) float sm[2 * BUF_FLOATS];

  const int t = threadIdx.x;
  const int lane = t & 63, wave = t >> 6;
  const int l15 = lane & 15, g = lane >> 4;
  const int M = p.Nimg * p.H * p.W;            // host guarantees < 2^31
  const int cblocks = (p.C + BCI - 1) / BCI;
  const int tap = blockIdx.x / cblocks;
  const int ci0 = (blockIdx.x - tap * cblocks) * BCI;
  const int n0 = blockIdx.y * BN;
  const int mbeg = blockIdx.z * p.rows_per_split;
  int mend = mbeg + p.rows_per_split;
  if (mend > M) mend = M;
  int dh = 0, dw = 0;
  if (p.ntaps > 1) {
    dh = (tap / 3 - 1) * p.dil;
    dw = (tap % 3 - 1) * p.dil;
  }
  const int HW = p.H * p.W;
  const int adv_h = BKM / p.W, adv_w = BKM - adv_h * p.W;   // one chunk = 32 pixels further along the flattened (n, h, w) index

  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, kBufRecords, 0x00020000);
  const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc((void*)p.dY, 0, kBufRecords, 0x00020000);
  const __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void*)(SC ? p.x_scale : p.X), 0, kBufRecords, 0x00020000);

  // ---- per-thread X rows: position of the row the NEXT load will fetch
  const int x_cq = t % (BCI / 4);
  const int x_r0 = t / (BCI / 4);
  const bool x_cok = ci0 + x_cq * 4 < p.C;
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
    x_off[i] = (unsigned)((((long long)m + (long long)dh * p.W + dw) * p.ldx + ci0 + x_cq * 4) * 4);
  }
  const unsigned x_step = (unsigned)(BKM * p.ldx * 4);
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
    d_off[i] = (unsigned)((((long long)mbeg + r) * p.lddy + n0 + nq * 4) * 4);
  }
  const unsigned d_step = (unsigned)(BKM * p.lddy * 4);

  float4 rx[PF][X_PER_THREAD], rs[PF][X_PER_THREAD], rd[PF][D_PER_THREAD];
  auto load_chunk = [&](float4* rx_, float4* rs_, float4* rd_) {   // next 32 pixel rows -> registers, then advance
#pragma unroll
    for (int i = 0; i < X_PER_THREAD; ++i) {
      const bool ok = x_cok & (x_m[i] < mend) & ((unsigned)(x_h[i] + dh) < (unsigned)p.H) & ((unsigned)(x_w[i] + dw) < (unsigned)p.W);
      rx_[i] = buf_ld4(rX, ok ? x_off[i] : kOob);
      if (SC) rs_[i] = buf_ld4(rS, ok ? (unsigned)((x_n[i] * p.C + ci0 + x_cq * 4) * 4) : kOob);
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
      rd_[i] = buf_ld4(rD, (d_ok[i] & (d_m[i] < mend)) ? d_off[i] : kOob);
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
  auto step = [&](auto UC) {
    constexpr int U = decltype(UC)::value;
    constexpr int NX = (U + 1) % PF;
    load_chunk(rx[U], rs[U], rd[U]);   // PF chunks ahead (rows beyond mend come back as zeros)
    const float* smX = sm + cur * BUF_FLOATS;
    const float* smD = smX + BKM * LDX;
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
      const int ci = ci0 + (wave * TMF + i) * 16 + g * 4 + r;
      if (ci >= p.C) continue;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = n0 + j * 16 + l15;
        if (n < p.Nout) p.partial[((long long)blockIdx.z * Ktot + (long long)tap * p.C + ci) * p.Nout + n] = acc[i][j][r];
      }
    }
}

// ------------------------------------------------------------------------------------------------ weight shadow (HWIO -> HWOI)
// dst[off + tap*Ci*Co + co*Ci + ci] = src[off + tap*Ci*Co + ci*Co + co] for every descriptor (off, taps, Ci, Co): one launch per
// inner step keeps a K-contiguous copy of all dense-conv weights so the FORWARD GEMM can use the same b128-fragment B path as
// backward-data.  32x32 LDS tiles: coalesced reads along co, coalesced writes along ci.
__global__ __launch_bounds__(256) void transpose_weights_k(const float* __restrict__ src, float* __restrict__ dst,
                                                           const int* __restrict__ desc) {
  __shared__ float tile[32][33];
  const int d = blockIdx.y;
  const int off = desc[4 * d + 0], taps = desc[4 * d + 1], Ci = desc[4 * d + 2], Co = desc[4 * d + 3];
  const int tci = (Ci + 31) / 32, tco = (Co + 31) / 32;
  const int ntiles = taps * tci * tco;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int tIdx = blockIdx.x; tIdx < ntiles; tIdx += gridDim.x) {
    const int tap = tIdx / (tci * tco);
    const int rem = tIdx - tap * tci * tco;
    const int ci0 = (rem / tco) * 32, co0 = (rem % tco) * 32;
    const float* s0 = src + off + (long long)tap * Ci * Co;
    float* d0 = dst + off + (long long)tap * Ci * Co;
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
      const int ci = ci0 + r, co = co0 + tx;
      tile[r][tx] = (ci < Ci && co < Co) ? s0[(long long)ci * Co + co] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
      const int co = co0 + r, ci = ci0 + tx;
      if (co < Co && ci < Ci) d0[(long long)co * Ci + ci] = tile[tx][r];
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------ host-side planning
struct GemmPlan {
  int tm, nt, gx, gy, gz, chunks_per_split;
};

// Column tiles per block (16 columns each, at most 8): the widest tile that does not pad the output width by much -- padded
// columns are wasted MFMAs (N = 136: 3 x 48 columns instead of 2 x 80), narrower tiles re-read the A tile more often.
static inline int pick_nt(int Nout) {
  int best = 1;
  double best_cost = 1e30;
  const int tiles = (Nout + 15) / 16;
  for (int nt = 1; nt <= 8; ++nt) {
    const int blocks = (tiles + nt - 1) / nt;
    const double cost = (double)(blocks * nt * 16) / (double)Nout * (1.0 + 0.03 * (8 - nt));
    if (cost < best_cost - 1e-9 || (cost < best_cost + 1e-9 && nt > best)) {
      best_cost = cost;
      best = nt;
    }
  }
  return best;
}

constexpr int kGemmFill = 4;     // fwd / bwd-data: narrow the column tiles of sub-chip grids until there are this many blocks per CU
constexpr int kFilterFill = 1;   // bwd-filter: same idea, one block per CU is enough (its slabs already split the pixel axis)

// 3x3 convs over fewer than 32 channels: a 32-wide K chunk spans several taps (conv_gemm_nk_k<..., NARROW = true>)
static inline bool gemm_narrow(int ntaps, int C) { return ntaps > 1 && C < 32; }

static inline GemmPlan plan_gemm(long long M, int Nout, int C, int ntaps, int num_cus, int allow_split) {
  GemmPlan g;
  g.nt = pick_nt(Nout);
  g.gy = (Nout + g.nt * 16 - 1) / (g.nt * 16);
  // measured: two co-resident 64-row blocks per CU hide LDS/barrier latency better than one 128-row block; only very tall, HBM-bound
  // layers (112x112 maps) take 128-row blocks, which halve the number of BN-statistics partials the consumer folds
  g.tm = (M + 63) / 64 > 4LL * num_cus ? 2 : 1;
  if (gemm_narrow(ntaps, C)) {   // (not a hot-path shape: one plain instance per column-tile width)
    g.tm = 1;
    allow_split = 0;
  }
  g.gx = (int)((M + 64 * g.tm - 1) / (64 * g.tm));
  // small grids (14x14 / 28x28 maps) are latency-bound: narrower column tiles (down to 32 columns) until the grid holds
  // kGemmFill blocks per CU -> a shorter MFMA chain per block and often no split-K pass at all (measured +3 % on the whole step)
  if (g.tm == 1 && g.gx < num_cus)
    while (g.nt > 2 && (long long)g.gx * g.gy < (long long)kGemmFill * num_cus) {
      g.nt = (g.nt + 1) / 2;
      g.gy = (Nout + g.nt * 16 - 1) / (g.nt * 16);
    }
  const int nchunks = (ntaps * C + 31) / 32;   // K chunks over the flattened (tap, channel) index
  g.gz = 1;
  if (allow_split && g.tm == 1) {   // (128-row tiles are only chosen for grids that fill the chip many times over)
    long long blocks = (long long)g.gx * g.gy;
    // split K when the grid cannot fill the chip and there is enough K to amortise the extra pass
    while (blocks * g.gz * 2 <= num_cus && nchunks / (g.gz * 2) >= 3 && g.gz < 16) g.gz *= 2;   // (a looser bound measured slower)
  }
  g.chunks_per_split = (nchunks + g.gz - 1) / g.gz;
  g.gz = (nchunks + g.chunks_per_split - 1) / g.chunks_per_split;
  return g;
}

static void launch_gemm(const GemmPlan& g, const ConvGemmParams& p, hipStream_t stream) {
  dim3 grid(g.gx, g.gy, g.gz), block(256);
  const bool sp = p.partial != nullptr, sc = p.a_scale != nullptr, narrow = gemm_narrow(p.ntaps, p.C);
#define NK(TM_, NT_, SC_, SP_) hipLaunchKernelGGL((conv_gemm_nk_k<TM_, NT_, (TM_ == 1 ? 2 : 1), SC_, SP_, false>), grid, block, 0, stream, p)
#define L(TM_, NT_)                                  \
  if (TM_ == 1 && narrow) {                          \
    hipLaunchKernelGGL((conv_gemm_nk_k<1, NT_, 2, false, false, true>), grid, block, 0, stream, p); \
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

struct FilterPlan {
  int tmf, nt, gx, gy, gz, rows_per_split;
};
static inline void plan_filter_split(FilterPlan& f, long long M, int C, int Nout, int ntaps, int num_cus) {
  const int bci = 64 * f.tmf;
  f.gy = (Nout + f.nt * 16 - 1) / (f.nt * 16);
  f.gx = ntaps * ((C + bci - 1) / bci);
  long long base = (long long)f.gx * f.gy;
  long long want = (2LL * num_cus) / base;  // one full round of two co-resident blocks per CU (no half-empty second round) ...
  {  // ... bounded by the slab count the fold has to read: 96 in general, up to 512 for tiny filters (fold cost ~ slabs * size)
    const long long total = (long long)ntaps * C * Nout;
    long long cap = (1LL << 19) / (total > 0 ? total : 1);
    if (cap > 512) cap = 512;
    if (cap < 96) cap = 96;
    if (want > cap) want = cap;
  }
  if (want < 1) want = 1;
  long long rps = (M + want - 1) / want;
  if (rps < 64) rps = 64;   // at least two 32-row steps per slab (measured: 32 is no faster, 128+ slower -- the kernel is latency-bound)
  rps = (rps + 31) / 32 * 32;
  f.rows_per_split = (int)rps;
  f.gz = (int)((M + rps - 1) / rps);
}

static inline FilterPlan plan_filter(long long M, int C, int Nout, int ntaps, int num_cus) {
  FilterPlan f;
  f.nt = pick_nt(Nout);
  // 128-wide ci blocks halve the dY re-reads; take them unless they pad more zero rows than 64-wide blocks would
  const int waste128 = (C + 127) / 128 * 128 - C, waste64 = (C + 63) / 64 * 64 - C;
  f.tmf = (C >= 128 && waste128 <= waste64) ? 2 : 1;
  plan_filter_split(f, M, C, Nout, ntaps, num_cus);
  // small maps cannot be split further along the pixels (64 rows per slab): narrower tiles instead (latency-bound, see plan_gemm)
  while ((long long)f.gx * f.gy * f.gz < (long long)kFilterFill * num_cus && (f.tmf == 2 || f.nt > 2)) {
    if (f.tmf == 2) f.tmf = 1;
    else f.nt = (f.nt + 1) / 2;
    plan_filter_split(f, M, C, Nout, ntaps, num_cus);
  }
  return f;
}

static void launch_filter(const FilterPlan& f, const FilterGradParams& p, hipStream_t stream) {
  dim3 grid(f.gx, f.gy, f.gz), block(256);
#define L(T_, NT_)                                                                                                   \
  if (p.x_scale) hipLaunchKernelGGL((conv_filter_grad2_k<T_, NT_, true>), grid, block, 0, stream, p);           \
  else hipLaunchKernelGGL((conv_filter_grad2_k<T_, NT_, false>), grid, block, 0, stream, p);                         \
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

static int g_num_cus = 0;
static int num_cus() {
  if (g_num_cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      g_num_cus = prop.multiProcessorCount;
    else
      g_num_cus = 256;
  }
  return g_num_cus;
}

static int conv_check(const char* name, int Nimg, int H, int W, int Cin, int Cout, int ksize, int dil) {
  MLIIS_REQUIRE(Nimg > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, MLIIS_ERR_ARG, "%s: bad shape", name);
  MLIIS_REQUIRE((Cin & 3) == 0 && (Cout & 3) == 0, MLIIS_ERR_ARG, "%s: channel counts must be multiples of 4 (Cin=%d Cout=%d)", name,
                Cin, Cout);
  MLIIS_REQUIRE(ksize == 1 || ksize == 3, MLIIS_ERR_UNSUPPORTED, "%s: kernel size %d unsupported (1 or 3)", name, ksize);
  MLIIS_REQUIRE(dil >= 1, MLIIS_ERR_ARG, "%s: dilation must be >= 1", name);
  return MLIIS_OK;
}

}  // namespace mliis

using namespace mliis;

extern "C" {

// Tiling the planner picks for a conv2d_fwd / conv2d_bwd_data call (profiling aid: row-tile factor, column tiles, split-K factor).
int mliis_conv2d_plan(int Nimg, int H, int W, int Cred, int Nout, int ksize, int* tm, int* nt, int* splits) {
  MLIIS_REQUIRE(tm && nt && splits, MLIIS_ERR_ARG, "conv2d_plan: null pointer");
  GemmPlan g = plan_gemm((long long)Nimg * H * W, Nout, Cred, ksize * ksize, num_cus(), 1);
  *tm = g.tm;
  *nt = g.nt;
  *splits = g.gz;
  return MLIIS_OK;
}

// Name of the kernel instantiation (as rocprofv3 prints it, without the mliis:: prefix and argument list) that a conv2d_fwd /
// conv2d_bwd_data call with these shapes launches (has_scale: an x_scale operand is given).
int mliis_conv2d_kernel_name(int Nimg, int H, int W, int Cred, int Nout, int ksize, int has_scale, char* buf, size_t buf_len) {
  MLIIS_REQUIRE(buf && buf_len >= 56, MLIIS_ERR_ARG, "conv2d_kernel_name: buffer too small");
  GemmPlan g = plan_gemm((long long)Nimg * H * W, Nout, Cred, ksize * ksize, num_cus(), 1);
  snprintf(buf, buf_len, "conv_gemm_nk_k<%d, %d, %d, %s, %s, %s>", g.tm, g.nt, g.tm == 1 ? 2 : 1, has_scale ? "true" : "false",
           g.gz > 1 ? "true" : "false", gemm_narrow(ksize * ksize, Cred) ? "true" : "false");
  return MLIIS_OK;
}

// Workspace (floats) that conv2d_fwd / conv2d_bwd_data may need for split-K partials.
size_t mliis_conv2d_workspace_floats(int Nimg, int H, int W, int Cred, int Nout, int ksize) {
  long long M = (long long)Nimg * H * W;
  const GemmPlan g = plan_gemm(M, Nout, Cred, ksize * ksize, num_cus(), 1);
  return g.gz > 1 ? (size_t)g.gz * M * Nout : 0;
}

// y[M, Cout] (ld = ldy) (+)= conv(x[M, Cin] (ld = ldx), w) + bias ; stride 1, TF-SAME, dilation dil; the weights come as the
// K-contiguous copy wt[k,k,Cout,Cin_total] of the HWIO tensor (mliis_transpose_weights)
int mliis_conv2d_fwd(const float* x, int ldx, const float* x_scale, const float* wt, const float* bias,
                     const float* border_bias, float* y, int ldy, int Nimg, int H, int W, int Cin_total, int ci_begin, int Cin, int Cout,
                     int ksize, int dil, int accumulate, float* stats_part, int stats_swish, int* stats_nblk, float* ws,
                     size_t ws_floats, hipStream_t stream) {
  int rc = conv_check("conv2d_fwd", Nimg, H, W, Cin, Cout, ksize, dil);
  if (rc) return rc;
  MLIIS_REQUIRE(x && wt && y, MLIIS_ERR_ARG, "conv2d_fwd: null pointer");
  MLIIS_REQUIRE((ldx & 3) == 0 && ldx >= Cin && (ldy & 3) == 0 && ldy >= Cout, MLIIS_ERR_ARG, "conv2d_fwd: bad leading dimensions");
  MLIIS_REQUIRE(aligned16(x) && aligned16(wt) && aligned16(bias) && aligned16(y), MLIIS_ERR_ALIGN,
                "conv2d_fwd: pointers must be 16-byte aligned");
  long long M = (long long)Nimg * H * W;
  MLIIS_REQUIRE(M * ldx * 4 < (1LL << 31) && (long long)ksize * ksize * Cin_total * Cout * 4 < (1LL << 31), MLIIS_ERR_UNSUPPORTED,
                "conv2d_fwd: operand larger than 2 GiB (32-bit buffer offsets)");
  GemmPlan g = plan_gemm(M, Cout, Cin, ksize * ksize, num_cus(), ws != nullptr);
  MLIIS_REQUIRE(ci_begin >= 0 && (ci_begin & 3) == 0 && ci_begin + Cin <= Cin_total, MLIIS_ERR_ARG,
                "conv2d_fwd: input-channel window out of range");
  MLIIS_REQUIRE(border_bias == nullptr || (ksize == 3 && dil == 1 && H >= 2 && W >= 2 && aligned16(border_bias)), MLIIS_ERR_ARG,
                "conv2d_fwd: border_bias needs a 3x3 dilation-1 conv on a map of at least 2x2");
  ConvGemmParams p{x, ldx, Nimg, H, W, Cin, ksize * ksize, dil, +1, wt + ci_begin, (long long)Cin_total * Cout, Cin_total, Cout, y, ldy,
                   bias, accumulate, nullptr, g.chunks_per_split, nullptr, 0, x_scale, border_bias};
  MLIIS_REQUIRE(aligned16(x_scale) && (x_scale == nullptr || ksize == 1), MLIIS_ERR_ARG,
                "conv2d_fwd: x_scale must be 16-byte aligned and is only supported for 1x1 convs");
  if (stats_nblk) *stats_nblk = 0;
  if (stats_part != nullptr) {   // fused BN statistics: GEMM epilogue (gz == 1) or the split-K fold (gz > 1)
    MLIIS_REQUIRE(!accumulate && stats_nblk, MLIIS_ERR_ARG, "conv2d_fwd: fused statistics need accumulate == 0 and a stats_nblk output");
    if (g.gz == 1) {
      p.stats_part = stats_part;
      p.stats_swish = stats_swish;
      *stats_nblk = g.gx;
    } else {
      MLIIS_REQUIRE(M < (1LL << 31), MLIIS_ERR_ARG, "conv2d_fwd: too many rows for the split-K statistics fold");
      *stats_nblk = (int)((M + kRedRows - 1) / kRedRows);
    }
  }
  if (g.gz > 1) {
    size_t need = (size_t)g.gz * M * Cout;
    MLIIS_REQUIRE(need <= ws_floats && aligned16(ws) && (ldy & 3) == 0 && aligned16(y), MLIIS_ERR_WORKSPACE,
                  "conv2d_fwd: split-K workspace too small (%zu needed, %zu given) or unaligned output", need, ws_floats);
    p.partial = ws;
  }
  launch_gemm(g, p, stream);
  MLIIS_CHECK_LAUNCH("conv2d_fwd");
  if (g.gz > 1 && stats_part != nullptr) {
    hipLaunchKernelGGL(splitk_reduce_stats_k, dim3((unsigned)((M + kRedRows - 1) / kRedRows), (Cout + 31) / 32), dim3(256), 0, stream, ws, g.gz,
                       (int)M, Cout, y, ldy, bias, border_bias, H, W, stats_part, stats_swish);
    MLIIS_CHECK_LAUNCH("conv2d_fwd_splitk_reduce_stats");
  } else if (g.gz > 1) {
    long long q = M * (Cout / 4);
    int blocks = (int)((q + 255) / 256 > 2048 ? 2048 : (q + 255) / 256);
    hipLaunchKernelGGL(splitk_reduce_k, dim3(blocks), dim3(256), 0, stream, ws, g.gz, M, Cout, y, ldy, bias, accumulate, border_bias, H, W);
    MLIIS_CHECK_LAUNCH("conv2d_fwd_splitk_reduce");
  }
  return MLIIS_OK;
}

// dx[M, Cin_out] (ld = lddx) (+)= conv_transpose(dy[M, Cout] (ld = lddy), w) restricted to input channels
// [ci_begin, ci_begin + Cin_out) of a weight tensor w[k,k,Cin_total,Cout].
int mliis_conv2d_bwd_data(const float* dy, int lddy, const float* w, float* dx, int lddx, int Nimg, int H, int W, int Cin_total,
                          int ci_begin, int Cin_out, int Cout, int ksize, int dil, int accumulate, float* ws, size_t ws_floats,
                          hipStream_t stream) {
  int rc = conv_check("conv2d_bwd_data", Nimg, H, W, Cin_out, Cout, ksize, dil);
  if (rc) return rc;
  MLIIS_REQUIRE(dy && w && dx, MLIIS_ERR_ARG, "conv2d_bwd_data: null pointer");
  MLIIS_REQUIRE(ci_begin >= 0 && ci_begin + Cin_out <= Cin_total, MLIIS_ERR_ARG, "conv2d_bwd_data: channel window out of range");
  MLIIS_REQUIRE((lddy & 3) == 0 && lddy >= Cout && (lddx & 3) == 0 && lddx >= Cin_out && (ci_begin & 3) == 0, MLIIS_ERR_ARG,
                "conv2d_bwd_data: bad leading dimensions");
  MLIIS_REQUIRE(aligned16(dy) && aligned16(w) && aligned16(dx), MLIIS_ERR_ALIGN, "conv2d_bwd_data: pointers must be 16-byte aligned");
  long long M = (long long)Nimg * H * W;
  MLIIS_REQUIRE(M * lddy * 4 < (1LL << 31) && (long long)ksize * ksize * Cin_total * Cout * 4 < (1LL << 31), MLIIS_ERR_UNSUPPORTED,
                "conv2d_bwd_data: operand larger than 2 GiB (32-bit buffer offsets)");
  GemmPlan g = plan_gemm(M, Cin_out, Cout, ksize * ksize, num_cus(), ws != nullptr);
  ConvGemmParams p{dy, lddy, Nimg, H, W, Cout, ksize * ksize, dil, -1, w + (long long)ci_begin * Cout, (long long)Cin_total * Cout,
                   Cout, Cin_out, dx, lddx, nullptr, accumulate, nullptr, g.chunks_per_split, nullptr, 0, nullptr, nullptr};
  if (g.gz > 1) {
    size_t need = (size_t)g.gz * M * Cin_out;
    MLIIS_REQUIRE(need <= ws_floats && aligned16(ws) && (lddx & 3) == 0 && aligned16(dx), MLIIS_ERR_WORKSPACE,
                  "conv2d_bwd_data: split-K workspace too small (%zu needed, %zu given) or unaligned output", need, ws_floats);
    p.partial = ws;
  }
  launch_gemm(g, p, stream);
  MLIIS_CHECK_LAUNCH("conv2d_bwd_data");
  if (g.gz > 1) {
    long long q = M * (Cin_out / 4);
    int blocks = (int)((q + 255) / 256 > 2048 ? 2048 : (q + 255) / 256);
    hipLaunchKernelGGL(splitk_reduce_k, dim3(blocks), dim3(256), 0, stream, ws, g.gz, M, Cin_out, dx, lddx, nullptr, accumulate, nullptr, H, W);
    MLIIS_CHECK_LAUNCH("conv2d_bwd_data_splitk_reduce");
  }
  return MLIIS_OK;
}

// desc: device int32 [ndesc][4] = {offset (floats), taps, Cin, Cout}; src/dst: arenas with identical layout.
int mliis_transpose_weights(const float* src, float* dst, const int* desc, int ndesc, hipStream_t stream) {
  MLIIS_REQUIRE(src && dst && desc && ndesc > 0, MLIIS_ERR_ARG, "transpose_weights: bad arguments");
  hipLaunchKernelGGL(transpose_weights_k, dim3(224, ndesc), dim3(256), 0, stream, src, dst, desc);   // (the largest tensor has ~2000 tiles)
  MLIIS_CHECK_LAUNCH("transpose_weights");
  return MLIIS_OK;
}

size_t mliis_conv2d_bwd_filter_workspace_floats(int Nimg, int H, int W, int Cin, int Cout, int ksize) {
  long long M = (long long)Nimg * H * W;
  FilterPlan f = plan_filter(M, Cin, Cout, ksize * ksize, num_cus());
  return (size_t)f.gz * ksize * ksize * Cin * Cout;
}

// dw[k,k,Cin,Cout] (+)= sum_pixels x[pixel + tap offset, ci] * dy[pixel, co]
int mliis_conv2d_bwd_filter(const float* x, int ldx, const float* x_scale, const float* dy, int lddy, float* dw, int Nimg, int H, int W,
                            int Cin_total, int ci_begin, int Cin, int Cout, int ksize, int dil, int accumulate, float* ws,
                            size_t ws_floats, hipStream_t stream) {
  int rc = conv_check("conv2d_bwd_filter", Nimg, H, W, Cin, Cout, ksize, dil);
  if (rc) return rc;
  MLIIS_REQUIRE(x && dy && ws, MLIIS_ERR_ARG, "conv2d_bwd_filter: null pointer");   // dw == NULL: leave the slabs in ws (mliis_fold_batched)
  MLIIS_REQUIRE((ldx & 3) == 0 && ldx >= Cin && (lddy & 3) == 0 && lddy >= Cout, MLIIS_ERR_ARG, "conv2d_bwd_filter: bad leading dimensions");
  MLIIS_REQUIRE(aligned16(x) && aligned16(dy) && aligned16(ws), MLIIS_ERR_ALIGN, "conv2d_bwd_filter: pointers must be 16-byte aligned");
  long long M = (long long)Nimg * H * W;
  MLIIS_REQUIRE((M + 64) * ldx * 4 < (1LL << 31) && (M + 64) * lddy * 4 < (1LL << 31), MLIIS_ERR_UNSUPPORTED,
                "conv2d_bwd_filter: operand larger than 2 GiB (32-bit buffer offsets)");
  FilterPlan f = plan_filter(M, Cin, Cout, ksize * ksize, num_cus());
  size_t total = (size_t)ksize * ksize * Cin * Cout;
  MLIIS_REQUIRE((size_t)f.gz * total <= ws_floats, MLIIS_ERR_WORKSPACE, "conv2d_bwd_filter: workspace too small (%zu needed, %zu given)",
                (size_t)f.gz * total, ws_floats);
  MLIIS_REQUIRE(aligned16(x_scale), MLIIS_ERR_ALIGN, "conv2d_bwd_filter: x_scale must be 16-byte aligned");
  FilterGradParams p{x, ldx, Nimg, H, W, Cin, ksize * ksize, dil, dy, lddy, Cout, ws, f.rows_per_split, x_scale};
  launch_filter(f, p, stream);
  MLIIS_CHECK_LAUNCH("conv2d_bwd_filter");
  MLIIS_REQUIRE(ci_begin >= 0 && ci_begin + Cin <= Cin_total, MLIIS_ERR_ARG, "conv2d_bwd_filter: input-channel window out of range");
  if (dw == nullptr) return MLIIS_OK;
  hipLaunchKernelGGL(fold_flat_k, dim3(ceil_div((long long)total, kFoldX)), dim3(kFoldX, kFoldY), 0, stream, ws, f.gz, (long long)total, 1.0f, dw,
                     accumulate, (long long)Cin * Cout, (long long)Cin_total * Cout, (long long)ci_begin * Cout);
  MLIIS_CHECK_LAUNCH("conv2d_bwd_filter_reduce");
  return MLIIS_OK;
}
}
