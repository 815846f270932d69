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
#include "conv_gemm_kernels.hpp"
#include "rng_masks.hpp"

namespace mliis {

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

// ------------------------------------------------------------------------------------------------ weight shadow (HWIO -> HWOI)
// dst[off + tap*Ci*Co + co*Ci + ci] = src[off + tap*Ci*Co + ci*Co + co] for every descriptor (off, taps, Ci, Co): one launch per
// inner step keeps a K-contiguous copy of all dense-conv weights so the FORWARD GEMM can use the same b128-fragment B path as
// backward-data.  32x32 LDS tiles: coalesced reads along co, coalesced writes along ci.
// ndesc_flat > 0: ONE tile per workgroup over a grid of exactly the tiles of all descriptors (the caller knows their total): workgroup b
// finds its descriptor with a wave prefix scan of the tile counts (64 descriptors per pass).  The two-dimensional form (descriptor =
// blockIdx.y, 224 workgroups each) launched 8512 workgroups for ~2000 tiles, most of them only to find nothing to do.
// x3 (mliis_weight_shadows; flat form only): the workgroups behind the first x3.first_block build the split-product weight images
// of conv_x3.hip -- two 128-thread pack blocks each -- so the two weight shadows of a step are ONE launch.
struct X3PackArgs {
  char* images;
  const long long* desc;
  int ndesc, blocks, first_block;   // first_block = the transposes' tile count; blocks == 0: no images
};
// rng (mliis_weight_shadows_rng; flat form only): the workgroups behind the pack blocks draw the masks of the training step (rng.hip's
// job table) -- the third launch every step starts with rides in the same grid.  blocks == 0: none.
struct RngArgs {
  unsigned* state;
  MaskJobs jobs;
  int blocks, first_block;
};
__global__ __launch_bounds__(256) void transpose_weights_k(const float* __restrict__ src, float* __restrict__ dst,
                                                           const int* __restrict__ desc, unsigned* __restrict__ amax_bits, int ndesc_flat,
                                                           X3PackArgs x3, RngArgs rng) {
  if (rng.blocks > 0 && (int)blockIdx.x >= rng.first_block) {   // (uniform)
    rng_masks_body(rng.state, rng.jobs, (int)blockIdx.x - rng.first_block, rng.blocks);
    return;
  }
  if (x3.blocks > 0 && (int)blockIdx.x >= x3.first_block) {   // (uniform)
    const int pb = ((int)blockIdx.x - x3.first_block) * 2 + (int)(threadIdx.x >> 7);
    if (pb < x3.blocks) x3_pack_block(src, x3.images, x3.desc, x3.ndesc, pb, (int)(threadIdx.x & 127));
    return;
  }
  __shared__ float tile[32][33];
  __shared__ float wmax[4];
  __shared__ int s_d[2];
  int d = blockIdx.y, t_first = blockIdx.x, t_step = gridDim.x;
  if (ndesc_flat > 0) {   // (uniform)
    if (threadIdx.x == 0) s_d[0] = -1;   // a grid larger than the table's tile total (stale host count): surplus workgroups leave
    __syncthreads();
    if (threadIdx.x < 64) {
      const int lane = threadIdx.x, b = blockIdx.x;
      int base = 0;
      for (int d0 = 0; d0 < ndesc_flat; d0 += 64) {
        const int dd = d0 + lane;
        int nt = 0;
        if (dd < ndesc_flat) nt = desc[4 * dd + 1] * ((desc[4 * dd + 2] + 31) / 32) * ((desc[4 * dd + 3] + 31) / 32);
        int incl = nt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int up = __shfl_up(incl, o, 64);
          if (lane >= o) incl += up;
        }
        const int lo = base + incl - nt, hi = base + incl;
        if (b >= lo && b < hi) {
          s_d[0] = dd;
          s_d[1] = b - lo;
        }
        base += __shfl(incl, 63, 64);
        if (b < base) break;   // (uniform: base is a wave-wide value)
      }
    }
    __syncthreads();
    d = s_d[0];
    if (d < 0) return;   // (uniform)
    t_first = s_d[1];
    t_step = 1 << 30;
  }
  const int off = desc[4 * d + 0], taps = desc[4 * d + 1], Ci = desc[4 * d + 2], Co = desc[4 * d + 3];
  const int tci = (Ci + 31) / 32, tco = (Co + 31) / 32;
  const int ntiles = taps * tci * tco;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  float amax = 0.f;
  for (int tIdx = t_first; tIdx < ntiles; tIdx += t_step) {
    const int tap = tIdx / (tci * tco);
    const int rem = tIdx - tap * tci * tco;
    const int ci0 = (rem / tco) * 32, co0 = (rem % tco) * 32;
    const float* s0 = src + off + (long long)tap * Ci * Co;
    float* d0 = dst + off + (long long)tap * Ci * Co;
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
      const int ci = ci0 + r, co = co0 + tx;
      const float v = (ci < Ci && co < Co) ? s0[(long long)ci * Co + co] : 0.f;
      tile[r][tx] = v;
      amax = fmaxf(amax, fabsf(v));
    }
    __syncthreads();
#pragma unroll
    for (int r = ty; r < 32; r += 8) {
      const int co = co0 + r, ci = ci0 + tx;
      if (co < Co && ci < Ci) d0[(long long)co * Ci + ci] = tile[tx][r];
    }
    __syncthreads();
  }
  if (amax_bits == nullptr) return;   // (uniform)
  // per-tensor max |w| for the fp8 operand scale: wave max, workgroup max, then one integer atomicMax per workgroup (non-negative
  // floats order like their bit patterns; a maximum does not depend on the order of the updates: deterministic)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o, 64));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = amax;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(amax_bits + d, __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
}

// ------------------------------------------------------------------------------------------------ host-side planning
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

static const bool kStreamK = getenv("MLIIS_NO_STREAMK") == nullptr;   // (A/B switch for profiles/r02_notes.md)
constexpr int kSkMinChunks = 16;  // stream-K remainder only for K >= 512 (shorter K: the fixed cost of a part outweighs the balance)
constexpr int kSkMinPart = 2;     // K chunks per part, at least
constexpr int kGemmFill = 4;     // fwd / bwd-data: narrow the column tiles of sub-chip grids until there are this many blocks per CU
constexpr int kFilterFill = 1;   // bwd-filter: same idea, one block per CU is enough (its slabs already split the pixel axis)

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
  // (3x3 convs -- the 14x14 decoder level, K = 1008 / 2016 -- stop at 64 columns and split K instead: a 64 x 32 tile re-reads its A
  //  chunk from LDS per 32 columns.  Round 6, same box, three alternations: floor 2: 20.5 us per launch, 3484 images/s; 4: 16.3 us + a
  //  split-K fold on the two backward-data convs, 3498; 7 (no narrowing): 15.5 us but 12 us folds of eight slabs, 3481.)
  const int min_nt = ntaps == 9 ? 4 : 2;
  if (g.tm == 1 && g.gx < num_cus)
    while (g.nt > min_nt && (long long)g.gx * g.gy < (long long)kGemmFill * num_cus) {
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
  // data-parallel + stream-K remainder (conv_gemm_sk_k): long-K layers whose tile count leaves the last round of the chip part-empty
  // (the 56x56 decoder convs: 392 / 784 / 1176 tiles on 256 CUs).  Needs a workspace for the segment slabs.
  g.sk_full = g.sk_rem = g.sk_parts = g.sk_ipp = g.sk_nchunks = g.sk_smax = 0;
  const long long tiles = (long long)g.gx * g.gy;
  if (allow_split && kStreamK && g.tm == 1 && g.gz == 1 && !gemm_narrow(ntaps, C) && nchunks >= kSkMinChunks && tiles > num_cus &&
      tiles % num_cus != 0 && tiles < (1 << 20)) {
    const int full = (int)(tiles / num_cus) * num_cus, rem = (int)(tiles - full);
    // (pure stream-K -- every tile through the parts, two equal parts per CU -- was measured and not adopted: profiles/r02_notes.md)
    long long total = (long long)rem * nchunks;
    int parts = num_cus;
    if (parts > 4 * rem) parts = 4 * rem;         // at most ~5 slabs per remainder tile for the fix-up to add (a short tail of 16 tiles
                                                  // cut 256 ways cost a 41 us fix-up: 16 workgroups adding 16 slabs each)
    if (parts < rem) parts = rem;                 // (at most two segments per part)
    if (total / parts < kSkMinPart) parts = (int)(total / kSkMinPart);   // parts of at least kSkMinPart chunks
    if (parts >= 1) {
      const int ipp = (int)((total + parts - 1) / parts);
      g.sk_full = full;
      g.sk_rem = rem;
      g.sk_parts = (int)((total + ipp - 1) / ipp);
      g.sk_ipp = ipp;
      g.sk_nchunks = nchunks;
      g.sk_smax = (nchunks + ipp - 1) / ipp + 1;
    }
  }
  return g;
}

static inline void plan_filter_split(FilterPlan& f, long long M, int C, int Nout, int ntaps, int num_cus) {
  const int bci = 64 * f.tmf;
  f.gy = (Nout + f.nt * 16 - 1) / (f.nt * 16);
  f.gx = f.multitap ? 1 : ntaps * ((C + bci - 1) / bci);
  long long base = (long long)f.gx * f.gy;
  long long want = (2LL * num_cus) / base;  // one full round of two co-resident blocks per CU (no half-empty second round) ...
  {  // ... bounded by the slab count the fold has to read: 96 in general, up to 512 for tiny filters (fold cost ~ slabs * size)
    const long long total = (long long)ntaps * C * Nout;
    long long cap = (1LL << 21) / (total > 0 ? total : 1);
    if (cap > 512) cap = 512;
    if (cap < 96) cap = 96;
    if (want > cap) want = cap;
  }
  if (want < 1) want = 1;
  long long rps = (M + want - 1) / want;
  // at least 256 pixel rows per slab.  (Round 1, every problem its own launch: 32 -> 2395, 64 -> 2404, 96 -> 2409, 128 -> 2416,
  // 192 -> 2408, 256 -> 2406 images/s.  Round 3, the problems of a kernel instantiation batched into one grid -- the other problems'
  // workgroups fill the chip, so the 14x14 layers' 13 slabs of 4 chunks each were mostly pipeline fill and slab traffic:
  // 128 -> 3081, 192 -> 3104, 256 -> 3113-3122, 384 -> 3110, 512 -> 3104, 800 -> 3082, 1568 -> 2917 images/s.)
  if (rps < 256) rps = 256;
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
  // a 3x3 conv over very few channels (the 8-channel sliver of the RSD concat): all taps in one channel block, dY read once
  f.multitap = (ntaps > 1 && ntaps * C <= 128) ? 1 : 0;
  if (f.multitap) f.tmf = ntaps * C > 64 ? 2 : 1;
  plan_filter_split(f, M, C, Nout, ntaps, num_cus);
  // small maps cannot be split further along the pixels (64 rows per slab): narrower tiles instead (latency-bound, see plan_gemm)
  // (round 3, batched launches: thresholds of 0 .. 8 x num_cus blocks: 3033 / 3060 / 3086 / 3096 / 3117-3122 / 3119 / 3120 / 3125 images/s)
  while ((long long)f.gx * f.gy * f.gz < (long long)kFilterFill * num_cus && ((f.tmf == 2 && !f.multitap) || f.nt > 2)) {
    if (f.tmf == 2 && !f.multitap) f.tmf = 1;
    else f.nt = (f.nt + 1) / 2;
    plan_filter_split(f, M, C, Nout, ntaps, num_cus);
  }
  return f;
}

// Matrix-core operand precision of a dense-conv call (the `precision` argument of the three entry points): MLIIS_PREC_FP32 = fp32
// operands (v_mfma_f32_16x16x4_f32), MLIIS_PREC_BF16 = operands rounded to bf16 in registers, fp32 accumulation
// (v_mfma_f32_16x16x32_bf16).  Per call: nothing process-wide, a captured HIP graph keeps what each launch was issued with.
static void launch_gemm(const GemmPlan& g, const ConvGemmParams& p, int precision, hipStream_t stream, float* sk_slab = nullptr) {
  if (precision == MLIIS_PREC_FP8) {   // (1x1 forward convs only: never a stream-K plan's long-K 3x3 layer)
    launch_gemm_fp8(g, p, stream);
    return;
  }
  if (g.sk_parts > 0 && sk_slab != nullptr) {
    if (precision == MLIIS_PREC_BF16) launch_gemm_sk_bf16(g, p, sk_slab, stream);
    else launch_gemm_sk_t<0>(g, p, sk_slab, stream);
    return;
  }
  if (precision == MLIIS_PREC_BF16) launch_gemm_bf16(g, p, stream);
  else launch_gemm_t<0>(g, p, stream);
}
static void launch_filter(const FilterPlan& f, const FilterGradParams& p, int precision, hipStream_t stream) {
  if (precision != MLIIS_PREC_FP32) launch_filter_bf16(f, p, stream);   // (fp8 mode: the backward passes take bf16 operands)
  else launch_filter_t<false>(f, p, stream);
}
static int prec_check(const char* name, int precision) {
  MLIIS_REQUIRE(precision == MLIIS_PREC_FP32 || precision == MLIIS_PREC_BF16 || precision == MLIIS_PREC_FP8, MLIIS_ERR_ARG,
                "%s: precision must be MLIIS_PREC_FP32 (0), MLIIS_PREC_BF16 (1) or MLIIS_PREC_FP8 (2), got %d", name, precision);
  return MLIIS_OK;
}

// ------------------------------------------------------------------------------------------------ short-K 1x1 convs (conv1x1_stream_k)
// KC = 16-wide K groups (K <= 112), NT = column tiles per wave, KC * NT <= 8 (the B fragments live in registers)
struct StreamPlan {
  int kc, nt, gx, gy, row_groups;
};
static inline bool stream_plan(long long M, int K, int Nout, int num_cus, StreamPlan* sp) {
  // (whole step, one box: off 2335, K <= 96 / M >= 2048: 2352, K <= 112 / M >= 1024: 2359 images/s; the 14x14 layers have M = 1568)
  if (K > 112 || M < 1024 || M >= (1LL << 27)) return false;
  sp->kc = (K + 15) / 16;
  // column tiles per wave: KC * NT <= 8 B-fragment quads in registers.  (Whole step on one box: cap 24 -> 2426, 16 -> 2440, 8 -> 2443,
  // 4 -> 2447, 2 -> 2439 images/s: more column blocks = more waves in flight beats fewer re-reads of A.)
  int nt = 8 / sp->kc;
  if (nt < 1) nt = 1;
  // (Round 3 capped NT at 4: the instances with 5-8 column tiles need 153-199 VGPRs, so only ONE 512-thread workgroup fits a CU and the
  // grid of two per CU ran in two rounds -- stamps build: the second half of the waves of the 16 -> 96 expand conv started 6-13 us after
  // the first; 23 us per launch.)
  // Round 6, re-measured on the present kernels (same box, two alternations each, whole step): cap 1: 3496, 2: 3505, 3: 3486, 4: 3487,
  // 8: 3455 images/s -- two column tiles per wave: more column blocks in flight, and a workgroup holds half the B fragments.
  if (nt > 2) nt = 2;
  const int tiles = (Nout + 15) / 16;
  if (nt > tiles) nt = tiles;
  const int gy = (tiles + nt - 1) / nt;   // balanced column tiles: Nout = 144 -> 9 tiles -> 2 x NT 5 rather than NT 8 + a sliver
  sp->nt = (tiles + gy - 1) / gy;
  sp->gy = gy;
  sp->row_groups = (int)((M + 15) / 16);
  long long gx = (16LL / kStreamWaves * num_cus) / gy;     // about 16 waves per CU in flight
  if (gx < 1) gx = 1;
  if (gx > (sp->row_groups + kStreamWaves - 1) / kStreamWaves) gx = (sp->row_groups + kStreamWaves - 1) / kStreamWaves;
  sp->gx = (int)gx;
  return true;
}
template <bool AIN>
static bool launch_stream_f32(const StreamPlan& sp, const ConvGemmParams& p, hipStream_t stream) {
  dim3 grid(sp.gx, sp.gy), block(64 * kStreamWaves);
#define S(KC_, NT_) hipLaunchKernelGGL((conv1x1_stream_k<KC_, NT_, 0, AIN>), grid, block, 0, stream, p, sp.row_groups); break;
#define SN(KC_, NT_) if constexpr (AIN) return false; else { S(KC_, NT_) }   // (stream_plan gives at most four column tiles: no AIN instance)
  switch (sp.kc) {
    case 1: switch (sp.nt) { case 1: S(1, 1) case 2: S(1, 2) case 3: S(1, 3) case 4: S(1, 4) case 5: SN(1, 5) case 6: SN(1, 6) case 7: SN(1, 7) case 8: SN(1, 8) default: return false; } break;
    case 2: switch (sp.nt) { case 1: S(2, 1) case 2: S(2, 2) case 3: S(2, 3) case 4: S(2, 4) default: return false; } break;
    case 3: switch (sp.nt) { case 1: S(3, 1) case 2: S(3, 2) default: return false; } break;
    case 4: switch (sp.nt) { case 1: S(4, 1) case 2: S(4, 2) default: return false; } break;
    case 5: switch (sp.nt) { case 1: S(5, 1) default: return false; } break;
    case 6: switch (sp.nt) { case 1: S(6, 1) default: return false; } break;
    case 7: switch (sp.nt) { case 1: S(7, 1) default: return false; } break;
    default: return false;
  }
#undef SN
#undef S
  return true;
}
static bool launch_stream(const StreamPlan& sp, const ConvGemmParams& p, hipStream_t stream, int precision = MLIIS_PREC_FP32, bool ain = false) {
  if (precision != MLIIS_PREC_FP32) return launch_stream_lowp(precision, sp.kc, sp.nt, dim3(sp.gx, sp.gy), p, sp.row_groups, stream, ain);
  return ain ? launch_stream_f32<true>(sp, p, stream) : launch_stream_f32<false>(sp, p, stream);
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

// Workgroups of a 1x1 kernel instance that fit one CU according to the runtime (profiling aid: the planners assume two 512-thread
// workgroups per CU -- a register count above 128 silently halves that and the grid runs in two rounds).  kind 0: conv1x1_stream_k<kc, nt>,
// 1: conv1x1_ksplit_k<kc, nt, 8>; fp32 instances.  Returns MLIIS_ERR_ARG for a combination that is not instantiated.
int mliis_conv1x1_occupancy(int kind, int kc, int nt, int* blocks_per_cu) {
  MLIIS_REQUIRE(blocks_per_cu, MLIIS_ERR_ARG, "conv1x1_occupancy: null pointer");
  const void* fn = nullptr;
#define ST(KC_, NT_) if (kind == 0 && kc == KC_ && nt == NT_) fn = reinterpret_cast<const void*>(&conv1x1_stream_k<KC_, NT_, 0>);
#define KS(KC_, NT_) if (kind == 1 && kc == KC_ && nt == NT_) fn = reinterpret_cast<const void*>(&conv1x1_ksplit_k<KC_, NT_, 8, 0>);
  ST(1, 1) ST(1, 2) ST(1, 3) ST(1, 4) ST(2, 1) ST(2, 2) ST(2, 3) ST(3, 1) ST(3, 2) ST(4, 1) ST(4, 2) ST(5, 1) ST(6, 1) ST(7, 1)
  KS(1, 1) KS(1, 2) KS(1, 3) KS(1, 4) KS(1, 5) KS(1, 6) KS(1, 7) KS(2, 1) KS(2, 2) KS(2, 3) KS(2, 4) KS(3, 1) KS(3, 2) KS(4, 1) KS(4, 2)
  KS(5, 1) KS(6, 1) KS(7, 1) KS(4, 3) KS(5, 2) KS(6, 2)
#undef ST
#undef KS
  MLIIS_REQUIRE(fn != nullptr, MLIIS_ERR_ARG, "conv1x1_occupancy: no instance kind %d <%d, %d>", kind, kc, nt);
  int nb = 0;
  const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 512, 0);
  MLIIS_REQUIRE(e == hipSuccess, MLIIS_ERR_LAUNCH, "conv1x1_occupancy: %s", hipGetErrorString(e));
  *blocks_per_cu = nb;
  return MLIIS_OK;
}

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
// ------------------------------------------------------------------------------------------------ long-K 1x1 convs on small maps (conv1x1_ksplit_k)
// K split over the 8 waves of a workgroup instead of over workgroups + a fold launch: the MBConv project convs forward and expand
// convs backward-data of the 56x56 / 28x28 / 14x14 maps (K = 144..672, 1568..25088 rows at N = 8).
constexpr long long kKsplitMaxRows = 32768;   // up to the 56x56 maps (measured: 8192 -> 2868, 32768 -> 2872, 131072 -> 2875 images/s)
static const bool kKsplit = getenv("MLIIS_NO_KSPLIT") == nullptr;   // (A/B switch for profiles/r02_notes.md)
static inline bool ksplit_plan(long long M, int K, int Nout, int num_cus, StreamPlan* sp) {
  if (!kKsplit || K <= 112 || K > 7 * 128 || M > kKsplitMaxRows || M < 16) return false;
  sp->kc = (K + 127) / 128;                // 16-wide K groups per wave
  // B fragments of the wave's K slice in registers: KC * NT <= 8 quads; on the small maps (one workgroup per CU below: 256 registers)
  // the long-K instances take 12 -- K = 480 / 672: three / two column tiles per wave, fewer column blocks re-reading A and fewer row
  // groups per workgroup (round 6, same box: <6,1> 11.9 -> <6,2> 10.9 us, <4,2> 8.7 -> <4,3> 8.1 us, +0.25 % on the step; wider tiles
  // for K <= 384 measured slower: <2,4> 5.3 -> <2,7> 8.0 us)
  int nt = ((M <= 8192 && sp->kc >= 4) ? 12 : 8) / sp->kc;
  if (nt < 1) nt = 1;
  if (nt > 7) nt = 7;                      // (the 8 x 16 x (16 NT + 4) staging tile must stay below 64 KB of static LDS)
  const int tiles = (Nout + 15) / 16;
  if (nt > tiles) nt = tiles;
  const int gy = (tiles + nt - 1) / nt;    // balanced column tiles
  sp->nt = (tiles + gy - 1) / gy;
  sp->gy = gy;
  sp->row_groups = (int)((M + 15) / 16);
  // Workgroups (ONE round: rounding up would spill a few workgroups into a second): every workgroup loads the B slice of its column block into registers once, and on the small maps that is most of the
  // launch's traffic (K = 672, 112 columns: 300 KB per workgroup against 4 MB of A in all) -- ONE workgroup per CU on the 28x28 / 14x14
  // maps, two on the 56x56 ones.  Round 6, same box, whole step: two everywhere 3491, one everywhere 3507-3520, one up to 8192 rows
  // 3526-3530; 0.75 / 0.5 per CU 3462 / 3381 images/s.  (The KC = 7 instances need 132 VGPRs: one workgroup per CU anyway.)
  long long gx = ((sp->kc >= 7 || M <= 8192 ? 1LL : 2LL) * num_cus) / gy;
  if (gx < 1) gx = 1;
  if (gx > sp->row_groups) gx = sp->row_groups;
  sp->gx = (int)gx;
  return true;
}
static bool launch_ksplit(const StreamPlan& sp, const ConvGemmParams& p_, hipStream_t stream, int precision) {
  ConvGemmParams p = p_;
#ifdef KS_DBG
  p.dbg_stamps = getenv("MLIIS_KS_STAMPS") ? (unsigned long long*)strtoull(getenv("MLIIS_KS_STAMPS"), nullptr, 0) : nullptr;
#endif
  dim3 grid(sp.gx, sp.gy);
  if (precision != MLIIS_PREC_FP32) return launch_ksplit_lowp(precision, sp.kc, sp.nt, grid, p, sp.row_groups, stream);
  return launch_ksplit_t<0>(sp.kc, sp.nt, grid, p, sp.row_groups, stream);
}

int mliis_conv2d_kernel_name(int Nimg, int H, int W, int Cred, int Nout, int ksize, int has_scale, int precision, char* buf, size_t buf_len) {
  MLIIS_REQUIRE(buf && buf_len >= 64, MLIIS_ERR_ARG, "conv2d_kernel_name: buffer too small");
  StreamPlan sp;
  if (ksize == 1 && !has_scale && stream_plan((long long)Nimg * H * W, Cred, Nout, num_cus(), &sp)) {
    snprintf(buf, buf_len, "conv1x1_stream_k<%d, %d, %d>", sp.kc, sp.nt, precision);   // (a call without accumulate / border bias)
    return MLIIS_OK;
  }
  if (ksize == 1 && (!has_scale || H * W >= 16) && ksplit_plan((long long)Nimg * H * W, Cred, Nout, num_cus(), &sp)) {
    snprintf(buf, buf_len, "conv1x1_ksplit_k<%d, %d, 8, %d>", sp.kc, sp.nt, precision);
    return MLIIS_OK;
  }
  GemmPlan g = plan_gemm((long long)Nimg * H * W, Nout, Cred, ksize * ksize, num_cus(), 1);
  if (g.sk_parts > 0 && !has_scale) {   // (+ sk_fixup_k<NT> for the remainder tiles)
    snprintf(buf, buf_len, "conv_gemm_sk_k<%d, 2, false, %d>", g.nt, precision == MLIIS_PREC_FP8 ? MLIIS_PREC_BF16 : precision);
    return MLIIS_OK;
  }
  snprintf(buf, buf_len, "conv_gemm_nk_k<%d, %d, %d, %s, %s, %s, %d>", g.tm, g.nt, g.tm == 1 ? 2 : 1, has_scale ? "true" : "false",
           g.gz > 1 ? "true" : "false", gemm_narrow(ksize * ksize, Cred) ? "true" : "false", (precision == MLIIS_PREC_FP8 && ksize != 1) ? MLIIS_PREC_BF16 : precision);
  return MLIIS_OK;
}

// Workspace (floats) that conv2d_fwd / conv2d_bwd_data may need for split-K partials.
size_t mliis_conv2d_workspace_floats(int Nimg, int H, int W, int Cred, int Nout, int ksize) {
  long long M = (long long)Nimg * H * W;
  const GemmPlan g = plan_gemm(M, Nout, Cred, ksize * ksize, num_cus(), 1);
  return g.gz > 1 ? (size_t)g.gz * M * Nout : g.sk_slab_floats();
}

// y[M, Cout] (ld = ldy) (+)= conv(x[M, Cin] (ld = ldx), w) + bias ; stride 1, TF-SAME, dilation dil; the weights come as the
// K-contiguous copy wt[k,k,Cout,Cin_total] of the HWIO tensor (mliis_transpose_weights)
int mliis_conv2d_fwd(const float* x, int ldx, const float* x_scale, const float* wt, const float* bias,
                     const float* border_bias, float* y, int ldy, int Nimg, int H, int W, int Cin_total, int ci_begin, int Cin, int Cout,
                     int ksize, int dil, int accumulate, float* stats_part, int stats_swish, int* stats_nblk, float* ws,
                     size_t ws_floats, int precision, float fp8_act_scale, const float* fp8_w_amax, int x_dtype, int y_dtype,
                     hipStream_t stream) {
  int rc = conv_check("conv2d_fwd", Nimg, H, W, Cin, Cout, ksize, dil);
  if (rc) return rc;
  if ((rc = prec_check("conv2d_fwd", precision))) return rc;
  // bf16 STORAGE of x and / or y (an expanded MBConv tensor): bf16 operands on the matrix cores, a 1x1 conv, no accumulate, and no
  // plan that finishes its tiles in a second launch (split-K / stream-K slabs) -- the statistics are formed from the rounded output
  const int y_block = (y_dtype >> 8) & 0xff;   // MLIIS_DT_BLOCKED(v): y in the group-blocked layout [Cout / v][N H W][v]
  y_dtype &= 0xff;
  MLIIS_REQUIRE(y_block == 0 || ((y_block == 2 || y_block == 4) && y_dtype == MLIIS_DT_F32 && Cout % y_block == 0), MLIIS_ERR_ARG,
                "conv2d_fwd: a group-blocked output takes v = 2 | 4 dividing Cout and fp32 storage");
  const bool xbf = x_dtype == MLIIS_DT_BF16, ybf = y_dtype == MLIIS_DT_BF16;
  MLIIS_REQUIRE((x_dtype == MLIIS_DT_F32 || xbf) && (y_dtype == MLIIS_DT_F32 || ybf), MLIIS_ERR_ARG, "conv2d_fwd: bad storage type");
  if (xbf || ybf) {
    MLIIS_REQUIRE(precision == MLIIS_PREC_BF16 && ksize == 1 && border_bias == nullptr && !(ybf && accumulate), MLIIS_ERR_UNSUPPORTED,
                  "conv2d_fwd: bf16 tensors need MLIIS_PREC_BF16, a 1x1 conv and (bf16 output) no accumulate");
    ws = nullptr;   // (no split-K / stream-K)
    ws_floats = 0;
  }
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
  // fp8 operands are for the 1x1 convs (BASELINE configs[4]: "fp8 MFMA on 1x1 pointwise convs"); a 3x3 call in that mode takes bf16
  if (precision == MLIIS_PREC_FP8 && ksize != 1) precision = MLIIS_PREC_BF16;
  MLIIS_REQUIRE(precision != MLIIS_PREC_FP8 || (fp8_act_scale > 0.0f && fp8_w_amax != nullptr), MLIIS_ERR_ARG,
                "conv2d_fwd: fp8 operands need a positive activation scale and the weight tensor's amax (mliis_transpose_weights)");
  ConvGemmParams p{x, ldx, Nimg, H, W, Cin, ksize * ksize, dil, +1, wt + ci_begin, (long long)Cin_total * Cout, Cin_total, Cout, y, ldy,
                   bias, accumulate, nullptr, g.chunks_per_split, nullptr, 0, x_scale, border_bias, fp8_act_scale, fp8_w_amax};
  p.a_bf16 = xbf;
  p.out_bf16 = ybf;
  MLIIS_REQUIRE(aligned16(x_scale) && (x_scale == nullptr || ksize == 1), MLIIS_ERR_ARG,
                "conv2d_fwd: x_scale must be 16-byte aligned and is only supported for 1x1 convs");
  if (stats_nblk) *stats_nblk = 0;
  {  // short-K 1x1 convs (the MBConv expand convs): barrier-free streaming kernel
    StreamPlan sp;
    // (the instance follows the call's operand precision: one rounding rule for every matrix-core conv of a reduced-precision step)
    if (ksize == 1 && x_scale == nullptr && border_bias == nullptr && !accumulate && !xbf && M * ldx * 4 < (1LL << 31) &&
        M * ldy * 4 < (1LL << 31) && stream_plan(M, Cin, Cout, num_cus(), &sp)) {   // (the streaming kernel reads an fp32 A)
      MLIIS_REQUIRE(stats_part == nullptr || stats_nblk, MLIIS_ERR_ARG, "conv2d_fwd: fused statistics need a stats_nblk output");
      p.stats_part = stats_part;
      p.stats_swish = stats_swish;
      p.c_block = y_block;
      if (launch_stream(sp, p, stream, precision)) {
        MLIIS_CHECK_LAUNCH("conv2d_fwd_stream");
        if (stats_part != nullptr) *stats_nblk = sp.gx;
        return MLIIS_OK;
      }
      p.stats_part = nullptr;
      p.c_block = 0;
    }
    MLIIS_REQUIRE(y_block == 0, MLIIS_ERR_UNSUPPORTED, "conv2d_fwd: a group-blocked output needs the streamed 1x1 plan (mliis_conv2d_kernel_name)");
    // long-K 1x1 convs on small maps (the MBConv project convs, SE gate on load): K split inside the workgroup, one launch
    if (ksize == 1 && border_bias == nullptr && !ybf && M * ldx * 4 < (1LL << 31) && M * ldy * 4 < (1LL << 31) &&
        (stats_part == nullptr || !accumulate) && (x_scale == nullptr || H * W >= 16) &&   // (the gate goes through LDS: two images per row group at most)
        ksplit_plan(M, Cin, Cout, num_cus(), &sp)) {   // (its finishing threads write fp32)
      MLIIS_REQUIRE(stats_part == nullptr || stats_nblk, MLIIS_ERR_ARG, "conv2d_fwd: fused statistics need a stats_nblk output");
      p.stats_part = stats_part;
      p.stats_swish = stats_swish;
      if (launch_ksplit(sp, p, stream, precision)) {
        MLIIS_CHECK_LAUNCH("conv2d_fwd_ksplit");
        if (stats_part != nullptr) *stats_nblk = sp.gx;
        return MLIIS_OK;
      }
      p.stats_part = nullptr;
    }
  }
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
  float* sk_slab = nullptr;
  if (g.sk_parts > 0 && x_scale == nullptr && ws != nullptr && aligned16(ws) && g.sk_slab_floats() <= ws_floats && (ldy & 3) == 0) sk_slab = ws;
  launch_gemm(g, p, precision, stream, sk_slab);
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

// 1 when a 1x1 conv of this shape runs on the streaming plan, i.e. when mliis_conv2d_fwd_bnin accepts it (the MBConv expand convs:
// Cin <= 112, N H W >= 1024); 0 otherwise.
int mliis_conv2d_fwd_bnin_ok(int Nimg, int H, int W, int Cin, int Cout) {
  StreamPlan sp;
  const long long M = (long long)Nimg * H * W;
  if (Nimg <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (Cin & 3) || (Cout & 3) || M * Cin * 4 >= (1LL << 31) || M * Cout * 4 >= (1LL << 31))
    return 0;
  return stream_plan(M, Cin, Cout, num_cus(), &sp) ? 1 : 0;
}

// y = conv1x1(a, w) with  a = ((z - mean) * rstd * gamma + beta) * img_scale[image] + res  formed while z is loaded: the plain batch
// norm in FRONT of the conv (an MBConv project BN, efficientnet_model.py:283-288: drop-connect and the identity skip behind it,
// utils.py:157-170) fused into the NEXT block's expand conv (efficientnet_model.py:175-182).  The launch folds the batch norm's
// stage-1 partials bn_part [bn_nblk][2][Cin] (left by the conv that produced z), publishes mean / rstd, advances the moving
// averages (nullable pair) and writes the finished tensor a to a_out [M, Cin] (ld = lda_out): exactly what mliis_bn_apply_fused
// (no activation) followed by mliis_conv2d_fwd computes, in one launch.  z, res, a_out: fp32.  stats_part / y_dtype (incl.
// MLIIS_DT_BLOCKED) / precision as for mliis_conv2d_fwd.  MLIIS_ERR_UNSUPPORTED unless mliis_conv2d_fwd_bnin_ok(...).
int mliis_conv2d_fwd_bnin(const float* z, int ldz, const float* bn_part, int bn_nblk, float eps, float momentum, float* mean, float* rstd,
                          float* moving_mean, float* moving_var, const float* gamma, const float* beta, const float* img_scale,
                          const float* res, int ldr, float* a_out, int lda_out, const float* wt, float* y, int ldy, int Nimg, int H, int W,
                          int Cin, int Cout, float* stats_part, int stats_swish, int* stats_nblk, int precision, float fp8_act_scale,
                          const float* fp8_w_amax, int y_dtype, hipStream_t stream) {
  int rc = conv_check("conv2d_fwd_bnin", Nimg, H, W, Cin, Cout, 1, 1);
  if (rc) return rc;
  if ((rc = prec_check("conv2d_fwd_bnin", precision))) return rc;
  MLIIS_REQUIRE(z && bn_part && mean && rstd && gamma && beta && a_out && wt && y, MLIIS_ERR_ARG, "conv2d_fwd_bnin: null pointer");
  MLIIS_REQUIRE(bn_nblk > 0 && (moving_mean == nullptr) == (moving_var == nullptr), MLIIS_ERR_ARG,
                "conv2d_fwd_bnin: needs stage-1 partials; the moving statistics come as a pair");
  const int y_block = (y_dtype >> 8) & 0xff;
  y_dtype &= 0xff;
  MLIIS_REQUIRE(y_block == 0 || ((y_block == 2 || y_block == 4) && y_dtype == MLIIS_DT_F32 && Cout % y_block == 0), MLIIS_ERR_ARG,
                "conv2d_fwd_bnin: a group-blocked output takes v = 2 | 4 dividing Cout and fp32 storage");
  const bool ybf = y_dtype == MLIIS_DT_BF16;
  MLIIS_REQUIRE(y_dtype == MLIIS_DT_F32 || ybf, MLIIS_ERR_ARG, "conv2d_fwd_bnin: bad storage type");
  MLIIS_REQUIRE(!ybf || precision == MLIIS_PREC_BF16, MLIIS_ERR_UNSUPPORTED, "conv2d_fwd_bnin: a bf16 output needs MLIIS_PREC_BF16");
  MLIIS_REQUIRE((ldz & 3) == 0 && ldz >= Cin && (lda_out & 3) == 0 && lda_out >= Cin && (ldy & 3) == 0 && ldy >= Cout &&
                    (res == nullptr || ((ldr & 3) == 0 && ldr >= Cin)),
                MLIIS_ERR_ARG, "conv2d_fwd_bnin: bad leading dimensions");
  MLIIS_REQUIRE(aligned16(z) && aligned16(bn_part) && aligned16(gamma) && aligned16(beta) && aligned16(res) && aligned16(a_out) &&
                    aligned16(wt) && aligned16(y),
                MLIIS_ERR_ALIGN, "conv2d_fwd_bnin: pointers must be 16-byte aligned");
  const long long M = (long long)Nimg * H * W;
  MLIIS_REQUIRE(M > 1 && M * ldz * 4 < (1LL << 31) && M * ldy * 4 < (1LL << 31) && M * lda_out * 4 < (1LL << 31) &&
                    (res == nullptr || M * ldr * 4 < (1LL << 31)),
                MLIIS_ERR_UNSUPPORTED, "conv2d_fwd_bnin: operand larger than 2 GiB (32-bit buffer offsets)");
  MLIIS_REQUIRE(precision != MLIIS_PREC_FP8 || (fp8_act_scale > 0.0f && fp8_w_amax != nullptr), MLIIS_ERR_ARG,
                "conv2d_fwd_bnin: fp8 operands need a positive activation scale and the weight tensor's amax");
  MLIIS_REQUIRE(stats_part == nullptr || stats_nblk, MLIIS_ERR_ARG, "conv2d_fwd_bnin: fused statistics need a stats_nblk output");
  StreamPlan sp;
  MLIIS_REQUIRE(stream_plan(M, Cin, Cout, num_cus(), &sp), MLIIS_ERR_UNSUPPORTED,
                "conv2d_fwd_bnin: not a streamed 1x1 shape (mliis_conv2d_fwd_bnin_ok)");
  ConvGemmParams p{z, ldz, Nimg, H, W, Cin, 1, 1, +1, wt, (long long)Cin * Cout, Cin, Cout, y, ldy,
                   nullptr, 0, nullptr, 0, nullptr, 0, nullptr, nullptr, fp8_act_scale, fp8_w_amax};
  p.out_bf16 = ybf;
  p.stats_part = stats_part;
  p.stats_swish = stats_swish;
  p.c_block = y_block;
  const double n = (double)M;
  p.ain = BnFold{bn_part, bn_nblk, 1.0 / n, eps, (float)(1.0 - (double)momentum), 1.0f, mean, rstd, moving_mean, moving_var};
  p.ain_gamma = gamma;
  p.ain_beta = beta;
  p.ain_res = res;
  p.ain_ldr = ldr;
  p.ain_scale = img_scale;
  p.ain_out = a_out;
  p.ain_ldo = lda_out;
  if (stats_nblk) *stats_nblk = 0;
  MLIIS_REQUIRE(launch_stream(sp, p, stream, precision, true), MLIIS_ERR_UNSUPPORTED, "conv2d_fwd_bnin: no instance <%d, %d>", sp.kc, sp.nt);
  MLIIS_CHECK_LAUNCH("conv2d_fwd_bnin");
  if (stats_part != nullptr) *stats_nblk = sp.gx;
  return MLIIS_OK;
}

// dx[M, Cin_out] (ld = lddx) (+)= conv_transpose(dy[M, Cout] (ld = lddy), w) restricted to input channels
// [ci_begin, ci_begin + Cin_out) of a weight tensor w[k,k,Cin_total,Cout].
struct BnbArgs {   // mliis_conv2d_bwd_data_bn: the batch norm whose output gradient this call produces;
                   // mliis_conv2d_bwd_data_gate (mean == NULL): the tensor the output is multiplied with under the squeeze-excite gate
  const float* x;
  int ldx;
  const float *mean, *rstd, *img_scale;
  float* part;
  size_t part_floats;
  int* nblk;
};
static int conv2d_bwd_data_impl(const float* dy, int lddy, const float* w, float* dx, int lddx, int Nimg, int H, int W, int Cin_total,
                                int ci_begin, int Cin_out, int Cout, int ksize, int dil, int accumulate, float* ws, size_t ws_floats,
                                int precision, hipStream_t stream, const BnbArgs* bnb, int dy_dtype, int dx_dtype);

int mliis_conv2d_bwd_data(const float* dy, int lddy, const float* w, float* dx, int lddx, int Nimg, int H, int W, int Cin_total,
                          int ci_begin, int Cin_out, int Cout, int ksize, int dil, int accumulate, float* ws, size_t ws_floats,
                          int precision, int dy_dtype, int dx_dtype, hipStream_t stream) {
  return conv2d_bwd_data_impl(dy, lddy, w, dx, lddx, Nimg, H, W, Cin_total, ci_begin, Cin_out, Cout, ksize, dil, accumulate, ws, ws_floats,
                              precision, stream, nullptr, dy_dtype, dx_dtype);
}

// Same, when dx is the gradient w.r.t. the output of a plain batch norm over bn_x [M, Cin_out] (the project BN of the MBConv block in
// front: efficientnet_model.py:225-236): on the plans that finish their rows in one workgroup (conv1x1_ksplit_k: the 28x28 / 14x14
// maps) the launch also leaves stage 1 of that batch norm's backward -- {sum g, sum g * xhat} per row-group block, g = dx *
// img_scale[image] -- in part [*nblk][2][Cin_out] for mliis_bn_bwd(stage1_part, stage1_nblk).  *nblk == 0: not produced (another
// plan, or part too small): run mliis_bn_bwd without stage-1 partials then.
int mliis_conv2d_bwd_data_bn(const float* dy, int lddy, const float* w, float* dx, int lddx, int Nimg, int H, int W, int Cin_total,
                             int ci_begin, int Cin_out, int Cout, int ksize, int dil, int accumulate, float* ws, size_t ws_floats,
                             int precision, const float* bn_x, int bn_ldx, const float* bn_mean, const float* bn_rstd,
                             const float* bn_img_scale, float* part, size_t part_floats, int* nblk, int dy_dtype, int dx_dtype,
                             hipStream_t stream) {
  MLIIS_REQUIRE(bn_x && bn_mean && bn_rstd && part && nblk, MLIIS_ERR_ARG, "conv2d_bwd_data_bn: null pointer");
  MLIIS_REQUIRE((bn_ldx & 3) == 0 && bn_ldx >= Cin_out && aligned16(bn_x) && aligned16(bn_mean) && aligned16(bn_rstd) && aligned16(part),
                MLIIS_ERR_ARG, "conv2d_bwd_data_bn: batch-norm operands misaligned or too narrow");
  *nblk = 0;
  const BnbArgs b{bn_x, bn_ldx, bn_mean, bn_rstd, bn_img_scale, part, part_floats, nblk};
  return conv2d_bwd_data_impl(dy, lddy, w, dx, lddx, Nimg, H, W, Cin_total, ci_begin, Cin_out, Cout, ksize, dil, accumulate, ws, ws_floats,
                              precision, stream, &b, dy_dtype, dx_dtype);
}

// Same as mliis_conv2d_bwd_data, when dx is the gradient w.r.t. the product gate_x * gate[image] (the squeeze-excite gating in front of
// an MBConv project conv, efficientnet_model.py:251): on the streaming plan (short K, no accumulate) with maps of at least 16 pixels the
// launch also leaves the column sums of dx * gate_x per 16-row group, split by image, in part [*groups][2][Cin_out] -- the gate's
// gradient dgate[n][c] = sum over image n of dx * gate_x, which mliis_se_mlp_bwd(dgate_row_groups = *groups) folds itself.
// *groups == 0: not produced (another plan, or part too small): run mliis_colsum then.
int mliis_conv2d_bwd_data_gate(const float* dy, int lddy, const float* w, float* dx, int lddx, int Nimg, int H, int W, int Cin_total,
                               int ci_begin, int Cin_out, int Cout, int ksize, int dil, float* ws, size_t ws_floats, int precision,
                               const float* gate_x, int gate_ldx, float* part, size_t part_floats, int* groups, int dy_dtype, int dx_dtype,
                               hipStream_t stream) {   // (gate_x has dx's storage type: a1 beside the gradient of a1 * gate)
  MLIIS_REQUIRE(gate_x && part && groups, MLIIS_ERR_ARG, "conv2d_bwd_data_gate: null pointer");
  MLIIS_REQUIRE((gate_ldx & 3) == 0 && gate_ldx >= Cin_out && aligned16(gate_x) && aligned16(part), MLIIS_ERR_ARG,
                "conv2d_bwd_data_gate: operands misaligned or too narrow");
  *groups = 0;
  const BnbArgs b{gate_x, gate_ldx, nullptr, nullptr, nullptr, part, part_floats, groups};
  return conv2d_bwd_data_impl(dy, lddy, w, dx, lddx, Nimg, H, W, Cin_total, ci_begin, Cin_out, Cout, ksize, dil, 0, ws, ws_floats, precision,
                              stream, &b, dy_dtype, dx_dtype);
}

static int conv2d_bwd_data_impl(const float* dy, int lddy, const float* w, float* dx, int lddx, int Nimg, int H, int W, int Cin_total,
                                int ci_begin, int Cin_out, int Cout, int ksize, int dil, int accumulate, float* ws, size_t ws_floats,
                                int precision, hipStream_t stream, const BnbArgs* bnb, int dy_dtype, int dx_dtype) {
  int rc = conv_check("conv2d_bwd_data", Nimg, H, W, Cin_out, Cout, ksize, dil);
  if (rc) return rc;
  if ((rc = prec_check("conv2d_bwd_data", precision))) return rc;
  const int dx_block = (dx_dtype >> 8) & 0xff;   // MLIIS_DT_BLOCKED(v): dx in the group-blocked layout (as y in mliis_conv2d_fwd)
  dx_dtype &= 0xff;
  MLIIS_REQUIRE(dx_block == 0 || ((dx_block == 2 || dx_block == 4) && dx_dtype == MLIIS_DT_F32 && Cin_out % dx_block == 0 && ci_begin == 0 &&
                                  Cin_out == Cin_total),
                MLIIS_ERR_ARG, "conv2d_bwd_data: a group-blocked output takes v = 2 | 4 dividing the channel count, all channels and fp32 storage");
  const bool dybf = dy_dtype == MLIIS_DT_BF16, dxbf = dx_dtype == MLIIS_DT_BF16;   // (as in mliis_conv2d_fwd)
  MLIIS_REQUIRE((dy_dtype == MLIIS_DT_F32 || dybf) && (dx_dtype == MLIIS_DT_F32 || dxbf), MLIIS_ERR_ARG, "conv2d_bwd_data: bad storage type");
  if (dybf || dxbf) {
    MLIIS_REQUIRE(precision != MLIIS_PREC_FP32 && ksize == 1 && !(dxbf && accumulate), MLIIS_ERR_UNSUPPORTED,
                  "conv2d_bwd_data: bf16 tensors need bf16 operands, a 1x1 conv and (bf16 output) no accumulate");
    ws = nullptr;
    ws_floats = 0;
  }
  MLIIS_REQUIRE(dy && w && dx, MLIIS_ERR_ARG, "conv2d_bwd_data: null pointer");
  MLIIS_REQUIRE(ci_begin >= 0 && ci_begin + Cin_out <= Cin_total, MLIIS_ERR_ARG, "conv2d_bwd_data: channel window out of range");
  MLIIS_REQUIRE((lddy & 3) == 0 && lddy >= Cout && (lddx & 3) == 0 && lddx >= Cin_out && (ci_begin & 3) == 0, MLIIS_ERR_ARG,
                "conv2d_bwd_data: bad leading dimensions");
  MLIIS_REQUIRE(aligned16(dy) && aligned16(w) && aligned16(dx), MLIIS_ERR_ALIGN, "conv2d_bwd_data: pointers must be 16-byte aligned");
  long long M = (long long)Nimg * H * W;
  MLIIS_REQUIRE(M * lddy * 4 < (1LL << 31) && (long long)ksize * ksize * Cin_total * Cout * 4 < (1LL << 31), MLIIS_ERR_UNSUPPORTED,
                "conv2d_bwd_data: operand larger than 2 GiB (32-bit buffer offsets)");
  GemmPlan g = plan_gemm(M, Cin_out, Cout, ksize * ksize, num_cus(), ws != nullptr);
  if (precision == MLIIS_PREC_FP8) precision = MLIIS_PREC_BF16;   // fp8 mode: forward 1x1 convs in e4m3, the backward passes in bf16
  ConvGemmParams p{dy, lddy, Nimg, H, W, Cout, ksize * ksize, dil, -1, w + (long long)ci_begin * Cout, (long long)Cin_total * Cout,
                   Cout, Cin_out, dx, lddx, nullptr, accumulate, nullptr, g.chunks_per_split, nullptr, 0, nullptr, nullptr, 1.0f, nullptr};
  p.a_bf16 = dybf;
  p.out_bf16 = dxbf;
  p.side_bf16 = dxbf;
  {  // short-K 1x1 convs (backward-data of the MBConv project convs): barrier-free streaming kernel
    StreamPlan sp;
    if (ksize == 1 && !accumulate && !dybf && M * lddx * 4 < (1LL << 31) && stream_plan(M, Cout, Cin_out, num_cus(), &sp)) {
      const bool gate = bnb != nullptr && bnb->mean == nullptr;
      const bool with_bn = bnb != nullptr && !gate && (size_t)sp.gx * 2 * Cin_out <= bnb->part_floats;
      const bool with_gate = gate && (long long)H * W >= 16 && (size_t)sp.row_groups * 2 * Cin_out <= bnb->part_floats;
      if (with_bn) {   // + stage 1 of the consumer batch norm's backward (per-wave sums, folded per block like the forward statistics)
        p.stats_part = bnb->part;
        p.bnb_x = bnb->x;
        p.bnb_ldx = bnb->ldx;
        p.bnb_mean = bnb->mean;
        p.bnb_rstd = bnb->rstd;
        p.bnb_scale = bnb->img_scale;
      }
      if (with_gate) {   // + per-row-group sums of dx * gate_x (the squeeze-excite gate's gradient)
        p.gp_x = bnb->x;
        p.gp_ldx = bnb->ldx;
        p.gp_part = bnb->part;
      }
      p.c_block = dx_block;
      if (launch_stream(sp, p, stream, precision)) {
        MLIIS_CHECK_LAUNCH("conv2d_bwd_data_stream");
        if (with_bn) *bnb->nblk = sp.gx;
        if (with_gate) *bnb->nblk = sp.row_groups;
        return MLIIS_OK;
      }
      p.stats_part = nullptr;
      p.bnb_x = nullptr;
      p.gp_part = nullptr;
      p.c_block = 0;
    }
    MLIIS_REQUIRE(dx_block == 0, MLIIS_ERR_UNSUPPORTED, "conv2d_bwd_data: a group-blocked output needs the streamed 1x1 plan");
    // long-K 1x1 convs on small maps (backward-data of the MBConv expand convs): K split inside the workgroup, one launch
    if (ksize == 1 && !dxbf && M * lddx * 4 < (1LL << 31) && M * lddy * 4 < (1LL << 31) && ksplit_plan(M, Cout, Cin_out, num_cus(), &sp)) {
      const bool with_bn = bnb != nullptr && bnb->mean != nullptr && (size_t)sp.gx * 2 * Cin_out <= bnb->part_floats;
      if (with_bn) {   // + stage 1 of the consumer batch norm's backward from the finishing threads
        p.stats_part = bnb->part;
        p.bnb_x = bnb->x;
        p.bnb_ldx = bnb->ldx;
        p.bnb_mean = bnb->mean;
        p.bnb_rstd = bnb->rstd;
        p.bnb_scale = bnb->img_scale;
      }
      if (launch_ksplit(sp, p, stream, precision)) {
        MLIIS_CHECK_LAUNCH("conv2d_bwd_data_ksplit");
        if (with_bn) *bnb->nblk = sp.gx;
        return MLIIS_OK;
      }
      p.stats_part = nullptr;
      p.bnb_x = nullptr;
    }
  }
  if (g.gz > 1) {
    size_t need = (size_t)g.gz * M * Cin_out;
    MLIIS_REQUIRE(need <= ws_floats && aligned16(ws) && (lddx & 3) == 0 && aligned16(dx), MLIIS_ERR_WORKSPACE,
                  "conv2d_bwd_data: split-K workspace too small (%zu needed, %zu given) or unaligned output", need, ws_floats);
    p.partial = ws;
  }
  float* sk_slab = nullptr;
  if (g.sk_parts > 0 && ws != nullptr && aligned16(ws) && g.sk_slab_floats() <= ws_floats && (lddx & 3) == 0) sk_slab = ws;
  launch_gemm(g, p, precision, stream, sk_slab);
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
// total_tiles > 0: the sum over the descriptors of taps * ceil(Cin / 32) * ceil(Cout / 32) (the caller built the table, it knows):
// one workgroup per tile.  0: a 224 x ndesc grid whose workgroups stride over their descriptor's tiles.
static int weight_shadows(const char* name, const float* src, float* dst, const int* desc, int ndesc, long long total_tiles, float* amax,
                          void* x3_images, const long long* x3_desc, int x3_ndesc, int x3_blocks, hipStream_t stream,
                          const RngArgs* rng_in = nullptr) {
  MLIIS_REQUIRE(src && dst && desc && ndesc > 0 && total_tiles >= 0 && total_tiles < (1LL << 30), MLIIS_ERR_ARG, "%s: bad arguments", name);
  MLIIS_REQUIRE(x3_blocks == 0 || (total_tiles > 0 && x3_images && x3_desc && x3_ndesc >= 1 && x3_ndesc <= 64 && x3_blocks > 0 &&
                                   aligned16(src) && aligned16(x3_images) && aligned16(x3_desc)),
                MLIIS_ERR_ARG, "%s: the weight images need the one-tile-per-workgroup form, a table of 1..64 rows and 16-byte aligned pointers", name);
  if (amax != nullptr) {   // (a memset node when captured into a graph)
    hipError_t e = hipMemsetAsync(amax, 0, (size_t)ndesc * sizeof(float), stream);
    MLIIS_REQUIRE(e == hipSuccess, MLIIS_ERR_LAUNCH, "%s: memset failed: %s", name, hipGetErrorString(e));
  }
  const X3PackArgs x3{reinterpret_cast<char*>(x3_images), x3_desc, x3_ndesc, x3_blocks, (int)total_tiles};
  RngArgs rng{};
  if (rng_in != nullptr) {
    MLIIS_REQUIRE(total_tiles > 0, MLIIS_ERR_ARG, "%s: the mask jobs need the one-tile-per-workgroup form", name);
    rng = *rng_in;
    rng.first_block = (int)total_tiles + (x3_blocks + 1) / 2;
  }
  if (total_tiles > 0)
    hipLaunchKernelGGL(transpose_weights_k, dim3((unsigned)(total_tiles + (x3_blocks + 1) / 2 + rng.blocks)), dim3(256), 0, stream, src, dst, desc,
                       reinterpret_cast<unsigned*>(amax), ndesc, x3, rng);
  else
    hipLaunchKernelGGL(transpose_weights_k, dim3(224, ndesc), dim3(256), 0, stream, src, dst, desc, reinterpret_cast<unsigned*>(amax),
                       0, x3, rng);   // (the largest tensor has ~2000 tiles)
  MLIIS_CHECK_LAUNCH(name);
  return MLIIS_OK;
}

int mliis_transpose_weights(const float* src, float* dst, const int* desc, int ndesc, long long total_tiles, float* amax, hipStream_t stream) {
  return weight_shadows("transpose_weights", src, dst, desc, ndesc, total_tiles, amax, nullptr, nullptr, 0, 0, stream);
}

// mliis_transpose_weights + mliis_x3_pack_weights (same arguments) in ONE launch: both read the weight arena once per optimizer step and
// depend on nothing else of the step.  total_tiles must be > 0 (one workgroup per transpose tile, the pack blocks behind them).
int mliis_weight_shadows(const float* src, float* dst, const int* desc, int ndesc, long long total_tiles, float* amax, void* x3_images,
                         const long long* x3_desc, int x3_ndesc, int x3_blocks, hipStream_t stream) {
  MLIIS_REQUIRE(x3_blocks > 0, MLIIS_ERR_ARG, "weight_shadows: no image blocks (use mliis_transpose_weights)");
  return weight_shadows("weight_shadows", src, dst, desc, ndesc, total_tiles, amax, x3_images, x3_desc, x3_ndesc, x3_blocks, stream);
}

// ... and the masks of the step (mliis_rng_masks, same arguments) in the same launch: x3_blocks may be 0 (no weight images).
int mliis_weight_shadows_rng(const float* src, float* dst, const int* desc, int ndesc, long long total_tiles, float* amax, void* x3_images,
                             const long long* x3_desc, int x3_ndesc, int x3_blocks, unsigned* rng_state, int njobs, float* const* outs,
                             const long long* numels, const float* keep, const float* const* keeps, const int* row_len, const int* floor_form,
                             hipStream_t stream) {
  MLIIS_REQUIRE(rng_state, MLIIS_ERR_ARG, "weight_shadows_rng: null generator state");
  RngArgs rng{};
  rng.state = rng_state;
  rng.blocks = rng_make_jobs("weight_shadows_rng", njobs, outs, numels, keep, keeps, row_len, floor_form, &rng.jobs);
  if (rng.blocks < 0) return rng.blocks;
  return weight_shadows("weight_shadows_rng", src, dst, desc, ndesc, total_tiles, amax, x3_images, x3_desc, x3_ndesc, x3_blocks, stream, &rng);
}

size_t mliis_conv2d_bwd_filter_workspace_floats(int Nimg, int H, int W, int Cin, int Cout, int ksize) {
  long long M = (long long)Nimg * H * W;
  FilterPlan f = plan_filter(M, Cin, Cout, ksize * ksize, num_cus());
  return (size_t)f.gz * ksize * ksize * Cin * Cout;
}

// The tiling of a filter-gradient call: plan[0..6] = {ci-block factor TMF, column tiles NT, multitap, gx, gy, gz (= slabs),
// rows_per_split} -- what a caller needs to lay out the descriptor table of mliis_conv2d_bwd_filter_batched.
int mliis_conv2d_bwd_filter_plan(int Nimg, int H, int W, int Cin, int Cout, int ksize, int* plan) {
  int rc = conv_check("conv2d_bwd_filter_plan", Nimg, H, W, Cin, Cout, ksize, 1);
  if (rc) return rc;
  MLIIS_REQUIRE(plan, MLIIS_ERR_ARG, "conv2d_bwd_filter_plan: null pointer");
  const FilterPlan f = plan_filter((long long)Nimg * H * W, Cin, Cout, ksize * ksize, num_cus());
  plan[0] = f.tmf; plan[1] = f.nt; plan[2] = f.multitap; plan[3] = f.gx; plan[4] = f.gy; plan[5] = f.gz; plan[6] = f.rows_per_split;
  return MLIIS_OK;
}

// nprob filter-gradient problems that share one kernel instantiation (TMF, NT, x_scale or not) as ONE launch of `blocks` workgroups;
// every problem leaves its gz slabs in its own workspace region exactly as mliis_conv2d_bwd_filter(dw = NULL) does.  desc: DEVICE
// table int64 [nprob][16], layout at conv_filter_grad2_batched_k; the caller owns it (it is read by the kernel, not by the host).
int mliis_conv2d_bwd_filter_batched(const long long* desc, int nprob, int blocks, int tmf, int nt, int has_scale, int precision,
                                    hipStream_t stream) {
  // MLIIS_PREC_F32X3: the groups of 128-channel tiles (TMF = 2, at least 4 column tiles, no x_scale) as fp32-equivalent split products
  // on the bf16 matrix cores (conv_x3.hip: conv_filter_x3_batched_k); every other group runs the native fp32 instruction
  bool x3_done = false;
  if (precision == MLIIS_PREC_F32X3) {
    precision = MLIIS_PREC_FP32;
    MLIIS_REQUIRE(desc && nprob >= 1 && nprob <= 64 && blocks >= 1 && aligned16(desc), MLIIS_ERR_ARG, "conv2d_bwd_filter_batched: bad table");
    // tmf == 4 (this precision only): the table was built for 256-channel tiles of the problems with more than 128 input channels
    // (gx = taps x ceil(C / 256) for those, first blocks accordingly: ops.FilterBatch) -- conv_filter_x3_batched_k stages dY once per
    // 256 input channels there
    if ((tmf == 2 || tmf == 4) && !has_scale) x3_done = launch_filter_batched_x3(nt, desc, nprob, blocks, tmf == 4 ? 256 : 128, stream);
    MLIIS_REQUIRE(tmf != 4 || x3_done, MLIIS_ERR_ARG, "conv2d_bwd_filter_batched: TMF = 4 exists for MLIIS_PREC_F32X3 with 4..8 column tiles and no x_scale only");
  }
  int rc = prec_check("conv2d_bwd_filter_batched", precision);
  if (rc) return rc;
  MLIIS_REQUIRE(desc && nprob >= 1 && nprob <= 64 && blocks >= 1, MLIIS_ERR_ARG, "conv2d_bwd_filter_batched: bad table (1..64 problems)");
  MLIIS_REQUIRE(aligned16(desc), MLIIS_ERR_ALIGN, "conv2d_bwd_filter_batched: table must be 16-byte aligned");
  const bool ok = x3_done ? true
                  : precision != MLIIS_PREC_FP32 ? launch_filter_batched_bf16(tmf, nt, has_scale != 0, desc, nprob, blocks, stream)
                                                 : launch_filter_batched_t<false>(tmf, nt, has_scale != 0, desc, nprob, blocks, stream);
  MLIIS_REQUIRE(ok, MLIIS_ERR_ARG, "conv2d_bwd_filter_batched: no instantiation TMF = %d, NT = %d", tmf, nt);
  MLIIS_CHECK_LAUNCH("conv2d_bwd_filter_batched");
  return MLIIS_OK;
}

// dw[k,k,Cin,Cout] (+)= sum_pixels x[pixel + tap offset, ci] * dy[pixel, co]
int mliis_conv2d_bwd_filter(const float* x, int ldx, const float* x_scale, const float* dy, int lddy, float* dw, int Nimg, int H, int W,
                            int Cin_total, int ci_begin, int Cin, int Cout, int ksize, int dil, int accumulate, float* ws,
                            size_t ws_floats, int precision, hipStream_t stream) {
  int rc = conv_check("conv2d_bwd_filter", Nimg, H, W, Cin, Cout, ksize, dil);
  if (rc) return rc;
  if ((rc = prec_check("conv2d_bwd_filter", precision))) return rc;
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
  FilterGradParams p{x, ldx, Nimg, H, W, Cin, ksize * ksize, dil, dy, lddy, Cout, ws, f.rows_per_split, x_scale, f.multitap};
  launch_filter(f, p, precision, stream);
  MLIIS_CHECK_LAUNCH("conv2d_bwd_filter");
  MLIIS_REQUIRE(ci_begin >= 0 && ci_begin + Cin <= Cin_total, MLIIS_ERR_ARG, "conv2d_bwd_filter: input-channel window out of range");
  if (dw == nullptr) return MLIIS_OK;
  hipLaunchKernelGGL(fold_flat_k, dim3(ceil_div((long long)total, kFoldX)), dim3(kFoldX, kFoldY), 0, stream, ws, f.gz, (long long)total, 1.0f, dw,
                     accumulate, (long long)Cin * Cout, (long long)Cin_total * Cout, (long long)ci_begin * Cout);
  MLIIS_CHECK_LAUNCH("conv2d_bwd_filter_reduce");
  return MLIIS_OK;
}
}
