// fp32-EQUIVALENT dense convolutions on the bf16 matrix cores: the long-K convs of the RSD decoder (models/efficientlab.py:185-190,
// 218-224: 3x3 and 3x3-dilated over 136..224 channels, K = 1008..2016) forward and backward-data.
//
// Every fp32 operand value is written EXACTLY as three bf16 terms x = hi + mid + lo (hi = bf16(x), mid = bf16(x - hi), lo = bf16(x -
// hi - mid): 3 x 8 = 24 significand bits) and six of the nine term products (all but mid*lo, lo*mid, lo*lo: below 2^-24 |a b|) run on
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation: 6 matrix instructions of 16 cycles per 16x16x32 block instead of 8 fp32
// instructions (v_mfma_f32_16x16x4_f32) of 32 -- 2.67x less matrix-pipe time for the same products.  With the matrix pipe that short,
// everything else in a chunk of the fp32 kernel (conv_gemm_tile: both operands through LDS, every wave re-reading the whole B tile,
// the split itself) became the bound (the split done inside that kernel: 120 -> 95 us, and 82 us with NO split arithmetic at all), so
// this kernel is laid out for the split product (how each choice measured: profiles/r05_notes.md):
//   * B (the weights) is split ONCE per inner step by x3_pack_k into an image that is already the LDS image of the kernel:
//     [K chunk of 32][16-column tile][term 3][lane group g 4][column 16][8 bf16] = 3072 bytes per (chunk, column tile); K is the
//     flattened (tap, channel) index without padding between taps (round 6).  A chunk of a workgroup's B
//     tile is ONE contiguous run of NT x 3072 bytes: linear 16-byte loads, linear ds_write_b128, conflict-free ds_read_b128
//     fragments, no arithmetic;
//   * A (the activations) does not go through LDS at all: a wave owns 32 rows (two 16-row blocks) and loads its own A fragments from
//     memory in MFMA operand layout (lane (row, g): k = 4g..4g+3 and 16+4g..16+4g+3 of the chunk: two 16-byte loads per row block,
//     each instruction 16 rows x 64 contiguous bytes), two chunks ahead, splits them in registers BETWEEN its matrix instructions
//     (each element is split exactly once on the chip) and multiplies them with every B fragment: 2 x NT x 6 MFMAs per 3 NT reads;
//   * workgroup = 8 waves = 256 rows x 16 NT columns sharing the B tile, one per CU; the barrier of a chunk waits for LDS only
//     (__syncthreads() also drains the wave's global loads: the prefetched chunks would be pinned to the chunk they are requested in);
//   * the K iterations of all tiles are cut stream-K fashion into equal contiguous parts, one per CU (98 - 294 tiles of 256 rows on 256
//     CUs: whole tiles would leave a quarter of the chip idle or take two rounds); x3_fixup_k adds a tile's segments in a fixed
//     order, applies bias / border bias / accumulate and emits the BN statistics: deterministic.
// Results are within the fp32 tolerances of every conv test (rel 2e-5 of max-abs forward, 1e-4 backward): error against float64
// 0.6-1.5e-6 of max-abs, against 1.4-2.0e-6 for the native fp32 instruction (tools/x3_probe.py) -- the 16x16x32 instruction adds its 32
// products in one pass, the fp32 path rounds after each of its eight 16x16x4 steps.
#include "conv_gemm_kernels.hpp"

namespace mliis {

// (kX3Block = 3072 bytes of one (chunk, 16-column tile) block of a weight image, x3_pack_block: conv_gemm_kernels.hpp -- the weight
//  shadows of a step, transposes and images, are one launch: mliis_weight_shadows)
constexpr int kX3BM = 256;       // rows of a workgroup tile (eight waves of 32)
#ifndef X3_PF
#define X3_PF 1
#endif
constexpr int kX3PF = X3_PF;     // column tiles of B fragments requested ahead of their products

// ------------------------------------------------------------------------------------------------ weight images
// desc rows (int64 [ndesc][8]): {source offset (floats) in `theta`, taps, Cin_total, Cout, ci_begin, Cin (window), mode | first block
// << 8, image offset (bytes)}.  mode 0 (forward): B[n = co][k = tap * Cin + c] = w[tap][ci_begin + c][co];  mode 1 (backward-data):
// B[n = ci - ci_begin][k = tap * Cout + co] = w[tap][ci][co].  One workgroup (128 threads) per (chunk, 16-column tile) block.
__global__ __launch_bounds__(128) void x3_pack_k(const float* __restrict__ theta, char* __restrict__ images, const long long* __restrict__ desc,
                                                  int ndesc) {
  x3_pack_block(theta, images, desc, ndesc, blockIdx.x, threadIdx.x);
}

// ------------------------------------------------------------------------------------------------ the tile
struct X3Params {
  ConvGemmParams g;      // A, lda, Nimg, H, W, C (K per tap), ntaps, dil, sign, Nout, Cmat, ldc, bias, accumulate, stats_part, stats_swish,
                         // border_bias as in conv_gemm_tile; B / ldb / b_tap_stride are not used
  const char* image;     // weight image of this conv and direction (x3_pack_k)
  int ncol16;            // its 16-column tiles
#ifdef X3_CLK
  unsigned long long* dbg;   // [workgroups][4]: {shader clock, 100 MHz clock} at the start and the end of the first segment's main loop
#endif
};

template <int NT>
struct X3Sm {
  static constexpr int BN = 16 * NT;
  static constexpr int B_BYTES = NT * kX3Block;
  static constexpr int STAGES = 2;
  static constexpr int BYTES = STAGES * B_BYTES + 64;   // the chunks of the B tile + a scratch slot for surplus store lanes
};

// One output tile (256-row tile bx, column tile by) over the K chunks [it0, it1): the raw partial tile to pdst[row * BN + column] (a
// stream-K segment slab; x3_fixup_k adds a tile's segments and applies bias / border bias / accumulate / statistics).
// Wave w of the eight owns rows 32 w .. 32 w + 31 of the tile.
template <int NT>
__device__ __forceinline__ void conv_x3_tile(const X3Params& q, char* __restrict__ sm, unsigned bx, int by, int it0, int it1,
                                             float* __restrict__ pdst) {
  constexpr int BN = 16 * NT;
  constexpr int B_BYTES = X3Sm<NT>::B_BYTES;
  constexpr int NB = (NT * 192 + 511) / 512;   // 16-byte pieces of a B chunk per thread
  const ConvGemmParams& p = q.g;
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const long long M = (long long)p.Nimg * p.H * p.W;
  const long long m0 = (long long)bx * kX3BM;
  const int n0 = by * BN;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, kBufRecords, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)q.image, 0, kBufRecords, 0x00020000);

  // ---- this lane's two rows (row blocks rb = 0, 1 of the wave's 32 rows): byte offset of the row and the 9-bit mask of the taps whose
  // source pixel lies inside the image.  K is the FLATTENED (tap, channel) index, tap-major, cut into chunks of 32 with no padding
  // between taps (round 6: a tap of 112 or 136 channels used to be padded to 128 / 160 -- 12-18 % of the matrix instructions multiplied
  // zeros); a lane's two k quads of a chunk (k = 32 chunk + 16 qd + 4 g ...) may therefore belong to different taps: tap, channel and
  // source offset are per-lane values, advanced incrementally (Cred >= 32: at most one tap boundary per advance; Cred % 4 == 0: a quad
  // never straddles one)
  const int wrow = wave * 32;   // first row of this wave inside the tile
  unsigned a_row[2], a_taps[2];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb) {
    const long long m = m0 + wrow + rb * 16 + l15;
    a_taps[rb] = 0;
    a_row[rb] = 0;
    if (m < M) {
      a_row[rb] = (unsigned)(m * p.lda * 4);
      if (p.ntaps > 1) {
        const int HWp = p.H * p.W;
        const int n = (int)(m / HWp);
        const int rem = (int)(m - (long long)n * HWp);
        const int h = rem / p.W, w_ = rem - h * p.W;
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
          const int dh = (tp / 3 - 1) * p.dil * p.sign, dw = (tp % 3 - 1) * p.dil * p.sign;
          if ((unsigned)(h + dh) < (unsigned)p.H && (unsigned)(w_ + dw) < (unsigned)p.W) a_taps[rb] |= 1u << tp;
        }
      } else {
        a_taps[rb] = 1u;
      }
    }
  }
  // ---- the next A chunk to LOAD: per lane and k quad its tap l_tap and channel l_c; the next B chunk to load: byte offset s_boff
  // (uniform per wave; past the last chunk the SAME chunk is requested again -- cache hits, never used -- so the hot path has no predicate)
  int l_tap[2], l_c[2];
#pragma unroll
  for (int qd = 0; qd < 2; ++qd) {
    const int k0 = it0 * 32 + qd * 16 + g * 4;
    l_tap[qd] = k0 / p.C;
    l_c[qd] = k0 - l_tap[qd] * p.C;
  }
  int s_aleft = it1 - it0, s_bleft = it1 - it0;
  int ncol_here = q.ncol16 - by * NT;
  if (ncol_here > NT) ncol_here = NT;
  const unsigned b_lim = (unsigned)ncol_here * 192u;   // 16-byte pieces of a chunk present in the image (beyond: zeros)
  const unsigned b_step = (unsigned)q.ncol16 * kX3Block;
  unsigned b_voff[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) b_voff[i] = ((unsigned)t + 512u * i < b_lim) ? (unsigned)(by * NT) * kX3Block + ((unsigned)t + 512u * i) * 16u : kOob;
  unsigned s_boff = (unsigned)it0 * b_step;

  float4 ra[2][2];    // [row block][quad]: the next A chunk, fp32
  float4 ra2[2][2];   // the A chunk after that
  u32x4 rbv[NB];      // this thread's pieces of the next B chunk
  const int tap_scale = p.dil * p.sign;
  auto load_a_into = [&](float4 (&ra)[2][2]) {
    const int adv = s_aleft > 1 ? 1 : 0;   // (branch-free advance)
    s_aleft -= adv;
#pragma unroll
    for (int qd = 0; qd < 2; ++qd) {
      const int tap = l_tap[qd];
      const int th = p.ntaps > 1 ? (tap * 11) >> 5 : 1, tw = p.ntaps > 1 ? tap - 3 * th : 1;   // (tap / 3 for tap <= 9)
      const int dh = (th - 1) * tap_scale, dw = (tw - 1) * tap_scale;
      const unsigned koff = (unsigned)(((dh * p.W + dw) * p.lda + l_c[qd]) * 4);   // (two's complement: the sum with a row's offset is exact)
      const unsigned tapbit = tap < p.ntaps ? 1u << tap : 0u;                       // (k beyond the last tap: zeros)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) ra[rb][qd] = buf_ld4(rA, (a_taps[rb] & tapbit) != 0 ? a_row[rb] + koff : kOob);
      const int c_next = l_c[qd] + 32;
      const int wrap = c_next >= p.C ? 1 : 0;
      l_c[qd] = adv ? (wrap ? c_next - p.C : c_next) : l_c[qd];
      l_tap[qd] += adv & wrap;
    }
  };
  auto load_b_into = [&](u32x4 (&rbv)[NB]) {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      rbv[i] = __builtin_amdgcn_raw_buffer_load_b128(rB, (int)b_voff[i], (int)s_boff, 0);
    }
    const int adv = s_bleft > 1 ? 1 : 0;
    s_bleft -= adv;
    s_boff += adv ? b_step : 0u;
  };
  auto store_b_from = [&](char* buf, const u32x4 (&rbv)[NB]) {
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int idx = t + 512 * i;
      char* d = (512 * (i + 1) <= NT * 192 || idx < NT * 192) ? buf + idx * 16 : sm + X3Sm<NT>::STAGES * B_BYTES + (t & 3) * 16;   // surplus lanes: scratch
      *reinterpret_cast<u32x4*>(d) = rbv[i];
    }
  };
  auto load_b = [&]() { load_b_into(rbv); };
  auto store_b = [&](char* buf) { store_b_from(buf, rbv); };
  bf16x8 a3[3][2], a3n[3][2];
  auto split_into = [&](const float4 (&ra)[2][2], bf16x8 (&a3)[3][2]) {
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      uint2 h0, m0_, l0, h1, m1, l1;
      split3(ra[rb][0], h0, m0_, l0);
      split3(ra[rb][1], h1, m1, l1);
      a3[0][rb] = __builtin_bit_cast(bf16x8, (u32x4){h0.x, h0.y, h1.x, h1.y});
      a3[1][rb] = __builtin_bit_cast(bf16x8, (u32x4){m0_.x, m0_.y, m1.x, m1.y});
      a3[2][rb] = __builtin_bit_cast(bf16x8, (u32x4){l0.x, l0.y, l1.x, l1.y});
    }
  };
  f32x4 acc[2][NT];
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[rb][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // One chunk's products out of LDS buffer `buf` and the registers a3; the fragments of column tile jn + 1 are requested before the
  // products of tile jn.  The split of the A chunk multiplied NEXT (rs -> a3n) is cut into its four 16-byte pieces and each piece is
  // placed behind the twelve matrix instructions of one column tile, fenced, so that the vector work is spread over the chunk
  // instead of following it (a matrix instruction leaves the SIMD's vector issue free for half of its 16 cycles, and the partner
  // wave's matrix instructions fill the pipe meanwhile)
  auto compute_split = [&](const char* buf, const float4 (&rs)[2][2]) {
    const char* pb = buf + g * 256 + l15 * 16;
    // B fragments kX3PF column tiles ahead of their products (-DX3_PF=2 | 3 measured: 79.0 / 80.7 against 79.9 us with one tile ahead on
    // the 224 -> 112 conv -- the LDS round trip is not what the slower wave of a SIMD waits for)
    constexpr int PF = kX3PF;
    bf16x8 b3[PF + 1][3];
    uint2 sh[2][2], sm_[2][2], sl[2][2];
#pragma unroll
    for (int d = 0; d < PF; ++d)
      if (d < NT) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) b3[d][pl] = *reinterpret_cast<const bf16x8*>(pb + d * kX3Block + pl * 1024);
      }
#pragma unroll
    for (int jn = 0; jn < NT; ++jn) {
#if defined(X3_ABL) && (X3_ABL & 2)   // ablation: every column tile multiplies the FIRST tile's fragments (no further LDS reads)
      if (jn + PF < NT && jn + PF < 2) {
#else
      if (jn + PF < NT) {
#endif
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) b3[(jn + PF) % (PF + 1)][pl] = *reinterpret_cast<const bf16x8*>(pb + (jn + PF) * kX3Block + pl * 1024);
      }
#define X3_MM(PA, PB)                                                                                                           \
  acc[0][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[PA][0], b3[jn % (PF + 1)][PB], acc[0][jn], 0, 0, 0);                \
  acc[1][jn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[PA][1], b3[jn % (PF + 1)][PB], acc[1][jn], 0, 0, 0);
#if defined(X3_ABL) && (X3_ABL & 1)   // ablation: one matrix instruction per accumulator instead of six
      X3_MM(0, 0)
#else
      X3_MM(2, 0) X3_MM(0, 2) X3_MM(1, 1) X3_MM(1, 0) X3_MM(0, 1) X3_MM(0, 0)
#endif
#undef X3_MM
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if ((k * NT) / 4 == jn) {
          uint2& h = sh[k >> 1][k & 1];
          uint2& m = sm_[k >> 1][k & 1];
          uint2& l = sl[k >> 1][k & 1];
#ifdef X3_NOSPLIT   // ablation: no split arithmetic (wrong numbers: what the loop costs without its vector work)
          h = make_uint2(__float_as_uint(rs[k >> 1][k & 1].x), __float_as_uint(rs[k >> 1][k & 1].y));
          m = make_uint2(__float_as_uint(rs[k >> 1][k & 1].z), __float_as_uint(rs[k >> 1][k & 1].w));
          l = h;
#else
          split3(rs[k >> 1][k & 1], h, m, l);
#endif
          // (an empty statement that "uses" the piece here: without it the optimiser sinks the arithmetic to the first real use, the
          // assembly of a3n behind the last matrix instruction)
          asm volatile("" : "+v"(h.x), "+v"(h.y), "+v"(m.x), "+v"(m.y), "+v"(l.x), "+v"(l.y));
        }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      a3n[0][rb] = __builtin_bit_cast(bf16x8, (u32x4){sh[rb][0].x, sh[rb][0].y, sh[rb][1].x, sh[rb][1].y});
      a3n[1][rb] = __builtin_bit_cast(bf16x8, (u32x4){sm_[rb][0].x, sm_[rb][0].y, sm_[rb][1].x, sm_[rb][1].y});
      a3n[2][rb] = __builtin_bit_cast(bf16x8, (u32x4){sl[rb][0].x, sl[rb][0].y, sl[rb][1].x, sl[rb][1].y});
    }
  };
#ifdef X3_CLK
  const unsigned long long ck0 = __builtin_amdgcn_s_memtime(), cr0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long c_comp = 0, c_skel = 0, c_bar = 0, c_t = ck0;
#define X3_TICK(acc_) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); acc_ += n_ - c_t; c_t = n_; } while (0)
#else
#define X3_TICK(acc_) do { } while (0)
#endif
  // ---- main loop: every wave multiplies chunk j and, between its own matrix instructions, splits the A chunk it multiplies next; the
  // A chunk after that and the B chunk after next are in flight meanwhile (requested a whole chunk before their first use)
  const int nper = it1 - it0;
  load_a_into(ra);     // A(it0)
  load_b();            // B(it0)
  store_b(sm);
  load_b();            // B(it0 + 1)
  split_into(ra, a3);
  load_a_into(ra);     // A(it0 + 1)
  load_a_into(ra2);    // A(it0 + 2)
  lds_barrier();
  auto chunk = [&](char* cur, char* nxt, float4 (&rs)[2][2]) {   // rs: the registers holding A of the chunk after this one
    compute_split(cur, rs);
    X3_TICK(c_comp);
#if !(defined(X3_ABL) && (X3_ABL & 8))   // (ablation 8: the B tile of the first chunks forever -- no B loads, no LDS stores)
    store_b(nxt);        // B of the next chunk (requested one chunk ago)
    load_b();            // B of the chunk after next
#endif
#if !(defined(X3_ABL) && (X3_ABL & 4))   // (ablation 4: the A registers of the prologue forever -- no A loads)
    load_a_into(rs);     // A two chunks after the one just split
#endif
    X3_TICK(c_skel);
    lds_barrier();
    X3_TICK(c_bar);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) a3[pl][rb] = a3n[pl][rb];
  };
  {
    int it = 0;
    for (; it + 2 <= nper; it += 2) {
      chunk(sm, sm + B_BYTES, ra);
      chunk(sm + B_BYTES, sm, ra2);
    }
    if (it < nper) chunk(sm, sm + B_BYTES, ra);
  }
#ifndef X3_NO_DRAIN
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the requests past the last chunk are never consumed: none may outlive the loop)
#endif
#ifdef X3_CLK
  if (q.dbg != nullptr && lane == 0 && (wave & 3) == 0) {
    unsigned long long* d = q.dbg + ((long long)blockIdx.x * 2 + (wave >> 2)) * 8;
    if (d[0] == 0) {
      d[0] = ck0;
      d[1] = cr0;
      d[2] = __builtin_amdgcn_s_memtime();
      d[3] = __builtin_amdgcn_s_memrealtime();
      d[4] = c_comp;
      d[5] = c_skel;
      d[6] = c_bar;
      d[7] = nper;
    }
  }
#endif
  // ---- the raw partial tile (C/D layout: col = lane & 15, row = 4 * (lane >> 4) + reg) to this segment's slab; x3_fixup_k finishes the tile
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = wrow + rb * 16 + g * 4 + r;
      if (m0 + row >= M) continue;
#pragma unroll
      for (int j = 0; j < NT; ++j)
        if (n0 + j * 16 + l15 < p.Nout) pdst[(long long)row * BN + j * 16 + l15] = acc[rb][j][r];
    }
}

// Every tile goes through the stream-K parts: workgroup `part` multiplies the contiguous range [part * ipp, (part + 1) * ipp) of the
// tiles' K iterations (the tail of one tile and the head of the next: at most two segments when parts >= tiles), each segment into its
// (tile, slot) slab [tiles][smax][256][BN]; one 512-thread workgroup per CU.  (A whole-tile form -- tiles beyond a multiple of the CU
// count finished inside this kernel -- gave results that depended on what ran beside it on the chip in one of 3 x 16 trials with
// two learners' graphs in flight, tests/test_step_gpu.py::test_concurrent_task_lanes_...; the all-parts form did not in 32, and it is
// one code path less.  SkPlan.full stays 0.  What actually disturbed the other learner's kernels was co-residency with this kernel's
// matrix instruction: see the register-file claim below.)
template <int NT>
__global__ __launch_bounds__(512) void conv_x3_k(X3Params q, SkPlan k) {
  __shared__ __attribute__((aligned(16))) char sm[X3Sm<NT>::BYTES];
  constexpr int BN = 16 * NT;
  // This kernel claims the CU's WHOLE register file (eight waves x 256 VGPRs: touching v255 makes the descriptor say so), so that no
  // wave of another kernel can be co-resident with it on a CU.  Why (measured on MI355X, ROCm 7.2: tools/interfere_probe.py,
  // profiles/r06_notes.md): while a wave that interleaves v_mfma_f32_16x16x32_bf16 with LDS or vector-memory instructions is resident
  // on a CU, a packed fp32 instruction (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) of ANY OTHER wave on that CU whose op_sel is [0,1]
  // (low result from src0's low and src1's HIGH register) intermittently returns a wrong LOW result for a 16-lane pass.
  // Every other select form, the high result, single fp32 instructions, loads, stores and integer work are never
  // affected; neither is anything on a CU the aggressor does not occupy; the fp32 matrix instruction is not an aggressor.  (Round 5 saw
  // it as "kernels of another stream return wrong values beside this kernel": the head's resize and the final layer's filter gradient
  // are two of the six kernels of this library that contain that form.)  With its natural 157-194 registers: victims wrong in 240 of 360
  // rounds; with the file claimed (this kernel and conv_filter_x3_batched_k below): 0 of 3600.  One workgroup per CU is the plan anyway:
  // the claim costs nothing.  -DX3_NO_CLAIM: probe builds only (tools/build_variants.sh).
#ifndef X3_NO_CLAIM
  asm volatile("v_mov_b32 v255, 0" ::: "v255");
#endif
  // Consecutive parts -- the K ranges of one row tile and of its neighbours, i.e. the nine taps over the same A rows -- on ONE XCD, so that
  // they share an L2: fabric reads of the 224 -> 112 conv at 56 x 56 199.2 -> 45.6 MB per launch (rocprofv3 --pmc FETCH_SIZE; algorithmic
  // 31.5 MB), 78.1 -> 76.0 us with its fix-up, +0.3 % on the step (profiles/r06_notes.md).  -DX3_NO_XCD: the plain order.
#ifndef X3_NO_XCD
  const int part = (int)xcd_remap(blockIdx.x, gridDim.x);
#else
  const int part = blockIdx.x;
#endif
  int lo = part * k.ipp;
  const int total = k.rem * k.nchunks;
  int hi = lo + k.ipp;
  if (hi > total) hi = total;
#pragma unroll 1
  while (lo < hi) {
    const int rt = lo / k.nchunks, c0 = lo - rt * k.nchunks;
    int c1 = c0 + (hi - lo);
    if (c1 > k.nchunks) c1 = k.nchunks;
    const int first = (rt * k.nchunks) / k.ipp;   // first part that touches this tile -> slot 0
    float* dst = k.slab + ((long long)rt * k.smax + (part - first)) * (kX3BM * BN);
    conv_x3_tile<NT>(q, sm, rt / k.gy, rt % k.gy, c0, c1, dst);
    lo += c1 - c0;
    __syncthreads();   // the next segment reuses the LDS buffers
  }
}

// Finishes the stream-K tiles: grid = (rem, kX3Fix): workgroup (rt, s) owns the 64 rows [64 s, 64 s + 64) of remainder tile rt (a
// 98-workgroup grid -- one per 256-row tile -- left most of the chip idle: 13.4 us per launch); 1024 threads = 32 column-quad slots x
// 32 row lanes, two rows per thread.  Statistics: kX3Fix blocks per tile, [bx * kX3Fix + s][2][Nout].
constexpr int kX3Fix = 4;
template <int NT>
__global__ __launch_bounds__(1024) void x3_fixup_k(ConvGemmParams p, SkPlan k) {
  constexpr int BN = 16 * NT, QN = BN / 4;
  constexpr int QS = QN <= 32 ? 32 : 64;              // column-quad slots (NT = 9: 36 quads)
  constexpr int RL = 1024 / QS;                       // row lanes
  constexpr int RPT = kX3BM / kX3Fix / RL;            // rows per thread
  __shared__ float4 red[2][RL][QS];
  const int t = threadIdx.x, q = t % QS, rl = t / QS;
  const int rt = blockIdx.x, sub = blockIdx.y;
  const unsigned tau = k.full + rt;
  const int bx = tau / k.gy, by = tau % k.gy;
  const long long M = (long long)p.Nimg * p.H * p.W;
  const int r0 = sub * (kX3BM / kX3Fix);            // first row of this workgroup inside the tile
  const long long m0 = (long long)bx * kX3BM + r0;
  const int n = by * BN + q * 4;
  const bool cok = q < QN && n < p.Nout;
  const int first = (rt * k.nchunks) / k.ipp;
  int last = ((rt + 1) * k.nchunks - 1) / k.ipp;
  if (last > k.parts - 1) last = k.parts - 1;
  const int nslots = last - first + 1;
  const float* base = k.slab + ((long long)rt * k.smax * kX3BM + r0) * BN + q * 4;
  float4 s1 = f4zero(), s2 = f4zero();
  if (cok) {
    const float4 bv = p.bias != nullptr ? ld4(p.bias + n) : f4zero();
    const long long HWp = (long long)p.H * p.W;
    float4 v[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) v[i] = (m0 + rl + RL * i < M) ? ld4(base + (long long)(rl + RL * i) * BN) : f4zero();
    for (int z = 1; z < nslots; ++z) {
      float4 u[RPT];
#pragma unroll
      for (int i = 0; i < RPT; ++i) u[i] = (m0 + rl + RL * i < M) ? ld4(base + ((long long)z * kX3BM + rl + RL * i) * BN) : f4zero();
#pragma unroll
      for (int i = 0; i < RPT; ++i) v[i] = f4add(v[i], u[i]);
    }
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
      const long long m = m0 + rl + RL * i;
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
  if (t < 2 * QS) {
    const int v = t / QS, qq = t % QS;
    const int nn = by * BN + qq * 4;
    if (qq < QN && nn < p.Nout) {
      float4 a = red[v][0][qq];
#pragma unroll 8
      for (int r = 1; r < RL; ++r) a = f4add(a, red[v][r][qq]);
      st4(p.stats_part + (((long long)bx * kX3Fix + sub) * 2 + v) * p.Nout + nn, a);
    }
  }
}

// ------------------------------------------------------------------------------------------------ backward-filter
// dW[tap][ci][co] = sum over pixels of X[pixel + tap offset][ci] * dY[pixel][co] with the same split products: the reduction runs over
// the PIXELS, so both operands need k-major fragments of tensors whose channels are contiguous.  A chunk of 32 pixels is staged once
// per workgroup: every fp32 value is split while it is stored (each element once), as three row-major bf16 planes [pixel][channel]
// (X: 128 input channels, dY: up to 128 output columns; 256-byte rows whose 32-byte column chunks are XOR-swizzled by the pixel), and
// the fragments come back TRANSPOSED through ds_read_b64_tr_b16 (a 16-lane group reads 4 pixel rows x 16 channels and every lane
// receives its channel's four pixels: two reads make the eight k values of a 16x16x32 operand).  Workgroup = 8 waves = 4 blocks of 32
// input channels x 2 halves of the column tiles (the two waves of a SIMD hold one half each: 12 + 9 fragment reads for 48 + 36 matrix
// instructions per chunk and SIMD pair); tile, slabs and descriptor table are those of conv_filter_grad2_batched_k<2, NT> (one tap per
// workgroup, the multitap form of the concat's sliver included), so the plan, the workspace and the batched fold do not change.
typedef short x3_v4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 ds_read_tr16(const char* p) {
  const x3_v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((x3_v4s __attribute__((address_space(3)))*)(p));
  return __builtin_bit_cast(uint2, r);
}
constexpr int kF3Rows = 32;                          // pixels per chunk
constexpr int kF3Plane = kF3Rows * 256;              // bytes of one term plane of one operand (32 rows x 128 channels x 2 bytes)
constexpr int kF3Stage = 6 * kF3Plane;               // X planes + dY planes
__device__ __forceinline__ int f3_swz(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }   // chunk XOR of a pixel row

// BCI = input channels of a workgroup tile.  128: eight waves = 4 channel blocks x 2 halves of the column tiles (round 5).  256 (round 6,
// problems with more than 128 input channels): eight waves = 8 channel blocks, every wave all column tiles -- dY is staged (loaded, split,
// stored) ONCE for up to 256 input channels instead of once per 128, twice the matrix instructions per staged chunk and barrier.
template <int BCI>
struct F3Sm {
  static constexpr int XQ = BCI / 4;                      // channel quads of an X plane row
  static constexpr int XP = BCI * 2;                      // bytes of an X plane row (bf16)
  static constexpr int X_PLANE = kF3Rows * XP;
  static constexpr int STAGE = 3 * X_PLANE + 3 * kF3Plane;   // X planes + dY planes
};
template <int NT, int BCI>
__device__ __forceinline__ void conv_filter_x3_body(const FilterGradParams& p, char* __restrict__ sm, int bx, int by, int bz) {
  constexpr int BN = 16 * NT;
  constexpr bool WIDE = BCI == 256;
  constexpr int XQ = F3Sm<BCI>::XQ, XP = F3Sm<BCI>::XP, X_PLANE = F3Sm<BCI>::X_PLANE, STAGE = F3Sm<BCI>::STAGE;
  constexpr int X_PER_THREAD = kF3Rows * XQ / 512;        // 2 | 4 quads of X per thread and chunk
  constexpr int X_ROWS_PASS = 512 / XQ;                   // 16 | 8 pixel rows per pass of the 512 threads
  constexpr int D_TOTAL = kF3Rows * (BN / 4);
  constexpr int D_PER_THREAD = (D_TOTAL + 511) / 512;
  constexpr int NT0 = WIDE ? NT : (NT + 1) / 2, NT1 = WIDE ? 0 : NT / 2;   // column tiles of the two wave halves (WIDE: no halves)
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int cib = WIDE ? wave : (wave & 3), half = WIDE ? 0 : (wave >> 2);   // 32-channel block of the tile; column half
  const int l15 = lane & 15, g = lane >> 4;
  const int M = p.Nimg * p.H * p.W;
  const bool mt = p.multitap != 0;
  const int cblocks = mt ? 1 : (p.C + BCI - 1) / BCI;
  const int tap = mt ? 0 : bx / cblocks;
  const int ci0 = mt ? 0 : (bx - tap * cblocks) * BCI;
  const int n0 = by * BN;
#ifdef F3_SAMEBZ   // ablation: every pixel split reads the FIRST pixel range (wrong numbers: the kernel with its operands L2-resident)
  const int mbeg = 0;
#else
  const int mbeg = bz * p.rows_per_split;
#endif
  int mend = mbeg + p.rows_per_split;
  if (mend > M) mend = M;
  const int HW = p.H * p.W;
  const int adv_h = kF3Rows / p.W, adv_w = kF3Rows - adv_h * p.W;
  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, kBufRecords, 0x00020000);
  const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc((void*)p.dY, 0, kBufRecords, 0x00020000);

  // ---- X: thread t loads the channel quad x_cq of pixel rows x_r0, x_r0 + X_ROWS_PASS, ... of every chunk
  const int x_cq = t % XQ, x_r0 = t / XQ;
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
  int x_m[X_PER_THREAD], x_h[X_PER_THREAD], x_w[X_PER_THREAD];
  unsigned x_off[X_PER_THREAD];
#pragma unroll
  for (int i = 0; i < X_PER_THREAD; ++i) {
    const int m = mbeg + x_r0 + X_ROWS_PASS * i;
    x_m[i] = m;
    const int n = m / HW;
    const int rem = m - n * HW;
    x_h[i] = rem / p.W;
    x_w[i] = rem - x_h[i] * p.W;
    x_off[i] = (unsigned)((((long long)m + (long long)dh * p.W + dw) * p.ldx + l_c) * 4);
  }
  const unsigned x_step = (unsigned)(kF3Rows * p.ldx * 4);
  int d_m[D_PER_THREAD];
  unsigned d_off[D_PER_THREAD];
  bool d_ok[D_PER_THREAD];
  int d_lds[D_PER_THREAD];
#pragma unroll
  for (int i = 0; i < D_PER_THREAD; ++i) {
    const int idx = t + 512 * i;
    const int r = idx / (BN / 4), nq = idx - r * (BN / 4);
    d_m[i] = mbeg + r;
    d_ok[i] = (idx < D_TOTAL) && (n0 + nq * 4 < p.Nout);
    d_off[i] = (unsigned)((((long long)mbeg + r) * p.lddy + n0 + nq * 4) * 4);
    d_lds[i] = idx < D_TOTAL ? 3 * X_PLANE + r * 256 + (((nq >> 2) ^ f3_swz(r)) * 32) + (nq & 3) * 8 : -1;
  }
  const unsigned d_step = (unsigned)(kF3Rows * p.lddy * 4);
  int x_lds[X_PER_THREAD];   // (the XOR swizzle acts on the low three bits of the 32-byte chunk index: rows of 256 or 512 bytes alike)
#pragma unroll
  for (int i = 0; i < X_PER_THREAD; ++i) {
    const int r = x_r0 + X_ROWS_PASS * i;
    const int ch = x_cq >> 2;
    x_lds[i] = r * XP + (((ch & ~7) | ((ch & 7) ^ f3_swz(r))) * 32) + (x_cq & 3) * 8;
  }

  float4 rxa[X_PER_THREAD], rda[D_PER_THREAD], rxb[X_PER_THREAD], rdb[D_PER_THREAD];   // two chunks in flight (requested a whole chunk before their split)
  auto load_chunk = [&](float4 (&rx)[X_PER_THREAD], float4 (&rd)[D_PER_THREAD]) {   // next 32 pixel rows -> registers, then advance (rows beyond
                                                                                   // mend / halo pixels come back as zeros)
#pragma unroll
    for (int i = 0; i < X_PER_THREAD; ++i) {
      const bool ok = x_cok & (x_m[i] < mend) & ((unsigned)(x_h[i] + dh) < (unsigned)p.H) & ((unsigned)(x_w[i] + dw) < (unsigned)p.W);
      rx[i] = buf_ld4(rX, ok ? x_off[i] : kOob);
      x_m[i] += kF3Rows;
      x_off[i] += x_step;
      x_w[i] += adv_w;
      x_h[i] += adv_h;
      if (x_w[i] >= p.W) {
        x_w[i] -= p.W;
        ++x_h[i];
      }
      if (x_h[i] >= p.H) x_h[i] -= (x_h[i] / p.H) * p.H;   // crossed into the next image(s)
    }
#pragma unroll
    for (int i = 0; i < D_PER_THREAD; ++i) {
      rd[i] = buf_ld4(rD, (d_ok[i] & (d_m[i] < mend)) ? d_off[i] : kOob);
      d_m[i] += kF3Rows;
      d_off[i] += d_step;
    }
  };
  // the split of the chunk in registers (this thread's 2 + D_PER_THREAD quads -> three terms each) is done in pieces BETWEEN the
  // matrix instructions of the chunk being multiplied (compute_split below), the terms are stored behind them
  constexpr int NP = X_PER_THREAD + D_PER_THREAD;
  uint2 th[NP], tm[NP], tl[NP];
  auto split_piece = [&](int k, const float4 (&rx)[X_PER_THREAD], const float4 (&rd)[D_PER_THREAD]) {
#ifdef F3_NOSPLIT   // ablation: the loads and the LDS stores of the staging, none of its arithmetic (wrong numbers)
    {
      const float4 v_ = k < X_PER_THREAD ? rx[k < X_PER_THREAD ? k : 0] : rd[k >= X_PER_THREAD ? k - X_PER_THREAD : 0];
      th[k] = make_uint2(__float_as_uint(v_.x), __float_as_uint(v_.y));
      tm[k] = make_uint2(__float_as_uint(v_.z), __float_as_uint(v_.w));
      tl[k] = th[k];
    }
#else
    split3(k < X_PER_THREAD ? rx[k < X_PER_THREAD ? k : 0] : rd[k >= X_PER_THREAD ? k - X_PER_THREAD : 0], th[k], tm[k], tl[k]);
#endif
    // (an empty statement that "uses" the piece here: without it the optimiser sinks the arithmetic to the stores behind the loop)
    asm volatile("" : "+v"(th[k].x), "+v"(th[k].y), "+v"(tm[k].x), "+v"(tm[k].y), "+v"(tl[k].x), "+v"(tl[k].y));
  };
  auto store_terms = [&](char* buf) {
#pragma unroll
    for (int i = 0; i < X_PER_THREAD; ++i) {
      char* d = buf + x_lds[i];
      *reinterpret_cast<uint2*>(d) = th[i];
      *reinterpret_cast<uint2*>(d + X_PLANE) = tm[i];
      *reinterpret_cast<uint2*>(d + 2 * X_PLANE) = tl[i];
    }
#pragma unroll
    for (int i = 0; i < D_PER_THREAD; ++i) {
      if (512 * (i + 1) <= D_TOTAL || d_lds[i] >= 0) {
        char* d = buf + d_lds[i];
        *reinterpret_cast<uint2*>(d) = th[X_PER_THREAD + i];
        *reinterpret_cast<uint2*>(d + kF3Plane) = tm[X_PER_THREAD + i];
        *reinterpret_cast<uint2*>(d + 2 * kF3Plane) = tl[X_PER_THREAD + i];
      }
    }
  };
  constexpr int NTW = NT0;   // accumulator tiles per wave (half 1 uses NT1 of them)
  f32x4 acc[2][NTW];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NTW; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // fragment addresses: lane (g, q = l15 >> 2, pq = l15 & 3) supplies pixel row 8 g + 4 hh + q, columns 4 pq .. 4 pq + 3 of a 16-channel chunk
  const int fq = l15 >> 2, fp = l15 & 3;
  int a_row[2], d_row[2], a_sw[2];   // byte offset of this lane's pixel row in an X plane / a dY plane, and the row's chunk XOR
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    const int r = 8 * g + 4 * hh + fq;
    a_row[hh] = r * XP + fp * 8;
    d_row[hh] = r * 256 + fp * 8;
    a_sw[hh] = f3_swz(r);
  }
  const bool wave_live = mt ? cib * 32 < p.ntaps * p.C : ci0 + cib * 32 < p.C;
  const int jn0 = half == 0 ? 0 : NT0;           // first column tile of this wave
  const int njn = half == 0 ? NT0 : NT1;
  auto frag = [&](const char* plane, const int (&row)[2], int chunk16) {   // the 8 k values of one 16-channel block: two transposed reads
    const uint2 lo = ds_read_tr16(plane + row[0] + (((chunk16 & ~7) | ((chunk16 & 7) ^ a_sw[0])) * 32));
    const uint2 hi = ds_read_tr16(plane + row[1] + (((chunk16 & ~7) | ((chunk16 & 7) ^ a_sw[1])) * 32));
    return __builtin_bit_cast(bf16x8, (u32x4){lo.x, lo.y, hi.x, hi.y});
  };
  // One chunk's products out of LDS stage `buf`; the fragments of column tile j + 1 are requested before the products of tile j, and
  // one piece of the NEXT chunk's split rides behind the matrix instructions of every column tile (fenced: conv_x3_tile)
  auto compute_split = [&](const char* buf, const float4 (&rx)[X_PER_THREAD], const float4 (&rd)[D_PER_THREAD]) {
    bf16x8 a3[3][2], b3[2][3];
    if (wave_live) {
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) a3[pl][rb] = frag(buf + pl * X_PLANE, a_row, cib * 2 + rb);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) b3[0][pl] = frag(buf + 3 * X_PLANE + pl * kF3Plane, d_row, jn0);
    }
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
      if (wave_live && j < njn) {   // (uniform)
        if (j + 1 < NTW && j + 1 < njn) {
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) b3[(j + 1) & 1][pl] = frag(buf + 3 * X_PLANE + pl * kF3Plane, d_row, jn0 + j + 1);
        }
#define X3_MM(PA, PB)                                                                                            \
  acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[PA][0], b3[j & 1][PB], acc[0][j], 0, 0, 0);           \
  acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[PA][1], b3[j & 1][PB], acc[1][j], 0, 0, 0);
#ifdef F3_NOMFMA   // ablation: the chunk skeleton without its matrix instructions (one per accumulator keeps the fragments live)
        X3_MM(0, 0)
#else
        X3_MM(2, 0) X3_MM(0, 2) X3_MM(1, 1) X3_MM(1, 0) X3_MM(0, 1) X3_MM(0, 0)
#endif
#undef X3_MM
      }
#ifndef F3_NOSTAGE
#pragma unroll
      for (int k = 0; k < NP; ++k)
        if ((k * NTW) / NP == j) split_piece(k, rx, rd);
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  const int nchunks = (mend - mbeg + kF3Rows - 1) / kF3Rows;
  load_chunk(rxa, rda);   // chunk 0
#pragma unroll
  for (int k = 0; k < NP; ++k) split_piece(k, rxa, rda);
  store_terms(sm);
  load_chunk(rxa, rda);   // chunk 1
  load_chunk(rxb, rdb);   // chunk 2
  lds_barrier();
  auto trip = [&](char* cur, char* nxt, float4 (&rx)[X_PER_THREAD], float4 (&rd)[D_PER_THREAD]) {   // (rx, rd): the chunk after the one in `cur`
    compute_split(cur, rx, rd);   // + the split of the next chunk
#ifndef F3_NOSTAGE                // (ablation: no split arithmetic, no LDS stores -- the products run on stale stages)
    store_terms(nxt);             // (zeros after the last chunk: nobody reads them)
#endif
    load_chunk(rx, rd);           // two chunks after the one just split
    lds_barrier();
  };
  int it = 0;
  for (; it + 2 <= nchunks; it += 2) {
    trip(sm, sm + STAGE, rxa, rda);
    trip(sm + STAGE, sm, rxb, rdb);
  }
  if (it < nchunks) trip(sm, sm + STAGE, rxa, rda);
#ifndef X3_NO_DRAIN
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (as in conv_x3_tile)
#endif

  const long long Ktot = (long long)p.ntaps * p.C;
#pragma unroll
  for (int rb = 0; rb < 2; ++rb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ci = ci0 + cib * 32 + rb * 16 + g * 4 + r;   // multitap: the flattened (tap, channel) row, tap == 0 in the index below
      if (ci >= (mt ? p.ntaps * p.C : p.C)) continue;
#pragma unroll
      for (int j = 0; j < NTW; ++j) {
        const int n = n0 + (jn0 + j) * 16 + l15;
        if (j < njn && n < p.Nout) p.partial[((long long)bz * Ktot + (long long)tap * p.C + ci) * p.Nout + n] = acc[rb][j][r];
      }
    }
}

// (Round 5, measured and removed: a form in which a workgroup multiplies the THREE taps of one kernel row per staged chunk -- a chunk
//  = a segment of one image row, the X window of 32 + 2 dil pixels staged once, dY staged once for three taps, 16 channels x all column
//  tiles per wave.  Ablations of the one-tap form on the decoder's two 56x56 problems, 178 us as one cold launch
//  (-DF3_NOMFMA / -DF3_NOSTAGE / -DF3_SAMEBZ): 1 of 6 matrix instructions 116 us; no split arithmetic and no LDS stores 97 us; both
//  40 us; every workgroup on the same pixels 163 us -- the parts add and traffic is not the bound.  The kernel-row form: 158 us cold,
//  149 against 151 us inside the step, 3424 / 3416 against 3409 / 3418 images/s: no gain where it counts, 330 lines: not kept.)
// the problems of one (TMF = 2, NT) group of a FilterBatch as one grid: descriptor table as conv_filter_grad2_batched_k
template <int NT>
__global__ __launch_bounds__(512) void conv_filter_x3_batched_k(const long long* __restrict__ desc, int nprob, int tile_ci) {
  __shared__ __attribute__((aligned(16))) char sm[2 * F3Sm<256>::STAGE];
  // the CU's whole register file, as conv_x3_k and for the same reason (one 512-thread workgroup per CU already: 96 KB of LDS);
  // -DF3_NO_CLAIM: probe builds only
#ifndef F3_NO_CLAIM
  asm volatile("v_mov_b32 v255, 0" ::: "v255");
#endif
  // (The 18 (tap, channel-block) workgroups of one pixel range read the same dY rows and overlapping X rows and have consecutive block
  //  ids, which the hardware deals round-robin over the eight XCDs.  Remapped so that they share ONE L2 -- xcd_remap -- the launch
  //  took 183 us instead of 152: eight L2s serve the hot lines faster than one.  Plain order.)
  const int b = blockIdx.x;
  int j = 0;
  for (int k = 1; k < nprob; ++k)
    if (b >= (int)desc[k * kFilterDescWords + 15]) j = k;
  const long long* d = desc + (long long)j * kFilterDescWords;
  const int ks = (int)(d[11] & 0xff);
  const FilterGradParams p{reinterpret_cast<const float*>(d[0]), (int)d[4], (int)d[6], (int)d[7], (int)d[8], (int)d[9], ks * ks, (int)d[12],
                           reinterpret_cast<const float*>(d[1]), (int)d[5], (int)d[10], reinterpret_cast<float*>(d[3]),
                           (int)(d[13] & 0xffffffffLL), nullptr, (int)(d[13] >> 32), 0, 0};
  const int gx = (int)(d[14] & 0xfffff), gy = (int)((d[14] >> 20) & 0xfffff);
  const int local = b - (int)d[15];
  const int bx = local % gx, r = local / gx;
  // problems with more than 128 input channels take 256-channel tiles (the table's gx of such a problem is taps x ceil(C / 256): the caller
  // builds the table for THIS kernel -- mliis_conv2d_bwd_filter_batched, tile_ci == 256), everything else, the multitap form included, 128
  if (tile_ci == 256 && p.multitap == 0 && p.C > 128) conv_filter_x3_body<NT, 256>(p, sm, bx, r % gy, r / gy);   // (uniform)
  else conv_filter_x3_body<NT, 128>(p, sm, bx, r % gy, r / gy);
}

bool launch_filter_batched_x3(int nt, const long long* desc, int nprob, int blocks, int tile_ci, hipStream_t stream) {
  dim3 grid(blocks), block(512);
#define L(NT_) hipLaunchKernelGGL((conv_filter_x3_batched_k<NT_>), grid, block, 0, stream, desc, nprob, tile_ci); return true;
  switch (nt) {
    case 4: L(4)
    case 5: L(5)
    case 6: L(6)
    case 7: L(7)
    case 8: L(8)
    default: return false;
  }
#undef L
}

// ------------------------------------------------------------------------------------------------ host side
struct X3Plan {
  int nt, gx, gy, nchunks;
  int full, rem, parts, ipp, smax;
  size_t slab_floats() const { return (size_t)rem * smax * kX3BM * (16 * nt); }
};
constexpr int kX3MinPart = 4;   // K chunks per stream-K part, at least

#ifndef X3_MAX_NT
#define X3_MAX_NT 9   // (-DX3_MAX_NT=8: probe builds, round 5's tiling)
#endif
static int x3_pick_nt(int Nout) {   // the widest column tile that does not pad the output width by much (as conv_gemm.hip: pick_nt, with a
                                    // stronger pull towards few column tiles)
  int best = 1;
  double best_cost = 1e30;
  const int tiles = (Nout + 15) / 16;
  for (int nt = 1; nt <= X3_MAX_NT; ++nt) {   // (9: the 136-column backward-data conv of the dilated branch as ONE column tile of 144 -- A loaded and split
                                      //  once instead of once per 80-column tile: round 6)
    const int blocks = (tiles + nt - 1) / nt;
    const double cost = (double)(blocks * nt * 16) / (double)Nout * (1.0 + 0.08 * (8 - nt));   // (A is loaded AND split once per column tile)
    if (cost < best_cost - 1e-9 || (cost < best_cost + 1e-9 && nt > best)) {
      best_cost = cost;
      best = nt;
    }
  }
  return best;
}

static inline int x3_chunks(int ntaps, int Cred) { return (ntaps * Cred + 31) / 32; }   // the flattened (tap, channel) index in chunks of 32
static X3Plan x3_plan(long long M, int Nout, int ntaps, int Cred, int num_cus) {
  X3Plan g;
  g.nt = x3_pick_nt(Nout);
  g.gy = (Nout + g.nt * 16 - 1) / (g.nt * 16);
  g.gx = (int)((M + kX3BM - 1) / kX3BM);
  g.nchunks = x3_chunks(ntaps, Cred);
  const int slots = num_cus;   // one 512-thread workgroup per CU
  // small maps (the 14x14 level: 13 row tiles): narrower column tiles until the K iterations can be dealt over the chip
  while (g.nt > 2 && (long long)g.gx * g.gy * g.nchunks < (long long)slots * 2 * kX3MinPart) {
    g.nt = (g.nt + 1) / 2;
    g.gy = (Nout + g.nt * 16 - 1) / (g.nt * 16);
  }
  const long long tiles = (long long)g.gx * g.gy;
  g.full = 0;   // (every tile through the parts: conv_x3_k)
  g.rem = (int)tiles;
  const long long total = tiles * g.nchunks;
  long long parts = slots;
  if (parts > 8LL * tiles) parts = 8LL * tiles;
  if (total / parts < kX3MinPart) parts = total / kX3MinPart;
  if (parts < 1) parts = 1;
  g.ipp = (int)((total + parts - 1) / parts);
  g.parts = (int)((total + g.ipp - 1) / g.ipp);
  g.smax = (g.nchunks + g.ipp - 1) / g.ipp + 1;
  return g;
}

static int g_x3_cus = 0;
static int x3_num_cus() {
  if (g_x3_cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    g_x3_cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
                   ? prop.multiProcessorCount
                   : 256;
  }
  return g_x3_cus;
}

static void x3_launch(const X3Plan& g, const X3Params& q_, float* slab, hipStream_t stream) {
  X3Params q = q_;
#ifdef X3_CLK
  q.dbg = getenv("MLIIS_X3_STAMPS") ? (unsigned long long*)strtoull(getenv("MLIIS_X3_STAMPS"), nullptr, 0) : nullptr;
#endif
  const SkPlan k{g.full, g.rem, g.parts, g.ipp, g.nchunks, g.smax, g.gy, slab};
  dim3 grid(g.parts), block(512);
#define L(NT_)                                                                                  \
  hipLaunchKernelGGL((conv_x3_k<NT_>), grid, block, 0, stream, q, k);                           \
  hipLaunchKernelGGL((x3_fixup_k<NT_>), dim3(g.rem, kX3Fix), dim3(1024), 0, stream, q.g, k);    \
  break;
  switch (g.nt) {
    case 1: L(1)
    case 2: L(2)
    case 3: L(3)
    case 4: L(4)
    case 5: L(5)
    case 6: L(6)
    case 7: L(7)
    case 8: L(8)
    default: L(9)
  }
#undef L
}

static int x3_shape_check(const char* name, int Nimg, int H, int W, int Cred, int Nout, int ksize, int dil) {
  MLIIS_REQUIRE(Nimg > 0 && H > 0 && W > 0 && Cred >= 32 && Nout > 0, MLIIS_ERR_ARG, "%s: bad shape (the reduced channel count must be >= 32)", name);
  MLIIS_REQUIRE((Cred & 3) == 0 && (Nout & 3) == 0, MLIIS_ERR_ARG, "%s: channel counts must be multiples of 4", name);
  MLIIS_REQUIRE(ksize == 1 || ksize == 3, MLIIS_ERR_UNSUPPORTED, "%s: kernel size %d unsupported (1 or 3)", name, ksize);
  MLIIS_REQUIRE(dil >= 1, MLIIS_ERR_ARG, "%s: dilation must be >= 1", name);
  return MLIIS_OK;
}

}  // namespace mliis

using namespace mliis;

extern "C" {

// bytes of the weight image of a conv direction that reduces over Cred channels per tap and produces Nout columns
size_t mliis_x3_image_bytes(int Cred, int Nout, int ksize) {
  return (size_t)x3_chunks(ksize * ksize, Cred) * ((Nout + 15) / 16) * kX3Block;
}

// blocks of x3_pack_k for that image (the `first block` column of the descriptor table is the running sum of these)
int mliis_x3_image_blocks(int Cred, int Nout, int ksize) { return x3_chunks(ksize * ksize, Cred) * ((Nout + 15) / 16); }

int mliis_x3_pack_weights(const float* theta, void* images, const long long* desc, int ndesc, int total_blocks, hipStream_t stream) {
  MLIIS_REQUIRE(theta && images && desc && ndesc >= 1 && ndesc <= 64 && total_blocks >= 1, MLIIS_ERR_ARG, "x3_pack_weights: bad arguments");
  MLIIS_REQUIRE(aligned16(theta) && aligned16(images) && aligned16(desc), MLIIS_ERR_ALIGN, "x3_pack_weights: pointers must be 16-byte aligned");
  hipLaunchKernelGGL(x3_pack_k, dim3(total_blocks), dim3(128), 0, stream, theta, reinterpret_cast<char*>(images), desc, ndesc);
  MLIIS_CHECK_LAUNCH("x3_pack_weights");
  return MLIIS_OK;
}

size_t mliis_conv2d_x3_workspace_floats(int Nimg, int H, int W, int Cred, int Nout, int ksize) {
  return x3_plan((long long)Nimg * H * W, Nout, ksize * ksize, Cred, x3_num_cus()).slab_floats();
}

// tiling of an x3 call: plan[0..5] = {column tiles NT, row tiles, column tiles per row, whole-tile workgroups, stream-K parts, K chunks per part}
int mliis_conv2d_x3_plan(int Nimg, int H, int W, int Cred, int Nout, int ksize, int* plan) {
  MLIIS_REQUIRE(plan, MLIIS_ERR_ARG, "conv2d_x3_plan: null pointer");
  const X3Plan g = x3_plan((long long)Nimg * H * W, Nout, ksize * ksize, Cred, x3_num_cus());
  plan[0] = g.nt; plan[1] = g.gx; plan[2] = g.gy; plan[3] = g.full; plan[4] = g.parts; plan[5] = g.ipp;
  return MLIIS_OK;
}

// y[M, Cout] (ld = ldy) (+)= conv(x[M, Cin] (ld = ldx), w) + bias (+ border_bias), stride 1, TF-SAME, dilation dil, the weights as the
// mode-0 image of mliis_x3_pack_weights over the same Cin window.  stats_part / stats_swish / stats_nblk as mliis_conv2d_fwd
// (*stats_nblk = four blocks per 256-row tile).  ws: stream-K slabs (mliis_conv2d_x3_workspace_floats).
int mliis_conv2d_fwd_x3(const float* x, int ldx, const void* image, size_t image_bytes, const float* bias, const float* border_bias, float* y, int ldy, int Nimg,
                        int H, int W, int Cin, int Cout, int ksize, int dil, int accumulate, float* stats_part, int stats_swish,
                        int* stats_nblk, float* ws, size_t ws_floats, hipStream_t stream) {
  int rc = x3_shape_check("conv2d_fwd_x3", Nimg, H, W, Cin, Cout, ksize, dil);
  if (rc) return rc;
  MLIIS_REQUIRE(x && image && y, MLIIS_ERR_ARG, "conv2d_fwd_x3: null pointer");
  MLIIS_REQUIRE(image_bytes == mliis_x3_image_bytes(Cin, Cout, ksize), MLIIS_ERR_ARG,
                "conv2d_fwd_x3: the weight image has %zu bytes, a forward image over %d reduced channels x %d columns (k = %d) has %zu -- packed for another "
                "window or direction?", image_bytes, Cin, Cout, ksize, mliis_x3_image_bytes(Cin, Cout, ksize));
  MLIIS_REQUIRE((ldx & 3) == 0 && ldx >= Cin && (ldy & 3) == 0 && ldy >= Cout, MLIIS_ERR_ARG, "conv2d_fwd_x3: bad leading dimensions");
  MLIIS_REQUIRE(aligned16(x) && aligned16(image) && aligned16(bias) && aligned16(y) && aligned16(border_bias) && aligned16(ws), MLIIS_ERR_ALIGN,
                "conv2d_fwd_x3: pointers must be 16-byte aligned");
  const long long M = (long long)Nimg * H * W;
  MLIIS_REQUIRE(M * ldx * 4 < (1LL << 31) && mliis_x3_image_bytes(Cin, Cout, ksize) < (1ULL << 31), MLIIS_ERR_UNSUPPORTED,
                "conv2d_fwd_x3: operand larger than 2 GiB (32-bit buffer offsets)");
  MLIIS_REQUIRE(border_bias == nullptr || (ksize == 3 && dil == 1 && H >= 2 && W >= 2), MLIIS_ERR_ARG,
                "conv2d_fwd_x3: border_bias needs a 3x3 dilation-1 conv on a map of at least 2x2");
  MLIIS_REQUIRE(stats_part == nullptr || (!accumulate && stats_nblk), MLIIS_ERR_ARG,
                "conv2d_fwd_x3: fused statistics need accumulate == 0 and a stats_nblk output");
  const X3Plan g = x3_plan(M, Cout, ksize * ksize, Cin, x3_num_cus());
  MLIIS_REQUIRE(g.slab_floats() == 0 || (ws != nullptr && g.slab_floats() <= ws_floats), MLIIS_ERR_WORKSPACE,
                "conv2d_fwd_x3: workspace too small (%zu needed, %zu given)", g.slab_floats(), ws_floats);
  X3Params q{};
  q.g = ConvGemmParams{x, ldx, Nimg, H, W, Cin, ksize * ksize, dil, +1, nullptr, 0, 0, Cout, y, ldy, bias, accumulate, nullptr, 0,
                       stats_part, stats_swish, nullptr, border_bias, 1.0f, nullptr};
  q.image = reinterpret_cast<const char*>(image);
  q.ncol16 = (Cout + 15) / 16;
  x3_launch(g, q, ws, stream);
  MLIIS_CHECK_LAUNCH("conv2d_fwd_x3");
  if (stats_nblk) *stats_nblk = stats_part != nullptr ? g.gx * kX3Fix : 0;
  return MLIIS_OK;
}

// dx[M, Cin_out] (ld = lddx) (+)= conv_transpose(dy[M, Cout] (ld = lddy), w) over the input-channel window the mode-1 image was packed for
int mliis_conv2d_bwd_data_x3(const float* dy, int lddy, const void* image, size_t image_bytes, float* dx, int lddx, int Nimg, int H, int W, int Cin_out, int Cout,
                             int ksize, int dil, int accumulate, float* ws, size_t ws_floats, hipStream_t stream) {
  int rc = x3_shape_check("conv2d_bwd_data_x3", Nimg, H, W, Cout, Cin_out, ksize, dil);
  if (rc) return rc;
  MLIIS_REQUIRE(dy && image && dx, MLIIS_ERR_ARG, "conv2d_bwd_data_x3: null pointer");
  MLIIS_REQUIRE(image_bytes == mliis_x3_image_bytes(Cout, Cin_out, ksize), MLIIS_ERR_ARG,
                "conv2d_bwd_data_x3: the weight image has %zu bytes, a backward-data image over %d reduced channels x %d columns (k = %d) has %zu -- packed "
                "for another window or direction?", image_bytes, Cout, Cin_out, ksize, mliis_x3_image_bytes(Cout, Cin_out, ksize));
  MLIIS_REQUIRE((lddy & 3) == 0 && lddy >= Cout && (lddx & 3) == 0 && lddx >= Cin_out, MLIIS_ERR_ARG, "conv2d_bwd_data_x3: bad leading dimensions");
  MLIIS_REQUIRE(aligned16(dy) && aligned16(image) && aligned16(dx) && aligned16(ws), MLIIS_ERR_ALIGN,
                "conv2d_bwd_data_x3: pointers must be 16-byte aligned");
  const long long M = (long long)Nimg * H * W;
  MLIIS_REQUIRE(M * lddy * 4 < (1LL << 31) && mliis_x3_image_bytes(Cout, Cin_out, ksize) < (1ULL << 31), MLIIS_ERR_UNSUPPORTED,
                "conv2d_bwd_data_x3: operand larger than 2 GiB (32-bit buffer offsets)");
  const X3Plan g = x3_plan(M, Cin_out, ksize * ksize, Cout, x3_num_cus());
  MLIIS_REQUIRE(g.slab_floats() == 0 || (ws != nullptr && g.slab_floats() <= ws_floats), MLIIS_ERR_WORKSPACE,
                "conv2d_bwd_data_x3: workspace too small (%zu needed, %zu given)", g.slab_floats(), ws_floats);
  X3Params q{};
  q.g = ConvGemmParams{dy, lddy, Nimg, H, W, Cout, ksize * ksize, dil, -1, nullptr, 0, 0, Cin_out, dx, lddx, nullptr, accumulate, nullptr, 0,
                       nullptr, 0, nullptr, nullptr, 1.0f, nullptr};
  q.image = reinterpret_cast<const char*>(image);
  q.ncol16 = (Cin_out + 15) / 16;
  x3_launch(g, q, ws, stream);
  MLIIS_CHECK_LAUNCH("conv2d_bwd_data_x3");
  return MLIIS_OK;
}
}
