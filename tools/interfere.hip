// Victims and synthetic aggressors of tools/interfere_probe.py: what goes wrong in a small kernel of one stream while the split-product
// kernels of another stream are on the chip (profiles/r05_notes.md, "An interference found on the way").
//
// Victims are SELF-CHECKING: every value a victim loads is a hash of its own address (so a wrong load is seen as such, and the wrong
// word itself says where it came from: every tensor of the probe carries a 4-bit tag in its low mantissa bits), and every piece of
// arithmetic is evaluated twice from laundered inputs (so a transient wrong result of a vector instruction is seen as a
// disagreement of the two evaluations).  A fault record carries the hardware id of the wave (XCC, SE, CU, SIMD) and every intermediate.
//
// Aggressors are register-/LDS-only loops of ONE instruction class each (bf16 matrix instruction, fp32 matrix instruction, the
// f32 -> bf16 conversion, LDS traffic behind an LDS-only barrier, raw buffer loads with out-of-range lanes, transposed LDS reads), with
// few registers so that victim waves are co-resident with them.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short v4s __attribute__((ext_vector_type(4)));

constexpr int kRecWords = 32, kMaxRec = 512;
struct Log {
  unsigned count;             // faults seen (records beyond kMaxRec are counted, not kept)
  unsigned cu_bitmap[16];     // (xcc 0..7) x 64 (se, cu) slots: where the kernel's waves ran
  unsigned pad[15];
  unsigned rec[kMaxRec][kRecWords];
};

#define GETREG(id, off, size) ((((size) - 1) << 11) | ((off) << 6) | (id))
__device__ __forceinline__ unsigned hw_id() { return __builtin_amdgcn_s_getreg(GETREG(4, 0, 32)); }     // HW_REG_HW_ID
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(GETREG(20, 0, 4)); }    // HW_REG_XCC_ID
__device__ __forceinline__ void mark_cu(Log* lg) {
  if ((threadIdx.x & 63) == 0) {
    const unsigned h = hw_id(), x = xcc_id();
    const unsigned cu = (h >> 8) & 15, se = (h >> 13) & 3;   // gfx9 HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]
    const unsigned slot = se * 16 + cu;
    atomicOr(&lg->cu_bitmap[x * 2 + (slot >> 5)], 1u << (slot & 31));
  }
}
__device__ __forceinline__ unsigned* new_rec(Log* lg, unsigned kind) {
  const unsigned i = atomicAdd(&lg->count, 1u);
  if (i >= kMaxRec) return nullptr;
  unsigned* r = lg->rec[i];
  r[0] = kind;
  r[1] = blockIdx.x;
  r[2] = threadIdx.x;
  r[3] = hw_id();
  r[4] = xcc_id();
  return r;
}

// value of element j of a tagged tensor: a float in about [-4, 4) whose low four mantissa bits are `tag`
__host__ __device__ __forceinline__ unsigned tagged_bits(unsigned j, unsigned tag) {
  unsigned h = j * 2654435761u + 0x9e3779b9u;
  h ^= h >> 15;
  h *= 2246822519u;
  h ^= h >> 13;
  // sign 1, exponent 0x7d..0x80 (0.25 .. 4), 23 mantissa bits
  const unsigned sign = h & 0x80000000u, ex = 0x7du + ((h >> 29) & 3u), man = (h >> 4) & 0x7ffff0u;
  return sign | (ex << 23) | man | (tag & 15u);
}
__global__ void fill_tagged_k(unsigned* p, unsigned n, unsigned tag) {
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = tagged_bits(i, tag);
}

__device__ __forceinline__ void src_coord(int o, float scale, int in_size, int& i0, int& i1, float& l) {
  const float f = (float)o * scale;
  i0 = (int)floorf(f);
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + 1 < in_size ? i0 + 1 : in_size - 1;
  l = f - (float)i0;
}

// ---- victim 1: the head's bilinear resize of a two-channel map (head.hip: resize_fwd_k<2>), loads verified, arithmetic twice
struct ResizeEval {
  int y0, y1, x0, x1;
  float ly, lx;
  float2 tl, tr, bl, br, o;
  unsigned e[4];   // element index (float2 units) of the four corners
};
__device__ __attribute__((noinline)) void resize_eval(const float* x, long long i, int Hi, int Wi, int Ho, int Wo, float sh, float sw, ResizeEval& r) {
  const int wo = (int)(i % Wo);
  const long long q = i / Wo;
  const int ho = (int)(q % Ho);
  const int n = (int)(q / Ho);
  src_coord(ho, sh, Hi, r.y0, r.y1, r.ly);
  src_coord(wo, sw, Wi, r.x0, r.x1, r.lx);
  const long long b = (long long)n * Hi * Wi;
  r.e[0] = (unsigned)(b + (long long)r.y0 * Wi + r.x0);
  r.e[1] = (unsigned)(b + (long long)r.y0 * Wi + r.x1);
  r.e[2] = (unsigned)(b + (long long)r.y1 * Wi + r.x0);
  r.e[3] = (unsigned)(b + (long long)r.y1 * Wi + r.x1);
  r.tl = *reinterpret_cast<const float2*>(x + 2ull * r.e[0]);
  r.tr = *reinterpret_cast<const float2*>(x + 2ull * r.e[1]);
  r.bl = *reinterpret_cast<const float2*>(x + 2ull * r.e[2]);
  r.br = *reinterpret_cast<const float2*>(x + 2ull * r.e[3]);
  float2 o = make_float2(0.f, 0.f);
  const float w0 = (1.f - r.ly) * (1.f - r.lx), w1 = (1.f - r.ly) * r.lx, w2 = r.ly * (1.f - r.lx), w3 = r.ly * r.lx;
  o.x = fmaf(w0, r.tl.x, o.x); o.y = fmaf(w0, r.tl.y, o.y);
  o.x = fmaf(w1, r.tr.x, o.x); o.y = fmaf(w1, r.tr.y, o.y);
  o.x = fmaf(w2, r.bl.x, o.x); o.y = fmaf(w2, r.bl.y, o.y);
  o.x = fmaf(w3, r.br.x, o.x); o.y = fmaf(w3, r.br.y, o.y);
  r.o = o;
}
__global__ __launch_bounds__(256) void v_resize_k(const float* __restrict__ x, float* __restrict__ y, int N, int Hi, int Wi, int Ho, int Wo, float sh,
                                                  float sw, unsigned tag, Log* lg) {
  mark_cu(lg);
  const long long total = (long long)N * Ho * Wo;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    ResizeEval a, b;
    resize_eval(x, i, Hi, Wi, Ho, Wo, sh, sw, a);
    // second evaluation from laundered inputs: the compiler must emit every instruction again
    long long i2 = i;
    const float* x2 = x;
    asm volatile("" : "+v"(i2));
    asm volatile("" : "+s"(x2));
    resize_eval(x2, i2, Hi, Wi, Ho, Wo, sh, sw, b);
    const float2* la = &a.tl;
    const float2* lb = &b.tl;
    unsigned bad_load = 0, bad_reload = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float2 va = k == 0 ? a.tl : k == 1 ? a.tr : k == 2 ? a.bl : a.br;
      const float2 vb = k == 0 ? b.tl : k == 1 ? b.tr : k == 2 ? b.bl : b.br;
      const unsigned ex = tagged_bits(2u * a.e[k], tag), ey = tagged_bits(2u * a.e[k] + 1u, tag);
      if (__float_as_uint(va.x) != ex || __float_as_uint(va.y) != ey) bad_load |= 1u << k;
      if (__float_as_uint(vb.x) != ex || __float_as_uint(vb.y) != ey) bad_reload |= 1u << k;
    }
    const bool bad_idx = a.e[0] != b.e[0] || a.e[1] != b.e[1] || a.e[2] != b.e[2] || a.e[3] != b.e[3] || __float_as_uint(a.ly) != __float_as_uint(b.ly) ||
                         __float_as_uint(a.lx) != __float_as_uint(b.lx);
    const bool bad_o = __float_as_uint(a.o.x) != __float_as_uint(b.o.x) || __float_as_uint(a.o.y) != __float_as_uint(b.o.y);
    if (bad_load | bad_reload | (unsigned)bad_idx | (unsigned)bad_o) {
      unsigned* r = new_rec(lg, 1u);
      if (r) {
        r[5] = (unsigned)i;
        r[6] = bad_load | (bad_reload << 4) | ((unsigned)bad_idx << 8) | ((unsigned)bad_o << 9);
        r[7] = a.e[0]; r[8] = a.e[1]; r[9] = a.e[2]; r[10] = a.e[3];
        r[11] = __float_as_uint(a.tl.x); r[12] = __float_as_uint(a.tl.y); r[13] = __float_as_uint(a.tr.x); r[14] = __float_as_uint(a.tr.y);
        r[15] = __float_as_uint(a.bl.x); r[16] = __float_as_uint(a.bl.y); r[17] = __float_as_uint(a.br.x); r[18] = __float_as_uint(a.br.y);
        r[19] = __float_as_uint(b.tl.x); r[20] = __float_as_uint(b.tl.y); r[21] = __float_as_uint(b.tr.x); r[22] = __float_as_uint(b.tr.y);
        r[23] = __float_as_uint(b.bl.x); r[24] = __float_as_uint(b.bl.y); r[25] = __float_as_uint(b.br.x); r[26] = __float_as_uint(b.br.y);
        r[27] = __float_as_uint(a.o.x); r[28] = __float_as_uint(a.o.y); r[29] = __float_as_uint(b.o.x); r[30] = __float_as_uint(b.o.y);
        r[31] = b.e[0];
      }
    }
    (void)la; (void)lb;
    *reinterpret_cast<float2*>(y + 2 * i) = a.o;
  }
}

// ---- victim 2: no loads at all -- fp32 fma chains, the 64-bit index divisions of the resize, float <-> int conversions, each twice
__device__ __attribute__((noinline)) void alu_eval(long long i, int Wo, int Ho, float sh, unsigned (&out)[6]) {
  const int wo = (int)(i % Wo);
  const long long q = i / Wo;
  const int ho = (int)(q % Ho);
  const int n = (int)(q / Ho);
  const float f = (float)ho * sh;
  const int y0 = (int)floorf(f);
  const float ly = f - (float)y0;
  float a = (float)wo * 0.0078125f + 0.5f, b = ly + 0.25f, c = (float)(n + 1) * 0.125f;
  float s0 = a, s1 = b;
#pragma unroll
  for (int k = 0; k < 24; ++k) {
    s0 = fmaf(s0, 0.9990234375f, c);
    s1 = fmaf(s1, -0.99951171875f, a);
    s0 = fmaf(s1, 0.001953125f, s0);
  }
  out[0] = (unsigned)wo; out[1] = (unsigned)ho; out[2] = (unsigned)n; out[3] = (unsigned)y0;
  out[4] = __float_as_uint(s0); out[5] = __float_as_uint(s1);
}
__global__ __launch_bounds__(256) void v_alu_k(float* __restrict__ y, long long total, int Wo, int Ho, float sh, Log* lg) {
  mark_cu(lg);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    unsigned a[6], b[6];
    alu_eval(i, Wo, Ho, sh, a);
    long long i2 = i;
    asm volatile("" : "+v"(i2));
    alu_eval(i2, Wo, Ho, sh, b);
    unsigned bad = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) bad |= (a[k] != b[k]) ? 1u << k : 0u;
    if (bad) {
      unsigned* r = new_rec(lg, 2u);
      if (r) {
        r[5] = (unsigned)i; r[6] = bad;
#pragma unroll
        for (int k = 0; k < 6; ++k) { r[7 + k] = a[k]; r[13 + k] = b[k]; }
      }
    }
    *reinterpret_cast<float2*>(y + 2 * i) = make_float2(__uint_as_float(a[4]), __uint_as_float(a[5]));
  }
}

// ---- victim 3: a copy with 4-, 8- and 16-byte loads of the tagged tensor, each checked against the hash
template <int V>
__global__ __launch_bounds__(256) void v_copy_k(const unsigned* __restrict__ x, unsigned* __restrict__ y, unsigned nelem, unsigned tag, Log* lg) {
  mark_cu(lg);
  typedef unsigned T __attribute__((ext_vector_type(V)));
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < nelem / V; i += gridDim.x * blockDim.x) {
    union { T t; unsigned u[V]; } v;
    if constexpr (V == 1) v.u[0] = x[i];
    else v.t = *reinterpret_cast<const T*>(x + (size_t)i * V);
    unsigned bad = 0;
#pragma unroll
    for (int k = 0; k < V; ++k) bad |= (v.u[k] != tagged_bits(i * V + k, tag)) ? 1u << k : 0u;
    if (bad) {
      unsigned* r = new_rec(lg, 3u);
      if (r) {
        r[5] = i; r[6] = bad; r[7] = V;
#pragma unroll
        for (int k = 0; k < V; ++k) { r[8 + k] = v.u[k]; r[12 + k] = tagged_bits(i * V + k, tag); }
      }
    }
    if constexpr (V == 1) y[i] = v.u[0];
    else *reinterpret_cast<T*>(y + (size_t)i * V) = v.t;
  }
}

// ------------------------------------------------------------------------------------------------ synthetic aggressors
// MASK bits: 1 bf16 matrix instruction (16x16x32), 2 f32 -> bf16 conversions + the split arithmetic, 4 LDS b128 traffic behind LDS-only
// barriers, 8 raw buffer b128 loads (scalar offset, out-of-range lanes), 16 fp32 matrix instruction (16x16x4), 32 transposed LDS reads
// (ds_read_b64_tr_b16), 64: the barrier of bit 4 is __syncthreads() instead of the LDS-only one, 128: plain global loads instead of
// the raw buffer loads of bit 8.  15 = a register-light model of conv_x3_k.
__device__ __forceinline__ unsigned pk_bf16x2(float a, float b) {
  typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
  bf16x2_ r;
  r[0] = (__bf16)a;
  r[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, r);
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int MASK>
__global__ __launch_bounds__(256) void a_synth_k(float* __restrict__ out, const unsigned* __restrict__ src, unsigned src_bytes, int iters, Log* lg) {
  constexpr bool LDS = (MASK & (4 | 32)) != 0;
  __shared__ __attribute__((aligned(16))) char sm[LDS ? 40960 : 16];
  mark_cu(lg);
  const int t = threadIdx.x;
  f32x4 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) acc[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float v0 = t * 1e-3f + 1.f, v1 = blockIdx.x * 1e-5f + 0.5f, v2 = 0.75f, v3 = 1.25f;
  u32x4 pa = {0x3f803f80u + t, 0x3f003f00u, 0x3f803f80u, 0x3f003f00u + blockIdx.x}, pb = {0x3f803f80u, 0x3f803f00u, 0x3f003f80u, 0x3f003f00u};
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x80000000u, 0x00020000);
  const unsigned span = src_bytes / 2;
  unsigned voff = ((unsigned)(blockIdx.x * 256 + t) * 16u) % (span - 4096u);
  const bool oob = (t & 7) == 7;
  unsigned soff = 0;
  u32x4 ld = {0, 0, 0, 0};
  if (LDS) {
    for (int k = t; k < 40960 / 16; k += 256) *reinterpret_cast<u32x4*>(sm + k * 16) = (u32x4){(unsigned)k, 1u, 2u, 3u};
    __syncthreads();
  }
  for (int it = 0; it < iters; ++it) {
    if (MASK & 1) {
      const bf16x8 a = __builtin_bit_cast(bf16x8, pa), b = __builtin_bit_cast(bf16x8, pb);
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j], 0, 0, 0);
    }
    if (MASK & 16) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(v0, v1, acc[j], 0, 0, 0);
    }
    if (MASK & 2) {
      const unsigned h = pk_bf16x2(v0, v1);
      const float rx = v0 - __uint_as_float(h << 16), ry = v1 - __uint_as_float(h & 0xffff0000u);
      const unsigned m = pk_bf16x2(rx, ry);
      const unsigned l = pk_bf16x2(rx - __uint_as_float(m << 16), ry - __uint_as_float(m & 0xffff0000u));
      pa.x ^= h & 0x00010001u; pa.y ^= m & 0x00010001u; pb.x ^= l & 0x00010001u;
      v0 = fmaf(v0, 0.999f, 0.001f * v2); v1 = fmaf(v1, 0.998f, 0.002f * v3);
      asm volatile("" : "+v"(v0), "+v"(v1));
    }
    if (MASK & 4) {
      const int o = ((t + it) & 255) * 16;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const u32x4 q = *reinterpret_cast<const u32x4*>(sm + o + k * 4096);
        ld.x += q.x; ld.y ^= q.y;
      }
      *reinterpret_cast<u32x4*>(sm + 16384 + ((it & 1) * 8192) + t * 16) = ld;
      if (MASK & 64) __syncthreads();
      else lds_barrier();
    }
    if (MASK & 8) {
      u32x4 q;
      if (MASK & 128) q = oob ? (u32x4){0, 0, 0, 0} : *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(src) + voff + soff);
      else q = __builtin_amdgcn_raw_buffer_load_b128(rs, oob ? (int)0xFFFFFFF0u : (int)voff, (int)soff, 0);
      ld.z += q.x ^ q.w;
      soff = (soff + 4096u) % (span - 8192u);
      soff &= ~15u;
    }
    if (MASK & 32) {
      const v4s q = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(sm + ((t * 8 + it * 512) & 32767)));
      ld.w += (unsigned)q[0] + (unsigned)q[3];
    }
  }
  float s = v0 + v1 + __uint_as_float(ld.x ^ ld.y ^ ld.z ^ ld.w) * 0.f + __uint_as_float(pa.x) * 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[blockIdx.x * 256 + t] = s;
}

// ---- the library's victims verbatim (head.hip: resize_fwd_k<2>), compiled into THIS code object: no log, no census, one evaluation
__global__ __launch_bounds__(256) void plain_resize_k(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int N, int Hi, int Wi, int Ho,
                                                      int Wo, int C, float sh, float sw) {
  const int Q = C / 2;
  const long long total = (long long)N * Ho * Wo * Q;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int c = (int)(i % Q) * 2;
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
    const float2 tl = *reinterpret_cast<const float2*>(base + ((long long)y0 * Wi + x0) * ldx);
    const float2 tr = *reinterpret_cast<const float2*>(base + ((long long)y0 * Wi + x1) * ldx);
    const float2 bl = *reinterpret_cast<const float2*>(base + ((long long)y1 * Wi + x0) * ldx);
    const float2 br = *reinterpret_cast<const float2*>(base + ((long long)y1 * Wi + x1) * ldx);
    float2 o = make_float2(0.f, 0.f);
    const float w0 = (1.f - ly) * (1.f - lx), w1 = (1.f - ly) * lx, w2 = ly * (1.f - lx), w3 = ly * lx;
    o = make_float2(fmaf(w0, tl.x, o.x), fmaf(w0, tl.y, o.y));
    o = make_float2(fmaf(w1, tr.x, o.x), fmaf(w1, tr.y, o.y));
    o = make_float2(fmaf(w2, bl.x, o.x), fmaf(w2, bl.y, o.y));
    o = make_float2(fmaf(w3, br.x, o.x), fmaf(w3, br.y, o.y));
    *reinterpret_cast<float2*>(y + p * ldy + c) = o;
  }
}
// a victim with NO floating-point arithmetic at all: y[i] = x[perm(i)] (8-byte elements), gathered like the resize's corner loads
__global__ __launch_bounds__(256) void plain_gather_k(const float* __restrict__ x, float* __restrict__ y, int N, int Hi, int Wi, int Ho, int Wo, float sh,
                                                      float sw) {
  const long long total = (long long)N * Ho * Wo;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int wo = (int)(i % Wo);
    const long long r = i / Wo;
    const int ho = (int)(r % Ho);
    const int n = (int)(r / Ho);
    int y0, y1, x0, x1;
    float ly, lx;
    src_coord(ho, sh, Hi, y0, y1, ly);
    src_coord(wo, sw, Wi, x0, x1, lx);
    const float2 v = *reinterpret_cast<const float2*>(x + (((long long)n * Hi + y1) * Wi + x0) * 2);
    *reinterpret_cast<float2*>(y + i * 2) = v;
  }
}


// ---- victim 4: ONE instruction form per kernel (inline asm: the exact encodings the failing library kernels contain), register-only,
// each result checked against the same arithmetic done with single (non-packed) instructions in the same thread.
// bad[FORM * 2 + half]: mismatching low / high halves.  FORM: 0 v_pk_fma_f32 (all VGPR)   1 v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[0,1]
// 2 v_pk_mul_f32 with an SGPR-pair source   3 v_pk_fma_f32 with an SGPR-pair source and neg modifiers   4 v_pk_add_f32 with the inline
// constant 1.0, op_sel_hi:[1,0] and neg   5 v_pk_fma_f32 with the literal 0 accumulator, op_sel_hi:[1,1,0]   6 v_pk_fma_f32 whose operand
// arrives from a global_load_dwordx2   7 control: v_fma_f32 against v_mul_f32 + v_add_f32-free re-evaluation (same instruction twice)
typedef float f2 __attribute__((ext_vector_type(2)));
template <int FORM>
__global__ __launch_bounds__(256) void v_form_k(float* __restrict__ y, unsigned* __restrict__ bad, const float* __restrict__ src, unsigned nsrc2, int iters,
                                                float s0, float s1) {
  const unsigned gid = blockIdx.x * 256 + threadIdx.x;
  f2 a = {1.0f + (gid & 1023) * 0.0009765625f, 0.5f + (gid >> 10) * 0.001953125f}, b = {0.75f + (gid & 255) * 0.00390625f, 1.25f - (gid & 63) * 0.0078125f};
  f2 c = {0.125f, -0.375f};
  const f2 sp = {s0, s1};
  const unsigned long long spq = ((unsigned long long)__builtin_amdgcn_readfirstlane(__float_as_uint(s1)) << 32) | __builtin_amdgcn_readfirstlane(__float_as_uint(s0));
  unsigned lo = 0, hi = 0;
  float sum0 = 0.f, sum1 = 0.f;
  float k0 = 0.999f, k1 = 0.0011f, k2 = 1.0005f, k3 = -0.0007f, k4 = 0.001f, k5 = -0.0005f;
  asm volatile("" : "+v"(k0), "+v"(k1), "+v"(k2), "+v"(k3), "+v"(k4), "+v"(k5));
  for (int it = 0; it < iters; ++it) {
    f2 r, e;
    if (FORM == 0) {
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e.x) : "v"(a.x), "v"(b.x), "v"(c.x));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e.y) : "v"(a.y), "v"(b.y), "v"(c.y));
    } else if (FORM == 1) {
      asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e.x) : "v"(a.x), "v"(b.y));
      e.y = e.x;
    } else if (FORM == 2) {
      asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "s"(spq), "v"(b));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e.x) : "s"(sp.x), "v"(b.x));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e.y) : "s"(sp.y), "v"(b.y));
    } else if (FORM == 3) {
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(r) : "s"(spq), "v"(b), "v"(c));
      asm volatile("v_fma_f32 %0, %1, %2, -%3" : "=v"(e.x) : "s"(sp.x), "v"(b.x), "v"(c.x));
      asm volatile("v_fma_f32 %0, %1, %2, -%3" : "=v"(e.y) : "s"(sp.y), "v"(b.y), "v"(c.y));
    } else if (FORM == 4) {
      asm volatile("v_pk_add_f32 %0, %1, 1.0 op_sel_hi:[1,0] neg_lo:[1,0] neg_hi:[1,0]" : "=v"(r) : "v"(a));
      asm volatile("v_sub_f32 %0, 1.0, %1" : "=v"(e.x) : "v"(a.x));
      asm volatile("v_sub_f32 %0, 1.0, %1" : "=v"(e.y) : "v"(a.y));
    } else if (FORM == 5) {
      asm volatile("v_pk_fma_f32 %0, %1, %2, 0 op_sel_hi:[1,1,0]" : "=v"(r) : "v"(a), "v"(b));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e.x) : "v"(a.x), "v"(b.x));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e.y) : "v"(a.y), "v"(b.y));
    } else if (FORM == 6) {
      const f2 g = *reinterpret_cast<const f2*>(src + 2ull * ((gid * 7u + it * 977u) % nsrc2));
      asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(g), "v"(c));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e.x) : "v"(a.x), "v"(g.x), "v"(c.x));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e.y) : "v"(a.y), "v"(g.y), "v"(c.y));
    } else {
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r.x) : "v"(a.x), "v"(b.x), "v"(c.x));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r.y) : "v"(a.y), "v"(b.y), "v"(c.y));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e.x) : "v"(a.x), "v"(b.x), "v"(c.x));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e.y) : "v"(a.y), "v"(b.y), "v"(c.y));
    }
    lo += __float_as_uint(r.x) != __float_as_uint(e.x);
    hi += __float_as_uint(r.y) != __float_as_uint(e.y);
    asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum0) : "v"(e.x));
    asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum1) : "v"(e.y));
    // next operands from the single-instruction results only (a packed fault must not propagate into the inputs)
    // (single instructions, written out: the SLP vectoriser would turn C arithmetic on the pairs into packed instructions again)
    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a.x) : "v"(k0), "v"(k1));
    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a.y) : "v"(k2), "v"(k3));
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(b.x) : "v"(k4), "v"(e.y));
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(b.y) : "v"(k5), "v"(a.x));
  }
  if (lo) atomicAdd(&bad[FORM * 2], lo);
  if (hi) atomicAdd(&bad[FORM * 2 + 1], hi);
  y[2ull * gid] = sum0;
  y[2ull * gid + 1] = sum1;
}


// ---- victim 5: every source-select form of the three packed fp32 instructions.  OP 0 v_pk_mul_f32, 1 v_pk_add_f32, 2 v_pk_fma_f32 (src2 in
// natural order); SEL = a | b << 1 | c << 2 | d << 3 for op_sel:[a,b] op_sel_hi:[c,d]: the LOW result is src0[a] (op) src1[b], the HIGH result
// src0[c] (op) src1[d] (0 = the low register of the pair, 1 = the high one).  Reference: single instructions.  bad[((OP * 16 + SEL) * 2 + half]
template <int OP, int SEL>
__global__ __launch_bounds__(256) void v_sel_k(float* __restrict__ y, unsigned* __restrict__ bad, int iters) {
  const unsigned gid = blockIdx.x * 256 + threadIdx.x;
  f2 a = {1.0f + (gid & 1023) * 0.0009765625f, 0.5f + (gid >> 10) * 0.001953125f}, b = {0.75f + (gid & 255) * 0.00390625f, 1.25f - (gid & 63) * 0.0078125f};
  f2 c = {0.125f, -0.375f};
  unsigned lo = 0, hi = 0;
  float sum0 = 0.f, sum1 = 0.f;
  float k0 = 0.999f, k1 = 0.0011f, k2 = 1.0005f, k3 = -0.0007f, k4 = 0.001f, k5 = -0.0005f;
  asm volatile("" : "+v"(k0), "+v"(k1), "+v"(k2), "+v"(k3), "+v"(k4), "+v"(k5));
  constexpr int A = SEL & 1, B = (SEL >> 1) & 1, Cc = (SEL >> 2) & 1, D = (SEL >> 3) & 1;
  for (int it = 0; it < iters; ++it) {
    f2 r, e;
#define SELCASE(N, SA, SB, SC, SD)                                                                                                              \
  if constexpr (SEL == N) {                                                                                                                     \
    if constexpr (OP == 0) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[" #SA "," #SB "] op_sel_hi:[" #SC "," #SD "]" : "=v"(r) : "v"(a), "v"(b));           \
    if constexpr (OP == 1) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[" #SA "," #SB "] op_sel_hi:[" #SC "," #SD "]" : "=v"(r) : "v"(a), "v"(b));           \
    if constexpr (OP == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[" #SA "," #SB ",0] op_sel_hi:[" #SC "," #SD ",1]" : "=v"(r) : "v"(a), "v"(b), "v"(c)); \
  }
    SELCASE(0, 0, 0, 0, 0) SELCASE(1, 1, 0, 0, 0) SELCASE(2, 0, 1, 0, 0) SELCASE(3, 1, 1, 0, 0)
    SELCASE(4, 0, 0, 1, 0) SELCASE(5, 1, 0, 1, 0) SELCASE(6, 0, 1, 1, 0) SELCASE(7, 1, 1, 1, 0)
    SELCASE(8, 0, 0, 0, 1) SELCASE(9, 1, 0, 0, 1) SELCASE(10, 0, 1, 0, 1) SELCASE(11, 1, 1, 0, 1)
    SELCASE(12, 0, 0, 1, 1) SELCASE(13, 1, 0, 1, 1) SELCASE(14, 0, 1, 1, 1) SELCASE(15, 1, 1, 1, 1)
#undef SELCASE
    const float al = A ? a.y : a.x, bl = B ? b.y : b.x, ah = Cc ? a.y : a.x, bh = D ? b.y : b.x;
    if constexpr (OP == 0) {
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e.x) : "v"(al), "v"(bl));
      asm volatile("v_mul_f32 %0, %1, %2" : "=v"(e.y) : "v"(ah), "v"(bh));
    } else if constexpr (OP == 1) {
      asm volatile("v_add_f32 %0, %1, %2" : "=v"(e.x) : "v"(al), "v"(bl));
      asm volatile("v_add_f32 %0, %1, %2" : "=v"(e.y) : "v"(ah), "v"(bh));
    } else {
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e.x) : "v"(al), "v"(bl), "v"(c.x));
      asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(e.y) : "v"(ah), "v"(bh), "v"(c.y));
    }
    lo += __float_as_uint(r.x) != __float_as_uint(e.x);
    hi += __float_as_uint(r.y) != __float_as_uint(e.y);
    if (__float_as_uint(r.x) != __float_as_uint(e.x)) {   // what IS the wrong low result?  The same operation on another choice of registers:
      // class 0: src0 as selected, src1's OTHER register; 1: src0's other, src1 as selected; 2: both others; 3: none of these
      const float ao = A ? a.x : a.y, bo = B ? b.x : b.y;
      float alt[3];
#define ALT(K, P, Q)                                                                                                    \
  if constexpr (OP == 0) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(alt[K]) : "v"(P), "v"(Q));                       \
  else if constexpr (OP == 1) asm volatile("v_add_f32 %0, %1, %2" : "=v"(alt[K]) : "v"(P), "v"(Q));                  \
  else asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(alt[K]) : "v"(P), "v"(Q), "v"(c.x));
      ALT(0, al, bo) ALT(1, ao, bl) ALT(2, ao, bo)
#undef ALT
      int cls = 3;
      for (int k = 2; k >= 0; --k)
        if (__float_as_uint(r.x) == __float_as_uint(alt[k])) cls = k;
      atomicAdd(&bad[96 + (OP * 16 + SEL) * 4 + cls], 1u);
    }
    asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum0) : "v"(e.x));
    asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum1) : "v"(e.y));
    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a.x) : "v"(k0), "v"(k1));
    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a.y) : "v"(k2), "v"(k3));
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(b.x) : "v"(k4), "v"(e.y));
    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(b.y) : "v"(k5), "v"(a.x));
  }
  if (lo) atomicAdd(&bad[(OP * 16 + SEL) * 2], lo);
  if (hi) atomicAdd(&bad[(OP * 16 + SEL) * 2 + 1], hi);
  y[2ull * gid] = sum0;
  y[2ull * gid + 1] = sum1;
}

template <int OP, int SEL>
static void launch_sel(float* y, unsigned* bad, int blocks, int iters, hipStream_t st) {
  hipLaunchKernelGGL((v_sel_k<OP, SEL>), dim3(blocks), dim3(256), 0, st, y, bad, iters);
  if constexpr (SEL + 1 < 16) launch_sel<OP, SEL + 1>(y, bad, blocks, iters, st);
}
extern "C" {
int ifp_log_bytes() { return (int)sizeof(Log); }
int ifp_fill_tagged(void* p, unsigned n, unsigned tag, hipStream_t st) {
  hipLaunchKernelGGL(fill_tagged_k, dim3(1024), dim3(256), 0, st, (unsigned*)p, n, tag);
  return (int)hipGetLastError();
}
unsigned ifp_tagged_bits(unsigned j, unsigned tag) { return tagged_bits(j, tag); }
int ifp_resize(const float* x, float* y, int N, int Hi, int Wi, int Ho, int Wo, unsigned tag, void* lg, hipStream_t st) {
  const float sh = (float)(Hi - 1) / (float)(Ho - 1), sw = (float)(Wi - 1) / (float)(Wo - 1);
  const long long total = (long long)N * Ho * Wo;
  hipLaunchKernelGGL(v_resize_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, x, y, N, Hi, Wi, Ho, Wo, sh, sw, tag, (Log*)lg);
  return (int)hipGetLastError();
}
int ifp_alu(float* y, long long total, int Wo, int Ho, void* lg, hipStream_t st) {
  hipLaunchKernelGGL(v_alu_k, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, y, total, Wo, Ho, 0.24663677f, (Log*)lg);
  return (int)hipGetLastError();
}
int ifp_copy(const void* x, void* y, unsigned nelem, int vec, unsigned tag, void* lg, hipStream_t st) {
  const unsigned blocks = (nelem / vec + 255) / 256;
  if (vec == 1) hipLaunchKernelGGL(v_copy_k<1>, dim3(blocks), dim3(256), 0, st, (const unsigned*)x, (unsigned*)y, nelem, tag, (Log*)lg);
  else if (vec == 2) hipLaunchKernelGGL(v_copy_k<2>, dim3(blocks), dim3(256), 0, st, (const unsigned*)x, (unsigned*)y, nelem, tag, (Log*)lg);
  else hipLaunchKernelGGL(v_copy_k<4>, dim3(blocks), dim3(256), 0, st, (const unsigned*)x, (unsigned*)y, nelem, tag, (Log*)lg);
  return (int)hipGetLastError();
}
int ifp_aggressor(int mask, float* out, const void* src, unsigned src_bytes, int blocks, int iters, void* lg, hipStream_t st) {
#define L(K) case K: hipLaunchKernelGGL(a_synth_k<K>, dim3(blocks), dim3(256), 0, st, out, (const unsigned*)src, src_bytes, iters, (Log*)lg); break;
  switch (mask) {
    L(1) L(2) L(3) L(4) L(5) L(6) L(7) L(8) L(9) L(10) L(11) L(12) L(13) L(14) L(15) L(16) L(32) L(20) L(28) L(30) L(79) L(143) L(207) L(31) L(47)
    default: return -1;
  }
#undef L
  return (int)hipGetLastError();
}
int ifp_forms(float* y, unsigned* bad, const float* src, unsigned nsrc2, int blocks, int iters, hipStream_t st) {
#define F(K) hipLaunchKernelGGL(v_form_k<K>, dim3(blocks), dim3(256), 0, st, y, bad, src, nsrc2, iters, 0.24663677f, 0.7531f);
  F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)
#undef F
  return (int)hipGetLastError();
}
int ifp_sel_matrix(float* y, unsigned* bad, int blocks, int iters, hipStream_t st) {   // bad: 96 + 192 counters
  launch_sel<0, 0>(y, bad, blocks, iters, st);
  launch_sel<1, 0>(y, bad, blocks, iters, st);
  launch_sel<2, 0>(y, bad, blocks, iters, st);
  return (int)hipGetLastError();
}
int ifp_plain_resize(const float* x, float* y, int N, int Hi, int Wi, int Ho, int Wo, int blocks, hipStream_t st) {
  const float sh = (float)(Hi - 1) / (float)(Ho - 1), sw = (float)(Wi - 1) / (float)(Wo - 1);
  hipLaunchKernelGGL(plain_resize_k, dim3(blocks), dim3(256), 0, st, x, 2, y, 2, N, Hi, Wi, Ho, Wo, 2, sh, sw);
  return (int)hipGetLastError();
}
int ifp_plain_gather(const float* x, float* y, int N, int Hi, int Wi, int Ho, int Wo, int blocks, hipStream_t st) {
  const float sh = (float)(Hi - 1) / (float)(Ho - 1), sw = (float)(Wi - 1) / (float)(Wo - 1);
  hipLaunchKernelGGL(plain_gather_k, dim3(blocks), dim3(256), 0, st, x, y, N, Hi, Wi, Ho, Wo, sh, sw);
  return (int)hipGetLastError();
}
}
