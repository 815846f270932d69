// On-device pixel half of the inner-loop augmentation (`--augment`, which the reference's run.sh turns on): the six operations of
// augmenters/np_augmenters.py:9-131 -- random eraser, translate, left-right flip, additive Gaussian noise, exposure, rotate -- applied
// to images [S,H,W,3] (0..255) and one-hot masks [S,H,W,2] resident in HBM.  The reference runs them in numpy / scipy on the host
// (27 ms for one cubic rotation, 7 ms for one noise field of a 224x224 image: 165 images/s against > 2 600 images/s of the device loop).
//
// Division of labour: the DRAWS stay on the host in the reference's own order (mliis_amd/augment.py: which operations, how many, their
// scalar parameters -- a few numbers per sample); this file does the pixel work.  One launch = one STAGE of a mini-batch: sample b
// applies its stage-k operation `ops[b]` (identity when it has fewer), reading the previous stage's buffer and writing the next one
// (ping-pong; the host arranges that the last stage lands in the learner's batch slots).  What is not draw-identical: the noise field
// and the noise fill of a `constant`-mode rotation come from Philox4x32-10 on the device (seeded per sample by a host draw) instead of
// numpy's Mersenne Twister -- same distributions (N(0, sd) per element; U{0..255}); and the image rotation interpolates with the
// Keys cubic convolution kernel (a = -1/2) where scipy.ndimage.rotate evaluates a prefiltered cubic B-spline -- both interpolate the
// samples; on smooth images they agree to < 1 grey level (tests/test_augment_gpu.py).  Mask rotation (nearest) and scipy's boundary
// modes -- reflect (d c b a | a b c d), mirror (d c b | a b c d), wrap (legacy: coordinates modulo n - 1), constant (cval outside
// [0, n - 1]) -- are reproduced exactly.
#include "common.hpp"

namespace mliis {

struct AugOp {          // 48 bytes, mirrors mliis_amd/augment.py:DeviceOps
  int op;               // 0 copy, 1 erase, 2 translate, 3 flip, 4 noise, 5 exposure, 6 rotate
  int i0, i1, i2, i3;
  float f0, f1, f2, f3;
  unsigned seed_lo, seed_hi;
  int src;              // source sample index in the input buffer
};

__device__ __forceinline__ void aug_philox(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned* out) {
  constexpr unsigned M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    const unsigned hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += W0; k1 += W1;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// scipy.ndimage boundary modes for an integer tap index
__device__ __forceinline__ int aug_tap(int x, int n, int mode) {
  if (mode == 0) {            // reflect: d c b a | a b c d | d c b a
    const int p = 2 * n;
    x %= p;
    if (x < 0) x += p;
    return x < n ? x : p - 1 - x;
  }
  if (mode == 3) {            // wrap: taps periodic with period n (the float coordinate was folded modulo n - 1 before)
    x %= n;
    return x < 0 ? x + n : x;
  }
  if (n == 1) return 0;       // mirror (also the taps of `constant` inside the image): d c b | a b c d | c b a
  const int p = 2 * n - 2;
  x %= p;
  if (x < 0) x += p;
  return x < n ? x : p - x;
}
__device__ __forceinline__ float aug_keys(float t) {   // cubic convolution kernel, a = -1/2
  t = fabsf(t);
  if (t <= 1.0f) return (1.5f * t - 2.5f) * t * t + 1.0f;
  if (t < 2.0f) return ((-0.5f * t + 2.5f) * t - 4.0f) * t + 2.0f;
  return 0.0f;
}

__global__ __launch_bounds__(256) void augment_stage_k(const float* __restrict__ xin, const float* __restrict__ yin, float* __restrict__ xout,
                                                       float* __restrict__ yout, const AugOp* __restrict__ ops, int H, int W, int out_base) {
  const int b = blockIdx.y;
  const AugOp o = ops[b];
  const int pix = blockIdx.x * 256 + threadIdx.x;
  if (pix >= H * W) return;
  const int h = pix / W, w = pix - h * W;
  const float* xs = xin + (long long)o.src * H * W * 3;
  const float* ys = yin + (long long)o.src * H * W * 2;
  float* xd = xout + ((long long)(out_base + b) * H * W + pix) * 3;
  float* yd = yout + ((long long)(out_base + b) * H * W + pix) * 2;
  float v0, v1, v2, m0, m1;
  auto fetch = [&](int hh, int ww) {
    const float* px = xs + ((long long)hh * W + ww) * 3;
    const float* py = ys + ((long long)hh * W + ww) * 2;
    v0 = px[0]; v1 = px[1]; v2 = px[2];
    m0 = py[0]; m1 = py[1];
  };
  switch (o.op) {
    case 1: {   // erase: box [i0, i0 + i2) x [i1, i1 + i3) <- grey value f0, mask <- background (np_augmenters.py:22-38)
      fetch(h, w);
      if (h >= o.i0 && h < o.i0 + o.i2 && w >= o.i1 && w < o.i1 + o.i3) {
        v0 = v1 = v2 = o.f0;
        m0 = 1.0f; m1 = 0.0f;
      }
      break;
    }
    case 2: {   // translate (:48-99): i0 = "vertical" flag, i1 = direction, i2 = shift, i3 = wrap-around, f0..f2 = band colour.
      // Quirk kept: the "left/right" branch rolls the ROW axis and paints COLUMNS, the "up/down" branch rolls the column axis and paints rows.
      const int s = o.i2;
      if (!o.i0) {                       // shift_img_lr: np.roll(axis 0, +-s); band = columns [:s] (direction) or [-s:]
        int hs = (o.i1 ? h - s : h + s) % H;
        if (hs < 0) hs += H;
        fetch(hs, w);
        if (!o.i3 && (o.i1 ? w < s : w >= W - s)) { v0 = o.f0; v1 = o.f1; v2 = o.f2; m0 = 1.0f; m1 = 0.0f; }
      } else {                           // shift_img_ud: np.roll(axis 1, +-s); band = rows [-s:] (direction) or [:s]
        int wsrc = (o.i1 ? w - s : w + s) % W;
        if (wsrc < 0) wsrc += W;
        fetch(h, wsrc);
        if (!o.i3 && (o.i1 ? h >= H - s : h < s)) { v0 = o.f0; v1 = o.f1; v2 = o.f2; m0 = 1.0f; m1 = 0.0f; }
      }
      break;
    }
    case 3: fetch(h, W - 1 - w); break;   // fliplr (:41-44)
    case 4: {   // additive Gaussian noise, sd f0 (:9-12): three normals per pixel (Box-Muller on Philox), clipped to 0..255
      fetch(h, w);
      unsigned r[4];
      aug_philox((unsigned)pix, 0u, 0x6e6f6973u, 0u, o.seed_lo, o.seed_hi, r);
      const float u1 = ((float)(r[0] >> 8) + 0.5f) * (1.0f / 16777216.0f), u2 = (float)(r[1] >> 8) * (1.0f / 16777216.0f);
      const float u3 = ((float)(r[2] >> 8) + 0.5f) * (1.0f / 16777216.0f), u4 = (float)(r[3] >> 8) * (1.0f / 16777216.0f);
      const float ra = sqrtf(-2.0f * logf(u1)), rb = sqrtf(-2.0f * logf(u3));
      const float tw = 6.283185307179586f;
      v0 = fminf(fmaxf(v0 + o.f0 * ra * cosf(tw * u2), 0.0f), 255.0f);
      v1 = fminf(fmaxf(v1 + o.f0 * ra * sinf(tw * u2), 0.0f), 255.0f);
      v2 = fminf(fmaxf(v2 + o.f0 * rb * cosf(tw * u4), 0.0f), 255.0f);
      break;
    }
    case 5:     // exposure: one offset for the whole image (:15-18)
      fetch(h, w);
      v0 = fminf(fmaxf(v0 + o.f0, 0.0f), 255.0f);
      v1 = fminf(fmaxf(v1 + o.f0, 0.0f), 255.0f);
      v2 = fminf(fmaxf(v2 + o.f0, 0.0f), 255.0f);
      break;
    case 6: {   // rotate about the centre by f0 degrees (:102-131): scipy.ndimage.rotate(reshape=False), modes i0 = 0 reflect | 1 constant |
      // 2 mirror | 3 wrap; image cubic, mask nearest; constant mode: cval f1 outside, or (i1) Philox noise U{0..255} per element
      const int mode = o.i0;
      const float a = o.f0 * 0.017453292519943295f, c = cosf(a), s = sinf(a);
      const float ch = 0.5f * (float)(H - 1), cw = 0.5f * (float)(W - 1);
      float ih = c * ((float)h - ch) + s * ((float)w - cw) + ch;
      float iw = -s * ((float)h - ch) + c * ((float)w - cw) + cw;
      const bool hole = mode == 1 && !(ih >= 0.0f && ih <= (float)(H - 1) && iw >= 0.0f && iw <= (float)(W - 1));
      if (hole) {
        m0 = 1.0f; m1 = 0.0f;
        if (o.i1) {
          unsigned r[4];
          aug_philox((unsigned)pix, 0u, 0x726f7461u, 0u, o.seed_lo, o.seed_hi, r);
          v0 = (float)(r[0] >> 24); v1 = (float)(r[1] >> 24); v2 = (float)(r[2] >> 24);
        } else {
          v0 = v1 = v2 = o.f1;
        }
        break;
      }
      if (mode == 3) {   // legacy scipy 'wrap': the coordinate is folded modulo n - 1
        const float ph = (float)(H - 1), pw = (float)(W - 1);
        ih -= ph * floorf(ih / ph);
        iw -= pw * floorf(iw / pw);
      }
      const int tmode = mode == 1 ? 2 : mode;   // inside the image a constant-mode rotation reads mirrored taps
      {
        const int rh = aug_tap((int)floorf(ih + 0.5f), H, tmode), rw = aug_tap((int)floorf(iw + 0.5f), W, tmode);
        const float* py = ys + ((long long)rh * W + rw) * 2;
        m0 = py[0]; m1 = py[1];
      }
      const int fh = (int)floorf(ih), fw = (int)floorf(iw);
      float wy[4], wx[4];
      int ty[4], tx[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        wy[d] = aug_keys(ih - (float)(fh - 1 + d));
        wx[d] = aug_keys(iw - (float)(fw - 1 + d));
        ty[d] = aug_tap(fh - 1 + d, H, tmode);
        tx[d] = aug_tap(fw - 1 + d, W, tmode);
      }
      v0 = v1 = v2 = 0.0f;
#pragma unroll
      for (int dy = 0; dy < 4; ++dy)
#pragma unroll
        for (int dx = 0; dx < 4; ++dx) {
          const float* px = xs + ((long long)ty[dy] * W + tx[dx]) * 3;
          const float wgt = wy[dy] * wx[dx];
          v0 = fmaf(wgt, px[0], v0);
          v1 = fmaf(wgt, px[1], v1);
          v2 = fmaf(wgt, px[2], v2);
        }
      break;
    }
    default: fetch(h, w); break;   // copy
  }
  xd[0] = v0; xd[1] = v1; xd[2] = v2;
  yd[0] = m0; yd[1] = m1;
}

}  // namespace mliis

using namespace mliis;

extern "C" {

// One augmentation stage of a mini-batch of B samples: sample b reads image / mask ops[b].src of (xin, yin) and writes sample
// out_base + b of (xout, yout); ops = device array of B 48-byte records (mliis_amd/augment.py: DeviceOps).  The buffers must not alias.
int mliis_augment_stage(const float* xin, const float* yin, float* xout, float* yout, const void* ops, int B, int H, int W, int out_base,
                        hipStream_t stream) {
  MLIIS_REQUIRE(xin && yin && xout && yout && ops && B > 0 && H > 1 && W > 1 && out_base >= 0, MLIIS_ERR_ARG, "augment_stage: bad arguments");
  MLIIS_REQUIRE(xin != xout && yin != yout, MLIIS_ERR_ARG, "augment_stage: input and output buffers must differ (gather operations)");
  hipLaunchKernelGGL(augment_stage_k, dim3((H * W + 255) / 256, B), dim3(256), 0, stream, xin, yin, xout, yout,
                     reinterpret_cast<const AugOp*>(ops), H, W, out_base);
  MLIIS_CHECK_LAUNCH("augment_stage");
  return MLIIS_OK;
}
}
