// Squeeze-excite weight gradients (efficientnet_model.py:238-251): dw1[c][j] = sum_n s[n][c] dpre1[n][j], dw2[j][c] = sum_n
// swish(hpre[n][j]) dpre2[n][c], db2 = sum_n dpre2, db1 = sum_n dpre1 -- one thread per element, images in order (deterministic).
// Shared by se.hip (its own launches) and bn.hip (mliis_fold_batched carries the batched form as extra workgroups of the slab fold:
// nothing but the optimizer reads these gradients, and the fold is the launch in front of it).
#pragma once
#include "common.hpp"

namespace mliis {

// one thread per weight element, loops over images (deterministic)
__device__ __forceinline__ void se_wgrad_elem(int i, const float* __restrict__ s, const float* __restrict__ hpre,
                                              const float* __restrict__ dpre1, const float* __restrict__ dpre2,
                                              float* __restrict__ dw1, float* __restrict__ db1, float* __restrict__ dw2,
                                              float* __restrict__ db2, int N, int C, int R) {
  const int CR = C * R;
  const int total = 2 * CR + C + R;
  if (i >= total) return;
  // sums over the images, eight at a time with every load of a batch issued first (fixed order: deterministic)
  auto dot_n = [&](const float* __restrict__ pa, long long sa, const float* __restrict__ pb, long long sb, bool swish_a) {
    float a = 0.f;
    for (int n0 = 0; n0 < N; n0 += 8) {
      float x[8], y[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int n = n0 + u < N ? n0 + u : N - 1;
        x[u] = pa[n * sa];
        y[u] = pb != nullptr ? pb[n * sb] : 1.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) a = fmaf(n0 + u < N ? (swish_a ? swish_f(x[u]) : x[u]) : 0.f, y[u], a);
    }
    return a;
  };
  if (i < CR) {  // dw1[c][j] = sum_n s[n][c] * dpre1[n][j]
    const int c = i / R, j = i - c * R;
    dw1[i] = dot_n(s + c, C, dpre1 + j, R, false);
  } else if (i < 2 * CR) {  // dw2[j][c] = sum_n swish(hpre[n][j]) * dpre2[n][c]
    const int k = i - CR;
    const int j = k / C, c = k - j * C;
    dw2[k] = dot_n(hpre + j, R, dpre2 + c, C, true);
  } else if (i < 2 * CR + C) {
    const int c = i - 2 * CR;
    db2[c] = dot_n(dpre2 + c, C, nullptr, 0, false);
  } else {
    const int j = i - 2 * CR - C;
    db1[j] = dot_n(dpre1 + j, R, nullptr, 0, false);
  }
}


// desc: device int64 [ndesc][12] = {s, hpre, dpre1, dpre2, dw1, db1, dw2, db2 (device addresses), N, C, R, tile_begin}; a tile = 256
// consecutive elements of the descriptor whose tile range contains it
__device__ __forceinline__ void se_wgrad_tile(const long long* __restrict__ desc, int ndesc, long long tile) {
  int lo = 0, hi = ndesc - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (desc[(long long)mid * 12 + 11] <= tile) lo = mid; else hi = mid - 1;
  }
  const long long* d = desc + (long long)lo * 12;
  se_wgrad_elem((int)(tile - d[11]) * 256 + threadIdx.x, reinterpret_cast<const float*>(d[0]), reinterpret_cast<const float*>(d[1]),
                reinterpret_cast<const float*>(d[2]), reinterpret_cast<const float*>(d[3]), reinterpret_cast<float*>(d[4]),
                reinterpret_cast<float*>(d[5]), reinterpret_cast<float*>(d[6]), reinterpret_cast<float*>(d[7]), (int)d[8], (int)d[9],
                (int)d[10]);
}

}  // namespace mliis
