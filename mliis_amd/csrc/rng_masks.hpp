// Device side of the mask generator (rng.hip): shared with conv_gemm.hip, whose weight-shadow launch can carry the mask jobs of a
// training step as extra workgroups (mliis_weight_shadows_rng: one launch less per step).
#pragma once
#include "common.hpp"

namespace mliis {

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned* out) {
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
// 24 random bits -> [0, 1): every value is exactly representable, 1.0 is never produced
__device__ __forceinline__ float u01(unsigned x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }

struct MaskJob {
  float* out;        // nullable: job skipped
  long long numel;
  float keep;        // keep probability (scalar jobs)
  const float* keeps;   // nullable: per-row keep probabilities, row = index / row_len (drop-connect: one row per block)
  int row_len;
  int floor_form;    // 1: floor(keep + u) / keep (drop-connect);  0: (u < keep) / keep (dropout)
};
constexpr int kMaxMaskJobs = 6;
struct MaskJobs {
  MaskJob j[kMaxMaskJobs];
  int n;
};

// The work of workgroup b of nb (256 threads) of a mask launch, and the ticket that advances the step once every workgroup has read it.
__device__ __forceinline__ void rng_masks_body(unsigned* __restrict__ state, const MaskJobs& jobs, int b, int nb) {
  const unsigned k0 = state[0], k1 = state[1], step = state[2];
#pragma unroll 1
  for (int ji = 0; ji < jobs.n; ++ji) {
    const MaskJob J = jobs.j[ji];
    if (J.out == nullptr) continue;
    const long long quads = (J.numel + 3) >> 2;
    for (long long q = (long long)b * 256 + threadIdx.x; q < quads; q += (long long)nb * 256) {
      unsigned r[4];
      philox4x32_10((unsigned)q, (unsigned)(q >> 32), step, (unsigned)ji, k0, k1, r);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const long long i = q * 4 + e;
        if (i >= J.numel) break;
        const float keep = J.keeps != nullptr ? J.keeps[i / J.row_len] : J.keep;
        const float u = u01(r[e]);
        J.out[i] = J.floor_form ? floorf(keep + u) / keep : (u < keep ? 1.0f / keep : 0.0f);
      }
    }
  }
  // advance the step once every workgroup has READ it: each workgroup read `step` before taking its ticket
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned ticket = atomicAdd(&state[3], 1u);
    if (ticket == (unsigned)nb - 1u) {
      state[3] = 0u;
      state[2] = step + 1u;
    }
  }
}

// host side: the job table of a launch from the C-ABI arrays (returns the workgroup count, or a negative error code)
int rng_make_jobs(const char* name, int njobs, float* const* outs, const long long* numels, const float* keep, const float* const* keeps,
                  const int* row_len, const int* floor_form, MaskJobs* jobs);

}  // namespace mliis
