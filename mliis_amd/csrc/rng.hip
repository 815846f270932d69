// Mask generation for the stochastic ops of an inner step, inside the step's HIP graph (no host RNG, no torch op on the path):
//   * drop-connect (models/efficientnet/utils.py:157-170): per block and image  scale = floor(keep + u) / keep,  u ~ U[0,1);
//   * dropout sites (tf.layers.dropout: final-layer dropout models/efficientlab.py:94-100,161-162; the ASPP's four rate-0.5 sites
//     :248-289): per element  mask = (u < keep) / keep.
// Generator: Philox4x32-10 (Salmon et al., SC'11), counter-based: key = the learner's 64-bit seed, counter = (step, launch-local index,
// stream id).  `state` = device uint32[4] {seed lo, seed hi, step, arrivals}: every launch reads `step`; the workgroup that arrives last
// (atomic ticket) increments it and resets the ticket, so a replayed graph draws a fresh stream each step without any host write.
// The reference seeds neither TF nor numpy (run_metasegnet.py:43 seeds only `random`), so there is no reference stream to match:
// parity tests inject masks; tests/test_ops_gpu.py checks the kernel against a Python restatement of Philox bit for bit.
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

__global__ __launch_bounds__(256) void rng_masks_k(unsigned* __restrict__ state, MaskJobs jobs, int total_blocks) {
  const unsigned k0 = state[0], k1 = state[1], step = state[2];
  long long b = blockIdx.x;
#pragma unroll 1
  for (int ji = 0; ji < jobs.n; ++ji) {
    const MaskJob J = jobs.j[ji];
    if (J.out == nullptr) continue;
    const long long quads = (J.numel + 3) >> 2;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < quads; q += (long long)gridDim.x * 256) {
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
  (void)b;
  // advance the step once every workgroup has READ it: each workgroup read `step` before taking its ticket
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned ticket = atomicAdd(&state[3], 1u);
    if (ticket == (unsigned)total_blocks - 1u) {
      state[3] = 0u;
      state[2] = step + 1u;
    }
  }
}

}  // namespace mliis

using namespace mliis;

extern "C" {

// state: device uint32[4] {seed lo, seed hi, step, 0}.  Up to six mask tensors per launch: job i draws from Philox stream i of the
// current step.  keeps[i] (device, nullable) overrides keep[i] per row of row_len[i] elements.
int mliis_rng_masks(unsigned* state, int njobs, float* const* outs, const long long* numels, const float* keep, const float* const* keeps,
                    const int* row_len, const int* floor_form, hipStream_t stream) {
  MLIIS_REQUIRE(state && njobs >= 1 && njobs <= kMaxMaskJobs && outs && numels && keep && keeps && row_len && floor_form, MLIIS_ERR_ARG,
                "rng_masks: bad arguments (1..%d jobs)", kMaxMaskJobs);
  MaskJobs jobs;
  jobs.n = njobs;
  long long most = 0;
  for (int i = 0; i < njobs; ++i) {
    MLIIS_REQUIRE(numels[i] >= 0 && (keeps[i] == nullptr || row_len[i] > 0), MLIIS_ERR_ARG, "rng_masks: job %d: bad size", i);
    MLIIS_REQUIRE(keeps[i] != nullptr || (keep[i] > 0.0f && keep[i] <= 1.0f), MLIIS_ERR_ARG, "rng_masks: job %d: keep probability must be in (0, 1]", i);
    jobs.j[i] = MaskJob{outs[i], numels[i], keep[i], keeps[i], row_len[i] > 0 ? row_len[i] : 1, floor_form[i]};
    if (outs[i] != nullptr && numels[i] > most) most = numels[i];
  }
  int blocks = (int)((most / 4 + 255) / 256);
  if (blocks < 1) blocks = 1;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(rng_masks_k, dim3(blocks), dim3(256), 0, stream, state, jobs, blocks);
  MLIIS_CHECK_LAUNCH("rng_masks");
  return MLIIS_OK;
}
}
