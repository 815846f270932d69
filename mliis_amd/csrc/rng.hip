// Mask generation for the stochastic ops of an inner step, inside the step's HIP graph (no host RNG, no torch op on the path):
//   * drop-connect (models/efficientnet/utils.py:157-170): per block and image  scale = floor(keep + u) / keep,  u ~ U[0,1);
//   * dropout sites (tf.layers.dropout: final-layer dropout models/efficientlab.py:94-100,161-162; the ASPP's four rate-0.5 sites
//     :248-289): per element  mask = (u < keep) / keep.
// Generator: Philox4x32-10 (Salmon et al., SC'11), counter-based: key = the learner's 64-bit seed, counter = (step, launch-local index,
// stream id).  `state` = device uint32[4] {seed lo, seed hi, step, arrivals}: every launch reads `step`; the workgroup that arrives last
// (atomic ticket) increments it and resets the ticket, so a replayed graph draws a fresh stream each step without any host write.
// The reference seeds neither TF nor numpy (run_metasegnet.py:43 seeds only `random`), so there is no reference stream to match:
// parity tests inject masks; tests/test_ops_gpu.py checks the kernel against a Python restatement of Philox bit for bit.
#include "rng_masks.hpp"

namespace mliis {

__global__ __launch_bounds__(256) void rng_masks_k(unsigned* __restrict__ state, MaskJobs jobs, int total_blocks) {
  rng_masks_body(state, jobs, (int)blockIdx.x, total_blocks);
}

int rng_make_jobs(const char* name, int njobs, float* const* outs, const long long* numels, const float* keep, const float* const* keeps,
                  const int* row_len, const int* floor_form, MaskJobs* jobs) {
  MLIIS_REQUIRE(njobs >= 1 && njobs <= kMaxMaskJobs && outs && numels && keep && keeps && row_len && floor_form, MLIIS_ERR_ARG,
                "%s: bad arguments (1..%d mask jobs)", name, kMaxMaskJobs);
  jobs->n = njobs;
  long long most = 0;
  for (int i = 0; i < njobs; ++i) {
    MLIIS_REQUIRE(numels[i] >= 0 && (keeps[i] == nullptr || row_len[i] > 0), MLIIS_ERR_ARG, "%s: job %d: bad size", name, i);
    MLIIS_REQUIRE(keeps[i] != nullptr || (keep[i] > 0.0f && keep[i] <= 1.0f), MLIIS_ERR_ARG, "%s: job %d: keep probability must be in (0, 1]", name, i);
    jobs->j[i] = MaskJob{outs[i], numels[i], keep[i], keeps[i], row_len[i] > 0 ? row_len[i] : 1, floor_form[i]};
    if (outs[i] != nullptr && numels[i] > most) most = numels[i];
  }
  int blocks = (int)((most / 4 + 255) / 256);
  if (blocks < 1) blocks = 1;
  if (blocks > 1024) blocks = 1024;
  return blocks;
}

}  // namespace mliis

using namespace mliis;

extern "C" {

// state: device uint32[4] {seed lo, seed hi, step, 0}.  Up to six mask tensors per launch: job i draws from Philox stream i of the
// current step.  keeps[i] (device, nullable) overrides keep[i] per row of row_len[i] elements.
int mliis_rng_masks(unsigned* state, int njobs, float* const* outs, const long long* numels, const float* keep, const float* const* keeps,
                    const int* row_len, const int* floor_form, hipStream_t stream) {
  MLIIS_REQUIRE(state, MLIIS_ERR_ARG, "rng_masks: null state");
  MaskJobs jobs;
  const int blocks = rng_make_jobs("rng_masks", njobs, outs, numels, keep, keeps, row_len, floor_form, &jobs);
  if (blocks < 0) return blocks;
  hipLaunchKernelGGL(rng_masks_k, dim3(blocks), dim3(256), 0, stream, state, jobs, blocks);
  MLIIS_CHECK_LAUNCH("rng_masks");
  return MLIIS_OK;
}
}
