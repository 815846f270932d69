/* libmliis_hip -- C ABI of the MI355X (gfx950) inner-loop kernels for EfficientLab.
 *
 * The reference (ml4ai/mliis) has no FFI: its hot path crosses from Python into the TensorFlow 1.15 runtime at
 * `session.run(model.minimize_op, feed_dict=...)` (meta_learners/supervised_reptile/supervised_reptile/reptile.py:114-121,
 * 639-643).  Each entry point below replaces the TF op(s) named in its comment (file:line of the graph-construction call
 * that creates the op).  INTEGRATION.md shows the ctypes binding a maintainer would add.
 *
 * Conventions
 *   - all device buffers are caller-owned fp32, NHWC; a tensor [N,H,W,C] is passed as a 2-D view [rows = N*H*W][C] with a
 *     row stride `ld*` in floats (so slices of channel-concatenated buffers are passed without copies);
 *   - pointers 16-byte aligned, channel counts and leading dimensions multiples of 4, unless a comment says otherwise;
 *   - no allocation, no synchronisation, no global mutable state: every call only enqueues kernels on `stream`;
 *     scratch comes from a caller-provided workspace whose size the matching *_workspace_floats() returns;
 *   - returns MLIIS_OK (0) or a negative MLIIS_ERR_*; mliis_last_error() returns a thread-local message;
 *   - reductions are deterministic (two-stage, no float atomics): identical inputs give bit-identical outputs.
 *
 * Streams.  Every entry point is re-entrant and may be called concurrently on different streams (own buffers and workspaces per
 * stream).  One property of the hardware is part of the contract (measured on MI355X, ROCm 7.2: tools/interfere_probe.py,
 * profiles/r06_notes.md, tests/test_interference_gpu.py): while a wave that interleaves v_mfma_f32_16x16x32_{bf16,fp8} with LDS or
 * vector-memory instructions is resident on a CU, a packed fp32 instruction (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) with
 * op_sel:[0,1] executed by ANY OTHER wave on that CU can return a wrong low result.  What that means per entry point:
 *   - mliis_conv2d_fwd_x3, mliis_conv2d_bwd_data_x3 and the MLIIS_PREC_F32X3 groups of mliis_conv2d_bwd_filter_batched (the default fp32
 *     path of the decoder convs) occupy their CUs ALONE -- one 512-thread workgroup with the CU's whole register file -- so no kernel
 *     of any other stream, the caller's own or a foreign one (framework ops, RCCL), can share a CU with them: safe beside anything;
 *   - calls with MLIIS_PREC_BF16 / MLIIS_PREC_FP8 (and MLIIS_DT_BF16 storage, which implies bf16 operands) launch ordinary multi-
 *     workgroup-per-CU kernels that contain those matrix instructions.  This library contains no packed fp32 instruction of the affected
 *     form (checked on the built code objects: tests/test_build_cpu.py), so ITS kernels on other streams stay exact beside them; a FOREIGN
 *     kernel that does contain the form must not run concurrently with such a call on the same device (order it with an event);
 *   - everything else (MLIIS_PREC_FP32 and all non-GEMM entry points) neither disturbs nor is disturbed.
 */
#ifndef MLIIS_HIP_H
#define MLIIS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP__
typedef struct ihipStream_t* hipStream_t;
#endif

/* The library is built with -fvisibility=hidden: ONLY the entry points declared between this push and the pop at the end of the
 * file are dynamic symbols of libmliis_hip.so (tests/test_abi.py checks `nm -D` against this header). */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define MLIIS_OK 0
#define MLIIS_ERR_ARG (-1)
#define MLIIS_ERR_UNSUPPORTED (-2)
#define MLIIS_ERR_ALIGN (-3)
#define MLIIS_ERR_LAUNCH (-4)
#define MLIIS_ERR_WORKSPACE (-5)

int mliis_version(void);
const char* mliis_last_error(void);

/* ---- stem: (x - MEAN_RGB) / STDDEV_RGB (models/efficientlab.py:113-114) fused into conv 3x3 s2 SAME, 3 -> Co, no bias
 *      (models/efficientnet/efficientnet_model.py:359-366,411-412).  x: [S,H,W,3] in 0..255; img_idx (nullable, device
 *      int32[N]) selects the batch images out of the S resident shots.  mean3 / std3 are HOST float[3]. */
int mliis_stem_conv_fwd(const float* x, const int* img_idx, const float* w, float* z, int N, int H, int W, int Co,
                        const float* mean3, const float* std3, hipStream_t stream);
/*      mliis_stem_conv_fwd_stats: the same conv (z bit-identical) from a row-strip kernel that stages and normalises every input value
 *      once, plus the stage-1 statistics of z for the batch norm that follows (efficientnet_model.py:411-413): stats_part (nullable)
 *      [*nblk][2][Co] = {sum, sum of squares} per workgroup -- the layout of mliis_bn_stats_partial, at most ~256 blocks; capacity from
 *      mliis_stem_conv_fwd_stats_floats.  MLIIS_ERR_UNSUPPORTED (nothing launched) when a row's staging window does not fit LDS
 *      (W > ~400): take mliis_stem_conv_fwd + mliis_bn_stats_partial then. */
size_t mliis_stem_conv_fwd_stats_floats(int N, int H, int W, int Co);
int mliis_stem_conv_fwd_stats(const float* x, const int* img_idx, const float* w, float* z, int N, int H, int W, int Co,
                              const float* mean3, const float* std3, float* stats_part, size_t stats_floats, int* nblk,
                              hipStream_t stream);
size_t mliis_stem_conv_bwd_filter_workspace_floats(int N, int H, int W, int Co);
int mliis_stem_conv_bwd_filter(const float* x, const int* img_idx, const float* dz, float* dw, int N, int H, int W, int Co,
                               const float* mean3, const float* std3, float* ws, size_t ws_floats, hipStream_t stream);

/* ---- depthwise k x k (k 3|5, stride 1|2, TF-SAME, no bias): keras DepthwiseConv2D
 *      (models/efficientnet/efficientnet_model.py:190-196,271; utils.py:219-222).  H, W are the INPUT size in all three;
 *      w / dw are [k,k,C] (TF [k,k,C,1]). */
/*      stats_part (nullable): also emit the following batch norm's stage-1 statistics {sum y, sum y^2} as [*stats_nblk][2][C]
 *      (same contract as mliis_conv2d_fwd; needs ceil(N*Ho*ceil(Wo/4) / 32) * 2 * C floats) */
int mliis_dwconv_fwd(const float* x, const float* w, float* y, int N, int H, int W, int C, int k, int stride, float* stats_part,
                     size_t stats_floats, int* stats_nblk, hipStream_t stream);
int mliis_dwconv_bwd_data(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int k, int stride,
                          hipStream_t stream);
/*      same, when dx is the gradient w.r.t. swish(bn(bn_z)) (the expand branch of an MBConv block, efficientnet_model.py:175-182):
 *      the launch also emits stage 1 of that batch norm's backward -- per-block column sums {sum g, sum g*xhat}, g = dx *
 *      swish'(gamma*xhat + beta) -- into part [*nblk][2][C] for mliis_bn_bwd(stage1_part, stage1_nblk).  *nblk == 0: not produced
 *      (part too small: needs N * ceil(H/7) * ceil(W/16) * 2 * C floats); run mliis_bn_bwd without stage-1 partials then. */
int mliis_dwconv_bwd_data_bn(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int k, int stride, const float* bn_z,
                             const float* bn_mean, const float* bn_rstd, const float* bn_gamma, const float* bn_beta, float* part,
                             size_t part_floats, int* nblk, hipStream_t stream);
size_t mliis_dwconv_bwd_filter_workspace_floats(int N, int H, int W, int C, int k, int stride);
int mliis_dwconv_bwd_filter(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int k, int stride, float* ws,
                            size_t ws_floats, hipStream_t stream);

/* ---- on-device pixel half of the inner-loop augmentation (augmenters/np_augmenters.py:9-131: random eraser, translate, flip,
 *      Gaussian noise, exposure, rotate).  One launch = one stage of a mini-batch: sample b applies ops[b] (48-byte records {int op;
 *      int i0..i3; float f0..f3; uint32 seed_lo, seed_hi; int src}: 0 copy, 1 erase, 2 translate, 3 flip, 4 noise, 5 exposure,
 *      6 rotate -- parameter meaning in csrc/augment.hip) to image / mask `src` of (xin [.,H,W,3], yin [.,H,W,2]) and writes sample
 *      out_base + b of (xout, yout).  The draws stay on the host in the reference's order; noise fields come from Philox on the device. */
int mliis_augment_stage(const float* xin, const float* yin, float* xout, float* yout, const void* ops, int B, int H, int W, int out_base,
                        hipStream_t stream);

/* ---- masks of the stochastic ops of a training step, generated on the device inside the step's HIP graph (Philox4x32-10, counter =
 *      (element, step, job), key = seed): drop-connect  floor(keep + u) / keep  per block and image (models/efficientnet/utils.py:
 *      157-170) and dropout  (u < keep) / keep  per element (tf.layers.dropout: models/efficientlab.py:94-100,161-162,248-289).
 *      state = device uint32[4] {seed lo, seed hi, step, 0}; the launch advances `step` itself (last workgroup), so graph replays
 *      draw fresh masks.  Up to 6 jobs per launch; keeps[i] (device, nullable) gives a keep probability per row of row_len[i]
 *      elements (drop-connect: one row per block), else keep[i] applies; outs[i] == NULL skips a job (its stream index is kept). */
int mliis_rng_masks(unsigned* state, int njobs, float* const* outs, const long long* numels, const float* keep, const float* const* keeps,
                    const int* row_len, const int* floor_form, hipStream_t stream);

/* ---- the depthwise half of an MBConv block on the LARGE maps (dwmarch.hip): the depthwise conv with the batch norm + swish IN FRONT
 *      of it applied while the input is staged -- expand conv -> BN -> swish -> depthwise k x k (efficientnet_model.py:175-196,
 *      266-271; utils.py:87-134), and for block 0 the stem's BN -> swish (efficientnet_model.py:409-414).  A workgroup marches down
 *      the rows of a (image, 32 channels, column band) with the input rows in an LDS ring: z is read ONCE, a = swish(bn(z)) is never
 *      written, the backward is ONE pass over (dy, z).  H, W = the layer's INPUT size; w / dw [k,k,C]; k 3|5, stride 1|2.
 *      forward: bn_gamma == NULL: no batch norm (y = dwconv(z)).  bn_nblk > 0: bn_part [bn_nblk][2][C] are the stage-1 sums {sum z,
 *      sum z^2} of the producer (mliis_conv2d_fwd's stats_part): the launch folds them, writes bn_mean / bn_rstd (kept for the
 *      backward pass) and updates the moving averages (nullable pair; biased variance, utils.py:118-131).  bn_nblk == 0: bn_mean /
 *      bn_rstd are INPUTS (inference: moving statistics).  stats_part (nullable): the FOLLOWING batch norm's stage-1 sums {sum y,
 *      sum y^2} as [*stats_nblk][2][C], *stats_nblk = mliis_dwconv_bn_fwd_blocks(...).
 *      backward: dx = gradient w.r.t. a (bn_gamma == NULL: w.r.t. z itself); dw_part [*nblk][k*k][C] = slabs of the filter gradient
 *      (for mliis_fold_batched; dw != NULL: also folded into dw here); bn_part (nullable) [*nblk][2][C] = stage 1 of the batch norm's
 *      backward {sum g, sum g*xhat}, g = dx * swish'(gamma*xhat + beta), for mliis_bn_bwd(stage1_part, stage1_nblk);
 *      *nblk = mliis_dwconv_bn_bwd_blocks(...). */
int mliis_dwconv_bn_supported(int N, int H, int W, int C, int k, int stride);   /* 1: the marching entry points take this layer */
int mliis_dwconv_bn_fwd_blocks(int N, int H, int W, int C, int k, int stride);
int mliis_dwconv_bn_bwd_blocks(int N, int H, int W, int C, int k, int stride);
int mliis_dwconv_bn_fwd(const float* z, const float* bn_part, int bn_nblk, const float* bn_gamma, const float* bn_beta, float* bn_mean,
                        float* bn_rstd, float* bn_moving_mean, float* bn_moving_var, float eps, float momentum, const float* w, float* y,
                        int N, int H, int W, int C, int k, int stride, float* stats_part, size_t stats_floats, int* stats_nblk,
                        int in_dtype, int out_dtype, hipStream_t stream);
int mliis_dwconv_bn_bwd(const float* dy, const float* z, const float* bn_mean, const float* bn_rstd, const float* bn_gamma,
                        const float* bn_beta, const float* w, float* dx, float* dw, int N, int H, int W, int C, int k, int stride,
                        float* dw_part, size_t dw_part_floats, float* bn_part, size_t bn_part_floats, int* nblk, int dy_dtype, int zx_dtype,
                        hipStream_t stream);
/*      Storage types (MLIIS_DT_*): forward (in_dtype of z, out_dtype of y) in {(F32, F32), (F32, BF16), (BF16, BF16)}; backward (dy_dtype of
 *      dy [and z1], zx_dtype of z and dx) in {(F32, F32), (BF16, F32), (BF16, BF16)} -- the fp32 sides are a block's fp32 input / input
 *      gradient (block 0 behind the stem, blocks without an expand conv).  Statistics are formed from the values as stored -- with ONE
 *      exception: the 5x5 FORWARD kernel with a bf16 output sums its fp32 accumulators (rounding them first costs the kernel
 *      registers it does not have); the backward normalises the stored z1 with that mean / rstd.  oracle/efficientlab_ref.py models
 *      exactly this (batch_norm(stats_of=...)). */

/*      mliis_mbconv_dw_bwd_march: the same backward with the depthwise batch norm's (bn1, efficientnet_model.py:271) backward APPLY
 *      formed while its operands are staged: da2 = the project conv's backward-data output (gradient w.r.t. a1 * gate), z1 = bn1's
 *      input, gate / chan_add [N,C] and stage1 [stage1_nimg][2][C] from mliis_se_mlp_bwd_bn.  The launch forms
 *      dz1 = gamma1 rstd1 (g - mean(g) - xhat1 mean(g xhat1)), g = (da2 gate + chan_add) swish'(gamma1 xhat1 + beta1), on the fly (the
 *      tensor dz1 is never written), writes dgamma1 / dbeta1, and continues as mliis_dwconv_bn_bwd with the batch norm in front (bn0:
 *      z0, mean0 .. beta0): dx, dw_part, bn_part, *nblk. */
int mliis_mbconv_dw_bwd_march(const float* da2, const float* z1, const float* mean1, const float* rstd1, const float* gamma1, const float* beta1,
                              const float* gate, const float* chan_add, const float* stage1, int stage1_nimg, float* dgamma1, float* dbeta1,
                              const float* z0, const float* mean0, const float* rstd0, const float* gamma0, const float* beta0, const float* w,
                              float* dx, int N, int H, int W, int C, int k, int stride, float* dw_part, size_t dw_part_floats, float* bn_part,
                              size_t bn_part_floats, int* nblk, int dy_dtype, int zx_dtype, hipStream_t stream);

/* ---- the depthwise half of an MBConv block on SMALL maps in ONE launch per direction (mbconv_small.hip): expand BN -> swish ->
 *      depthwise k x k (stride 1) -> BN -> swish -> squeeze-excite mean (efficientnet_model.py:183-200,266-271,247; utils.py:87-134)
 *      and the whole backward of that chain.  Every op in it is per channel, so a workgroup that owns a group of `group_width`
 *      channels over all of [N,H,W] needs no grid-wide dependency.  group_width: 0 = the planner's choice
 *      (mliis_mbconv_dw_small_group_width: quads -- 16-byte accesses -- except pairs for 5x5 layers whose C / 2 workgroups fit
 *      one round of the chip), or 2 | 4.  Eligible shapes (else MLIIS_ERR_UNSUPPORTED, nothing launched; use
 *      the op-by-op entry points): stride 1, k 3|5, C % 4 == 0, N*H*W <= 2048, N*ceil(H/4)*W <= 512.
 *      forward: z0 = expand conv output with its stage-1 statistics part0 [nblk0][2][C] (mliis_conv2d_fwd's stats_part); writes the
 *      batch statistics of both batch norms (mean / rstd, for the backward pass), updates both pairs of moving averages (nullable),
 *      z1 = depthwise output, a1 = swish(bn1(z1)), s [N,C] = per-image mean of a1, and (nullable) a0 = swish(bn0(z0)).
 *      backward: da2 = gradient w.r.t. a1 * gate (the project conv's backward-data), gate / chan_add [N,C] as in mliis_bn_bwd's
 *      chan_scale / chan_add (nullable); writes both BN parameter gradients, the COMPLETE depthwise filter gradient dw [k,k,C] (no
 *      slabs) and dz0 = gradient w.r.t. the expand conv's output.
 *      Blocked operands (group-blocked layout [C / group_width][N*H*W][group_width], what a workgroup reads contiguously): forward
 *      z0_blocked = a buffer that receives a copy of z0 in that layout -- or z0 itself when the expand conv wrote it blocked
 *      (MLIIS_DT_BLOCKED: z0_blocked == z0, nothing is copied); z1_blocked != 0: z1 is written blocked.  backward: z0_blocked read
 *      instead of z0; z1_blocked bit 0: z1 is blocked, bit 1: da2 is blocked (mliis_conv2d_bwd_data_gate with MLIIS_DT_BLOCKED). */
int mliis_mbconv_dw_small_supported(int N, int H, int W, int C, int k, int stride);
int mliis_mbconv_dw_small_group_width(int C, int k);
int mliis_mbconv_dw_fwd_small(const float* z0, const float* part0, int nblk0, const float* gamma0, const float* beta0, float* mean0,
                              float* rstd0, float* moving_mean0, float* moving_var0, const float* w, const float* gamma1, const float* beta1,
                              float* mean1, float* rstd1, float* moving_mean1, float* moving_var1, float* a0, float* z1, float* a1, float* s,
                              int N, int H, int W, int C, int k, float eps, float momentum, int group_width, int act_dtype,
                              float* z0_blocked, int z1_blocked, hipStream_t stream);
int mliis_mbconv_dw_bwd_small(const float* da2, const float* gate, const float* chan_add, const float* z1, const float* mean1,
                              const float* rstd1, const float* gamma1, const float* beta1, const float* w, const float* z0, const float* mean0,
                              const float* rstd0, const float* gamma0, const float* beta0, float* dgamma1, float* dbeta1, float* dw,
                              float* dgamma0, float* dbeta0, float* dz0, int N, int H, int W, int C, int k, int group_width, int act_dtype,
                              const float* z0_blocked, int z1_blocked, hipStream_t stream);
/*      z0_blocked (nullable) / z1_blocked: the two saved tensors of the layer in the GROUP-BLOCKED layout [C / group_width][N H W][group_width]
 *      -- the forward launch writes a blocked copy of z0 and / or writes z1 blocked, the backward launch of the same layer (same
 *      group_width) reads them: a workgroup's channel group is then contiguous (196 cache lines per tensor instead of 16 bytes of each of
 *      1568 lines), which is what the backward pass's cold re-read of the two forward tensors costs (profiles/r04_notes.md). */

/* ---- dense conv (k 1|3, stride 1, TF-SAME, dilation >= 1, optional bias) on the fp32 matrix cores:
 *      tf.layers.Conv2D 1x1 expand/project (efficientnet_model.py:175-182,225-232) and tf.layers.conv2d of the RSD decoder /
 *      ASPP (models/efficientlab.py:185-190,218-224,258-283).  w is TF HWIO [k,k,Cin,Cout]; the forward reads its K-contiguous
 *      copy wt [k,k,Cout,Cin] (mliis_transpose_weights, once per weight update), backward-data reads w itself.  `accumulate` != 0
 *      adds into the destination.  ws may be NULL (disables split-K). */
size_t mliis_conv2d_workspace_floats(int Nimg, int H, int W, int Cred, int Nout, int ksize);
/*      `precision` (per call; nothing process-wide): operand precision of the matrix cores.  MLIIS_PREC_FP32 = fp32 operands
 *      (BASELINE configs 1-3), MLIIS_PREC_BF16 = operands rounded to bf16 in registers with fp32 accumulation (configs 4-5 flavour:
 *      tensors, BN, depthwise and the optimiser stay fp32). */
/*      MLIIS_PREC_FP8 (BASELINE configs[4], "fp8 MFMA on 1x1 pointwise convs"): forward 1x1 convs with OCP e4m3 operands
 *      (v_mfma_f32_16x16x32_fp8_fp8) -- activations times fp8_act_scale, weights times 2^floor(log2(224 / *fp8_w_amax)), both saturated
 *      at +-448 and converted in registers, the fp32 accumulators divided by the two scales; *fp8_w_amax = max |w| of the weight
 *      tensor, written by mliis_transpose_weights.  A 3x3 forward call and every backward call given MLIIS_PREC_FP8 runs bf16. */
/*      Storage type of the EXPANDED tensors of an MBConv block (z0, z1, a1 and their gradients), `act_dtype` / `a_dtype` / `out_dtype`
 *      arguments below: MLIIS_DT_F32, or MLIIS_DT_BF16 = bf16 in HBM (BASELINE configs[3]: models/efficientnet/efficientnet_model.py:
 *      175-236, utils.py:87-134 run on bf16 activations there) -- the pointer then addresses 2-byte elements (leading dimensions stay in
 *      elements), loads widen exactly, stores round to nearest even, every sum / statistic / accumulator is fp32. */
#define MLIIS_DT_F32 0
#define MLIIS_DT_BF16 1
/* OR into y_dtype of mliis_conv2d_fwd / dx_dtype of mliis_conv2d_bwd_data_gate: the output is written in the group-blocked layout
 * [C / v][N*H*W][v] (v = 2 | 4, fp32 storage; ldy / lddx are then not used) that the small-map fused MBConv kernels read with
 * contiguous accesses (a workgroup there owns v channels over all pixels): mliis_mbconv_dw_fwd_small(z0 == z0_blocked),
 * mliis_mbconv_dw_bwd_small(z1_blocked bit 1).  Only calls that take the streamed 1x1 plan (mliis_conv2d_kernel_name:
 * conv1x1_stream_k) support it; others return MLIIS_ERR_UNSUPPORTED. */
#define MLIIS_DT_BLOCKED(v) ((v) << 8)
#define MLIIS_PREC_FP32 0
#define MLIIS_PREC_BF16 1
#define MLIIS_PREC_FP8 2
/*      MLIIS_PREC_F32X3 (mliis_conv2d_bwd_filter_batched only; the forward / backward-data form has its own entry points,
 *      mliis_conv2d_fwd_x3 / mliis_conv2d_bwd_data_x3): fp32-EQUIVALENT products on the bf16 matrix cores -- every fp32 operand value split
 *      exactly into three bf16 terms, six of the nine term products, fp32 accumulation -- for the groups of 128-channel tiles; the
 *      other groups run the fp32 instruction under this value too. */
#define MLIIS_PREC_F32X3 3
/*      tiling chosen for a fwd / bwd-data call (row-tile factor, column tiles, split-K factor) and the name of the kernel
 *      instantiation it launches, as rocprofv3 prints it (profiling aids: bench.py matches its live timings to the trace) */
int mliis_conv2d_plan(int Nimg, int H, int W, int Cred, int Nout, int ksize, int* tm, int* nt, int* splits);
/*      workgroups of a 1x1 kernel instance the runtime fits on one CU (kind 0: conv1x1_stream_k<kc, nt>, 1: conv1x1_ksplit_k<kc, nt, 8>):
 *      the planners launch two per CU, a test checks that two fit */
int mliis_conv1x1_occupancy(int kind, int kc, int nt, int* blocks_per_cu);
int mliis_conv2d_kernel_name(int Nimg, int H, int W, int Cred, int Nout, int ksize, int has_scale, int precision, char* buf, size_t buf_len);
/*      stats_part (nullable): the epilogue also emits the following batch norm's stage-1 statistics -- per row-block column sums
 *      {sum v, sum v^2} (of swish(v) when stats_swish) as [*stats_nblk][2][Cout], written by the GEMM epilogue or, on a split-K
 *      plan, by the slab fold.  *stats_nblk == 0 means "not produced": the caller must run mliis_bn_stats_partial instead.
 *      Needs >= ceil(M/16) * 2 * Cout floats (at most one partial per 16-row group; most plans write one per 64 or 128 rows);
 *      requires accumulate == 0. */
/*      x_scale (nullable, [Nimg,Cin], 1x1 convs): x[m,c] is multiplied by x_scale[image(m),c] while it is staged -- the
 *      squeeze-excite gate (efficientnet_model.py:251) applied on the fly, so the gated tensor is never materialised. */
/*      The weight tensor has Cin_total input channels; the conv reads its channels [ci_begin, ci_begin+Cin) against x's Cin
 *      channels.  border_bias (nullable, [Nimg,9,Cout], 3x3 dilation 1 only): per-pixel bias selected by the pixel's border
 *      class -- the exact contribution of spatially constant input channels (mliis_rsd_pool_fwd). */
int mliis_conv2d_fwd(const float* x, int ldx, const float* x_scale, const float* wt, const float* bias,
                     const float* border_bias, float* y, int ldy, int Nimg, int H, int W, int Cin_total, int ci_begin, int Cin, int Cout,
                     int ksize, int dil, int accumulate, float* stats_part, int stats_swish, int* stats_nblk, float* ws,
                     size_t ws_floats, int precision, float fp8_act_scale, const float* fp8_w_amax, int x_dtype, int y_dtype, hipStream_t stream);
/*      mliis_conv2d_fwd_bnin: the 1x1 conv of the NEXT MBConv block's expand step (efficientnet_model.py:175-182) with the plain batch
 *      norm in front of it -- the previous block's project BN, its drop-connect scale and identity skip (efficientnet_model.py:
 *      283-288, utils.py:157-170) -- applied while the rows are loaded:
 *          a = ((z - mean) * rstd * gamma + beta) * img_scale[image] + res;   y = conv1x1(a, wt)
 *      The launch folds the batch norm's stage-1 partials bn_part [bn_nblk][2][Cin] (the stats_part of the conv that produced z),
 *      publishes mean / rstd (kept for the backward pass), advances the moving averages (nullable pair; biased variance) and writes
 *      the finished tensor a to a_out [N*H*W, Cin] (ld = lda_out: the block output -- residual of the next block, endpoint, X operand
 *      of the filter gradient).  Replaces mliis_bn_apply_fused (no activation) + mliis_conv2d_fwd: one launch instead of two, z read
 *      once.  img_scale [Nimg] and res (ld = ldr) nullable; z, res, a_out fp32; wt [Cout][Cin] (mliis_transpose_weights);
 *      stats_part / stats_swish / stats_nblk, precision, fp8 scales and y_dtype (MLIIS_DT_BF16, MLIIS_DT_BLOCKED) as for
 *      mliis_conv2d_fwd.  Only the streamed 1x1 plan implements it (Cin <= 112, N*H*W >= 1024: mliis_conv2d_fwd_bnin_ok returns 1);
 *      other shapes: MLIIS_ERR_UNSUPPORTED, nothing launched. */
int mliis_conv2d_fwd_bnin_ok(int Nimg, int H, int W, int Cin, int Cout);
int mliis_conv2d_fwd_bnin(const float* z, int ldz, const float* bn_part, int bn_nblk, float eps, float momentum, float* mean, float* rstd,
                          float* moving_mean, float* moving_var, const float* gamma, const float* beta, const float* img_scale,
                          const float* res, int ldr, float* a_out, int lda_out, const float* wt, float* y, int ldy, int Nimg, int H, int W,
                          int Cin, int Cout, float* stats_part, int stats_swish, int* stats_nblk, int precision, float fp8_act_scale,
                          const float* fp8_w_amax, int y_dtype, hipStream_t stream);
/*      batched HWIO -> HWOI copy of dense-conv weights between two arenas of identical layout; desc = device int32
 *      [ndesc][4] {offset, taps, Cin, Cout} */
/*      amax (nullable, device float[ndesc]): also max |w| per descriptor (the fp8 operand scale of MLIIS_PREC_FP8).
 *      total_tiles > 0: sum over the descriptors of taps * ceil(Cin / 32) * ceil(Cout / 32) -- one workgroup per 32 x 32 tile; 0: a
 *      fixed 224 x ndesc grid (the caller does not know the table's contents) */
int mliis_transpose_weights(const float* src, float* dst, const int* desc, int ndesc, long long total_tiles, float* amax, hipStream_t stream);
/*      mliis_weight_shadows: mliis_transpose_weights and mliis_x3_pack_weights (same arguments) in ONE launch -- the two per-step
 *      shadows of the weight arena; total_tiles > 0 required */
int mliis_weight_shadows(const float* src, float* dst, const int* desc, int ndesc, long long total_tiles, float* amax, void* x3_images,
                         const long long* x3_desc, int x3_ndesc, int x3_blocks, hipStream_t stream);
/*      ... and mliis_rng_masks (the masks of the training step: same rng_state / job arrays) in the SAME launch -- the three launches a
 *      step used to start with are one (x3_blocks may be 0: no weight images). */
int mliis_weight_shadows_rng(const float* src, float* dst, const int* desc, int ndesc, long long total_tiles, float* amax, void* x3_images,
                             const long long* x3_desc, int x3_ndesc, int x3_blocks, unsigned* rng_state, int njobs, float* const* outs,
                             const long long* numels, const float* keep, const float* const* keeps, const int* row_len, const int* floor_form,
                             hipStream_t stream);
/*      gradient w.r.t. input channels [ci_begin, ci_begin+Cin_out) of a conv whose weight has Cin_total input channels */
int mliis_conv2d_bwd_data(const float* dy, int lddy, const float* w, float* dx, int lddx, int Nimg, int H, int W, int Cin_total,
                          int ci_begin, int Cin_out, int Cout, int ksize, int dil, int accumulate, float* ws, size_t ws_floats,
                          int precision, int dy_dtype, int dx_dtype, hipStream_t stream);
/*      same, when dx is the gradient w.r.t. the OUTPUT of a plain batch norm over bn_x [M, Cin_out] (the project BN of the MBConv
 *      block in front, efficientnet_model.py:225-236; bn_img_scale = its drop-connect scales, nullable): on the plans that finish
 *      their rows inside one workgroup (the 28x28 / 14x14 maps) the launch also leaves stage 1 of that batch norm's backward
 *      {sum g, sum g * xhat} in part [*nblk][2][Cin_out] for mliis_bn_bwd(stage1_part, stage1_nblk).  *nblk == 0: not produced. */
int mliis_conv2d_bwd_data_bn(const float* dy, int lddy, const float* w, float* dx, int lddx, int Nimg, int H, int W, int Cin_total,
                             int ci_begin, int Cin_out, int Cout, int ksize, int dil, int accumulate, float* ws, size_t ws_floats,
                             int precision, const float* bn_x, int bn_ldx, const float* bn_mean, const float* bn_rstd,
                             const float* bn_img_scale, float* part, size_t part_floats, int* nblk, int dy_dtype, int dx_dtype, hipStream_t stream);
/*      same as mliis_conv2d_bwd_data, when dx is the gradient w.r.t. gate_x * gate[image] (the squeeze-excite gating in front of a
 *      project conv, efficientnet_model.py:251): on the streaming plan the launch also leaves the column sums of dx * gate_x per
 *      16-row group, split by image, in part [*groups][2][Cin_out]; mliis_se_mlp_bwd(dgate = part, dgate_row_groups = *groups) folds
 *      them -- the gate's gradient without a pass over the two tensors.  *groups == 0: not produced (use mliis_colsum). */
int mliis_conv2d_bwd_data_gate(const float* dy, int lddy, const float* w, float* dx, int lddx, int Nimg, int H, int W, int Cin_total,
                               int ci_begin, int Cin_out, int Cout, int ksize, int dil, float* ws, size_t ws_floats, int precision,
                               const float* gate_x, int gate_ldx, float* part, size_t part_floats, int* groups, int dy_dtype, int dx_dtype, hipStream_t stream);
/* ---- MLIIS_PREC_F32X3 for the long-K dense convs (the RSD decoder's 3x3 / 3x3-dilated convs, models/efficientlab.py:185-190,218-224;
 *      same products as tf.layers.conv2d in fp32): a kernel of its own whose B operand is a pre-split WEIGHT IMAGE.
 *      mliis_x3_pack_weights splits the weights of every listed conv direction once per inner step (after the optimizer step, like
 *      mliis_transpose_weights) into [K chunk of 32][16-column tile][term hi|mid|lo][lane group 4][column 16][8 bf16];
 *      desc: DEVICE int64 [ndesc][8] rows {source offset in theta (floats), taps, Cin_total, Cout, ci_begin, Cin (window),
 *      mode | first block << 8, image offset (bytes)} with mode 0 = forward (columns = Cout, K = (tap, ci of the window)) and mode 1 =
 *      backward-data (columns = the ci window, K = (tap, co)); an image takes mliis_x3_image_bytes(Cred, Nout, ksize) bytes and
 *      mliis_x3_image_blocks(...) workgroups of the pack launch (`first block` = running sum, total_blocks = their sum).
 *      mliis_conv2d_fwd_x3 / mliis_conv2d_bwd_data_x3: arguments as mliis_conv2d_fwd / mliis_conv2d_bwd_data with the image in place of
 *      wt / w (the channel window is the image's); activations are read straight from memory in matrix-core operand layout and split
 *      in registers; 256-row tiles, one 512-thread workgroup per CU, stream-K remainder with a deterministic fix-up launch (slabs in ws:
 *      mliis_conv2d_x3_workspace_floats); *stats_nblk = four blocks per 256-row tile.  Cred >= 32.
 *      Stream contract (see "Streams" at the top): the workgroup claims its CU's whole register file, nothing of another stream is ever
 *      co-resident with these kernels -- they may run beside any other stream's work.  `image` must be the image packed for exactly this
 *      (Cred window, Nout, ksize, direction); image_bytes = its size, checked against mliis_x3_image_bytes(Cred, Nout, ksize). */
size_t mliis_x3_image_bytes(int Cred, int Nout, int ksize);
int mliis_x3_image_blocks(int Cred, int Nout, int ksize);
int mliis_x3_pack_weights(const float* theta, void* images, const long long* desc, int ndesc, int total_blocks, hipStream_t stream);
size_t mliis_conv2d_x3_workspace_floats(int Nimg, int H, int W, int Cred, int Nout, int ksize);
int mliis_conv2d_x3_plan(int Nimg, int H, int W, int Cred, int Nout, int ksize, int* plan);
int mliis_conv2d_fwd_x3(const float* x, int ldx, const void* image, size_t image_bytes, const float* bias, const float* border_bias, float* y, int ldy, int Nimg,
                        int H, int W, int Cin, int Cout, int ksize, int dil, int accumulate, float* stats_part, int stats_swish,
                        int* stats_nblk, float* ws, size_t ws_floats, hipStream_t stream);
int mliis_conv2d_bwd_data_x3(const float* dy, int lddy, const void* image, size_t image_bytes, float* dx, int lddx, int Nimg, int H, int W, int Cin_out, int Cout,
                             int ksize, int dil, int accumulate, float* ws, size_t ws_floats, hipStream_t stream);
size_t mliis_conv2d_bwd_filter_workspace_floats(int Nimg, int H, int W, int Cin, int Cout, int ksize);
/*      writes rows [ci_begin, ci_begin+Cin) (per tap) of the full [k,k,Cin_total,Cout] gradient tensor dw */
int mliis_conv2d_bwd_filter(const float* x, int ldx, const float* x_scale, const float* dy, int lddy, float* dw, int Nimg, int H, int W,
                            int Cin_total, int ci_begin, int Cin, int Cout, int ksize, int dil, int accumulate, float* ws,
                            size_t ws_floats, int precision, hipStream_t stream);
/*      Weight gradients are off the critical path of the backward pass (nothing but the optimizer reads them): several
 *      mliis_conv2d_bwd_filter(dw = NULL) calls that resolve to the same kernel instantiation can be issued as ONE launch, so that
 *      the small-map layers share the chip.  mliis_conv2d_bwd_filter_plan: plan[0..6] = {TMF, NT, multitap, gx, gy, gz (slabs left
 *      in the workspace), rows_per_split} of a call.  mliis_conv2d_bwd_filter_batched: desc = DEVICE table int64 [nprob][16] rows
 *      {x, dy, x_scale (0: none), workspace, ldx, lddy, Nimg, H, W, Cin, Cout, ksize, dil, rows_per_split | multitap << 32,
 *       gx | gy << 20 | gz << 40, first workgroup of the problem in the grid}, blocks = sum of gx * gy * gz, all problems with the
 *      same (TMF, NT, x_scale present); each problem's slabs land in its workspace exactly as the single call leaves them. */
int mliis_conv2d_bwd_filter_plan(int Nimg, int H, int W, int Cin, int Cout, int ksize, int* plan);
int mliis_conv2d_bwd_filter_batched(const long long* desc, int nprob, int blocks, int tmf, int nt, int has_scale, int precision,
                                    hipStream_t stream);

/* ---- RSD pooled branch (models/efficientlab.py:192-197,220-224) without convolving it: the Cp spatially constant channels
 *      [c_begin, c_begin+Cp) of the 3x3 fuse conv's input become a per-image, per-border-class bias (forward) and need only
 *      per-image sums of the output gradient over the map, its border rows/columns and corners (backward).  tot[n][co] =
 *      per-image column sums of dz -- an OUTPUT of mliis_rsd_pool_bwd (formed by its border-sum launch).  dpool = (dL/dpool) / (H*W).
 *      mliis_rsd_concat_pool writes the module's input cat = [deep map, copied or bilinearly resized (efficientlab.py:205-206) | skip
 *      feature] (tf.concat, efficientlab.py:208) and, in the same pass, the per-image column sums of cat in *chunks partials per image,
 *      pool_part [N][*chunks][Cd+Cs]; mliis_rsd_pool_fwd folds them (pool = scale * sum; chunks = 1, scale = 1 for a finished vector),
 *      keeps the folded vectors in pool_out (nullable) for the backward pass and forms the border-class bias -- one launch each. */
size_t mliis_rsd_concat_pool_floats(int N, int H, int W, int C);
int mliis_rsd_concat_pool(const float* deep, int ld_deep, int Hi, int Wi, int Cd, const float* skip, int ld_skip, int Cs, float* cat, int ldcat,
                          int N, int H, int W, float* pool_part, size_t pool_part_floats, int* chunks, hipStream_t stream);
int mliis_rsd_pool_fwd(const float* pool_part, int chunks, float scale, float* pool_out, const float* w, float* border_bias, int N, int Cp,
                       int Cin_total, int c_begin, int Co, hipStream_t stream);
size_t mliis_rsd_pool_bwd_workspace_floats(int N, int Co);
int mliis_rsd_pool_bwd(const float* dz, int lddz, float* tot, const float* pool, const float* w, float* dw, float* dbias,
                       float* dpool, int N, int H, int W, int Cp, int Cin_total, int c_begin, int Co, float* ws, size_t ws_floats,
                       hipStream_t stream);

/* ---- batch norm, training mode (TpuBatchNormalization, models/efficientnet/utils.py:87-134; tf.layers.batch_normalization,
 *      models/efficientlab.py:190).  pre_swish: statistics/normalisation act on swish(x) (decoder order conv->swish->BN);
 *      post_swish: y = swish(bn(x)) (backbone order conv->BN->swish).  bn_stats also applies the moving-average update
 *      m -= (m - stat) * (1 - momentum) (moving_* nullable); unbiased_moving_var selects the fused-BN rule.
 *      bn_apply: y = [swish](gamma * xhat + beta) * img_scale[n] + res   (img_scale = drop-connect scale, utils.py:157-170;
 *      res = identity skip, efficientnet_model.py:281-288 / RSD residual, efficientlab.py:226-229).
 *      bn_bwd: upstream gradient = dy * img_scale[n] * chan_scale[n,c] + chan_add[n,c]  (SE gate / pooled gradient fused in). */
size_t mliis_colreduce_workspace_floats(long long rows_per_seg, int C, int nseg, int nv);
int mliis_bn_stats(const float* x, int ldx, long long rows, int C, int pre_swish, float eps, float momentum, int unbiased_moving_var,
                   float* mean, float* rstd, float* moving_mean, float* moving_var, float* ws, size_t ws_floats, hipStream_t stream);
int mliis_bn_apply(const float* x, int ldx, float* y, int ldy, long long rows, int C, int rows_per_img, const float* mean,
                   const float* rstd, const float* gamma, const float* beta, int pre_swish, int post_swish, const float* img_scale,
                   const float* res, int ldr, hipStream_t stream);
/*      training fast path (3 launches -> 1): stage-1 statistics from the producer's epilogue or mliis_bn_stats_partial
 *      (part [nblk][2][C], mliis_colreduce_workspace_floats(rows, C, 1, 2) floats), then fold + moving-average update + apply. */
int mliis_bn_stats_partial(const float* x, int ldx, long long rows, int C, int pre_swish, float* part, size_t part_floats, int* nblk_out,
                           hipStream_t stream);
/*      pool_part (nullable, training): the pass also leaves per-image partial sums of its OUTPUT, [rows/rows_per_img][*pool_chunks][C]
 *      (needs ceil(rows_per_img/128) * images * C floats) -- the squeeze-excite pooling (efficientnet_model.py:247) without a pass of
 *      its own; mliis_se_mlp_fwd folds the chunks. */
int mliis_bn_apply_fused(const float* x, int ldx, float* y, int ldy, long long rows, int C, int rows_per_img, const float* part, int nblk,
                         float eps, float momentum, int unbiased_moving_var, float* mean, float* rstd, float* moving_mean,
                         float* moving_var, const float* gamma, const float* beta, int pre_swish, int post_swish, const float* img_scale,
                         const float* res, int ldr, float* pool_part, size_t pool_floats, int* pool_chunks, int act_dtype, hipStream_t stream);
/*      workspace: mliis_colreduce_workspace_floats(rows, C, 1, 2) floats.  dskip (nullable): the same pass also writes the
 *      identity-skip gradient dskip[r,c] (+)= dy[r,c] (MBConv residual, efficientnet_model.py:286-288), so it needs no launch of
 *      its own.  dxsum_part (nullable): per-row-chunk column sums of dx, [chunks][C] with chunks * C = mliis_bn_bwd_dxsum_floats(rows, C) --
 *      slabs for mliis_fold_batched: the bias gradient of a conv -> swish -> BN stack (efficientlab.py:185-190) without a pass of
 *      its own. */
/*      Two batch norms of the same shape, flags and leading dimensions in ONE launch each -- the 1x1 and the 3x3-dilated branch of an
 *      RSD module are independent (efficientlab.py:185-197): forward = fold + apply of both (arguments as mliis_bn_apply_fused, per
 *      problem; no per-image scale, residual or pooling); backward = one reduce launch + one apply launch for both (plain batch norms:
 *      no per-image vectors, no skip output; dxsum0 / dxsum1 as dxsum_part above, dxsum_floats each; ws = 2 x
 *      mliis_colreduce_workspace_floats(rows, C, 1, 2) floats). */
int mliis_bn_apply_fused_pair(const float* x0, float* y0, const float* part0, int nblk0, float* mean0, float* rstd0, float* moving_mean0,
                              float* moving_var0, const float* gamma0, const float* beta0, const float* x1, float* y1, const float* part1,
                              int nblk1, float* mean1, float* rstd1, float* moving_mean1, float* moving_var1, const float* gamma1,
                              const float* beta1, int ldx, int ldy, long long rows, int C, float eps, float momentum,
                              int unbiased_moving_var, int pre_swish, int post_swish, hipStream_t stream);
int mliis_bn_bwd_pair(const float* x0, const float* dy0, float* dx0, const float* mean0, const float* rstd0, const float* gamma0,
                      const float* beta0, float* dgamma0, float* dbeta0, float* dxsum0, const float* x1, const float* dy1, float* dx1,
                      const float* mean1, const float* rstd1, const float* gamma1, const float* beta1, float* dgamma1, float* dbeta1,
                      float* dxsum1, int ldx, int lddy, int lddx, long long rows, int C, int pre_swish, int post_swish, size_t dxsum_floats,
                      float* ws, size_t ws_floats, hipStream_t stream);
size_t mliis_bn_bwd_dxsum_floats(long long rows, int C);
int mliis_bn_bwd(const float* x, int ldx, const float* dy, int lddy, float* dx, int lddx, long long rows, int C, int rows_per_img,
                 const float* mean, const float* rstd, const float* gamma, const float* beta, int pre_swish, int post_swish,
                 const float* img_scale, const float* chan_scale, const float* chan_add, float* dgamma, float* dbeta, float* dskip,
                 int lddskip, int dskip_accumulate, float* dxsum_part, size_t dxsum_floats, float* ws, size_t ws_floats,
                 const float* stage1_part, int stage1_nblk, int act_dtype, hipStream_t stream);
/*      stage1_part (nullable, [stage1_nblk][2][C]): the reduce pass's partial sums {sum g, sum g*xhat} already produced elsewhere with the
 *      SAME g (mliis_dwconv_bwd_data_bn, mliis_dwconv_bn_bwd, mliis_conv2d_bwd_data_bn, mliis_se_mlp_bwd_bn); mliis_bn_bwd then runs its apply pass only */

/* ---- per-image column sums: out[seg,c] (+)= scale * sum_rows a[row,c] * b[row,c]  (b nullable).  Serves tf.reduce_mean over
 *      H,W of squeeze-excite (efficientnet_model.py:247) and of the RSD pooled branch (efficientlab.py:192-197), their
 *      gradients, and conv bias gradients.  workspace: mliis_colreduce_workspace_floats(rows_per_seg, C, nseg, 1). */
int mliis_colsum(const float* a, int lda, const float* b, int ldb, long long rows_per_seg, int nseg, int C, float scale, float* out,
                 int accumulate, float* ws, size_t ws_floats, int ab_dtype, hipStream_t stream);

/* ---- squeeze-excite gate (efficientnet_model.py:238-251): hpre = W1.s + b1; gate = sigmoid(W2.swish(hpre) + b2).
 *      w1 [C,R], w2 [R,C] (TF HWIO 1x1).  R <= 128. */
/*      s_part [N][chunks][C]: pooled sums in `chunks` partials per image (mliis_bn_apply_fused's pool_part, or chunks = 1 for a
 *      finished vector); s = scale * sum_chunks is what the MLP sees and, if s_out is given, what is kept for the backward pass. */
int mliis_se_mlp_fwd(const float* s_part, int chunks, float scale, float* s_out, const float* w1, const float* b1, const float* w2,
                     const float* b2, float* hpre, float* gate, int N, int C, int R, hipStream_t stream);
/*      dw1 .. db2 all NULL: the weight gradients are left to mliis_se_wgrad_batched (one launch for every block of a backward pass;
 *      desc = device int64 [ndesc][12] {s, hpre, dpre1, dpre2, dw1, db1, dw2, db2 as device addresses, N, C, R, tile_begin}, a tile =
 *      256 of the 2*C*R + C + R gradient elements of a block, tile_begin = running sum of ceil(elements / 256)). */
/*      w1t (nullable): w1 transposed to [R,C] (the shadow copy mliis_transpose_weights maintains): the last phase needs a column of w1
 *      per channel, coalesced only from the transpose.  Same results either way. */
int mliis_se_mlp_bwd(const float* dgate, int dgate_row_groups, const float* gate, const float* s, const float* hpre, const float* w1,
                     const float* w1t, const float* w2, float* dpre1, float* dpre2, float* chan_add, float* dw1, float* db1, float* dw2, float* db2, int N,
                     int C, int R, int HW, hipStream_t stream);
/*      The squeeze-excite backward and the depthwise batch norm's backward of an MBConv block (efficientnet_model.py:238-251,271) share
 *      ONE pass over (da2, z1): mliis_se_bn_bwd_sums leaves, per image and row chunk, part [N][*nblk][5][C] = {sum da2*a1, sum da2*s',
 *      sum da2*s'*xhat, sum s', sum s'*xhat} (a1 = swish(u), s' = swish'(u), u = gamma*xhat + beta); mliis_se_mlp_bwd_bn folds value 0
 *      into the gate's gradient, runs the MLP backward (dpre1, dpre2, chan_add as mliis_se_mlp_bwd) and, since the batch norm's upstream
 *      gradient is g = (da2*gate[n] + chan_add[n]) * s', emits its stage-1 sums per image, stage1 [N][2][C], for
 *      mliis_bn_bwd(chan_scale = gate, chan_add, stage1_part = stage1, stage1_nblk = N).  Replaces mliis_colsum(da2, a1) and the
 *      batch norm's own reduce pass. */
size_t mliis_se_bn_bwd_sums_floats(int N, int rows_per_img, int C);
int mliis_se_bn_bwd_sums(const float* x, int ldx, const float* dy, int lddy, int N, int rows_per_img, int C, const float* mean,
                         const float* rstd, const float* gamma, const float* beta, float* part, size_t part_floats, int* nblk,
                         int act_dtype, hipStream_t stream);
int mliis_se_mlp_bwd_bn(const float* sums, int sums_nblk, const float* gate, const float* hpre, const float* w1, const float* w1t, const float* w2,
                        float* dpre1, float* dpre2, float* chan_add, float* stage1, int N, int C, int R, int HW, hipStream_t stream);
int mliis_se_wgrad_batched(const long long* desc, int ndesc, long long total_tiles, hipStream_t stream);
/*      y0[m, c] (+)= x[m, c] + A[n(m), c] for c < c0 and y1[m, c - c0] (+)= the same for c >= c0: the gradient of a channel concat
 *      routed to its two inputs in one pass (tail of the RSD module's backward, models/efficientlab.py:206-208,226-228) */
int mliis_chan_split(const float* x, int ldx, const float* A, float* y0, int ld0, int c0, int accumulate0, float* y1, int ld1, int accumulate1,
                     long long rows, int C, int rows_per_img, hipStream_t stream);
/*      y[m,c] (+)= x[m,c] * S[n(m),c] + A[n(m),c]  (x, S, A optional): gate apply, tf.tile of pooled vectors, pooled-gradient
 *      broadcast, strided channel-slice copy (tf.concat, efficientlab.py:208,222). */
int mliis_chan_affine(const float* x, int ldx, const float* S, const float* A, float* y, int ldy, long long rows, int C,
                      int rows_per_img, int accumulate, hipStream_t stream);

/* ---- ASPP activation sites (models/efficientlab.py:258-286, --spatial_pyramid_pooling): y = swish(z) * mask (conv -> swish ->
 *      tf.layers.dropout; mask = 0 or 1/keep per element, NULL = inference) or, pre_mask != 0, y = swish(z * mask) (the pooled branch
 *      drops BEFORE the swish); backward dz = dy * mask * swish'(z)  |  dy * swish'(z * mask) * mask.  Row-strided operands. */
int mliis_swish_mask_fwd(const float* z, int ldz, const float* mask, int ldm, float* y, int ldy, long long rows, int C, int pre_mask,
                         hipStream_t stream);
int mliis_swish_mask_bwd(const float* dy, int lddy, const float* z, int ldz, const float* mask, int ldm, float* dz, int lddz, long long rows,
                         int C, int pre_mask, hipStream_t stream);

/* ---- tf.image.resize_images(BILINEAR, align_corners=True) (efficientlab.py:171-172,205-206) and its transpose.
 *      C % 2 == 0 (the 2-channel logits map uses 8-byte vectors). */
int mliis_resize_bilinear_fwd(const float* x, int ldx, float* y, int ldy, int N, int Hi, int Wi, int Ho, int Wo, int C,
                              hipStream_t stream);
int mliis_resize_bilinear_bwd(const float* dy, int lddy, float* dx, int lddx, int N, int Hi, int Wi, int Ho, int Wo, int C,
                              int accumulate, hipStream_t stream);

/* ---- final layer: [dropout mask *] 1x1 conv C -> 2 + bias (efficientlab.py:161-167).  w [C,2], y/dy [rows,2] dense.
 *      mask (nullable) has the layout of x and holds 0 or 1/(1-rate). */
int mliis_final_conv_fwd(const float* x, int ldx, const float* mask, const float* w, const float* b, float* y, long long rows, int C,
                         hipStream_t stream);
int mliis_final_conv_bwd_data(const float* dy, const float* w, const float* mask, float* dx, int lddx, long long rows, int C,
                              hipStream_t stream);
/*      ... with the loss fold of a mliis_head_ce_fused call that was given out == NULL riding in workgroup 0 (fin_ws = that call's
 *      workspace, untouched since; N .. extra_loss as given there): one launch less between the head and the backward pass. */
int mliis_final_conv_bwd_data_fin(const float* dy, const float* w, const float* mask, float* dx, int lddx, long long rows, int C, float* fin_ws,
                                  int N, int Hd, int Wd, int H, int W, float extra_loss, float* out, hipStream_t stream);
/*      workspace: mliis_colreduce_workspace_floats(rows, C, 1, 2) */
int mliis_final_conv_bwd_filter(const float* x, int ldx, const float* mask, const float* dy, long long rows, int C, float* dw, float* db,
                                float* ws, size_t ws_floats, hipStream_t stream);

/* ---- loss: mean-over-pixels softmax CE with label smoothing [- ln(2 IoU / (IoU + 1))] (efficientlab.py:294-303,319-396),
 *      gradient w.r.t. logits, and predictions = (softmax > 0.5) (efficientlab.py:174-176,291-292).
 *      out (device float[3]) = {loss + extra_loss, ce, iou}. */
size_t mliis_softmax_ce_workspace_floats(int N, int H, int W);
int mliis_softmax_ce(const float* logits, const float* labels, const int* img_idx, int N, int H, int W, float label_smoothing, int dice,
                     float extra_loss, float* dlogits, float* pred, float* out, float* ws, size_t ws_floats, hipStream_t stream);
/*      mliis_head_ce_fused: the tail of a training step WITHOUT the dice term in two launches instead of five -- logits = bilinear
 *      resize (align_corners, efficientlab.py:166-173) of small [N,Hd,Wd,2] to [H,W]; per-pixel softmax cross-entropy with label
 *      smoothing (efficientlab.py:294-303) against labels [S,H,W,2] (through img_idx, nullable); dsmall = the gradient w.r.t. small --
 *      what mliis_resize_bilinear_fwd -> mliis_softmax_ce -> mliis_resize_bilinear_bwd compute, without writing the full-resolution
 *      logits and their gradient; out[0..2] = {loss + extra_loss, ce, iou}.  out == NULL: the second launch (the fold of the per-workgroup
 *      loss partials in ws) is left to mliis_final_conv_bwd_data_fin -- ws must stay untouched until then. */
int mliis_head_ce_fused_supported(int Hd, int Wd, int H, int W);   /* 1: up-sampling factor within the kernel's tile footprint (<= ~4.5) */
size_t mliis_head_ce_fused_workspace_floats(int N, int Hd, int Wd);
int mliis_head_ce_fused(const float* small, const float* labels, const int* img_idx, int N, int Hd, int Wd, int H, int W,
                        float label_smoothing, float extra_loss, float* dsmall, float* out, float* ws, size_t ws_floats,
                        hipStream_t stream);

/*      DARC1 regulariser (models/regularizers.py:20-22): weight * max over positions of sum_n |logits[n, pos]| added to out[0]
 *      (nullable) and its gradient weight * sign(logits[n, argmax]) added to dlogits (nullable); logits [N, per_img]; ws >= 2048 floats. */
int mliis_darc1(const float* logits, int N, long long per_img, float weight, float* dlogits, float* out, float* ws, size_t ws_floats,
                hipStream_t stream);

/* ---- optimizer + arena algebra.  tf.train.GradientDescentOptimizer / AdamOptimizer(beta1=0) apply
 *      (efficientlab.py:16,301,315-317; meta_learners/args.py:151-154) with the gradients of the weight regularisers -- l2 * w
 *      (models/regularizers.py:4-10) and l1 * sign(w) (:13-19), non batch-norm variables only -- folded in through a per-quad
 *      byte mask; lr_dev (nullable device float) overrides lr so captured graphs can vary it.
 *      axpby / lincomb implement meta_learners/variables.py:9-55 on the flat arena. */
int mliis_sgd_fused(float* w, const float* g, const uint8_t* l2_quad_mask, long long n, float lr, const float* lr_dev, float l2, float l1,
                    hipStream_t stream);
/*      Adam: step_dev (device float) = number of steps applied so far.  step_ticket == NULL: the caller advanced it before the call
 *      (this launch is step *step_dev).  step_ticket != NULL (device uint32, zero): this launch is step *step_dev + 1 and advances
 *      the count itself (last workgroup to finish) -- the form a captured HIP graph replays. */
int mliis_adam_b1zero_fused(float* w, const float* g, float* v, const uint8_t* l2_quad_mask, long long n, float lr, const float* lr_dev,
                            float l2, float l1, float beta2, float eps, float* step_dev, unsigned* step_ticket, hipStream_t stream);
int mliis_axpby(float a, const float* x, float b, float* y, long long n, hipStream_t stream);
int mliis_lincomb(float a, const float* x, float b, const float* y, float* out, long long n, hipStream_t stream);
/*      dst[0..n) = src[0..n) (32-bit words, n <= 1024) by ONE small kernel: the upload of a step's batch indices out of pinned host
 *      memory (src may be a device-visible HOST pointer).  A hipMemcpyAsync in its place runs on a copy engine: 4 us + a ~5 us engine
 *      switch between two steps of the compute queue (profiles/r06_notes.md). */
int mliis_copy_words(const void* src, void* dst, int n, hipStream_t stream);

/* ---- weight-gradient producers (conv2d / dwconv / stem *_bwd_filter) called with dw == NULL leave their per-split slabs in
 *      `ws`; one mliis_fold_batched launch then folds all of them into the gradient arena.  desc: device int64 [ndesc][8] =
 *      {part_off, out_off, total, seg_len, seg_stride, seg_off, nblk, tile_begin} (offsets in floats; one tile =
 *      mliis_fold_tile_outputs() consecutive outputs of a descriptor, tile_begin = running sum of ceil(total / tile)). */
int mliis_fold_tile_outputs(void);
/*      se_desc (nullable, with n_se rows and se_tiles tiles, else NULL, 0, 0): the table of mliis_se_wgrad_batched -- the squeeze-excite
 *      weight gradients of the pass ride in the same launch as se_tiles more workgroups (nothing but the optimizer reads either). */
int mliis_fold_batched(const float* part_base, float* out_base, const long long* desc, int ndesc, long long total_tiles,
                       const long long* se_desc, int n_se, long long se_tiles, hipStream_t stream);

/* ---- HIP-graph capture of one inner step (replaces the per-op dispatch of session.run) */
int mliis_graph_begin_capture(hipStream_t stream);
int mliis_graph_end_capture(hipStream_t stream, void** graph_exec_out);
int mliis_graph_launch(void* graph_exec, hipStream_t stream);
int mliis_graph_destroy(void* graph_exec);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif /* MLIIS_HIP_H */
