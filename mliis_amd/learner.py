"""EfficientLab learner: the MI355X replacement for `session.run(model.minimize_op, feed_dict)`.

Mirrors the handles the reference's meta-learner uses on the model object (models/efficientlab.py:42-61,94-100,108,173-176,
298-317) as methods:

    reference handle                                   here
    -------------------------------------------------  ------------------------------------------------
    sess.run(minimize_op, {X, Y[, lr_ph]})             Learner.inner_step(batch_idx, lr=None)
    sess.run(predictions, {X, is_training_ph: False})  Learner.predict(images, training=False)
    VariableState(trainable).export/import             Learner.export_trainable() / import_trainable()
    VariableState(global).export/import                Learner.export_all() / import_all()
    model.loss                                         Learner.last_loss (device scalar, read lazily)

One inner step = forward + backward + BN moving-average update + fused SGD/Adam apply, entirely on the device, in a fixed
sequence of C-ABI kernel launches over preallocated HBM buffers; after the first (eager) step of a given batch size the
sequence is captured into a HIP graph and replayed, so the host does no per-op work in the hot loop.

Graph semantics restated from: models/efficientlab.py:111-119,126-231,294-317; models/efficientnet/efficientnet_model.py:
253-290,396-441; models/efficientnet/utils.py:87-170.  Backward formulas: SURVEY.md Appendix B.
"""
from __future__ import annotations

import ctypes as C
import functools
import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import ops, spec
from ._lib import MliisError, lib
from .arena import Arena
from .passes import _Passes
from .plan import _Plan


class Learner(_Passes):
    def __init__(self, feature_extractor_name: str = "efficientnet-b0", image_size: int = 224, rsd: Optional[Sequence[int]] = (2, 4),
                 learning_rate: float = 1e-3, optimizer: str = "sgd", l2: bool = False, l1: bool = False, darc1: bool = False,
                 dice: bool = False, label_smoothing: float = 0.0, final_layer_dropout_rate: float = 0.0,
                 spatial_pyramid_pooling: bool = False, skip_decoding: bool = False, drop_connect: bool = True, seed: int = 0,
                 device="cuda:0", use_graph: bool = True, max_shots: int = 16, matmul_precision: str = "fp32",
                 small_fused: Optional[bool] = None, dw_march: Optional[bool] = None, fuse_bn2: Optional[bool] = None, fuse_head: Optional[bool] = None,
                 augment_batch_capacity: int = 0, rng_stream: int = 0):
        if optimizer not in ("sgd", "adam"):
            raise ValueError("optimizer must be 'sgd' or 'adam' (Adam with beta1=0, the reference default)")
        if not torch.cuda.is_available():
            raise MliisError("mliis_amd.Learner needs an MI355X (HIP device); there is no CPU path")
        lib.load()  # fail loudly if the HIP extension is missing
        # operand precision of the matrix cores in the dense convs: "fp32" (BASELINE configs 1-3) or "bf16" (operands rounded to bf16 on
        # the fly, fp32 accumulation; everything else stays fp32).  Passed with every dense-conv call (nothing process-wide).
        # "bf16-storage" (BASELINE configs[3]): bf16 operands AND the expanded tensors of the MBConv blocks (z0, z1, a1 and their
        # gradients) stored as bf16 in HBM during training steps; statistics, sums, accumulators, block outputs, the decoder and the
        # master weights stay fp32; inference (predict) runs on fp32 tensors
        self.act_dtype = torch.float32
        if matmul_precision == "bf16-storage":
            matmul_precision, self.act_dtype = "bf16", torch.bfloat16
        # "fp32" multiplies the long-K decoder convs as fp32 EQUIVALENT products on the bf16 matrix cores (every operand value split
        # exactly into three bf16 terms, six term products, fp32 accumulation: csrc/conv_x3.hip -- closer to float64 than the fp32
        # instruction's own accumulation, 1.5x faster on the dominant launch); every other conv takes the native fp32 instruction.
        # "fp32-native": the native fp32 matrix instruction (v_mfma_f32_16x16x4_f32) everywhere.  Same tolerances in every parity test.
        self.x3_on = matmul_precision == "fp32"
        if matmul_precision == "fp32-native":
            matmul_precision = "fp32"
        if matmul_precision not in ops.PRECISIONS:
            raise ValueError("matmul_precision must be one of {}, 'fp32-native' or 'bf16-storage', got {!r}".format(sorted(ops.PRECISIONS), matmul_precision))
        self.matmul_precision = matmul_precision
        self._conv_fwd = functools.partial(ops.conv2d_fwd, precision=matmul_precision)
        self._conv_bwd_data = functools.partial(ops.conv2d_bwd_data, precision=matmul_precision)
        self._conv_bwd_filter = functools.partial(ops.conv2d_bwd_filter, precision=matmul_precision)
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.arch = spec.derive(feature_extractor_name, image_size, list(rsd or []), final_layer_dropout_rate, spatial_pyramid_pooling,
                                skip_decoding=bool(skip_decoding))
        self.feature_extractor_name = feature_extractor_name
        self.final_layer_scope = "decode/final_layer_weights"
        self.lr, self.optimizer = float(learning_rate), optimizer
        self.l2, self.dice, self.label_smoothing = bool(l2), bool(dice), float(label_smoothing)
        self.l1, self.darc1 = bool(l1), bool(darc1)
        self.final_layer_dropout_rate = float(final_layer_dropout_rate or 0.0)
        self.drop_connect = drop_connect
        # small-map blocks through the one-launch-per-direction kernels of mbconv_small.hip, the other blocks through the row-marching
        # kernels of dwmarch.hip; a block whose shape neither family takes (their *_supported queries) runs op by op (dwconv.hip).
        # small_fused / dw_march = False force that op-by-op path (tests of the fallback).
        self.small_fused = True if small_fused is None else bool(small_fused)
        self.dw_march = True if dw_march is None else bool(dw_march)
        # fuse_bn2 = True: a block's project batch norm (+ drop-connect, + identity skip) is applied by the NEXT block's expand conv while
        # it loads its rows (ops.conv2d_fwd_bnin: ten launches fewer per step).  Built and parity-tested in round 5, measured SLOWER
        # than the stand-alone apply launches (-0.8 % on the step: every one of the 390-512 workgroups of an expand conv needs all K
        # channels' statistics and folds all the producer's partial blocks itself -- 24-44 MB of redundant L2 reads per launch against
        # 3 MB in the 39-52 workgroups of the apply kernel; profiles/r05_notes.md), so OFF by default; MLIIS_FUSE_BN2=1 turns it on
        self.fuse_bn2 = (os.environ.get("MLIIS_FUSE_BN2") == "1") if fuse_bn2 is None else bool(fuse_bn2)
        # fuse_head = False: resize -> softmax cross-entropy -> gradient -> resize^T as four launches (the form before round 5)
        self.fuse_head = (os.environ.get("MLIIS_NO_FUSE_HEAD") is None) if fuse_head is None else bool(fuse_head)
        self.use_graph = use_graph
        self.stream = torch.cuda.Stream(device=self.device)
        # batch indices go up through a ring of pinned slots: an upload from pageable memory makes the host wait for this stream
        self._idx_pin = torch.empty((16, 64), dtype=torch.int32).pin_memory()
        self._idx_ev = [None] * 16
        self._idx_n = 0
        self._idx_by_kernel = os.environ.get("MLIIS_IDX_MEMCPY", "0") != "1"
        self.stem_stats = os.environ.get("MLIIS_STEM_STATS", "1") != "0"   # the stem conv's launch also emits its batch norm's stage-1 sums
        self.defer_loss_fold = os.environ.get("MLIIS_NO_DEFER_LOSS_FOLD", "0") != "1"
        self.arena = Arena(self.arch, self.device)
        self.arena.init_weights(seed)
        self.variables_initialized = True
        self._capturing = False
        self.ws = ops.Workspace(self.device, 1 << 22, on_grow=self._on_workspace_grow)
        # K-contiguous ("HWOI") shadow of the dense-conv weights, refreshed by ONE batched transpose at the top of every
        # forward, so the forward GEMMs read their B operand as 16-byte k-fragments like backward-data does
        A = self.arena
        self.theta_t = torch.zeros_like(A.theta)
        fe = self.arch.name
        desc = []
        for p in A.trainable:
            if p.kind == "conv" and p.executed and "/se/" not in p.name and p.name not in (f"{fe}/stem/conv2d/kernel", "decode/final_layer_weights/kernel"):
                desc.append([A.t_off[p.name], p.shape[0] * p.shape[1], p.shape[2], p.shape[3]])
        n_dense = len(desc)
        for p in A.trainable:   # the squeeze-excite reduce weights [C, R] too: the MLP's backward reads a column of them per channel
            if p.kind == "conv" and p.executed and p.name.endswith("/se/conv2d/kernel"):
                desc.append([A.t_off[p.name], 1, p.shape[2], p.shape[3]])
        self.wt_desc = torch.tensor(desc, dtype=torch.int32, device=self.device)
        self.wt_tiles = ops.transpose_tiles(desc)
        # fp8 mode: max |w| of every dense-conv weight, refreshed by the same launch that refreshes the shadow copies
        self.w_amax = torch.zeros(len(desc), dtype=torch.float32, device=self.device) if matmul_precision == "fp8" else None   # ([:n_dense] used)
        self._amax_of = {}
        if self.w_amax is not None:
            names = [p.name for p in A.trainable if p.kind == "conv" and p.executed and "/se/" not in p.name and
                     p.name not in (f"{fe}/stem/conv2d/kernel", "decode/final_layer_weights/kernel")]
            self._amax_of = {n: self.w_amax[i:i + 1] for i, n in enumerate(names)}
        self.wt = {p.name: self.theta_t[A.t_off[p.name]:A.t_off[p.name] + p.size] for p in A.trainable}
        self.lr_dev = torch.tensor([self.lr], dtype=torch.float32, device=self.device)
        self._lr_dev_val = float(self.lr)
        self.adam_v = torch.zeros_like(self.arena.theta) if optimizer == "adam" else None
        self.adam_t = torch.zeros(1, dtype=torch.float32, device=self.device)          # Adam steps applied so far
        self.adam_ticket = torch.zeros(1, dtype=torch.int32, device=self.device)   # the optimizer launch advances adam_t itself
        self.adam_epoch = 0     # bumped whenever the Adam slots are REPLACED from outside (restore / import_all): lanes follow (reptile.py)
        self.plans: Dict[object, _Plan] = {}   # key: batch size (training plan) or (batch size, "infer") in bf16-storage mode
        self.max_shots = max_shots
        H = image_size
        # resident shots [0, max_shots) + (on-device augmentation) the slots of one augmented mini-batch behind them: the step's kernels
        # address images through an index vector, so an augmented batch is just another set of indices -- same HIP graph
        self.aug_capacity = int(augment_batch_capacity)
        self.shots_x = torch.zeros(max_shots + self.aug_capacity, H, H, 3, dtype=torch.float32, device=self.device)
        self.shots_y = torch.zeros(max_shots + self.aug_capacity, H, H, 2, dtype=torch.float32, device=self.device)
        if self.aug_capacity:
            self._aug_tmp_x = torch.zeros(self.aug_capacity, H, H, 3, dtype=torch.float32, device=self.device)
            self._aug_tmp_y = torch.zeros(self.aug_capacity, H, H, 2, dtype=torch.float32, device=self.device)
            self._aug_ops_dev = torch.zeros(8 * self.aug_capacity * 48, dtype=torch.uint8, device=self.device)
            self._aug_ops_pin = torch.zeros((4, 8 * self.aug_capacity * 48), dtype=torch.uint8).pin_memory()
            self._aug_ev = [None] * 4
            self._aug_n = 0
        self.n_shots = 0
        # device RNG of the stochastic ops (drop-connect, dropout): Philox state advanced by the mask kernel itself (csrc/rng.hip)
        # `seed` is the same on every rank (the weights must start identical); `rng_stream` -- the rank -- goes into the high word of
        # the Philox KEY only, so ranks that step in lock-step on different tasks do not draw identical drop-connect / dropout masks
        self.rng_state = ops.rng_state((int(seed) & 0xFFFFFFFF) ^ ((int(rng_stream) & 0xFFFFFFFF) << 32) if rng_stream else seed, self.device)
        ex_ = [b for b in self.arch.blocks if b.executed and b.skip]
        self._dc_keeps = torch.tensor([1.0 - b.drop_rate for b in ex_] or [1.0], dtype=torch.float32, device=self.device)
        self.drop_keep_dev = torch.tensor([1.0 - self.final_layer_dropout_rate], dtype=torch.float32, device=self.device)
        self._drop_keep_val = 1.0 - self.final_layer_dropout_rate
        self._pname()
        # fp32x3: weight images of the decoder's 3x3 convs (the dilated branch over the whole concat, the fuse conv over its convolved
        # channels), both directions, re-split once per step beside the shadow transpose
        self.x3 = None
        if self.x3_on:
            self.x3 = ops.X3Images(self.device)
            shapes = {p.name: p.shape for p in A.trainable}
            for m, nm in zip(self.arch.rsd, self.n_rsd):
                (k0, _, _), (k1, _, _), (kf, _, _) = nm
                for name, cin in ((k1, None), (kf, 2 * m.c_out)):
                    kk, _, cin_total, cout = shapes[name]
                    win = cin_total if cin is None else cin
                    if ops.X3Images.eligible(win, cout, kk) and ops.X3Images.eligible(cout, win, kk):
                        self.x3.add(name, "fwd", A.t_off[name], kk, cin_total, cout, 0, win)
                        self.x3.add(name, "bwd", A.t_off[name], kk, cin_total, cout, 0, win)
            self.x3.finish()
        torch.cuda.synchronize(self.device)   # arena was initialised on the default stream; steps run on self.stream

    # ------------------------------------------------------------------------------------------- names
    def _pname(self):
        fe = self.arch.name
        self.n_stem = (f"{fe}/stem/conv2d/kernel", f"{fe}/stem/tpu_batch_normalization")
        self.n_blocks = []
        for b in self.arch.blocks:
            if not b.executed:
                continue
            s = f"{fe}/blocks_{b.idx}"
            bns = [f"{s}/tpu_batch_normalization", f"{s}/tpu_batch_normalization_1", f"{s}/tpu_batch_normalization_2"]
            cvs = [f"{s}/conv2d/kernel", f"{s}/conv2d_1/kernel"]
            d = {}
            if b.expand != 1:
                d["w_exp"], d["bn0"] = cvs.pop(0), bns.pop(0)
            d["w_dw"], d["bn1"] = f"{s}/depthwise_conv2d/depthwise_kernel", bns.pop(0)
            d["se"] = (f"{s}/se/conv2d/kernel", f"{s}/se/conv2d/bias", f"{s}/se/conv2d_1/kernel", f"{s}/se/conv2d_1/bias")
            d["w_proj"], d["bn2"] = cvs.pop(0), bns.pop(0)
            self.n_blocks.append(d)
        self.n_rsd, self.n_rsd_up = [], []
        for m in self.arch.rsd:
            s = f"decode/decode_skip_connections_{m.scope_index}"
            sfx = ["", "_1", "_2", "_3"]
            trip = lambda x: (f"{s}/conv2d{x}/kernel", f"{s}/conv2d{x}/bias", f"{s}/batch_normalization{x}")   # noqa: E731
            self.n_rsd_up.append(trip(sfx.pop(0)) if m.upsample_conv else None)   # (created first in the scope, efficientlab.py:213-215)
            self.n_rsd.append([trip(sfx.pop(0)) for _ in range(3)])
        s = "decode/decode_skip_connections"   # --skip_decoding: (1x1 kernel, its BN), then per sep_conv (dw kernel, BN, 1x1 kernel, BN)
        self.n_skipdec = ((f"{s}/conv2d/kernel", f"{s}/batch_normalization"),
                          [(f"{s}/depthwise_conv2d{'' if j == 0 else '_%d' % j}/depthwise_kernel", f"{s}/batch_normalization_{2 * j + 1}",
                            f"{s}/conv2d_{j + 1}/kernel", f"{s}/batch_normalization_{2 * j + 2}") for j in range(2)])
        self.n_final = ("decode/final_layer_weights/kernel", "decode/final_layer_weights/bias")
        s = "decode/spatial_pyramid_pooling"
        self.n_aspp = [(f"{sc}/conv2d/kernel", f"{sc}/conv2d/bias") for sc in (f"{s}/branch_0", f"{s}/branch_1", f"{s}/branch_2", s)]

    # ------------------------------------------------------------------------------------------- variable state
    @property
    def n_trainable(self) -> int:
        return self.arena.n_trainable

    def _out(self, t):
        """Hand a tensor produced on the learner stream to the caller's current stream."""
        cur = torch.cuda.current_stream(self.device)
        if cur != self.stream:
            cur.wait_stream(self.stream)
            if torch.is_tensor(t):
                t.record_stream(cur)
        return t

    def _in(self):
        """Order the learner stream after work the caller queued on its current stream."""
        cur = torch.cuda.current_stream(self.device)
        if cur != self.stream:
            self.stream.wait_stream(cur)

    def export_trainable(self) -> torch.Tensor:
        """Device clone of the (padded) trainable arena -- VariableState.export_variables (variables.py:70-74)."""
        with torch.cuda.stream(self.stream):
            t = self.arena.theta.clone()
        return self._out(t)

    def import_trainable(self, flat: torch.Tensor):
        self._in()
        with torch.cuda.stream(self.stream):
            self.arena.theta.copy_(flat)

    def export_bn(self) -> torch.Tensor:
        with torch.cuda.stream(self.stream):
            t = self.arena.bn_moving.clone()
        return self._out(t)

    def import_bn(self, flat: torch.Tensor):
        self._in()
        with torch.cuda.stream(self.stream):
            self.arena.bn_moving.copy_(flat)

    def export_all(self):
        """All global variables (trainable + BN moving [+ Adam slots]) -- Gecko._full_state (reptile.py:35-36,258)."""
        with torch.cuda.stream(self.stream):
            st = {"theta": self.arena.theta.clone(), "bn": self.arena.bn_moving.clone()}
            if self.adam_v is not None:
                st["adam_v"], st["adam_t"] = self.adam_v.clone(), self.adam_t.clone()
        for v in st.values():
            self._out(v)
        return st

    def import_all(self, st):
        """Restore every variable from an export_all() state of THIS run (the evaluation path's save / restore around a fine-tune,
        reptile.py: _evaluate): the lanes' second-moment history stays.  State from outside the run (a checkpoint) arrives through
        load_named, which advances adam_epoch so that concurrent lanes take over its Adam slots too."""
        self._in()
        with torch.cuda.stream(self.stream):
            self.arena.theta.copy_(st["theta"])
            self.arena.bn_moving.copy_(st["bn"])
            if self.adam_v is not None and "adam_v" in st:
                self.adam_v.copy_(st["adam_v"])
                self.adam_t.copy_(st["adam_t"])

    def import_adam(self, adam_v: torch.Tensor, adam_t: torch.Tensor):
        """Replace only the Adam slots (second moments + step count): how a lane takes over the main learner's restored optimizer state."""
        if self.adam_v is None:
            return
        self._in()
        with torch.cuda.stream(self.stream):
            self.adam_v.copy_(adam_v)
            self.adam_t.copy_(adam_t)
        self.adam_epoch += 1

    def axpby(self, a: float, x: torch.Tensor, b: float, y: torch.Tensor):
        """y <- a*x + b*y on flat arena-shaped buffers (meta_learners/variables.py:9-45 on the device)."""
        self._in()
        with torch.cuda.stream(self.stream):
            ops.axpby(a, x, b, y)

    def comm_context(self):
        """Context in which collectives on arena buffers must be issued (orders them after this learner's stream)."""
        return torch.cuda.stream(self.stream)

    def synchronize(self):
        self.stream.synchronize()

    def loss_value(self) -> float:
        self.stream.synchronize()
        return float(self.last_loss[0].item())

    ADAM_BETA2 = 0.999

    def named_numpy(self) -> Dict[str, np.ndarray]:
        """All global variables under their TF names, read after everything queued on the learner's stream has finished: what
        `tf.train.Saver()` stores (train.py:54,131) -- trainables, BN moving statistics and, for the Adam inner optimizer
        (--sgd absent), its persistent slots: `<var>/Adam_1` (second moment), `beta1_power` / `beta2_power` in TF's convention
        (beta^(t+1) after t steps; beta1 = 0).  The first-moment slot `<var>/Adam` equals the last gradient when beta1 = 0 and is
        not stored."""
        self.stream.synchronize()
        torch.cuda.synchronize(self.device)
        out = self.arena.named_numpy()
        if self.adam_v is not None:
            A = self.arena
            v = self.adam_v.detach().cpu().numpy()
            for p in A.trainable:
                o = A.t_off[p.name]
                out[p.name + "/Adam_1"] = v[o:o + p.size].reshape(p.shape).copy()
            from .checkpoint import adam_step_entries
            out.update(adam_step_entries(int(round(float(self.adam_t.item()))), self.ADAM_BETA2))
        return out

    def load_named(self, values, **kw) -> int:
        """Restore by name with the scope filters of EfficientLab.restore_model (prefixes / exclude_prefix / strict, forwarded to the
        arena).  The Adam slots follow the SAME filters (a final layer that is not restored keeps fresh slots too).  The step count
        (`beta2_power` / `adam_step`, top-level names) is restored unless a `prefixes` whitelist excludes it: restore_model's
        `filter_out_scope=final_layer_scope` keeps every global variable that does not start with that scope, the beta powers among
        them (efficientlab.py restore_model) -- so a --pretrained --do_not_restore_final_layer_weights run fine-tunes with a
        warm bias correction, as the reference does."""
        from .checkpoint import adam_step_from
        self.stream.synchronize()
        n = self.arena.load_named(values, **kw)
        if self.adam_v is not None:
            A = self.arena
            prefixes, exclude = kw.get("prefixes"), kw.get("exclude_prefix")
            for p in A.trainable:
                key = p.name + "/Adam_1"
                if key not in values:
                    continue
                if prefixes is not None and not any(p.name.startswith(x) for x in prefixes):
                    continue
                if exclude is not None and p.name.startswith(exclude):
                    continue
                o = A.t_off[p.name]
                self.adam_v[o:o + p.size].copy_(torch.from_numpy(np.asarray(values[key], dtype=np.float32).reshape(-1)))
            t = adam_step_from(values, self.ADAM_BETA2)
            if t is not None and prefixes is None:
                self.adam_t.fill_(float(t))
            self.adam_epoch += 1
        torch.cuda.synchronize(self.device)
        return n

    def disable_split_products(self):
        """Back to the native fp32 matrix instruction for the decoder convs (matmul_precision "fp32-native") on a live learner.
        Captured graphs hold the x3 launches: dropped, the next step of each plan is eager again."""
        if self.x3 is None:
            return
        self.stream.synchronize()
        self.x3 = None
        self.x3_on = False
        for P in self.plans.values():
            if P.graphs:
                for gexec in P.graphs.values():
                    lib.call("mliis_graph_destroy", gexec)
                P.graphs = {}
                P.steps_run = 0
            P.wbatch_ready = False   # (the deferred filter-gradient batch is launched with the learner's precision: rebuilt)

    def _on_workspace_grow(self, floats: int):
        """The shared scratch buffer is about to be replaced by a larger one (a plan with more images than any before).  Captured
        HIP graphs hold the old address: finish the queued work and drop them (the next step of each plan captures again)."""
        if self._capturing:
            raise MliisError("workspace would grow to {} floats during HIP-graph capture (the eager first step of a plan sizes it)".format(floats))
        self.stream.synchronize()
        for P in self.plans.values():
            if P.graphs:
                for gexec in P.graphs.values():
                    lib.call("mliis_graph_destroy", gexec)
                P.graphs = {}
                P.steps_run = 0   # next step eager (re-sizes), the one after captures

    # ------------------------------------------------------------------------------------------- task data
    def load_task(self, images, labels):
        """Make a task's shots resident: images [S,H,W,3] f32 0..255, labels [S,H,W,2] (numpy or tensors)."""
        images = torch.as_tensor(images)
        labels = torch.as_tensor(labels)
        S = images.shape[0]
        if S > self.max_shots:
            raise ValueError("task has {} shots; Learner was built with max_shots={}".format(S, self.max_shots))
        H = self.arch.image_size
        if tuple(images.shape[1:]) != (H, H, 3) or tuple(labels.shape) != (S, H, H, 2):
            raise ValueError("expected images [S,{0},{0},3] and labels [S,{0},{0},2], got {1} / {2}".format(H, tuple(images.shape),
                                                                                                    tuple(labels.shape)))
        self._in()
        with torch.cuda.stream(self.stream):
            self.shots_x[:S].copy_(images.to(torch.float32), non_blocking=True)
            self.shots_y[:S].copy_(labels.to(torch.float32), non_blocking=True)
        self.n_shots = S
        self._aug_valid = 0

    def augment_batch(self, src_idx: Sequence[int], recipes) -> List[int]:
        """On-device augmentation of one mini-batch (csrc/augment.hip): sample b = shot src_idx[b] of the resident task through the
        planned steps recipes[b] (mliis_amd.augment.Augmenter(fields=False).plan; None = the original).  Returns the indices of the
        augmented samples -- slots behind the resident shots -- to hand to inner_step()."""
        from .augment import encode_device_ops
        B = len(src_idx)
        if not self.aug_capacity or B > self.aug_capacity:
            raise ValueError("Learner was built with augment_batch_capacity={}; batch has {} samples".format(self.aug_capacity, B))
        if max(src_idx) >= self.n_shots or min(src_idx) < 0:
            raise ValueError("augment_batch: shot index out of range of the resident task ({} shots)".format(self.n_shots))
        ops_np = encode_device_ops(recipes, src_idx)
        n = ops_np.shape[0]
        if n > 8:
            raise ValueError("at most 8 augmentation stages per sample")
        base = self.max_shots
        for i in range(n):    # stage i reads: the shots (i = 0) or the previous stage's output; the LAST stage writes the batch slots
            prev_in_shots = i > 0 and (n - 1 - (i - 1)) % 2 == 0
            ops_np[i]["src"] = np.asarray(src_idx, dtype=np.int32) if i == 0 else (np.arange(B, dtype=np.int32) + (base if prev_in_shots else 0))
        H = self.arch.image_size
        nbytes = ops_np.nbytes
        with torch.cuda.stream(self.stream):
            slot = self._aug_n % 4
            self._aug_n += 1
            if self._aug_ev[slot] is not None:
                self._aug_ev[slot].synchronize()
            else:
                self._aug_ev[slot] = torch.cuda.Event()
            pin = self._aug_ops_pin[slot, :nbytes]
            pin.copy_(torch.from_numpy(ops_np.reshape(-1).view(np.uint8)))
            dev = self._aug_ops_dev[:nbytes]
            dev.copy_(pin, non_blocking=True)
            self._aug_ev[slot].record(self.stream)
            for i in range(n):
                to_shots = (n - 1 - i) % 2 == 0
                from_shots = i == 0 or (n - 1 - (i - 1)) % 2 == 0
                xin, yin = (self.shots_x, self.shots_y) if from_shots else (self._aug_tmp_x, self._aug_tmp_y)
                xout, yout = (self.shots_x, self.shots_y) if to_shots else (self._aug_tmp_x, self._aug_tmp_y)
                if xin is xout:   # (stage 0 reading the shots and writing the batch slots of the same allocation: disjoint sample ranges)
                    xin_p, yin_p = xin.data_ptr(), yin.data_ptr()
                    xout_p, yout_p = xout[base:].data_ptr(), yout[base:].data_ptr()
                    ob = 0
                else:
                    xin_p, yin_p, xout_p, yout_p, ob = xin.data_ptr(), yin.data_ptr(), xout.data_ptr(), yout.data_ptr(), (base if to_shots else 0)
                lib.call("mliis_augment_stage", xin_p, yin_p, xout_p, yout_p, dev.data_ptr() + i * B * 48, B, H, H, ob, self.stream.cuda_stream)
        self._aug_valid = B
        return [base + b for b in range(B)]

    def _plan(self, N: int, infer: bool = False) -> _Plan:
        """The buffers of a batch size.  bf16-storage mode keeps a second, fp32 plan for inference (the op-by-op inference kernels read
        fp32 tensors); in every other mode training and inference share one."""
        key = (N, "infer") if (infer and self.act_dtype != torch.float32) else N
        if key not in self.plans:
            with torch.cuda.stream(self.stream):
                self.plans[key] = _Plan(self, N, torch.float32 if isinstance(key, tuple) else self.act_dtype)
        return self.plans[key]

    # ------------------------------------------------------------------------------------------- one optimisation step
    def _apply(self):
        A = self.arena
        l2 = spec.L2_WEIGHT if self.l2 else 0.0
        l1 = spec.L2_WEIGHT if self.l1 else 0.0     # (models/regularizers.py:13: the same 5e-4 default)
        if self.optimizer == "sgd":
            ops.sgd_fused(A.theta, A.grad, self.lr, A.l2_quad_mask, l2, self.lr_dev, l1=l1)
        else:
            # (the launch reads the step count from device memory, uses t + 1 and stores it: the same graph replays every step)
            ops.adam_b1zero_fused(A.theta, A.grad, self.adam_v, self.adam_t, self.lr, A.l2_quad_mask, l2, self.lr_dev, l1=l1,
                                  ticket=self.adam_ticket)

    def _train_sequence(self, P: _Plan, draw_masks: bool):
        # (the masks of the step are drawn by the launch that builds the weight shadows at the head of the forward pass: passes.py)
        self._rng_now = (self.rng_state, P.mask_plan) if (draw_masks and P.mask_plan is not None) else None
        # Without the dice term (and DARC1, which reads the full-resolution logits) the tail of the step -- resize to the image size,
        # softmax cross-entropy, its gradient, the resize's transpose -- is ONE launch on the decoder's map (ops.head_ce_fused)
        hd_, H_ = self.arch.h_dec, self.arch.image_size
        head_fused = bool(self.fuse_head and not self.dice and not self.darc1 and lib.size("mliis_head_ce_fused_supported", hd_, hd_, H_, H_))
        logits = self._forward(P, self.shots_x, P.idx, True, upsample=not head_fused)
        P.head_fin = None
        if head_fused:
            H = self.arch.image_size
            if self.defer_loss_fold:   # ONE launch: the fold of the loss partials rides in the final conv's backward-data launch (passes.py)
                if getattr(P, "head_ws", None) is None:
                    P.head_ws = ops.Workspace(self.device, lib.size("mliis_head_ce_fused_workspace_floats", P.N, hd_, hd_))
                _, _, buf = ops.head_ce_fused(P.small, self.shots_y, P.idx, (H, H), self.label_smoothing, P.dsmall, P.loss_out, ws=P.head_ws,
                                              finalize=False)
                P.head_fin = (buf, (H, H), 0.0, P.loss_out)
            else:
                ops.head_ce_fused(P.small, self.shots_y, P.idx, (H, H), self.label_smoothing, P.dsmall, P.loss_out, ws=self.ws)
        else:
            ops.softmax_ce(logits, self.shots_y, P.idx, self.label_smoothing, self.dice, 0.0, want_grad=True, want_pred=False,
                           dlogits=P.dlogits, out=P.loss_out, ws=self.ws)
            if self.darc1:
                ops.darc1(logits, spec.L2_WEIGHT, dlogits=P.dlogits, out=P.loss_out, ws=self.ws)
        self._backward(P, self.shots_x, P.idx, head_fused=head_fused)
        self._apply()

    def inner_step(self, batch_idx: Sequence[int], lr: Optional[float] = None, dc_scales: Optional[Dict[int, torch.Tensor]] = None,
                   dropout_mask: Optional[torch.Tensor] = None, weight_decay_rate: float = 1.0, drop_rate: Optional[float] = None,
                   aspp_masks: Optional[Sequence[torch.Tensor]] = None):
        """One `session.run(minimize_op)` on images `batch_idx` of the resident task.  Returns the device loss scalar.
        `drop_rate` overrides the final-layer dropout rate for this step (the reference feeds `final_layer_dropout_rate_ph`, a
        placeholder that only exists when the model was built with dropout, models/efficientlab.py:94-100)."""
        N = len(batch_idx)
        if N == 0:
            raise ValueError("empty mini-batch")
        lim = self.max_shots + self._aug_valid if getattr(self, "_aug_valid", 0) else self.n_shots
        if min(batch_idx) < 0 or max(batch_idx) >= lim or any(self.n_shots <= i < self.max_shots for i in batch_idx):
            raise ValueError("batch index out of range of the resident task ({} shots)".format(self.n_shots))
        P = self._plan(N)
        with torch.cuda.stream(self.stream):
            if N <= self._idx_pin.shape[1]:
                slot = self._idx_n % len(self._idx_ev)
                self._idx_n += 1
                if self._idx_ev[slot] is not None:
                    self._idx_ev[slot].synchronize()   # the upload that last used this slot (16 steps ago) has been consumed
                else:
                    self._idx_ev[slot] = torch.cuda.Event()
                src = self._idx_pin[slot, :N]
                src.copy_(torch.tensor(list(batch_idx), dtype=torch.int32))
                if self._idx_by_kernel:    # one small kernel reading the pinned slot: no copy engine between two steps
                    ops.copy_words(src, P.idx)
                else:
                    P.idx.copy_(src, non_blocking=True)
                self._idx_ev[slot].record(self.stream)
            else:
                P.idx.copy_(torch.tensor(list(batch_idx), dtype=torch.int32), non_blocking=True)
            lr_now = self.lr if lr is None else float(lr)
            if lr_now != self._lr_dev_val:   # the fused optimizer kernels read the rate from device memory (graph-replay safe)
                self.lr_dev.fill_(lr_now)
                self._lr_dev_val = lr_now
            if weight_decay_rate != 1.0:  # pre_step_op (variables.py:48-55)
                ops.axpby(0.0, None, float(weight_decay_rate), self.arena.theta)
            draw = self._fill_masks(P, dc_scales, dropout_mask, drop_rate, aspp_masks)
            if self.use_graph and P.steps_run >= 1:
                if draw not in P.graphs:
                    gexec = C.c_void_p()
                    lib.call("mliis_graph_begin_capture", self.stream.cuda_stream)
                    self._capturing = True
                    try:
                        self._train_sequence(P, draw)
                    finally:
                        self._capturing = False
                        lib.call("mliis_graph_end_capture", self.stream.cuda_stream, C.byref(gexec))
                    P.graphs[draw] = gexec
                lib.call("mliis_graph_launch", P.graphs[draw], self.stream.cuda_stream)
            else:
                self._train_sequence(P, draw)
            P.steps_run += 1
        self.last_loss = P.loss_out
        return P.loss_out

    def _fill_masks(self, P: _Plan, dc_scales, dropout_mask, drop_rate=None, aspp_masks=None) -> bool:
        """Masks handed in by the caller (parity tests) are copied into the plan's buffers; returns True when the step has to draw its
        masks itself -- on the device, inside the step (ops.rng_masks: no host RNG and no torch op on the path)."""
        if drop_rate is not None and P.drop_mask is None:
            raise ValueError("drop_rate given but the model was built without final-layer dropout (final_layer_dropout_rate = 0)")
        if drop_rate is not None and not 0.0 <= float(drop_rate) < 1.0:
            raise ValueError("drop_rate must be in [0, 1), got {}".format(drop_rate))
        injected = dc_scales is not None or dropout_mask is not None or aspp_masks is not None
        if P.drop_mask is not None and dropout_mask is None:   # keep probability of the final-layer dropout: a device scalar (graph-replay safe)
            keep = 1.0 - (self.final_layer_dropout_rate if drop_rate is None else float(drop_rate))
            if keep != self._drop_keep_val:
                self.drop_keep_dev.fill_(keep)
                self._drop_keep_val = keep
        if not injected:
            return P.mask_plan is not None
        # ---- injected masks: every stochastic site not given explicitly is drawn once here with the same device generator
        if P.mask_plan is not None:
            ops.rng_masks(self.rng_state, P.mask_plan)
        if P.aspp is not None and aspp_masks is not None:   # four tf.layers.dropout(rate=0.5) sites: scale 0 or 1/keep
            for i, mbuf in enumerate(P.aspp["masks"]):
                mbuf.copy_(torch.as_tensor(aspp_masks[i], dtype=torch.float32).reshape(mbuf.shape))
        ex = [b for b in self.arch.blocks if b.executed]
        if self.drop_connect and dc_scales is not None:
            for b, B in zip(ex, P.blocks):
                if b.skip and "dc" in B:
                    v = dc_scales.get(b.idx)
                    if v is None:
                        B["dc"].fill_(1.0)
                    else:
                        B["dc"].copy_(torch.as_tensor(v, dtype=torch.float32))
        if P.drop_mask is not None and dropout_mask is not None:
            P.drop_mask.copy_(torch.as_tensor(dropout_mask, dtype=torch.float32))
        return False

    # ------------------------------------------------------------------------------------------- inference
    def predict(self, images, training: bool = False, return_logits: bool = False):
        """predictions tensor of the reference: (softmax(logits) > 0.5) as float, [N,H,W,2]."""
        images = torch.as_tensor(images)
        N = images.shape[0]
        P = self._plan(N, infer=True)
        with torch.cuda.stream(self.stream):
            x = images.to(device=self.device, dtype=torch.float32).contiguous()
            logits = self._forward(P, x, None, training)
            dummy = torch.zeros_like(logits)
            ops.softmax_ce(logits, dummy, None, 0.0, False, 0.0, want_grad=False, want_pred=True, pred=P.pred, out=P.loss_out, ws=self.ws)
            out = P.pred.clone()
            lg = logits.clone() if return_logits else None
        self.stream.synchronize()
        return (out, lg) if return_logits else out

    def predict_resident(self, idx: Sequence[int], training: bool = False):
        """predict() on images `idx` of the task made resident by load_task()."""
        with torch.cuda.stream(self.stream):
            x = self.shots_x[torch.tensor(list(idx), dtype=torch.long, device=self.device)]
        return self.predict(x, training=training)

    def close(self):
        """Destroy the captured HIP graphs (the buffers themselves are torch tensors and go with the object)."""
        self.synchronize()
        for P in self.plans.values():
            for gexec in P.graphs.values():
                lib.call("mliis_graph_destroy", gexec)
            P.graphs = {}

    def gradients_packed(self) -> torch.Tensor:
        return self.arena.export_grad_packed()
