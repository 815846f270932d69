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
import math
import os
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import ops, spec
from ._lib import MliisError, lib
from .arena import Arena


class _Plan:
    """All activation / gradient buffers of one inner step for a fixed batch size N."""

    def __init__(self, L: "Learner", N: int, act_dtype=torch.float32):
        a, dev = L.arch, L.device
        self.N = N
        # storage type of the EXPANDED tensors of the MBConv blocks (z0, z1, a1, da2, da0 / dz0): fp32, or bf16 for the training plan of
        # `--precision bf16-storage` (BASELINE configs[3]).  Everything else -- block inputs / outputs, the decoder, statistics, sums,
        # parameters -- is fp32 in every mode.
        self.act_dtype = act_dtype

        def buf(*shape):
            return torch.empty(shape, dtype=torch.float32, device=dev)

        def xbuf(*shape):
            return torch.empty(shape, dtype=act_dtype, device=dev)

        def vec(c):
            return buf(c), buf(c)
        self.idx = torch.zeros(N, dtype=torch.int32, device=dev)
        H = a.image_size
        hs = a.h_stem
        self.z_stem, self.a_stem = buf(N, hs, hs, a.stem_out), None   # (a_stem: only when block 0 does not take the stem's BN + swish, below)
        self.st_stem = vec(a.stem_out)
        self.blocks = []
        nskip = sum(1 for b in a.blocks if b.executed and b.skip)
        self.dc_all = torch.ones(max(nskip, 1), N, dtype=torch.float32, device=dev)
        si = 0
        gmax = 0
        for b in a.blocks:
            if not b.executed:
                continue
            B = {}
            hi, ho, ce = b.h_in, b.h_out, b.cexp
            # small maps (14x14 at 224x224 inputs): the depthwise half of the block runs as ONE launch per direction (mbconv_small.hip)
            B["small"] = bool(L.small_fused and b.expand != 1 and ops.mbconv_dw_small_supported(N, hi, hi, ce, b.k, b.stride))
            # every other block: the row-marching kernels (batch norm + swish in front of the depthwise conv applied while its input is
            # staged; one-pass backward).  A shape neither family takes (>= 2 GiB tensors, other k / stride) runs op by op (dwconv.hip)
            B["march"] = bool(L.dw_march and not B["small"] and lib.raw("mliis_dwconv_bn_supported")(N, hi, hi, ce, b.k, b.stride))
            if act_dtype != torch.float32 and not (B["march"] or B["small"]):
                raise MliisError("bf16 storage: block {} ({}x{}x{}, k {}, stride {}) is taken by neither fused depthwise family".format(
                    b.idx, hi, hi, ce, b.k, b.stride))
            if b.expand != 1:
                B["z0"], B["st0"] = xbuf(N, hi, hi, ce), vec(ce)
                if not B["march"]:   # (the marching kernels apply bn0 + swish on load: a0 is never written; small-map blocks run op by op in inference)
                    B["a0"] = xbuf(N, hi, hi, ce)
            B["z1"], B["a1"], B["st1"] = xbuf(N, ho, ho, ce), xbuf(N, ho, ho, ce), vec(ce)
            if B["small"]:   # the fused small-map kernels save z0 (a copy) and z1 in their group-blocked layout for the backward launch
                B["z0b"] = xbuf(N, hi, hi, ce)
            B["s"], B["hpre"], B["gate"] = buf(N, ce), buf(N, b.se), buf(N, ce)
            B["z2"], B["st2"], B["out"] = buf(N, ho, ho, b.cout), vec(b.cout), buf(N, ho, ho, b.cout)
            B["dout"] = buf(N, ho, ho, b.cout)
            B["dgate"], B["dpre1"], B["dpre2"], B["chan_add"] = buf(N, ce), buf(N, b.se), buf(N, ce), buf(N, ce)
            if b.skip:
                B["dc"] = self.dc_all[si]
                si += 1
            # gradients w.r.t. the expanded activations: one pair PER BLOCK (not a shared scratch) so the weight-gradient kernels of a
            # block can run on the side stream while the main stream already works on the next block
            B["da2"] = xbuf(N, ho, ho, ce)
            # (a block without an expand conv: the depthwise backward's output is the gradient of the block's fp32 input)
            B["da0"] = xbuf(N, hi, hi, ce) if b.expand != 1 else buf(N, hi, hi, ce)
            self.blocks.append(B)
        self.dstem = buf(N, hs, hs, a.stem_out)
        ex0 = [b for b in a.blocks if b.executed]
        # block 0 without an expand conv (EfficientNet-B0 ... B7) takes the stem's BN + swish into its depthwise launch: the activated
        # stem output is only read there (no identity skip), so it is never written
        self.fuse_stem = bool(ex0 and self.blocks[0]["march"] and ex0[0].expand == 1 and not ex0[0].skip)
        if not self.fuse_stem:
            self.a_stem = buf(N, hs, hs, a.stem_out)
        self.rsd = []
        for m in a.rsd:
            D = {}
            h = m.h
            D["cat"] = buf(N, h, h, m.c_cat)
            D["z0"], D["z1"], D["zf"] = buf(N, h, h, m.c_out), buf(N, h, h, m.c_out), buf(N, h, h, m.c_out)
            D["st0"], D["st1"], D["stf"] = vec(m.c_out), vec(m.c_out), vec(m.c_out)
            D["pyr"] = buf(N, h, h, 2 * m.c_out)        # the pooled third of the reference's "pyramid" is never materialised
            D["pool"], D["dpool"] = buf(N, m.c_cat), buf(N, m.c_cat)
            D["pool_part"] = buf(max(1, ops.rsd_concat_pool_floats(N, h, h, m.c_cat)))
            D["bbias"], D["tot"] = buf(N, 9, m.c_out), buf(N, m.c_out)
            D["out"], D["dout"] = buf(N, h, h, m.c_out), buf(N, h, h, m.c_out)
            D["dzf"], D["dpyr"], D["dcat"] = buf(N, h, h, m.c_out), buf(N, h, h, 2 * m.c_out), buf(N, h, h, m.c_cat)
            if m.upsample_conv:   # the residual operand's own 1x1 branch (efficientlab.py:213-215), deep channels != c_out
                D["zu"], D["stu"], D["up2"] = buf(N, h, h, m.c_out), vec(m.c_out), buf(N, h, h, m.c_out)
                D["dzu"], D["dup"] = buf(N, h, h, m.c_out), buf(N, h, h, m.c_deep)
            self.rsd.append(D)
        self.skipdec = None
        if a.skipdec is not None:   # --skip_decoding (efficientlab.py:133-149)
            sd, h = a.skipdec, a.skipdec.h
            T = dict(cat=buf(N, h, h, sd.c_cat), dcat=None, z0=buf(N, h, h, sd.c_skip), st0=vec(sd.c_skip), dz0=buf(N, h, h, sd.c_skip),
                     dout=buf(N, h, h, sd.c_sep), sep=[])
            cin = sd.c_cat
            for _ in range(2):
                T["sep"].append(dict(zd=buf(N, h, h, cin), std=vec(cin), ad=buf(N, h, h, cin), zp=buf(N, h, h, sd.c_sep), stp=vec(sd.c_sep),
                                     out=buf(N, h, h, sd.c_sep), dad=buf(N, h, h, cin), din=buf(N, h, h, cin)))
                cin = sd.c_sep
            self.skipdec = T
        self.aspp = None
        if a.aspp:   # --spatial_pyramid_pooling (models/efficientlab.py:248-289)
            h, ci, d = a.aspp_h, a.aspp_cin, a.aspp_dimension
            self.aspp = dict(z0=buf(N, h, h, d), z1=buf(N, h, h, d), cat=buf(N, h, h, 3 * d), dcat=buf(N, h, h, 3 * d), zo=buf(N, h, h, d),
                             out=buf(N, h, h, d), dout=buf(N, h, h, d), dzo=buf(N, h, h, d), pool=buf(N, ci), dpool=buf(N, ci),
                             z2=buf(N, d), b2=buf(N, d), db2=buf(N, d),
                             masks=[buf(N, h, h, d), buf(N, h, h, d), buf(N, d), buf(N, h, h, d)])
        hd = a.h_dec
        self.small, self.dsmall = buf(N, hd, hd, 2), buf(N, hd, hd, 2)
        self.logits, self.dlogits, self.pred = buf(N, H, H, 2), buf(N, H, H, 2), buf(N, H, H, 2)
        self.drop_mask = buf(N, hd, hd, a.c_final) if L.final_layer_dropout_rate > 0 else None
        self.loss_out = torch.zeros(4, dtype=torch.float32, device=dev)
        # stage-1 BN statistics handed from a producer (GEMM epilogue / stats kernel) to the fused fold+apply kernel
        need = 0
        for b in a.blocks:
            if b.executed:
                for rows, c in ((N * b.h_in ** 2, b.cexp), (N * b.h_out ** 2, b.cexp), (N * b.h_out ** 2, b.cout)):
                    need = max(need, -(-rows // 16) * 2 * c, ops.bn_stats_partial_floats(rows, c))
        for m in a.rsd:
            need = max(need, -(-(N * m.h * m.h) // 16) * 2 * m.c_out, ops.bn_stats_partial_floats(N * m.h * m.h, m.c_out))
        if a.skipdec is not None:
            rows = N * a.skipdec.h ** 2
            for c in (a.skipdec.c_skip, a.skipdec.c_cat, a.skipdec.c_sep):
                need = max(need, -(-rows // 16) * 2 * c, ops.bn_stats_partial_floats(rows, c))
        need = max(need, ops.bn_stats_partial_floats(N * hs * hs, a.stem_out))
        self.stats_part = buf(need + 64)
        # the row-marching depthwise kernels (ops.dwconv_bn_fwd / _bwd) READ the producer's partial sums from stats_part while other
        # workgroups of the same launch already WRITE theirs: a second buffer
        need2 = 0
        for b in a.blocks:
            if b.executed:
                need2 = max(need2, lib.raw("mliis_dwconv_bn_fwd_blocks")(N, b.h_in, b.h_in, b.cexp, b.k, b.stride) * 2 * b.cexp,
                            lib.raw("mliis_dwconv_bn_bwd_blocks")(N, b.h_in, b.h_in, b.cexp, b.k, b.stride) * 2 * b.cexp)
        for m in a.rsd:   # (and the second RSD branch GEMM's statistics, folded together with the first's by ops.bn_apply_fused_pair)
            need2 = max(need2, -(-(N * m.h * m.h) // 16) * 2 * m.c_out, ops.bn_stats_partial_floats(N * m.h * m.h, m.c_out))
        self.stats_part2 = buf(need2 + 64)
        # the squeeze-excite backward and the depthwise batch norm's backward share ONE pass over (da2, z1) (ops.se_bn_bwd_sums): its
        # per-image chunk sums, and the batch norm's stage-1 sums per image that ops.se_mlp_bwd_bn forms from them
        self.sums_part = buf(max([ops.se_bn_bwd_sums_floats(N, b.h_out * b.h_out, b.cexp) for b in a.blocks if b.executed] + [0]) + 64)
        self.stage1_se = buf(max([2 * N * b.cexp for b in a.blocks if b.executed] + [0]) + 64)
        # squeeze-excite pooling partials of the bn1 apply pass: [N][ceil(rows_per_img / 128)][C]
        self.pool_part = buf(max(N * (-(-(b.h_out * b.h_out) // 128)) * b.cexp for b in a.blocks if b.executed) + 64)
        # gate-gradient partials of the project backward-data launch on the small maps: [16-row groups][2][C]
        self.gate_part = buf(max([(-(-(N * b.h_out * b.h_out) // 16)) * 2 * b.cexp for b in a.blocks if b.executed and 16 <= b.h_out * b.h_out <= 256]
                                 + [0]) + 64)
        # ---- deferred weight-gradient folds: every *_bwd_filter leaves its per-split slabs in a region of fold_buf and ONE
        #      mliis_fold_batched launch at the end of the backward pass reduces them all into the gradient arena
        A = L.arena
        regs, rows, off, tile = {}, [], 0, 0
        fold_tile = lib.raw("mliis_fold_tile_outputs")()

        def add(name, ws_floats, total, seg=None, key=None):
            nonlocal off, tile
            seg_len, seg_stride, seg_off = seg or (total, 0, 0)
            regs[key or name] = (off, ws_floats)
            rows.append([off, A.t_off[name], total, seg_len, seg_stride, seg_off, ws_floats // total, tile])
            off += (ws_floats + 3) // 4 * 4
            tile += -(-total // fold_tile)
        fe = a.name
        add(f"{fe}/stem/conv2d/kernel", lib.size("mliis_stem_conv_bwd_filter_workspace_floats", N, H, H, a.stem_out), 27 * a.stem_out)
        for b, nm, B in zip([b for b in a.blocks if b.executed], L.n_blocks, self.blocks):
            ce = b.cexp
            if b.expand != 1:
                add(nm["w_exp"], lib.size("mliis_conv2d_bwd_filter_workspace_floats", N, b.h_in, b.h_in, b.cin, ce, 1), b.cin * ce)
            if B["march"]:
                add(nm["w_dw"], lib.raw("mliis_dwconv_bn_bwd_blocks")(N, b.h_in, b.h_in, ce, b.k, b.stride) * b.k * b.k * ce, b.k * b.k * ce)
            elif not B["small"]:   # (the small-map backward kernel writes the complete depthwise filter gradient itself: no slabs)
                add(nm["w_dw"], lib.size("mliis_dwconv_bwd_filter_workspace_floats", N, b.h_in, b.h_in, ce, b.k, b.stride), b.k * b.k * ce)
            add(nm["w_proj"], lib.size("mliis_conv2d_bwd_filter_workspace_floats", N, b.h_out, b.h_out, ce, b.cout, 1), ce * b.cout)
        if a.skipdec is not None:
            sd, h = a.skipdec, a.skipdec.h
            ksk, seps = L.n_skipdec
            add(ksk[0], lib.size("mliis_conv2d_bwd_filter_workspace_floats", N, h, h, sd.c_skip_in, sd.c_skip, 1), sd.c_skip_in * sd.c_skip)
            cin = sd.c_cat
            for (dwn, _, pwn, _) in seps:
                add(dwn, lib.size("mliis_dwconv_bwd_filter_workspace_floats", N, h, h, cin, 3, 1), 9 * cin)
                add(pwn, lib.size("mliis_conv2d_bwd_filter_workspace_floats", N, h, h, cin, sd.c_sep, 1), cin * sd.c_sep)
                cin = sd.c_sep
        self.filter_tail = {}
        for j_rsd, (m, nm) in enumerate(zip(a.rsd, L.n_rsd)):
            (k0, b0_, _), (k1, b1_, _), (kf, _, _) = nm
            co = m.c_out
            if m.upsample_conv:
                ku, bu, _ = L.n_rsd_up[j_rsd]
                add(ku, lib.size("mliis_conv2d_bwd_filter_workspace_floats", N, m.h, m.h, m.c_deep, co, 1), m.c_deep * co)
                add(bu, ops.bn_bwd_dxsum_floats(N * m.h * m.h, co), co)
            for bias in (b0_, b1_):   # conv-bias gradients: column sums of dz leave the BN backward pass as slabs
                add(bias, ops.bn_bwd_dxsum_floats(N * m.h * m.h, co), co)
            # filter gradients over the concatenated [deep | skip] channels.  A channel count like 136 = 2 * 64 + 8 leaves a third of
            # the 64-channel blocks of the filter-gradient kernel nearly empty while they still occupy a CU slot each: the sliver
            # (c_cat mod 64 <= 16 channels) gets its own small launch and fold region instead (profiles/r01_notes.md).
            tail = m.c_cat % 64 if (m.c_cat > 64 and 0 < m.c_cat % 64 <= 16) else 0
            self.filter_tail[j_rsd] = tail
            for kk, kname in ((1, k0), (3, k1)):
                main_c = m.c_cat - tail
                add(kname, lib.size("mliis_conv2d_bwd_filter_workspace_floats", N, m.h, m.h, main_c, co, kk), kk * kk * main_c * co,
                    seg=(main_c * co, m.c_cat * co, 0) if tail else None)
                if tail:
                    add(kname, lib.size("mliis_conv2d_bwd_filter_workspace_floats", N, m.h, m.h, tail, co, kk), kk * kk * tail * co,
                        seg=(tail * co, m.c_cat * co, main_c * co), key=kname + "#tail")
            add(kf, lib.size("mliis_conv2d_bwd_filter_workspace_floats", N, m.h, m.h, 2 * co, co, 3), 9 * 2 * co * co,
                seg=(2 * co * co, m.c_pyr * co, 0))
        # squeeze-excite weight gradients of all blocks: one launch (descriptor table of device addresses)
        rows_se, se_tile = [], 0
        for b, B, nm in zip([b for b in a.blocks if b.executed], self.blocks, L.n_blocks):
            se = nm["se"]
            rows_se.append([B["s"].data_ptr(), B["hpre"].data_ptr(), B["dpre1"].data_ptr(), B["dpre2"].data_ptr()] +
                           [A.g[k].data_ptr() for k in se] + [N, b.cexp, b.se, se_tile])
            se_tile += -(-(2 * b.cexp * b.se + b.cexp + b.se) // 256)
        self.se_desc = torch.tensor(rows_se, dtype=torch.int64, device=dev)
        self.se_tiles = se_tile
        self.fold_buf = buf(off + 16)
        self.fold_part = {k: self.fold_buf[o:o + n] for k, (o, n) in regs.items()}
        self.fold_desc = torch.tensor(rows, dtype=torch.int64, device=dev)
        self.fold_tiles = tile
        # captured hipGraphExecs of the training step: key True = the step draws its masks on the device (mliis_rng_masks inside the graph),
        # False = masks were handed in by the caller (parity tests inject them) and the graph starts after them
        self.graphs = {}
        self.steps_run = 0
        # deferred dense-conv filter gradients: collected during the first (eager) backward pass of this plan, then one launch per
        # kernel instantiation at the end of every backward pass (ops.FilterBatch)
        self.wbatch = ops.FilterBatch(dev)
        self.wbatch_ready = False
        # ---- mask generation inside the step (ops.rng_masks): drop-connect scales of all skip blocks, final-layer dropout, ASPP dropouts
        jobs = []
        if L.drop_connect and nskip:
            jobs.append((self.dc_all, L._dc_keeps, N, True))
        if self.drop_mask is not None:
            jobs.append((self.drop_mask, L.drop_keep_dev, self.drop_mask.numel(), False))
        if self.aspp is not None:
            for mbuf in self.aspp["masks"]:
                jobs.append((mbuf, 1.0 - spec.ASPP_DROPOUT, 1, False))
        self.mask_plan = ops.MaskPlan(jobs) if jobs else None

    @property
    def graph(self):
        """Any captured graph of this plan (None: none yet)."""
        for g in self.graphs.values():
            return g
        return None


class Learner:
    def __init__(self, feature_extractor_name: str = "efficientnet-b0", image_size: int = 224, rsd: Optional[Sequence[int]] = (2, 4),
                 learning_rate: float = 1e-3, optimizer: str = "sgd", l2: bool = False, l1: bool = False, darc1: bool = False,
                 dice: bool = False, label_smoothing: float = 0.0, final_layer_dropout_rate: float = 0.0,
                 spatial_pyramid_pooling: bool = False, skip_decoding: bool = False, drop_connect: bool = True, seed: int = 0,
                 device="cuda:0", use_graph: bool = True, max_shots: int = 16, matmul_precision: str = "fp32",
                 small_fused: Optional[bool] = None, dw_march: Optional[bool] = None,
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
        if matmul_precision not in ops.PRECISIONS:
            raise ValueError("matmul_precision must be one of {} or 'bf16-storage', got {!r}".format(sorted(ops.PRECISIONS), matmul_precision))
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
        self.use_graph = use_graph
        self.stream = torch.cuda.Stream(device=self.device)
        # batch indices go up through a ring of pinned slots: an upload from pageable memory makes the host wait for this stream
        self._idx_pin = torch.empty((16, 64), dtype=torch.int32).pin_memory()
        self._idx_ev = [None] * 16
        self._idx_n = 0
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
        self._in()
        with torch.cuda.stream(self.stream):
            self.arena.theta.copy_(st["theta"])
            self.arena.bn_moving.copy_(st["bn"])
            if self.adam_v is not None and "adam_v" in st:
                self.adam_v.copy_(st["adam_v"])
                self.adam_t.copy_(st["adam_t"])
                self.adam_epoch += 1

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

    # ------------------------------------------------------------------------------------------- forward
    def _forward(self, P: _Plan, x, idx, training: bool):
        A, a, ws, N = self.arena, self.arch, self.ws, P.N
        w, mv = A.w, A.mv

        def bn(xin, st, prefix, y, pre=False, post=False, img_scale=None, res=None, fused=False, nblk=0, pool_part=None, always_batch=False,
               part=None):
            """nblk > 0: the producing conv already left the stage-1 statistics in P.stats_part.  always_batch: a batch norm the
            reference builds with training=True (the --skip_decoding decoder): batch statistics in inference too, moving averages
            untouched there."""
            if training or always_batch:
                part = P.stats_part if (part is None or nblk == 0) else part
                if nblk == 0:
                    nblk = ops.bn_stats_partial(xin, pre, P.stats_part)
                return ops.bn_apply_fused(xin, part, nblk, st[0], st[1], w[prefix + "/gamma"], w[prefix + "/beta"],
                                          moving=(mv[prefix + "/moving_mean"], mv[prefix + "/moving_variance"]) if training else None,
                                          unbiased_moving_var=fused,
                                          pre_swish=pre, post_swish=post, img_scale=img_scale, res=res, out=y, pool_part=pool_part)
            st[0].copy_(mv[prefix + "/moving_mean"])
            torch.rsqrt(mv[prefix + "/moving_variance"] + spec.BN_EPS, out=st[1])
            return ops.bn_apply(xin, st[0], st[1], w[prefix + "/gamma"], w[prefix + "/beta"], pre, post, img_scale, res, out=y)

        ops.transpose_weights(A.theta, self.theta_t, self.wt_desc, self.w_amax, tiles=self.wt_tiles)

        def conv(xin, wname, bname, dil, out, swish_stats, x_scale=None, border_bias=None):
            """dense conv; in training the epilogue also emits the following BN's statistics (returns their block count)."""
            am = self._amax_of.get(wname)
            if training:
                return self._conv_fwd(xin, w[wname], w[bname] if bname else None, dil, out=out, ws=ws, stats_part=P.stats_part,
                                      stats_swish=swish_stats, wt=self.wt[wname], x_scale=x_scale, border_bias=border_bias, fp8_w_amax=am)[1]
            self._conv_fwd(xin, w[wname], w[bname] if bname else None, dil, out=out, ws=ws, wt=self.wt[wname], x_scale=x_scale,
                           border_bias=border_bias, fp8_w_amax=am)
            return 0

        ops.stem_conv_fwd(x, w[self.n_stem[0]], idx, out=P.z_stem)
        ex = [b for b in a.blocks if b.executed]
        fuse_stem = P.fuse_stem   # (block 0 takes the stem's BN + swish into its depthwise launch: _Plan)
        if fuse_stem:
            cur = None
        else:
            cur = bn(P.z_stem, P.st_stem, self.n_stem[1], P.a_stem, post=True)

        def bn_in(z, st, prefix, nblk):
            """The batch norm in front of a marching depthwise launch: (bn tuple, nblk) for ops.dwconv_bn_fwd.  Training: the launch
            folds the producer's partial sums (P.stats_part) and updates the moving averages; inference: moving statistics given."""
            g_, b_ = w[prefix + "/gamma"], w[prefix + "/beta"]
            if training:
                if nblk == 0:
                    nblk = ops.bn_stats_partial(z, False, P.stats_part)
                return (g_, b_, st[0], st[1], mv[prefix + "/moving_mean"], mv[prefix + "/moving_variance"]), nblk
            st[0].copy_(mv[prefix + "/moving_mean"])
            torch.rsqrt(mv[prefix + "/moving_variance"] + spec.BN_EPS, out=st[1])
            return (g_, b_, st[0], st[1], None, None), 0

        for bi_, (b, B, nm) in enumerate(zip(ex, P.blocks, self.n_blocks)):
            B["x_in"] = cur
            t = cur
            if training and B["small"]:
                # expand GEMM (+ stage-1 statistics) -> ONE launch: bn0 fold + apply + swish, depthwise, bn1 statistics + apply + swish,
                # squeeze-excite means, both moving averages -> SE MLP -> project GEMM
                nb = conv(t, nm["w_exp"], None, 1, B["z0"], False)
                if nb == 0:
                    nb = ops.bn_stats_partial(B["z0"], False, P.stats_part)
                p0, p1 = nm["bn0"], nm["bn1"]
                ops.mbconv_dw_fwd_small(B["z0"], P.stats_part, nb,
                                        (w[p0 + "/gamma"], w[p0 + "/beta"], B["st0"][0], B["st0"][1], mv[p0 + "/moving_mean"], mv[p0 + "/moving_variance"]),
                                        w[nm["w_dw"]],
                                        (w[p1 + "/gamma"], w[p1 + "/beta"], B["st1"][0], B["st1"][1], mv[p1 + "/moving_mean"], mv[p1 + "/moving_variance"]),
                                        B["z1"], B["a1"], B["s"], z0_blocked=B["z0b"], z1_blocked=True)   # (the backward's re-reads: contiguous)
                se = nm["se"]
                ops.se_mlp_fwd(B["s"], w[se[0]], w[se[1]], w[se[2]], w[se[3]], B["hpre"], B["gate"])
                nb = conv(B["a1"], nm["w_proj"], None, 1, B["z2"], False, x_scale=B["gate"])
                use_dc = self.drop_connect and b.skip and b.drop_rate > 0
                B["use_dc"] = use_dc
                cur = bn(B["z2"], B["st2"], nm["bn2"], B["out"], img_scale=B["dc"] if use_dc else None, res=B["x_in"] if b.skip else None, nblk=nb)
                continue
            if B["march"]:
                # expand GEMM (+ stage-1 statistics) -> ONE launch: bn0 fold + apply + swish while the rows are staged, depthwise conv,
                # bn1 stage-1 statistics (P.stats_part2)
                if b.expand != 1:
                    nb = conv(t, nm["w_exp"], None, 1, B["z0"], False)
                    bn0, nb = bn_in(B["z0"], B["st0"], nm["bn0"], nb)
                    zin = B["z0"]
                elif bi_ == 0 and fuse_stem:
                    bn0, nb = bn_in(P.z_stem, P.st_stem, self.n_stem[1], 0)
                    zin = P.z_stem
                else:
                    bn0, nb, zin = None, 0, t
                if training:
                    nb = ops.dwconv_bn_fwd(zin, w[nm["w_dw"]], b.stride, bn=bn0, part=P.stats_part, nblk=nb, out=B["z1"],
                                           stats_part=P.stats_part2)[1]
                else:
                    ops.dwconv_bn_fwd(zin, w[nm["w_dw"]], b.stride, bn=bn0, out=B["z1"])
                    nb = 0
                st_part = P.stats_part2
            else:
                if b.expand != 1:
                    nb = conv(t, nm["w_exp"], None, 1, B["z0"], False)
                    t = bn(B["z0"], B["st0"], nm["bn0"], B["a0"], post=True, nblk=nb)
                if training:   # the depthwise launch also leaves bn1's stage-1 statistics in P.stats_part
                    nb = ops.dwconv_fwd(t, w[nm["w_dw"]], b.stride, out=B["z1"], stats_part=P.stats_part)[1]
                else:
                    ops.dwconv_fwd(t, w[nm["w_dw"]], b.stride, out=B["z1"])
                    nb = 0
                st_part = P.stats_part
            hw = b.h_out * b.h_out
            se = nm["se"]
            if training:   # bn1's apply pass also pools its output per image (partial sums); the SE kernel folds them
                chunks = bn(B["z1"], B["st1"], nm["bn1"], B["a1"], post=True, nblk=nb, pool_part=P.pool_part, part=st_part)[1]
                ops.se_mlp_fwd(P.pool_part, w[se[0]], w[se[1]], w[se[2]], w[se[3]], B["hpre"], B["gate"], chunks=chunks, scale=1.0 / hw,
                               s_out=B["s"])
            else:
                bn(B["z1"], B["st1"], nm["bn1"], B["a1"], post=True, nblk=nb)
                ops.colsum(B["a1"], None, nseg=N, scale=1.0 / hw, out=B["s"], ws=ws)
                ops.se_mlp_fwd(B["s"], w[se[0]], w[se[1]], w[se[2]], w[se[3]], B["hpre"], B["gate"])
            # squeeze-excite gate applied inside the project GEMM's A loader (the gated tensor is never written)
            nb = conv(B["a1"], nm["w_proj"], None, 1, B["z2"], False, x_scale=B["gate"])
            use_dc = training and self.drop_connect and b.skip and b.drop_rate > 0
            B["use_dc"] = use_dc
            cur = bn(B["z2"], B["st2"], nm["bn2"], B["out"], img_scale=B["dc"] if use_dc else None, res=B["x_in"] if b.skip else None, nblk=nb)
        ends = {r: P.blocks[bi]["out"] for r, bi in a.reductions.items() if bi < len(P.blocks)}
        dec = ends[4]
        if a.aspp:
            dec = self._aspp_forward(P, dec, training)
        if a.skipdec is not None:
            # efficientlab.py:133-149: [resize(embedded, input // 4) | swish(BN(conv1x1(reduction_2)))] -> two sep_convs (dw 3x3 -> BN ->
            # swish -> 1x1 -> BN -> swish).  Every BN here is built with training=True in the reference.
            sd, T = a.skipdec, P.skipdec
            (k0, n0), seps = self.n_skipdec
            cat = T["cat"]
            ops.resize_bilinear_fwd(dec, (sd.h, sd.h), out=cat[..., :sd.c_in])
            nb = conv(ends[2], k0, None, 1, T["z0"], False)
            bn(T["z0"], T["st0"], n0, cat[..., sd.c_in:], post=True, fused=True, nblk=nb, always_batch=True)
            cur_sd = cat
            for S, (dwn, dbn, pwn, pbn) in zip(T["sep"], seps):
                S["x_in"] = cur_sd
                if training:
                    nb = ops.dwconv_fwd(cur_sd, w[dwn], 1, out=S["zd"], stats_part=P.stats_part)[1]
                else:
                    ops.dwconv_fwd(cur_sd, w[dwn], 1, out=S["zd"])
                    nb = 0
                bn(S["zd"], S["std"], dbn, S["ad"], post=True, fused=True, nblk=nb, always_batch=True)
                nb = conv(S["ad"], pwn, None, 1, S["zp"], False)
                cur_sd = bn(S["zp"], S["stp"], pbn, S["out"], post=True, fused=True, nblk=nb, always_batch=True)
            dec = cur_sd
        for j_rsd, (m, D, nm, r) in enumerate(zip(a.rsd, P.rsd, self.n_rsd, sorted([x.scope_index + 1 for x in a.rsd], reverse=True))):
            skip = ends[r]
            cat = D["cat"]
            up = cat[..., :m.c_deep]
            # the concat of the (resized) deep map and the skip feature, and the pooled branch's per-image sums of it: one launch
            pool_chunks = 0
            if m.c_deep % 4 == 0 and (m.c_cat - m.c_deep) % 4 == 0 and m.h > 1:
                pool_chunks = ops.rsd_concat_pool(dec, skip, cat, D["pool_part"])
            else:
                if m.h_in == m.h:
                    ops.chan_affine(dec, out=up)
                else:
                    ops.resize_bilinear_fwd(dec, (m.h, m.h), out=up)
                ops.chan_affine(skip, out=cat[..., m.c_deep:])
            res_up = up
            if m.upsample_conv:   # the residual operand through its own conv -> swish -> BN branch; the concat keeps the resized map
                ku, bu, nu = self.n_rsd_up[j_rsd]
                nb = conv(up, ku, bu, 1, D["zu"], True)
                res_up = bn(D["zu"], D["stu"], nu, D["up2"], pre=True, fused=True, nblk=nb)
            pyr = D["pyr"]
            (k0, b0, n0), (k1, b1, n1), (kf, bf, nf) = nm
            if training:
                # the 1x1 and the 3x3-dilated branch are independent: both GEMMs first (statistics in two buffers), then ONE launch for
                # the two conv -> swish -> BN tails
                nb0 = self._conv_fwd(cat, w[k0], w[b0], 1, out=D["z0"], ws=ws, stats_part=P.stats_part, stats_swish=True, wt=self.wt[k0],
                                     fp8_w_amax=self._amax_of.get(k0))[1]
                nb1 = self._conv_fwd(cat, w[k1], w[b1], 2, out=D["z1"], ws=ws, stats_part=P.stats_part2, stats_swish=True, wt=self.wt[k1],
                                     fp8_w_amax=self._amax_of.get(k1))[1]
                ops.bn_apply_fused_pair([(D["z" + i], pt, nb_, D["st" + i][0], D["st" + i][1], w[nn + "/gamma"], w[nn + "/beta"],
                                          (mv[nn + "/moving_mean"], mv[nn + "/moving_variance"]), out_)
                                         for i, pt, nb_, nn, out_ in (("0", P.stats_part, nb0, n0, pyr[..., :m.c_out]),
                                                                      ("1", P.stats_part2, nb1, n1, pyr[..., m.c_out:2 * m.c_out]))],
                                        pre_swish=True, unbiased_moving_var=True)
            else:
                nb = conv(cat, k0, b0, 1, D["z0"], True)
                bn(D["z0"], D["st0"], n0, pyr[..., :m.c_out], pre=True, fused=True, nblk=nb)
                nb = conv(cat, k1, b1, 2, D["z1"], True)
                bn(D["z1"], D["st1"], n1, pyr[..., m.c_out:2 * m.c_out], pre=True, fused=True, nblk=nb)
            # pooled branch: per-image mean of `cat`, folded into the fuse conv as a border-class bias (rsd.hip)
            if pool_chunks:
                ops.rsd_pool_fwd(D["pool_part"], w[kf], 2 * m.c_out, out=D["bbias"], chunks=pool_chunks, scale=1.0 / (m.h * m.h), pool_out=D["pool"])
            else:
                ops.colsum(cat, None, nseg=N, scale=1.0 / (m.h * m.h), out=D["pool"], ws=ws)
                ops.rsd_pool_fwd(D["pool"], w[kf], 2 * m.c_out, out=D["bbias"])
            nb = conv(pyr, kf, bf, 1, D["zf"], True, border_bias=D["bbias"])
            dec = bn(D["zf"], D["stf"], nf, D["out"], pre=True, res=res_up, fused=True, nblk=nb)
        mask = P.drop_mask if (training and P.drop_mask is not None) else None
        P.dec_in = dec
        ops.final_conv_fwd(dec, w[self.n_final[0]], w[self.n_final[1]], mask, out=P.small)
        H = a.image_size
        ops.resize_bilinear_fwd(P.small, (H, H), out=P.logits)
        return P.logits

    # ------------------------------------------------------------------------------------------- ASPP (--spatial_pyramid_pooling)
    def _aspp_forward(self, P: _Plan, x, training: bool):
        """models/efficientlab.py:248-289 on the encoder output x [N,h,h,Cin]: 1x1 / 3x3-dilation-6 / image-pooling branches written
        straight into channel slices of the concat buffer ([pooled | 3x3 | 1x1], the reference's order), then 1x1 conv + swish +
        dropout.  The dense convs are the MFMA implicit GEMM, the activations mliis_swish_mask_*."""
        a, w, ws, T, N = self.arch, self.arena.w, self.ws, P.aspp, P.N
        d, hw = a.aspp_dimension, a.aspp_h * a.aspp_h
        (k0, c0), (k1, c1), (k2, c2), (ko, co) = self.n_aspp
        m = T["masks"] if training else [None] * 4
        cat = T["cat"]
        self._conv_fwd(x, w[k0], w[c0], 1, out=T["z0"], ws=ws, wt=self.wt[k0], fp8_w_amax=self._amax_of.get(k0))
        ops.swish_mask_fwd(T["z0"], m[0], out=cat[..., 2 * d:])
        self._conv_fwd(x, w[k1], w[c1], spec.ASPP_DILATION, out=T["z1"], ws=ws, wt=self.wt[k1])
        ops.swish_mask_fwd(T["z1"], m[1], out=cat[..., d:2 * d])
        ops.colsum(x, None, nseg=N, scale=1.0 / hw, out=T["pool"], ws=ws)
        self._conv_fwd(T["pool"].view(N, 1, 1, -1), w[k2], w[c2], 1, out=T["z2"].view(N, 1, 1, d), ws=ws, wt=self.wt[k2],
                       fp8_w_amax=self._amax_of.get(k2))
        ops.swish_mask_fwd(T["z2"], m[2], out=T["b2"], pre_mask=True)
        ops.chan_affine(None, A=T["b2"], out=cat[..., :d])      # bilinear resize of the 1x1 pooled map = broadcast
        self._conv_fwd(cat, w[ko], w[co], 1, out=T["zo"], ws=ws, wt=self.wt[ko], fp8_w_amax=self._amax_of.get(ko))
        ops.swish_mask_fwd(T["zo"], m[3], out=T["out"])
        T["trained"] = training
        return T["out"]

    def _aspp_backward(self, P: _Plan, x, dx, dx_has: bool):
        """Gradients of the ASPP parameters (straight into the gradient arena) and of its input (accumulated into dx when dx_has)."""
        a, A, ws, T, N = self.arch, self.arena, self.ws, P.aspp, P.N
        w, g = A.w, A.g
        d, hw = a.aspp_dimension, a.aspp_h * a.aspp_h
        (k0, c0), (k1, c1), (k2, c2), (ko, co) = self.n_aspp
        m = T["masks"]
        cat, dcat = T["cat"], T["dcat"]
        dzo = ops.swish_mask_bwd(T["dout"], T["zo"], m[3], out=T["dzo"])
        self._conv_bwd_filter(cat, dzo, 1, 1, out=g[ko], ws=ws)
        ops.colsum(dzo, out=g[co], ws=ws)
        self._conv_bwd_data(dzo, w[ko], 1, out=dcat, ws=ws)
        # 1x1 branch (the pre-activation gradient overwrites its slice of dcat)
        d0 = ops.swish_mask_bwd(dcat[..., 2 * d:], T["z0"], m[0], out=dcat[..., 2 * d:])
        self._conv_bwd_filter(x, d0, 1, 1, out=g[k0], ws=ws)
        ops.colsum(d0, out=g[c0], ws=ws)
        self._conv_bwd_data(d0, w[k0], 1, out=dx, accumulate=dx_has, ws=ws)
        # 3x3 dilation-6 branch
        d1 = ops.swish_mask_bwd(dcat[..., d:2 * d], T["z1"], m[1], out=dcat[..., d:2 * d])
        self._conv_bwd_filter(x, d1, 3, spec.ASPP_DILATION, out=g[k1], ws=ws)
        ops.colsum(d1, out=g[c1], ws=ws)
        self._conv_bwd_data(d1, w[k1], spec.ASPP_DILATION, out=dx, accumulate=True, ws=ws)
        # image-pooling branch: per-image sums of the broadcast slice -> [N, d] chain -> mean's gradient on every pixel
        ops.colsum(dcat[..., :d], None, nseg=N, out=T["db2"], ws=ws)
        d2 = ops.swish_mask_bwd(T["db2"], T["z2"], m[2], out=T["db2"], pre_mask=True)
        pool4, d24 = T["pool"].view(N, 1, 1, -1), d2.view(N, 1, 1, d)
        self._conv_bwd_filter(pool4, d24, 1, 1, out=g[k2], ws=ws)
        ops.colsum(d2, out=g[c2], ws=ws)
        self._conv_bwd_data(d24, w[k2], 1, out=T["dpool"].view(N, 1, 1, -1), ws=ws)
        ops.axpby(0.0, None, 1.0 / hw, T["dpool"])                       # d(mean)/dx = 1 / (h*w) on every pixel
        ops.chan_affine(None, A=T["dpool"], out=dx, accumulate=True)

    # ------------------------------------------------------------------------------------------- backward
    def _backward(self, P: _Plan, x, idx):
        A, a, ws, N = self.arena, self.arch, self.ws, P.N
        w, g = A.w, A.g
        hd = a.h_dec
        ops.resize_bilinear_bwd(P.dlogits, (hd, hd), out=P.dsmall)
        mask = P.drop_mask
        ops.final_conv_bwd_filter(P.dec_in, P.dsmall, mask, dw=g[self.n_final[0]], db=g[self.n_final[1]], ws=ws)
        ex = [b for b in a.blocks if b.executed]
        has_grad = [False] * len(P.blocks)
        dtop = P.rsd[-1]["dout"] if P.rsd else (P.skipdec["dout"] if a.skipdec is not None else
                                                 (P.aspp["dout"] if a.aspp else P.blocks[-1]["dout"]))
        ops.final_conv_bwd_data(P.dsmall, w[self.n_final[0]], a.c_final, mask, out=dtop)
        if not P.rsd and not a.aspp and a.skipdec is None:
            has_grad[-1] = True

        def bn_b(xin, dy, st, prefix, dx, pre=False, post=False, img_scale=None, chan_scale=None, chan_add=None, dskip=None,
                 dskip_accumulate=False, dxsum_part=None, stage1=None):
            ops.bn_bwd(xin, dy, st[0], st[1], w[prefix + "/gamma"], w[prefix + "/beta"], pre, post, img_scale, chan_scale, chan_add, dx=dx,
                       dgamma=g[prefix + "/gamma"], dbeta=g[prefix + "/beta"], ws=ws, dskip=dskip, dskip_accumulate=dskip_accumulate,
                       dxsum_part=dxsum_part, stage1=stage1)

        if not P.wbatch_ready:   # (a first backward pass that raised half-way must not leave half a table behind)
            P.wbatch = ops.FilterBatch(self.device)

        def wgrad_conv(xin, dz, kk, dil, key, x_scale=None):
            """filter gradient of a dense conv (slabs into P.fold_part[key]): deferred into the plan's batch, launched at the end of the
            pass, one launch per kernel instantiation.  (Round 4 ran the decoder's share on a side branch of the captured step with
            capped grids beside the encoder's backward chain: measured neutral to negative -- profiles/r04_notes.md -- and removed.)"""
            if not P.wbatch_ready:
                P.wbatch.add(xin, dz, kk, dil, P.fold_part[key], x_scale=x_scale)

        def wgrad_1x1(xin, dz, kname, x_scale=None):
            wgrad_conv(xin, dz, 1, 1, kname, x_scale=x_scale)

        rs = sorted([x.scope_index + 1 for x in a.rsd], reverse=True)
        for j in range(len(a.rsd) - 1, -1, -1):
            m, D, nm, r = a.rsd[j], P.rsd[j], self.n_rsd[j], rs[j]
            (k0, b0, n0), (k1, b1, n1), (kf, bf, nf) = nm
            co, hw = m.c_out, m.h * m.h
            dO, cat, pyr, dpyr, dcat = D["dout"], D["cat"], D["pyr"], D["dpyr"], D["dcat"]
            bn_b(D["zf"], dO, D["stf"], nf, D["dzf"], pre=True)
            ops.rsd_pool_bwd(D["dzf"], D["tot"], D["pool"], w[kf], 2 * co, dw=g[kf], dbias=g[bf], dpool=D["dpool"], ws=ws)
            wgrad_conv(pyr, D["dzf"], 3, 1, kf)   # rows of the 2*co convolved channels
            self._conv_bwd_data(D["dzf"], w[kf], 1, ci_begin=0, ci_count=2 * co, out=dpyr, ws=ws)
            d0, d1 = dpyr[..., :co], dpyr[..., co:2 * co]
            # both branches' batch norms: one reduce launch + one apply launch (+ conv-bias gradient slabs for the batched fold)
            ops.bn_bwd_pair([(D["z" + i], d_, D["st" + i][0], D["st" + i][1], w[nn + "/gamma"], w[nn + "/beta"], d_, g[nn + "/gamma"],
                              g[nn + "/beta"], P.fold_part[bb]) for i, d_, nn, bb in (("0", d0, n0, b0), ("1", d1, n1, b1))],
                            pre_swish=True, ws=ws)
            tail = P.filter_tail[j]
            cmain = cat[..., :m.c_cat - tail] if tail else cat

            def wgrad(dz, kname, kk, dil, cmain=cmain, ctail=cat[..., m.c_cat - tail:] if tail else None):
                wgrad_conv(cmain, dz, kk, dil, kname)
                if ctail is not None:   # the <= 16-channel sliver of the concat (see _Plan)
                    wgrad_conv(ctail, dz, kk, dil, kname + "#tail")
            wgrad(d0, k0, 1, 1)
            self._conv_bwd_data(d0, w[k0], 1, out=dcat, ws=ws)
            wgrad(d1, k1, 3, 2)
            self._conv_bwd_data(d1, w[k1], 2, out=dcat, accumulate=True, ws=ws)
            # gradient of the concat = dcat + dpool / (H*W) on every pixel (the pooled branch); its deep half joins the residual
            # gradient, its skip half goes to the endpoint's gradient: one pass (mliis_chan_split)
            bi_skip = a.reductions[r]
            if m.upsample_conv:
                # the residual operand came through its own conv -> swish -> BN branch (efficientlab.py:213-215): dO is its gradient;
                # back through that branch to the resized deep map, where the concat's share joins
                ku, bu, nu = self.n_rsd_up[j]
                bn_b(D["zu"], dO, D["stu"], nu, D["dzu"], pre=True, dxsum_part=P.fold_part[bu])
                wgrad_conv(cat[..., :m.c_deep], D["dzu"], 1, 1, ku)
                self._conv_bwd_data(D["dzu"], w[ku], 1, out=D["dup"], ws=ws)
                dU = D["dup"]
            else:
                dU = dO      # dU = dO + dcat[:, :c_deep] (residual)
            ops.chan_split(dcat, m.c_deep, dU, True, P.blocks[bi_skip]["dout"], has_grad[bi_skip], A=D["dpool"])
            has_grad[bi_skip] = True
            # gradient w.r.t. the deep input (for RSD(4) without a decoder in front it is the same endpoint the skip half just went to)
            if j > 0:
                tgt, tgt_has = P.rsd[j - 1]["dout"], False
            elif a.skipdec is not None:
                tgt, tgt_has = P.skipdec["dout"], False
            elif a.aspp:
                tgt, tgt_has = P.aspp["dout"], False
            else:
                bi = a.reductions[4]
                tgt, tgt_has = P.blocks[bi]["dout"], has_grad[bi]
            if m.h_in == m.h:
                ops.chan_affine(dU, out=tgt, accumulate=tgt_has)
            else:
                ops.resize_bilinear_bwd(dU, (m.h_in, m.h_in), out=tgt, accumulate=tgt_has)
            if j == 0 and not a.aspp and a.skipdec is None:
                has_grad[a.reductions[4]] = True

        if a.skipdec is not None:
            # --skip_decoding decoder backward: the two sep_convs in reverse, then the concat's two halves -- the projected reduction_2
            # endpoint (conv1x1 -> BN -> swish) and the resized embedded image
            sd, T = a.skipdec, P.skipdec
            (k0, n0), seps = self.n_skipdec
            d = T["dout"]     # from the first RSD module (or, without RSD modules, the final conv's input gradient)
            for S, (dwn, dbn, pwn, pbn) in zip(reversed(T["sep"]), reversed(seps)):
                bn_b(S["zp"], d, S["stp"], pbn, d, post=True)
                wgrad_conv(S["ad"], d, 1, 1, pwn)
                self._conv_bwd_data(d, w[pwn], 1, out=S["dad"], ws=ws)
                bn_b(S["zd"], S["dad"], S["std"], dbn, S["dad"], post=True)
                ops.dwconv_bwd_filter(S["x_in"], S["dad"], 3, 1, partial=P.fold_part[dwn])
                ops.dwconv_bwd_data(S["dad"], w[dwn], 1, (sd.h, sd.h), out=S["din"])
                d = S["din"]
            dcat_sd = d                                   # [N, h, h, c_in + c_skip]
            bi2 = a.reductions[2]
            bn_b(T["z0"], dcat_sd[..., sd.c_in:], T["st0"], n0, T["dz0"], post=True)
            wgrad_conv(P.blocks[bi2]["out"], T["dz0"], 1, 1, k0)
            self._conv_bwd_data(T["dz0"], w[k0], 1, out=P.blocks[bi2]["dout"], accumulate=has_grad[bi2], ws=ws)
            has_grad[bi2] = True
            if a.aspp:
                tgt, tgt_has = P.aspp["dout"], False
            else:
                bi = a.reductions[4]
                tgt, tgt_has = P.blocks[bi]["dout"], has_grad[bi]
                has_grad[bi] = True
            ops.resize_bilinear_bwd(dcat_sd[..., :sd.c_in], (sd.h_in, sd.h_in), out=tgt, accumulate=tgt_has)
        if a.aspp:
            bi = a.reductions[4]
            self._aspp_backward(P, P.blocks[bi]["out"], P.blocks[bi]["dout"], has_grad[bi])
            has_grad[bi] = True
        stage1_next = None   # stage 1 of the NEXT block's (bi - 1) project-BN backward, when the expand backward-data launch produced it

        def expand_bwd_data(bi, da0, wname, tgt, tgt_has):
            """Backward-data of block bi's expand conv into the gradient of block bi - 1's output -- the last contribution to it, so the
            launch can also emit stage 1 of that block's project-BN backward (mliis_conv2d_bwd_data_bn; small maps only)."""
            if bi == 0:
                self._conv_bwd_data(da0, w[wname], 1, out=tgt, accumulate=tgt_has, ws=ws)
                return None
            Bp = P.blocks[bi - 1]
            _, nb = self._conv_bwd_data(da0, w[wname], 1, out=tgt, accumulate=tgt_has, ws=ws,
                                        bn=(Bp["z2"], Bp["st2"][0], Bp["st2"][1], Bp["dc"] if Bp["use_dc"] else None), part=P.stats_part)
            return (P.stats_part, nb) if nb else None

        for bi in range(len(P.blocks) - 1, -1, -1):
            b, B, nm = ex[bi], P.blocks[bi], self.n_blocks[bi]
            if not has_grad[bi]:
                raise MliisError("internal: block {} has no upstream gradient".format(bi))
            dout = B["dout"]
            ce, hw = b.cexp, b.h_out * b.h_out
            # gradient for the block input: identity-skip part first (before dout is overwritten in place)
            tgt = P.blocks[bi - 1]["dout"] if bi > 0 else P.dstem
            tgt_has = has_grad[bi - 1] if bi > 0 else False
            # identity-skip part of the block-input gradient: written by the same pass that turns dout into the bn2 input gradient
            bn_b(B["z2"], dout, B["st2"], nm["bn2"], dout, img_scale=B["dc"] if B["use_dc"] else None,
                 dskip=tgt if b.skip else None, dskip_accumulate=tgt_has, stage1=stage1_next)
            stage1_next = None
            if b.skip:
                tgt_has = True
            wgrad_1x1(B["a1"], dout, nm["w_proj"], x_scale=B["gate"])
            da2 = B["da2"]
            se = nm["se"]
            groups = 0
            if 16 <= hw <= 256:
                # small maps: the project backward-data launch also leaves the gate gradient's per-row-group partial sums of da2 * a1
                # and the SE kernel folds them -- no pass over the two tensors (mliis_conv2d_bwd_data_gate)
                _, groups = self._conv_bwd_data(dout, w[nm["w_proj"]], 1, out=da2, ws=ws, gate=B["a1"], part=P.gate_part)
            else:
                self._conv_bwd_data(dout, w[nm["w_proj"]], 1, out=da2, ws=ws)
            se_outs = dict(dpre1=B["dpre1"], dpre2=B["dpre2"], chan_add=B["chan_add"])
            bn1_stage1 = None
            if not groups and not B["small"]:
                # ONE pass over (da2, z1): the gate's gradient and everything bn1's backward needs from the two tensors; the SE kernel
                # folds it and emits bn1's stage-1 sums per image -- no column-sum launch, no reduce pass of the batch norm
                st1, p1 = B["st1"], nm["bn1"]
                nbs = ops.se_bn_bwd_sums(B["z1"], da2, st1[0], st1[1], w[p1 + "/gamma"], w[p1 + "/beta"], P.sums_part)
                ops.se_mlp_bwd_bn(P.sums_part, nbs, B["gate"], B["hpre"], w[se[0]], w[se[2]], hw, se_outs, P.stage1_se, w1t=self.wt[se[0]])
                bn1_stage1 = (P.stage1_se, N)
            else:
                if not groups:
                    ops.colsum(da2, B["a1"], nseg=N, out=B["dgate"], ws=ws)
                # (the SE weight gradients of all blocks are computed by one batched launch after the loop: P.se_desc)
                ops.se_mlp_bwd(P.gate_part if groups else B["dgate"], B["gate"], B["s"], B["hpre"], w[se[0]], w[se[2]], hw, se_outs,
                               dgate_groups=groups, w1t=self.wt[se[0]])
            if B["small"]:   # bn1 backward, depthwise filter gradient + backward-data, bn0 backward: one launch
                da0, st0, st1, p0, p1 = B["da0"], B["st0"], B["st1"], nm["bn0"], nm["bn1"]
                ops.mbconv_dw_bwd_small(da2, B["gate"], B["chan_add"], B["z1"], (st1[0], st1[1], w[p1 + "/gamma"], w[p1 + "/beta"]),
                                        w[nm["w_dw"]], B["z0"], (st0[0], st0[1], w[p0 + "/gamma"], w[p0 + "/beta"]),
                                        g[p1 + "/gamma"], g[p1 + "/beta"], g[nm["w_dw"]], g[p0 + "/gamma"], g[p0 + "/beta"], da0,
                                        z0_blocked=B["z0b"], z1_blocked=True)
                wgrad_1x1(B["x_in"], da0, nm["w_exp"])
                stage1_next = expand_bwd_data(bi, da0, nm["w_exp"], tgt, tgt_has)
                if bi > 0:
                    has_grad[bi - 1] = True
                continue
            # bn1's backward apply inside the depthwise backward launch (its operands are staged there anyway; dz1 is never written)
            # (not the 5x5 stride-1 layer: that instantiation spills, measured without gain -- profiles/r03_notes.md)
            fuse_bn1 = bool(B["march"] and bn1_stage1 is not None and not (b.k == 5 and b.stride == 1) and
                            (b.expand != 1 or (bi == 0 and P.fuse_stem)))
            if not fuse_bn1:
                bn_b(B["z1"], da2, B["st1"], nm["bn1"], da2, post=True, chan_scale=B["gate"], chan_add=B["chan_add"], stage1=bn1_stage1)
            if fuse_bn1:
                st1, p1 = B["st1"], nm["bn1"]
                if b.expand != 1:
                    zin, st0, p0, dxo = B["z0"], B["st0"], nm["bn0"], B["da0"]
                else:
                    zin, st0, p0, dxo = P.z_stem, P.st_stem, self.n_stem[1], tgt
                nb1 = ops.mbconv_dw_bwd_march(da2, B["z1"], (st1[0], st1[1], w[p1 + "/gamma"], w[p1 + "/beta"]), B["gate"], B["chan_add"],
                                              P.stage1_se[:2 * N * ce].view(N, 2, ce), g[p1 + "/gamma"], g[p1 + "/beta"], zin,
                                              (st0[0], st0[1], w[p0 + "/gamma"], w[p0 + "/beta"]), w[nm["w_dw"]], b.stride, dxo,
                                              P.fold_part[nm["w_dw"]], P.stats_part2)
                if b.expand != 1:
                    bn_b(B["z0"], dxo, st0, p0, dxo, post=True, stage1=(P.stats_part2, nb1))
                    wgrad_1x1(B["x_in"], dxo, nm["w_exp"])
                    stage1_next = expand_bwd_data(bi, dxo, nm["w_exp"], tgt, tgt_has)
                else:
                    P.stem_stage1 = (P.stats_part2, nb1)
                if bi > 0:
                    has_grad[bi - 1] = True
                continue
            if B["march"]:
                # ONE pass over (dz1, z0): depthwise backward-data, filter-gradient slabs and stage 1 of bn0's backward
                wdw, slabs = w[nm["w_dw"]], P.fold_part[nm["w_dw"]]
                if b.expand != 1:
                    da0, st0, p0 = B["da0"], B["st0"], nm["bn0"]
                    _, _, nb1 = ops.dwconv_bn_bwd(da2, B["z0"], wdw, b.stride, bn=(st0[0], st0[1], w[p0 + "/gamma"], w[p0 + "/beta"]), out=da0,
                                                  dw_part=slabs, bn_part=P.stats_part2)
                    bn_b(B["z0"], da0, st0, p0, da0, post=True, stage1=(P.stats_part2, nb1))
                    wgrad_1x1(B["x_in"], da0, nm["w_exp"])
                    stage1_next = expand_bwd_data(bi, da0, nm["w_exp"], tgt, tgt_has)
                elif bi == 0 and P.fuse_stem:
                    # (the stem's BN + swish went into this block's depthwise launch: P.dstem = gradient w.r.t. the activated stem output)
                    st0, p0 = P.st_stem, self.n_stem[1]
                    _, _, nb1 = ops.dwconv_bn_bwd(da2, P.z_stem, wdw, b.stride, bn=(st0[0], st0[1], w[p0 + "/gamma"], w[p0 + "/beta"]), out=tgt,
                                                  dw_part=slabs, bn_part=P.stats_part2)
                    P.stem_stage1 = (P.stats_part2, nb1)
                elif tgt_has:   # no-expand block with identity skip (EfficientNet-B3 stage-1 repeats)
                    ops.dwconv_bn_bwd(da2, B["x_in"], wdw, b.stride, out=B["da0"], dw_part=slabs)
                    ops.chan_affine(B["da0"], out=tgt, accumulate=True)
                else:
                    ops.dwconv_bn_bwd(da2, B["x_in"], wdw, b.stride, out=tgt, dw_part=slabs)
                if bi > 0:
                    has_grad[bi - 1] = True
                continue
            dw_in = B["a0"] if b.expand != 1 else B["x_in"]
            ops.dwconv_bwd_filter(dw_in, da2, b.k, b.stride, partial=P.fold_part[nm["w_dw"]])
            if b.expand != 1:
                da0 = B["da0"]
                # the depthwise backward-data launch also emits stage 1 of bn0's backward (sums over (z0, da0)): no reduce pass
                st0 = B["st0"]
                _, nb1 = ops.dwconv_bwd_data(da2, w[nm["w_dw"]], b.stride, (b.h_in, b.h_in), out=da0, part=P.stats_part,
                                             bn=(B["z0"], st0[0], st0[1], w[nm["bn0"] + "/gamma"], w[nm["bn0"] + "/beta"]))
                bn_b(B["z0"], da0, st0, nm["bn0"], da0, post=True, stage1=(P.stats_part, nb1) if nb1 else None)
                wgrad_1x1(B["x_in"], da0, nm["w_exp"])
                stage1_next = expand_bwd_data(bi, da0, nm["w_exp"], tgt, tgt_has)
            else:
                if tgt_has:  # no-expand block with identity skip (EfficientNet-B3 stage-1 repeats)
                    tmp = B["da0"]
                    ops.dwconv_bwd_data(da2, w[nm["w_dw"]], b.stride, (b.h_in, b.h_in), out=tmp)
                    ops.chan_affine(tmp, out=tgt, accumulate=True)
                else:
                    ops.dwconv_bwd_data(da2, w[nm["w_dw"]], b.stride, (b.h_in, b.h_in), out=tgt)
            if bi > 0:
                has_grad[bi - 1] = True
        bn_b(P.z_stem, P.dstem, P.st_stem, self.n_stem[1], P.dstem, post=True, stage1=P.stem_stage1 if P.fuse_stem else None)
        ops.stem_conv_bwd_filter(x, P.dstem, idx, partial=P.fold_part[self.n_stem[0]])
        P.wbatch_ready = True
        P.wbatch.launch(self.matmul_precision)
        ops.se_wgrad_batched(P.se_desc, P.se_tiles)
        # all slabs written -> one batched fold into the gradient arena
        ops.fold_batched(P.fold_buf, A.grad, P.fold_desc, P.fold_tiles)

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
        if draw_masks and P.mask_plan is not None:
            ops.rng_masks(self.rng_state, P.mask_plan)
        logits = self._forward(P, self.shots_x, P.idx, True)
        ops.softmax_ce(logits, self.shots_y, P.idx, self.label_smoothing, self.dice, 0.0, want_grad=True, want_pred=False,
                       dlogits=P.dlogits, out=P.loss_out, ws=self.ws)
        if self.darc1:
            ops.darc1(logits, spec.L2_WEIGHT, dlogits=P.dlogits, out=P.loss_out, ws=self.ws)
        self._backward(P, self.shots_x, P.idx)
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
