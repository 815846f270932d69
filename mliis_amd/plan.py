"""The buffers of one inner step: every activation, gradient, statistics and scratch tensor of a fixed batch size N, allocated once in
HBM (`_Plan`), plus the launch tables that are built once per plan (deferred filter gradients, slab folds, SE weight gradients, the
mask draws) and the captured HIP graphs of the step.  Which kernel family runs a block (row-marching / small-map fused / op by op) is
decided here, from the C library's own eligibility queries."""
from __future__ import annotations

import torch

from . import ops, spec
from ._lib import MliisError, lib


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
            B["blk"] = 0
            if B["small"]:   # the fused small-map kernels save z0 (a copy) and z1 in their group-blocked layout for the backward launch
                B["z0b"] = xbuf(N, hi, hi, ce)
                # fp32 storage and both 1x1 convs beside the fused launches on the streamed plan: they WRITE the group-blocked layout
                # themselves (z0 by the expand conv: no copy; da2 by the project conv's backward-data) -- every access of the fused
                # launches to the expanded tensors but a1 / dz0 is then contiguous
                if (act_dtype == torch.float32 and b.expand != 1 and ops.conv1x1_stream_eligible(N, hi, hi, b.cin, ce, L.matmul_precision)
                        and ops.conv1x1_stream_eligible(N, ho, ho, b.cout, ce, L.matmul_precision) and 16 <= ho * ho <= 256):
                    B["blk"] = ops.mbconv_dw_small_group_width(ce, b.k)
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
        need = max(need, ops.bn_stats_partial_floats(N * hs * hs, a.stem_out), ops.stem_conv_fwd_stats_floats(N, H, H, a.stem_out))
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
        # project-BN-on-load (ops.conv2d_fwd_bnin): block i's project conv leaves its statistics in a THIRD buffer -- block i + 1's expand
        # conv folds them while its own workgroups already write the next batch norm's into stats_part.  bn2_deferred[i]: block i's
        # project batch norm (+ drop-connect, + identity skip) is applied by block i + 1's expand conv while it loads its rows
        # (training passes; inference keeps the stand-alone apply)
        ex_ = [b for b in a.blocks if b.executed]
        self.bn2_deferred = [bool(L.fuse_bn2 and i + 1 < len(ex_) and ex_[i + 1].expand != 1 and act_dtype in (torch.float32, torch.bfloat16) and
                                  ops.conv2d_fwd_bnin_ok(N, ex_[i].h_out, ex_[i].h_out, ex_[i].cout, ex_[i + 1].cexp)) for i in range(len(ex_))]
        self.stats_part3 = buf(max([-(-(N * b.h_out ** 2) // 16) * 2 * b.cout for b, d in zip(ex_, self.bn2_deferred) if d] + [0]) + 64)
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
