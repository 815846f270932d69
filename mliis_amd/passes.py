"""The two passes of an inner step as launch sequences over a plan's buffers (`_Plan`, plan.py): forward (stem -> MBConv blocks ->
[ASPP | DeepLabv3+-style decoder] -> residual skip decoders -> head) and backward, each a fixed sequence of C-ABI kernel launches
(ops.py) -- no torch op, no host decision that depends on data.  A mixin of `Learner` (learner.py), which owns the weights (arena),
the streams, graph capture and the checkpoint / meta-learner interface.

Graph semantics restated from: models/efficientlab.py:111-119,126-231,248-289,294-317; models/efficientnet/efficientnet_model.py:
175-290,396-441; models/efficientnet/utils.py:87-170.  Backward formulas: SURVEY.md Appendix B."""
from __future__ import annotations

import torch

from . import ops, spec
from ._lib import MliisError
from .plan import _Plan


class _Passes:
    # ------------------------------------------------------------------------------------------- forward
    X3_MIN_ROWS = 8192   # fp32x3 only on maps with at least this many pixels in the batch (32 row tiles of 256): below, the stream-K
                         # parts are a few chunks each and the native kernel wins (14x14 level: 31 against 20 us per launch)

    def _x3_takes(self, wname, xin) -> bool:
        return self.x3 is not None and self.x3.has(wname, "fwd") and xin.shape[0] * xin.shape[1] * xin.shape[2] >= self.X3_MIN_ROWS

    def _forward(self, P: _Plan, x, idx, training: bool, upsample: bool = True):
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

        # (the weight shadows of the step and -- in a training step that draws its masks on the device -- the masks: ONE launch)
        ops.transpose_weights(A.theta, self.theta_t, self.wt_desc, self.w_amax, tiles=self.wt_tiles, x3=self.x3, rng=getattr(self, "_rng_now", None))
        self._rng_now = None

        def conv(xin, wname, bname, dil, out, swish_stats, x_scale=None, border_bias=None, out_block=0, part=None, bnin=None):
            """dense conv; in training the epilogue also emits the following BN's statistics (returns their block count) into
            P.stats_part, or `part`.  bnin: the batch norm in front of this conv (the previous block's project BN, deferred: see the end
            of the block loop) applied while the conv loads its rows -- xin is then that batch norm's OUTPUT buffer, written by the launch."""
            am = self._amax_of.get(wname)
            if bnin is not None:
                st, prefix = bnin["st"], bnin["prefix"]
                return ops.conv2d_fwd_bnin(bnin["z"], P.stats_part3, bnin["nblk"], st[0], st[1], w[prefix + "/gamma"], w[prefix + "/beta"], xin,
                                           w[wname], out, moving=(mv[prefix + "/moving_mean"], mv[prefix + "/moving_variance"]),
                                           img_scale=bnin["img_scale"], res=bnin["res"], stats_part=P.stats_part, stats_swish=swish_stats,
                                           wt=self.wt[wname], precision=self.matmul_precision, fp8_w_amax=am, out_block=out_block)[1]
            if self._x3_takes(wname, xin):   # (fp32x3: a long-K decoder conv on a map large enough to fill the chip)
                k_, co_ = w[wname].shape[0], w[wname].shape[3]
                if training:
                    return ops.conv2d_fwd_x3(xin, self.x3.image(wname, "fwd"), k_, co_, w[bname] if bname else None, dil, out=out, ws=ws,
                                             stats_part=P.stats_part, stats_swish=swish_stats, border_bias=border_bias)[1]
                ops.conv2d_fwd_x3(xin, self.x3.image(wname, "fwd"), k_, co_, w[bname] if bname else None, dil, out=out, ws=ws,
                                  border_bias=border_bias)
                return 0
            if training:
                return self._conv_fwd(xin, w[wname], w[bname] if bname else None, dil, out=out, ws=ws, stats_part=P.stats_part if part is None else part,
                                      stats_swish=swish_stats, wt=self.wt[wname], x_scale=x_scale, border_bias=border_bias, fp8_w_amax=am,
                                      out_block=out_block)[1]
            self._conv_fwd(xin, w[wname], w[bname] if bname else None, dil, out=out, ws=ws, wt=self.wt[wname], x_scale=x_scale,
                           border_bias=border_bias, fp8_w_amax=am)
            return 0

        # training: the stem conv's launch also leaves the stage-1 statistics of its batch norm (row-strip kernel; -1: the rows are too
        # wide for it -- the plain kernel ran and the consumer takes the statistics launch); MLIIS_STEM_STATS=0: always the two launches
        nb_stem = 0
        if training and self.stem_stats:
            nb_stem = max(0, ops.stem_conv_fwd(x, w[self.n_stem[0]], idx, out=P.z_stem, stats_part=P.stats_part)[1])
        else:
            ops.stem_conv_fwd(x, w[self.n_stem[0]], idx, out=P.z_stem)
        ex = [b for b in a.blocks if b.executed]
        fuse_stem = P.fuse_stem   # (block 0 takes the stem's BN + swish into its depthwise launch: _Plan)
        if fuse_stem:
            cur = None
        else:
            cur = bn(P.z_stem, P.st_stem, self.n_stem[1], P.a_stem, post=True, nblk=nb_stem)

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

        pend = None   # the previous block's project batch norm, when this block's expand conv applies it on load (P.bn2_deferred)

        def block_end(bi_, b, B, nm, nb, use_dc):
            """The block's project batch norm (+ drop-connect scale, + identity skip): a launch of its own, or -- training, the next
            block's expand conv on the streamed plan -- handed to that conv, which forms the block output while it loads its rows
            and writes it to B["out"] (ops.conv2d_fwd_bnin: one launch and one pass over z2 less per block)."""
            nonlocal pend
            img_scale, res = (B["dc"] if use_dc else None), (B["x_in"] if b.skip else None)
            if training and P.bn2_deferred[bi_]:
                if nb == 0:
                    nb = ops.bn_stats_partial(B["z2"], False, P.stats_part3)
                pend = dict(z=B["z2"], nblk=nb, st=B["st2"], prefix=nm["bn2"], img_scale=img_scale, res=res)
                return B["out"]
            return bn(B["z2"], B["st2"], nm["bn2"], B["out"], img_scale=img_scale, res=res, nblk=nb)

        for bi_, (b, B, nm) in enumerate(zip(ex, P.blocks, self.n_blocks)):
            B["x_in"] = cur
            t = cur
            bnin, pend = pend, None
            defer = training and P.bn2_deferred[bi_]
            if training and B["small"]:
                # expand GEMM (+ stage-1 statistics) -> ONE launch: bn0 fold + apply + swish, depthwise, bn1 statistics + apply + swish,
                # squeeze-excite means, both moving averages -> SE MLP -> project GEMM
                z0 = B["z0b"] if B["blk"] else B["z0"]     # (blk: the expand conv writes the group-blocked layout itself)
                nb = conv(t, nm["w_exp"], None, 1, z0, False, out_block=B["blk"], bnin=bnin)
                if nb == 0:
                    if B["blk"]:
                        raise MliisError("internal: the streamed expand conv of block {} left no statistics".format(b.idx))
                    nb = ops.bn_stats_partial(z0, False, P.stats_part)
                p0, p1 = nm["bn0"], nm["bn1"]
                ops.mbconv_dw_fwd_small(z0, P.stats_part, nb,
                                        (w[p0 + "/gamma"], w[p0 + "/beta"], B["st0"][0], B["st0"][1], mv[p0 + "/moving_mean"], mv[p0 + "/moving_variance"]),
                                        w[nm["w_dw"]],
                                        (w[p1 + "/gamma"], w[p1 + "/beta"], B["st1"][0], B["st1"][1], mv[p1 + "/moving_mean"], mv[p1 + "/moving_variance"]),
                                        B["z1"], B["a1"], B["s"], z0_blocked=B["z0b"], z1_blocked=True)   # (the backward's re-reads: contiguous)
                se = nm["se"]
                ops.se_mlp_fwd(B["s"], w[se[0]], w[se[1]], w[se[2]], w[se[3]], B["hpre"], B["gate"])
                nb = conv(B["a1"], nm["w_proj"], None, 1, B["z2"], False, x_scale=B["gate"], part=P.stats_part3 if defer else None)
                use_dc = self.drop_connect and b.skip and b.drop_rate > 0
                B["use_dc"] = use_dc
                cur = block_end(bi_, b, B, nm, nb, use_dc)
                continue
            if B["march"]:
                # expand GEMM (+ stage-1 statistics) -> ONE launch: bn0 fold + apply + swish while the rows are staged, depthwise conv,
                # bn1 stage-1 statistics (P.stats_part2)
                if b.expand != 1:
                    nb = conv(t, nm["w_exp"], None, 1, B["z0"], False, bnin=bnin)
                    bn0, nb = bn_in(B["z0"], B["st0"], nm["bn0"], nb)
                    zin = B["z0"]
                elif bi_ == 0 and fuse_stem:
                    bn0, nb = bn_in(P.z_stem, P.st_stem, self.n_stem[1], nb_stem)
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
                    nb = conv(t, nm["w_exp"], None, 1, B["z0"], False, bnin=bnin)
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
            nb = conv(B["a1"], nm["w_proj"], None, 1, B["z2"], False, x_scale=B["gate"], part=P.stats_part3 if defer else None)
            use_dc = training and self.drop_connect and b.skip and b.drop_rate > 0
            B["use_dc"] = use_dc
            cur = block_end(bi_, b, B, nm, nb, use_dc)
        if pend is not None:
            raise MliisError("internal: a deferred project batch norm was not consumed")
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
                if self._x3_takes(k1, cat):
                    nb1 = ops.conv2d_fwd_x3(cat, self.x3.image(k1, "fwd"), 3, m.c_out, w[b1], 2, out=D["z1"], ws=ws, stats_part=P.stats_part2,
                                            stats_swish=True)[1]
                else:
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
        if not upsample:   # (the fused head launch reads the decoder-resolution logits: Learner._train_sequence)
            return None
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
    def _backward(self, P: _Plan, x, idx, head_fused: bool = False):
        A, a, ws, N = self.arena, self.arch, self.ws, P.N
        w, g = A.w, A.g
        hd = a.h_dec
        if not head_fused:   # (ops.head_ce_fused left the gradient on the decoder's map already)
            ops.resize_bilinear_bwd(P.dlogits, (hd, hd), out=P.dsmall)
        mask = P.drop_mask
        ops.final_conv_bwd_filter(P.dec_in, P.dsmall, mask, dw=g[self.n_final[0]], db=g[self.n_final[1]], ws=ws)
        ex = [b for b in a.blocks if b.executed]
        has_grad = [False] * len(P.blocks)
        dtop = P.rsd[-1]["dout"] if P.rsd else (P.skipdec["dout"] if a.skipdec is not None else
                                                 (P.aspp["dout"] if a.aspp else P.blocks[-1]["dout"]))
        ops.final_conv_bwd_data(P.dsmall, w[self.n_final[0]], a.c_final, mask, out=dtop, fin=getattr(P, "head_fin", None) if head_fused else None)
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
            capped grids beside the encoder's backward chain: neutral to negative.  Round 5 ran it as a graph of its own on a stream
            masked to 64-96 CUs beside the chain on the other 160-192: the masks hold, the chain slows by what the move saves --
            profiles/r05_notes.md, tools/cumask_probe.py -- removed again.)"""
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
            if self._x3_takes(kf, D["dzf"]):
                ops.conv2d_bwd_data_x3(D["dzf"], self.x3.image(kf, "bwd"), 3, 2 * co, 1, out=dpyr, ws=ws)
            else:
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
            if self._x3_takes(k1, d1):
                ops.conv2d_bwd_data_x3(d1, self.x3.image(k1, "bwd"), 3, m.c_cat, 2, out=dcat, accumulate=True, ws=ws)
            else:
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
        hook = getattr(self, "_segment_hook", None)   # (tools/cumask_probe.py cuts the pass into separately captured graphs here)
        if hook is not None:
            hook("encoder_backward")

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
                _, groups = self._conv_bwd_data(dout, w[nm["w_proj"]], 1, out=da2, ws=ws, gate=B["a1"], part=P.gate_part,
                                                out_block=B["blk"] if B["small"] else 0)
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
                    if B["small"] and B["blk"]:   # (da2 is group-blocked here, a1 is not: their product needs the launch's own partial sums)
                        raise MliisError("internal: the project backward-data launch of block {} left no gate-gradient partials for its "
                                         "group-blocked output".format(b.idx))
                    ops.colsum(da2, B["a1"], nseg=N, out=B["dgate"], ws=ws)
                # (the SE weight gradients of all blocks are computed by one batched launch after the loop: P.se_desc)
                ops.se_mlp_bwd(P.gate_part if groups else B["dgate"], B["gate"], B["s"], B["hpre"], w[se[0]], w[se[2]], hw, se_outs,
                               dgate_groups=groups, w1t=self.wt[se[0]])
            if B["small"]:   # bn1 backward, depthwise filter gradient + backward-data, bn0 backward: one launch
                da0, st0, st1, p0, p1 = B["da0"], B["st0"], B["st1"], nm["bn0"], nm["bn1"]
                ops.mbconv_dw_bwd_small(da2, B["gate"], B["chan_add"], B["z1"], (st1[0], st1[1], w[p1 + "/gamma"], w[p1 + "/beta"]),
                                        w[nm["w_dw"]], B["z0"], (st0[0], st0[1], w[p0 + "/gamma"], w[p0 + "/beta"]),
                                        g[p1 + "/gamma"], g[p1 + "/beta"], g[nm["w_dw"]], g[p0 + "/gamma"], g[p0 + "/beta"], da0,
                                        z0_blocked=B["z0b"], z1_blocked=True, da2_blocked=bool(B["blk"]))
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
        if hook is not None:
            hook("filter_gradients")
        P.wbatch.launch("fp32x3" if self.x3 is not None else self.matmul_precision)
        if hook is not None:
            hook("fold")
        # all slabs written -> one batched fold into the gradient arena; the squeeze-excite weight gradients of every block ride in it
        ops.fold_batched(P.fold_buf, A.grad, P.fold_desc, P.fold_tiles, se_desc=P.se_desc, se_tiles=P.se_tiles)
