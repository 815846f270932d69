"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product package (mliis_amd/).

CPU restatement (PyTorch-CPU, float64 by default, autograd supplies every backward) of the reference's
EfficientLab graph and one inner optimisation step.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.

PARITY STATUS: **parity unpinned** for the numeric graph.  The reference computes this path inside
tensorflow==1.15.4 (requirements.txt:1), which is neither vendored under /root/reference nor installable here
(Python 3.10, no network), and the reference ships no tests / golden vectors for it.  This file
restates the graph from the reference's graph-construction code and TF's documented op semantics; it is anchored by
hand-derivable known-answer tests (tests/test_oracle_kat.py).  The *host logic* oracle (oracle/host_ref.py) IS pinned
against values produced by importing the reference's own numpy code (tests/golden/host_logic.json).

Reference lines followed:
  graph         models/efficientlab.py:111-119 (normalise), :126-177 (decode), :179-231 (RSD), :294-327 (loss/opt)
  backbone      models/efficientnet/efficientnet_model.py:170-290 (MBConv), :326-371,396-441 (stem/blocks/endpoints)
  architecture  models/efficientnet/efficientnet_builder.py:29-42,90-109,125-149
  BN/dropconn   models/efficientnet/utils.py:87-134,157-170
  L2            models/regularizers.py:4-10
  init          models/efficientnet/efficientnet_model.py:61-82; TF glorot_uniform default for tf.layers.conv2d
TF semantics restated from documentation: SAME padding, non-fused BN (biased variance everywhere; moving stats
`m -= (m - stat) * (1 - 0.99)`), fused BN in the decoder (unbiased variance into the moving average), bilinear
resize with align_corners=True, tf.losses.softmax_cross_entropy (mean over rows), GradientDescentOptimizer.
"""
from __future__ import annotations

import math
import re
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

MEAN = [0.485 * 255, 0.456 * 255, 0.406 * 255]
STD = [0.229 * 255, 0.224 * 255, 0.225 * 255]
EPS = 1e-3
MOMENTUM = 0.99

_B0_NOTATION = "r1k3s1e1i32o16 r2k3s2e6i16o24 r2k5s2e6i24o40 r3k3s2e6i40o80 r3k5s1e6i80o112 r4k5s2e6i112o192 r1k3s1e6i192o320"
_SCALE = {"efficientnet-b0": (1.0, 1.0, 10, 112), "efficientnet-b3": (1.2, 1.4, 17, 136)}


# ------------------------------------------------------------------------------------------------ architecture
def _rf(f, w):
    f2 = f * w
    n = max(8, int(f2 + 4) // 8 * 8)
    return int(n + 8 if n < 0.9 * f2 else n)


def arch(name="efficientnet-b0", image_size=224, rsd=(2, 4), aspp=False, skip_decoding=False):
    w, d, max_block, dec_c = _SCALE[name]
    blocks, cum = [], 0
    for tok in _B0_NOTATION.split():
        r, k, s, e, i, o = map(int, re.match(r"r(\d+)k(\d+)s(\d+)e(\d+)i(\d+)o(\d+)", tok).groups())
        cum += r
        if cum > max_block + 1:
            break
        i, o, r = _rf(i, w), _rf(o, w), int(math.ceil(d * r))
        for j in range(r):
            blocks.append(dict(k=k, s=s if j == 0 else 1, e=e, i=i if j == 0 else o, o=o))
    n = len(blocks)
    red, endpoints = 0, {}
    for j, b in enumerate(blocks):
        b["idx"] = j
        b["se"] = max(1, int(b["i"] * 0.25))
        b["drop"] = 0.2 * j / n
        if j == n - 1 or blocks[j + 1]["s"] > 1:
            red += 1
            endpoints[red] = j
    return dict(name=name, blocks=blocks, endpoints=endpoints, dec_c=dec_c, rsd=sorted(rsd or [], reverse=True),
                stem=_rf(32, w), last=endpoints[4], image_size=image_size, aspp=bool(aspp), skipdec=bool(skip_decoding))


def param_specs(a) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(name, shape, init) of all TRAINABLE variables in creation order, then names of BN layers."""
    fe = a["name"]
    out = [(f"{fe}/stem/conv2d/kernel", (3, 3, 3, a["stem"]), "normal")]
    out += [(f"{fe}/stem/tpu_batch_normalization/gamma", (a["stem"],), "ones"),
            (f"{fe}/stem/tpu_batch_normalization/beta", (a["stem"],), "zeros")]
    for b in a["blocks"]:
        s, ce = f"{fe}/blocks_{b['idx']}", b["i"] * b["e"]
        bn = ["tpu_batch_normalization", "tpu_batch_normalization_1", "tpu_batch_normalization_2"]
        cv = ["conv2d", "conv2d_1"]
        if b["e"] != 1:
            out += [(f"{s}/{cv.pop(0)}/kernel", (1, 1, b["i"], ce), "normal")]
            n = bn.pop(0)
            out += [(f"{s}/{n}/gamma", (ce,), "ones"), (f"{s}/{n}/beta", (ce,), "zeros")]
        out += [(f"{s}/depthwise_conv2d/depthwise_kernel", (b["k"], b["k"], ce, 1), "normal")]
        n = bn.pop(0)
        out += [(f"{s}/{n}/gamma", (ce,), "ones"), (f"{s}/{n}/beta", (ce,), "zeros")]
        out += [(f"{s}/se/conv2d/kernel", (1, 1, ce, b["se"]), "normal"), (f"{s}/se/conv2d/bias", (b["se"],), "zeros"),
                (f"{s}/se/conv2d_1/kernel", (1, 1, b["se"], ce), "normal"), (f"{s}/se/conv2d_1/bias", (ce,), "zeros")]
        out += [(f"{s}/{cv.pop(0)}/kernel", (1, 1, ce, b["o"]), "normal")]
        n = bn.pop(0)
        out += [(f"{s}/{n}/gamma", (b["o"],), "ones"), (f"{s}/{n}/beta", (b["o"],), "zeros")]
    deep = a["blocks"][a["last"]]["o"]
    if a.get("aspp"):   # models/efficientlab.py:248-289 (tf.layers.conv2d defaults: glorot-uniform kernel, zero bias)
        s, d = "decode/spatial_pyramid_pooling", a["dec_c"]
        for scope, k, ci in ((f"{s}/branch_0", 1, deep), (f"{s}/branch_1", 3, deep), (f"{s}/branch_2", 1, deep), (s, 1, 3 * d)):
            out += [(f"{scope}/conv2d/kernel", (k, k, ci, d), "glorot"), (f"{scope}/conv2d/bias", (d,), "zeros")]
        deep = d
    if a.get("skipdec"):   # models/efficientlab.py:133-149, 445-474 (DeepLabv3+-style decoder, --skip_decoding)
        s = "decode/decode_skip_connections"
        c2, csk = a["blocks"][a["endpoints"][2]]["o"], a["dec_c"] // 2
        csep = a["dec_c"] + csk
        out += [(f"{s}/conv2d/kernel", (1, 1, c2, csk), "glorot"),
                (f"{s}/batch_normalization/gamma", (csk,), "ones"), (f"{s}/batch_normalization/beta", (csk,), "zeros")]
        cin = deep + csk
        for j in range(2):
            dwn = "depthwise_conv2d" + ("" if j == 0 else f"_{j}")
            out += [(f"{s}/{dwn}/depthwise_kernel", (3, 3, cin, 1), "normal"),
                    (f"{s}/batch_normalization_{2 * j + 1}/gamma", (cin,), "ones"), (f"{s}/batch_normalization_{2 * j + 1}/beta", (cin,), "zeros"),
                    (f"{s}/conv2d_{j + 1}/kernel", (1, 1, cin, csep), "normal"),
                    (f"{s}/batch_normalization_{2 * j + 2}/gamma", (csep,), "ones"), (f"{s}/batch_normalization_{2 * j + 2}/beta", (csep,), "zeros")]
            cin = csep
        deep = csep
    for r in a["rsd"]:
        s = f"decode/decode_skip_connections_{r - 1}"
        cs = a["blocks"][a["endpoints"][r]]["o"]
        cc, co = deep + cs, a["dec_c"]
        convs = [(1, cc), (3, cc), (3, 2 * co + cc)]
        if deep != co:   # "Increasing upsampled skip connection number of filters with 1x1 conv" (efficientlab.py:213-215): created first
            convs = [(1, deep)] + convs
        for j, (k, ci) in enumerate(convs):
            sfx = "" if j == 0 else f"_{j}"
            out += [(f"{s}/conv2d{sfx}/kernel", (k, k, ci, co), "glorot"), (f"{s}/conv2d{sfx}/bias", (co,), "zeros"),
                    (f"{s}/batch_normalization{sfx}/gamma", (co,), "ones"),
                    (f"{s}/batch_normalization{sfx}/beta", (co,), "zeros")]
        deep = co
    out += [("decode/final_layer_weights/kernel", (1, 1, deep, 2), "normal"),
            ("decode/final_layer_weights/bias", (2,), "zeros")]
    return out


def init_state(a, seed=0, dtype=torch.float64):
    """Returns (params: ordered dict name->tensor, bn: dict bn_prefix->(moving_mean, moving_var))."""
    g = np.random.default_rng(seed)
    params, bn = {}, {}
    for name, shape, init in param_specs(a):
        if init == "normal":     # conv_kernel_initializer: N(0, sqrt(2 / (kh*kw*out)))
            v = g.standard_normal(shape) * math.sqrt(2.0 / (shape[0] * shape[1] * shape[3]))
        elif init == "glorot":   # glorot_uniform: U(-l, l), l = sqrt(6/(fan_in+fan_out)), receptive field included
            rf = shape[0] * shape[1]
            lim = math.sqrt(6.0 / (rf * shape[2] + rf * shape[3]))
            v = g.uniform(-lim, lim, shape)
        elif init == "ones":
            v = np.ones(shape)
        else:
            v = np.zeros(shape)
        params[name] = torch.tensor(v, dtype=dtype)
        if name.endswith("/gamma"):
            p = name[: -len("/gamma")]
            bn[p] = (torch.zeros(shape, dtype=dtype), torch.ones(shape, dtype=dtype))
    return params, bn


# ------------------------------------------------------------------------------------------------------ ops
def same_pad_amounts(size, k, s, d=1):
    out = -(-size // s)
    tot = max((out - 1) * s + (k - 1) * d + 1 - size, 0)
    return tot // 2, tot - tot // 2


def conv2d_same(x, w_hwio, stride=1, dilation=1, bias=None, groups=1):
    """x: NCHW; w: TF HWIO (depthwise: HW,C,1 with groups=C)."""
    kh, kw = w_hwio.shape[:2]
    pt, pb = same_pad_amounts(x.shape[2], kh, stride, dilation)
    pl, pr = same_pad_amounts(x.shape[3], kw, stride, dilation)
    x = F.pad(x, (pl, pr, pt, pb))
    if groups == 1:
        w = w_hwio.permute(3, 2, 0, 1)
    else:
        w = w_hwio.permute(2, 3, 0, 1)  # (C,1,kh,kw)
    return F.conv2d(x, w, bias, stride=stride, dilation=dilation, groups=groups)


def swish(x):
    return x * torch.sigmoid(x)


# ---- reduced-precision matrix-core operands (BASELINE configs 4-5), emulated: the product converts the fp32 operands of every dense
#      conv that runs on the matrix cores (MBConv 1x1 expand / project, RSD and ASPP convs) in registers -- bf16: round to nearest
#      even; fp8 (forward 1x1 convs only): OCP e4m3 after a power-of-two scale, saturating at +-448 -- and accumulates in fp32.  The
#      backward passes use bf16-rounded operands in both modes: dX = convT(round(dY), round(W)), dW = corr(round(X), round(dY)).
FP8_ACT_SCALE = 16.0


def round_bf16(t):
    return t.to(torch.float32).to(torch.bfloat16).to(t.dtype)


def fp8_weight_scale(w):
    amax = float(w.detach().abs().max())
    return 2.0 ** math.floor(math.log2(224.0 / amax)) if amax > 0 else 1.0


def round_fp8(t, scale):
    q = (t.to(torch.float32) * scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).to(torch.float32) / scale
    return q.to(t.dtype)


class _RoundedConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, stride, dilation, mode):
        k = w.shape[0]
        if mode == "fp8" and k == 1:
            xr, wr = round_fp8(x, FP8_ACT_SCALE), round_fp8(w, fp8_weight_scale(w))
        else:
            xr, wr = round_bf16(x), round_bf16(w)
        ctx.save_for_backward(round_bf16(x), round_bf16(w))
        ctx.geom = (stride, dilation)
        return conv2d_same(xr, wr, stride, dilation)

    @staticmethod
    def backward(ctx, dy):
        xb, wb = ctx.saved_tensors
        stride, dilation = ctx.geom
        dyr = round_bf16(dy)
        with torch.enable_grad():
            x_, w_ = xb.detach().requires_grad_(True), wb.detach().requires_grad_(True)
            y = conv2d_same(x_, w_, stride, dilation)
            gx, gw = torch.autograd.grad(y, [x_, w_], dyr)
        return gx, gw, None, None, None


# ---- bf16 STORAGE points (`--precision bf16-storage`, BASELINE configs[3]): the product keeps the expanded tensors of an MBConv block
#      (z0, z1, a1) and their gradients as bf16 in HBM.  A tensor that is written and read back is rounded to nearest even where it is
#      written: in the forward pass (`fwd`), in the backward pass (its gradient, `bwd`), or both.  Which gradients exist as tensors
#      depends on the kernel family that runs the block (`store(block)` -> "march" | "small" | None, given by the test from the
#      product's plan): the fused families form some of them in registers only.
class _StorePoint(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return round_bf16(x) if fwd else x.clone()

    @staticmethod
    def backward(ctx, g):
        return (round_bf16(g) if ctx.bwd else g), None, None


def store_point(x, fwd=True, bwd=True):
    return _StorePoint.apply(x, fwd, bwd)


def mm_conv(x, w_hwio, stride=1, dilation=1, bias=None, round_ops=None):
    """A dense conv that the product runs on the matrix cores: exact, or with emulated reduced-precision operands."""
    if round_ops is None:
        return conv2d_same(x, w_hwio, stride, dilation, bias=bias)
    y = _RoundedConv.apply(x, w_hwio, stride, dilation, round_ops)
    return y if bias is None else y + bias[None, :, None, None]


def batch_norm(x, gamma, beta, moving, training, new_moving: Optional[dict], key, fused):
    """x: NCHW.  training: batch stats (biased var for normalisation).  Records the EMA target in new_moving."""
    if training:
        mean = x.mean(dim=(0, 2, 3))
        var = ((x - mean[None, :, None, None]) ** 2).mean(dim=(0, 2, 3))
        if new_moving is not None:
            n = x.numel() // x.shape[1]
            v_ema = var * (n / (n - 1.0)) if fused else var
            mm, mv = moving
            d = 1.0 - MOMENTUM
            new_moving[key] = ((mm - (mm - mean.detach()) * d), (mv - (mv - v_ema.detach()) * d))
    else:
        mean, var = moving
    inv = torch.rsqrt(var + EPS)
    return (x - mean[None, :, None, None]) * (inv * gamma)[None, :, None, None] + beta[None, :, None, None]


def resize_bilinear_ac(x, size):
    if tuple(x.shape[2:]) == tuple(size):
        return x
    return F.interpolate(x, size=size, mode="bilinear", align_corners=True)


# -------------------------------------------------------------------------------------------------- forward
def forward(a, params, bn, x_nhwc, training=True, dc_scales: Optional[Dict[int, torch.Tensor]] = None,
            dropout_mask: Optional[torch.Tensor] = None, taps: Optional[dict] = None, aspp_masks=None, round_ops=None, store=None):
    """x_nhwc: [N,H,W,3] in 0..255.  dc_scales[block_idx]: [N] tensor of 0 or 1/keep (training only; None -> no
    drop-connect).  dropout_mask: [N,h,w,C] of 0 or 1/(1-rate) applied before the final 1x1.  Returns
    (logits NHWC, new_moving dict).  `taps` (optional dict) receives named intermediates in NHWC."""
    fe = a["name"]
    P = params
    new_moving = {} if training else None
    dt = P[f"{fe}/stem/conv2d/kernel"].dtype
    x = x_nhwc.to(dt)
    x = (x - torch.tensor(MEAN, dtype=dt)) / torch.tensor(STD, dtype=dt)
    x = x.permute(0, 3, 1, 2)

    def BN(t, prefix, fused=False):
        return batch_norm(t, P[prefix + "/gamma"], P[prefix + "/beta"], bn[prefix], training, new_moving, prefix, fused)

    def tap(name, t):
        if taps is not None:
            taps[name] = t.permute(0, 2, 3, 1)

    x = swish(BN(conv2d_same(x, P[f"{fe}/stem/conv2d/kernel"], 2), f"{fe}/stem/tpu_batch_normalization"))
    tap("stem", x)
    ends = {}
    inv_end = {v: k for k, v in a["endpoints"].items()}
    for b in a["blocks"][: a["last"] + 1]:
        s = f"{fe}/blocks_{b['idx']}"
        bns = [f"{s}/tpu_batch_normalization", f"{s}/tpu_batch_normalization_1", f"{s}/tpu_batch_normalization_2"]
        cvs = [f"{s}/conv2d/kernel", f"{s}/conv2d_1/kernel"]
        inp = x
        fam = store(b) if (store is not None and training) else None   # bf16 storage points of this block (None: fp32 tensors)
        if b["e"] != 1:
            z0 = mm_conv(x, P[cvs.pop(0)], round_ops=round_ops)
            if fam:   # z0 and its gradient dz0 are bf16 tensors
                z0 = store_point(z0)
            x = swish(BN(z0, bns.pop(0)))
            if fam == "march":   # the marching backward writes da0 (the gradient of the activation), the bn0 backward apply reads it
                x = store_point(x, fwd=False, bwd=True)
        ce = x.shape[1]
        z1 = conv2d_same(x, P[f"{s}/depthwise_conv2d/depthwise_kernel"], b["s"], groups=ce)
        if fam:   # z1 is a bf16 tensor; its gradient dz1 exists as a tensor only where the depthwise batch norm's backward apply is a
            #       launch of its own (the 5x5 stride-1 marching layers) -- the fused forms keep it in registers
            z1 = store_point(z1, bwd=(fam == "march" and b["k"] == 5 and b["s"] == 1))
        x = swish(BN(z1, bns.pop(0)))
        if fam:   # a1 is a bf16 tensor (its gradient is formed in registers from da2)
            x = store_point(x, bwd=False)
        sq = x.mean(dim=(2, 3), keepdim=True)
        sq = swish(conv2d_same(sq, P[f"{s}/se/conv2d/kernel"], bias=P[f"{s}/se/conv2d/bias"]))
        sq = conv2d_same(sq, P[f"{s}/se/conv2d_1/kernel"], bias=P[f"{s}/se/conv2d_1/bias"])
        x = torch.sigmoid(sq) * x
        if fam:   # da2, the gradient of the gated activation (the project conv's backward-data output), is a bf16 tensor
            x = store_point(x, fwd=False, bwd=True)
        x = BN(mm_conv(x, P[cvs.pop(0)], round_ops=round_ops), bns.pop(0))
        if b["s"] == 1 and b["i"] == b["o"]:
            if training and b["drop"] and dc_scales is not None and b["idx"] in dc_scales:
                x = x * dc_scales[b["idx"]].to(dt)[:, None, None, None]
            x = x + inp
        tap(f"block_{b['idx']}", x)
        if b["idx"] in inv_end:
            ends[inv_end[b["idx"]]] = x
    dec = ends[4]
    if a.get("aspp"):
        # Atrous spatial pyramid pooling, models/efficientlab.py:248-289: 1x1 / 3x3-dilation-6 / image-pooling branches (swish, dropout
        # 0.5 -- the pooled branch drops BEFORE its swish), concat [pooled, 3x3, 1x1] -> 1x1 conv -> swish -> dropout.  aspp_masks: the
        # four dropout scale tensors (0 or 2) in NHWC for branch_0, branch_1, branch_2 ([N,1,1,C]) and the output (training only).
        s = "decode/spatial_pyramid_pooling"
        mk = [None] * 4 if (aspp_masks is None or not training) else [m.to(dt).permute(0, 3, 1, 2) for m in aspp_masks]

        def drop(t, m):
            return t if m is None else t * m
        cv = lambda t, scope, d=1: mm_conv(t, P[f"{scope}/conv2d/kernel"], 1, d, bias=P[f"{scope}/conv2d/bias"], round_ops=round_ops)  # noqa: E731
        b0 = drop(swish(cv(dec, f"{s}/branch_0")), mk[0])
        b1 = drop(swish(cv(dec, f"{s}/branch_1", 6)), mk[1])
        b2 = swish(drop(cv(dec.mean(dim=(2, 3), keepdim=True), f"{s}/branch_2"), mk[2]))
        b2 = b2.expand(-1, -1, dec.shape[2], dec.shape[3])     # bilinear align_corners resize of a 1x1 map = broadcast
        dec = drop(swish(cv(torch.cat([b2, b1, b0], dim=1), s)), mk[3])
        tap("aspp", dec)
    if a.get("skipdec"):
        # efficientlab.py:133-149: resize to input // 4, concat with swish(BN(conv1x1(reduction_2 endpoint))), two sep_convs
        # (depthwise 3x3 -> BN -> swish -> 1x1 -> BN -> swish, :445-474).  These batch norms are built with training=True: batch
        # statistics in inference too (their moving averages only move when the train op's update ops run).
        s = "decode/decode_skip_connections"

        def BNT(t, prefix):
            return batch_norm(t, P[prefix + "/gamma"], P[prefix + "/beta"], bn[prefix], True, new_moving, prefix, True)
        sk = ends[2]
        up4 = resize_bilinear_ac(dec, sk.shape[2:])
        dsk = swish(BNT(mm_conv(sk, P[f"{s}/conv2d/kernel"], round_ops=round_ops), f"{s}/batch_normalization"))
        dec = torch.cat([up4, dsk], dim=1)
        for j in range(2):
            dwn = "depthwise_conv2d" + ("" if j == 0 else f"_{j}")
            dec = swish(BNT(conv2d_same(dec, P[f"{s}/{dwn}/depthwise_kernel"], 1, groups=dec.shape[1]), f"{s}/batch_normalization_{2 * j + 1}"))
            dec = swish(BNT(mm_conv(dec, P[f"{s}/conv2d_{j + 1}/kernel"], round_ops=round_ops), f"{s}/batch_normalization_{2 * j + 2}"))
        tap("skipdec", dec)
    for r in a["rsd"]:
        s = f"decode/decode_skip_connections_{r - 1}"
        skip = ends[r]
        up = resize_bilinear_ac(dec, skip.shape[2:])
        cat = torch.cat([up, skip], dim=1)
        j0 = [0]

        def branch(t, k, d):
            sfx = "" if j0[0] == 0 else f"_{j0[0]}"
            j0[0] += 1
            t = mm_conv(t, P[f"{s}/conv2d{sfx}/kernel"], 1, d, bias=P[f"{s}/conv2d{sfx}/bias"], round_ops=round_ops)
            return BN(swish(t), f"{s}/batch_normalization{sfx}", fused=True)
        if up.shape[1] != a["dec_c"]:   # efficientlab.py:213-215: the residual operand gets its own 1x1 branch; the concat keeps `up`
            up = branch(up, 1, 1)
        b0 = branch(cat, 1, 1)
        b1 = branch(cat, 3, 2)
        b2 = cat.mean(dim=(2, 3), keepdim=True).expand_as(cat)
        dec = branch(torch.cat([b0, b1, b2], dim=1), 3, 1) + up
        tap(f"rsd_{r}", dec)
    if dropout_mask is not None and training:
        dec = dec * dropout_mask.to(dt).permute(0, 3, 1, 2)
    dec = conv2d_same(dec, P["decode/final_layer_weights/kernel"], bias=P["decode/final_layer_weights/bias"])
    tap("final_small", dec)
    H = x_nhwc.shape[1]
    logits = resize_bilinear_ac(dec, (H, x_nhwc.shape[2]))
    return logits.permute(0, 2, 3, 1), new_moving


def loss_fn(a, params, logits, labels, label_smoothing=0.0, dice=False, l2=False, l1=False, darc1=False):
    """models/efficientlab.py:294-313.  logits/labels NHWC [N,H,W,2]."""
    t = labels.to(logits.dtype)
    if label_smoothing:
        t = t * (1 - label_smoothing) + label_smoothing / 2
    logp = F.log_softmax(logits, dim=-1)
    # tf.losses.softmax_cross_entropy: weights=1, SUM_BY_NONZERO_WEIGHTS -> mean over rows
    loss = -(t * logp).sum(-1).mean()
    if dice:
        p1 = torch.softmax(logits, dim=-1)[..., 1].flatten(1)
        t1 = labels.to(logits.dtype)[..., 1].flatten(1)
        inter = (p1 * t1).sum(1)
        den = p1.sum(1) + t1.sum(1) - inter
        iou = ((inter + 1e-7) / (den + 1e-7)).mean()
        loss = loss - torch.log(2 * iou / (iou + 1))
    if l2:
        loss = loss + 0.0005 * sum(0.5 * (v ** 2).sum() for k, v in params.items() if "batch_normalization" not in k)
    if darc1:   # models/regularizers.py:20-22 (the reference adds it first, efficientlab.py:304-306)
        loss = loss + 0.0005 * logits.abs().sum(dim=0).max()
    if l1:   # models/regularizers.py:13-19
        loss = loss + 0.0005 * sum(v.abs().sum() for k, v in params.items() if "batch_normalization" not in k)
    return loss


def predictions(logits):
    """(softmax > 0.5) as float, models/efficientlab.py:174-176,291-292."""
    return (torch.softmax(logits, dim=-1) > 0.5).to(logits.dtype)


def inner_step(a, params, bn, x, y, lr, dc_scales=None, dropout_mask=None, label_smoothing=0.0, dice=False, l2=False,
               weight_decay_rate=1.0, adam_state=None, aspp_masks=None, round_ops=None, l1=False, darc1=False, store=None):
    """One `session.run(minimize_op)` (reptile.py:114-121,639-643): fwd + bwd + BN moving update + SGD apply.
    Mutates params / bn in place; returns (loss, grads dict, logits)."""
    if weight_decay_rate != 1.0:  # pre_step_op, meta_learners/variables.py:48-55
        for k in params:
            params[k] = params[k] * weight_decay_rate
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    logits, new_moving = forward(a, leaves, bn, x, True, dc_scales, dropout_mask, aspp_masks=aspp_masks, round_ops=round_ops, store=store)
    loss = loss_fn(a, leaves, logits, y, label_smoothing, dice, l2, l1, darc1)
    names = list(leaves)
    grads = torch.autograd.grad(loss, [leaves[k] for k in names], allow_unused=True)
    out_g = {}
    for k, g in zip(names, grads):
        if g is None:
            g = torch.zeros_like(params[k])
        out_g[k] = g
        if adam_state is None:
            params[k] = params[k] - lr * g          # tf.train.GradientDescentOptimizer
        else:
            # tf.train.AdamOptimizer(beta1=0, beta2=0.999, epsilon=1e-8) (models/efficientlab.py:16): m = g,
            # v = b2 v + (1-b2) g^2, w -= lr * sqrt(1 - b2^t) / (1 - b1^t) * m / (sqrt(v) + eps)
            t = adam_state["t"] + 1
            v = adam_state["v"].get(k, torch.zeros_like(g)) * 0.999 + 0.001 * g * g
            adam_state["v"][k] = v
            params[k] = params[k] - lr * math.sqrt(1.0 - 0.999 ** t) * g / (v.sqrt() + 1e-8)
    if adam_state is not None:
        adam_state["t"] += 1
    for k in []:
        pass
    for k, v in new_moving.items():
        bn[k] = v
    return float(loss.detach()), out_g, logits.detach()


# ------------------------------------------------------------------------------- Learner protocol (for tests / baseline)
class OracleLearner:
    """Implements the same Learner protocol as mliis_amd.learner.Learner on the CPU restatement, so the
    meta-learner host logic (Gecko/FOMLIS, sharding, all-reduce) can be exercised without a GPU."""

    def __init__(self, name="efficientnet-b0", image_size=224, rsd=(2, 4), seed=0, dtype=torch.float64, lr=1e-3,
                 l2=False, dice=False, label_smoothing=0.0, drop_connect=True, aspp=False, round_ops=None, l1=False, darc1=False,
                 skip_decoding=False):
        self.l1, self.darc1 = l1, darc1
        self.round_ops = round_ops    # None | "bf16" | "fp8": emulated reduced-precision matrix-core operands
        self.store = None             # callable(block) -> "march" | "small" | None: bf16 storage points of the training step (forward())
        self.a = arch(name, image_size, rsd, aspp, skip_decoding)
        self.params, self.bn = init_state(self.a, seed, dtype)
        self.dtype, self.lr, self.l2, self.dice, self.ls = dtype, lr, l2, dice, label_smoothing
        self.drop_connect = drop_connect
        self.names = list(self.params)
        self.sizes = [self.params[k].numel() for k in self.names]
        self.bn_names = list(self.bn)
        self.n_trainable = sum(self.sizes)
        self.feature_extractor_name = name
        self.final_layer_scope = "decode/final_layer_weights"
        self.optimizer = "sgd"

    # -- checkpoint surface (all global variables under their TF names)
    def named_numpy(self):
        out = {k: v.detach().numpy().astype("float32") for k, v in self.params.items()}
        for k, (mm, mv) in self.bn.items():
            out[k + "/moving_mean"], out[k + "/moving_variance"] = mm.numpy().astype("float32"), mv.numpy().astype("float32")
        return out

    def load_named(self, values, strict=True, prefixes=None, exclude_prefix=None):
        n = 0
        names = list(self.params) + [k + s for k in self.bn for s in ("/moving_mean", "/moving_variance")]
        for name in names:
            if prefixes is not None and not any(name.startswith(x) for x in prefixes):
                continue
            if exclude_prefix is not None and name.startswith(exclude_prefix):
                continue
            if name not in values:
                if strict:
                    raise KeyError("variable {} missing from checkpoint".format(name))
                continue
            v = torch.as_tensor(values[name]).to(self.dtype)
            if name in self.params:
                self.params[name] = v.reshape(self.params[name].shape).clone()
            else:
                k, which = name.rsplit("/", 1)
                mm, mv = self.bn[k]
                self.bn[k] = (v.clone(), mv) if which == "moving_mean" else (mm, v.clone())
            n += 1
        return n

    def close(self):
        pass

    # -- variable state (meta_learners/variables.py:58-80)
    def export_trainable(self) -> torch.Tensor:
        return torch.cat([self.params[k].reshape(-1) for k in self.names]).clone()

    def import_trainable(self, flat: torch.Tensor):
        off = 0
        for k, n in zip(self.names, self.sizes):
            self.params[k] = flat[off:off + n].reshape(self.params[k].shape).to(self.dtype).clone()
            off += n

    def export_bn(self) -> torch.Tensor:
        return torch.cat([torch.cat([self.bn[k][0], self.bn[k][1]]) for k in self.bn_names]).clone()

    def import_bn(self, flat: torch.Tensor):
        off = 0
        for k in self.bn_names:
            c = self.bn[k][0].numel()
            self.bn[k] = (flat[off:off + c].clone().to(self.dtype), flat[off + c:off + 2 * c].clone().to(self.dtype))
            off += 2 * c

    def axpby(self, a, x, b, y):
        y.mul_(b).add_(x, alpha=a)

    def comm_context(self):
        import contextlib
        return contextlib.nullcontext()

    def load_task(self, images, labels):
        self._x, self._y = torch.as_tensor(images), torch.as_tensor(labels)

    def inner_step(self, x, y=None, lr=None, dc_scales=None, dropout_mask=None, weight_decay_rate=1.0, drop_rate=None, aspp_masks=None):
        """inner_step(x, y, ...) on explicit tensors, or inner_step(batch_idx, ...) on the task given to load_task().
        (drop_rate is accepted for protocol compatibility; the oracle applies dropout only through an explicit dropout_mask.)"""
        if y is None:
            i = list(x)
            x, y = self._x[i], self._y[i]
        loss, _, _ = inner_step(self.a, self.params, self.bn, x, y, self.lr if lr is None else lr,
                                dc_scales if self.drop_connect else None, dropout_mask, self.ls, self.dice, self.l2,
                                weight_decay_rate, aspp_masks=aspp_masks, round_ops=self.round_ops, l1=self.l1, darc1=self.darc1, store=self.store)
        return loss

    def export_all(self):
        return {"params": {k: v.clone() for k, v in self.params.items()}, "bn": {k: (a.clone(), b.clone()) for k, (a, b) in self.bn.items()}}

    def import_all(self, st):
        self.params = {k: v.clone() for k, v in st["params"].items()}
        self.bn = {k: (a.clone(), b.clone()) for k, (a, b) in st["bn"].items()}

    def predict_resident(self, idx, training=False):
        return self.predict(self._x[list(idx)], training)

    def synchronize(self):
        pass

    def predict(self, x, training=False):
        with torch.no_grad():
            logits, _ = forward(self.a, self.params, self.bn, x, training, round_ops=self.round_ops)
        return predictions(logits)
