"""Architecture derivation for EfficientLab (EfficientNet-B0/B3 encoder + residual-skip decoder).

Pure host logic, no device code.  Re-derives, from the published EfficientNet stage table and
compound-scaling coefficients, the exact layer list the reference builds:

  * stage table + truncation by ``max_block_num``: models/efficientnet/efficientnet_builder.py:90-109,125-149
  * width/depth rounding:                            models/efficientnet/efficientnet_model.py:106-130
  * block expansion (repeats, stride-1 tail):        models/efficientnet/efficientnet_model.py:326-349
  * reduction endpoints:                             models/efficientnet/efficientnet_model.py:417-434
  * drop-connect rate per block:                     models/efficientnet/efficientnet_model.py:426-431
  * decoder (RSD) wiring:                            models/efficientlab.py:126-231
  * variable names / creation order:                 SURVEY.md Appendix D

The parameter order produced by :func:`param_table` is the TF variable *creation* order, which is what
``tf.trainable_variables()`` returns and therefore what ``VariableState`` (meta_learners/variables.py:58-80)
exports/imports; the flat arena uses the same order.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

# (repeats, kernel, stride, expand, in, out, se_ratio) -- the seven published EfficientNet-B0 stages.
_STAGES = (
    (1, 3, 1, 1, 32, 16, 0.25),
    (2, 3, 2, 6, 16, 24, 0.25),
    (2, 5, 2, 6, 24, 40, 0.25),
    (3, 3, 2, 6, 40, 80, 0.25),
    (3, 5, 1, 6, 80, 112, 0.25),
    (4, 5, 2, 6, 112, 192, 0.25),
    (1, 3, 1, 6, 192, 320, 0.25),
)
# name -> (width, depth) compound coefficients.
_COEFFS = {
    "efficientnet-b0": (1.0, 1.0),
    "efficientnet-b3": (1.2, 1.4),
}
# models/efficientlab.py:73-78
_DECODER = {
    "efficientnet-b0": dict(aspp_dimension=112, max_block_num=10),
    "efficientnet-b3": dict(aspp_dimension=136, max_block_num=17),
}
DEPTH_DIVISOR = 8
BN_MOMENTUM = 0.99
BN_EPS = 1e-3
DROP_CONNECT_RATE = 0.2
L2_WEIGHT = 0.0005
# models/efficientnet/constants.py:1-2
MEAN_RGB = (0.485 * 255, 0.456 * 255, 0.406 * 255)
STDDEV_RGB = (0.229 * 255, 0.224 * 255, 0.225 * 255)


def round_filters(filters: int, width: float, divisor: int = DEPTH_DIVISOR) -> int:
    if not width:
        return filters
    f = filters * width
    new = max(divisor, int(f + divisor / 2) // divisor * divisor)
    if new < 0.9 * f:
        new += divisor
    return int(new)


def round_repeats(repeats: int, depth: float) -> int:
    if not depth:
        return repeats
    return int(math.ceil(depth * repeats))


def same_pad(size: int, k: int, stride: int, dil: int = 1) -> Tuple[int, int, int]:
    """TF 'SAME': returns (out, pad_before, pad_after)."""
    out = -(-size // stride)
    eff = (k - 1) * dil + 1
    total = max((out - 1) * stride + eff - size, 0)
    return out, total // 2, total - total // 2


@dataclass
class Block:
    idx: int
    k: int
    stride: int
    expand: int
    cin: int
    cout: int
    se: int            # squeeze-excite reduced width
    h_in: int
    h_out: int
    skip: bool         # identity skip (+ drop-connect) present
    drop_rate: float   # drop-connect rate of this block (0 for block 0)
    executed: bool     # False for B3 blocks past reduction_4: variables exist, never run
    reduction: int = 0  # 1-indexed reduction endpoint this block's output is, or 0

    @property
    def cexp(self) -> int:
        return self.cin * self.expand


@dataclass
class RSD:
    """One residual-skip-decoder module (models/efficientlab.py:179-231)."""
    scope_index: int   # reduction_index = i-1  -> scope decode/decode_skip_connections_<idx>
    h_in: int          # deep map side
    h: int             # skip map side (= output side)
    c_deep: int        # channels of the deep (upsampled) map
    c_skip: int
    c_out: int
    upsample_conv: bool  # extra 1x1 branch when c_deep != c_out (not hit by b0/b3 with rsd 2 4)

    @property
    def c_cat(self) -> int:
        return self.c_deep + self.c_skip

    @property
    def c_pyr(self) -> int:
        return 2 * self.c_out + self.c_cat


@dataclass
class SkipDec:
    """The DeepLabv3+-style decoder of --skip_decoding (models/efficientlab.py:133-149): the embedded image resized to input / 4,
    concatenated with a 1x1-projected reduction_2 endpoint, refined by two depthwise-separable convs (sep_conv, :445-474)."""
    h_in: int          # embedded map side
    h: int             # image_size // 4 = the reduction_2 map side
    c_in: int          # channels of the embedded image (encoder output or ASPP output)
    c_skip_in: int     # channels of the reduction_2 endpoint
    c_skip: int        # aspp_dimension // 2
    c_sep: int         # aspp_dimension + c_skip: output channels of both sep_convs

    @property
    def c_cat(self) -> int:
        return self.c_in + self.c_skip


@dataclass
class Arch:
    name: str
    image_size: int
    stem_out: int
    blocks: List[Block]
    reductions: Dict[int, int]      # reduction idx (1..) -> block idx
    rsd: List[RSD]                  # in execution order (deepest first)
    aspp_dimension: int
    n_out: int = 2
    final_dropout: bool = False
    h_stem: int = 0
    executed_blocks: int = 0
    h_dec: int = 0                  # side of the final decoded map (input of final 1x1)
    aspp: bool = False              # --spatial_pyramid_pooling: ASPP between the encoder output and the RSD modules
    aspp_cin: int = 0               # channels / map side of the encoder output it reads (reduction_4)
    aspp_h: int = 0
    skipdec: Optional[SkipDec] = None   # --skip_decoding: between the (ASPP'd) embedded image and the RSD modules
    c_final: int = 0                # channels of the decoded map the final 1x1 conv reads


ASPP_DILATION = 6        # models/efficientlab.py:265-267 (96 / downsample factor 16)
ASPP_DROPOUT = 0.5       # models/efficientlab.py:248


def derive(name: str = "efficientnet-b0", image_size: int = 224, rsd: Optional[List[int]] = (2, 4),
           final_layer_dropout_rate: float = 0.0, spatial_pyramid_pooling: bool = False, skip_decoding: bool = False) -> Arch:
    if name not in _COEFFS:
        raise ValueError("feature_extractor_name must be in {} but is: {}".format(sorted(_COEFFS), name))
    width, depth = _COEFFS[name]
    dec = _DECODER[name]
    # stage truncation compares UN-rounded repeat sums (efficientnet_builder.py:104-107)
    stages = []
    total = 0
    for st in _STAGES:
        total += st[0]
        if total > dec["max_block_num"] + 1:
            break
        stages.append(st)
    blocks: List[Block] = []
    flat = []
    for (r, k, s, e, ci, co, se) in stages:
        ci, co, r = round_filters(ci, width), round_filters(co, width), round_repeats(r, depth)
        flat.append((k, s, e, ci, co, se))
        for _ in range(r - 1):
            flat.append((k, 1, e, co, co, se))
    stem_out = round_filters(32, width)
    h, _, _ = same_pad(image_size, 3, 2)
    h_stem = h
    nblocks = len(flat)
    reductions: Dict[int, int] = {}
    ridx = 0
    for i, (k, s, e, ci, co, se) in enumerate(flat):
        h_out = same_pad(h, k, s)[0]
        is_red = (i == nblocks - 1) or flat[i + 1][1] > 1
        if is_red:
            ridx += 1
            reductions[ridx] = i
        blocks.append(Block(idx=i, k=k, stride=s, expand=e, cin=ci, cout=co,
                            se=max(1, int(ci * se)), h_in=h, h_out=h_out,
                            skip=(s == 1 and ci == co),
                            drop_rate=DROP_CONNECT_RATE * float(i) / nblocks,
                            executed=True, reduction=ridx if is_red else 0))
        h = h_out
    # blocks after reduction_4 are pruned from execution (only endpoints reduction_1..4 are consumed)
    last = reductions[4]
    for b in blocks:
        b.executed = b.idx <= last
    mods: List[RSD] = []
    deep_c, deep_h = blocks[last].cout, blocks[last].h_out
    aspp_cin, aspp_h = deep_c, deep_h
    if spatial_pyramid_pooling:   # the ASPP output (aspp_dimension channels, same map) replaces the embedded image (efficientlab.py:129-131)
        deep_c = dec["aspp_dimension"]
    skipdec = None
    if skip_decoding:   # efficientlab.py:133-149: resize to input // 4, concat with the projected reduction_2 endpoint, two sep_convs
        sb = blocks[reductions[2]]
        h4 = image_size // 4
        if sb.h_out != h4:
            raise ValueError("--skip_decoding needs the reduction_2 map ({0}x{0}) to be image_size // 4 = {1}".format(sb.h_out, h4))
        c_skip = dec["aspp_dimension"] // 2
        skipdec = SkipDec(h_in=deep_h, h=h4, c_in=deep_c, c_skip_in=sb.cout, c_skip=c_skip, c_sep=dec["aspp_dimension"] + c_skip)
        deep_c, deep_h = skipdec.c_sep, h4
    for i in sorted(rsd or [], reverse=True):
        if not 1 <= i <= 4:
            raise ValueError("rsd entries must be reduction indices 1..4, got {}".format(i))
        sb = blocks[reductions[i]]
        m = RSD(scope_index=i - 1, h_in=deep_h, h=sb.h_out, c_deep=deep_c, c_skip=sb.cout,
                c_out=dec["aspp_dimension"], upsample_conv=(deep_c != dec["aspp_dimension"]))
        mods.append(m)
        deep_c, deep_h = m.c_out, m.h
    return Arch(name=name, image_size=image_size, stem_out=stem_out, blocks=blocks, reductions=reductions,
                rsd=mods, aspp_dimension=dec["aspp_dimension"],
                final_dropout=bool(final_layer_dropout_rate and final_layer_dropout_rate > 0),
                h_stem=h_stem, executed_blocks=last + 1, h_dec=deep_h, aspp=bool(spatial_pyramid_pooling), aspp_cin=aspp_cin,
                aspp_h=aspp_h, skipdec=skipdec, c_final=deep_c)


# ----------------------------------------------------------------------------------------------------------
# Parameter table (names, shapes, kinds) in TF creation order.
# ----------------------------------------------------------------------------------------------------------
@dataclass
class Param:
    name: str
    shape: Tuple[int, ...]
    kind: str          # 'conv' | 'dw' | 'bias' | 'gamma' | 'beta' | 'moving_mean' | 'moving_variance'
    trainable: bool
    l2: bool           # receives the 5e-4 L2 gradient when --l2 (models/regularizers.py:4-10)
    init: str          # 'normal_fanout' | 'glorot_uniform' | 'zeros' | 'ones'
    offset: int = 0    # filled by the arena
    executed: bool = True

    @property
    def size(self) -> int:
        n = 1
        for s in self.shape:
            n *= s
        return n


def _bn(prefix: str, c: int, out: List[Param], executed: bool = True):
    out.append(Param(prefix + "/gamma", (c,), "gamma", True, False, "ones", executed=executed))
    out.append(Param(prefix + "/beta", (c,), "beta", True, False, "zeros", executed=executed))
    out.append(Param(prefix + "/moving_mean", (c,), "moving_mean", False, False, "zeros", executed=executed))
    out.append(Param(prefix + "/moving_variance", (c,), "moving_variance", False, False, "ones", executed=executed))


def param_table(arch: Arch) -> List[Param]:
    """All global variables (trainable + BN moving stats) in creation order."""
    P: List[Param] = []
    fe = arch.name
    P.append(Param(f"{fe}/stem/conv2d/kernel", (3, 3, 3, arch.stem_out), "conv", True, True, "normal_fanout"))
    _bn(f"{fe}/stem/tpu_batch_normalization", arch.stem_out, P)
    for b in arch.blocks:
        s = f"{fe}/blocks_{b.idx}"
        ex = b.executed
        nb = 0   # running BN counter inside the block scope
        nconv = 0

        def bn_name():
            nonlocal nb
            n = "tpu_batch_normalization" + ("" if nb == 0 else f"_{nb}")
            nb += 1
            return n

        def conv_name():
            nonlocal nconv
            n = "conv2d" + ("" if nconv == 0 else f"_{nconv}")
            nconv += 1
            return n
        if b.expand != 1:
            P.append(Param(f"{s}/{conv_name()}/kernel", (1, 1, b.cin, b.cexp), "conv", True, True, "normal_fanout", executed=ex))
            _bn(f"{s}/{bn_name()}", b.cexp, P, ex)
        P.append(Param(f"{s}/depthwise_conv2d/depthwise_kernel", (b.k, b.k, b.cexp, 1), "dw", True, True, "normal_fanout", executed=ex))
        _bn(f"{s}/{bn_name()}", b.cexp, P, ex)
        P.append(Param(f"{s}/se/conv2d/kernel", (1, 1, b.cexp, b.se), "conv", True, True, "normal_fanout", executed=ex))
        P.append(Param(f"{s}/se/conv2d/bias", (b.se,), "bias", True, True, "zeros", executed=ex))
        P.append(Param(f"{s}/se/conv2d_1/kernel", (1, 1, b.se, b.cexp), "conv", True, True, "normal_fanout", executed=ex))
        P.append(Param(f"{s}/se/conv2d_1/bias", (b.cexp,), "bias", True, True, "zeros", executed=ex))
        P.append(Param(f"{s}/{conv_name()}/kernel", (1, 1, b.cexp, b.cout), "conv", True, True, "normal_fanout", executed=ex))
        _bn(f"{s}/{bn_name()}", b.cout, P, ex)
    if arch.aspp:   # tf.layers.conv2d defaults: glorot-uniform kernels, zero biases (efficientlab.py:258-283)
        s, d = "decode/spatial_pyramid_pooling", arch.aspp_dimension
        for scope, k, ci in ((f"{s}/branch_0", 1, arch.aspp_cin), (f"{s}/branch_1", 3, arch.aspp_cin), (f"{s}/branch_2", 1, arch.aspp_cin),
                             (s, 1, 3 * d)):
            P.append(Param(f"{scope}/conv2d/kernel", (k, k, ci, d), "conv", True, True, "glorot_uniform"))
            P.append(Param(f"{scope}/conv2d/bias", (d,), "bias", True, True, "zeros"))
    if arch.skipdec is not None:
        # variable scope decode/decode_skip_connections (efficientlab.py:135): tf.layers.conv2d (glorot-uniform, no bias) + BN, then two
        # sep_convs = keras DepthwiseConv2D + tf.layers.conv2d, both with conv_kernel_initializer, each followed by a BN.  Default layer
        # names count up inside the scope.  (The keras layer's variable name under a tf.variable_scope is restated from memory.)
        sd, s = arch.skipdec, "decode/decode_skip_connections"
        P.append(Param(f"{s}/conv2d/kernel", (1, 1, sd.c_skip_in, sd.c_skip), "conv", True, True, "glorot_uniform"))
        _bn(f"{s}/batch_normalization", sd.c_skip, P)
        cin = sd.c_cat
        for j in range(2):
            dw = "depthwise_conv2d" + ("" if j == 0 else f"_{j}")
            P.append(Param(f"{s}/{dw}/depthwise_kernel", (3, 3, cin, 1), "dw", True, True, "normal_fanout"))
            _bn(f"{s}/batch_normalization_{2 * j + 1}", cin, P)
            P.append(Param(f"{s}/conv2d_{j + 1}/kernel", (1, 1, cin, sd.c_sep), "conv", True, True, "normal_fanout"))
            _bn(f"{s}/batch_normalization_{2 * j + 2}", sd.c_sep, P)
            cin = sd.c_sep
    for m in arch.rsd:
        s = f"decode/decode_skip_connections_{m.scope_index}"
        convs = []
        if m.upsample_conv:
            convs.append((1, m.c_deep, m.c_out))
        convs += [(1, m.c_cat, m.c_out), (3, m.c_cat, m.c_out), (3, m.c_pyr, m.c_out)]
        for j, (k, ci, co) in enumerate(convs):
            cn = "conv2d" + ("" if j == 0 else f"_{j}")
            bn = "batch_normalization" + ("" if j == 0 else f"_{j}")
            P.append(Param(f"{s}/{cn}/kernel", (k, k, ci, co), "conv", True, True, "glorot_uniform"))
            P.append(Param(f"{s}/{cn}/bias", (co,), "bias", True, True, "zeros"))
            _bn(f"{s}/{bn}", co, P)
    P.append(Param("decode/final_layer_weights/kernel", (1, 1, arch.c_final, arch.n_out), "conv", True, True, "normal_fanout"))
    P.append(Param("decode/final_layer_weights/bias", (arch.n_out,), "bias", True, True, "zeros"))
    # l2_term's name filter ('batch_normalization' substring) exempts only BN gamma/beta.
    for p in P:
        if p.kind in ("gamma", "beta", "moving_mean", "moving_variance"):
            p.l2 = False
    return P


def count_trainable(arch: Arch, executed_only: bool = False) -> Tuple[int, int]:
    ps = [p for p in param_table(arch) if p.trainable and (p.executed or not executed_only)]
    return len(ps), sum(p.size for p in ps)


def forward_macs_per_image(arch: Arch) -> Dict[str, int]:
    """Algorithmic multiply-accumulates per image of the forward pass, per op family."""
    out = dict(stem=0, depthwise=0, pointwise=0, se=0, decoder=0)
    out["stem"] = arch.h_stem ** 2 * 27 * arch.stem_out
    for b in arch.blocks:
        if not b.executed:
            continue
        if b.expand != 1:
            out["pointwise"] += b.h_in ** 2 * b.cin * b.cexp
        out["depthwise"] += b.h_out ** 2 * b.k ** 2 * b.cexp
        out["se"] += 2 * b.cexp * b.se
        out["pointwise"] += b.h_out ** 2 * b.cexp * b.cout
    if arch.aspp:
        d = arch.aspp_dimension
        out["decoder"] += arch.aspp_h ** 2 * (10 * arch.aspp_cin * d + 3 * d * d) + arch.aspp_cin * d
    if arch.skipdec is not None:
        sd = arch.skipdec
        out["decoder"] += sd.h ** 2 * (sd.c_skip_in * sd.c_skip + 9 * sd.c_cat + sd.c_cat * sd.c_sep + 9 * sd.c_sep + sd.c_sep * sd.c_sep)
    for m in arch.rsd:
        px = m.h ** 2
        if m.upsample_conv:
            out["decoder"] += px * m.c_deep * m.c_out
        out["decoder"] += px * (m.c_cat * m.c_out + 9 * m.c_cat * m.c_out + 9 * m.c_pyr * m.c_out)
    out["decoder"] += arch.h_dec ** 2 * arch.c_final * arch.n_out
    return out


def depthwise_algorithmic_bytes(arch: Arch, n: int, elem: int = 4) -> Tuple[int, int]:
    """(fwd, bwd) algorithmic HBM bytes of all depthwise layers for a batch of n."""
    fwd = bwd = 0
    for b in arch.blocks:
        if not b.executed:
            continue
        i = n * b.h_in ** 2 * b.cexp
        o = n * b.h_out ** 2 * b.cexp
        w = b.k ** 2 * b.cexp
        fwd += elem * (i + o + w)
        bwd += elem * (2 * i + o + 2 * w)
    return fwd, bwd
