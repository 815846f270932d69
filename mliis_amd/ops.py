"""Thin tensor-level wrappers over the C ABI (one function per entry point family).

PyTorch is used only as the owner of device memory and streams: every function takes fp32 CUDA(HIP) tensors, passes raw
pointers + leading dimensions to libmliis_hip.so on the current stream and returns preallocated/new output tensors.
There is no eager/PyTorch fallback -- a missing library or a non-device tensor raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import torch

from ._lib import MliisError, lib
from .spec import BN_EPS, BN_MOMENTUM, MEAN_RGB, STDDEV_RGB

_MEAN3 = (C.c_float * 3)(*MEAN_RGB)
_STD3 = (C.c_float * 3)(*STDDEV_RGB)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _ptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise MliisError("mliis_amd ops need device tensors (no CPU path)")
    if t.dtype == torch.bfloat16:
        raise MliisError("this entry point reads fp32 tensors; a bfloat16 tensor was passed (bf16 storage is taken by the fused MBConv kernels only)")
    return t.data_ptr()


def _aptr(t: Optional[torch.Tensor]):
    """Pointer of an activation tensor whose storage type (float32 | bfloat16) is passed beside it (_dt)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise MliisError("mliis_amd ops need device tensors (no CPU path)")
    return t.data_ptr()


def _chk(t: torch.Tensor, dtype=torch.float32):
    if t.dtype != dtype:
        raise MliisError("expected {} tensor, got {}".format(dtype, t.dtype))
    return t


DT_F32, DT_BF16 = 0, 1   # MLIIS_DT_* of include/mliis_hip.h: storage type of the expanded MBConv tensors (z0, z1, a1 and their gradients)


def _dt(*tensors) -> int:
    """Storage-type code of tensors that must share it: float32 -> MLIIS_DT_F32, bfloat16 -> MLIIS_DT_BF16."""
    d = tensors[0].dtype
    if d not in (torch.float32, torch.bfloat16) or any(t.dtype != d for t in tensors):
        raise MliisError("expected float32 or bfloat16 tensors of ONE storage type, got {}".format([t.dtype for t in tensors]))
    return DT_BF16 if d == torch.bfloat16 else DT_F32


# ---- optional per-call timing (bench.py / tools): PROFILE = [] enables it; every wrapped call appends
#      {"op", "ms", "fn", **meta} with HIP events recorded on the stream the kernels are launched on ("fn" re-issues the launch).
PROFILE = None


def _timed(op, meta, fn):
    if PROFILE is None:
        return fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    st = torch.cuda.current_stream()
    s.record(st)
    r = fn()
    e.record(st)
    PROFILE.append(dict(op=op, ev=(s, e), fn=fn, **meta))
    return r


def profile_resolve(records):
    """Turn recorded event pairs into milliseconds (call after a stream/device synchronize)."""
    for r in records:
        if "ev" in r:
            s, e = r.pop("ev")
            r["ms"] = s.elapsed_time(e)
    return records


PRECISIONS = {"fp32": 0, "bf16": 1, "fp8": 2}   # MLIIS_PREC_* of include/mliis_hip.h: operand precision of the matrix cores, per call
FP8_ACT_SCALE = 16.0   # default power-of-two multiplier of activations before the e4m3 conversion (saturating at 448 / 16 = 28)


def _prec(precision) -> int:
    if precision not in PRECISIONS:
        raise MliisError("matmul precision must be one of {}, got {!r}".format(sorted(PRECISIONS), precision))
    return PRECISIONS[precision]


def conv2d_kernel_name(N, H, W, cred, nout, k, has_scale=False, precision="fp32"):
    buf = C.create_string_buffer(80)
    lib.call("mliis_conv2d_kernel_name", N, H, W, cred, nout, k, int(has_scale), _prec(precision), C.cast(buf, C.c_void_p), 80)
    return buf.value.decode()


def conv2d_plan(N, H, W, cred, nout, k):
    tm, nt, sp = C.c_int(), C.c_int(), C.c_int()
    lib.call("mliis_conv2d_plan", N, H, W, cred, nout, k, C.byref(tm), C.byref(nt), C.byref(sp))
    return tm.value, nt.value, sp.value


class Workspace:
    """Grow-only scratch buffer (floats) shared by all calls on a stream.  `on_grow` (optional) is called BEFORE the buffer is
    replaced: an owner whose captured HIP graphs have the old address baked in must drop them there (Learner does)."""

    def __init__(self, device="cuda", floats: int = 1 << 20, on_grow=None):
        self.device = device
        self.on_grow = on_grow
        self.buf = torch.empty(floats, dtype=torch.float32, device=device)

    def get(self, floats: int) -> torch.Tensor:
        if floats > self.buf.numel():
            if self.on_grow is not None:
                self.on_grow(floats)
            self.buf = torch.empty(int(floats * 1.25) + 1024, dtype=torch.float32, device=self.device)
        return self.buf


_default_ws = None


def default_ws() -> Workspace:
    global _default_ws
    if _default_ws is None:
        _default_ws = Workspace()
    return _default_ws


def rows_ld(t: torch.Tensor) -> Tuple[int, int, int]:
    """[..., C] view whose leading dims are contiguous rows with stride ld: returns (rows, C, ld)."""
    C_ = t.shape[-1]
    if t.stride(-1) != 1:
        raise MliisError("channel dim must be contiguous")
    ld = t.stride(-2)
    rows = t.numel() // C_
    # all outer dims must collapse onto a uniform row stride
    exp = ld
    for d in range(t.dim() - 2, -1, -1):
        if t.shape[d] != 1 and t.stride(d) != exp:
            raise MliisError("tensor is not a uniform-row-stride view: shape {} strides {}".format(tuple(t.shape), t.stride()))
        exp *= t.shape[d]
    return rows, C_, ld


# ------------------------------------------------------------------------------------------------ stem
def stem_conv_fwd(x, w, idx=None, out=None, stats_part=None, rows=None):
    """With stats_part (a float buffer, or rows=True for the same kernel without statistics): the row-strip kernel, z bit-identical,
    returns (out, nblk) -- nblk stage-1 partial blocks [nblk][2][Co] of z for the batch norm that follows; nblk == -1: the shape does
    not fit that kernel (rows wider than ~400 pixels), z comes from the plain kernel and the caller takes bn_stats_partial."""
    S, H, W, _ = x.shape
    N = S if idx is None else idx.numel()
    Co = w.shape[-1]
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    out = torch.empty((N, Ho, Wo, Co), dtype=torch.float32, device=x.device) if out is None else out
    if stats_part is not None or rows:
        nblk = C.c_int(0)
        if stem_conv_fwd_stats_fits(N, H, W, Co):
            lib.call("mliis_stem_conv_fwd_stats", _ptr(_chk(x)), _ptr(idx), _ptr(w), _ptr(out), N, H, W, Co, _MEAN3, _STD3, _ptr(stats_part),
                     stats_part.numel() if stats_part is not None else 0, C.byref(nblk), _stream())
            return out, nblk.value
        lib.call("mliis_stem_conv_fwd", _ptr(_chk(x)), _ptr(idx), _ptr(w), _ptr(out), N, H, W, Co, _MEAN3, _STD3, _stream())
        return out, -1
    lib.call("mliis_stem_conv_fwd", _ptr(_chk(x)), _ptr(idx), _ptr(w), _ptr(out), N, H, W, Co, _MEAN3, _STD3, _stream())
    return out


def stem_conv_fwd_stats_fits(N, H, W, Co):
    """whether mliis_stem_conv_fwd_stats takes this shape (its staging window: 9 input rows of 2 * ceil(W / 2) + 1 pixels, in LDS)"""
    Wo = (W + 1) // 2
    return (9 * (2 * Wo + 1) * 3 + 512 * 8) * 4 <= 64 * 1024


def stem_conv_fwd_stats_floats(N, H, W, Co):
    return lib.size("mliis_stem_conv_fwd_stats_floats", N, H, W, Co)


def stem_conv_bwd_filter(x, dz, idx=None, out=None, ws: Optional[Workspace] = None, partial=None):
    S, H, W, _ = x.shape
    N, _, _, Co = dz.shape
    if partial is not None:
        lib.call("mliis_stem_conv_bwd_filter", _ptr(x), _ptr(idx), _ptr(dz), None, N, H, W, Co, _MEAN3, _STD3, _ptr(partial), partial.numel(),
                 _stream())
        return None
    ws = ws or default_ws()
    need = lib.size("mliis_stem_conv_bwd_filter_workspace_floats", N, H, W, Co)
    buf = ws.get(need)
    out = torch.empty((3, 3, 3, Co), dtype=torch.float32, device=x.device) if out is None else out
    lib.call("mliis_stem_conv_bwd_filter", _ptr(x), _ptr(idx), _ptr(dz), _ptr(out), N, H, W, Co, _MEAN3, _STD3, _ptr(buf), buf.numel(),
             _stream())
    return out


# ------------------------------------------------------------------------------------------------ depthwise
def dwconv_fwd(x, w, stride, out=None, stats_part=None):
    """With stats_part (a float buffer) the launch also emits the next batch norm's stage-1 statistics; returns (out, nblk)."""
    N, H, W, C_ = x.shape
    k = w.shape[0]
    Ho, Wo = -(-H // stride), -(-W // stride)
    out = torch.empty((N, Ho, Wo, C_), dtype=torch.float32, device=x.device) if out is None else out
    meta = dict(bytes=4.0 * (x.numel() + out.numel() + k * k * C_), shape=(N, H, W, C_, k, stride)) if PROFILE is not None else {}
    nblk = C.c_int(0)
    _timed("dwconv_fwd", meta, lambda: lib.call("mliis_dwconv_fwd", _ptr(_chk(x)), _ptr(w), _ptr(out), N, H, W, C_, k, stride, _ptr(stats_part),
                                                stats_part.numel() if stats_part is not None else 0, C.byref(nblk), _stream()))
    if stats_part is not None:
        return out, nblk.value
    return out


def dwconv_bwd_data(dy, w, stride, in_hw, out=None, bn=None, part=None):
    """bn = (z, mean, rstd, gamma, beta) with a float buffer `part`: the result is the gradient w.r.t. swish(bn(z)) and the launch also
    leaves stage 1 of that batch norm's backward in `part`; returns (out, nblk) -- pass (part, nblk) to bn_bwd(stage1=...) when
    nblk > 0."""
    N, _, _, C_ = dy.shape
    H, W = in_hw
    k = w.shape[0]
    out = torch.empty((N, H, W, C_), dtype=torch.float32, device=dy.device) if out is None else out
    if bn is not None:
        z, mean, rstd, gamma, beta = bn
        nblk = C.c_int(0)
        _chk(z)
        lib.call("mliis_dwconv_bwd_data_bn", _ptr(_chk(dy)), _ptr(w), _ptr(out), N, H, W, C_, k, stride, _ptr(z), _ptr(mean), _ptr(rstd),
                 _ptr(gamma), _ptr(beta), _ptr(part), part.numel(), C.byref(nblk), _stream())
        return out, nblk.value
    # bwd algorithmic bytes: read dY, read X, write dX, read W, write dW -- split here as data: dY + dX + W
    meta = dict(bytes=4.0 * (dy.numel() + out.numel() + k * k * C_), shape=(N, H, W, C_, k, stride)) if PROFILE is not None else {}
    _timed("dwconv_bwd_data", meta, lambda: lib.call("mliis_dwconv_bwd_data", _ptr(_chk(dy)), _ptr(w), _ptr(out), N, H, W, C_, k, stride, _stream()))
    return out


def dwconv_bwd_filter(x, dy, k, stride, out=None, ws: Optional[Workspace] = None, partial=None):
    N, H, W, C_ = x.shape
    if partial is not None:
        lib.call("mliis_dwconv_bwd_filter", _ptr(x), _ptr(dy), None, N, H, W, C_, k, stride, _ptr(partial), partial.numel(), _stream())
        return None
    ws = ws or default_ws()
    buf = ws.get(lib.size("mliis_dwconv_bwd_filter_workspace_floats", N, H, W, C_, k, stride))
    out = torch.empty((k, k, C_, 1), dtype=torch.float32, device=x.device) if out is None else out
    meta = dict(bytes=4.0 * (x.numel() + k * k * C_), shape=(N, H, W, C_, k, stride)) if PROFILE is not None else {}   # filter: X + dW
    _timed("dwconv_bwd_filter", meta, lambda: lib.call("mliis_dwconv_bwd_filter", _ptr(x), _ptr(dy), _ptr(out), N, H, W, C_, k, stride, _ptr(buf),
                                                       buf.numel(), _stream()))
    return out


# ------------------------------------------------------------------------------------------------ large-map MBConv depthwise half
def dwconv_bn_fwd(z, w, stride, bn=None, part=None, nblk=0, out=None, stats_part=None, eps=BN_EPS, momentum=BN_MOMENTUM):
    """y = dwconv(swish(bn(z))) with the batch norm applied while z is staged (csrc/dwmarch.hip); bn = (gamma, beta, mean, rstd,
    moving_mean | None, moving_var | None) or None (plain depthwise conv of z).  nblk > 0: `part` holds the producer's stage-1 sums
    [nblk][2][C] -- folded here, mean / rstd written, moving averages updated; nblk == 0: mean / rstd are inputs.  With stats_part
    the launch also emits the next batch norm's stage-1 sums; returns (out, blocks)."""
    N, H, W, C_ = z.shape
    k = w.shape[0]
    Ho, Wo = -(-H // stride), -(-W // stride)
    out = torch.empty((N, Ho, Wo, C_), dtype=z.dtype, device=z.device) if out is None else out
    g, b, m, r, mm, mv = bn if bn is not None else (None,) * 6
    meta = dict(bytes=float(z.numel() * z.element_size() + out.numel() * out.element_size() + 4 * k * k * C_),
                shape=(N, H, W, C_, k, stride)) if PROFILE is not None else {}
    nb = C.c_int(0)
    _timed("dwconv_bn_fwd", meta, lambda: lib.call(
        "mliis_dwconv_bn_fwd", _aptr(z), _ptr(part) if nblk else None, int(nblk), _ptr(g), _ptr(b), _ptr(m), _ptr(r), _ptr(mm), _ptr(mv),
        float(eps), float(momentum), _ptr(w), _aptr(out), N, H, W, C_, k, stride, _ptr(stats_part),
        stats_part.numel() if stats_part is not None else 0, C.byref(nb), _dt(z), _dt(out), _stream()))
    if stats_part is not None:
        return out, nb.value
    return out


def dwconv_bn_bwd_blocks(N, H, W, C_, k, stride) -> int:
    return int(lib.raw("mliis_dwconv_bn_bwd_blocks")(N, H, W, C_, k, stride))


def dwconv_bn_bwd(dy, z, w, stride, bn=None, out=None, dw=None, dw_part=None, bn_part=None, ws: Optional[Workspace] = None):
    """ONE pass over (dy, z): dx = gradient w.r.t. a = swish(bn(z)) (bn = (mean, rstd, gamma, beta); None: a = z), the depthwise filter
    gradient as slabs in dw_part [blocks][k*k][C] (folded into `dw` when given; dw_part None: taken from ws and folded into a new /
    the given dw) and, with bn_part, stage 1 of the batch norm's backward [blocks][2][C].  Returns (dx, dw | None, blocks)."""
    N, H, W, C_ = z.shape
    k = w.shape[0]
    out = torch.empty((N, H, W, C_), dtype=z.dtype, device=z.device) if out is None else out
    blocks = dwconv_bn_bwd_blocks(N, H, W, C_, k, stride)
    if dw_part is None:
        dw_part = (ws or default_ws()).get(blocks * k * k * C_)
        dw = torch.empty((k, k, C_, 1), dtype=torch.float32, device=z.device) if dw is None else dw
    m, r, g, b = bn if bn is not None else (None,) * 4
    meta = dict(bytes=4.0 * (2 * z.numel() + dy.numel() + 2 * k * k * C_), shape=(N, H, W, C_, k, stride)) if PROFILE is not None else {}
    nb = C.c_int(0)
    _timed("dwconv_bn_bwd", meta, lambda: lib.call(
        "mliis_dwconv_bn_bwd", _aptr(dy), _aptr(z), _ptr(m), _ptr(r), _ptr(g), _ptr(b), _ptr(w), _aptr(out), _ptr(dw), N, H, W, C_, k,
        stride, _ptr(dw_part), dw_part.numel(), _ptr(bn_part), bn_part.numel() if bn_part is not None else 0, C.byref(nb), _dt(dy), _dt(z, out),
        _stream()))
    return out, dw, nb.value


def mbconv_dw_bwd_march(da2, z1, bn1, gate, chan_add, stage1, dgamma1, dbeta1, z0, bn0, w, stride, out, dw_part, bn_part):
    """dwconv_bn_bwd with the depthwise batch norm's backward apply formed while (da2, z1) are staged: bn1 / bn0 = (mean, rstd, gamma,
    beta); stage1 [N][2][C] from se_mlp_bwd_bn.  Returns the block count (slabs of dw_part, blocks of bn_part)."""
    N, H, W, C_ = z0.shape
    k = w.shape[0]
    nb = C.c_int(0)
    meta = dict(bytes=4.0 * (2 * z0.numel() + 2 * z1.numel() + 2 * k * k * C_), shape=(N, H, W, C_, k, stride)) if PROFILE is not None else {}
    _timed("dwconv_bn_bwd", meta, lambda: lib.call(
        "mliis_mbconv_dw_bwd_march", _aptr(da2), _aptr(z1), _ptr(bn1[0]), _ptr(bn1[1]), _ptr(bn1[2]), _ptr(bn1[3]), _ptr(gate),
        _ptr(chan_add), _ptr(stage1), int(stage1.shape[0]) if stage1.dim() == 3 else N, _ptr(dgamma1), _ptr(dbeta1), _aptr(z0), _ptr(bn0[0]),
        _ptr(bn0[1]), _ptr(bn0[2]), _ptr(bn0[3]), _ptr(w), _aptr(out), N, H, W, C_, k, stride, _ptr(dw_part), dw_part.numel(), _ptr(bn_part),
        bn_part.numel(), C.byref(nb), _dt(da2, z1), _dt(z0, out), _stream()))
    return nb.value


# ------------------------------------------------------------------------------------------------ small-map MBConv depthwise half
def mbconv_dw_small_supported(N, H, W, C_, k, stride) -> bool:
    return bool(lib.raw("mliis_mbconv_dw_small_supported")(N, H, W, C_, k, stride))


def mbconv_dw_small_group_width(C_, k) -> int:
    """Channels per workgroup (2 or 4) the small-map launches choose for C channels and a k x k filter."""
    return int(lib.raw("mliis_mbconv_dw_small_group_width")(C_, k))


def _blocked_ptr(z0, z0_blocked):
    if z0_blocked is None:
        return None
    if z0_blocked.dtype != z0.dtype or z0_blocked.numel() < z0.numel():
        raise MliisError("z0_blocked must have z0's storage type and size")
    return _aptr(z0_blocked)


def mbconv_dw_fwd_small(z0, part0, nblk0, bn0, w, bn1, z1, a1, s, a0=None, eps=BN_EPS, momentum=BN_MOMENTUM, group_width=0, z0_blocked=None,
                        z1_blocked=False):
    """bn0 / bn1 = (gamma, beta, mean_out, rstd_out, moving_mean | None, moving_var | None).  One launch: fold bn0's statistics,
    a0 = swish(bn0(z0)), depthwise k x k (stride 1), exact statistics of z1, a1 = swish(bn1(z1)), s = per-image mean of a1.
    group_width: channels per workgroup (0 = the planner's choice).  z0_blocked (a buffer of z0's size and type) / z1_blocked: leave a
    copy of z0 / write z1 in the group-blocked layout that mbconv_dw_bwd_small(z0_blocked=..., z1_blocked=True) re-reads contiguously."""
    N, H, W, C_ = z0.shape
    k = w.shape[0]
    g0, b0, m0, r0, mm0, mv0 = bn0
    g1, b1, m1, r1, mm1, mv1 = bn1
    lib.call("mliis_mbconv_dw_fwd_small", _aptr(z0), _ptr(part0), int(nblk0), _ptr(g0), _ptr(b0), _ptr(m0), _ptr(r0), _ptr(mm0), _ptr(mv0),
             _ptr(w), _ptr(g1), _ptr(b1), _ptr(m1), _ptr(r1), _ptr(mm1), _ptr(mv1), _aptr(a0), _aptr(z1), _aptr(a1), _ptr(s), N, H, W, C_, k,
             float(eps), float(momentum), int(group_width), _dt(z0, z1, a1) if a0 is None else _dt(z0, z1, a1, a0),
             _blocked_ptr(z0, z0_blocked), int(bool(z1_blocked)), _stream())
    return z1, a1, s


def mbconv_dw_bwd_small(da2, gate, chan_add, z1, bn1, w, z0, bn0, dgamma1, dbeta1, dw, dgamma0, dbeta0, dz0, group_width=0, z0_blocked=None,
                        z1_blocked=False, da2_blocked=False):
    """bn1 / bn0 = (mean, rstd, gamma, beta).  One launch: bn1 backward, depthwise filter gradient (complete) and backward-data, bn0
    backward; dz0 = gradient w.r.t. the expand conv's output.  da2_blocked: da2 arrives in the group-blocked layout
    (conv2d_bwd_data(gate=..., out_block=group width))."""
    N, H, W, C_ = z1.shape
    k = w.shape[0]
    lib.call("mliis_mbconv_dw_bwd_small", _aptr(da2), _ptr(gate), _ptr(chan_add), _aptr(z1), _ptr(bn1[0]), _ptr(bn1[1]), _ptr(bn1[2]),
             _ptr(bn1[3]), _ptr(w), _aptr(z0), _ptr(bn0[0]), _ptr(bn0[1]), _ptr(bn0[2]), _ptr(bn0[3]), _ptr(dgamma1), _ptr(dbeta1), _ptr(dw),
             _ptr(dgamma0), _ptr(dbeta0), _aptr(dz0), N, H, W, C_, k, int(group_width), _dt(da2, z1, z0, dz0),
             _blocked_ptr(z0, z0_blocked), int(bool(z1_blocked)) | (2 if da2_blocked else 0), _stream())
    return dz0


# ------------------------------------------------------------------------------------------------ dense conv
def transpose_tiles(desc_rows):
    """Tile count of a transpose table given as HOST rows (offset, taps, Cin, Cout): pass it to transpose_weights(tiles=...)."""
    return int(sum(int(t) * ((int(ci) + 31) // 32) * ((int(co) + 31) // 32) for _, t, ci, co in desc_rows))


def transpose_weights(src, dst, desc, amax=None, tiles=0, x3=None, rng=None):
    """dst <- HWOI copies of the dense-conv weights listed in desc (device int32 [n,4] = offset, taps, Cin, Cout); amax (optional,
    float [n]): also max |w| per tensor -- the fp8 operand scale.  tiles = transpose_tiles(rows of desc): one workgroup per 32 x 32
    tile; 0: a fixed grid that strides over the tiles.  x3 (an X3Images over the same arena): its images are rebuilt too.
    rng = (generator state, MaskPlan): the masks of the training step are drawn by the same launch (rng_masks)."""
    if rng is not None and tiles > 0:
        st, plan = rng
        has = x3 is not None and x3.desc is not None
        lib.call("mliis_weight_shadows_rng", _ptr(src), _ptr(dst), _ptr(desc), int(desc.shape[0]), int(tiles), _ptr(amax),
                 _ptr(x3.images) if has else None, _ptr(x3.desc) if has else None, len(x3.rows) if has else 0, x3.blocks if has else 0,
                 _ptr(st), plan.n, plan.outs, plan.numels, plan.keep, plan.keeps, plan.row_len, plan.floor_form, _stream())
        return
    if rng is not None:
        rng_masks(*rng)
    if x3 is not None and x3.desc is not None and tiles > 0:   # the split-product weight images ride in the same launch
        lib.call("mliis_weight_shadows", _ptr(src), _ptr(dst), _ptr(desc), int(desc.shape[0]), int(tiles), _ptr(amax), _ptr(x3.images), _ptr(x3.desc),
                 len(x3.rows), x3.blocks, _stream())
        return
    lib.call("mliis_transpose_weights", _ptr(src), _ptr(dst), _ptr(desc), int(desc.shape[0]), int(tiles), _ptr(amax), _stream())
    if x3 is not None:
        x3.pack(src)


def hwoi(w):
    """K-contiguous copy [k,k,Cout,Cin] of one HWIO weight tensor (what the learner keeps for its whole arena, refreshed per step)."""
    k, _, cin, cout = w.shape
    dst = torch.empty((k, k, cout, cin), dtype=torch.float32, device=w.device)
    rows = [[0, k * k, cin, cout]]
    transpose_weights(w.contiguous(), dst, torch.tensor(rows, dtype=torch.int32, device=w.device), tiles=transpose_tiles(rows))
    return dst


def conv2d_fwd(x, w, bias=None, dil=1, out=None, accumulate=False, ws: Optional[Workspace] = None, nhw=None, stats_part=None,
               stats_swish=False, wt=None, x_scale=None, border_bias=None, ci_begin=0, precision="fp32", fp8_act_scale=FP8_ACT_SCALE,
               fp8_w_amax=None, out_block=0):
    """x: [N,H,W,>=Cin] view (channel slice allowed); w: [k,k,Cin,Cout]; wt: its K-contiguous copy (built here when not given --
    the kernels only read wt).  With stats_part (a float buffer) the epilogue also emits the next batch norm's stage-1 statistics
    and the function returns (out, nblk); nblk == 0 means they were not produced.  out_block = v (2 | 4): `out` receives the
    group-blocked layout [Cout / v][N H W][v] (streamed 1x1 plan only: conv1x1_stream_eligible)."""
    N, H, W = nhw if nhw is not None else x.shape[:3]
    k, _, Cin_total, Cout = w.shape
    wt = hwoi(w) if wt is None else wt
    rows, Cin, ldx = rows_ld(x)
    if ci_begin + Cin > Cin_total:
        raise MliisError("conv2d_fwd: x has {} channels, weight has {} (window starts at {})".format(Cin, Cin_total, ci_begin))
    out = torch.empty((N, H, W, Cout), dtype=torch.float32, device=x.device) if out is None else out
    _, co, ldy = rows_ld(out)
    ws = ws or default_ws()
    buf = ws.get(lib.size("mliis_conv2d_workspace_floats", N, H, W, Cin, Cout, k))
    meta = {}
    if PROFILE is not None:
        tm, nt, sp = conv2d_plan(N, H, W, Cin, Cout, k)
        meta = dict(kernel=conv2d_kernel_name(N, H, W, Cin, Cout, k, x_scale is not None, precision), splits=sp,
                    flops=2.0 * N * H * W * k * k * Cin * Cout, shape=(N, H, W, Cin, Cout, k, dil))
    nblk = C.c_int(0)
    prec = _prec(precision)
    if prec == 2 and k == 1 and fp8_w_amax is None:   # stand-alone call: take the tensor's amax here (the learner gets it from the
        fp8_w_amax = w.abs().max().reshape(1)          # per-step weight-shadow launch)
    if stats_part is not None and stats_part.numel() < (-(-N * H * W // 16)) * 2 * Cout:
        raise MliisError("conv2d_fwd: stats_part too small")
    _timed("conv2d_fwd", meta, lambda: lib.call("mliis_conv2d_fwd", _aptr(x), ldx, _ptr(x_scale), _ptr(wt), _ptr(bias), _ptr(border_bias), _aptr(out), ldy, N, H, W,
                                                Cin_total, ci_begin, Cin, Cout, k,
                                                dil, int(accumulate), _ptr(stats_part), int(stats_swish), C.byref(nblk), _ptr(buf),
                                                buf.numel(), prec, float(fp8_act_scale), _ptr(fp8_w_amax), _dt(x), _dt(out) | (int(out_block) << 8),
                                                _stream()))
    if stats_part is not None:
        return out, nblk.value
    return out


def conv2d_fwd_bnin_ok(N, H, W, Cin, Cout) -> bool:
    """True when a 1x1 conv of this shape takes the streamed plan, i.e. when conv2d_fwd_bnin accepts it."""
    return bool(lib.size("mliis_conv2d_fwd_bnin_ok", N, H, W, Cin, Cout))


def conv2d_fwd_bnin(z, part, nblk, mean, rstd, gamma, beta, a_out, w, out, moving=None, img_scale=None, res=None, stats_part=None,
                    stats_swish=False, wt=None, precision="fp32", fp8_act_scale=FP8_ACT_SCALE, fp8_w_amax=None, out_block=0, eps=BN_EPS,
                    momentum=BN_MOMENTUM):
    """out = conv1x1(a, w) with a = bn(z) * img_scale[image] + res formed while z is loaded (the project batch norm of the MBConv block
    in front, fused into this block's expand conv); a is also written to a_out, mean / rstd / moving averages as bn_apply_fused does.
    Returns (out, stats_nblk)."""
    N, H, W = z.shape[:3]
    _, Cin, ldz = rows_ld(z)
    k, _, Cin_w, Cout = w.shape
    if k != 1 or Cin_w != Cin:
        raise MliisError("conv2d_fwd_bnin: a 1x1 weight over the {} channels of z expected, got {}".format(Cin, tuple(w.shape)))
    wt = hwoi(w) if wt is None else wt
    _, _, ldy = rows_ld(out)
    _, _, ldo = rows_ld(a_out)
    ldr = rows_ld(res)[2] if res is not None else 0
    mm, mv = (None, None) if moving is None else moving
    prec = _prec(precision)
    if prec == 2 and fp8_w_amax is None:
        fp8_w_amax = w.abs().max().reshape(1)
    if stats_part is not None and stats_part.numel() < (-(-N * H * W // 16)) * 2 * Cout:
        raise MliisError("conv2d_fwd_bnin: stats_part too small")
    meta = {}
    if PROFILE is not None:
        meta = dict(kernel=conv2d_kernel_name(N, H, W, Cin, Cout, 1, False, precision), splits=1, flops=2.0 * N * H * W * Cin * Cout,
                    shape=(N, H, W, Cin, Cout, 1, 1))
    nb = C.c_int(0)
    _timed("conv2d_fwd", meta, lambda: lib.call("mliis_conv2d_fwd_bnin", _ptr(_chk(z)), ldz, _ptr(part), int(nblk), eps, momentum, _ptr(mean), _ptr(rstd),
                                                _ptr(mm), _ptr(mv), _ptr(gamma), _ptr(beta), _ptr(img_scale), _ptr(res), ldr, _ptr(a_out), ldo,
                                                _ptr(wt), _aptr(out), ldy, N, H, W, Cin, Cout, _ptr(stats_part), int(stats_swish), C.byref(nb), prec,
                                                float(fp8_act_scale), _ptr(fp8_w_amax), _dt(out) | (int(out_block) << 8), _stream()))
    return out, nb.value


class X3Images:
    """Pre-split weight images of the dense convs that run under MLIIS_PREC_F32X3 (csrc/conv_x3.hip): add() every (conv, direction)
    once, finish() builds the device descriptor table, pack(theta) re-splits all of them in ONE launch (once per inner step: the
    weights change with every optimizer step)."""

    X3_MIN_K = 512   # reductions shorter than this stay on the native fp32 instruction (those launches are not matrix-pipe-bound)

    def __init__(self, device):
        self.device = device
        self.rows, self.view, self.blocks, self.bytes = [], {}, 0, 0
        self.images = self.desc = None

    @staticmethod
    def eligible(cred, nout, k) -> bool:
        return cred >= 32 and cred % 4 == 0 and nout % 4 == 0 and k * k * cred >= X3Images.X3_MIN_K

    def add(self, key, mode, theta_off, k, cin_total, cout, ci_begin=0, cin=None):
        """mode "fwd": columns = cout, reduction over the window [ci_begin, ci_begin + cin); "bwd": columns = that window, reduction over cout."""
        cin = cin_total - ci_begin if cin is None else cin
        cred, nout = (cin, cout) if mode == "fwd" else (cout, cin)
        nbytes = lib.size("mliis_x3_image_bytes", cred, nout, k)
        nblk = lib.size("mliis_x3_image_blocks", cred, nout, k)
        self.rows.append([int(theta_off), k * k, cin_total, cout, ci_begin, cin, (0 if mode == "fwd" else 1) | (self.blocks << 8), self.bytes])
        self.view[(key, mode)] = (self.bytes, nbytes)
        self.blocks += nblk
        self.bytes += nbytes

    def finish(self):
        self.images = torch.zeros(max(self.bytes, 16), dtype=torch.uint8, device=self.device)
        self.desc = torch.tensor(self.rows, dtype=torch.int64, device=self.device) if self.rows else None
        return self

    def pack(self, theta):
        if self.desc is not None:
            lib.call("mliis_x3_pack_weights", _ptr(theta), _ptr(self.images), _ptr(self.desc), len(self.rows), self.blocks, _stream())

    def image(self, key, mode):
        off, n = self.view[(key, mode)]
        return self.images[off:off + n]

    def has(self, key, mode):
        return (key, mode) in self.view


def x3_image_of(w, mode, ci_begin=0, cin=None):
    """Stand-alone image of one weight tensor [k,k,Cin,Cout] (tests / probes: the learner keeps an X3Images over its arena)."""
    k, _, cin_total, cout = w.shape
    im = X3Images(w.device)
    im.add("w", mode, 0, k, cin_total, cout, ci_begin, cin)
    im.finish().pack(w.contiguous().view(-1))
    return im.image("w", mode)


def conv2d_x3_kernel_name(N, H, W, cred, nout, k):
    """The instantiation an x3 call launches, as rocprofv3 prints it (+ x3_fixup_k<NT> for the stream-K tiles)."""
    plan = (C.c_int * 8)()
    lib.call("mliis_conv2d_x3_plan", N, H, W, cred, nout, k, plan)
    return "conv_x3_k<%d>" % plan[0]


def conv2d_fwd_x3(x, image, k, cout, bias=None, dil=1, out=None, accumulate=False, ws: Optional[Workspace] = None, stats_part=None,
                  stats_swish=False, border_bias=None):
    """conv2d_fwd under MLIIS_PREC_F32X3 with the conv's forward weight image (X3Images / x3_image_of); x may be a channel-sliced view
    -- the window is the one the image was packed for.  Returns out, or (out, nblk) with stats_part."""
    N, H, W = x.shape[:3]
    rows, Cin, ldx = rows_ld(x)
    out = torch.empty((N, H, W, cout), dtype=torch.float32, device=x.device) if out is None else out
    _, co, ldy = rows_ld(out)
    ws = ws or default_ws()
    buf = ws.get(lib.size("mliis_conv2d_x3_workspace_floats", N, H, W, Cin, cout, k))
    nblk = C.c_int(0)
    meta = {}
    if PROFILE is not None:
        meta = dict(kernel=conv2d_x3_kernel_name(N, H, W, Cin, cout, k), flops=2.0 * N * H * W * k * k * Cin * cout, shape=(N, H, W, Cin, cout, k, dil))
    _timed("conv2d_fwd_x3", meta, lambda: lib.call("mliis_conv2d_fwd_x3", _aptr(x), ldx, _ptr(image), image.numel() * image.element_size(), _ptr(bias), _ptr(border_bias), _aptr(out), ldy,
                                                   N, H, W, Cin, cout, k, dil, int(accumulate), _ptr(stats_part), int(stats_swish), C.byref(nblk),
                                                   _ptr(buf), buf.numel(), _stream()))
    if stats_part is not None:
        return out, nblk.value
    return out


def conv2d_bwd_data_x3(dy, image, k, cin_out, dil=1, out=None, accumulate=False, ws: Optional[Workspace] = None):
    """conv2d_bwd_data under MLIIS_PREC_F32X3 with the conv's backward weight image (its input-channel window has cin_out channels)."""
    N, H, W = dy.shape[:3]
    rows, Cout, lddy = rows_ld(dy)
    out = torch.empty((N, H, W, cin_out), dtype=torch.float32, device=dy.device) if out is None else out
    _, ci, lddx = rows_ld(out)
    ws = ws or default_ws()
    buf = ws.get(lib.size("mliis_conv2d_x3_workspace_floats", N, H, W, Cout, cin_out, k))
    meta = {}
    if PROFILE is not None:
        meta = dict(kernel=conv2d_x3_kernel_name(N, H, W, Cout, cin_out, k), flops=2.0 * N * H * W * k * k * cin_out * Cout, shape=(N, H, W, Cout, cin_out, k, dil))
    _timed("conv2d_bwd_data_x3", meta, lambda: lib.call("mliis_conv2d_bwd_data_x3", _aptr(dy), lddy, _ptr(image), image.numel() * image.element_size(), _aptr(out), lddx, N, H, W, cin_out,
                                                        Cout, k, dil, int(accumulate), _ptr(buf), buf.numel(), _stream()))
    return out


def conv1x1_stream_eligible(N, H, W, cred, nout, precision="fp32"):
    """True when a plain 1x1 conv of this shape (no input scale, accumulate or border bias) takes the streamed kernel -- the plan that can
    write its output group-blocked (conv2d_fwd(out_block=...), conv2d_bwd_data(gate=..., out_block=...))."""
    return conv2d_kernel_name(N, H, W, cred, nout, 1, False, precision).startswith("conv1x1_stream_k")


def conv2d_bwd_data(dy, w, dil=1, ci_begin=0, ci_count=None, out=None, accumulate=False, ws: Optional[Workspace] = None, precision="fp32",
                    bn=None, part=None, gate=None, out_block=0):
    """bn = (x, mean, rstd, img_scale or None) with a float buffer `part`: `out` is the gradient w.r.t. the output of a plain batch norm
    over x and the launch may also leave stage 1 of that batch norm's backward in `part`; returns (out, nblk) then -- pass
    (part, nblk) to bn_bwd(stage1=...) when nblk > 0."""
    N, H, W = dy.shape[:3]
    k, _, Cin, Cout = w.shape
    ci_count = Cin - ci_begin if ci_count is None else ci_count
    _, cy, lddy = rows_ld(dy)
    if cy != Cout:
        raise MliisError("conv2d_bwd_data: dy has {} channels, weight has {}".format(cy, Cout))
    out = torch.empty((N, H, W, ci_count), dtype=torch.float32, device=dy.device) if out is None else out
    _, _, lddx = rows_ld(out)
    ws = ws or default_ws()
    buf = ws.get(lib.size("mliis_conv2d_workspace_floats", N, H, W, Cout, ci_count, k))
    prec = _prec(precision)
    meta = {}
    if PROFILE is not None:
        tm, nt, sp = conv2d_plan(N, H, W, Cout, ci_count, k)
        meta = dict(kernel=conv2d_kernel_name(N, H, W, Cout, ci_count, k, False, precision), splits=sp, flops=2.0 * N * H * W * k * k * ci_count * Cout,
                    shape=(N, H, W, ci_count, Cout, k, dil))
    if gate is not None:   # out is the gradient w.r.t. gate_x * gate: the launch may leave the gate-gradient partials in `part`
        groups = C.c_int(0)
        _timed("conv2d_bwd_data", meta, lambda: lib.call("mliis_conv2d_bwd_data_gate", _aptr(dy), lddy, _ptr(w), _aptr(out), lddx, N, H, W, Cin,
                                                         ci_begin, ci_count, Cout, k, dil, _ptr(buf), buf.numel(), prec, _aptr(gate),
                                                         rows_ld(gate)[2], _ptr(part), part.numel(), C.byref(groups), _dt(dy),
                                                         _dt(out, gate) | (int(out_block) << 8), _stream()))   # (out_block: out group-blocked)
        return out, groups.value
    if bn is not None:
        bx, bmean, brstd, bscale = bn
        nblk = C.c_int(0)
        _timed("conv2d_bwd_data", meta, lambda: lib.call("mliis_conv2d_bwd_data_bn", _aptr(dy), lddy, _ptr(w), _aptr(out), lddx, N, H, W, Cin,
                                                         ci_begin, ci_count, Cout, k, dil, int(accumulate), _ptr(buf), buf.numel(), prec,
                                                         _ptr(bx), rows_ld(bx)[2], _ptr(bmean), _ptr(brstd), _ptr(bscale), _ptr(part),
                                                         part.numel(), C.byref(nblk), _dt(dy), _dt(out), _stream()))
        return out, nblk.value
    _timed("conv2d_bwd_data", meta, lambda: lib.call("mliis_conv2d_bwd_data", _aptr(dy), lddy, _ptr(w), _aptr(out), lddx, N, H, W, Cin, ci_begin,
                                                     ci_count, Cout, k, dil, int(accumulate), _ptr(buf), buf.numel(), prec, _dt(dy), _dt(out),
                                                     _stream()))
    return out


def conv2d_bwd_filter(x, dy, k, dil=1, out=None, accumulate=False, ws: Optional[Workspace] = None, x_scale=None, ci_begin=0, partial=None,
                      precision="fp32"):
    """Writes rows [ci_begin, ci_begin + x.channels) of `out` ([k,k,Cin_total,Cout]; Cin_total = x.channels when out is None).
    With `partial` (float buffer of conv2d_bwd_filter_floats(...) elements) only the per-split slabs are produced there; a later
    fold_batched() call reduces them."""
    N, H, W = dy.shape[:3]
    _, Cin, ldx = rows_ld(x)
    _, Cout, lddy = rows_ld(dy)
    prec = _prec(precision)
    if partial is not None:
        lib.call("mliis_conv2d_bwd_filter", _ptr(x), ldx, _ptr(x_scale), _ptr(dy), lddy, None, N, H, W, Cin, 0, Cin, Cout, k, dil, 0,
                 _ptr(partial), partial.numel(), prec, _stream())
        return None
    out = torch.empty((k, k, Cin, Cout), dtype=torch.float32, device=x.device) if out is None else out
    Cin_total = out.shape[2]
    ws = ws or default_ws()
    buf = ws.get(lib.size("mliis_conv2d_bwd_filter_workspace_floats", N, H, W, Cin, Cout, k))
    meta = dict(flops=2.0 * N * H * W * k * k * Cin * Cout, shape=(N, H, W, Cin, Cout, k, dil)) if PROFILE is not None else {}
    _timed("conv2d_bwd_filter", meta, lambda: lib.call("mliis_conv2d_bwd_filter", _ptr(x), ldx, _ptr(x_scale), _ptr(dy), lddy, _ptr(out), N, H, W, Cin_total, ci_begin,
                                                       Cin, Cout, k,
                                                       dil, int(accumulate), _ptr(buf), buf.numel(), prec, _stream()))
    return out


class FilterBatch:
    """Deferred filter gradients (the `partial` form of conv2d_bwd_filter: slabs only) grouped by kernel instantiation: add() the
    calls of a backward pass once (the pointers are fixed for the life of a plan), then launch() issues ONE
    mliis_conv2d_bwd_filter_batched per group.  The device tables are built here, on the host, once."""

    def __init__(self, device):
        self.device = device
        self.groups = {}          # (tmf, nt, has_scale) -> list of rows
        self.tables = None
        self._keep = []           # the tensors whose addresses the tables hold

    def add(self, x, dy, k, dil, partial, x_scale=None):
        N, H, W = dy.shape[:3]
        _, Cin, ldx = rows_ld(x)
        _, Cout, lddy = rows_ld(dy)
        plan = (C.c_int * 8)()
        lib.call("mliis_conv2d_bwd_filter_plan", N, H, W, Cin, Cout, k, plan)
        tmf, nt, multitap, gx, gy, gz, rps = [int(v) for v in plan[:7]]
        if partial.numel() < gz * k * k * Cin * Cout:
            raise MliisError("FilterBatch: slab region too small")
        # (ksize word: bits 8 / 9 = X / dY stored as bf16 -- an expanded MBConv tensor under `--precision bf16-storage`)
        row = [x.data_ptr(), dy.data_ptr(), x_scale.data_ptr() if x_scale is not None else 0, partial.data_ptr(), ldx, lddy, N, H, W, Cin,
               Cout, k | (_dt(x) << 8) | (_dt(dy) << 9), dil, rps | (multitap << 32), gx | (gy << 20) | (gz << 40), 0]
        self.groups.setdefault((tmf, nt, x_scale is not None), []).append((row, gx * gy * gz, bool(multitap)))
        self.flops = getattr(self, "flops", 0.0) + 2.0 * N * H * W * k * k * Cin * Cout
        self._keep += [x, dy, partial, x_scale]
        self.tables = None

    # Small groups join the 64-column group.  The planner narrows the column tiles of a problem until ITS grid fills the chip; in a batched
    # launch the other problems' workgroups do that, and every tile width that occurs is a launch of its own: EfficientLab-6-3 at N = 8 has
    # seven native launches per step, four of them under 20 us.  A (64-channel-block, nt, gated) group with fewer than MERGE_MAX_BLOCKS
    # workgroups is re-tiled to four column tiles (the kernels mask the columns beyond Cout; slabs, pixel splits and the fold are
    # unchanged, results bit-identical) and rides in that group's launch.  Round 6, same box (profiles/r06_notes.md section 8): config 2
    # as planned 3547, widths 3 and 6 re-tiled 3574, 2 too (four launches instead of seven) 3588 images/s; EfficientNet-B3 (its width-3
    # group is thousands of workgroups of 48-column problems: padding them to 64 costs 2 %) 2079 as planned -- hence the size bound.
    # MLIIS_FB_MERGE_MAX overrides the bound (0: off).
    MERGE_MAX_BLOCKS = int(os.environ.get("MLIIS_FB_MERGE_MAX", "2500"))

    # fp32x3 launches: 256-channel workgroup tiles for the problems with more than 128 input channels (MLIIS_X3_NARROW=1: round 5's 128)
    X3_WIDE = os.environ.get("MLIIS_X3_NARROW", "0") != "1"

    def _merged_groups(self):
        groups = {k: list(v) for k, v in self.groups.items()}
        for (tmf, nt, sc) in sorted(groups):
            items = groups[(tmf, nt, sc)]
            if not (tmf == 1 and nt in (1, 2, 3, 6) and items and not any(mt for _, _, mt in items) and
                    sum(b for _, b, _ in items) < self.MERGE_MAX_BLOCKS):
                continue
            moved = []
            for row, _, mt in items:
                cout = row[10]
                gx, gz = row[14] & 0xfffff, row[14] >> 40
                gy = -(-cout // 64)
                moved.append((row[:14] + [gx | (gy << 20) | (gz << 40), 0], gx * gy * gz, mt))
            groups.setdefault((1, 4, sc), []).extend(moved)
            del groups[(tmf, nt, sc)]
        return groups

    def _build(self):
        self.tables = []
        for (tmf, nt, sc), items in sorted(self._merged_groups().items()):
            for i0 in range(0, len(items), 64):          # (the kernel scans at most 64 rows)
                rows, first = [], 0
                for row, blocks, _ in items[i0:i0 + 64]:
                    rows.append(row[:15] + [first])
                    first += blocks
                # the same problems tiled for conv_filter_x3_batched_k's 256-channel form (MLIIS_PREC_F32X3, TMF passed as 4): a problem with
                # more than 128 input channels takes gx = taps * ceil(Cin / 256) workgroups along x; slabs, pixel splits and fold unchanged
                wide = None
                if tmf == 2 and not sc and 4 <= nt <= 8 and self.X3_WIDE:
                    wrows, wfirst = [], 0
                    for row, _, _ in items[i0:i0 + 64]:
                        cin, kk, multitap = row[9], row[11] & 0xff, (row[13] >> 32) & 1
                        gx, gy, gz = row[14] & 0xfffff, (row[14] >> 20) & 0xfffff, row[14] >> 40
                        if not multitap and cin > 128:
                            gx = kk * kk * ((cin + 255) // 256)
                        wrows.append(row[:14] + [gx | (gy << 20) | (gz << 40), wfirst])
                        wfirst += gx * gy * gz
                    if wfirst != first:   # (at least one problem is tiled differently)
                        wide = (torch.tensor(wrows, dtype=torch.int64, device=self.device), wfirst)
                self.tables.append((torch.tensor(rows, dtype=torch.int64, device=self.device), len(rows), first, tmf, nt, int(sc), wide))

    def launch(self, precision="fp32"):
        """precision "fp32x3": the groups of 128-channel tiles as fp32-equivalent split products on the bf16 matrix cores
        (MLIIS_PREC_F32X3), the other groups on the fp32 instruction."""
        if self.tables is None:
            self._build()
        prec = 3 if precision == "fp32x3" else _prec(precision)
        if prec in (0, 3) and any((row[11] >> 8) for items in self.groups.values() for row, _, _ in items):
            raise MliisError("FilterBatch: bf16 tensors need the bf16-operand instances (precision 'bf16')")
        def issue():
            for table, nprob, blocks, tmf, nt, sc, wide in self.tables:
                if prec == 3 and wide is not None:
                    lib.call("mliis_conv2d_bwd_filter_batched", _ptr(wide[0]), nprob, wide[1], 4, nt, sc, prec, _stream())
                else:
                    lib.call("mliis_conv2d_bwd_filter_batched", _ptr(table), nprob, blocks, tmf, nt, sc, prec, _stream())
        _timed("conv2d_bwd_filter_batched", dict(flops=getattr(self, "flops", 0.0)) if PROFILE is not None else {}, issue)

    def __len__(self):
        return sum(len(v) for v in self.groups.values())


def rsd_concat_pool(deep, skip, cat, pool_part):
    """cat = [deep (copied, or bilinearly resized to cat's map) | skip] and the per-image column sums of cat as chunk partials in
    pool_part; returns the chunk count (feed pool_part, chunks and scale = 1 / (H W) to rsd_pool_fwd)."""
    N, H, W = cat.shape[:3]
    _, Cd, ldd = rows_ld(deep)
    _, Cs, lds = rows_ld(skip)
    ch = C.c_int(0)
    lib.call("mliis_rsd_concat_pool", _ptr(_chk(deep)), ldd, deep.shape[1], deep.shape[2], Cd, _ptr(_chk(skip)), lds, Cs, _ptr(cat), rows_ld(cat)[2],
             N, H, W, _ptr(pool_part), pool_part.numel(), C.byref(ch), _stream())
    return ch.value


def rsd_concat_pool_floats(N, H, W, C_):
    return lib.size("mliis_rsd_concat_pool_floats", N, H, W, C_)


def rsd_pool_fwd(pool, w, c_begin, out=None, chunks=1, scale=1.0, pool_out=None):
    """border-class bias [N,9,Cout] of the constant channels [c_begin, c_begin+Cp) of a 3x3 conv with weights w.  pool [N, Cp], or
    (chunks > 1) the chunk partials [N, chunks, Cp] of rsd_concat_pool with their scale; pool_out [N, Cp] keeps the folded vectors."""
    N, Cp = (pool.shape[0], pool.shape[-1]) if pool_out is None else pool_out.shape   # (the partials may arrive as a flat buffer)
    _, _, Cin_total, Co = w.shape
    out = torch.empty((N, 9, Co), dtype=torch.float32, device=pool.device) if out is None else out
    lib.call("mliis_rsd_pool_fwd", _ptr(pool), int(chunks), float(scale), _ptr(pool_out), _ptr(w), _ptr(out), N, Cp, Cin_total, c_begin, Co, _stream())
    return out


def rsd_pool_bwd(dz, tot, pool, w, c_begin, dw, dbias=None, dpool=None, ws: Optional[Workspace] = None):
    """tot [N, Co] is an OUTPUT (per-image column sums of dz, formed by the border-sum launch)."""
    N, H, W = dz.shape[:3]
    _, Co, lddz = rows_ld(dz)
    Cp = pool.shape[1]
    Cin_total = w.shape[2]
    dpool = torch.empty((N, Cp), dtype=torch.float32, device=dz.device) if dpool is None else dpool
    ws = ws or default_ws()
    buf = ws.get(lib.size("mliis_rsd_pool_bwd_workspace_floats", N, Co))
    lib.call("mliis_rsd_pool_bwd", _ptr(dz), lddz, _ptr(tot), _ptr(pool), _ptr(w), _ptr(dw), _ptr(dbias), _ptr(dpool), N, H, W, Cp, Cin_total,
             c_begin, Co, _ptr(buf), buf.numel(), _stream())
    return dpool


# ------------------------------------------------------------------------------------------------ batch norm
def bn_stats(x, pre_swish=False, moving=None, unbiased_moving_var=False, mean=None, rstd=None, eps=BN_EPS, momentum=BN_MOMENTUM,
             ws: Optional[Workspace] = None):
    rows, C_, ldx = rows_ld(x)
    mean = torch.empty(C_, dtype=torch.float32, device=x.device) if mean is None else mean
    rstd = torch.empty(C_, dtype=torch.float32, device=x.device) if rstd is None else rstd
    ws = ws or default_ws()
    buf = ws.get(lib.size("mliis_colreduce_workspace_floats", rows, C_, 1, 2))
    mm, mv = (None, None) if moving is None else moving
    lib.call("mliis_bn_stats", _ptr(x), ldx, rows, C_, int(pre_swish), eps, momentum, int(unbiased_moving_var), _ptr(mean), _ptr(rstd),
             _ptr(mm), _ptr(mv), _ptr(buf), buf.numel(), _stream())
    return mean, rstd


def bn_stats_partial(x, pre_swish, part):
    """Stage-1 statistics of x into `part` ([nblk][2][C]); returns nblk."""
    rows, C_, ldx = rows_ld(x)
    nblk = C.c_int(0)
    lib.call("mliis_bn_stats_partial", _ptr(x), ldx, rows, C_, int(pre_swish), _ptr(part), part.numel(), C.byref(nblk), _stream())
    return nblk.value


def bn_stats_partial_floats(rows, C_):
    return lib.size("mliis_colreduce_workspace_floats", rows, C_, 1, 2)


def bn_apply_fused(x, part, nblk, mean, rstd, gamma, beta, moving=None, unbiased_moving_var=False, pre_swish=False, post_swish=False,
                   img_scale=None, res=None, out=None, rows_per_img=None, eps=BN_EPS, momentum=BN_MOMENTUM, pool_part=None):
    """pool_part (a float buffer): the pass also leaves per-image partial sums of its output there and the function returns
    (out, chunks_per_image) -- feed both to se_mlp_fwd."""
    rows, C_, ldx = rows_ld(x)
    out = torch.empty(x.shape, dtype=x.dtype, device=x.device) if out is None else out
    _, _, ldy = rows_ld(out)
    rpi = rows_per_img or (rows // x.shape[0])
    ldr = rows_ld(res)[2] if res is not None else 0
    mm, mv = (None, None) if moving is None else moving
    chunks = C.c_int(0)
    lib.call("mliis_bn_apply_fused", _aptr(x), ldx, _aptr(out), ldy, rows, C_, rpi, _ptr(part), int(nblk), eps, momentum,
             int(unbiased_moving_var), _ptr(mean), _ptr(rstd), _ptr(mm), _ptr(mv), _ptr(gamma), _ptr(beta), int(pre_swish), int(post_swish),
             _ptr(img_scale), _ptr(res), ldr, _ptr(pool_part), pool_part.numel() if pool_part is not None else 0, C.byref(chunks), _dt(x, out),
             _stream())
    if pool_part is not None:
        return out, chunks.value
    return out


def bn_apply(x, mean, rstd, gamma, beta, pre_swish=False, post_swish=False, img_scale=None, res=None, out=None, rows_per_img=None):
    rows, C_, ldx = rows_ld(x)
    out = torch.empty(x.shape, dtype=torch.float32, device=x.device) if out is None else out
    _, _, ldy = rows_ld(out)
    rpi = rows_per_img or (rows // x.shape[0])
    ldr = rows_ld(res)[2] if res is not None else 0
    lib.call("mliis_bn_apply", _ptr(x), ldx, _ptr(out), ldy, rows, C_, rpi, _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta), int(pre_swish),
             int(post_swish), _ptr(img_scale), _ptr(res), ldr, _stream())
    return out


def bn_bwd(x, dy, mean, rstd, gamma, beta, pre_swish=False, post_swish=False, img_scale=None, chan_scale=None, chan_add=None, dx=None,
           dgamma=None, dbeta=None, rows_per_img=None, ws: Optional[Workspace] = None, dskip=None, dskip_accumulate=False, dxsum_part=None,
           stage1=None):
    """dskip (optional): the pass also writes dskip (+)= dy, the gradient of an identity skip around the normalised branch.
    dxsum_part (optional, bn_bwd_dxsum_floats(rows, C) floats): per-row-chunk column sums of dx (slabs for fold_batched)."""
    rows, C_, ldx = rows_ld(x)
    _, _, lddy = rows_ld(dy)
    dx = torch.empty(x.shape, dtype=x.dtype, device=x.device) if dx is None else dx
    _, _, lddx = rows_ld(dx)
    dgamma = torch.empty(C_, dtype=torch.float32, device=x.device) if dgamma is None else dgamma
    dbeta = torch.empty(C_, dtype=torch.float32, device=x.device) if dbeta is None else dbeta
    rpi = rows_per_img or (rows // x.shape[0])
    ws = ws or default_ws()
    buf = ws.get(lib.size("mliis_colreduce_workspace_floats", rows, C_, 1, 2))
    lib.call("mliis_bn_bwd", _aptr(x), ldx, _aptr(dy), lddy, _aptr(dx), lddx, rows, C_, rpi, _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta),
             int(pre_swish), int(post_swish), _ptr(img_scale), _ptr(chan_scale), _ptr(chan_add), _ptr(dgamma), _ptr(dbeta), _ptr(dskip),
             rows_ld(dskip)[2] if dskip is not None else 0, int(dskip_accumulate), _ptr(dxsum_part),
             dxsum_part.numel() if dxsum_part is not None else 0, _ptr(buf), buf.numel(),
             _ptr(stage1[0]) if stage1 else None, int(stage1[1]) if stage1 else 0, _dt(x, dy, dx), _stream())
    return dx, dgamma, dbeta


def bn_apply_fused_pair(probs, pre_swish=False, post_swish=False, unbiased_moving_var=False, eps=BN_EPS, momentum=BN_MOMENTUM):
    """Two batch norms of the same shape in one launch.  probs = 2 x (x, part, nblk, mean, rstd, gamma, beta, (moving_mean, moving_var) |
    None, out)."""
    (x0, pt0, nb0, m0, r0, g0, b0, mv0, y0), (x1, pt1, nb1, m1, r1, g1, b1, mv1, y1) = probs
    rows, C_, ldx = rows_ld(x0)
    _, _, ldy = rows_ld(y0)
    if rows_ld(x1) != (rows, C_, ldx) or rows_ld(y1) != (rows, C_, ldy):
        raise MliisError("bn_apply_fused_pair: the two problems must have the same shape and leading dimensions")
    mm0, mw0 = mv0 if mv0 is not None else (None, None)
    mm1, mw1 = mv1 if mv1 is not None else (None, None)
    lib.call("mliis_bn_apply_fused_pair", _ptr(_chk(x0)), _ptr(y0), _ptr(pt0), int(nb0), _ptr(m0), _ptr(r0), _ptr(mm0), _ptr(mw0), _ptr(g0), _ptr(b0),
             _ptr(_chk(x1)), _ptr(y1), _ptr(pt1), int(nb1), _ptr(m1), _ptr(r1), _ptr(mm1), _ptr(mw1), _ptr(g1), _ptr(b1), ldx, ldy, rows, C_,
             float(eps), float(momentum), int(unbiased_moving_var), int(pre_swish), int(post_swish), _stream())
    return y0, y1


def bn_bwd_pair(probs, pre_swish=False, post_swish=False, ws: Optional[Workspace] = None):
    """The backward of two plain batch norms of the same shape: one reduce launch + one apply launch for both.
    probs = 2 x (x, dy, mean, rstd, gamma, beta, dx, dgamma, dbeta, dxsum_part | None)."""
    (x0, dy0, m0, r0, g0, b0, dx0, dg0, db0, ds0), (x1, dy1, m1, r1, g1, b1, dx1, dg1, db1, ds1) = probs
    rows, C_, ldx = rows_ld(x0)
    _, _, lddy = rows_ld(dy0)
    _, _, lddx = rows_ld(dx0)
    if rows_ld(x1) != (rows, C_, ldx) or rows_ld(dy1) != (rows, C_, lddy) or rows_ld(dx1) != (rows, C_, lddx):
        raise MliisError("bn_bwd_pair: the two problems must have the same shape and leading dimensions")
    ws = ws or default_ws()
    buf = ws.get(lib.size("mliis_colreduce_workspace_floats", rows, C_, 2, 2))
    lib.call("mliis_bn_bwd_pair", _ptr(_chk(x0)), _ptr(dy0), _ptr(dx0), _ptr(m0), _ptr(r0), _ptr(g0), _ptr(b0), _ptr(dg0), _ptr(db0), _ptr(ds0),
             _ptr(_chk(x1)), _ptr(dy1), _ptr(dx1), _ptr(m1), _ptr(r1), _ptr(g1), _ptr(b1), _ptr(dg1), _ptr(db1), _ptr(ds1), ldx, lddy, lddx, rows, C_,
             int(pre_swish), int(post_swish), min(ds0.numel(), ds1.numel()) if ds0 is not None else 0, _ptr(buf), buf.numel(), _stream())
    return dx0, dx1


def bn_bwd_dxsum_floats(rows, C_):
    return lib.size("mliis_bn_bwd_dxsum_floats", rows, C_)


def colsum(a, b=None, nseg=1, scale=1.0, out=None, accumulate=False, ws: Optional[Workspace] = None):
    rows, C_, lda = rows_ld(a)
    ldb = rows_ld(b)[2] if b is not None else 0
    out = torch.empty((nseg, C_), dtype=torch.float32, device=a.device) if out is None else out
    ws = ws or default_ws()
    buf = ws.get(lib.size("mliis_colreduce_workspace_floats", rows // nseg, C_, nseg, 1))
    lib.call("mliis_colsum", _aptr(a), lda, _aptr(b), ldb, rows // nseg, nseg, C_, float(scale), _ptr(out), int(accumulate), _ptr(buf),
             buf.numel(), _dt(a) if b is None else _dt(a, b), _stream())
    return out


# ------------------------------------------------------------------------------------------------ squeeze-excite
def se_mlp_fwd(s, w1, b1, w2, b2, hpre=None, gate=None, chunks=0, scale=1.0, s_out=None):
    """s: the pooled vector [N, C]; or, with chunks > 0, per-image partial sums [N, chunks, C] (bn_apply_fused's pool_part) that the
    kernel folds and scales (s_out [N, C] receives the finished vector for the backward pass)."""
    if chunks > 0 and s_out is None:
        raise MliisError("se_mlp_fwd: partial sums need an s_out [N, C] tensor")
    N, C_ = s.shape if chunks == 0 else s_out.shape
    R = b1.numel()
    hpre = torch.empty((N, R), dtype=torch.float32, device=s.device) if hpre is None else hpre
    gate = torch.empty((N, C_), dtype=torch.float32, device=s.device) if gate is None else gate
    lib.call("mliis_se_mlp_fwd", _ptr(s), max(1, int(chunks)), float(scale), _ptr(s_out), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(hpre),
             _ptr(gate), N, C_, R, _stream())
    return hpre, gate


def se_wgrad_batched(desc, total_tiles):
    """Deferred squeeze-excite weight gradients of every block in one launch (desc: device int64 [n,12], include/mliis_hip.h)."""
    lib.call("mliis_se_wgrad_batched", _ptr(desc), int(desc.shape[0]), int(total_tiles), _stream())


def se_mlp_bwd(dgate, gate, s, hpre, w1, w2, hw, outs=None, dgate_groups=0, w1t=None):
    """outs without "dw1".."db2": the weight gradients are deferred to se_wgrad_batched.  dgate_groups > 0: `dgate` holds the
    per-row-group partial sums of conv2d_bwd_data(gate=...) and the kernel folds them.  w1t: w1 transposed to [R, C] (optional; the
    kernel's last phase reads it coalesced)."""
    N, C_ = s.shape
    R = hpre.shape[1]
    dev = s.device
    if outs is None:
        outs = dict(dpre1=torch.empty((N, R), device=dev), dpre2=torch.empty((N, C_), device=dev), chan_add=torch.empty((N, C_), device=dev),
                    dw1=torch.empty((1, 1, C_, R), device=dev), db1=torch.empty(R, device=dev), dw2=torch.empty((1, 1, R, C_), device=dev),
                    db2=torch.empty(C_, device=dev))
    lib.call("mliis_se_mlp_bwd", _ptr(dgate), int(dgate_groups), _ptr(gate), _ptr(s), _ptr(hpre), _ptr(w1), _ptr(w1t), _ptr(w2), _ptr(outs["dpre1"]),
             _ptr(outs["dpre2"]), _ptr(outs["chan_add"]), _ptr(outs.get("dw1")), _ptr(outs.get("db1")), _ptr(outs.get("dw2")), _ptr(outs.get("db2")), N, C_, R, hw,
             _stream())
    return outs


def se_bn_bwd_sums(z1, da2, mean, rstd, gamma, beta, part):
    """ONE pass over (da2, z1) for the squeeze-excite backward and the depthwise batch norm's backward: part [N][nblk][5][C];
    returns nblk (feed part + nblk to se_mlp_bwd_bn)."""
    N = z1.shape[0]
    rows, C_, ldx = rows_ld(z1)
    _, _, ldd = rows_ld(da2)
    nb = C.c_int(0)
    lib.call("mliis_se_bn_bwd_sums", _aptr(z1), ldx, _aptr(da2), ldd, N, rows // N, C_, _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(beta),
             _ptr(part), part.numel(), C.byref(nb), _dt(z1, da2), _stream())
    return nb.value


def se_bn_bwd_sums_floats(N, rows_per_img, C_):
    return lib.size("mliis_se_bn_bwd_sums_floats", N, rows_per_img, C_)


def se_mlp_bwd_bn(sums, nblk, gate, hpre, w1, w2, hw, outs, stage1, w1t=None):
    """se_mlp_bwd fed by se_bn_bwd_sums; also leaves stage 1 of the depthwise batch norm's backward in stage1 [N][2][C] (pass
    (stage1, N) to bn_bwd(stage1=...) together with chan_scale = gate and chan_add)."""
    N, C_ = gate.shape
    R = hpre.shape[1]
    lib.call("mliis_se_mlp_bwd_bn", _ptr(sums), int(nblk), _ptr(gate), _ptr(hpre), _ptr(w1), _ptr(w1t), _ptr(w2), _ptr(outs["dpre1"]),
             _ptr(outs["dpre2"]), _ptr(outs["chan_add"]), _ptr(stage1), N, C_, R, hw, _stream())
    return outs


def chan_affine(x, S=None, A=None, out=None, accumulate=False, rows_per_img=None, like=None):
    ref = x if x is not None else (out if out is not None else like)
    rows, C_, _ = rows_ld(ref)
    ldx = rows_ld(x)[2] if x is not None else 0
    out = torch.empty(ref.shape, dtype=torch.float32, device=ref.device) if out is None else out
    _, _, ldy = rows_ld(out)
    rpi = rows_per_img or (rows // ref.shape[0])
    lib.call("mliis_chan_affine", _ptr(x), ldx, _ptr(S), _ptr(A), _ptr(out), ldy, rows, C_, rpi, int(accumulate), _stream())
    return out


def chan_split(x, c0, out0, acc0, out1, acc1, A=None, rows_per_img=None):
    """out0 (+)= x[..., :c0] + A[n, :c0];  out1 (+)= x[..., c0:] + A[n, c0:]   (one pass; channel-slice views allowed)"""
    rows, C_, ldx = rows_ld(x)
    rpi = rows_per_img or (rows // x.shape[0])
    lib.call("mliis_chan_split", _ptr(x), ldx, _ptr(A), _ptr(out0), rows_ld(out0)[2], int(c0), int(acc0), _ptr(out1), rows_ld(out1)[2], int(acc1),
             rows, C_, rpi, _stream())


def swish_mask_fwd(z, mask=None, out=None, pre_mask=False):
    """ASPP activation: out = swish(z) * mask, or swish(z * mask) with pre_mask (mask None = inference).  Channel-slice views allowed."""
    rows, C_, ldz = rows_ld(z)
    out = torch.empty(z.shape, dtype=torch.float32, device=z.device) if out is None else out
    ldm = rows_ld(mask)[2] if mask is not None else 0
    lib.call("mliis_swish_mask_fwd", _ptr(z), ldz, _ptr(mask), ldm, _ptr(out), rows_ld(out)[2], rows, C_, int(pre_mask), _stream())
    return out


def swish_mask_bwd(dy, z, mask=None, out=None, pre_mask=False):
    rows, C_, lddy = rows_ld(dy)
    out = torch.empty(z.shape, dtype=torch.float32, device=z.device) if out is None else out
    ldm = rows_ld(mask)[2] if mask is not None else 0
    lib.call("mliis_swish_mask_bwd", _ptr(dy), lddy, _ptr(z), rows_ld(z)[2], _ptr(mask), ldm, _ptr(out), rows_ld(out)[2], rows, C_,
             int(pre_mask), _stream())
    return out


# ------------------------------------------------------------------------------------------------ resize / head / loss
def resize_bilinear_fwd(x, out_hw, out=None):
    N, Hi, Wi = x.shape[:3]
    _, C_, ldx = rows_ld(x)
    Ho, Wo = out_hw
    out = torch.empty((N, Ho, Wo, C_), dtype=torch.float32, device=x.device) if out is None else out
    _, _, ldy = rows_ld(out)
    lib.call("mliis_resize_bilinear_fwd", _ptr(x), ldx, _ptr(out), ldy, N, Hi, Wi, Ho, Wo, C_, _stream())
    return out


def resize_bilinear_bwd(dy, in_hw, out=None, accumulate=False):
    N, Ho, Wo = dy.shape[:3]
    _, C_, lddy = rows_ld(dy)
    Hi, Wi = in_hw
    out = torch.empty((N, Hi, Wi, C_), dtype=torch.float32, device=dy.device) if out is None else out
    _, _, lddx = rows_ld(out)
    lib.call("mliis_resize_bilinear_bwd", _ptr(dy), lddy, _ptr(out), lddx, N, Hi, Wi, Ho, Wo, C_, int(accumulate), _stream())
    return out


def final_conv_fwd(x, w, b, mask=None, out=None):
    rows, C_, ldx = rows_ld(x)
    out = torch.empty(tuple(x.shape[:-1]) + (2,), dtype=torch.float32, device=x.device) if out is None else out
    lib.call("mliis_final_conv_fwd", _ptr(x), ldx, _ptr(mask), _ptr(w), _ptr(b), _ptr(out), rows, C_, _stream())
    return out


def final_conv_bwd_data(dy, w, C_, mask=None, out=None, fin=None):
    """fin = (partials buffer of head_ce_fused(finalize=False), image size (H, W), extra_loss, loss_out): the loss fold rides in this launch."""
    rows = dy.numel() // 2
    out = torch.empty(tuple(dy.shape[:-1]) + (C_,), dtype=torch.float32, device=dy.device) if out is None else out
    _, _, lddx = rows_ld(out)
    if fin is not None:
        buf, size, extra_loss, loss_out = fin
        N, Hd, Wd = dy.shape[:3]
        lib.call("mliis_final_conv_bwd_data_fin", _ptr(dy), _ptr(w), _ptr(mask), _ptr(out), lddx, rows, C_, _ptr(buf), N, Hd, Wd, int(size[0]), int(size[1]),
                 float(extra_loss), _ptr(loss_out), _stream())
        return out
    lib.call("mliis_final_conv_bwd_data", _ptr(dy), _ptr(w), _ptr(mask), _ptr(out), lddx, rows, C_, _stream())
    return out


def final_conv_bwd_filter(x, dy, mask=None, dw=None, db=None, ws: Optional[Workspace] = None):
    rows, C_, ldx = rows_ld(x)
    dw = torch.empty((1, 1, C_, 2), dtype=torch.float32, device=x.device) if dw is None else dw
    db = torch.empty(2, dtype=torch.float32, device=x.device) if db is None else db
    ws = ws or default_ws()
    buf = ws.get(lib.size("mliis_colreduce_workspace_floats", rows, C_, 1, 2))
    lib.call("mliis_final_conv_bwd_filter", _ptr(x), ldx, _ptr(mask), _ptr(dy), rows, C_, _ptr(dw), _ptr(db), _ptr(buf), buf.numel(), _stream())
    return dw, db


def softmax_ce(logits, labels, idx=None, label_smoothing=0.0, dice=False, extra_loss=0.0, want_grad=True, want_pred=False, dlogits=None,
               pred=None, out=None, ws: Optional[Workspace] = None):
    N, H, W, _ = logits.shape
    dev = logits.device
    if want_grad and dlogits is None:
        dlogits = torch.empty_like(logits)
    if want_pred and pred is None:
        pred = torch.empty_like(logits)
    out = torch.empty(4, dtype=torch.float32, device=dev) if out is None else out
    ws = ws or default_ws()
    buf = ws.get(lib.size("mliis_softmax_ce_workspace_floats", N, H, W))
    lib.call("mliis_softmax_ce", _ptr(logits), _ptr(labels), _ptr(idx), N, H, W, float(label_smoothing), int(dice), float(extra_loss),
             _ptr(dlogits) if want_grad else None, _ptr(pred) if want_pred else None, _ptr(out), _ptr(buf), buf.numel(), _stream())
    return out, dlogits, pred


def head_ce_fused(small, labels, idx, size, label_smoothing, dsmall, out, extra_loss=0.0, ws: Optional[Workspace] = None, finalize=True):
    """resize(small -> size) -> softmax cross-entropy (no dice term) -> gradient -> resize^T: dsmall and out[0..2] = {loss, ce, iou};
    the full-resolution logits are never written (two launches instead of five).  finalize=False: ONE launch -- the loss partials stay
    in the workspace buffer (returned third; nothing else may use it meanwhile) for final_conv_bwd_data(fin=...) to fold."""
    N, Hd, Wd, _ = small.shape
    ws = ws or default_ws()
    buf = ws.get(lib.size("mliis_head_ce_fused_workspace_floats", N, Hd, Wd))
    _timed("head_ce_fused", {}, lambda: lib.call("mliis_head_ce_fused", _ptr(_chk(small)), _ptr(labels), _ptr(idx), N, Hd, Wd, int(size[0]), int(size[1]),
                                                 float(label_smoothing), float(extra_loss), _ptr(dsmall), _ptr(out) if finalize else None, _ptr(buf),
                                                 buf.numel(), _stream()))
    return (out, dsmall) if finalize else (out, dsmall, buf)


def darc1(logits, weight, dlogits=None, out=None, ws: Optional[Workspace] = None):
    """out[0] += weight * max_pos sum_n |logits[n, pos]|;  dlogits += its gradient (models/regularizers.py:20-22)."""
    N = logits.shape[0]
    ws = ws or default_ws()
    buf = ws.get(2048)
    lib.call("mliis_darc1", _ptr(_chk(logits)), N, logits.numel() // N, float(weight), _ptr(dlogits), _ptr(out), _ptr(buf), buf.numel(), _stream())


def fold_batched(part_base, out_base, desc, total_tiles, se_desc=None, se_tiles=0):
    """One launch folding every deferred weight-gradient slab set (desc: device int64 [n,8], see include/mliis_hip.h).  se_desc /
    se_tiles (the arguments of se_wgrad_batched): the squeeze-excite weight gradients ride in the same launch."""
    lib.call("mliis_fold_batched", _ptr(part_base), _ptr(out_base), _ptr(desc), int(desc.shape[0]), int(total_tiles),
             _ptr(se_desc), int(se_desc.shape[0]) if se_desc is not None else 0, int(se_tiles) if se_desc is not None else 0, _stream())


# ------------------------------------------------------------------------------------------------ device RNG (masks)
class MaskPlan:
    """Argument block of one mliis_rng_masks launch (built once per plan: the host arrays must outlive every launch / graph capture).
    jobs: list of (out tensor | None, keep probability | device tensor of per-row keep probabilities, row_len, floor_form)."""

    def __init__(self, jobs):
        n = len(jobs)
        self.n = n
        self.outs = (C.c_void_p * n)(*[_ptr(j[0]) for j in jobs])
        self.numels = (C.c_longlong * n)(*[int(j[0].numel()) if j[0] is not None else 0 for j in jobs])
        self.keep = (C.c_float * n)(*[float(j[1]) if not torch.is_tensor(j[1]) else 1.0 for j in jobs])
        self.keeps = (C.c_void_p * n)(*[_ptr(j[1]) if torch.is_tensor(j[1]) else None for j in jobs])
        self.row_len = (C.c_int * n)(*[int(j[2]) for j in jobs])
        self.floor_form = (C.c_int * n)(*[int(bool(j[3])) for j in jobs])
        self._keepalive = jobs


def rng_state(seed: int, device) -> torch.Tensor:
    """Device uint32[4] {seed lo, seed hi, step = 0, 0} (stored as int32 bits)."""
    s = int(seed) & 0xFFFFFFFFFFFFFFFF
    lo, hi = s & 0xFFFFFFFF, s >> 32
    to_i32 = lambda v: v - (1 << 32) if v >= (1 << 31) else v  # noqa: E731
    return torch.tensor([to_i32(lo), to_i32(hi), 0, 0], dtype=torch.int32, device=device)


def rng_masks(state: torch.Tensor, plan: MaskPlan):
    lib.call("mliis_rng_masks", _ptr(state), plan.n, plan.outs, plan.numels, plan.keep, plan.keeps, plan.row_len, plan.floor_form, _stream())


# ------------------------------------------------------------------------------------------------ optimizer / arena
def sgd_fused(w, g, lr, l2_quad_mask=None, l2=0.0, lr_dev=None, l1=0.0):
    lib.call("mliis_sgd_fused", _ptr(w), _ptr(g), _ptr(l2_quad_mask), w.numel(), float(lr), _ptr(lr_dev), float(l2), float(l1), _stream())


def adam_b1zero_fused(w, g, v, step_dev, lr, l2_quad_mask=None, l2=0.0, lr_dev=None, beta2=0.999, eps=1e-8, l1=0.0, ticket=None):
    """step_dev: device float, steps applied so far.  ticket None: the caller advanced it before the call; ticket (a zeroed device
    int32): the launch is step step_dev + 1 and advances the count itself (graph-replay safe)."""
    lib.call("mliis_adam_b1zero_fused", _ptr(w), _ptr(g), _ptr(v), _ptr(l2_quad_mask), w.numel(), float(lr), _ptr(lr_dev), float(l2), float(l1),
             float(beta2), float(eps), _ptr(step_dev), _ptr(ticket), _stream())


def axpby(a, x, b, y):
    lib.call("mliis_axpby", float(a), _ptr(x), float(b), _ptr(y), y.numel(), _stream())


def copy_words(src, dst):
    """dst <- src (<= 1024 32-bit words) by one small kernel on the current stream; src may be a pinned HOST tensor."""
    lib.call("mliis_copy_words", C.c_void_p(src.data_ptr()), _ptr(dst), src.numel(), _stream())


def lincomb(a, x, b, y, out):
    lib.call("mliis_lincomb", float(a), _ptr(x), float(b), _ptr(y), _ptr(out), out.numel(), _stream())
