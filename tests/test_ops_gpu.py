"""Per-kernel parity: every C-ABI entry point (called through mliis_amd.ops) against the float64 CPU oracle ops
(oracle/efficientlab_ref.py + autograd) on the same seeded inputs.  Tolerances: forward rel 2e-5, backward rel 1e-4
(fp32 kernels vs fp64 oracle), normalised by the tensor's max-abs."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import efficientlab_ref as R  # noqa: E402


def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g, dtype=torch.float64) * scale)


def close(got, ref, tol, what=""):
    got = got.detach().cpu().double()
    ref = ref.detach().cpu().double()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    den = max(ref.abs().max().item(), 1e-30)
    err = (got - ref).abs().max().item() / den
    assert err <= tol, "{}: rel err {:.3e} > {:.1e}".format(what, err, tol)


def nchw(t):
    return t.permute(0, 3, 1, 2)


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def f32(t, d):
    return t.float().contiguous().to(d)


# ------------------------------------------------------------------------------------------------ depthwise
@pytest.mark.parametrize("k,s,H,W,C", [(3, 1, 14, 14, 32), (3, 2, 16, 16, 24), (5, 1, 14, 14, 40), (5, 2, 28, 28, 16),
                                       (3, 2, 15, 17, 8), (5, 2, 9, 11, 8), (3, 1, 5, 3, 4), (5, 1, 7, 30, 144)])
def test_dwconv_all(k, s, H, W, C):
    _dwconv_case(k, s, H, W, C, 2)


# every depthwise layer of EfficientLab-6-3 at the BASELINE config-2 batch (N = 8, 224x224 input): blocks 0-5 are the large-map kernels,
# blocks 6-10 the shapes the op-by-op path still serves at evaluation time (training uses the fused small-map kernels, tested below)
@pytest.mark.parametrize("k,s,H,C", [(3, 1, 112, 32), (3, 2, 112, 96), (3, 1, 56, 144), (5, 2, 56, 144), (5, 1, 28, 240), (3, 2, 28, 240),
                                     (3, 1, 14, 480), (5, 1, 14, 480), (5, 1, 14, 672)])
def test_dwconv_baseline_shapes(k, s, H, C):
    _dwconv_case(k, s, H, H, C, 8, part_floats=1 << 20)


def _dwconv_case(k, s, H, W, C, N, part_floats=1 << 16):
    from mliis_amd import ops
    d = dev()
    x = rnd(N, H, W, C, seed=1).requires_grad_(True)
    w = rnd(k, k, C, 1, seed=2).requires_grad_(True)
    y = R.conv2d_same(nchw(x), w, s, groups=C)
    dy = rnd(*y.shape, seed=3)
    gx, gw = torch.autograd.grad(y, [x, w], dy)
    yg = ops.dwconv_fwd(f32(x, d), f32(w, d), s)
    close(yg, nhwc(y), 2e-5, "dw fwd")
    # training variant: same output plus the next batch norm's stage-1 statistics
    part = torch.full((part_floats,), 7.0, device=d)
    ys, nblk = ops.dwconv_fwd(f32(x, d), f32(w, d), s, stats_part=part)
    assert torch.equal(ys, yg) and nblk > 0
    sums = part[: nblk * 2 * C].view(nblk, 2, C).double().sum(0).cpu()
    yr = nhwc(y).detach()
    close(sums[0], yr.sum(dim=(0, 1, 2)), 1e-5, "dw fused sum")
    close(sums[1], (yr * yr).sum(dim=(0, 1, 2)), 1e-5, "dw fused sum of squares")
    dyg = f32(nhwc(dy), d)
    close(ops.dwconv_bwd_data(dyg, f32(w, d), s, (H, W)), gx, 1e-4, "dw bwd data")
    close(ops.dwconv_bwd_filter(f32(x, d), dyg, k, s), gw, 1e-4, "dw bwd filter")


# ------------------------------------------------------------------------------------------------ dense conv
@pytest.mark.parametrize("k,dil,H,W,Cin,Cout,N", [
    (1, 1, 8, 8, 16, 96, 2), (1, 1, 7, 9, 96, 24, 3), (1, 1, 14, 14, 40, 240, 2), (1, 1, 4, 4, 672, 112, 2),
    (3, 1, 14, 14, 24, 112, 2), (3, 2, 14, 14, 136, 112, 2), (3, 1, 9, 11, 360, 112, 1), (3, 6, 14, 14, 112, 112, 1),
    (1, 1, 56, 56, 24, 144, 4), (3, 2, 6, 5, 8, 8, 1),
    # narrow 3x3 convs: a 32-wide K chunk of the flattened (tap, channel) index crosses up to 8 taps
    (3, 1, 10, 10, 4, 8, 2), (3, 2, 9, 9, 12, 20, 1), (3, 1, 8, 8, 20, 4, 2), (3, 3, 8, 8, 28, 16, 1),
])
def test_conv2d_all(k, dil, H, W, Cin, Cout, N):
    _conv2d_case(k, dil, H, W, Cin, Cout, N)


# the dense convs of BASELINE config 2 at their real sizes (N = 8, 224x224 input): MBConv expand / project convs of the large maps
# (streaming kernel, 128-row tiles), a 14x14 pair (split-K), the RSD(4) convs and the three stream-K decoder launches of RSD(2)
@pytest.mark.parametrize("k,dil,H,Cin,Cout", [(1, 1, 112, 16, 96), (1, 1, 112, 32, 16), (1, 1, 56, 144, 24), (1, 1, 28, 240, 40),
                                               (1, 1, 14, 80, 480), (1, 1, 14, 672, 112), (1, 1, 14, 224, 112), (3, 2, 14, 224, 112),
                                               (1, 1, 56, 136, 112), (3, 2, 56, 136, 112), (3, 1, 56, 224, 112)])
def test_conv2d_baseline_shapes(k, dil, H, Cin, Cout):
    _conv2d_case(k, dil, H, H, Cin, Cout, 8)


def _conv2d_case(k, dil, H, W, Cin, Cout, N):
    from mliis_amd import ops
    d = dev()
    x = rnd(N, H, W, Cin, seed=4).requires_grad_(True)
    w = rnd(k, k, Cin, Cout, seed=5, scale=1.0 / math.sqrt(k * k * Cin)).requires_grad_(True)
    b = rnd(Cout, seed=6).requires_grad_(True)
    y = R.conv2d_same(nchw(x), w, 1, dil, bias=b)
    dy = rnd(*y.shape, seed=7)
    gx, gw, gb = torch.autograd.grad(y, [x, w, b], dy)
    xg, wg, bg = f32(x, d), f32(w, d), f32(b, d)
    close(ops.conv2d_fwd(xg, wg, bg, dil), nhwc(y), 2e-5, "conv fwd")   # (builds the K-contiguous weight copy itself)
    # forward through a slot of a K-contiguous shadow arena (batched HWIO -> HWOI transpose)
    wt = torch.full((wg.numel() + 8,), float("nan"), device=d)
    desc = torch.tensor([[4, k * k, Cin, Cout]], dtype=torch.int32, device=d)
    src = torch.zeros(wg.numel() + 8, device=d)
    src[4:4 + wg.numel()] = wg.reshape(-1)
    ops.transpose_weights(src, wt, desc)
    assert torch.equal(wt[4:4 + wg.numel()].view(k, k, Cout, Cin), wg.permute(0, 1, 3, 2)), "shadow weights"
    assert torch.isnan(wt[:4]).all() and torch.isnan(wt[4 + wg.numel():]).all()
    close(ops.conv2d_fwd(xg, wg, bg, dil, wt=wt[4:4 + wg.numel()]), nhwc(y), 2e-5, "conv fwd (shadow weights)")
    dyg = f32(nhwc(dy), d)
    close(ops.conv2d_bwd_data(dyg, wg, dil), gx, 1e-4, "conv bwd data")
    close(ops.conv2d_bwd_filter(xg, dyg, k, dil), gw, 1e-4, "conv bwd filter")
    close(ops.colsum(dyg), gb[None], 1e-4, "bias grad")


def test_conv2d_slices_and_accumulate():
    """Channel-sliced input/output views of concat buffers, partial-input-channel bwd-data, accumulate flags."""
    from mliis_amd import ops
    d = dev()
    N, H, W = 2, 10, 10
    cat = rnd(N, H, W, 48, seed=8)
    w = rnd(3, 3, 24, 16, seed=9, scale=0.1)
    xs = cat[..., 16:40]
    y = R.conv2d_same(nchw(xs), w, 1, 2)
    catg = f32(cat, d)
    outbuf = torch.zeros(N, H, W, 40, device=d)
    ops.conv2d_fwd(catg[..., 16:40], f32(w, d), None, 2, out=outbuf[..., 8:24])
    close(outbuf[..., 8:24], nhwc(y), 2e-5, "sliced fwd")
    assert outbuf[..., :8].abs().max().item() == 0 and outbuf[..., 24:].abs().max().item() == 0
    ops.conv2d_fwd(catg[..., 16:40], f32(w, d), None, 2, out=outbuf[..., 8:24], accumulate=True)
    close(outbuf[..., 8:24], 2 * nhwc(y), 2e-5, "accumulate fwd")
    # bwd-data for input channels [8, 20) only
    dy = rnd(N, H, W, 16, seed=10)
    xr = xs.clone().requires_grad_(True)
    (gx,) = torch.autograd.grad(R.conv2d_same(nchw(xr), w, 1, 2), [xr], nchw(dy))
    got = ops.conv2d_bwd_data(f32(dy, d), f32(w, d), 2, ci_begin=8, ci_count=12)
    close(got, gx[..., 8:20], 1e-4, "partial bwd data")


# ------------------------------------------------------------------------------------------------ stem
@pytest.mark.parametrize("H,W,Co", [(16, 16, 32), (15, 13, 32), (8, 8, 40), (224, 224, 32), (131, 67, 40), (66, 130, 8)])
def test_stem(H, W, Co):
    from mliis_amd import ops
    d = dev()
    S, idx = 4, [2, 0, 3, 2, 1]
    g = torch.Generator().manual_seed(11)
    x = torch.randint(0, 256, (S, H, W, 3), generator=g).double()
    w = rnd(3, 3, 3, Co, seed=12).requires_grad_(True)
    xn = (x[idx] - torch.tensor(R.MEAN, dtype=torch.float64)) / torch.tensor(R.STD, dtype=torch.float64)
    z = R.conv2d_same(nchw(xn), w, 2)
    dz = rnd(*z.shape, seed=13)
    (gw,) = torch.autograd.grad(z, [w], dz)
    ig = torch.tensor(idx, dtype=torch.int32, device=d)
    zd = ops.stem_conv_fwd(f32(x, d), f32(w, d), ig)
    close(zd, nhwc(z), 2e-5, "stem fwd")
    close(ops.stem_conv_bwd_filter(f32(x, d), f32(nhwc(dz), d), ig), gw, 1e-4, "stem bwd filter")
    # the row-strip kernel (a training step's stem launch): z bit-identical to the plain kernel's, and the stage-1 statistics of z
    # ([nblk][2][Co] partial sums / sums of squares) fold to the sums of that z
    part = torch.full((ops.stem_conv_fwd_stats_floats(len(idx), H, W, Co) + 8,), float("nan"), device=d)
    z2, nblk = ops.stem_conv_fwd(f32(x, d), f32(w, d), ig, stats_part=part)
    assert nblk > 0 and nblk * 2 * Co <= part.numel() - 8 and torch.isnan(part[nblk * 2 * Co:]).all()
    assert torch.equal(z2, zd), "row-strip stem kernel differs from the plain one"
    folded = part[:nblk * 2 * Co].view(nblk, 2, Co).double().sum(0).cpu()
    zz = zd.double().cpu().reshape(-1, Co)
    close(folded[0], zz.sum(0), 1e-5, "stem statistics: sum")
    close(folded[1], (zz * zz).sum(0), 1e-5, "stem statistics: sum of squares")
    z3, nb3 = ops.stem_conv_fwd(f32(x, d), f32(w, d), ig, rows=True)
    assert nb3 == 0 and torch.equal(z3, zd)


def test_stem_rows_wider_than_the_staging_window_take_the_plain_kernel():
    """mliis_stem_conv_fwd_stats stages nine input rows in LDS: rows wider than ~400 pixels do not fit -- the C entry refuses (nothing
    launched), the wrapper runs the plain kernel and reports -1 so that the caller takes the statistics launch."""
    import ctypes as C
    from mliis_amd import ops
    d = dev()
    x = torch.randint(0, 256, (1, 6, 900, 3), generator=torch.Generator().manual_seed(3)).float().to(d)
    w = f32(rnd(3, 3, 3, 32, seed=12), d)
    part = torch.zeros(ops.stem_conv_fwd_stats_floats(1, 6, 900, 32) + 8, device=d)
    z, nblk = ops.stem_conv_fwd(x, w, None, stats_part=part)
    assert nblk == -1 and torch.equal(z, ops.stem_conv_fwd(x, w, None))
    with pytest.raises(Exception):
        ops.lib.call("mliis_stem_conv_fwd_stats", C.c_void_p(x.data_ptr()), None, C.c_void_p(w.data_ptr()), C.c_void_p(z.data_ptr()), 1, 6, 900, 32,
                     ops._MEAN3, ops._STD3, C.c_void_p(part.data_ptr()), part.numel(), None, C.c_void_p(torch.cuda.current_stream().cuda_stream))


# ------------------------------------------------------------------------------------------------ batch norm
@pytest.mark.parametrize("pre,post,C,rows_hw,N", [(0, 1, 32, 36, 3), (0, 0, 24, 49, 2), (1, 0, 112, 25, 2), (0, 1, 672, 9, 2), (0, 1, 8, 1000, 2),
                                                  (0, 1, 32, 1500, 3), (1, 0, 40, 3001, 2), (0, 1, 672, 196, 8), (0, 0, 20, 832, 2), (1, 0, 20, 833, 2),
                                                  # BASELINE config-2 sizes: block-1 expand BN (112x112x96), a project BN, the RSD(2) branch BN
                                                  (0, 1, 96, 12544, 8), (0, 0, 24, 3136, 8), (1, 0, 112, 3136, 8)])
def test_bn_train_fwd_bwd(pre, post, C, rows_hw, N):
    from mliis_amd import ops
    d = dev()
    x = (rnd(N, rows_hw, 1, C, seed=14) * 2 + 0.5).requires_grad_(True)
    gamma = (rnd(C, seed=15) * 0.5 + 1).requires_grad_(True)
    beta = rnd(C, seed=16).requires_grad_(True)
    res = rnd(N, rows_hw, 1, C, seed=17)
    img_scale = torch.tensor(([0.0, 1.25, 1.25] * 3)[:N], dtype=torch.float64)
    chan_scale = rnd(N, C, seed=18)
    chan_add = rnd(N, C, seed=19) * 0.1
    mm0, mv0 = rnd(C, seed=20), rnd(C, seed=21).abs() + 0.5
    xin = R.swish(x) if pre else x
    mean = xin.mean(dim=(0, 1, 2))
    var = ((xin - mean) ** 2).mean(dim=(0, 1, 2))
    xhat = (xin - mean) * torch.rsqrt(var + 1e-3)
    u = xhat * gamma + beta
    bn_out = R.swish(u) if post else u
    y = bn_out * img_scale[:, None, None, None] + res
    dy = rnd(*y.shape, seed=22)
    # upstream of the BN output inside the kernel: dy*img_scale*chan_scale + chan_add
    up = dy * img_scale[:, None, None, None] * chan_scale[:, None, None, :] + chan_add[:, None, None, :]
    gx, gg, gb = torch.autograd.grad(bn_out, [x, gamma, beta], up)
    xg = f32(x, d)
    mm, mv = f32(mm0, d), f32(mv0, d)
    m, r = ops.bn_stats(xg, bool(pre), moving=(mm, mv), unbiased_moving_var=bool(pre))
    close(m, mean, 1e-5, "mean")
    close(r, torch.rsqrt(var + 1e-3), 1e-5, "rstd")
    n = N * rows_hw
    close(mm, mm0 - (mm0 - mean) * 0.01, 1e-5, "moving mean")
    close(mv, mv0 - (mv0 - var * (n / (n - 1.0) if pre else 1.0)) * 0.01, 1e-5, "moving var")
    yg = ops.bn_apply(xg, m, r, f32(gamma, d), f32(beta, d), bool(pre), bool(post), f32(img_scale, d), f32(res, d))
    close(yg, y, 2e-5, "bn apply")
    dx, dg, db = ops.bn_bwd(xg, f32(dy, d), m, r, f32(gamma, d), f32(beta, d), bool(pre), bool(post), f32(img_scale, d),
                            f32(chan_scale, d), f32(chan_add, d))
    close(dx, gx, 1e-4, "bn dx")
    close(dg, gg, 1e-4, "bn dgamma")
    close(db, gb, 1e-4, "bn dbeta")
    # identity-skip gradient written by the same pass (fresh and accumulating), dx unchanged
    skip0 = rnd(*y.shape, seed=23)
    for acc in (False, True):
        sk = f32(skip0, d)
        dx2, _, _ = ops.bn_bwd(xg, f32(dy, d), m, r, f32(gamma, d), f32(beta, d), bool(pre), bool(post), f32(img_scale, d),
                               f32(chan_scale, d), f32(chan_add, d), dskip=sk, dskip_accumulate=acc)
        assert torch.equal(dx2, dx)
        close(sk, dy + skip0 if acc else dy, 1e-6, "skip gradient acc={}".format(acc))
    # column sums of dx as per-row-chunk slabs (conv-bias gradient of a conv -> swish -> BN stack)
    nfl = ops.bn_bwd_dxsum_floats(n, C)
    slab = torch.full((nfl + 8,), 5.0, device=d)
    dx3, _, _ = ops.bn_bwd(xg, f32(dy, d), m, r, f32(gamma, d), f32(beta, d), bool(pre), bool(post), f32(img_scale, d),
                           f32(chan_scale, d), f32(chan_add, d), dxsum_part=slab)
    assert torch.equal(dx3, dx) and slab[nfl:].eq(5.0).all()
    got = slab[:nfl].view(-1, C).double().sum(0).cpu()
    scale = gx.abs().sum(dim=(0, 1, 2)).max().item()          # (without pre-swish the true column sums are ~0: compare on the sum's scale)
    assert (got - gx.sum(dim=(0, 1, 2))).abs().max().item() <= 1e-5 * scale, "dx column sums"
    # fused forward: statistics from bn_stats_partial folded inside the apply launch
    for _ in range(1):
        mm2, mv2 = f32(mm0, d), f32(mv0, d)
        m2, r2 = torch.empty(C, device=d), torch.empty(C, device=d)
        part = torch.empty(ops.bn_stats_partial_floats(n, C) + 16, device=d)
        nblk = ops.bn_stats_partial(xg, bool(pre), part)
        y2 = ops.bn_apply_fused(xg, part, nblk, m2, r2, f32(gamma, d), f32(beta, d), moving=(mm2, mv2), unbiased_moving_var=bool(pre),
                                pre_swish=bool(pre), post_swish=bool(post), img_scale=f32(img_scale, d), res=f32(res, d))
        close(y2, y, 2e-5, "fused apply")
        close(m2, mean, 1e-5, "fused mean")
        close(r2, torch.rsqrt(var + 1e-3), 1e-5, "fused rstd")
        close(mm2, mm0 - (mm0 - mean) * 0.01, 1e-5, "fused moving mean")
        close(mv2, mv0 - (mv0 - var * (n / (n - 1.0) if pre else 1.0)) * 0.01, 1e-5, "fused moving var")


def test_colsum_segments_and_product():
    from mliis_amd import ops
    d = dev()
    a, b = rnd(3, 50, 1, 24, seed=23), rnd(3, 50, 1, 24, seed=24)
    close(ops.colsum(f32(a, d), f32(b, d), nseg=3, scale=0.5), 0.5 * (a * b).sum(dim=(1, 2)), 1e-5, "colsum prod seg")
    out = torch.ones(3, 24, device=d)
    ops.colsum(f32(a, d), None, nseg=3, out=out, accumulate=True)
    close(out, 1 + a.sum(dim=(1, 2)), 1e-5, "colsum accumulate")


# ------------------------------------------------------------------------------------------------ squeeze-excite
@pytest.mark.parametrize("C,Rr,N,HW", [(32, 8, 3, 16), (96, 4, 2, 9), (672, 28, 2, 4), (144, 6, 5, 49), (1536, 64, 2, 16), (1000, 34, 2, 4)])
def test_se(C, Rr, N, HW):
    from mliis_amd import ops
    d = dev()
    x = rnd(N, HW, 1, C, seed=25).requires_grad_(True)
    w1 = rnd(1, 1, C, Rr, seed=26, scale=0.3).requires_grad_(True)
    b1 = rnd(Rr, seed=27).requires_grad_(True)
    w2 = rnd(1, 1, Rr, C, seed=28, scale=0.3).requires_grad_(True)
    b2 = rnd(C, seed=29).requires_grad_(True)
    s = x.mean(dim=(1, 2))
    h = s @ w1[0, 0] + b1
    gate = torch.sigmoid(R.swish(h) @ w2[0, 0] + b2)
    y = x * gate[:, None, None, :]
    dy = rnd(*y.shape, seed=30)
    gx, gw1, gb1, gw2, gb2 = torch.autograd.grad(y, [x, w1, b1, w2, b2], dy)
    xg, dyg = f32(x, d), f32(dy, d)
    sg = ops.colsum(xg, None, nseg=N, scale=1.0 / HW)
    close(sg, s, 1e-5, "se pool")
    hp, gg = ops.se_mlp_fwd(sg, f32(w1, d), f32(b1, d), f32(w2, d), f32(b2, d))
    close(gg, gate, 2e-5, "se gate")
    close(ops.chan_affine(xg, S=gg), y, 2e-5, "se scale")
    dgate = ops.colsum(dyg, xg, nseg=N)
    o = ops.se_mlp_bwd(dgate, gg, sg, hp, f32(w1, d), f32(w2, d), HW)
    ot = ops.se_mlp_bwd(dgate, gg, sg, hp, f32(w1, d), f32(w2, d), HW, w1t=f32(w1, d)[0, 0].t().contiguous())   # coalesced column reads
    for k in o:
        assert torch.equal(o[k], ot[k]), "se_mlp_bwd with the transposed w1: " + k
    close(o["dw1"], gw1, 1e-4, "dw1")
    close(o["db1"], gb1, 1e-4, "db1")
    close(o["dw2"], gw2, 1e-4, "dw2")
    close(o["db2"], gb2, 1e-4, "db2")
    dx = ops.chan_affine(dyg, S=gg, A=o["chan_add"])
    close(dx, gx, 1e-4, "se dx")


@pytest.mark.parametrize("rows,C", [(1568, 112), (25088, 112), (300, 136), (77, 8)])
def test_batch_norm_pairs_equal_the_single_launches(rows, C):
    """mliis_bn_apply_fused_pair / mliis_bn_bwd_pair (the two independent branch batch norms of an RSD module in one launch per pass)
    against two single launches: forward bit-identical (outputs into channel slices of a shared buffer, statistics, moving averages),
    backward to fp32 rounding (gradients, bias-gradient slabs)."""
    from mliis_amd import ops
    d = dev()
    xs = [f32(rnd(rows, C, seed=80 + i) * (1 + i) + 0.3 * i, d) for i in range(2)]
    gam = [f32(rnd(C, seed=82 + i) * 0.3 + 1, d) for i in range(2)]
    bet = [f32(rnd(C, seed=84 + i), d) for i in range(2)]
    parts, nbs = [], []
    for i in range(2):
        pt = torch.empty(ops.bn_stats_partial_floats(rows, C) + 8, device=d)
        nbs.append(ops.bn_stats_partial(xs[i], True, pt))
        parts.append(pt)

    def fresh():
        return ([torch.zeros(C, device=d) for _ in range(2)], [torch.zeros(C, device=d) for _ in range(2)],
                [f32(rnd(C, seed=86 + i), d) for i in range(2)], [f32(rnd(C, seed=88 + i).abs() + 0.5, d) for i in range(2)],
                torch.zeros(rows, 2 * C, device=d))
    m1, r1, mm1, mv1, pyr1 = fresh()
    for i in range(2):
        ops.bn_apply_fused(xs[i], parts[i], nbs[i], m1[i], r1[i], gam[i], bet[i], moving=(mm1[i], mv1[i]), unbiased_moving_var=True,
                           pre_swish=True, out=pyr1[:, i * C:(i + 1) * C])
    m2, r2, mm2, mv2, pyr2 = fresh()
    ops.bn_apply_fused_pair([(xs[i], parts[i], nbs[i], m2[i], r2[i], gam[i], bet[i], (mm2[i], mv2[i]), pyr2[:, i * C:(i + 1) * C]) for i in range(2)],
                            pre_swish=True, unbiased_moving_var=True)
    assert torch.equal(pyr1, pyr2)
    for a_, b_ in zip(m1 + r1 + mm1 + mv1, m2 + r2 + mm2 + mv2):
        assert torch.equal(a_, b_)
    # backward, in place on channel slices of a shared gradient buffer
    dy = f32(rnd(rows, 2 * C, seed=90), d)
    nsl = ops.bn_bwd_dxsum_floats(rows, C)
    res = []
    for pair in (False, True):
        dp = dy.clone()
        dg, db = [torch.zeros(C, device=d) for _ in range(2)], [torch.zeros(C, device=d) for _ in range(2)]
        sl = [torch.zeros(nsl, device=d) for _ in range(2)]
        if pair:
            ops.bn_bwd_pair([(xs[i], dp[:, i * C:(i + 1) * C], m1[i], r1[i], gam[i], bet[i], dp[:, i * C:(i + 1) * C], dg[i], db[i], sl[i])
                             for i in range(2)], pre_swish=True)
        else:
            for i in range(2):
                v = dp[:, i * C:(i + 1) * C]
                ops.bn_bwd(xs[i], v, m1[i], r1[i], gam[i], bet[i], pre_swish=True, dx=v, dgamma=dg[i], dbeta=db[i], dxsum_part=sl[i])
        res.append([dp] + dg + db + sl)
    # (the paired reduce pass cuts the rows into chunks for two problems at once: another summation order on large inputs, so the
    #  gradients agree to fp32 rounding, not bit for bit)
    for a_, b_ in zip(*res):
        close(a_, b_.double().cpu(), 2e-6, "paired batch-norm backward")


@pytest.mark.parametrize("C,Rr,N,HW", [(40, 10, 3, 196), (144, 6, 8, 3136), (24, 6, 2, 784), (96, 4, 8, 300), (32, 8, 2, 12544), (1152, 48, 2, 196)])
def test_se_and_bn_backward_share_one_pass(C, Rr, N, HW):
    """mliis_se_bn_bwd_sums + mliis_se_mlp_bwd_bn + mliis_bn_bwd(stage1 = per-image sums, chan_scale, chan_add): the backward of
    z1 -> BN -> swish -> squeeze-excite -> (a1 * gate) (efficientnet_model.py:238-251,271) with ONE reduce pass over (da2, z1), against
    autograd in float64: the batch norm's input gradient and parameter gradients, the SE weight gradients, and against the
    separate column-sum + reduce-pass path on the device."""
    from mliis_amd import ops
    d = dev()
    z = (rnd(N, HW, 1, C, seed=70) * 1.3 + 0.4).requires_grad_(True)
    gamma, beta = (rnd(C, seed=71) * 0.3 + 1).requires_grad_(True), rnd(C, seed=72).requires_grad_(True)
    w1, b1 = rnd(1, 1, C, Rr, seed=73, scale=0.3).requires_grad_(True), rnd(Rr, seed=74).requires_grad_(True)
    w2, b2 = rnd(1, 1, Rr, C, seed=75, scale=0.3).requires_grad_(True), rnd(C, seed=76).requires_grad_(True)
    mean = z.mean(dim=(0, 1, 2))
    var = z.var(dim=(0, 1, 2), unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-3)
    a1 = R.swish((z - mean) * rstd * gamma + beta)
    sp = a1.mean(dim=(1, 2))
    hpre = sp @ w1[0, 0] + b1
    gate = torch.sigmoid(R.swish(hpre) @ w2[0, 0] + b2)
    y = a1 * gate[:, None, None, :]
    dy = rnd(*y.shape, seed=77)
    gz, gga, gbe, gw1, gb1, gw2, gb2 = torch.autograd.grad(y, [z, gamma, beta, w1, b1, w2, b2], dy)
    zg, dyg = f32(z, d), f32(dy, d)
    mg, rg, gg, bg = f32(mean, d), f32(rstd, d), f32(gamma, d), f32(beta, d)
    gateg, hpg, spg = f32(gate, d), f32(hpre, d), f32(sp, d)
    w1g, w2g = f32(w1, d), f32(w2, d)
    part = torch.full((ops.se_bn_bwd_sums_floats(N, HW, C) + 8,), 7.0, device=d)
    nb = ops.se_bn_bwd_sums(zg, dyg, mg, rg, gg, bg, part)
    assert nb > 0 and N * nb * 5 * C == part.numel() - 8 and (part[-8:] == 7.0).all()
    outs = dict(dpre1=torch.empty(N, Rr, device=d), dpre2=torch.empty(N, C, device=d), chan_add=torch.empty(N, C, device=d))
    stage1 = torch.empty(N, 2, C, device=d)
    ops.se_mlp_bwd_bn(part, nb, gateg, hpg, w1g, w2g, HW, outs, stage1)
    outs_t = {k: torch.empty_like(v) for k, v in outs.items()}
    stage1_t = torch.empty_like(stage1)
    ops.se_mlp_bwd_bn(part, nb, gateg, hpg, w1g, w2g, HW, outs_t, stage1_t, w1t=w1g[0, 0].t().contiguous())
    assert torch.equal(stage1, stage1_t) and all(torch.equal(outs[k], outs_t[k]) for k in outs), "se_mlp_bwd_bn with the transposed w1"
    dz = torch.empty_like(zg)
    dga, dbe = torch.empty(C, device=d), torch.empty(C, device=d)
    ops.bn_bwd(zg, dyg, mg, rg, gg, bg, post_swish=True, chan_scale=gateg, chan_add=outs["chan_add"], dx=dz, dgamma=dga, dbeta=dbe,
               stage1=(stage1, N))
    close(dz, gz, 1e-4, "dz1 (one-pass sums)")
    close(dga, gga, 1e-4, "dgamma1")
    close(dbe, gbe, 1e-4, "dbeta1")
    # the SE weight gradients from the kernel's dpre1 / dpre2
    close(spg.t() @ outs["dpre1"], gw1[0, 0], 1e-4, "se dw1")
    close(outs["dpre1"].sum(0), gb1, 1e-4, "se db1")
    close(f32(R.swish(hpre), d).t() @ outs["dpre2"], gw2[0, 0], 1e-4, "se dw2")
    close(outs["dpre2"].sum(0), gb2, 1e-4, "se db2")
    # the separate path on the device: column sum of da2 * a1, SE backward, the batch norm's own reduce pass
    a1g = f32(a1, d)
    o2 = ops.se_mlp_bwd(ops.colsum(dyg, a1g, nseg=N), gateg, spg, hpg, w1g, w2g, HW)
    close(outs["chan_add"], o2["chan_add"].double().cpu(), 2e-5, "chan_add vs the column-sum path")
    dz2, dga2, dbe2 = ops.bn_bwd(zg, dyg, mg, rg, gg, bg, post_swish=True, chan_scale=gateg, chan_add=o2["chan_add"])
    close(dz, dz2.double().cpu(), 2e-5, "dz1 vs the reduce-pass path")
    close(dga, dga2.double().cpu(), 2e-5, "dgamma1 vs the reduce-pass path")


@pytest.mark.parametrize("C,Rr,N,HW", [(40, 10, 3, 196), (672, 28, 2, 49), (24, 6, 2, 784), (144, 6, 8, 300)])
def test_bn_apply_pools_for_squeeze_excite(C, Rr, N, HW):
    """bn_apply_fused(pool_part=...) leaves per-image partial sums of its output; se_mlp_fwd(chunks, scale) folds them: the pooled
    vector and the gate equal the separate colsum + MLP path and the float64 oracle."""
    from mliis_amd import ops
    d = dev()
    x = rnd(N, HW, 1, C, seed=60) * 1.5 + 0.2
    gamma, beta = rnd(C, seed=61) * 0.3 + 1, rnd(C, seed=62)
    w1, b1 = rnd(1, 1, C, Rr, seed=63, scale=0.3), rnd(Rr, seed=64)
    w2, b2 = rnd(1, 1, Rr, C, seed=65, scale=0.3), rnd(C, seed=66)
    mean, var = x.mean(dim=(0, 1, 2)), x.var(dim=(0, 1, 2), unbiased=False)
    a = R.swish((x - mean) * torch.rsqrt(var + 1e-3) * gamma + beta)
    s = a.mean(dim=(1, 2))
    gate = torch.sigmoid(R.swish(s @ w1[0, 0] + b1) @ w2[0, 0] + b2)
    xg = f32(x, d)
    part = torch.empty(ops.bn_stats_partial_floats(N * HW, C) + 16, device=d)
    nblk = ops.bn_stats_partial(xg, False, part)
    m_o, r_o = torch.empty(C, device=d), torch.empty(C, device=d)
    pool = torch.full((N * (-(-HW // 128)) * C + 16,), 3.0, device=d)
    ag, chunks = ops.bn_apply_fused(xg, part, nblk, m_o, r_o, f32(gamma, d), f32(beta, d), post_swish=True, pool_part=pool)
    assert chunks == -(-HW // 128)
    close(ag, a, 3e-5, "bn apply (pooling variant)")
    sg = torch.empty(N, C, device=d)
    hp, gg = ops.se_mlp_fwd(pool, f32(w1, d), f32(b1, d), f32(w2, d), f32(b2, d), chunks=chunks, scale=1.0 / HW, s_out=sg)
    close(sg, s, 1e-5, "pooled vector from the apply pass")
    close(gg, gate, 3e-5, "gate from pooled partials")
    assert pool[N * chunks * C:].eq(3.0).all(), "pool partials written out of range"
    s2 = ops.colsum(ag, None, nseg=N, scale=1.0 / HW)
    hp2, g2 = ops.se_mlp_fwd(s2, f32(w1, d), f32(b1, d), f32(w2, d), f32(b2, d))
    close(g2, gg, 1e-5, "same gate as the separate pooling pass")


def test_chan_affine_broadcast_and_copy():
    from mliis_amd import ops
    d = dev()
    v = rnd(2, 12, seed=31)
    buf = torch.zeros(2, 3, 3, 20, device=d)
    ops.chan_affine(None, A=f32(v, d), out=buf[..., 4:16])
    close(buf[..., 4:16], v[:, None, None, :].expand(2, 3, 3, 12), 1e-6, "tile")
    src = rnd(2, 3, 3, 8, seed=32)
    ops.chan_affine(f32(src, d), out=buf[..., 12:20])
    close(buf[..., 12:20], src, 1e-6, "copy")
    ops.chan_affine(None, A=f32(v, d), out=buf[..., 4:16], accumulate=True)
    assert buf[..., :4].abs().max().item() == 0


# ------------------------------------------------------------------------------------------------ resize
@pytest.mark.parametrize("Hi,Wi,Ho,Wo,C", [(14, 14, 56, 56, 112), (7, 5, 20, 13, 8), (56, 56, 224, 224, 2), (4, 4, 4, 4, 8), (3, 6, 2, 2, 4)])
def test_resize(Hi, Wi, Ho, Wo, C):
    from mliis_amd import ops
    d = dev()
    x = rnd(2, Hi, Wi, C, seed=33).requires_grad_(True)
    y = F.interpolate(nchw(x), size=(Ho, Wo), mode="bilinear", align_corners=True)
    dy = rnd(*y.shape, seed=34)
    (gx,) = torch.autograd.grad(y, [x], dy)
    close(ops.resize_bilinear_fwd(f32(x, d), (Ho, Wo)), nhwc(y), 2e-5, "resize fwd")
    close(ops.resize_bilinear_bwd(f32(nhwc(dy), d), (Hi, Wi)), gx, 1e-4, "resize bwd")


@pytest.mark.parametrize("Hi,Wi,Ho,Wo,C,pad", [(14, 14, 56, 56, 112, 24), (7, 9, 28, 28, 40, 0), (14, 14, 14, 14, 48, 8), (28, 28, 14, 14, 16, 0),
                                               (5, 70, 9, 130, 36, 4)])
def test_resize_bwd_separable_form_strided_and_accumulating(Hi, Wi, Ho, Wo, C, pad):
    """The wide-channel resize transpose (resize_bwd_rows_k: per input row, candidate output rows folded into LDS, then the columns)
    against float64 autograd: channel counts that are not a multiple of its 32-channel groups, gradient and target as channel slices of
    wider tensors (leading dimension > C, as the decoder's concat gradient is), accumulation into an existing gradient, up- and
    down-sampling, more output columns than the 64 pixel lanes of a workgroup."""
    from mliis_amd import ops
    d = dev()
    x = rnd(2, Hi, Wi, C, seed=35).requires_grad_(True)
    y = F.interpolate(nchw(x), size=(Ho, Wo), mode="bilinear", align_corners=True)
    dy = rnd(*y.shape, seed=36)
    (gx,) = torch.autograd.grad(y, [x], dy)
    wide = torch.full((2, Ho, Wo, C + pad), float("nan"), device=d)
    wide[..., :C] = f32(nhwc(dy), d)
    prev = rnd(2, Hi, Wi, C, seed=37)
    tgt = torch.full((2, Hi, Wi, C + pad), 7.0, device=d)
    tgt[..., pad:] = f32(prev, d)
    ops.resize_bilinear_bwd(wide[..., :C], (Hi, Wi), out=tgt[..., pad:], accumulate=True)
    close(tgt[..., pad:], gx + prev, 1e-4, "resize bwd, accumulate, strided")
    if pad:
        assert (tgt[..., :pad] == 7.0).all(), "channels outside the slice are untouched"
    ops.resize_bilinear_bwd(wide[..., :C], (Hi, Wi), out=tgt[..., pad:])
    close(tgt[..., pad:], gx, 1e-4, "resize bwd, overwrite, strided")


# ------------------------------------------------------------------------------------------------ final conv + loss
@pytest.mark.parametrize("C,rows,use_mask", [(112, 300, False), (136, 77, True), (8, 5, True)])
def test_final_conv(C, rows, use_mask):
    from mliis_amd import ops
    d = dev()
    x = rnd(1, rows, 1, C, seed=35).requires_grad_(True)
    w = rnd(1, 1, C, 2, seed=36).requires_grad_(True)
    b = rnd(2, seed=37).requires_grad_(True)
    mask = ((torch.rand(1, rows, 1, C, generator=torch.Generator().manual_seed(38)) > 0.5).double() * 2.0) if use_mask else None
    xm = x * mask if use_mask else x
    y = xm @ w[0, 0] + b
    dy = rnd(*y.shape, seed=39)
    gx, gw, gb = torch.autograd.grad(y, [x, w, b], dy)
    mg = f32(mask, d) if use_mask else None
    close(ops.final_conv_fwd(f32(x, d), f32(w, d), f32(b, d), mg), y, 2e-5, "final fwd")
    close(ops.final_conv_bwd_data(f32(dy, d), f32(w, d), C, mg), gx, 1e-4, "final dx")
    dw, db = ops.final_conv_bwd_filter(f32(x, d), f32(dy, d), mg)
    close(dw, gw, 1e-4, "final dw")
    close(db, gb, 1e-4, "final db")


@pytest.mark.parametrize("dice,ls,H,W", [(False, 0.0, 16, 16), (True, 0.0, 24, 20), (True, 0.1, 9, 7), (False, 0.2, 224, 224)])
def test_softmax_ce(dice, ls, H, W):
    from mliis_amd import ops
    d = dev()
    S, idx = 3, [1, 1, 0, 2]
    N = len(idx)
    z = (rnd(N, H, W, 2, seed=40) * 3).requires_grad_(True)
    g = torch.Generator().manual_seed(41)
    m = torch.rand(S, H, W, generator=g).double()
    m = torch.where(m < 0.6, (m < 0.3).double(), m)  # mostly binary, some fractional labels
    labels = torch.stack([1 - m, m], -1)
    a = R.arch()
    loss = R.loss_fn(a, {}, z, labels[idx], ls, dice, False)
    (gz,) = torch.autograd.grad(loss, [z])
    out, dl, pred = ops.softmax_ce(f32(z, d), f32(labels, d), torch.tensor(idx, dtype=torch.int32, device=d), ls, dice, want_pred=True)
    assert abs(out[0].item() - loss.item()) <= 2e-5 * max(1.0, abs(loss.item()))
    close(dl, gz, 1e-4, "dlogits")
    # mask: bit-exact against the oracle rule applied to the same fp32 logits
    zf = z.detach().float()
    ref_pred = (torch.softmax(zf.double(), -1) > 0.5).float()
    margin = (zf[..., 0] - zf[..., 1]).abs() > 1e-6
    assert torch.equal(pred.cpu()[margin], ref_pred[margin])


# ------------------------------------------------------------------------------------------------ optimizer / arena
def test_sgd_l2_mask_and_arena_algebra():
    from mliis_amd import ops
    d = dev()
    n = 4096 + 8
    w, g = rnd(n, seed=42), rnd(n, seed=43)
    mask = (torch.arange(n // 4) % 3 != 0)
    wg, gg = f32(w, d), f32(g, d)
    lr_dev = torch.tensor([0.05], device=d)
    ops.sgd_fused(wg, gg, 123.0, mask.to(torch.uint8).to(d), 5e-4, lr_dev)  # lr_dev overrides the host lr
    l2m = mask.repeat_interleave(4).double()
    close(wg, w - 0.05 * (g + 5e-4 * w * l2m), 1e-6, "sgd")
    x, y = rnd(n, seed=44), rnd(n, seed=45)
    yg = f32(y, d)
    ops.axpby(0.3, f32(x, d), -1.5, yg)
    close(yg, 0.3 * x - 1.5 * y, 1e-6, "axpby")
    o = torch.empty(n, device=d)
    ops.lincomb(2.0, f32(x, d), 0.25, f32(y, d), o)
    close(o, 2 * x + 0.25 * y, 1e-6, "lincomb")
    # Adam(beta1=0) single step, t = 1
    v = torch.zeros(n, device=d)
    w2 = f32(w, d)
    ops.adam_b1zero_fused(w2, gg, v, torch.tensor([1.0], device=d), 1e-3)
    vv = 0.001 * g * g
    close(w2, w - 1e-3 * math.sqrt(1 - 0.999) * g / (vv.sqrt() + 1e-8), 1e-5, "adam b1=0")
    # the self-advancing form a captured graph replays: steps applied so far on the device, the launch uses t + 1 and stores it
    v5, w5 = torch.zeros(n, device=d), f32(w, d)
    t_dev, ticket = torch.zeros(1, device=d), torch.zeros(1, dtype=torch.int32, device=d)
    ops.adam_b1zero_fused(w5, gg, v5, t_dev, 1e-3, ticket=ticket)
    assert torch.equal(w5, w2) and torch.equal(v5, v) and t_dev.item() == 1.0 and ticket.item() == 0
    ops.adam_b1zero_fused(w5, gg, v5, t_dev, 1e-3, ticket=ticket)
    ops.adam_b1zero_fused(w2, gg, v, torch.tensor([2.0], device=d), 1e-3)
    assert torch.equal(w5, w2) and torch.equal(v5, v) and t_dev.item() == 2.0 and ticket.item() == 0
    # --l1 (models/regularizers.py:13-19): + l1 * sign(w) on the same (non batch-norm) quads, together with L2; sign(0) = 0
    w[8:12] = 0.0
    w3 = f32(w, d)
    ops.sgd_fused(w3, gg, 0.05, mask.to(torch.uint8).to(d), 5e-4, None, l1=2e-3)
    close(w3, w - 0.05 * (g + (5e-4 * w + 2e-3 * torch.sign(w)) * l2m), 1e-6, "sgd + l1")
    v = torch.zeros(n, device=d)
    w4 = f32(w, d)
    ops.adam_b1zero_fused(w4, gg, v, torch.tensor([1.0], device=d), 1e-3, mask.to(torch.uint8).to(d), 0.0, None, l1=2e-3)
    g1 = g + 2e-3 * torch.sign(w) * l2m
    close(w4, w - 1e-3 * math.sqrt(1 - 0.999) * g1 / ((0.001 * g1 * g1).sqrt() + 1e-8), 1e-5, "adam + l1")


def test_errors_are_loud():
    from mliis_amd import ops
    from mliis_amd._lib import MliisError
    d = dev()
    with pytest.raises(MliisError):
        ops.dwconv_fwd(torch.zeros(1, 4, 4, 6, device=d), torch.zeros(3, 3, 6, 1, device=d), 1)  # C % 4 != 0
    with pytest.raises(MliisError):
        ops.dwconv_fwd(torch.zeros(1, 4, 4, 8, device=d), torch.zeros(7, 7, 8, 1, device=d), 1)  # k = 7 unsupported
    with pytest.raises(MliisError):
        ops.dwconv_fwd(torch.zeros(1, 4, 4, 8), torch.zeros(3, 3, 8, 1), 1)  # CPU tensors: no fallback


# ------------------------------------------------------------------------------------------------ fused BN fast path
@pytest.mark.parametrize("k,dil,H,W,Cin,Cout,N,swish", [(1, 1, 14, 14, 40, 240, 2, False), (3, 2, 20, 20, 136, 112, 2, True), (1, 1, 56, 56, 24, 144, 4, False),
                                                        (3, 1, 9, 11, 72, 112, 1, True), (1, 1, 7, 7, 96, 24, 3, False)])
def test_conv_epilogue_statistics_and_fused_bn(k, dil, H, W, Cin, Cout, N, swish):
    """conv2d_fwd's fused stage-1 statistics + bn_apply_fused == separate bn_stats + bn_apply == float64 oracle."""
    from mliis_amd import ops
    d = dev()
    x = rnd(N, H, W, Cin, seed=50)
    w = rnd(k, k, Cin, Cout, seed=51, scale=1.0 / math.sqrt(k * k * Cin))
    b = rnd(Cout, seed=52)
    gamma, beta = rnd(Cout, seed=53) * 0.3 + 1, rnd(Cout, seed=54)
    res = rnd(N, H, W, Cout, seed=55)
    z = nhwc(R.conv2d_same(nchw(x), w, 1, dil, bias=b))
    u = R.swish(z) if swish else z
    mean, var = u.mean(dim=(0, 1, 2)), u.var(dim=(0, 1, 2), unbiased=False)
    y = (u - mean) * torch.rsqrt(var + 1e-3) * gamma + beta
    y = (y if swish else R.swish(y)) + res
    part = torch.empty(1 << 18, device=d)
    zg, nblk = ops.conv2d_fwd(f32(x, d), f32(w, d), f32(b, d), dil, stats_part=part, stats_swish=swish)
    close(zg, z, 2e-5, "conv out")
    if nblk == 0:   # split-K plan: the caller falls back to the stats kernel
        nblk = ops.bn_stats_partial(zg, swish, part)
    assert nblk > 0
    n = N * H * W
    sums = part[: nblk * 2 * Cout].view(nblk, 2, Cout).double().sum(0).cpu()
    close(sums[0] / n, mean, 1e-5, "epilogue mean")
    close(sums[1] / n, var + mean ** 2, 1e-5, "epilogue E[x^2]")
    mm, mv = torch.zeros(Cout, device=d), torch.ones(Cout, device=d)
    m_o, r_o = torch.empty(Cout, device=d), torch.empty(Cout, device=d)
    yg = ops.bn_apply_fused(zg, part, nblk, m_o, r_o, f32(gamma, d), f32(beta, d), moving=(mm, mv), unbiased_moving_var=swish, pre_swish=swish,
                            post_swish=not swish, res=f32(res, d))
    close(yg, y, 3e-5, "fused bn apply")
    close(m_o, mean, 1e-5, "mean out")
    close(r_o, torch.rsqrt(var + 1e-3), 1e-5, "rstd out")
    close(mm, 0.01 * mean, 1e-5, "moving mean")
    close(mv, 0.99 + 0.01 * var * (n / (n - 1.0) if swish else 1.0), 1e-5, "moving var")
    # stats kernel path gives the same normalisation
    nb2 = ops.bn_stats_partial(zg, swish, part)
    y2 = ops.bn_apply_fused(zg, part, nb2, m_o, r_o, f32(gamma, d), f32(beta, d), pre_swish=swish, post_swish=not swish, res=f32(res, d))
    close(y2, y, 3e-5, "fused bn apply (stats kernel)")


@pytest.mark.parametrize("N,hd,H,ls,S", [(8, 56, 224, 0.0, 5), (3, 16, 64, 0.1, 3), (2, 24, 96, 0.0, 2), (5, 7, 28, 0.2, 5), (1, 56, 224, 0.0, 1),
                                         (2, 96, 384, 0.0, 2), (2, 13, 37, 0.1, 2), (3, 21, 21, 0.0, 3)])
def test_head_ce_fused_equals_the_four_launch_tail(N, hd, H, ls, S):
    """mliis_head_ce_fused (resize -> softmax cross-entropy -> gradient -> resize^T on the decoder's map) == the chain
    mliis_resize_bilinear_fwd -> mliis_softmax_ce -> mliis_resize_bilinear_bwd to fp32 rounding; both against the float64 oracle
    (efficientlab.py:166-173,294-303); a second launch gives the same bits."""
    from mliis_amd import ops
    d = dev()
    small = rnd(N, hd, hd, 2, seed=90) * 3
    lab = (torch.rand(S, H, H, 1, generator=torch.Generator().manual_seed(91)) > 0.6).double()
    labels = torch.cat([1 - lab, lab], dim=-1)
    idx = torch.tensor([i % S for i in range(N)], dtype=torch.int32)
    sg, lg, ig = f32(small, d), f32(labels, d), idx.to(d)
    # oracle
    sm = small.clone().requires_grad_(True)
    logits = nhwc(F.interpolate(nchw(sm), size=(H, H), mode="bilinear", align_corners=True))
    t = labels[idx.long()] * (1 - ls) + 0.5 * ls
    ce = -(t * torch.log_softmax(logits, dim=-1)).sum(-1).mean()
    (gs,) = torch.autograd.grad(ce, [sm])
    # four launches
    lo = ops.resize_bilinear_fwd(sg, (H, H))
    out1 = torch.zeros(4, device=d)
    _, dl, _ = ops.softmax_ce(lo, lg, ig, ls, False, 0.0, want_grad=True, out=out1)
    ds1 = ops.resize_bilinear_bwd(dl, (hd, hd))
    # fused
    out2, ds2 = torch.zeros(4, device=d), torch.full_like(sg, float("nan"))
    ops.head_ce_fused(sg, lg, ig, (H, H), ls, ds2, out2)
    close(ds2, ds1, 3e-6, "gradient vs the five-launch tail")   # (the same arithmetic per element; fp contraction may differ between kernels)
    close(ds2, gs, 2e-5, "gradient on the decoder's map")
    assert abs(out2[0].item() - ce.item()) <= 2e-6 * max(1.0, abs(ce.item())), (out2, ce)
    assert torch.allclose(out2[:3], out1[:3], rtol=2e-6, atol=1e-7), (out1, out2)
    out3, ds3 = torch.zeros(4, device=d), torch.empty_like(sg)
    ops.head_ce_fused(sg, lg, ig, (H, H), ls, ds3, out3)
    assert torch.equal(ds3, ds2) and torch.equal(out3, out2)


# (N, H, W, Cin, Cout, residual, drop-connect, precision): the five expand convs of EfficientLab-6-3 at 224x224 (16 -> 96 at 112x112,
#  24 -> 144 at 56x56, 40 -> 240 at 28x28, 80 -> 480 and 112 -> 672 at 14x14), a map whose row count is no multiple of 16 with images
#  that end inside a row group, a channel count that pads a K group, and the reduced-precision instances
@pytest.mark.parametrize("N,H,W,Cin,Cout,res,dc,prec", [(8, 112, 112, 16, 96, False, False, "fp32"), (8, 56, 56, 24, 144, True, True, "fp32"),
                                                        (8, 28, 28, 40, 240, True, True, "fp32"), (8, 14, 14, 80, 480, True, False, "fp32"),
                                                        (8, 14, 14, 112, 672, True, True, "fp32"), (3, 19, 21, 24, 40, True, True, "fp32"),
                                                        (5, 15, 15, 36, 100, False, True, "fp32"), (8, 14, 14, 112, 672, True, True, "bf16"),
                                                        (8, 28, 28, 40, 240, True, True, "fp8")])
def test_conv1x1_with_the_batch_norm_in_front_applied_on_load(N, H, W, Cin, Cout, res, dc, prec):
    """mliis_conv2d_fwd_bnin: the project batch norm of the MBConv block in front (+ drop-connect scale per image, + identity skip:
    efficientnet_model.py:283-288, utils.py:157-170) applied while the next block's expand conv loads its rows == bn_apply_fused +
    conv2d_fwd (two launches) == float64 oracle: the conv output and its fused statistics, the finished block tensor a_out, mean /
    rstd and the moving averages."""
    from mliis_amd import ops
    d = dev()
    if not ops.conv2d_fwd_bnin_ok(N, H, W, Cin, Cout):
        pytest.skip("not a streamed 1x1 shape")
    z = rnd(N, H, W, Cin, seed=70) * (rnd(Cin, seed=71).abs() + 0.5) + rnd(Cin, seed=72)
    w = rnd(1, 1, Cin, Cout, seed=73, scale=1.0 / math.sqrt(Cin))
    gamma, beta = rnd(Cin, seed=74) * 0.3 + 1, rnd(Cin, seed=75)
    r = rnd(N, H, W, Cin, seed=76) if res else None
    sc = (torch.rand(N, generator=torch.Generator().manual_seed(77), dtype=torch.float64) > 0.3).double() / 0.7 if dc else None
    mean, var = z.mean(dim=(0, 1, 2)), z.var(dim=(0, 1, 2), unbiased=False)
    a = (z - mean) * torch.rsqrt(var + 1e-3) * gamma + beta
    if dc:
        a = a * sc[:, None, None, None]
    if res:
        a = a + r
    y = nhwc(R.conv2d_same(nchw(a), w, 1, 1))
    zg, wg = f32(z, d), f32(w, d)
    nst = (-(-N * H * W // 16)) * 2 * Cout + 64
    # (NaN behind the last partial block: the fused fold reads its batches of eight slots through a buffer descriptor that ends there -- a
    #  slot past the end must come back as zero, never as what lies behind: ADVICE r05)
    part, part2, part3 = torch.full((1 << 18,), float("nan"), device=d), torch.empty(nst, device=d), torch.empty(nst, device=d)
    nblk = ops.bn_stats_partial(zg, False, part)
    m1, r1, m2, r2 = (torch.empty(Cin, device=d) for _ in range(4))
    mm1, mv1, mm2, mv2 = torch.zeros(Cin, device=d), torch.ones(Cin, device=d), torch.zeros(Cin, device=d), torch.ones(Cin, device=d)
    rg_, scg = (f32(r, d) if res else None), (f32(sc, d) if dc else None)
    # two launches
    a1 = ops.bn_apply_fused(zg, part, nblk, m1, r1, f32(gamma, d), f32(beta, d), moving=(mm1, mv1), img_scale=scg, res=rg_)
    y1, nb1 = ops.conv2d_fwd(a1, wg, None, 1, stats_part=part2, precision=prec)
    # one launch
    a2, y2 = torch.full_like(a1, float("nan")), torch.empty_like(y1)
    _, nb2 = ops.conv2d_fwd_bnin(zg, part, nblk, m2, r2, f32(gamma, d), f32(beta, d), a2, wg, y2, moving=(mm2, mv2), img_scale=scg, res=rg_,
                                 stats_part=part3, precision=prec)
    tol = {"fp32": 2e-5, "bf16": 1e-2, "fp8": 1e-1}[prec]
    close(a2, a, 3e-5, "block tensor")
    close(a2, a1, 2e-6, "block tensor vs the stand-alone apply")
    close(y2, y, tol, "conv of the normalised tensor")
    close(y2, y1, 2e-6 if prec == "fp32" else tol, "conv vs the two-launch form")
    close(m2, mean, 1e-5, "mean")
    close(r2, torch.rsqrt(var + 1e-3), 1e-5, "rstd")
    assert torch.equal(m2, m1) and torch.equal(r2, r1) and torch.equal(mm2, mm1) and torch.equal(mv2, mv1)   # (the same fold, bit for bit)
    close(mm2, 0.01 * mean, 1e-5, "moving mean")
    close(mv2, 0.99 + 0.01 * var, 1e-5, "moving var")
    assert nb1 == nb2 and nb2 > 0
    s1 = part2[: nb1 * 2 * Cout].view(nb1, 2, Cout).double().sum(0)
    s2 = part3[: nb2 * 2 * Cout].view(nb2, 2, Cout).double().sum(0)
    close(s2, s1, 1e-5 if prec == "fp32" else tol, "statistics of the conv output")
    # group-blocked output (what the small-map fused kernels read) through the same loader
    if prec == "fp32" and Cout % 4 == 0 and H * W <= 256:
        y3 = torch.empty(N, H, W, Cout, device=d)
        ops.conv2d_fwd_bnin(zg, part, nblk, m2, r2, f32(gamma, d), f32(beta, d), a2, wg, y3, img_scale=scg, res=rg_, stats_part=part3, out_block=4)
        assert torch.equal(y3.view(Cout // 4, N * H * W, 4).permute(1, 0, 2).reshape(N, H, W, Cout), y2)


@pytest.mark.parametrize("H,W,Cin,Cout,N", [(14, 14, 240, 40, 3), (7, 9, 96, 24, 2), (28, 28, 144, 40, 2)])
def test_conv1x1_with_se_gate_on_the_fly(H, W, Cin, Cout, N):
    """x_scale: the squeeze-excite gate applied inside the GEMM loaders == conv / filter-gradient of the gated tensor."""
    from mliis_amd import ops
    d = dev()
    x = rnd(N, H, W, Cin, seed=60)
    gate = torch.sigmoid(rnd(N, Cin, seed=61))
    w = rnd(1, 1, Cin, Cout, seed=62, scale=1.0 / math.sqrt(Cin)).requires_grad_(True)
    xs = x * gate[:, None, None, :]
    y = R.conv2d_same(nchw(xs), w, 1, 1)
    dy = rnd(*y.shape, seed=63)
    (gw,) = torch.autograd.grad(y, [w], dy)
    xg, gg, wg = f32(x, d), f32(gate, d), f32(w, d)
    wt = wg.permute(0, 1, 3, 2).contiguous().view(-1)
    close(ops.conv2d_fwd(xg, wg, None, 1, x_scale=gg, wt=wt), nhwc(y), 2e-5, "gated conv")
    close(ops.conv2d_bwd_filter(xg, f32(nhwc(dy), d), 1, 1, x_scale=gg), gw, 1e-4, "gated filter grad")


@pytest.mark.parametrize("H,W,Cc,Cp,Co,N", [(14, 14, 224, 224, 112, 2), (9, 12, 48, 24, 16, 3), (56, 56, 224, 136, 112, 2), (2, 2, 16, 8, 16, 2), (6, 6, 32, 24, 16, 37)])   # (37 images: three groups of the per-image accumulators)
def test_rsd_pooled_branch_as_border_bias(H, W, Cc, Cp, Co, N):
    """3x3 conv over [convolved Cc channels | Cp spatially-constant channels] == conv over the Cc channels + per-border-class
    bias; gradients of the constant channels / their weight rows from per-image border sums (rsd.hip)."""
    from mliis_amd import ops
    d = dev()
    xc = rnd(N, H, W, Cc, seed=70).requires_grad_(True)
    pool = rnd(N, Cp, seed=71).requires_grad_(True)
    w = rnd(3, 3, Cc + Cp, Co, seed=72, scale=1.0 / math.sqrt(9 * (Cc + Cp))).requires_grad_(True)
    b = rnd(Co, seed=73).requires_grad_(True)
    full = torch.cat([xc, pool[:, None, None, :].expand(N, H, W, Cp)], dim=-1)
    z = nhwc(R.conv2d_same(nchw(full), w, 1, 1, bias=b))
    dz = rnd(*z.shape, seed=74)
    gx, gp, gw, gb = torch.autograd.grad(z, [xc, pool, w, b], dz)
    xg, pg, wg, bg, dzg = f32(xc, d), f32(pool, d), f32(w, d), f32(b, d), f32(dz, d)
    wt = wg.permute(0, 1, 3, 2).contiguous().view(-1)
    E = ops.rsd_pool_fwd(pg, wg, Cc)
    # the same vectors as chunk partials with a scale (what rsd_concat_pool hands over), folded by the launch
    parts = torch.stack([f32(pool * a, d) for a in (0.5, 1.25, 0.25)], dim=1).contiguous()   # [N, 3, Cp], sum = 2 * pool
    pool_out = torch.empty(N, Cp, device=d)
    E2 = ops.rsd_pool_fwd(parts, wg, Cc, chunks=3, scale=0.5, pool_out=pool_out)
    close(pool_out, pool.detach(), 1e-6, "folded pool")
    close(E2, E.double().cpu(), 1e-5, "border bias from chunk partials")
    for kw in (dict(), dict(wt=wt)):
        close(ops.conv2d_fwd(xg, wg, bg, 1, border_bias=E, **kw), z, 2e-5, "fwd with border bias")
    dw = torch.full((3, 3, Cc + Cp, Co), float("nan"), device=d)
    db = torch.empty(Co, device=d)
    tot = torch.full((N, Co), float("nan"), device=d)   # an output: the per-image column sums of dz come from the border-sum launch
    dpool = ops.rsd_pool_bwd(dzg, tot, pg, wg, Cc, dw=dw, dbias=db)
    close(tot, dz.sum(dim=(1, 2)), 1e-4, "per-image sums of dz")
    ops.conv2d_bwd_filter(xg, dzg, 3, 1, out=dw)
    assert not torch.isnan(dw).any()
    close(dw, gw, 1e-4, "dw (convolved + constant rows)")
    close(db, gb, 1e-4, "dbias")
    close(dpool * (H * W), gp, 1e-4, "dpool")
    close(ops.conv2d_bwd_data(dzg, wg, 1, ci_begin=0, ci_count=Cc), gx, 1e-4, "dx of the convolved channels")


@pytest.mark.parametrize("N,Hi,H,Cd,Cs", [(8, 14, 14, 112, 112), (8, 14, 56, 112, 24), (3, 5, 9, 8, 4), (2, 7, 28, 40, 16), (2, 96, 96, 112, 24)])
def test_rsd_concat_and_pooled_sums(N, Hi, H, Cd, Cs):
    """mliis_rsd_concat_pool: cat = [deep copied / bilinearly resized (bit-identical to mliis_resize_bilinear_fwd) | skip] and the
    per-image column sums of cat as chunk partials (models/efficientlab.py:192-197,205-208)."""
    from mliis_amd import ops
    d = dev()
    deep, skip = f32(rnd(N, Hi, Hi, Cd, seed=75), d), f32(rnd(N, H, H, Cs, seed=76), d)
    cat = torch.full((N, H, H, Cd + Cs), float("nan"), device=d)
    part = torch.full((ops.rsd_concat_pool_floats(N, H, H, Cd + Cs) + 4,), 7.0, device=d)
    chunks = ops.rsd_concat_pool(deep, skip, cat, part)
    assert chunks >= 1 and N * chunks * (Cd + Cs) == part.numel() - 4 and (part[-4:] == 7.0).all()
    ref_up = deep if Hi == H else ops.resize_bilinear_fwd(deep, (H, H))
    assert torch.equal(cat[..., :Cd], ref_up) and torch.equal(cat[..., Cd:], skip)
    sums = part[:-4].view(N, chunks, Cd + Cs).double().sum(1).cpu()
    close(sums, cat.double().sum(dim=(1, 2)).cpu(), 1e-5, "per-image column sums of the concat")


def test_the_1x1_kernel_instances_fit_the_residency_their_planner_assumes():
    """stream_plan / ksplit_plan launch two 512-thread workgroups per CU in ONE round.  An instance whose register count crosses 128
    silently fits one, and the grid then runs in two rounds (round 3: the 16 -> 96 expand conv at 23 us instead of 12.6).  The runtime's
    own occupancy figure for every instance the planners can pick (mliis_conv1x1_occupancy)."""
    import ctypes as C
    from mliis_amd._lib import lib
    dev()
    stream = [(1, 1), (1, 2), (1, 3), (1, 4), (2, 1), (2, 2), (2, 3), (3, 1), (3, 2), (4, 1), (4, 2), (5, 1), (6, 1), (7, 1)]
    ksplit = [(1, n) for n in range(1, 8)] + [(2, 1), (2, 2), (2, 3), (2, 4), (3, 1), (3, 2), (4, 1), (4, 2), (5, 1), (6, 1), (7, 1)]
    for kind, combos in ((0, stream), (1, ksplit)):
        for kc, nt in combos:
            nb = C.c_int(0)
            lib.call("mliis_conv1x1_occupancy", kind, kc, nt, C.byref(nb))
            need = 1 if (kind == 1 and kc >= 7) else 2     # (ksplit_plan launches one per CU for KC = 7)
            assert nb.value >= need, ("stream" if kind == 0 else "ksplit", kc, nt, nb.value)


# ------------------------------------------------------------------------------------------------ batched slab fold
def test_fold_batched_dense_segmented_and_ragged():
    """mliis_fold_batched: several descriptors in one launch -- a dense vector path, a segmented (channel-window) output, a total that
    is not a multiple of 4 (scalar path) and slab counts from 1 to 37 -- against a float64 sum."""
    from mliis_amd import ops
    from mliis_amd._lib import lib
    d = dev()
    tile = lib.raw("mliis_fold_tile_outputs")()
    g = torch.Generator().manual_seed(7)
    # (total, nblk, seg_len, seg_stride, seg_off)
    cases = [(4096, 37, 4096, 0, 0), (9 * 8 * 12, 5, 8 * 12, 20 * 12, 4 * 12), (27 * 5 + 2, 3, 27 * 5 + 2, 0, 0), (520, 1, 520, 0, 0), (260, 16, 52, 100, 8)]
    parts, rows, outs, poff, ooff, tcount = [], [], [], 0, 0, 0
    for total, nblk, sl, ss, so in cases:
        p = torch.randn(nblk, total, generator=g, dtype=torch.float64)
        parts.append((poff, p))
        span = (total // sl - 1) * ss + so + sl if ss else total
        span = (span + 3) // 4 * 4
        rows.append([poff, ooff, total, sl, ss, so, nblk, tcount])
        outs.append((ooff, span, total, sl, ss, so, p.sum(0)))
        poff += (nblk * total + 3) // 4 * 4
        ooff += span
        tcount += -(-total // tile)
    buf = torch.zeros(poff + 16, device=d)
    for off, p in parts:
        buf[off:off + p.numel()] = p.float().reshape(-1).to(d)
    out = torch.full((ooff + 16,), -7.0, device=d)
    ops.fold_batched(buf, out, torch.tensor(rows, dtype=torch.int64, device=d), tcount)
    out = out.cpu().double()
    for off, span, total, sl, ss, so, ref in outs:
        want = torch.full((span,), -7.0, dtype=torch.float64)
        i = torch.arange(total)
        want[(i // sl) * ss + so + i % sl] = ref
        got = out[off:off + span]
        assert torch.equal(got == -7.0, want == -7.0), "fold wrote outside its window"
        close(got, want, 1e-6, "fold_batched total={}".format(total))
    # the squeeze-excite weight gradients of a pass as extra workgroups of the same launch: the fold's outputs and the gradients are bit
    # for bit what the two launches (mliis_fold_batched alone, mliis_se_wgrad_batched) leave
    se_rows, se_ref, se_got, keep, se_tile = [], [], [], [], 0
    for i, (N, C, Rr) in enumerate([(8, 96, 4), (8, 672, 28), (5, 240, 10)]):
        s_, hp, d1, d2 = f32(rnd(N, C, seed=40 + i), d), f32(rnd(N, Rr, seed=50 + i), d), f32(rnd(N, Rr, seed=60 + i), d), f32(rnd(N, C, seed=70 + i), d)
        a = [torch.full(sh, 3.0, device=d) for sh in ((C, Rr), (Rr,), (Rr, C), (C,))]
        b = [torch.full(sh, 4.0, device=d) for sh in ((C, Rr), (Rr,), (Rr, C), (C,))]
        keep += [s_, hp, d1, d2]
        se_ref.append(a); se_got.append(b)
        for dst, tgt in ((se_rows, b),):
            dst.append([s_.data_ptr(), hp.data_ptr(), d1.data_ptr(), d2.data_ptr()] + [t.data_ptr() for t in tgt] + [N, C, Rr, se_tile])
        se_tile += -(-(2 * C * Rr + C + Rr) // 256)
    ref_rows = [r[:4] + [t.data_ptr() for t in a] + r[8:] for r, a in zip(se_rows, se_ref)]
    ops.se_wgrad_batched(torch.tensor(ref_rows, dtype=torch.int64, device=d), se_tile)
    out2 = torch.full((ooff + 16,), -7.0, device=d)
    ops.fold_batched(buf, out2, torch.tensor(rows, dtype=torch.int64, device=d), tcount, se_desc=torch.tensor(se_rows, dtype=torch.int64, device=d),
                     se_tiles=se_tile)
    assert torch.equal(out2.cpu().double(), out)
    for a, b in zip(se_ref, se_got):
        for x_, y_ in zip(a, b):
            assert torch.equal(x_, y_) and not torch.any(y_ == 4.0)


@pytest.mark.parametrize("pre_mask", [False, True])
def test_swish_mask_fwd_bwd(pre_mask):
    """ASPP activation sites: swish(z) * mask and swish(z * mask), forward and backward, on channel-slice views, vs autograd."""
    from mliis_amd import ops
    d = dev()
    z = rnd(3, 5, 7, 24, seed=70, scale=2.0).requires_grad_(True)
    mask = torch.tensor(2.0 * (np.random.default_rng(1).random((3, 5, 7, 24)) < 0.5))
    y = R.swish(z * mask) if pre_mask else R.swish(z) * mask
    dy = rnd(3, 5, 7, 24, seed=71)
    (gz,) = torch.autograd.grad(y, [z], dy)
    cat = torch.full((3, 5, 7, 56), 9.0, device=d)
    ops.swish_mask_fwd(f32(z, d), f32(mask, d), out=cat[..., 8:32], pre_mask=pre_mask)
    close(cat[..., 8:32], y, 2e-6, "swish-mask fwd")
    assert (cat[..., :8] == 9).all() and (cat[..., 32:] == 9).all()
    dcat = torch.zeros(3, 5, 7, 56, device=d)
    dcat[..., 8:32] = f32(dy, d)
    ops.swish_mask_bwd(dcat[..., 8:32], f32(z, d), f32(mask, d), out=dcat[..., 8:32], pre_mask=pre_mask)     # in place on the slice
    close(dcat[..., 8:32], gz, 1e-5, "swish-mask bwd")
    # inference: no mask
    close(ops.swish_mask_fwd(f32(z, d), None, pre_mask=pre_mask), R.swish(z), 2e-6, "swish fwd, no mask")


@pytest.mark.parametrize("k,dil,H,W,Cin,Cout,N", [(1, 1, 14, 14, 40, 240, 2), (3, 2, 14, 14, 136, 112, 2), (1, 1, 56, 56, 136, 144, 4), (3, 1, 8, 8, 20, 16, 2),
                                                  (1, 1, 14, 14, 480, 80, 8), (1, 1, 14, 14, 672, 112, 5), (1, 1, 28, 28, 240, 40, 2),   # (conv1x1_ksplit_k, bf16 instances)
                                                  (3, 1, 14, 14, 224, 112, 8), (3, 2, 16, 16, 136, 112, 8)])   # (the decoder's small-map 3x3 convs: 64-column tiles, split K -- forward and backward-data)
def test_conv2d_bf16_operands(k, dil, H, W, Cin, Cout, N):
    """precision = MLIIS_PREC_BF16 (per call): bf16 operands on the matrix cores (v_mfma_f32_16x16x32_bf16), fp32 accumulation.  Exactly the fp32
    result of the bf16-ROUNDED operands up to accumulation order (tolerance 2e-5), i.e. within bf16 rounding (2^-9 relative per operand)
    of the full-precision result.  (The memory-bound short-K 1x1 convs -- conv1x1_stream_k, K <= 112 and >= 1024 pixels -- keep fp32
    operands in this mode too; the shapes here are the ones that do switch.)"""
    from mliis_amd import ops
    d = dev()
    bf = lambda t: t.to(torch.bfloat16).to(torch.float64)   # noqa: E731  round to nearest even, like v_cvt_pk_bf16_f32
    x32, w32, dy32 = rnd(N, H, W, Cin, seed=80).float(), rnd(k, k, Cin, Cout, seed=81, scale=1.0 / math.sqrt(k * k * Cin)).float(), None
    x, w = bf(x32).requires_grad_(True), bf(w32).requires_grad_(True)
    b = rnd(Cout, seed=82)
    y = R.conv2d_same(nchw(x), w, 1, dil, bias=b)
    dy32 = rnd(*y.shape, seed=83).float()
    dy = bf(dy32)
    gx, gw = torch.autograd.grad(y, [x, w], dy)
    xg, wg, dyg = x32.to(d), w32.to(d), nhwc(dy32).contiguous().to(d)
    close(ops.conv2d_fwd(xg, wg, f32(b, d), dil, precision="bf16"), nhwc(y), 2e-5, "bf16 conv fwd")
    close(ops.conv2d_bwd_data(dyg, wg, dil, precision="bf16"), gx, 1e-4, "bf16 conv bwd data")
    close(ops.conv2d_bwd_filter(xg, dyg, k, dil, precision="bf16"), gw, 1e-4, "bf16 conv bwd filter")
    # the precision is an argument of each call: an fp32 call right after gives the fp32 result of the UNROUNDED operands
    y32 = R.conv2d_same(nchw(x32.double()), w32.double(), 1, dil, bias=b)
    close(ops.conv2d_fwd(xg, wg, f32(b, d), dil), nhwc(y32), 2e-5, "fp32 conv fwd after a bf16 call")
    with pytest.raises(Exception):
        ops.conv2d_fwd(xg, wg, f32(b, d), dil, precision="fp4")


@pytest.mark.parametrize("H,Cin,Cout,swish", [(32, 24, 144, False), (32, 112, 40, True), (32, 16, 20, False), (40, 96, 672, True), (23, 40, 8, False)])
def test_conv1x1_short_k_streaming_kernel(H, Cin, Cout, swish):
    """1x1 convs with K <= 112 and >= 1024 pixels take conv1x1_stream_k (no LDS, no barriers): forward with bias and fused BN
    statistics into a channel slice of a wider buffer, from a channel slice of a wider input; backward-data likewise -- K tails
    (24, 40), column tails (20, 40, 8), every KC 1..7."""
    from mliis_amd import ops
    d = dev()
    N = 2
    assert "conv1x1_stream_k" in ops.conv2d_kernel_name(N, H, H, Cin, Cout, 1)
    xw = rnd(N, H, H, Cin + 8, seed=90)
    x = xw[..., 4:4 + Cin].clone().requires_grad_(True)
    w = rnd(1, 1, Cin, Cout, seed=91, scale=1.0 / math.sqrt(Cin)).requires_grad_(True)
    b = rnd(Cout, seed=92)
    y = R.conv2d_same(nchw(x), w, 1, 1, bias=b)
    dy = rnd(*y.shape, seed=93)
    (gx,) = torch.autograd.grad(y, [x], dy)
    xg = f32(xw, d)
    out = torch.full((N, H, H, Cout + 12), 5.0, device=d)
    part = torch.full((1 << 19,), 7.0, device=d)
    _, nblk = ops.conv2d_fwd(xg[..., 4:4 + Cin], f32(w, d), f32(b, d), 1, out=out[..., 8:8 + Cout], stats_part=part, stats_swish=swish)
    close(out[..., 8:8 + Cout], nhwc(y), 2e-5, "stream fwd")
    assert (out[..., :8] == 5).all() and (out[..., 8 + Cout:] == 5).all() and nblk > 0
    v = nhwc(y).detach()
    v = R.swish(v) if swish else v
    sums = part[: nblk * 2 * Cout].view(nblk, 2, Cout).double().sum(0).cpu()
    close(sums[0], v.sum(dim=(0, 1, 2)), 2e-5, "stream fused sum")
    close(sums[1], (v * v).sum(dim=(0, 1, 2)), 2e-5, "stream fused sum of squares")
    assert (part[nblk * 2 * Cout:] == 7).all()
    # backward-data of a conv whose OUTPUT has few channels: K = Cout of the forward
    if Cout <= 112:
        assert "conv1x1_stream_k" in ops.conv2d_kernel_name(N, H, H, Cout, Cin, 1)
        dxw = torch.full((N, H, H, Cin + 4), -3.0, device=d)
        ops.conv2d_bwd_data(f32(nhwc(dy), d), f32(w, d), 1, out=dxw[..., :Cin])
        close(dxw[..., :Cin], gx, 1e-4, "stream bwd data")
        assert (dxw[..., Cin:] == -3).all()


@pytest.mark.parametrize("k,s,H,W,C", [(3, 1, 14, 14, 32), (3, 2, 16, 16, 24), (5, 1, 14, 14, 40), (5, 2, 28, 28, 16), (3, 2, 15, 17, 8), (5, 1, 7, 30, 144)])
def test_dwconv_bwd_data_emits_bn_backward_stage1(k, s, H, W, C):
    """mliis_dwconv_bwd_data_bn: same dx as the plain call, plus {sum g, sum g * xhat} partials that make mliis_bn_bwd skip its reduce
    pass -- dx / dgamma / dbeta of the batch norm then equal the two-pass path and the float64 oracle."""
    from mliis_amd import ops
    d = dev()
    N = 2
    z = rnd(N, H, W, C, seed=100).requires_grad_(True)                    # expand conv output (the BN input)
    gamma, beta = (rnd(C, seed=101) * 0.3 + 1.0).requires_grad_(True), (rnd(C, seed=102) * 0.2).requires_grad_(True)
    wdw = rnd(k, k, C, 1, seed=103, scale=0.3)
    mean = z.mean(dim=(0, 1, 2)).detach()
    var = z.var(dim=(0, 1, 2), unbiased=False).detach()
    rstd = 1.0 / torch.sqrt(var + 1e-3)
    zh = (z - z.mean(dim=(0, 1, 2))) / torch.sqrt(z.var(dim=(0, 1, 2), unbiased=False) + 1e-3)
    a = R.swish(zh * gamma + beta)                                        # batch-statistics BN + swish, as in training
    y = R.conv2d_same(nchw(a), wdw, s, 1, groups=C)
    dy = rnd(*y.shape, seed=104)
    gz, gg, gb = torch.autograd.grad(y, [z, gamma, beta], dy)
    dyg, wg = f32(nhwc(dy), d), f32(wdw, d)
    part = torch.full((1 << 16,), 3.0, device=d)
    ref_da = ops.dwconv_bwd_data(dyg, wg, s, (H, W))
    da, nblk = ops.dwconv_bwd_data(dyg, wg, s, (H, W), bn=(f32(z, d), f32(mean, d), f32(rstd, d), f32(gamma, d), f32(beta, d)), part=part)
    assert nblk > 0
    close(da, ref_da, 1e-6, "dx with statistics == dx without")
    args = (f32(z, d), da, f32(mean, d), f32(rstd, d), f32(gamma, d), f32(beta, d), False, True)
    dx1, dg1, db1 = ops.bn_bwd(*args, stage1=(part, nblk))
    dx2, dg2, db2 = ops.bn_bwd(*args)
    close(dx1, dx2, 1e-5, "fused stage 1 == reduce pass (dx)")
    close(dg1, dg2, 1e-5, "dgamma")
    close(db1, db2, 1e-5, "dbeta")
    close(dx1, gz, 2e-4, "dx vs oracle")
    close(dg1, gg, 2e-4, "dgamma vs oracle")
    close(db1, gb, 2e-4, "dbeta vs oracle")


# ------------------------------------------------------------------------------------------------ small-map MBConv depthwise half
@pytest.mark.parametrize("k,N,H,W,C,V", [(3, 8, 14, 14, 480, 0), (5, 8, 14, 14, 672, 0), (5, 5, 14, 14, 480, 0), (3, 8, 4, 4, 480, 0), (5, 8, 16, 16, 40, 0),
                                         (3, 2, 24, 24, 8, 0), (5, 3, 7, 9, 72, 0), (3, 2, 2, 3, 16, 0),
                                         (5, 8, 14, 14, 816, 0), (3, 8, 14, 14, 1152, 0),          # EfficientNet-B3's 14x14 / widest layers: quads
                                         (5, 8, 14, 14, 480, 4), (3, 8, 14, 14, 672, 2), (5, 8, 14, 14, 672, 2), (3, 8, 14, 14, 480, 2),
                                         (5, 3, 7, 9, 72, 4), (3, 2, 2, 3, 12, 2)])
def test_mbconv_small_fused_fwd_bwd(k, N, H, W, C, V):
    """mliis_mbconv_dw_fwd_small / _bwd_small (one launch per direction for the depthwise half of an MBConv block on small maps) vs the
    float64 oracle ops + autograd: expand BN (statistics handed over as stage-1 partials) -> swish -> depthwise k x k -> BN -> swish ->
    per-image means; both moving averages; backward with the squeeze-excite gate / pooled-gradient terms.  Shapes: the 14x14 layers of
    EfficientLab-6-3 at N = 8 and N = 5 (FOMAML tail batch), the 64x64-input test sizes, ragged maps, a channel count that leaves
    workgroups of the XCD-grouped grid idle, a map smaller than one strip.  V = channels per workgroup: 0 = the planner's choice (quads; pairs for
    5x5 layers of up to 512 channels), or forced: every (k, V) instantiation is covered."""
    from mliis_amd import ops
    from mliis_amd.spec import BN_EPS
    d = dev()
    assert ops.mbconv_dw_small_supported(N, H, W, C, k, 1)
    assert ops.mbconv_dw_small_group_width(C, k) == (2 if (k == 5 and C <= 512) else 4)
    assert not ops.mbconv_dw_small_supported(N, H, W, C, k, 2) and not ops.mbconv_dw_small_supported(64, 14, 14, C, k, 1)
    z0 = (rnd(N, H, W, C, seed=1) * 1.5 + 0.3).requires_grad_(True)
    wd = rnd(k, k, C, 1, seed=2, scale=0.4).requires_grad_(True)
    g0, b0 = (1 + 0.2 * rnd(C, seed=3)).requires_grad_(True), (0.3 * rnd(C, seed=4)).requires_grad_(True)
    g1, b1 = (1 + 0.2 * rnd(C, seed=5)).requires_grad_(True), (0.3 * rnd(C, seed=6)).requires_grad_(True)
    gate, cadd, da2 = torch.sigmoid(rnd(N, C, seed=7)), 0.01 * rnd(N, C, seed=8), rnd(N, H, W, C, seed=9)
    mm0, mv0, mm1, mv1 = 0.1 * rnd(C, seed=10), 1 + 0.1 * rnd(C, seed=11).abs(), 0.1 * rnd(C, seed=12), 1 + 0.1 * rnd(C, seed=13).abs()

    def bn_ref(x, g, b):
        m = x.mean(dim=(0, 1, 2))
        v = ((x - m) ** 2).mean(dim=(0, 1, 2))
        return (x - m) / torch.sqrt(v + BN_EPS) * g + b, m, v
    y0, m0, v0 = bn_ref(z0, g0, b0)
    a0 = R.swish(y0)
    z1 = nhwc(R.conv2d_same(nchw(a0), wd, 1, groups=C))
    y1, m1, v1 = bn_ref(z1, g1, b1)
    a1 = R.swish(y1)
    s_ref = a1.mean(dim=(1, 2))
    up = da2 * gate[:, None, None, :] + cadd[:, None, None, :]
    grads = torch.autograd.grad((a1 * up).sum(), [z0, wd, g0, b0, g1, b1])
    # device: stage-1 statistics of z0 as the expand conv's epilogue would leave them
    z0d = f32(z0, d)
    part = torch.zeros(1 << 18, device=d)
    nblk = ops.bn_stats_partial(z0d, False, part)
    st = [torch.zeros(C, device=d) for _ in range(4)]
    mov = [f32(t, d) for t in (mm0, mv0, mm1, mv1)]
    a0d, z1d, a1d, sd = (torch.full((N, H, W, C), 9.0, device=d) for _ in range(3)), None, None, None
    a0d, z1d, a1d = a0d
    sd = torch.full((N, C), 9.0, device=d)
    ops.mbconv_dw_fwd_small(z0d, part, nblk, (f32(g0, d), f32(b0, d), st[0], st[1], mov[0], mov[1]), f32(wd, d),
                            (f32(g1, d), f32(b1, d), st[2], st[3], mov[2], mov[3]), z1d, a1d, sd, a0=a0d, group_width=V)
    close(a0d, a0, 2e-5, "small fwd a0")
    close(z1d, z1, 2e-5, "small fwd z1")
    close(a1d, a1, 5e-5, "small fwd a1")
    close(sd, s_ref, 2e-5, "small fwd pooled mean")
    close(st[0], m0, 1e-5, "mean0"); close(st[1], 1 / torch.sqrt(v0 + BN_EPS), 1e-5, "rstd0")
    close(st[2], m1, 2e-5, "mean1"); close(st[3], 1 / torch.sqrt(v1 + BN_EPS), 2e-5, "rstd1")
    for got, old, stat in ((mov[0], mm0, m0), (mov[1], mv0, v0), (mov[2], mm1, m1), (mov[3], mv1, v1)):
        close(got, old - (old - stat.detach()) * 0.01, 1e-5, "moving average")   # biased variance (non-fused BN, utils.py:87-134)
    # a0 is optional
    z1b, a1b, sb = torch.zeros_like(z1d), torch.zeros_like(a1d), torch.zeros_like(sd)
    mov2 = [f32(t, d) for t in (mm0, mv0, mm1, mv1)]
    ops.mbconv_dw_fwd_small(z0d, part, nblk, (f32(g0, d), f32(b0, d), st[0], st[1], mov2[0], mov2[1]), f32(wd, d),
                            (f32(g1, d), f32(b1, d), st[2], st[3], mov2[2], mov2[3]), z1b, a1b, sb, group_width=V)
    assert torch.equal(z1b, z1d) and torch.equal(a1b, a1d) and torch.equal(sb, sd)      # deterministic
    # backward
    outs = dict(dg1=torch.zeros(C, device=d), db1=torch.zeros(C, device=d), dw=torch.zeros(k, k, C, 1, device=d), dg0=torch.zeros(C, device=d),
                db0=torch.zeros(C, device=d), dz0=torch.full((N, H, W, C), 9.0, device=d))
    ops.mbconv_dw_bwd_small(f32(da2, d), f32(gate, d), f32(cadd, d), z1d, (st[2], st[3], f32(g1, d), f32(b1, d)), f32(wd, d), z0d,
                            (st[0], st[1], f32(g0, d), f32(b0, d)), outs["dg1"], outs["db1"], outs["dw"], outs["dg0"], outs["db0"], outs["dz0"],
                            group_width=V)
    for name, ref in (("dz0", grads[0]), ("dw", grads[1]), ("dg0", grads[2]), ("db0", grads[3]), ("dg1", grads[4]), ("db1", grads[5])):
        close(outs[name], ref, 2e-4, "small bwd " + name)
    # the saved tensors in the group-blocked layout [C / V][N H W][V] (what the learner uses): same results bit for bit
    Vb = V or ops.mbconv_dw_small_group_width(C, k)
    z0blk, z1blk, a1c, sc = torch.zeros_like(z0d), torch.zeros_like(z1d), torch.zeros_like(a1d), torch.zeros_like(sd)
    mov3 = [f32(t, d) for t in (mm0, mv0, mm1, mv1)]
    ops.mbconv_dw_fwd_small(z0d, part, nblk, (f32(g0, d), f32(b0, d), st[0], st[1], mov3[0], mov3[1]), f32(wd, d),
                            (f32(g1, d), f32(b1, d), st[2], st[3], mov3[2], mov3[3]), z1blk, a1c, sc, group_width=V, z0_blocked=z0blk, z1_blocked=True)
    unblock = lambda t: t.view(C // Vb, N * H * W, Vb).permute(1, 0, 2).reshape(N, H, W, C)  # noqa: E731
    assert torch.equal(unblock(z1blk), z1d) and torch.equal(unblock(z0blk), z0d) and torch.equal(a1c, a1d) and torch.equal(sc, sd)
    outs2 = {k_: torch.zeros_like(v) for k_, v in outs.items()}
    ops.mbconv_dw_bwd_small(f32(da2, d), f32(gate, d), f32(cadd, d), z1blk, (st[2], st[3], f32(g1, d), f32(b1, d)), f32(wd, d), z0d,
                            (st[0], st[1], f32(g0, d), f32(b0, d)), outs2["dg1"], outs2["db1"], outs2["dw"], outs2["dg0"], outs2["db0"], outs2["dz0"],
                            group_width=V, z0_blocked=z0blk, z1_blocked=True)
    for name in outs:
        assert torch.equal(outs2[name], outs[name]), "blocked layout: " + name
    # ... and with the operands ARRIVING blocked (z0 written blocked by the expand conv: z0_blocked is z0; da2 by the project conv's
    # backward-data): same bits again
    blk = lambda t: t.reshape(N * H * W, C // Vb, Vb).permute(1, 0, 2).contiguous().view(N, H, W, C)   # noqa: E731
    z0in, z1c, a1d2, sd2 = blk(z0d), torch.zeros_like(z1d), torch.zeros_like(a1d), torch.zeros_like(sd)
    mov4 = [f32(t, d) for t in (mm0, mv0, mm1, mv1)]
    ops.mbconv_dw_fwd_small(z0in, part, nblk, (f32(g0, d), f32(b0, d), st[0], st[1], mov4[0], mov4[1]), f32(wd, d),
                            (f32(g1, d), f32(b1, d), st[2], st[3], mov4[2], mov4[3]), z1c, a1d2, sd2, group_width=V, z0_blocked=z0in, z1_blocked=True)
    assert torch.equal(z1c, z1blk) and torch.equal(a1d2, a1d) and torch.equal(sd2, sd) and torch.equal(z0in, blk(z0d))
    outs3 = {k_: torch.zeros_like(v) for k_, v in outs.items()}
    ops.mbconv_dw_bwd_small(blk(f32(da2, d)), f32(gate, d), f32(cadd, d), z1blk, (st[2], st[3], f32(g1, d), f32(b1, d)), f32(wd, d), z0d,
                            (st[0], st[1], f32(g0, d), f32(b0, d)), outs3["dg1"], outs3["db1"], outs3["dw"], outs3["dg0"], outs3["db0"], outs3["dz0"],
                            group_width=V, z0_blocked=z0in, z1_blocked=True, da2_blocked=True)
    for name in outs:
        assert torch.equal(outs3[name], outs[name]), "blocked operands: " + name
    with pytest.raises(Exception):
        ops.mbconv_dw_fwd_small(torch.zeros(64, 14, 14, C, device=d), part, nblk, (f32(g0, d), f32(b0, d), st[0], st[1], None, None), f32(wd, d),
                                (f32(g1, d), f32(b1, d), st[2], st[3], None, None), z1b, a1b, sb)


@pytest.mark.parametrize("N,H,Cin,Cout,V", [(8, 14, 80, 480, 4), (8, 14, 112, 672, 4), (8, 14, 80, 480, 2), (6, 14, 112, 672, 4), (8, 16, 40, 96, 2)])
def test_streamed_1x1_convs_write_the_group_blocked_layout(N, H, Cin, Cout, V):
    """MLIIS_DT_BLOCKED: the streamed 1x1 conv (MBConv expand forward; project backward-data with the gate-gradient partials) writes its
    output as [C / v][N H W][v] -- bit for bit the row-major result, permuted; statistics and gate partials unchanged.  A call the streamed
    plan does not take refuses the layout."""
    from mliis_amd import ops
    d = dev()
    assert ops.conv1x1_stream_eligible(N, H, H, Cin, Cout)
    blk = lambda t, C_: t.reshape(N * H * H, C_ // V, V).permute(1, 0, 2).contiguous().view(N, H, H, C_)   # noqa: E731
    x, w = f32(rnd(N, H, H, Cin, seed=1), d), f32(rnd(1, 1, Cin, Cout, seed=2, scale=0.1), d)
    pa, pb = torch.zeros(1 << 20, device=d), torch.zeros(1 << 20, device=d)
    y, na = ops.conv2d_fwd(x, w, None, 1, stats_part=pa)
    yb, nb = ops.conv2d_fwd(x, w, None, 1, stats_part=pb, out=torch.full_like(y, 9.0), out_block=V)
    assert na == nb and na > 0 and torch.equal(pa[:na * 2 * Cout], pb[:nb * 2 * Cout])
    assert torch.equal(yb, blk(y, Cout))
    # project backward-data: dy [.., Cin] -> da2 [.., Cout] through w2 [1,1,Cout,Cin], gate partials beside it
    dy, w2, a1 = f32(rnd(N, H, H, Cin, seed=3), d), f32(rnd(1, 1, Cout, Cin, seed=4, scale=0.1), d), f32(rnd(N, H, H, Cout, seed=5), d)
    ga, gb = torch.zeros(1 << 20, device=d), torch.zeros(1 << 20, device=d)
    da, gra = ops.conv2d_bwd_data(dy, w2, 1, gate=a1, part=ga)
    db, grb = ops.conv2d_bwd_data(dy, w2, 1, gate=a1, part=gb, out=torch.full_like(da, 9.0), out_block=V)
    assert gra == grb and gra > 0 and torch.equal(ga[:gra * 2 * Cout], gb[:grb * 2 * Cout])
    assert torch.equal(db, blk(da, Cout))
    with pytest.raises(Exception):     # a long-K conv (K-split plan) cannot write it
        ops.conv2d_fwd(f32(rnd(N, H, H, 480, seed=6), d), f32(rnd(1, 1, 480, 80, seed=7), d), None, 1, out_block=4)


# ------------------------------------------------------------------------------------------------ data-parallel + stream-K remainder
@pytest.mark.parametrize("k,dil,H,Cin,Cout,N", [(3, 1, 56, 72, 112, 8), (3, 2, 56, 64, 136, 8), (3, 1, 56, 56, 224, 8), (1, 1, 64, 512, 112, 9)])   # (the 1x1 case: 36864 rows, beyond the in-workgroup K split of the small maps)
def test_conv2d_stream_k_remainder(k, dil, H, Cin, Cout, N):
    """Long-K layers whose tile count is not a multiple of the 256 CUs (the 56x56 decoder convs: 392 / 784 / 1176 tiles of 64 rows) run
    their first floor(T / 256) * 256 tiles whole and cut the K range of the rest into equal parts (conv_gemm_sk_k + sk_fixup_k): forward
    with bias, border-class bias and fused BN statistics (of swish(z)), accumulate, and backward-data, against the float64 oracle."""
    from mliis_amd import ops
    d = dev()
    name = ops.conv2d_kernel_name(N, H, H, Cin, Cout, k)
    assert name.startswith("conv_gemm_sk_k"), name
    x = rnd(N, H, H, Cin, seed=90).requires_grad_(True)
    w = rnd(k, k, Cin, Cout, seed=91, scale=1.0 / math.sqrt(k * k * Cin)).requires_grad_(True)
    b = rnd(Cout, seed=92)
    z = nhwc(R.conv2d_same(nchw(x), w, 1, dil, bias=b))
    dy = rnd(*z.shape, seed=93)
    (gx,) = torch.autograd.grad(z, [x], dy)
    xg, wg, bg = f32(x, d), f32(w, d), f32(b, d)
    part = torch.full((1 << 20,), 3.0, device=d)
    zg, nblk = ops.conv2d_fwd(xg, wg, bg, dil, stats_part=part, stats_swish=True)
    close(zg, z, 2e-5, "stream-K conv fwd")
    assert nblk == -(-N * H * H // 64)
    u = R.swish(z.detach())
    sums = part[: nblk * 2 * Cout].view(nblk, 2, Cout).double().sum(0).cpu()
    close(sums[0], u.sum(dim=(0, 1, 2)), 1e-5, "stream-K fused sum")
    close(sums[1], (u * u).sum(dim=(0, 1, 2)), 1e-5, "stream-K fused sum of squares")
    # deterministic, and identical to the plain data-parallel result up to the order of the K-range partial sums
    zg2, _ = ops.conv2d_fwd(xg, wg, bg, dil, stats_part=part, stats_swish=True)
    assert torch.equal(zg, zg2)
    # accumulate into an existing tensor
    base = f32(rnd(N, H, H, Cout, seed=94), d)
    acc = base.clone()
    ops.conv2d_fwd(xg, wg, bg, dil, out=acc, accumulate=True)
    close(acc - base, z, 3e-5, "stream-K accumulate")
    if k == 3 and dil == 1:   # border-class bias of spatially constant channels (the RSD pooled branch)
        bb = rnd(N, 9, Cout, seed=95)
        hh = torch.arange(H)
        cls = (torch.where(hh == 0, 0, torch.where(hh == H - 1, 2, 1))[:, None] * 3 + torch.where(hh == 0, 0, torch.where(hh == H - 1, 2, 1))[None, :])
        ref = z.detach() + bb[:, cls.reshape(-1), :].reshape(N, H, H, Cout)
        close(ops.conv2d_fwd(xg, wg, bg, dil, border_bias=f32(bb, d)), ref, 2e-5, "stream-K border bias")
    # backward-data: reduction over (taps, Cout), output Cin columns -- its own plan (stream-K when it qualifies)
    close(ops.conv2d_bwd_data(f32(dy, d), wg, dil), gx, 1e-4, "conv bwd data")


# ------------------------------------------------------------------------------------------------ device RNG (masks)
def _philox4x32_10(c, k):
    """Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11), restated for the test."""
    M0, M1, W0, W1, mask = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85, 0xFFFFFFFF
    c, k = list(c), list(k)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [(p1 >> 32) ^ c[1] ^ k[0], p1 & mask, (p0 >> 32) ^ c[3] ^ k[1], p0 & mask]
        k = [(k[0] + W0) & mask, (k[1] + W1) & mask]
    return c


def test_rng_masks_are_philox_and_advance_per_launch():
    """mliis_rng_masks: drop-connect scales floor(keep + u) / keep per (block, image) and dropout masks (u < keep) / keep per element,
    u = 24 high bits of Philox4x32-10(counter = (quad index, 0, step, job), key = seed) / 2^24 -- bit for bit against a Python
    restatement (known-answer: the all-zero counter / key block of the Random123 test vectors); the launch advances the step itself."""
    from mliis_amd import ops
    d = dev()
    assert _philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]     # Random123 kat_vectors
    assert _philox4x32_10([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    seed = 0x1234_5678_9ABC_DEF1
    state = ops.rng_state(seed, d)
    keeps = torch.tensor([0.9636, 0.9273, 0.8182], device=d)
    dc = torch.zeros(3, 5, device=d)
    drop = torch.zeros(1003, device=d)       # not a multiple of 4: ragged last quad
    keepv = torch.tensor([0.5], device=d)
    skipped = None
    plan = ops.MaskPlan([(dc, keeps, 5, True), (skipped, 0.5, 1, False), (drop, keepv, drop.numel(), False)])
    k = [seed & 0xFFFFFFFF, seed >> 32]
    for step in range(3):
        ops.rng_masks(state, plan)
        torch.cuda.synchronize()
        assert state.cpu().tolist()[2:] == [step + 1, 0]
        u = lambda job, i: (_philox4x32_10([i // 4, 0, step, job], k)[i % 4] >> 8) / 16777216.0  # noqa: E731
        kk = keeps.cpu().numpy()
        exp_dc = [[float(np.float32(np.floor(np.float32(kk[b]) + np.float32(u(0, b * 5 + n)))) / np.float32(kk[b])) for n in range(5)] for b in range(3)]
        np.testing.assert_array_equal(dc.cpu().numpy(), np.array(exp_dc, dtype=np.float32))
        exp_drop = np.array([2.0 if u(2, i) < 0.5 else 0.0 for i in range(1003)], dtype=np.float32)
        np.testing.assert_array_equal(drop.cpu().numpy(), exp_drop)
    big = torch.zeros(1 << 20, device=d)
    ops.rng_masks(state, ops.MaskPlan([(big, 0.8, 1, False)]))
    assert abs((big == 0).float().mean().item() - 0.2) < 2e-3 and abs(big.max().item() - 1.25) < 1e-6
    state2 = ops.rng_state(seed + 1, d)      # another seed, another stream
    big2 = torch.zeros(1 << 20, device=d)
    ops.rng_masks(state2, ops.MaskPlan([(big2, 0.8, 1, False)]))
    assert 0.6 < (big == big2).float().mean().item() < 0.75     # P(agree) = 0.8^2 + 0.2^2 = 0.68 for independent streams


def test_masks_drawn_by_the_weight_shadow_launch_equal_the_mask_launch():
    """mliis_weight_shadows_rng (round 6: the masks of a training step ride in the launch that builds the weight shadows -- one launch less
    per step): same masks and same generator state as mliis_rng_masks, same K-contiguous weight copies and split-product images as
    mliis_weight_shadows, over three steps; with and without weight images; a mask tensor large enough for several mask workgroups."""
    from mliis_amd import ops
    d = dev()
    theta = f32(rnd(3 * 3 * 40 * 48 + 1 * 1 * 48 * 24, seed=5), d)
    rows = [[0, 9, 40, 48], [9 * 40 * 48, 1, 48, 24]]
    desc = torch.tensor(rows, dtype=torch.int32, device=d)
    tiles = ops.transpose_tiles(rows)

    def images():
        im = ops.X3Images(d)
        im.add("w", "fwd", 0, 3, 40, 48)
        im.add("w", "bwd", 0, 3, 40, 48)
        return im.finish()
    for with_images in (True, False):
        sa, sb = ops.rng_state(77, d), ops.rng_state(77, d)
        keeps = torch.tensor([0.9, 0.8], device=d)
        ma, mb = [torch.zeros(2, 8, device=d), torch.zeros(5000, device=d)], [torch.zeros(2, 8, device=d), torch.zeros(5000, device=d)]
        pa = ops.MaskPlan([(ma[0], keeps, 8, True), (ma[1], 0.5, 1, False)])
        pb = ops.MaskPlan([(mb[0], keeps, 8, True), (mb[1], 0.5, 1, False)])
        ta, tb = torch.zeros_like(theta), torch.zeros_like(theta)
        xa, xb = (images(), images()) if with_images else (None, None)
        for step in range(3):
            ops.rng_masks(sa, pa)
            ops.transpose_weights(theta, ta, desc, tiles=tiles, x3=xa)
            ops.transpose_weights(theta, tb, desc, tiles=tiles, x3=xb, rng=(sb, pb))
            torch.cuda.synchronize()
            assert sb.cpu().tolist() == sa.cpu().tolist() and sa.cpu().tolist()[2:] == [step + 1, 0]
            assert all(torch.equal(u, v) for u, v in zip(ma, mb)) and torch.equal(ta, tb)
            if with_images:
                assert torch.equal(xa.images, xb.images)
        assert 0.45 < (ma[1] == 0).float().mean().item() < 0.55


# ------------------------------------------------------------------------------------------------ fp8 (OCP e4m3) operands, 1x1 forward
@pytest.mark.parametrize("H,Cin,Cout,N,gated", [(32, 24, 144, 2, False), (28, 240, 40, 2, True), (14, 672, 112, 2, True), (48, 40, 240, 2, False),
                                                (14, 112, 672, 8, False)])
def test_conv1x1_fp8_operands(H, Cin, Cout, N, gated):
    """precision = MLIIS_PREC_FP8 (BASELINE configs[4]): 1x1 forward convs with e4m3 operands -- activations x 16, weights x
    2^floor(log2(224 / max|w|)), saturated at +-448, converted in registers; fp32 accumulation, the scales divided out.  Equals the
    float64 conv of the identically quantised operands (torch.float8_e4m3fn) to 2e-5: the streaming kernel (K <= 112), the LDS-tiled
    kernel with the squeeze-excite gate applied on load, split-K plans; fused BN statistics ride along.  Backward calls in this mode take
    bf16 operands."""
    from mliis_amd import ops
    d = dev()
    x32 = (rnd(N, H, H, Cin, seed=70) * 1.3).float()
    x32[0, 0, 0, 0] = 40.0                                        # beyond 448 / 16: saturates
    w32 = rnd(1, 1, Cin, Cout, seed=71, scale=1.0 / math.sqrt(Cin)).float()
    gate = torch.sigmoid(rnd(N, Cin, seed=72)).float() if gated else None
    xs = x32 * gate[:, None, None, :] if gated else x32          # (the product multiplies in fp32, then converts)
    xq = R.round_fp8(xs, R.FP8_ACT_SCALE).double()
    wq = R.round_fp8(w32, R.fp8_weight_scale(w32)).double()
    b = rnd(Cout, seed=73)
    z = nhwc(R.conv2d_same(nchw(xq), wq, 1, 1, bias=b))
    part = torch.zeros(1 << 18, device=d)
    amax = w32.abs().max().reshape(1).to(d)
    zg, nblk = ops.conv2d_fwd(x32.to(d), w32.to(d), f32(b, d), 1, precision="fp8", x_scale=gate.to(d) if gated else None,
                              stats_part=part, fp8_w_amax=amax)
    close(zg, z, 2e-4, "fp8 conv fwd")   # (the fp8 MFMA aligns the 32 products of a K block before adding: ~2^-14 relative, measured)
    if nblk:
        sums = part[: nblk * 2 * Cout].view(nblk, 2, Cout).double().sum(0).cpu()
        close(sums[0], z.sum(dim=(0, 1, 2)), 2e-4, "fp8 fused sum")
    # weights' amax comes from the weight-shadow launch in the learner
    wt = torch.empty(Cin * Cout, device=d)
    am2 = torch.full((1,), 5.0, device=d)
    ops.transpose_weights(w32.to(d).view(-1), wt, torch.tensor([[0, 1, Cin, Cout]], dtype=torch.int32, device=d), am2)
    assert am2.item() == w32.abs().max().item()
    close(wt.view(Cout, Cin), w32.view(Cin, Cout).t().double(), 0.0, "weight shadow")
    # backward calls given the fp8 mode take bf16 operands
    bf = lambda t: t.to(torch.bfloat16).to(torch.float64)   # noqa: E731
    dy32 = rnd(N, H, H, Cout, seed=74).float()
    xb, wb = bf(xs).requires_grad_(True), bf(w32).requires_grad_(True)
    gx, gw = torch.autograd.grad(R.conv2d_same(nchw(xb), wb, 1, 1), [xb, wb], nchw(bf(dy32)))
    close(ops.conv2d_bwd_data(dy32.to(d), w32.to(d), 1, precision="fp8"), gx if not gated else gx, 1e-4, "fp8-mode bwd data (bf16)")
    close(ops.conv2d_bwd_filter(x32.to(d), dy32.to(d), 1, 1, precision="fp8", x_scale=gate.to(d) if gated else None), gw, 1e-4,
          "fp8-mode bwd filter (bf16)")


def test_conv2d_bwd_filter_batched_equals_the_single_calls():
    """ops.FilterBatch / mliis_conv2d_bwd_filter_batched: several filter-gradient problems as one launch per kernel instantiation
    leave exactly the slabs the single calls leave (bit for bit: same blocks, same order of summation), for 1x1 convs with and
    without the squeeze-excite gate on load, two map sizes and a 3x3 conv."""
    from mliis_amd import ops
    d = dev()
    probs = []
    for i, (N, H, Cin, Cout, k, dil, gated) in enumerate([(8, 14, 480, 80, 1, 1, True), (8, 14, 672, 112, 1, 1, True), (8, 14, 480, 112, 1, 1, True),
                                                          (8, 14, 80, 480, 1, 1, False), (8, 14, 112, 672, 1, 1, False), (5, 28, 40, 240, 1, 1, False),
                                                          (8, 28, 240, 40, 1, 1, True), (2, 14, 224, 112, 3, 2, False)]):
        x = f32(rnd(N, H, H, Cin, seed=200 + i), d)
        dy = f32(rnd(N, H, H, Cout, seed=300 + i), d)
        gate = f32(torch.sigmoid(rnd(N, Cin, seed=400 + i)), d) if gated else None
        n = ops.lib.size("mliis_conv2d_bwd_filter_workspace_floats", N, H, H, Cin, Cout, k)
        probs.append((x, dy, k, dil, gate, torch.full((n,), 9.0, device=d), torch.full((n,), -9.0, device=d)))
    fb = ops.FilterBatch(d)
    for x, dy, k, dil, gate, pa, pb in probs:
        ops.conv2d_bwd_filter(x, dy, k, dil, x_scale=gate, partial=pa)
        fb.add(x, dy, k, dil, pb, x_scale=gate)
    assert len(fb) == len(probs)
    fb.launch()
    fb.launch()          # idempotent: the slabs are overwritten, not accumulated
    torch.cuda.synchronize()
    assert len(fb.tables) < len(probs)     # problems sharing an instantiation share a launch
    for i, (x, dy, k, dil, gate, pa, pb) in enumerate(probs):
        assert torch.equal(pa, pb), i


def test_small_filter_gradient_groups_join_the_64_column_launch(monkeypatch):
    """FilterBatch re-tiles a (64-channel-block, 1 / 2 / 3 / 6 column tiles) group of fewer than MERGE_MAX_BLOCKS workgroups to four
    column tiles so that it rides in that group's launch (profiles/r06_notes.md section 8): fewer launches, and every slab bit-identical
    to the launch-per-width form -- output widths of 16, 24, 40, 96 and 144 columns (tile widths 1, 2, 3, 6, 3 of their own plans; the
    kernels mask the columns beyond Cout), gated and ungated; a group above the bound keeps its own launch."""
    from mliis_amd import ops
    d = dev()
    shapes = [(8, 28, 32, 16, True), (8, 28, 96, 24, True), (8, 28, 144, 40, True), (8, 28, 16, 96, False), (8, 28, 24, 144, False),
              (8, 14, 240, 80, True), (8, 14, 80, 480, False), (4, 28, 24, 144, False)]
    probs = []
    for i, (N, H, Cin, Cout, gated) in enumerate(shapes):
        x = f32(rnd(N, H, H, Cin, seed=600 + i), d)
        dy = f32(rnd(N, H, H, Cout, seed=700 + i), d)
        gate = f32(torch.sigmoid(rnd(N, Cin, seed=800 + i)), d) if gated else None
        n = ops.lib.size("mliis_conv2d_bwd_filter_workspace_floats", N, H, H, Cin, Cout, 1)
        probs.append((x, dy, gate, n))

    def run(bound):
        monkeypatch.setattr(ops.FilterBatch, "MERGE_MAX_BLOCKS", bound)
        fb = ops.FilterBatch(d)
        outs = []
        for x, dy, gate, n in probs:
            outs.append(torch.full((n,), float("nan"), device=d))
            fb.add(x, dy, 1, 1, outs[-1], x_scale=gate)
        fb.launch()
        torch.cuda.synchronize()
        return outs, [(t[3], t[4], t[5]) for t in fb.tables]

    plain, launches_plain = run(0)
    merged, launches_merged = run(2500)
    assert len(launches_merged) < len(launches_plain), (launches_plain, launches_merged)
    assert all(nt >= 4 or tmf != 1 for tmf, nt, _ in launches_merged), launches_merged
    for i, (a, b) in enumerate(zip(plain, merged)):
        assert not torch.isnan(a).any() and torch.equal(a, b), i
    some, launches_some = run(200)      # (only the groups below 200 workgroups move)
    assert len(launches_merged) <= len(launches_some) <= len(launches_plain)
    for i, (a, b) in enumerate(zip(plain, some)):
        assert torch.equal(a, b), i


def test_conv1x1_ksplit_random_shapes():
    """conv1x1_ksplit_k (long-K 1x1 convs on small maps) over 28 seeded random shapes: K 113..896 (every KC 1..7, K tails), 16..32768
    rows (row-group tails, maps from 2x2 up), 8..240 output channels (column tails), with / without bias, SE gate on load, fused
    statistics (plain / swish), accumulate, channel-slice views on both sides; forward and backward-data against float64."""
    from mliis_amd import ops
    d = dev()
    g = np.random.default_rng(123)
    seen = set()
    for case in range(28):
        K = int(g.integers(29, 225)) * 4
        N = int(g.integers(1, 9))
        H = int(g.integers(2, 33))
        W = int(g.integers(2, 33))
        while N * H * W > 32768 or N * H * W < 16:
            H, W = max(2, H // 2 + 1), max(2, W // 2 + 1)
            N = max(N, 2)
        Co = int(g.integers(2, 61)) * 4
        gated, bias, stats, swish, acc = [bool(v) for v in g.integers(0, 2, 5)]
        name = ops.conv2d_kernel_name(N, H, W, K, Co, 1, gated)
        assert "conv1x1_ksplit_k" in name, (name, N, H, W, K, Co)
        seen.add(name.split("<")[1].split(",")[0])
        xw = rnd(N, H, W, K + 8, seed=500 + case)
        x = xw[..., 4:4 + K]
        w = rnd(1, 1, K, Co, seed=600 + case, scale=1.0 / math.sqrt(K))
        b = rnd(Co, seed=700 + case) if bias else None
        gate = torch.sigmoid(rnd(N, K, seed=800 + case)) if gated else None
        xs = x * gate[:, None, None, :] if gated else x
        y = (xs.reshape(-1, K) @ w.reshape(K, Co) + (b if bias else 0)).reshape(N, H, W, Co)
        out = torch.full((N, H, W, Co + 12), 3.0, device=d)
        prev = f32(rnd(N, H, W, Co, seed=900 + case), d)
        if acc:
            out[..., 8:8 + Co] = prev
        part = torch.full((1 << 20,), 7.0, device=d) if (stats and not acc) else None
        r = ops.conv2d_fwd(f32(xw, d)[..., 4:4 + K], f32(w, d), f32(b, d) if bias else None, 1, out=out[..., 8:8 + Co], accumulate=acc,
                           stats_part=part, stats_swish=swish, x_scale=f32(gate, d) if gated else None)
        ref = y + (prev.cpu().double() if acc else 0)
        close(out[..., 8:8 + Co], ref, 2e-5, "ksplit fwd case %d" % case)
        assert (out[..., :8] == 3).all() and (out[..., 8 + Co:] == 3).all()
        if part is not None:
            nblk = r[1]
            assert nblk > 0
            v = R.swish(y) if swish else y
            sums = part[: nblk * 2 * Co].view(nblk, 2, Co).double().sum(0).cpu()
            close(sums[0], v.sum(dim=(0, 1, 2)), 2e-5, "ksplit fused sum %d" % case)
            close(sums[1], (v * v).sum(dim=(0, 1, 2)), 2e-5, "ksplit fused sum of squares %d" % case)
            assert (part[nblk * 2 * Co:] == 7).all()
        # backward-data of a conv with Cout = K of this case (the expand convs): dx = dy . W^T, here with K in the role of Cout
        dy = rnd(N, H, W, K, seed=1000 + case)
        w2 = rnd(1, 1, Co, K, seed=1100 + case, scale=1.0 / math.sqrt(K))
        if "conv1x1_ksplit_k" in ops.conv2d_kernel_name(N, H, W, K, Co, 1):
            dx = torch.full((N, H, W, Co + 4), -2.0, device=d)
            if acc:
                dx[..., :Co] = prev
            ops.conv2d_bwd_data(f32(dy, d), f32(w2, d), 1, out=dx[..., :Co], accumulate=acc)
            refx = (dy.reshape(-1, K) @ w2.reshape(Co, K).T).reshape(N, H, W, Co) + (prev.cpu().double() if acc else 0)
            close(dx[..., :Co], refx, 1e-4, "ksplit bwd data case %d" % case)
            assert (dx[..., Co:] == -2).all()
    assert len(seen) >= 5, seen          # several KC instances were hit


@pytest.mark.parametrize("N,H,K,C,scaled,acc", [(8, 14, 480, 80, True, True), (8, 14, 672, 112, False, False), (5, 28, 240, 40, True, False),
                                                (8, 14, 672, 112, True, True), (3, 7, 144, 24, False, True),
                                                # short K, no accumulate: the streaming kernel's per-wave epilogue (block 1's expand conv)
                                                (8, 112, 96, 16, False, False), (2, 40, 96, 24, True, False)])
def test_conv2d_bwd_data_emits_bn_backward_stage1(N, H, K, C, scaled, acc):
    """mliis_conv2d_bwd_data_bn: the backward-data of an expand conv (K = expanded channels) produces the gradient of the block in
    front, which is the OUTPUT gradient of that block's project batch norm; on the small-map plan the launch also leaves
    {sum g, sum g * xhat} (g = dx * drop-connect scale) so that mliis_bn_bwd skips its reduce pass.  Same dx as the plain call; the
    batch norm's dx / dgamma / dbeta equal the two-pass path (itself checked against the oracle in test_bn_train_fwd_bwd)."""
    from mliis_amd import ops
    d = dev()
    dy = f32(rnd(N, H, H, K, seed=31), d)
    w = f32(rnd(1, 1, C, K, seed=32, scale=1.0 / math.sqrt(K)), d)
    prev = f32(rnd(N, H, H, C, seed=33), d)
    z64 = rnd(N, H, H, C, seed=34) * 1.5 + 0.3                          # the batch norm's input
    z = f32(z64, d)
    gamma, beta = f32(rnd(C, seed=35) * 0.5 + 1, d), f32(rnd(C, seed=36), d)
    scale = f32(torch.tensor(np.random.default_rng(3).choice([0.0, 1.25], N)), d) if scaled else None
    zz = z64.float().double().reshape(-1, C)
    mean, var = zz.mean(0), zz.var(0, unbiased=False)
    mean_g, rstd_g = mean.float().to(d), torch.rsqrt(var + 1e-3).float().to(d)
    dx_plain = prev.clone() if acc else torch.empty_like(prev)
    ops.conv2d_bwd_data(dy, w, 1, out=dx_plain, accumulate=acc)
    dx_bn = prev.clone() if acc else torch.empty_like(prev)
    part = torch.full((1 << 18,), 5.0, device=d)
    _, nblk = ops.conv2d_bwd_data(dy, w, 1, out=dx_bn, accumulate=acc, bn=(z, mean_g, rstd_g, scale), part=part)
    assert torch.equal(dx_plain, dx_bn)
    assert nblk > 0 and ("conv1x1_ksplit_k" if K > 112 else "conv1x1_stream_k") in ops.conv2d_kernel_name(N, H, H, K, C, 1)
    assert (part[nblk * 2 * C:] == 5).all()
    # sums against float64
    g64 = dx_bn.double().cpu().reshape(-1, C) * (scale.double().cpu().repeat_interleave(H * H)[:, None] if scaled else 1.0)
    mean, var = mean.cpu(), var.cpu()
    xh = (zz - mean) * torch.rsqrt(var + 1e-3)
    sums = part[: nblk * 2 * C].view(nblk, 2, C).double().sum(0).cpu()
    close(sums[0], g64.sum(0), 2e-5, "stage-1 sum g")
    close(sums[1], (g64 * xh).sum(0), 2e-5, "stage-1 sum g xhat")
    # the batch norm's backward with and without the handed-over stage 1
    outs = []
    for st in (None, (part, nblk)):
        dxo = torch.empty_like(dx_bn)
        dg, db = torch.empty(C, device=d), torch.empty(C, device=d)
        ops.bn_bwd(z, dx_bn, mean_g, rstd_g, gamma, beta, False, False, scale, None, None, dx=dxo, dgamma=dg, dbeta=db, stage1=st)
        outs.append((dxo, dg, db))
    for a, b, what in zip(outs[0], outs[1], ("dx", "dgamma", "dbeta")):
        close(b, a.double().cpu(), 2e-5, "bn backward with stage 1 from the GEMM: " + what)


@pytest.mark.parametrize("N,H,Co,C", [(8, 14, 80, 480), (8, 14, 112, 672), (6, 14, 112, 672), (64, 4, 40, 240), (13, 9, 24, 144), (5, 14, 80, 480)])
def test_conv2d_bwd_data_emits_gate_gradient_partials(N, H, Co, C):
    """mliis_conv2d_bwd_data_gate + mliis_se_mlp_bwd(dgate_row_groups): the project conv's backward-data (K = its few output channels: the
    streaming kernel) leaves per-16-row-group sums of dx * a1 split by image; the SE backward kernel folds the groups of each image.
    Same dx as the plain call, same dpre1 / dpre2 / chan_add as with the colsum-made gate gradient, and the partials against float64
    (maps of 196 / 16 / 81 pixels: groups that straddle two images)."""
    from mliis_amd import ops
    d = dev()
    dy = f32(rnd(N, H, H, Co, seed=41), d)
    w = f32(rnd(1, 1, C, Co, seed=42, scale=1.0 / math.sqrt(Co)), d)
    a1 = f32(rnd(N, H, H, C, seed=43), d)
    R_ = max(1, C // 24)
    gate = f32(torch.sigmoid(rnd(N, C, seed=44)), d)
    s_ = f32(rnd(N, C, seed=45), d)
    hpre = f32(rnd(N, R_, seed=46), d)
    w1, w2 = f32(rnd(1, 1, C, R_, seed=47, scale=0.2), d), f32(rnd(1, 1, R_, C, seed=48, scale=0.2), d)
    dx_plain = ops.conv2d_bwd_data(dy, w, 1)
    part = torch.full((1 << 20,), 9.0, device=d)
    dx2 = torch.empty_like(dx_plain)
    _, groups = ops.conv2d_bwd_data(dy, w, 1, out=dx2, gate=a1, part=part)
    assert torch.equal(dx_plain, dx2)
    if N * H * H < 1024:      # below the streaming kernel's range: the partials are not produced and the caller takes the colsum path
        assert groups == 0 and "conv1x1_stream_k" not in ops.conv2d_kernel_name(N, H, H, Co, C, 1)
        return
    assert groups == -(-N * H * H // 16)
    assert (part[groups * 2 * C:] == 9).all()
    ref = (dx2.double() * a1.double()).reshape(N, H * H, C).sum(1).cpu()
    p = part[: groups * 2 * C].view(groups, 2, C).double().cpu()
    got = torch.zeros(N, C, dtype=torch.float64)
    HW = H * H
    for rg in range(groups):
        n0 = (rg * 16) // HW
        got[n0] += p[rg, 0]
        if n0 + 1 < N:
            got[n0 + 1] += p[rg, 1]
        else:
            assert p[rg, 1].abs().max().item() == 0
    close(got, ref, 2e-5, "gate gradient from the row-group partials")
    dgate = ops.colsum(dx2, a1, nseg=N)
    o1 = ops.se_mlp_bwd(dgate.reshape(N, C), gate, s_, hpre, w1, w2, HW)
    o2 = ops.se_mlp_bwd(part, gate, s_, hpre, w1, w2, HW, dgate_groups=groups)
    for k in ("dpre1", "dpre2", "chan_add", "dw1", "db1", "dw2", "db2"):
        close(o2[k], o1[k].double().cpu(), 2e-5, "se_mlp_bwd with folded partials: " + k)
