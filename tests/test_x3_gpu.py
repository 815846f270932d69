"""fp32-equivalent dense convs on the bf16 matrix cores (csrc/conv_x3.hip: every operand value split exactly into three bf16 terms,
six term products, fp32 accumulation) against the float64 CPU oracle ops -- with the tolerances of the native fp32 kernels
(tests/test_ops_gpu.py: forward rel 2e-5, backward rel 1e-4 of the tensor's max-abs), and beside the native fp32 instruction."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import efficientlab_ref as R  # noqa: E402
from tests.test_ops_gpu import close, dev, f32, nchw, nhwc, rnd  # noqa: E402


def _case(k, dil, H, W, Cin, Cout, N, bias=True):
    d = dev()
    x = rnd(N, H, W, Cin, seed=4).requires_grad_(True)
    w = rnd(k, k, Cin, Cout, seed=5, scale=1.0 / math.sqrt(k * k * Cin)).requires_grad_(True)
    b = rnd(Cout, seed=6).requires_grad_(True)
    y = R.conv2d_same(nchw(x), w, 1, dil, bias=b if bias else None)
    dy = rnd(*y.shape, seed=7)
    gx = torch.autograd.grad(y, [x], dy)[0]
    return d, x, w, b, y, dy, gx


# the decoder convs of BASELINE config 2 at their real sizes (N = 8, 224x224 input: the three stream-K launches of RSD(2)), the 14x14
# level, column counts that are not a multiple of 16 x NT, channel counts that pad a tap (136, 112, 40), odd maps with a ragged last
# row tile, a 1x1 conv with a long reduction, a dilation-6 conv, a map large enough for whole tiles beside the stream-K parts
@pytest.mark.parametrize("k,dil,H,W,Cin,Cout,N", [
    (3, 1, 56, 56, 224, 112, 8), (3, 2, 56, 56, 136, 112, 8), (3, 1, 14, 14, 224, 112, 8), (3, 2, 14, 14, 224, 112, 8),
    (3, 1, 9, 11, 360, 112, 1), (3, 2, 14, 14, 136, 48, 2), (3, 6, 14, 14, 112, 112, 1), (1, 1, 10, 13, 672, 112, 3),
    (3, 1, 17, 15, 64, 48, 5), (3, 1, 13, 13, 40, 136, 2), (1, 1, 7, 9, 512, 32, 3), (3, 1, 96, 96, 64, 32, 16),
])
def test_x3_conv_fwd_and_bwd_data_match_the_oracle(k, dil, H, W, Cin, Cout, N):
    from mliis_amd import ops
    d, x, w, b, y, dy, gx = _case(k, dil, H, W, Cin, Cout, N)
    xg, wg, bg = f32(x, d), f32(w, d), f32(b, d)
    close(ops.conv2d_fwd_x3(xg, ops.x3_image_of(wg, "fwd"), k, Cout, bg, dil), nhwc(y), 2e-5, "x3 conv fwd")
    close(ops.conv2d_bwd_data_x3(f32(nhwc(dy), d), ops.x3_image_of(wg, "bwd"), k, Cin, dil), gx, 1e-4, "x3 conv bwd data")


def test_x3_products_are_closer_to_float64_than_the_native_fp32_instruction():
    """The dominant launch of the step (3x3, 224 -> 112 at 56x56, K = 2016) on both paths against float64.  The split product keeps every
    bit of both operands (the three dropped term products are below 2^-24 of |a b|) and the 16x16x32 instruction adds its 32 products
    in one pass; the fp32 instruction rounds its accumulator after each of its K / 4 steps.  Measured: 0.7e-6 of max-abs against
    2.0e-6.  (A bar of "2 ulp of max-abs between the two kernels" is not meaningful: the native kernel itself is ~17 ulp-scale from
    float64 at K = 2016; what can be asserted is that the split form is no further from the truth than the native one.)"""
    from mliis_amd import ops
    d, x, w, b, y, dy, gx = _case(3, 1, 56, 56, 224, 112, 8, bias=False)
    xg, wg = f32(x, d), f32(w, d)
    ref = nhwc(y).detach()
    den = ref.abs().max().item()
    e3 = (ops.conv2d_fwd_x3(xg, ops.x3_image_of(wg, "fwd"), 3, 112, None, 1).cpu().double() - ref).abs().max().item() / den
    e0 = (ops.conv2d_fwd(xg, wg, None, 1).cpu().double() - ref).abs().max().item() / den
    print("fwd, K = 2016: split product %.2e, native fp32 %.2e of max-abs from float64" % (e3, e0))
    assert e3 <= 1.1 * e0 + 1e-7 and e3 <= 3e-6
    dyg = f32(nhwc(dy), d)
    den = gx.abs().max().item()
    e3 = (ops.conv2d_bwd_data_x3(dyg, ops.x3_image_of(wg, "bwd"), 3, 224, 1).cpu().double() - gx).abs().max().item() / den
    e0 = (ops.conv2d_bwd_data(dyg, wg, 1).cpu().double() - gx).abs().max().item() / den
    print("bwd-data, K = 1008: split product %.2e, native fp32 %.2e" % (e3, e0))
    assert e3 <= 1.1 * e0 + 1e-7 and e3 <= 3e-6


def test_x3_split_is_exact_for_values_with_24_significant_bits():
    """x = hi + mid + lo exactly: a conv whose weight is a one-hot tap copies its input channel bit for bit (every product is x * 1)."""
    from mliis_amd import ops
    d = dev()
    g = torch.Generator().manual_seed(3)
    N, H, C = 2, 16, 64
    x = (torch.randn(N, H, H, C, generator=g) * torch.exp(torch.randn(N, H, H, C, generator=g) * 8)).to(d)   # wide exponent range
    w = torch.zeros(3, 3, C, C)
    for c in range(C):
        w[1, 1, c, (c * 7) % C] = 1.0   # centre tap: output channel 7c mod C = input channel c
    w = w.to(d)
    y = ops.conv2d_fwd_x3(x, ops.x3_image_of(w, "fwd"), 3, C, None, 1)
    perm = torch.tensor([(c * 7) % C for c in range(C)], device=d)
    exp = torch.empty_like(x)
    exp[..., perm] = x
    assert torch.equal(y, exp)


@pytest.mark.parametrize("swish", [False, True])
def test_x3_epilogue_statistics_border_bias_accumulate_and_views(swish):
    """The epilogue extras of mliis_conv2d_fwd on the x3 path: the next batch norm's stage-1 statistics (whole tiles and the stream-K
    fix-up), the per-border-class bias of the RSD pooled branch, accumulate, a channel-sliced input view (the fuse conv reads the
    first 2 c_out channels of the pyramid buffer) and a sliced output view."""
    from mliis_amd import ops
    d = dev()
    N, H, Cbuf, Cin, Cout = 8, 56, 360, 224, 112
    buf = rnd(N, H, H, Cbuf, seed=11)
    x = buf[..., :Cin]
    w = rnd(3, 3, Cbuf, Cout, seed=12, scale=1.0 / math.sqrt(9 * Cin))
    b = rnd(Cout, seed=13)
    bb = rnd(N, 9, Cout, seed=14)
    y = R.conv2d_same(nchw(x), w[:, :, :Cin, :], 1, 1, bias=b)
    yb = nhwc(y).clone()
    for n in range(N):
        for h in range(H):
            for wq in range(H):
                cls = (0 if h == 0 else (2 if h == H - 1 else 1)) * 3 + (0 if wq == 0 else (2 if wq == H - 1 else 1))
                if cls != 4:
                    yb[n, h, wq] += bb[n, cls]
        yb[n, 1:H - 1, 1:H - 1] += bb[n, 4]
    bufg, wg, bg, bbg = f32(buf, d), f32(w, d), f32(b, d), f32(bb, d)
    im = ops.X3Images(d)
    im.add("w", "fwd", 0, 3, Cbuf, Cout, 0, Cin)
    im.finish().pack(wg.view(-1))
    part = torch.full((1 << 17,), 7.0, device=d)
    out_buf = torch.zeros(N, H, H, Cout + 16, device=d)
    out = out_buf[..., 8:8 + Cout]
    yg, nblk = ops.conv2d_fwd_x3(bufg[..., :Cin], im.image("w", "fwd"), 3, Cout, bg, 1, out=out, stats_part=part, stats_swish=swish,
                                 border_bias=bbg)
    close(yg, yb, 2e-5, "x3 fwd + border bias")
    assert nblk == 4 * -(-N * H * H // 256)
    assert float(out_buf[..., :8].abs().max()) == 0.0 and float(out_buf[..., 8 + Cout:].abs().max()) == 0.0
    sums = part[: nblk * 2 * Cout].view(nblk, 2, Cout).double().sum(0).cpu()
    u = yb * torch.sigmoid(yb) if swish else yb
    close(sums[0], u.sum(dim=(0, 1, 2)), 1e-5, "x3 fused sum")
    close(sums[1], (u * u).sum(dim=(0, 1, 2)), 1e-5, "x3 fused sum of squares")
    # accumulate: a second call adds onto the first
    y2 = ops.conv2d_fwd_x3(bufg[..., :Cin], im.image("w", "fwd"), 3, Cout, None, 1, out=yg.clone(), accumulate=True)
    close(y2, yb + nhwc(y) - b, 2e-5, "x3 fwd accumulate")
    # backward-data over the same channel window, accumulated into a sliced gradient buffer
    dy = rnd(N, H, H, Cout, seed=15)
    xr = x.clone().requires_grad_(True)
    gx = torch.autograd.grad(R.conv2d_same(nchw(xr), w[:, :, :Cin, :], 1, 1), [xr], nchw(dy))[0]
    im2 = ops.X3Images(d)
    im2.add("w", "bwd", 0, 3, Cbuf, Cout, 0, Cin)
    im2.finish().pack(wg.view(-1))
    base = rnd(N, H, H, Cbuf, seed=16)
    dbuf = f32(base, d)
    ops.conv2d_bwd_data_x3(f32(dy, d), im2.image("w", "bwd"), 3, Cin, 1, out=dbuf[..., :Cin], accumulate=True)
    close(dbuf[..., :Cin], base[..., :Cin] + gx, 1e-4, "x3 bwd data accumulate into a view")
    assert torch.equal(dbuf[..., Cin:].cpu().double(), base[..., Cin:].float().double())


def test_x3_results_are_deterministic_and_independent_of_the_launch_history():
    from mliis_amd import ops
    d, x, w, b, y, dy, gx = _case(3, 2, 56, 56, 136, 112, 8)
    xg, im = f32(x, d), ops.x3_image_of(f32(w, d), "fwd")
    a = ops.conv2d_fwd_x3(xg, im, 3, 112, None, 2)
    for _ in range(3):
        assert torch.equal(ops.conv2d_fwd_x3(xg, im, 3, 112, None, 2), a)


def test_learner_takes_the_split_products_for_the_large_decoder_maps_only():
    """matmul_precision "fp32" (the default): the 56x56 decoder level runs conv_x3_k, the 14x14 level and every MBConv conv the native
    fp32 kernels; "fp32-native" never builds a weight image."""
    from mliis_amd.learner import Learner
    dev()
    L = Learner(image_size=224, use_graph=False)
    assert L.x3 is not None and len(L.x3.rows) == 8   # two modules x {dilated branch, fuse conv} x {fwd, bwd}
    big = torch.empty(8, 56, 56, 4)
    small = torch.empty(8, 14, 14, 4)
    k1 = L.n_rsd[0][1][0]
    assert L._x3_takes(k1, big) and not L._x3_takes(k1, small) and not L._x3_takes(L.n_blocks[3]["w_proj"], big)
    L.close()
    L0 = Learner(image_size=64, use_graph=False, matmul_precision="fp32-native")
    assert L0.x3 is None
    L0.close()


def _fold(partial, total):
    n = partial.numel() // total
    return partial[: n * total].view(n, total).double().sum(0)


@pytest.mark.parametrize("N,H,Cbuf,Cin,Cout,k,dil,x3", [(8, 56, 360, 224, 112, 3, 1, True), (8, 56, 136, 128, 112, 3, 2, True),
                                                        (8, 28, 224, 224, 112, 3, 2, None), (6, 37, 200, 200, 80, 1, 1, None),
                                                        (4, 40, 128, 128, 128, 3, 1, None), (8, 56, 136, 128, 112, 1, 1, None),
                                                        (2, 14, 224, 224, 112, 3, 2, False)])
def test_x3_filter_gradients_match_the_oracle(N, H, Cbuf, Cin, Cout, k, dil, x3):
    """FilterBatch.launch("fp32x3"): the groups of 128-channel tiles as split products (conv_filter_x3_batched_k): the decoder's two
    56x56 problems at their real sizes (the fuse conv over a channel window of the pyramid buffer, the dilated branch's 128-channel
    main part), a 14x14 problem, a ragged map with channel / column tails, a full 128 x 128 tile, a 1x1 conv -- each with a second
    problem in the same launch -- against float64 autograd at the tolerance of the native kernel (1e-4 of max-abs), and no further
    from float64 than the native kernel."""
    from mliis_amd import ops
    d = dev()
    buf = rnd(N, H, H, Cbuf, seed=21)
    x = buf[..., :Cin].clone().requires_grad_(True)
    w = rnd(k, k, Cin, Cout, seed=22, scale=1.0 / math.sqrt(k * k * Cin)).requires_grad_(True)
    dy = rnd(N, H, H, Cout, seed=23)
    gw = torch.autograd.grad(R.conv2d_same(nchw(x), w, 1, dil), [w], nchw(dy))[0]
    bufg, dyg = f32(buf, d), f32(dy, d)
    plan = (ops.C.c_int * 8)()
    ops.lib.call("mliis_conv2d_bwd_filter_plan", N, H, H, Cin, Cout, k, plan)
    takes = plan[0] == 2 and plan[1] >= 4   # a group of 128-channel tiles: split products; any other plan runs the fp32 instruction
    assert x3 is None or takes == x3, (takes, list(plan[:7]))
    n = ops.lib.size("mliis_conv2d_bwd_filter_workspace_floats", N, H, H, Cin, Cout, k)
    total = k * k * Cin * Cout
    res = {}
    for prec in ("fp32x3", "fp32"):
        pa, pb = torch.full((n,), 9.0, device=d), torch.full((n,), -9.0, device=d)
        fb = ops.FilterBatch(d)
        fb.add(bufg[..., :Cin], dyg, k, dil, pa)
        fb.add(bufg[..., :Cin], dyg * 0.5, k, dil, pb)      # a second problem in the same grid
        fb.launch(prec)
        torch.cuda.synchronize()
        ga, gb = _fold(pa, total).view(k, k, Cin, Cout).cpu(), _fold(pb, total).view(k, k, Cin, Cout).cpu()
        close(ga, gw, 1e-4, prec + " filter gradient")
        close(gb, 0.5 * gw, 1e-4, prec + " filter gradient, second problem")
        res[prec] = (ga - gw).abs().max().item() / gw.abs().max().item()
    print("filter gradient, %d pixels: split product %.2e, native fp32 %.2e of max-abs from float64" % (N * H * H, res["fp32x3"], res["fp32"]))
    print("   (plan: TMF %d, NT %d -> %s)" % (plan[0], plan[1], "conv_filter_x3_batched_k" if takes else "native instruction"))
    assert res["fp32x3"] <= 1.5 * res["fp32"] + 2e-7


def test_x3_filter_gradient_of_the_concat_sliver_takes_the_multitap_form():
    """The 8-channel tail of the 136-channel concat under a 3x3 conv: all nine taps in ONE 128-row block (72 flattened (tap, channel)
    rows), in the same launch as a plain problem of the group."""
    from mliis_amd import ops
    d = dev()
    N, H, Cbuf, Cout = 8, 56, 136, 112
    buf = rnd(N, H, H, Cbuf, seed=31)
    dy = rnd(N, H, H, Cout, seed=32)
    xt = buf[..., 128:].clone().requires_grad_(True)
    w = rnd(3, 3, 8, Cout, seed=33).requires_grad_(True)
    gw = torch.autograd.grad(R.conv2d_same(nchw(xt), w, 1, 2), [w], nchw(dy))[0]
    plan = (ops.C.c_int * 8)()
    ops.lib.call("mliis_conv2d_bwd_filter_plan", N, H, H, 8, Cout, 3, plan)
    assert plan[2] == 1 and plan[0] == 2, "multitap plan expected"
    bufg, dyg = f32(buf, d), f32(dy, d)
    n = ops.lib.size("mliis_conv2d_bwd_filter_workspace_floats", N, H, H, 8, Cout, 3)
    pa = torch.full((n,), 9.0, device=d)
    fb = ops.FilterBatch(d)
    fb.add(bufg[..., 128:], dyg, 3, 2, pa)
    fb.launch("fp32x3")
    close(_fold(pa, 9 * 8 * Cout).view(3, 3, 8, Cout).cpu(), gw, 1e-4, "x3 multitap filter gradient")


def test_x3_filter_gradient_wide_tiles_equal_the_128_channel_tiles_bit_for_bit():
    """Round 6: problems with more than 128 input channels take 256-channel workgroup tiles (dY staged once per 256 input channels;
    FilterBatch builds a second table, the slabs / pixel splits / fold do not change).  The decoder's launch -- fuse conv 224 -> 112, the
    dilated branch's 128-channel part, its 8-channel multitap sliver -- and a ragged 200-channel problem: every slab element equal to the
    128-channel form's (the reduction order over the pixels is the same)."""
    from mliis_amd import ops
    d = dev()
    N, H = 8, 56
    buf1, buf2, buf3 = f32(rnd(N, H, H, 224, seed=41), d), f32(rnd(N, H, H, 136, seed=42), d), f32(rnd(2, 19, 19, 200, seed=44), d)
    dy, dy3 = f32(rnd(N, H, H, 112, seed=43), d), f32(rnd(2, 19, 19, 112, seed=45), d)
    specs = [(buf1, dy, 3, 1), (buf2[..., :128], dy, 3, 2), (buf2[..., 128:], dy, 3, 2), (buf3, dy3, 3, 1)]
    out = {}
    for wide in (False, True):
        fb = ops.FilterBatch(d)
        fb.X3_WIDE = wide
        slabs = []
        for x, g, k, dil in specs:
            n = ops.lib.size("mliis_conv2d_bwd_filter_workspace_floats", x.shape[0], x.shape[1], x.shape[2], x.shape[3], g.shape[3], k)
            slabs.append(torch.full((n,), 7.0, device=d))
            fb.add(x, g, k, dil, slabs[-1])
        fb.launch("fp32x3")
        torch.cuda.synchronize()
        assert any(t[6] is not None for t in fb.tables) == wide
        out[wide] = slabs
    for a, b in zip(out[False], out[True]):
        assert torch.equal(a, b)


def test_concurrent_lanes_keep_the_split_product_kernels():
    """Several learners' graphs in flight (Gecko(lanes=...), --concurrent-tasks) run the default fp32 path, split-product decoder convs
    included.  (Round 5 dropped them there: kernels of one learner went wrong beside the split-product kernels of another.  Round 6:
    the fault is a packed fp32 select form on a CU shared with bf16 matrix instructions; conv_x3_k / conv_filter_x3_batched_k now occupy
    their CUs alone and the library does not contain the form -- tests/test_interference_gpu.py, tests/test_build_cpu.py,
    profiles/r06_notes.md.  That the concurrent meta-update is bit-identical to the sequential one:
    tests/test_step_gpu.py::test_concurrent_task_lanes_equal_...)  Learner.disable_split_products stays as a switch."""
    from mliis_amd.learner import Learner
    from mliis_amd.reptile import Gecko
    dev()
    L = Learner(image_size=64, use_graph=False)
    lane = Learner(image_size=64, seed=5, use_graph=False)
    Gecko(L, lanes=[lane])
    assert L.x3 is not None and lane.x3 is not None and L.x3_on
    lane.disable_split_products()
    assert lane.x3 is None and not lane.x3_on
    k1 = lane.n_rsd[0][1][0]
    assert not lane._x3_takes(k1, torch.empty(8, 56, 56, 4))
    L.close()
    lane.close()
