"""bf16 STORAGE of the expanded MBConv tensors (`--precision bf16-storage`, BASELINE configs[3]; reference ops
models/efficientnet/efficientnet_model.py:175-236, utils.py:87-134 on bf16 activations): every kernel family that moves z0, z1, a1 or
their gradients, called through the C ABI with MLIIS_DT_BF16 tensors, against the SAME entry point on fp32 tensors that hold the same
(bf16-exact) values.  The arithmetic is fp32 in both, so

  * a bf16 output equals the fp32 output rounded to nearest even (one bf16 ulp of slack where a sum was formed in another order);
  * fp32 side outputs (statistics, pooled sums, filter-gradient slabs, BN-backward sums) agree to fp32 accuracy where they are formed
    from inputs, and to the rounding of the output (2^-9 per element, averaging out) where they are formed from the rounded output.

The fp32 forms themselves are oracle-tested in test_ops_gpu.py / test_dwmarch_gpu.py; the whole step against the oracle with the same
storage roundings is tests/test_step_gpu.py::test_config4_bf16_storage_*."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.test_ops_gpu import close, dev, rnd  # noqa: E402

EPS = 1e-3


def bfx(t, d):
    """bf16-exact values: (bf16 tensor, the same values as fp32), both on the device."""
    b = t.float().to(d).to(torch.bfloat16).contiguous()
    return b, b.float().contiguous()


def rb(t):
    return t.to(torch.bfloat16).float()


def ulp_close(got_bf, ref_f32, what, ulps=1.0):
    """got (bf16) against round_bf16(ref): at most `ulps` bf16 steps away (a step is 2^-8 .. 2^-7 of the value: 8 significant bits),
    relative to max(|value|, 1e-3 max|ref|).  Identical fp32 arithmetic gives 0; a sum formed in another order may flip a rounding."""
    g, r = got_bf.float().cpu().double(), ref_f32.float().cpu().double()
    assert g.shape == r.shape, (what, g.shape, r.shape)
    den = torch.maximum(r.abs(), torch.full_like(r, 1e-3 * r.abs().max().item() + 1e-30))
    err = ((g - rb(ref_f32.float()).cpu().double()).abs() / den).max().item()
    assert err <= ulps * 2.0 ** -7 + 1e-7, "{}: {:.3e} relative (> {} bf16 ulp)".format(what, err, ulps)


def totals(part, nblk, nv, C):
    return part[:nblk * nv * C].view(nblk, nv, C).double().sum(0)


# ------------------------------------------------------------------------------------------------ batch norm family
@pytest.mark.parametrize("N,H,C", [(8, 28, 240), (2, 14, 40), (3, 9, 8)])
def test_bn_apply_fused_bf16(N, H, C):
    from mliis_amd import ops
    d = dev()
    xb, xf = bfx(rnd(N, H, H, C, seed=1, scale=1.5) + 0.2, d)
    gamma, beta = (1 + 0.2 * rnd(C, seed=2)).float().to(d), (0.2 * rnd(C, seed=3)).float().to(d)
    part = torch.zeros(1 << 18, device=d)
    nblk = ops.bn_stats_partial(xf, False, part)
    outs = {}
    for tag, x in (("f", xf), ("b", xb)):
        mean, rstd = torch.zeros(C, device=d), torch.zeros(C, device=d)
        pool = torch.zeros(1 << 16, device=d)
        y, chunks = ops.bn_apply_fused(x, part, nblk, mean, rstd, gamma, beta, post_swish=True, pool_part=pool)
        outs[tag] = (y, pool[:N * chunks * C].view(N, chunks, C).sum(1), mean, rstd)
    assert outs["b"][0].dtype == torch.bfloat16
    ulp_close(outs["b"][0], outs["f"][0], "a1")
    close(outs["b"][2], outs["f"][2], 1e-7, "mean")
    close(outs["b"][3], outs["f"][3], 1e-7, "rstd")
    close(outs["b"][1], outs["b"][0].float().sum(dim=(1, 2)), 2e-5, "pooled sums are sums of the ROUNDED output")


@pytest.mark.parametrize("se", [False, True])
def test_bn_bwd_bf16_with_producer_stage1(se):
    from mliis_amd import ops
    d = dev()
    N, H, C = 4, 14, 48
    xb, xf = bfx(rnd(N, H, H, C, seed=1, scale=1.3), d)
    gb, gf = bfx(rnd(N, H, H, C, seed=2), d)
    gamma, beta = (1 + 0.2 * rnd(C, seed=3)).float().to(d), (0.2 * rnd(C, seed=4)).float().to(d)
    mean = xf.mean(dim=(0, 1, 2))
    rstd = 1.0 / torch.sqrt(xf.var(dim=(0, 1, 2), unbiased=False) + EPS)
    cs = torch.sigmoid(rnd(N, C, seed=5)).float().to(d) if se else None
    ca = (0.01 * rnd(N, C, seed=6)).float().to(d) if se else None
    # stage 1 as a producer would leave it: {sum g, sum g xhat}, g = (dy cs + ca) swish'(gamma xhat + beta), in ONE block
    xh = (xf.double() - mean.double()) * rstd.double()
    u = xh * gamma.double() + beta.double()
    sg = torch.sigmoid(u)
    g = gf.double() * (cs.double()[:, None, None, :] if se else 1.0) + (ca.double()[:, None, None, :] if se else 0.0)
    g = g * (sg * (1 + u * (1 - sg)))
    stage1 = torch.stack([g.sum(dim=(0, 1, 2)), (g * xh).sum(dim=(0, 1, 2))]).float().contiguous().view(-1)
    res = {}
    for tag, x, dy in (("f", xf, gf), ("b", xb, gb)):
        dx, dg, db = ops.bn_bwd(x, dy, mean, rstd, gamma, beta, post_swish=True, chan_scale=cs, chan_add=ca, stage1=(stage1, 1))
        res[tag] = (dx, dg, db)
    assert res["b"][0].dtype == torch.bfloat16
    ulp_close(res["b"][0], res["f"][0], "dx")
    close(res["b"][1], res["f"][1], 1e-6, "dgamma")
    close(res["b"][2], res["f"][2], 1e-6, "dbeta")


def test_se_bn_bwd_sums_bf16():
    from mliis_amd import ops
    d = dev()
    N, H, C = 8, 28, 96
    zb, zf = bfx(rnd(N, H, H, C, seed=1, scale=1.3), d)
    gb, gf = bfx(rnd(N, H, H, C, seed=2), d)
    vec = lambda s, a=0.0, b=1.0: (a + b * rnd(C, seed=s)).float().to(d)  # noqa: E731
    mean, rstd, gamma, beta = vec(3, 0, 0.1), vec(4, 1.0, 0.05).abs(), vec(5, 1.0, 0.2), vec(6, 0, 0.2)
    out = {}
    for tag, z, g in (("f", zf, gf), ("b", zb, gb)):
        part = torch.zeros(ops.se_bn_bwd_sums_floats(N, H * H, C) + 64, device=d)
        nb = ops.se_bn_bwd_sums(z, g, mean, rstd, gamma, beta, part)
        out[tag] = part[:N * nb * 5 * C].view(N, nb, 5, C).sum(1)
    close(out["b"], out["f"], 1e-6, "five sums per image")


# ------------------------------------------------------------------------------------------------ marching depthwise
@pytest.mark.parametrize("k,s,H,C,in_bf", [(3, 1, 28, 48, True), (3, 2, 28, 48, True), (5, 1, 28, 40, True), (5, 2, 28, 40, True),
                                           (3, 1, 30, 32, False), (5, 2, 17, 8, False)])
def test_dwconv_bn_fwd_bf16(k, s, H, C, in_bf):
    """(bf16 in, bf16 out) = an MBConv block with an expand conv; (fp32 in, bf16 out) = block 0 behind the stem / a block without one."""
    from mliis_amd import ops
    d = dev()
    N = 3
    zb, zf = bfx(rnd(N, H, H, C, seed=1, scale=1.3) + 0.2, d)
    w = rnd(k, k, C, 1, seed=2, scale=0.4).float().to(d)
    gamma, beta = (1 + 0.2 * rnd(C, seed=3)).float().to(d), (0.2 * rnd(C, seed=4)).float().to(d)
    part0 = torch.zeros(1 << 18, device=d)
    nblk0 = ops.bn_stats_partial(zf, False, part0)
    ho = -(-H // s)
    res = {}
    for tag in ("f", "b"):
        z = zf if (tag == "f" or not in_bf) else zb
        out = torch.empty(N, ho, ho, C, device=d, dtype=torch.float32 if tag == "f" else torch.bfloat16)
        mean, rstd = torch.zeros(C, device=d), torch.zeros(C, device=d)
        sp = torch.zeros(1 << 18, device=d)
        y, nb = ops.dwconv_bn_fwd(z, w, s, bn=(gamma, beta, mean, rstd, None, None), part=part0, nblk=nblk0, out=out, stats_part=sp)
        res[tag] = (y, totals(sp, nb, 2, C), mean, rstd)
    ulp_close(res["b"][0], res["f"][0], "z1")
    close(res["b"][2], res["f"][2], 1e-7, "bn0 mean")
    yb = res["b"][0].float().double()
    if k == 3:   # statistics of the values as stored
        close(res["b"][1][0], yb.sum(dim=(0, 1, 2)), 2e-5, "sum of the rounded z1")
        close(res["b"][1][1], (yb * yb).sum(dim=(0, 1, 2)), 2e-5, "sum of squares of the rounded z1")
    else:        # 5x5 forward: statistics of the unrounded accumulators (rounding them in registers spills: csrc/dwmarch.hip)
        close(res["b"][1], res["f"][1], 1e-6, "statistics")


@pytest.mark.parametrize("k,s,H,C,zx_bf,dybn", [(3, 1, 28, 48, True, True), (3, 2, 28, 48, True, True), (5, 2, 28, 40, True, True),
                                                 (5, 1, 28, 40, True, False), (3, 1, 28, 32, False, True), (3, 1, 14, 24, False, False)])
def test_dwconv_bn_bwd_bf16(k, s, H, C, zx_bf, dybn):
    """One-pass backward with a bf16 dy (and z1): z / dx bf16 (a block with an expand conv) or fp32 (block 0 behind the stem, a block
    without an expand conv); with and without the depthwise batch norm's backward apply formed on load (mliis_mbconv_dw_bwd_march)."""
    from mliis_amd import ops
    d = dev()
    N, ho = 3, -(-H // s)
    z0b, z0f = bfx(rnd(N, H, H, C, seed=1, scale=1.3) + 0.2, d)
    dyb, dyf = bfx(rnd(N, ho, ho, C, seed=2), d)
    z1b, z1f = bfx(rnd(N, ho, ho, C, seed=3, scale=1.2), d)
    w = rnd(k, k, C, 1, seed=4, scale=0.4).float().to(d)
    v = lambda sd, a=0.0, b=1.0: (a + b * rnd(C, seed=sd)).float().to(d)  # noqa: E731
    bn0 = (v(5, 0.2, 0.1), v(6, 1.0, 0.05).abs(), v(7, 1.0, 0.2), v(8, 0, 0.2))
    bn1 = (v(9, 0.0, 0.1), v(10, 1.0, 0.05).abs(), v(11, 1.0, 0.2), v(12, 0, 0.2))
    gate, cadd = torch.sigmoid(rnd(N, C, seed=13)).float().to(d), (0.01 * rnd(N, C, seed=14)).float().to(d)
    stage1 = (0.1 * rnd(N, 2, C, seed=15)).float().to(d)
    blocks = ops.dwconv_bn_bwd_blocks(N, H, H, C, k, s)
    res = {}
    for tag in ("f", "b"):
        dy, z1 = (dyf, z1f) if tag == "f" else (dyb, z1b)
        z0 = z0f if (tag == "f" or not zx_bf) else z0b
        dx = torch.empty(N, H, H, C, device=d, dtype=z0.dtype)
        dwp, bnp = torch.zeros(blocks * k * k * C + 64, device=d), torch.zeros(blocks * 2 * C + 64, device=d)
        if dybn:
            dg1, db1 = torch.zeros(C, device=d), torch.zeros(C, device=d)
            nb = ops.mbconv_dw_bwd_march(dy, z1, bn1, gate, cadd, stage1, dg1, db1, z0, bn0, w, s, dx, dwp, bnp)
        else:
            _, _, nb = ops.dwconv_bn_bwd(dy, z0, w, s, bn=bn0, out=dx, dw_part=dwp, bn_part=bnp)
        res[tag] = (dx, totals(dwp, nb, k * k, C), totals(bnp, nb, 2, C))
    if zx_bf:
        ulp_close(res["b"][0], res["f"][0], "dx")
        close(res["b"][2], res["f"][2], 3e-3, "bn0 stage-1 sums (formed from the rounded dx)")
    else:
        close(res["b"][0], res["f"][0], 1e-6, "dx")
        close(res["b"][2], res["f"][2], 1e-6, "bn0 stage-1 sums")
    close(res["b"][1], res["f"][1], 1e-6, "filter-gradient slabs")


# ------------------------------------------------------------------------------------------------ fused small-map kernels
@pytest.mark.parametrize("k,C,V", [(3, 480, 0), (5, 480, 0), (5, 672, 0), (3, 40, 2), (5, 72, 4)])
def test_mbconv_small_bf16(k, C, V):
    from mliis_amd import ops
    d = dev()
    N, H = 8, 14
    z0b, z0f = bfx(rnd(N, H, H, C, seed=1, scale=1.5) + 0.3, d)
    w = rnd(k, k, C, 1, seed=2, scale=0.4).float().to(d)
    v = lambda sd, a=0.0, b=1.0: (a + b * rnd(C, seed=sd)).float().to(d)  # noqa: E731
    g0, b0, g1, b1 = v(3, 1.0, 0.2), v(4, 0, 0.3), v(5, 1.0, 0.2), v(6, 0, 0.3)
    part = torch.zeros(1 << 18, device=d)
    nblk = ops.bn_stats_partial(z0f, False, part)
    gate, cadd = torch.sigmoid(rnd(N, C, seed=7)).float().to(d), (0.01 * rnd(N, C, seed=8)).float().to(d)
    da2b, da2f = bfx(rnd(N, H, H, C, seed=9), d)
    fw, bw = {}, {}
    for tag, z0, dt in (("f", z0f, torch.float32), ("b", z0b, torch.bfloat16)):
        st = [torch.zeros(C, device=d) for _ in range(4)]
        z1, a1, s = torch.empty(N, H, H, C, device=d, dtype=dt), torch.empty(N, H, H, C, device=d, dtype=dt), torch.zeros(N, C, device=d)
        ops.mbconv_dw_fwd_small(z0, part, nblk, (g0, b0, st[0], st[1], None, None), w, (g1, b1, st[2], st[3], None, None), z1, a1, s, group_width=V)
        fw[tag] = (z1, a1, s, st)
    ulp_close(fw["b"][0], fw["f"][0], "z1")
    close(fw["b"][3][0], fw["f"][3][0], 1e-7, "mean0")
    # bn1 statistics / a1 / pooled means are formed from the ROUNDED z1: against torch on the device's own z1
    z1r = fw["b"][0].float().double()
    m1 = z1r.mean(dim=(0, 1, 2))
    r1 = 1.0 / torch.sqrt(z1r.var(dim=(0, 1, 2), unbiased=False) + EPS)
    close(fw["b"][3][2], m1, 2e-5, "mean1 of the rounded z1")
    close(fw["b"][3][3], r1, 2e-5, "rstd1 of the rounded z1")
    u = (z1r - m1) * r1 * g1.double() + b1.double()
    ulp_close(fw["b"][1], (u * torch.sigmoid(u)).float(), "a1")
    close(fw["b"][2], fw["b"][1].float().mean(dim=(1, 2)), 2e-5, "pooled mean of the rounded a1")
    # backward: both runs on the SAME saved tensors (the bf16 run's z1, as fp32 values for the fp32 run)
    z1b = fw["b"][0]
    z1f = z1b.float().contiguous()
    stb = fw["b"][3]
    for tag, da2, z1, z0, dt in (("f", da2f, z1f, z0f, torch.float32), ("b", da2b, z1b, z0b, torch.bfloat16)):
        outs = [torch.zeros(C, device=d), torch.zeros(C, device=d), torch.zeros(k, k, C, 1, device=d), torch.zeros(C, device=d), torch.zeros(C, device=d),
                torch.empty(N, H, H, C, device=d, dtype=dt)]
        ops.mbconv_dw_bwd_small(da2, gate, cadd, z1, (stb[2], stb[3], g1, b1), w, z0, (stb[0], stb[1], g0, b0), *outs, group_width=V)
        bw[tag] = outs
    ulp_close(bw["b"][5], bw["f"][5], "dz0")
    for i, name in enumerate(("dgamma1", "dbeta1", "dw", "dgamma0", "dbeta0")):
        close(bw["b"][i], bw["f"][i], 1e-6, name)


# ------------------------------------------------------------------------------------------------ 1x1 convs
@pytest.mark.parametrize("H,Cin,Cout", [(112, 16, 96), (56, 24, 144), (14, 80, 480), (14, 136, 816), (7, 16, 96)])
def test_expand_conv_fwd_writes_bf16(H, Cin, Cout):
    """x fp32 (a block's input) -> z0 bf16 + the batch norm's stage-1 sums of the rounded values: the streaming kernel (K <= 112) and
    the generic tile kernel (K = 136, small maps)."""
    from mliis_amd import ops
    d = dev()
    N = 8
    x = rnd(N, H, H, Cin, seed=1).float().to(d)
    w = (rnd(1, 1, Cin, Cout, seed=2) * 0.2).float().to(d)
    yf, spf = torch.empty(N, H, H, Cout, device=d), torch.zeros(1 << 22, device=d)
    yb, spb = torch.empty(N, H, H, Cout, device=d, dtype=torch.bfloat16), torch.zeros(1 << 22, device=d)
    _, nf = ops.conv2d_fwd(x, w, out=yf, stats_part=spf, precision="bf16")
    _, nb = ops.conv2d_fwd(x, w, out=yb, stats_part=spb, precision="bf16")
    assert nb > 0
    ulp_close(yb, yf, "z0")
    v = yb.float().double()
    close(totals(spb, nb, 2, Cout)[0], v.sum(dim=(0, 1, 2)), 2e-5, "sum of the rounded z0")
    close(totals(spb, nb, 2, Cout)[1], (v * v).sum(dim=(0, 1, 2)), 2e-5, "sum of squares of the rounded z0")


@pytest.mark.parametrize("H,C,Cout,gated", [(14, 480, 80, True), (56, 144, 24, True), (112, 32, 16, True), (28, 240, 40, True), (14, 672, 112, False),
                                            (112, 96, 16, False)])
def test_project_conv_fwd_and_expand_bwd_data_read_bf16(H, C, Cout, gated):
    """A = a1 bf16 (x gate) -> z2 fp32 (+ statistics), and A = dz0 bf16 -> dx fp32 accumulated: the in-workgroup K-split kernel on the
    small maps, the generic tile kernel on the large ones."""
    from mliis_amd import ops
    d = dev()
    N = 8
    ab, af = bfx(rnd(N, H, H, C, seed=1), d)
    w = (rnd(1, 1, C, Cout, seed=2) * 0.1).float().to(d)
    gate = torch.sigmoid(rnd(N, C, seed=3)).float().to(d) if gated else None
    outs = {}
    for tag, a in (("f", af), ("b", ab)):
        y, sp = torch.empty(N, H, H, Cout, device=d), torch.zeros(1 << 22, device=d)
        _, nb = ops.conv2d_fwd(a, w, out=y, stats_part=sp, x_scale=gate, precision="bf16")
        outs[tag] = (y, totals(sp, nb, 2, Cout) if nb else None)
    close(outs["b"][0], outs["f"][0], 1e-6, "z2")
    if outs["f"][1] is not None and outs["b"][1] is not None:
        close(outs["b"][1], outs["f"][1], 1e-5, "statistics")
    # backward-data of a conv with Cin = Cout_, weights [1,1,Cout_,C]: dy = the bf16 tensor, dx fp32 accumulated
    w2 = (rnd(1, 1, Cout, C, seed=4) * 0.1).float().to(d)
    base = rnd(N, H, H, Cout, seed=5).float().to(d)
    res = {}
    for tag, a in (("f", af), ("b", ab)):
        dx = base.clone()
        ops.conv2d_bwd_data(a, w2, out=dx, accumulate=True, precision="bf16")
        res[tag] = dx
    close(res["b"], res["f"], 1e-6, "dx (accumulated)")


@pytest.mark.parametrize("H,Cout,C", [(14, 80, 480), (56, 24, 144), (112, 16, 32), (28, 40, 240)])
def test_project_conv_bwd_data_writes_bf16_and_gate_partials(H, Cout, C):
    """dy fp32 (gradient of the project conv's output) -> da2 bf16; on the small maps the launch also leaves the squeeze-excite gate's
    gradient partials, formed from the rounded da2 and the bf16 a1 beside it."""
    from mliis_amd import ops
    d = dev()
    N = 8
    dy = rnd(N, H, H, Cout, seed=1).float().to(d)
    w = (rnd(1, 1, C, Cout, seed=2) * 0.1).float().to(d)
    a1b, a1f = bfx(rnd(N, H, H, C, seed=3), d)
    use_gate = 16 <= H * H <= 256
    res = {}
    for tag, dt, a1 in (("f", torch.float32, a1f), ("b", torch.bfloat16, a1b)):
        dx = torch.empty(N, H, H, C, device=d, dtype=dt)
        part = torch.zeros(1 << 22, device=d)
        if use_gate:
            _, groups = ops.conv2d_bwd_data(dy, w, out=dx, precision="bf16", gate=a1, part=part)
            res[tag] = (dx, part[:groups * 2 * C].view(groups, 2, C).sum(dim=(0, 1)) if groups else None)
        else:
            ops.conv2d_bwd_data(dy, w, out=dx, precision="bf16")
            res[tag] = (dx, None)
    ulp_close(res["b"][0], res["f"][0], "da2")
    if res["b"][1] is not None:
        ref = (res["b"][0].float().double() * a1f.double()).sum(dim=(0, 1, 2))
        close(res["b"][1], ref, 2e-5, "gate-gradient partials = sum of (rounded da2) * a1")


def test_filter_gradients_read_bf16_operands():
    """ops.FilterBatch with X = a1 bf16 (gated) / dY fp32 (project conv) and X fp32 / dY = dz0 bf16 (expand conv): the slabs equal the
    fp32-storage launch on the same values."""
    from mliis_amd import ops
    d = dev()
    N, H = 8, 14
    a1b, a1f = bfx(rnd(N, H, H, 480, seed=1), d)
    dz2 = rnd(N, H, H, 80, seed=2).float().to(d)
    gate = torch.sigmoid(rnd(N, 480, seed=3)).float().to(d)
    xin = rnd(N, H, H, 80, seed=4).float().to(d)
    dz0b, dz0f = bfx(rnd(N, H, H, 480, seed=5), d)
    slabs = {}
    for tag, a1, dz0 in (("f", a1f, dz0f), ("b", a1b, dz0b)):
        fb = ops.FilterBatch(d)
        p1 = torch.zeros(lib_size(N, H, 480, 80), device=d)
        p2 = torch.zeros(lib_size(N, H, 80, 480), device=d)
        fb.add(a1, dz2, 1, 1, p1, x_scale=gate)
        fb.add(xin, dz0, 1, 1, p2)
        fb.launch("bf16")
        slabs[tag] = (p1, p2)
    close(slabs["b"][0], slabs["f"][0], 1e-6, "project filter-gradient slabs")
    close(slabs["b"][1], slabs["f"][1], 1e-6, "expand filter-gradient slabs")
    fb = ops.FilterBatch(d)
    fb.add(a1b, dz2, 1, 1, torch.zeros(lib_size(N, H, 480, 80), device=d), x_scale=gate)
    with pytest.raises(Exception):
        fb.launch("fp32")       # bf16 tensors need the bf16-operand instances


def lib_size(N, H, Cin, Cout):
    from mliis_amd._lib import lib
    return lib.size("mliis_conv2d_bwd_filter_workspace_floats", N, H, H, Cin, Cout, 1)
