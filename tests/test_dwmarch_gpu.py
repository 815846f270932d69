"""Parity of the row-marching depthwise kernels (csrc/dwmarch.hip: mliis_dwconv_bn_fwd / mliis_dwconv_bn_bwd) against the float64 CPU
oracle ops on the same seeded inputs: the depthwise conv with the batch norm + swish in front of it applied while the input is
staged (efficientnet_model.py:185-196,266-271; utils.py:87-134), its one-pass backward (input gradient, filter gradient, stage 1 of
the batch norm's backward), the plain forms without a batch norm, and the geometry corner cases of the march (one-step chunks, odd
step counts, narrow last bands, channel tails, maps smaller than a band).  Tolerances as in test_ops_gpu.py: forward rel 2e-5,
backward rel 1e-4 of the tensor's max-abs (fp32 kernels vs fp64 oracle)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import efficientlab_ref as R  # noqa: E402
from tests.test_ops_gpu import close, dev, f32, nchw, nhwc, rnd  # noqa: E402

EPS, MOM = 1e-3, 0.99


def _swish(u):
    return u * torch.sigmoid(u)


def _case(k, s, H, W, C, N, pre=True, seed=0):
    from mliis_amd import ops
    d = dev()
    z = rnd(N, H, W, C, seed=seed + 1, scale=1.5) + 0.3
    w = rnd(k, k, C, 1, seed=seed + 2)
    gamma = 1.0 + 0.2 * rnd(C, seed=seed + 3)
    beta = 0.1 * rnd(C, seed=seed + 4)
    cnt = N * H * W
    mean = z.mean(dim=(0, 1, 2))
    var = z.var(dim=(0, 1, 2), unbiased=False)
    rstd = 1.0 / torch.sqrt(var + EPS)
    xhat = (z - mean) * rstd
    u = xhat * gamma + beta
    a = (_swish(u) if pre else z).detach().requires_grad_(True)
    wl = w.detach().requires_grad_(True)
    y = R.conv2d_same(nchw(a), wl, s, groups=C)
    dy = rnd(*y.shape, seed=seed + 5)
    ga, gw = torch.autograd.grad(y, [a, wl], dy)
    yr = nhwc(y).detach()

    zg, wg = f32(z, d), f32(w, d)
    part1 = torch.full((1 << 20,), 7.0, device=d)
    if pre:
        # the producer's stage-1 sums, cut into 5 uneven partial blocks (+ noise that cancels) as a GEMM epilogue would leave them
        s1, s2 = z.sum(dim=(0, 1, 2)), (z * z).sum(dim=(0, 1, 2))
        cuts = torch.tensor([0.1, 0.35, 0.05, 0.3, 0.2], dtype=torch.float64)
        part0 = torch.stack([torch.stack([s1 * c_, s2 * c_]) for c_ in cuts]).float().contiguous().to(d)
        mm0, mv0 = 0.2 * rnd(C, seed=seed + 6), 1.0 + 0.1 * rnd(C, seed=seed + 7).abs()
        mg, rg = torch.zeros(C, device=d), torch.zeros(C, device=d)
        mmg, mvg = f32(mm0, d), f32(mv0, d)
        bn = (f32(gamma, d), f32(beta, d), mg, rg, mmg, mvg)
        yg, nb = ops.dwconv_bn_fwd(zg, wg, s, bn=bn, part=part0, nblk=5, stats_part=part1)
        close(mg, mean, 1e-5, "bn mean")
        close(rg, rstd, 1e-5, "bn rstd")
        close(mmg, mm0 - (mm0 - mean) * (1.0 - MOM), 1e-5, "moving mean")
        close(mvg, mv0 - (mv0 - var) * (1.0 - MOM), 1e-5, "moving variance")
        # inference form: mean / rstd given, nothing folded or updated
        yi = ops.dwconv_bn_fwd(zg, wg, s, bn=(bn[0], bn[1], f32(mean, d), f32(rstd, d), None, None))
        close(yi, yr, 2e-5, "dw-bn fwd (given statistics)")
    else:
        yg, nb = ops.dwconv_bn_fwd(zg, wg, s, stats_part=part1)
    close(yg, yr, 2e-5, "dw-bn fwd")
    assert nb > 0 and nb == ops.lib.raw("mliis_dwconv_bn_fwd_blocks")(N, H, W, C, k, s)
    sums = part1[: nb * 2 * C].view(nb, 2, C).double().sum(0).cpu()
    close(sums[0], yr.sum(dim=(0, 1, 2)), 1e-5, "fused sum")
    close(sums[1], (yr * yr).sum(dim=(0, 1, 2)), 1e-5, "fused sum of squares")
    assert (part1[nb * 2 * C:] == 7.0).all(), "statistics written past their blocks"
    # without the statistics output: same tensor
    y2 = ops.dwconv_bn_fwd(zg, wg, s, bn=(bn[0], bn[1], mg, rg, None, None) if pre else None)
    assert torch.equal(y2, yg)

    # ---- backward: one pass over (dy, z)
    dyg = f32(nhwc(dy), d)
    bnp = torch.full((1 << 20,), 7.0, device=d)
    bnb = (f32(mean, d), f32(rstd, d), f32(gamma, d), f32(beta, d)) if pre else None
    dxg, dwg, nbb = ops.dwconv_bn_bwd(dyg, zg, wg, s, bn=bnb, bn_part=bnp if pre else None)
    close(dxg, ga, 1e-4, "dw-bn bwd data")
    close(dwg, gw, 1e-4, "dw-bn bwd filter")
    assert nbb == ops.dwconv_bn_bwd_blocks(N, H, W, C, k, s)
    if pre:
        sg = torch.sigmoid(u)
        g = ga * (sg * (1.0 + u * (1.0 - sg)))
        sums = bnp[: nbb * 2 * C].view(nbb, 2, C).double().sum(0).cpu()
        close(sums[0], g.sum(dim=(0, 1, 2)), 1e-4, "bn bwd sum g")
        close(sums[1], (g * xhat).sum(dim=(0, 1, 2)), 1e-4, "bn bwd sum g xhat")
        assert (bnp[nbb * 2 * C:] == 7.0).all()
    # slabs handed to the caller (the learner's deferred fold): their sum is the filter gradient
    slabs = torch.zeros(nbb * k * k * C, device=d)
    dx2, none_, _ = ops.dwconv_bn_bwd(dyg, zg, wg, s, bn=bnb, dw_part=slabs, bn_part=bnp if pre else None)
    assert none_ is None and torch.equal(dx2, dxg)
    close(slabs.view(nbb, k, k, C).double().sum(0).cpu()[..., None], gw, 1e-4, "filter-gradient slabs")


@pytest.mark.parametrize("pre", [True, False])
@pytest.mark.parametrize("k,s,H,W,C,N", [
    (3, 1, 14, 14, 32, 2), (3, 2, 16, 16, 24, 2), (5, 1, 14, 14, 40, 2), (5, 2, 28, 28, 16, 2),
    (3, 2, 15, 17, 8, 2), (5, 2, 9, 11, 8, 3), (3, 1, 5, 3, 4, 2), (5, 1, 7, 30, 144, 1),
    (3, 1, 33, 61, 36, 1), (5, 1, 37, 35, 12, 2), (3, 2, 37, 65, 20, 1), (5, 2, 41, 63, 44, 1),   # several bands / odd step counts
    (3, 1, 1, 1, 4, 1), (5, 2, 2, 2, 4, 2), (3, 2, 1, 9, 8, 1),                                   # maps smaller than the filter
])
def test_dwmarch_shapes(k, s, H, W, C, N, pre):
    _case(k, s, H, W, C, N, pre=pre)


# the large-map depthwise layers of EfficientLab-6-3 at the BASELINE config-2 batch (N = 8, 224x224 input: blocks 0-5), the 14x14 layers
# (served by the fused small-map kernels in training, by these at other sizes) and the 384x384 / EfficientNet-B3 widths
@pytest.mark.parametrize("k,s,H,C", [(3, 1, 112, 32), (3, 2, 112, 96), (3, 1, 56, 144), (5, 2, 56, 144), (5, 1, 28, 240), (3, 2, 28, 240),
                                     (3, 1, 14, 480), (5, 1, 14, 672), (5, 2, 96, 144), (3, 1, 48, 192)])
def test_dwmarch_baseline_shapes(k, s, H, C):
    _case(k, s, H, H, C, 8 if H <= 112 else 2, pre=True)


@pytest.mark.parametrize("k,s,H,W,C,N", [(3, 1, 28, 28, 40, 3), (3, 2, 30, 34, 24, 2), (5, 2, 28, 28, 16, 2), (5, 1, 14, 20, 8, 2), (3, 1, 56, 56, 144, 8),
                                         (3, 2, 112, 112, 96, 8), (5, 2, 56, 56, 144, 8), (3, 1, 9, 7, 4, 1)])
def test_depthwise_backward_with_the_batch_norm_backward_formed_on_load(k, s, H, W, C, N):
    """mliis_mbconv_dw_bwd_march = mliis_bn_bwd (depthwise batch norm bn1, squeeze-excite vectors, stage 1 given per image) followed by
    mliis_dwconv_bn_bwd, without the tensor dz1 in between: against that composition on the device (each half is tested against the
    float64 oracle on its own: test_se_and_bn_backward_share_one_pass, test_dwmarch_shapes) -- dx, the filter gradient, bn0's stage-1
    sums, dgamma1 / dbeta1."""
    from mliis_amd import ops
    d = dev()
    Ho, Wo = -(-H // s), -(-W // s)
    z0 = f32(rnd(N, H, W, C, seed=1) * 1.5 + 0.3, d)
    z1 = f32(rnd(N, Ho, Wo, C, seed=2) * 1.2 - 0.2, d)
    da2 = f32(rnd(N, Ho, Wo, C, seed=3), d)
    w = f32(rnd(k, k, C, 1, seed=4), d)
    gate = f32(torch.sigmoid(rnd(N, C, seed=5)), d)
    ca = f32(rnd(N, C, seed=6) * 0.01, d)

    def stats(t):
        td = t.double()
        m = td.mean(dim=(0, 1, 2))
        return f32(m, d), f32(1.0 / torch.sqrt(td.var(dim=(0, 1, 2), unbiased=False) + EPS), d)
    m0, r0 = stats(z0.cpu())
    m1, r1 = stats(z1.cpu())
    g0, b0, g1, b1 = (f32(1.0 + 0.2 * rnd(C, seed=7 + i), d) for i in range(4))
    # stage 1 of bn1 per image, as mliis_se_mlp_bwd_bn leaves it
    xh = (z1.double().cpu() - m1.double().cpu()) * r1.double().cpu()
    u = xh * g1.double().cpu() + b1.double().cpu()
    sg = torch.sigmoid(u)
    gg = (da2.double().cpu() * gate.double().cpu()[:, None, None, :] + ca.double().cpu()[:, None, None, :]) * (sg * (1 + u * (1 - sg)))
    stage1 = f32(torch.stack([gg.sum(dim=(1, 2)), (gg * xh).sum(dim=(1, 2))], dim=1), d)     # [N, 2, C]
    # composition
    dz1, dga_ref, dbe_ref = ops.bn_bwd(z1, da2, m1, r1, g1, b1, post_swish=True, chan_scale=gate, chan_add=ca, stage1=(stage1, N))
    nbb = ops.dwconv_bn_bwd_blocks(N, H, W, C, k, s)
    slabs_ref, part_ref = torch.zeros(nbb * k * k * C, device=d), torch.zeros(nbb * 2 * C, device=d)
    dx_ref, _, _ = ops.dwconv_bn_bwd(dz1, z0, w, s, bn=(m0, r0, g0, b0), dw_part=slabs_ref, bn_part=part_ref)
    # fused
    slabs, part = torch.zeros_like(slabs_ref), torch.zeros_like(part_ref)
    dx = torch.empty_like(z0)
    dga, dbe = torch.empty(C, device=d), torch.empty(C, device=d)
    nb = ops.mbconv_dw_bwd_march(da2, z1, (m1, r1, g1, b1), gate, ca, stage1, dga, dbe, z0, (m0, r0, g0, b0), w, s, dx, slabs, part)
    assert nb == nbb
    close(dx, dx_ref.double().cpu(), 2e-5, "dx")
    close(slabs.view(nbb, -1).sum(0), slabs_ref.view(nbb, -1).double().sum(0).cpu(), 5e-5, "filter gradient")
    close(part.view(nbb, 2, C).sum(0), part_ref.view(nbb, 2, C).double().sum(0).cpu(), 5e-5, "bn0 stage 1")
    close(dga, dga_ref.double().cpu(), 1e-5, "dgamma1")
    close(dbe, dbe_ref.double().cpu(), 1e-5, "dbeta1")


def test_dwmarch_workgroup_target_does_not_change_results(monkeypatch):
    """The row-chunk count only changes which workgroup owns a row (MLIIS_DWM_TARGET is read once per process: checked through the
    block-count query against the default)."""
    from mliis_amd import ops
    assert ops.dwconv_bn_bwd_blocks(8, 112, 112, 32, 3, 1) > 0
    assert ops.dwconv_bn_bwd_blocks(8, 112, 112, 30, 3, 1) == 0   # C % 4 != 0: rejected
