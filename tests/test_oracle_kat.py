"""Known-answer tests that anchor the CPU oracle (oracle/efficientlab_ref.py) on hand-derivable TF semantics.  The numeric
graph of the reference cannot run here (TensorFlow 1.15 absent) and the reference ships no golden vectors, so these KATs +
the host-logic goldens are what pins the oracle ("parity unpinned" otherwise; see the oracle's header)."""
import math

import numpy as np
import pytest
import torch

from oracle import efficientlab_ref as R


def _naive_same_conv(x, w, stride, dil):
    """Direct loops from TF's documented SAME rule: out = ceil(in/stride); pad_total = max((out-1)*s + (k-1)*d + 1 - in, 0);
    pad_before = pad_total // 2 (extra padding goes to the bottom/right)."""
    H, W = x.shape
    k = w.shape[0]
    Ho, Wo = -(-H // stride), -(-W // stride)
    pt = max((Ho - 1) * stride + (k - 1) * dil + 1 - H, 0) // 2
    pl = max((Wo - 1) * stride + (k - 1) * dil + 1 - W, 0) // 2
    y = np.zeros((Ho, Wo))
    for i in range(Ho):
        for j in range(Wo):
            for a in range(k):
                for b in range(k):
                    ii, jj = i * stride - pt + a * dil, j * stride - pl + b * dil
                    if 0 <= ii < H and 0 <= jj < W:
                        y[i, j] += x[ii, jj] * w[a, b]
    return y


@pytest.mark.parametrize("H,W,k,s,d", [(8, 8, 3, 1, 1), (8, 8, 3, 2, 1), (7, 9, 3, 2, 1), (8, 8, 5, 2, 1), (9, 7, 5, 2, 1), (8, 8, 5, 1, 1),
                                       (10, 10, 3, 1, 2), (14, 14, 3, 1, 6), (6, 6, 3, 1, 2)])
def test_same_padding_placement(H, W, k, s, d):
    g = np.random.default_rng(0)
    x, w = g.standard_normal((H, W)), g.standard_normal((k, k))
    got = R.conv2d_same(torch.tensor(x)[None, None], torch.tensor(w)[:, :, None, None], s, d)[0, 0].numpy()
    np.testing.assert_allclose(got, _naive_same_conv(x, w, s, d), atol=1e-12)
    # corner impulses: stride-2 on an even side puts the padding at the bottom/right only (pad (0,1) for k=3)
    if (H, k, s) == (8, 3, 2):
        assert R.same_pad_amounts(8, 3, 2) == (0, 1) and R.same_pad_amounts(8, 5, 2) == (1, 2) and R.same_pad_amounts(224, 3, 2) == (0, 1)


def test_batch_norm_known_answer_and_moving_average():
    x = torch.tensor([[[[1.0]], [[3.0]]], [[[5.0]], [[7.0]]]], dtype=torch.float64).permute(0, 3, 1, 2)  # N=2,H=2,W=1,C=1 -> NCHW
    G, Bt, Z, O = (torch.tensor([v], dtype=torch.float64) for v in (2.0, 0.5, 0.0, 1.0))
    nm = {}
    y = R.batch_norm(x, G, Bt, (Z, O), True, nm, "bn", fused=False)
    mean, var = 4.0, 5.0                             # biased variance of {1,3,5,7}
    exp = (np.array([1.0, 3, 5, 7]) - mean) / math.sqrt(var + 1e-3) * 2 + 0.5
    np.testing.assert_allclose(y.permute(0, 2, 3, 1).reshape(-1).numpy(), exp, rtol=1e-12)
    mm, mv = nm["bn"]
    assert mm.item() == pytest.approx(0.99 * 0 + 0.01 * mean) and mv.item() == pytest.approx(0.99 * 1 + 0.01 * var)
    nm2 = {}
    R.batch_norm(x, G, Bt, (Z, O), True, nm2, "bn", fused=True)
    assert nm2["bn"][1].item() == pytest.approx(0.99 + 0.01 * var * 4 / 3)   # fused BN feeds the unbiased variance
    yi = R.batch_norm(x, G, Bt, (O, 4 * O), False, None, "bn", False)
    np.testing.assert_allclose(yi.reshape(-1).numpy(), (np.array([1.0, 3, 5, 7]) - 1) / math.sqrt(4 + 1e-3) * 2 + 0.5, rtol=1e-12)


def test_cross_entropy_uniform_logits_and_gradient():
    a = R.arch()
    z = torch.zeros(2, 3, 3, 2, dtype=torch.float64, requires_grad=True)
    t1 = (torch.arange(18).reshape(2, 3, 3) % 2).double()
    t = torch.stack([1 - t1, t1], -1)
    loss = R.loss_fn(a, {}, z, t)
    assert loss.item() == pytest.approx(math.log(2.0), rel=1e-12)
    (g,) = torch.autograd.grad(loss, [z])
    np.testing.assert_allclose(g.numpy(), ((0.5 - t) / 18).numpy(), atol=1e-15)
    # label smoothing 0.2: targets become 0.9/0.1 -> loss still ln 2 at uniform logits
    assert R.loss_fn(a, {}, z, t, 0.2).item() == pytest.approx(math.log(2.0))
    # dice term: -ln(2 iou / (iou + 1)) with iou = mean_n (sum p t + e)/(sum p + sum t - sum p t + e), p = 0.5
    iou = np.mean([(0.5 * s + 1e-7) / (4.5 + s - 0.5 * s + 1e-7) for s in (t1[0].sum().item(), t1[1].sum().item())])
    assert R.loss_fn(a, {}, z, t, 0.0, True).item() == pytest.approx(math.log(2.0) - math.log(2 * iou / (iou + 1)), rel=1e-12)


def test_predictions_threshold_and_tie():
    z = torch.tensor([[[[0.0, 0.0], [1.0, -1.0], [-2.0, 3.0]]]])
    p = R.predictions(z)
    assert p.tolist() == [[[[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]]]]   # tie -> neither class


def test_bilinear_align_corners():
    x = torch.arange(14, dtype=torch.float64).reshape(1, 1, 1, 14).expand(1, 1, 14, 14).contiguous()
    y = R.resize_bilinear_ac(x, (56, 56))
    np.testing.assert_allclose(y[0, 0, 0].numpy(), np.arange(56) * 13.0 / 55.0, atol=1e-12)   # src = dst*(in-1)/(out-1)
    assert y[0, 0, 0, 0] == 0 and y[0, 0, 0, 55] == 13
    assert R.resize_bilinear_ac(x, (14, 14)) is x


def test_l2_term_and_sgd_rule():
    a = R.arch(image_size=32)
    params, bn = R.init_state(a, 0)
    z = torch.zeros(1, 32, 32, 2, dtype=torch.float64)
    t = torch.stack([torch.ones(1, 32, 32), torch.zeros(1, 32, 32)], -1).double()
    l2 = 0.0005 * sum(0.5 * (v ** 2).sum().item() for k, v in params.items() if "batch_normalization" not in k)
    assert R.loss_fn(a, params, z, t, l2=True).item() == pytest.approx(math.log(2) + l2)
    n_bn = sum(1 for k in params if "batch_normalization" in k)
    assert n_bn == 2 * 39                       # only gamma/beta of the 39 BN layers are exempt from L2
    x = torch.zeros(2, 32, 32, 3, dtype=torch.float64)
    before = {k: v.clone() for k, v in params.items()}
    _, g, _ = R.inner_step(a, params, bn, x, t.expand(2, -1, -1, -1), 0.1)
    k = "decode/final_layer_weights/bias"
    np.testing.assert_allclose(params[k].numpy(), (before[k] - 0.1 * g[k]).numpy(), atol=1e-15)


def test_drop_connect_zero_scale_is_identity_block_and_init_stats():
    a = R.arch(image_size=32)
    params, bn = R.init_state(a, 1)
    x = torch.rand(2, 32, 32, 3, dtype=torch.float64) * 255
    taps0, taps1 = {}, {}
    R.forward(a, params, bn, x, True, {2: torch.tensor([0.0, 0.0])}, None, taps0)
    R.forward(a, params, bn, x, True, None, None, taps1)
    np.testing.assert_allclose(taps0["block_2"].numpy(), taps0["block_1"].numpy())       # dropped residual branch
    assert not np.allclose(taps1["block_2"].numpy(), taps1["block_1"].numpy())
    # initialisers: N(0, sqrt(2/(k*k*out))) and glorot-uniform limits
    w = params["efficientnet-b0/blocks_3/depthwise_conv2d/depthwise_kernel"]
    assert w.std().item() == pytest.approx(math.sqrt(2.0 / 25), rel=0.1)
    wf = params["decode/decode_skip_connections_1/conv2d_2/kernel"]
    lim = math.sqrt(6.0 / (9 * 360 + 9 * 112))
    assert wf.abs().max().item() <= lim and wf.abs().max().item() > 0.95 * lim
