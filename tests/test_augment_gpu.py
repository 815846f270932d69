"""On-device pixel half of the inner-loop augmentation (csrc/augment.hip via Learner.augment_batch) against the host augmenter
(mliis_amd/augment.py, itself pinned bit-exactly to the reference's module by tests/golden/augment.npz): exact for the deterministic
operations (eraser, translate incl. the reference's axis quirk, flip, exposure, mask rotation in all four scipy boundary modes,
multi-stage recipes), statistically for the two per-pixel noise fields (device Philox instead of numpy's Mersenne Twister), and to
< 1 grey level on a smooth image for the cubic image rotation (Keys kernel vs scipy's prefiltered B-spline)."""
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

H = 64


def _learner():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mliis_amd.learner import Learner
    return Learner(image_size=H, seed=0, use_graph=True, drop_connect=False, augment_batch_capacity=16)


def _images(n, smooth=False, seed=0):
    g = np.random.default_rng(seed)
    if smooth:
        yy, xx = np.mgrid[0:H, 0:H]
        x = np.stack([(128 + 100 * np.sin(yy / (5.0 + i)) * np.cos(xx / (7.0 + i)))[..., None].repeat(3, 2) for i in range(n)]).astype(np.float32)
    else:
        x = g.integers(0, 256, (n, H, H, 3)).astype(np.float32)
    m = (g.random((n, H // 8, H // 8)) < 0.4).astype(np.float32).repeat(8, 1).repeat(8, 2)
    return x, np.stack([1 - m, m], -1).astype(np.float32)


def _run(L, x, y, src, recipes):
    L.load_task(x, y)
    idx = L.augment_batch(src, recipes)
    L.synchronize()
    sel = torch.tensor(idx, device=L.device)
    return L.shots_x[sel].cpu().numpy(), L.shots_y[sel].cpu().numpy()


def test_deterministic_operations_equal_the_host_augmenter():
    from mliis_amd import augment as A
    L = _learner()
    x, y = _images(5)
    recipes, src = [], []
    recipes.append(None); src.append(3)                                                       # the original
    recipes.append([("flip",)]); src.append(0)
    recipes.append([("erase", 10, 50, 30, 40, 77.5)]); src.append(1)                          # box clipped at the right edge
    recipes.append([("exposure", np.array([-20.25]))]); src.append(2)
    for ud in (0, 1):
        for positive in (0, 1):
            for wrap in (0, 1):
                recipes.append([("translate", ud, positive, 7 + ud + 2 * positive, wrap, None if wrap else np.array([10.0, 200.0, 99.5]))])
                src.append(4)
    recipes.append([("flip",), ("erase", 0, 0, 5, 64, 3.0), ("translate", 1, 0, 11, 0, np.array([1.0, 2.0, 3.0]))]); src.append(0)   # 3 stages
    recipes.append([("exposure", np.array([300.0])), ("flip",)]); src.append(1)                # 2 stages, clipping
    recipes.append([("flip",)] * 6); src.append(2)                                             # 6 stages
    gx, gy = _run(L, x, y, src, recipes)
    for b, (r, s) in enumerate(zip(recipes, src)):
        ex, ey = A.apply_recipe(r, x[s], y[s], as_list=False)
        np.testing.assert_array_equal(gx[b], np.asarray(ex, dtype=np.float32), err_msg="image of sample %d" % b)
        np.testing.assert_array_equal(gy[b], np.asarray(ey, dtype=np.float32), err_msg="mask of sample %d" % b)
    # the resident shots are untouched and an inner step runs on the augmented slots
    np.testing.assert_array_equal(L.shots_x[:5].cpu().numpy(), x)
    idx = L.augment_batch(src[:8], recipes[:8])
    L.inner_step(idx)
    assert np.isfinite(L.loss_value())
    with pytest.raises(ValueError):
        L.inner_step([L.max_shots + 12])          # beyond the batch that was just augmented
    L.close()


@pytest.mark.parametrize("mode", ["reflect", "constant", "mirror", "wrap"])
def test_rotation_matches_scipy(mode):
    from scipy import ndimage
    L = _learner()
    x, y = _images(4, smooth=True)
    angles = [-45, -17, 8, 44]
    recipes = [[("rotate", a, mode, -256 if mode == "constant" else 0, None)] for a in angles]
    gx, gy = _run(L, x, y, [0, 1, 2, 3], recipes)
    for b, a in enumerate(angles):
        rm = ndimage.rotate(y[b], angle=a, reshape=False, mode=mode, cval=-256, order=0)
        if mode == "constant":
            rm[rm[:, :, 0] == -256] = (1, 0)
        # nearest neighbour: the same source pixel, except where the source coordinate is a tie (x.5 exactly -- rotations by 45 degrees
        # on this grid) that fp32 and scipy's float64 round to different sides: at most a handful of the 4096 pixels
        assert (gy[b] != rm).any(-1).mean() <= (5e-3 if abs(a) == 45 or mode == "wrap" else 0.0), (mode, a, (gy[b] != rm).any(-1).sum())
        ri = ndimage.rotate(x[b], angle=a, reshape=False, mode=mode, cval=-256 if mode == "constant" else 0)
        hole_r, hole_g = ri == -256, gx[b] == -256
        assert (hole_r != hole_g).mean() <= 1e-3
        ok = ~hole_r & ~hole_g
        d = np.abs(ri - gx[b])[ok]
        assert d.mean() < (0.3 if mode != "wrap" else 1.5), (mode, a, d.mean())       # Keys cubic vs prefiltered B-spline, smooth image
        assert np.quantile(d, 0.99) < (2.0 if mode != "wrap" else 40.0)               # (wrap: the seam, where the spline prefilter differs)
    L.close()


def test_rotation_deviation_from_scipy_on_images_with_edges_and_noise():
    """The device rotation interpolates with the Keys cubic (a = -0.5) on the raw samples; the reference's scipy.ndimage.rotate
    (augmenters/np_augmenters.py:104-129) prefilters to cubic B-spline coefficients first.  On smooth images the two agree to a fraction
    of a grey level (test above); here the deviation is QUANTIFIED where they differ most: a piecewise-constant image with sharp edges
    (photograph-like structure) and white noise (every pixel independent: the worst case for any interpolator pair).  The bounds are the
    measured values with ~1.5x headroom; parity runs use --augment-on-host, which calls scipy itself."""
    from scipy import ndimage
    L = _learner()
    g = np.random.default_rng(5)
    edges = (g.integers(0, 256, (2, H // 16, H // 16, 3)).astype(np.float32)).repeat(16, 1).repeat(16, 2)     # 16 x 16 constant patches
    noise = g.integers(0, 256, (2, H, H, 3)).astype(np.float32)
    x = np.concatenate([edges, noise])
    _, y = _images(4)
    angles = [-31, 23, -31, 23]
    recipes = [[("rotate", a, "reflect", 0, None)] for a in angles]
    gx, _ = _run(L, x, y, [0, 1, 2, 3], recipes)
    stats = []
    for b, a in enumerate(angles):
        ri = ndimage.rotate(x[b], angle=a, reshape=False, mode="reflect")
        d = np.abs(ri - gx[b])
        stats.append((float(d.mean()), float(np.quantile(d, 0.99)), float(d.max())))
    print("rotation |device - scipy| grey levels (mean, p99, max): edges %s %s, noise %s %s" % tuple(stats))
    for mean, p99, mx in stats[:2]:        # piecewise-constant patches: differences only along the edges
        assert mean < 1.5 and p99 < 12.0, stats            # measured 0.97 / 7.2 (max 17)
    for mean, p99, mx in stats[2:]:        # white noise
        assert mean < 11.0 and p99 < 38.0, stats           # measured 7.1 / 25.5 (max 48)
    L.close()


def test_noise_fields_have_the_reference_distributions():
    L = _learner()
    x = np.full((2, H, H, 3), 128.0, dtype=np.float32)
    _, y = _images(2)
    rec = [[("noise", 5.1, (123, 456))], [("noise", 5.1, (123, 457))], [("noise", 12.0, (9, 9))],
           [("rotate", 30, "constant", -256, (77, 78))], [("noise", 5.1, (123, 456))]]
    gx, gy = _run(L, x, y, [0, 0, 1, 1, 0], rec)
    n0 = gx[0] - 128.0
    assert abs(n0.mean()) < 0.08 and abs(n0.std() - 5.1) < 0.1                      # N(0, sd) per element (np_augmenters.py:9-12)
    assert abs((gx[2] - 128.0).std() - 12.0) < 0.25
    assert np.array_equal(gx[0], gx[4]) and not np.array_equal(gx[0], gx[1])       # seeded: reproducible, and seeds differ
    ch = np.corrcoef(n0[..., 0].ravel(), n0[..., 1].ravel())[0, 1]
    assert abs(ch) < 0.05                                                          # channels independent
    from scipy import stats
    assert stats.kstest(n0.ravel()[::7] / 5.1, "norm").pvalue > 1e-3
    np.testing.assert_array_equal(gy[0], y[0])                                     # masks untouched by noise
    # constant-mode rotation with the noise fill: hole pixels are integers U{0..255}, mask background there
    inside = np.abs(gx[3] - 128.0) < 1e-3
    hole = ~inside.all(-1)
    assert 0.05 < hole.mean() < 0.4
    hv = gx[3][hole]
    assert np.array_equal(hv, np.round(hv)) and hv.min() >= 0 and hv.max() <= 255 and abs(hv.mean() - 127.5) < 6
    assert (gy[3][hole] == np.array([1.0, 0.0])).all()
    # clipping to [0, 255]
    xb = np.zeros((1, H, H, 3), dtype=np.float32)
    gx2, _ = _run(L, xb, y[:1], [0], [[("noise", 5.0, (1, 2))]])
    assert gx2.min() == 0.0 and (gx2 == 0).mean() > 0.45
    L.close()


def test_draws_are_the_reference_draws_and_the_meta_step_runs():
    """Augmenter(fields=False) makes the same scalar draws in the same order as the host augmenter (identical plans up to the first
    step that carries a per-pixel field); Gecko / FOMLIS with augment='device' train reproducibly, FOMAML's tail batch stays raw."""
    from mliis_amd import augment as A
    from mliis_amd.metaseg import DeviceTask
    from mliis_amd.reptile import FOMLIS, Gecko
    for seed in range(40):
        a = A.Augmenter(py=random.Random(seed), npr=np.random.RandomState(seed), verbose=False)
        d = A.Augmenter(py=random.Random(seed), npr=np.random.RandomState(seed), verbose=False, fields=False)
        pa, pd = a.plan((H, H, 3), 0.3), d.plan((H, H, 3), 0.3)
        assert (pa is None) == (pd is None)
        for sa, sd in zip(pa or [], pd or []):
            assert sa[0] == sd[0]
            if sa[0] == "noise":
                break
            if sa[0] == "rotate":
                assert sa[1:4] == sd[1:4]
                if sa[4] is not None:
                    break
            elif sa[0] == "translate":
                assert sa[1:5] == sd[1:5] and (sa[5] is None or np.array_equal(sa[5], sd[5]))
            else:
                assert all(np.array_equal(u, v) for u, v in zip(sa[1:], sd[1:]))
    ops = A.encode_device_ops([None, [("flip",), ("noise", 2.0, (5, 6))]], [3, 1])
    assert ops.shape == (2, 2) and ops.dtype.itemsize == 48 and ops[0, 0]["op"] == 0 and ops[1, 1]["op"] == 4 and ops[1, 1]["seed_hi"] == 6
    dev = torch.device("cuda", 0)
    tasks = []
    for i in range(2):
        x, y = _images(10, seed=30 + i)
        tasks.append(DeviceTask("t%d" % i, torch.tensor(x).to(dev), torch.tensor(y).to(dev)))
    outs = []
    for rep in range(2):
        for fomaml in (False, True):
            L = _learner()
            seen = []
            orig = L.inner_step
            L.inner_step = lambda idx, **kw: (seen.append(list(idx)), orig(idx, **kw))[1]
            kw = dict(rng_mode="per_task", seed=3, augment="device", aug_rate=0.7)
            m = FOMLIS(L, train_shots=10, tail_shots=5, **kw) if fomaml else Gecko(L, **kw)
            for _ in range(2):
                m.train_step(tasks, num_shots=10 if fomaml else 5, inner_batch_size=4, inner_iters=3, meta_step_size=0.5, meta_batch_size=2)
            L.synchronize()
            outs.append(L.export_trainable().cpu())
            assert any(i >= L.max_shots for b in seen[:2] for i in b)          # augmented batches live behind the resident shots
            if fomaml:
                assert all(i < L.max_shots for i in seen[2]) and len(seen[2]) == 5      # the tail batch is raw
            L.close()
    assert torch.equal(outs[0], outs[2]) and torch.equal(outs[1], outs[3]) and not torch.equal(outs[0], outs[1])
