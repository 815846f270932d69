"""mliis_amd.augment vs the reference augmenter's own outputs (tests/golden/augment.npz, made by importing
/root/reference/augmenters/np_augmenters.py): bit-exact for the same seeds, same generator consumption."""
import os
import random

import numpy as np
import pytest

from mliis_amd import augment
from mliis_amd.augment import Augmenter


@pytest.fixture(autouse=True)
def _pristine_order():
    augment._SHARED_ORDER[:] = augment.PRISTINE_ORDER   # the operation order is process-wide state (like the reference's)
    yield

HERE = os.path.dirname(os.path.abspath(__file__))
SIZE = 24


def _inputs(n, seed=123):
    g = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        img = (g.rand(SIZE, SIZE, 3) * 255).astype(np.float32)
        fg = (g.rand(SIZE, SIZE) < 0.35).astype(np.float32)
        out.append((img, np.stack([1 - fg, fg], axis=2)))
    return out


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(HERE, "golden", "augment.npz"))


def _same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and a.dtype == b.dtype, (what, a.shape, b.shape, a.dtype, b.dtype)
    assert np.array_equal(a, b), what


def test_each_operation_bit_exact(gold):
    names = ["erase", "translate", "flip", "noise", "exposure", "rotate"]
    data = _inputs(3 * len(names))
    k = 0
    for name in names:
        for s in range(3):
            random.seed(100 + s)
            np.random.seed(200 + s)
            img, msk = data[k]
            k += 1
            a, b = getattr(Augmenter(verbose=False), name)(img.copy(), msk.copy())
            _same(a, gold["op_%s_%d_image" % (name, s)], "%s image seed %d" % (name, s))
            _same(b, gold["op_%s_%d_mask" % (name, s)], "%s mask seed %d" % (name, s))


def test_driver_sequence_and_generator_consumption(gold):
    random.seed(7)
    np.random.seed(11)
    aug = Augmenter(verbose=False)
    probs = [None, 0.0, 0.5]
    for i, (img, msk) in enumerate(_inputs(36, seed=321)):
        keep_img = img.copy()
        res = aug.apply_augmentations(img, msk, probs[i % 3])
        assert isinstance(res, list) == bool(gold["seq_%02d_islist" % i]), i
        _same(res[0], gold["seq_%02d_image" % i], "call %d image" % i)
        _same(res[1], gold["seq_%02d_mask" % i], "call %d mask" % i)
        assert np.array_equal(img, keep_img), "inputs must not be modified"
    assert random.random() == float(gold["final_py_random"]) and np.random.rand() == float(gold["final_np_random"])


def test_private_generators_do_not_touch_the_global_ones():
    random.seed(1)
    np.random.seed(1)
    a0, b0 = random.random(), np.random.rand()
    random.seed(1)
    np.random.seed(1)
    aug = Augmenter(py=random.Random(5), npr=np.random.RandomState(6), verbose=False)
    img, msk = _inputs(1)[0]
    for _ in range(5):
        aug.apply_augmentations(img, msk, 0.0)
    assert (random.random(), np.random.rand()) == (a0, b0)
    aug2 = Augmenter(py=random.Random(5), npr=np.random.RandomState(6), verbose=False)
    aug3 = Augmenter(py=random.Random(5), npr=np.random.RandomState(6), verbose=False)
    r2, r3 = aug2.apply_augmentations(img, msk, 0.0), aug3.apply_augmentations(img, msk, 0.0)
    assert np.array_equal(r2[0], r3[0]) and np.array_equal(r2[1], r3[1])


@pytest.mark.parametrize("tag,repl", [("wrap", False), ("repl", True)])
def test_mini_batches_with_augmentation_match_the_reference_schedule(gold, tag, repl):
    """metaseg._mini_batches / augmented_batches with an augmenter == the reference's _mini_batches (5 samples, batches of 4 with
    wrap-around or with replacement): same batches bit for bit and the same generator states afterwards."""
    from mliis_amd import metaseg
    samples = _inputs(5, seed=555)
    random.seed(31)
    np.random.seed(32)
    got = list(metaseg._mini_batches(samples, 4, 4, replacement=repl, augmenter=Augmenter(verbose=False), aug_rate=0.5))
    assert len(got) == 4
    for bi, batch in enumerate(got):
        _same(np.stack([np.asarray(b[0], dtype=np.float32) for b in batch]), gold["mb_%s_%d_images" % (tag, bi)], "batch %d images" % bi)
        _same(np.stack([np.asarray(b[1], dtype=np.float32) for b in batch]), gold["mb_%s_%d_masks" % (tag, bi)], "batch %d masks" % bi)
    assert random.random() == float(gold["mb_%s_final_py" % tag]) and np.random.rand() == float(gold["mb_%s_final_np" % tag])
    # the array form used by the device learner draws identically (from the same starting state of the shared operation order)
    augment._SHARED_ORDER[:] = augment.PRISTINE_ORDER
    random.seed(31)
    np.random.seed(32)
    x, y = np.stack([s[0] for s in samples]), np.stack([s[1] for s in samples])
    for bi, (xb, yb) in enumerate(metaseg.augmented_batches(x, y, 4, 4, repl, Augmenter(verbose=False), 0.5)):
        _same(xb, gold["mb_%s_%d_images" % (tag, bi)], "array form batch %d" % bi)
        _same(yb, gold["mb_%s_%d_masks" % (tag, bi)], "array form batch %d masks" % bi)


def test_pooled_schedule_is_bit_identical_to_the_inline_one():
    """AugmentedSchedule draws first and defers the pixels: on an AugmentPool (forked workers) the batches -- Reptile and FOMAML
    flavours -- equal the inline ones bit for bit, and the lazy generator (early-stopping loops) produces the same batches too."""
    import random
    from mliis_amd import augment, metaseg
    x, y = metaseg.synthetic_task(6, 48, seed=3, block=4)
    pool = augment.AugmentPool(2)
    try:
        for fomaml in (False, True):
            outs = []
            for mode in ("inline", "pool", "lazy"):
                augment._SHARED_ORDER[:] = augment.PRISTINE_ORDER
                random.seed(11)
                np.random.seed(12)
                A = augment.Augmenter(verbose=False)
                if mode == "lazy":
                    if fomaml:
                        continue
                    outs.append(list(metaseg.lazy_augmented_batches(x, y, 4, 5, False, A, 0.5)))
                else:
                    outs.append(list(metaseg.augmented_batches(x, y, 4, 5, False, A, 0.5, tail_shots=2 if fomaml else None, fomaml=fomaml,
                                                               pool=pool if mode == "pool" else None)))
                tail_state = (random.getstate(), np.random.get_state()[1][:8].tolist())
                outs[-1] = (outs[-1], tail_state)
            ref = outs[0]
            for other in outs[1:]:
                assert len(other[0]) == len(ref[0]) == 5
                for (ax, ay), (bx, by) in zip(ref[0], other[0]):
                    assert ax.dtype == bx.dtype == np.float32 and np.array_equal(ax, bx) and np.array_equal(ay, by)
                assert other[1] == ref[1]                       # the generators end in the same state
            assert any(not np.array_equal(b[0][i], x[j]) for b in ref[0][:4] for i in range(len(b[0])) for j in range(6))   # something was augmented
    finally:
        pool.close()


def test_device_path_draws_and_record_encoding():
    """The on-device pixel path (csrc/augment.hip) keeps every scalar draw of the host augmenter: Augmenter(fields=False) plans are the
    host plans up to the first step that carries a per-pixel field (which the device generates from a drawn seed instead), and the
    48-byte records the kernel reads encode the steps faithfully."""
    import random

    from mliis_amd import augment as A
    n_checked = 0
    for seed in range(60):
        a = A.Augmenter(py=random.Random(seed), npr=np.random.RandomState(seed), verbose=False)
        d = A.Augmenter(py=random.Random(seed), npr=np.random.RandomState(seed), verbose=False, fields=False)
        pa, pd = a.plan((32, 32, 3), 0.3), d.plan((32, 32, 3), 0.3)
        assert (pa is None) == (pd is None)
        if pa is None:
            continue
        assert [s[0] for s in pa] == [s[0] for s in pd]             # same operations, same order, same count
        for sa, sd in zip(pa, pd):
            if sa[0] == "noise":
                assert isinstance(sd[1], float) and isinstance(sd[2], tuple)
                break
            if sa[0] == "rotate":
                assert sa[1:4] == sd[1:4]
                if sa[4] is not None:
                    assert isinstance(sd[4], tuple)
                    break
            elif sa[0] == "translate":
                assert sa[1:5] == sd[1:5] and (sa[5] is None or np.array_equal(sa[5], sd[5]))
            else:
                assert all(np.array_equal(u, v) for u, v in zip(sa[1:], sd[1:]))
            n_checked += 1
    assert n_checked > 40
    with pytest.raises(ValueError):
        A.Augmenter(verbose=False, fields=False).noise(np.zeros((4, 4, 3), np.float32), np.zeros((4, 4, 2), np.float32))
    rec = A.encode_device_ops([None, [("erase", 1, 2, 3, 4, 9.5), ("rotate", -13, "wrap", 0, None)],
                               [("translate", 1, 0, 7, 0, np.array([1.0, 2.0, 3.0])), ("rotate", 30, "constant", -256, (11, 12)), ("exposure", np.array([4.5]))]],
                              [5, 6, 7])
    assert rec.shape == (3, 3) and rec.dtype.itemsize == 48
    assert rec[0, 0]["op"] == 0 and rec[1, 0]["op"] == 0 and rec[2, 1]["op"] == 0            # padding = copies
    e, r = rec[0, 1], rec[1, 1]
    assert (e["op"], e["i0"], e["i1"], e["i2"], e["i3"], e["f0"]) == (1, 1, 2, 3, 4, 9.5)
    assert (r["op"], r["i0"], r["i1"], r["f0"], r["f1"]) == (6, 3, 0, -13.0, 0.0)
    t, r2, x = rec[0, 2], rec[1, 2], rec[2, 2]
    assert (t["op"], t["i0"], t["i1"], t["i2"], t["i3"], t["f0"], t["f1"], t["f2"]) == (2, 1, 0, 7, 0, 1.0, 2.0, 3.0)
    assert (r2["op"], r2["i0"], r2["i1"], r2["f1"], r2["seed_lo"], r2["seed_hi"]) == (6, 1, 1, -256.0, 11, 12)
    assert (x["op"], x["f0"]) == (5, 4.5)
    with pytest.raises(ValueError):
        A.encode_device_ops([[("noise", np.zeros((2, 2, 3)))]], [0])          # a host field cannot go to the device
