"""Generates tests/golden/augment.npz by running the REFERENCE augmenter (augmenters/np_augmenters.py, pure numpy/scipy -- importable
without TensorFlow) on seeded inputs.  Run in the build container only (reads /root/reference); the fixture holds outputs only.

    python tests/golden/make_augment_golden.py
"""
import os
import random
import sys

import numpy as np

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from augmenters import np_augmenters as ref  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SIZE = 24


def inputs(n, seed=123):
    """Deterministic (image, mask) pairs from a PRIVATE generator (the global streams are the augmenter's)."""
    g = np.random.RandomState(seed)
    out = []
    for _ in range(n):
        img = (g.rand(SIZE, SIZE, 3) * 255).astype(np.float32)
        fg = (g.rand(SIZE, SIZE) < 0.35).astype(np.float32)
        out.append((img, np.stack([1 - fg, fg], axis=2)))
    return out


def main():
    out = {}
    # 1. each operation on its own, three seeds
    ops = [("erase", ref.random_eraser), ("translate", ref.translate), ("flip", ref.fliplr), ("noise", ref.additive_gaussian_noise),
           ("exposure", ref.exposure), ("rotate", ref.rotate_img_mask)]
    data = inputs(3 * len(ops))
    k = 0
    for name, fn in ops:
        for s in range(3):
            random.seed(100 + s)
            np.random.seed(200 + s)
            img, msk = data[k]
            k += 1
            a, b = fn(img.copy(), msk.copy())
            out["op_%s_%d_image" % (name, s)] = np.asarray(a)
            out["op_%s_%d_mask" % (name, s)] = np.asarray(b)
    # 2. the driver: one persistent augmenter, 36 consecutive calls, keep-probability cycling None / 0.0 / 0.5
    random.seed(7)
    np.random.seed(11)
    aug = ref.Augmenter()
    data = inputs(36, seed=321)
    probs = [None, 0.0, 0.5]
    for i, (img, msk) in enumerate(data):
        res = aug.apply_augmentations(img, msk, probs[i % 3])
        out["seq_%02d_image" % i] = np.asarray(res[0])
        out["seq_%02d_mask" % i] = np.asarray(res[1])
        out["seq_%02d_islist" % i] = np.array(isinstance(res, list))
    # 3. the generator states afterwards (the product must have consumed exactly as much)
    out["final_py_random"] = np.array(random.random())
    out["final_np_random"] = np.array(np.random.rand())
    # 4. the batch schedule WITH augmentation: the reference's own _mini_batches (meta_learners/metaseg.py, imported under stub
    #    tensorflow modules like make_host_logic_golden.py) -- shuffles and augmenter draws interleave sample by sample
    import make_host_logic_golden as stubs
    stubs._install_stubs()
    from meta_learners import metaseg as ref_metaseg
    samples = inputs(5, seed=555)
    pristine = [ref.random_eraser, ref.translate, ref.fliplr, ref.additive_gaussian_noise, ref.exposure, ref.rotate_img_mask]
    for tag, repl in (("wrap", False), ("repl", True)):
        random.seed(31)
        np.random.seed(32)
        ref.cur_aug_funcs[:] = pristine   # the operation order is a module-level list shared by every Augmenter: start from the import state
        aug = ref.Augmenter()
        for bi, batch in enumerate(ref_metaseg._mini_batches(samples, 4, 4, replacement=repl, augmenter=aug, aug_rate=0.5)):
            out["mb_%s_%d_images" % (tag, bi)] = np.stack([np.asarray(b[0], dtype=np.float32) for b in batch])
            out["mb_%s_%d_masks" % (tag, bi)] = np.stack([np.asarray(b[1], dtype=np.float32) for b in batch])
        out["mb_%s_final_py" % tag] = np.array(random.random())
        out["mb_%s_final_np" % tag] = np.array(np.random.rand())
    np.savez_compressed(os.path.join(HERE, "augment.npz"), **out)
    print("wrote augment.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
