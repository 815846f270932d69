"""Host-side image/mask augmentation of the inner-loop batches (`--augment`, `--aug_rate`; the reference's run.sh enables it).

Behavioural restatement of the reference's numpy augmenter (augmenters/np_augmenters.py:9-160) with the SAME random-number
consumption -- which generator (`random` vs `numpy.random`), which call, in which order -- so that a seeded run draws the same
augmentations; pinned bit-exactly by tests/golden/augment.npz, produced by importing the reference module itself
(tests/golden/make_augment_golden.py).  Images are float32 HxWx3 in 0..255, masks float32 HxWx2 one-hot (background, foreground).

The six operations and their draws (np = numpy generator, py = Python generator):
  erase      np.uniform area fraction, np.uniform aspect, np.randint top, np.randint left, np.uniform grey value  (:22-38)
  translate  py.bit axis, py.bit direction, np.randint shift 1..23, py.bit wrap-around; uncovered image band <- np.uniform colour,
             mask band <- background                                                                                  (:48-99)
  flip       none                                                                                                      (:41-44)
  noise      np.normal sd (|N(5.1, 1)|), np.normal field                                                               (:9-12)
  exposure   np.normal sd (|N(12.75, 1)|), np.normal offset (one value)                                                 (:15-18)
  rotate     np.randint angle, py.sample border mode, [constant: py.bit noise-or-grey, np.randint grey | np.randint noise field]
             scipy.ndimage.rotate (cubic for the image, nearest for the mask)                                        (:102-131)
`apply` draws np.rand (keep the original?), shuffles the PERSISTENT operation order with py.shuffle and draws how many of its
leading operations to run with np.randint (:146-160).  Quirk kept: "left/right" translation rolls the ROW axis and paints COLUMNS,
"up/down" rolls the column axis and paints rows (:48-81).

Every operation is split into a DRAW half (`plan_*`: consumes the generators, touches no pixel -- none of the reference's draws
depends on pixel values) and a pure PIXEL half (`apply_recipe`).  `apply_augmentations` = plan + apply, so single calls behave like
the reference's; the split is what lets a whole task's schedule be drawn sequentially (bit-identical generator consumption) while
the pixel work -- scipy's cubic rotation is 27 ms and the noise field 7 ms per 224x224 image, 134 images/s on one core against
2 300 images/s of the device loop -- runs on a pool of worker processes (`AugmentPool`) and overlaps the device.
"""
from __future__ import annotations

import os
import random as _py_random
from typing import List, Optional, Sequence, Tuple

import numpy as np
from scipy import ndimage

BACKGROUND = (1, 0)   # one-hot mask value painted where an operation uncovers pixels

# The reference keeps the operation order in ONE module-level list that every Augmenter instance shuffles in place
# (np_augmenters.py:134,139-141): the training and the evaluation meta-learner of a process therefore share it.  Same here.
_SHARED_ORDER = ["erase", "translate", "flip", "noise", "exposure", "rotate"]
PRISTINE_ORDER = tuple(_SHARED_ORDER)


# ---------------------------------------------------------------------------------------------------- pixel halves (pure functions)
def _shift(a, shift: int, wrap, positive, along_rows: bool, colour):
    """along_rows: roll axis 0 and paint a band of COLUMNS (the reference's 'lr'); else roll axis 1 and paint ROWS ('ud')."""
    a = np.roll(a, shift if positive else -shift, 0 if along_rows else 1)
    if not wrap:
        if along_rows:
            if positive:
                a[:, :shift] = colour
            else:
                a[:, -shift:] = colour
        else:
            if positive:
                a[-shift:, :] = colour
            else:
                a[:shift, :] = colour
    return a


def _apply_one(step, image, mask):
    op = step[0]
    if op == "noise":
        return np.clip(image + step[1], 0.0, 255.0).astype(np.float32), mask.astype(np.float32)
    if op == "exposure":
        return np.clip(image + step[1], 0.0, 255.0).astype(np.float32), mask.astype(np.float32)
    if op == "erase":
        _, y0, x0, bh, bw, value = step
        image[y0:y0 + bh, x0:x0 + bw, :] = value
        mask[y0:y0 + bh, x0:x0 + bw, :] = BACKGROUND
        return image.astype(np.float32), mask.astype(np.float32)
    if op == "flip":
        return np.fliplr(image).astype(np.float32), np.fliplr(mask).astype(np.float32)
    if op == "translate":
        _, ud, positive, shift, wrap, colour = step
        image = _shift(image, shift, wrap, positive, not ud, colour)
        mask = _shift(mask, shift, wrap, positive, not ud, list(BACKGROUND))
        return image.astype(np.float32), mask.astype(np.float32)
    if op == "rotate":
        _, angle, mode, cval, fill = step
        image = ndimage.rotate(image, angle=angle, reshape=False, mode=mode, cval=cval)
        if fill is not None:
            hole = image == -256
            image[hole] = fill[hole]
        mask = ndimage.rotate(mask, angle=angle, reshape=False, mode=mode, cval=-256, order=0)
        if mode == "constant":
            mask[mask[:, :, 0] == -256] = BACKGROUND
        return image, mask
    raise ValueError("unknown augmentation step {!r}".format(op))


def apply_recipe(recipe, image, mask, as_list: bool = True):
    """Pixel half of apply_augmentations: recipe None -> the inputs untouched (a tuple), else the steps on copies ([image, mask])."""
    if recipe is None:
        return image, mask
    image, mask = image.copy(), mask.copy()
    for step in recipe:
        image, mask = _apply_one(step, image, mask)
    return [image, mask] if as_list else (image, mask)


def _apply_job(job):
    recipe, image, mask = job
    out = apply_recipe(recipe, image, mask)
    return np.asarray(out[0], dtype=np.float32), np.asarray(out[1], dtype=np.float32)


class AugmentPool:
    """Worker processes for the pixel halves.  Forked, so create it BEFORE the process initialises the GPU (run_metasegnet.py and
    bench.py do); the workers only ever run numpy / scipy."""

    def __init__(self, workers: Optional[int] = None):
        import multiprocessing as mp
        if workers is None:
            try:
                workers = len(os.sched_getaffinity(0))
            except AttributeError:
                workers = os.cpu_count() or 1
            workers = max(1, min(32, workers - 1))
        self.workers = workers
        self._pool = mp.get_context("fork").Pool(workers)

    def map_async(self, jobs: Sequence[Tuple]):
        """jobs: (recipe, image, mask) triples -> handle whose .get() returns the (image, mask) float32 pairs in order."""
        jobs = list(jobs)
        return self._pool.map_async(_apply_job, jobs, chunksize=max(1, len(jobs) // (4 * self.workers)))

    def close(self):
        self._pool.terminate()
        self._pool.join()


class Augmenter:
    """`py` / `npr`: generator objects with the `random` / `numpy.random` call surface; default = the global modules (what the
    reference consumes), pass `random.Random(seed)` / `numpy.random.RandomState(seed)` for a private stream.
    fields=False (the on-device pixel path, csrc/augment.hip): the two per-pixel random FIELDS -- the Gaussian noise image and the noise
    fill of a constant-mode rotation -- are not drawn here; their steps carry a 64-bit seed for the device generator instead.  Every
    scalar draw (which operations, how many, their parameters) is made exactly as before."""

    def __init__(self, py=None, npr=None, verbose: bool = True, fields: bool = True):
        self.py = py if py is not None else _py_random
        self.npr = npr if npr is not None else np.random
        self.fields = fields
        # persistent and process-wide (see _SHARED_ORDER), shuffled in place on every non-trivial call (np_augmenters.py:154);
        # an augmenter with private generators gets a private order
        self.order: List[str] = _SHARED_ORDER if (py is None and npr is None) else list(PRISTINE_ORDER)
        self.default_keep_probability = 1.0 / (len(self.order) + 1)
        if verbose:
            print("Initialized image segmentation augmenter.")

    # ------------------------------------------------------------------------------------------------ draw halves
    def _seed(self):
        lo, hi = self.npr.randint(0, 2 ** 31 - 1, 2)
        return int(lo), int(hi)

    def plan_noise(self, shape, mean_sd: float = 5.1):
        sd = np.abs(self.npr.normal(mean_sd, 1, 1))
        if not self.fields:
            return ("noise", float(sd[0]), self._seed())
        return ("noise", self.npr.normal(0, sd, shape))

    def plan_exposure(self, shape, mean_sd: float = 12.75):
        sd = np.abs(self.npr.normal(mean_sd, 1, 1))
        return ("exposure", self.npr.normal(0, sd, 1))

    def plan_erase(self, shape, area=(0.02, 0.10), aspect=(0.3, 1 / 0.3), grey=(0, 255)):
        rows, cols = shape[:2]
        a = self.npr.uniform(area[0], area[1]) * rows * cols
        r = self.npr.uniform(aspect[0], aspect[1])
        bw, bh = int(np.sqrt(a / r)), int(np.sqrt(a * r))
        y0 = self.npr.randint(0, rows)
        x0 = self.npr.randint(0, cols)
        return ("erase", y0, x0, bh, bw, self.npr.uniform(grey[0], grey[1]))

    def plan_flip(self, shape):
        return ("flip",)

    def plan_translate(self, shape, max_shift: int = 23):
        ud = self.py.getrandbits(1)
        positive = self.py.getrandbits(1)
        shift = self.npr.randint(1, max_shift + 1, 1)[0]
        wrap = self.py.getrandbits(1)
        colour = None if wrap else self.npr.uniform(0, 255, shape[2])
        return ("translate", ud, positive, shift, wrap, colour)

    def plan_rotate(self, shape, max_angle: int = 45):
        angle = self.npr.randint(-max_angle, max_angle)
        mode = self.py.sample(["reflect", "constant", "mirror", "wrap"], 1)[0]
        cval, fill = 0, None
        if mode == "constant":
            if self.py.getrandbits(1):
                cval, fill = -256, (self.npr.randint(0, 256, size=shape) if self.fields else self._seed())
            else:
                cval = self.npr.randint(0, 256)
        return ("rotate", angle, mode, cval, fill)

    def plan(self, shape, prob_to_return_original: Optional[float] = 0.0):
        """Draw half of apply_augmentations for an image of `shape`: None (keep the original) or the list of steps."""
        keep = self.default_keep_probability if prob_to_return_original is None else prob_to_return_original
        if self.npr.rand() <= keep:
            return None
        self.py.shuffle(self.order)
        count = self.npr.randint(1, len(self.order) + 1)
        return [getattr(self, "plan_" + name)(shape) for name in self.order[:count]]

    # ------------------------------------------------------------------------------------------------ reference call surface
    def _single(self, name, image, mask):
        if not self.fields:
            raise ValueError("an Augmenter built with fields=False only plans; the pixels are computed on the device (encode_device_ops)")
        return _apply_one(getattr(self, "plan_" + name)(image.shape), image, mask)

    def noise(self, image, mask):
        return self._single("noise", image, mask)

    def exposure(self, image, mask):
        return self._single("exposure", image, mask)

    def erase(self, image, mask):
        return self._single("erase", image, mask)

    @staticmethod
    def flip(image, mask):
        return _apply_one(("flip",), image, mask)

    def translate(self, image, mask):
        return self._single("translate", image, mask)

    def rotate(self, image, mask):
        return self._single("rotate", image, mask)

    def apply_augmentations(self, image, mask, prob_to_return_original: Optional[float] = 0.0, return_image_mask_in_list: bool = True):
        """With probability `prob_to_return_original` (None -> 1/7) the inputs come back untouched (as a tuple); otherwise a random
        number of the shuffled operations is applied to copies and [image, mask] (a list, like the reference) is returned."""
        return apply_recipe(self.plan(image.shape, prob_to_return_original), image, mask, return_image_mask_in_list)


# ---------------------------------------------------------------------------------------------------- on-device pixel half
# 48-byte records of csrc/augment.hip (struct AugOp)
DEVICE_OP_DTYPE = np.dtype([("op", "<i4"), ("i0", "<i4"), ("i1", "<i4"), ("i2", "<i4"), ("i3", "<i4"), ("f0", "<f4"), ("f1", "<f4"), ("f2", "<f4"),
                            ("f3", "<f4"), ("seed_lo", "<u4"), ("seed_hi", "<u4"), ("src", "<i4")])
_ROTATE_MODES = {"reflect": 0, "constant": 1, "mirror": 2, "wrap": 3}


def _encode_step(step, rec):
    op = step[0]
    if op == "erase":
        _, y0, x0, bh, bw, value = step
        rec["op"], rec["i0"], rec["i1"], rec["i2"], rec["i3"], rec["f0"] = 1, y0, x0, bh, bw, value
    elif op == "translate":
        _, ud, positive, shift, wrap, colour = step
        rec["op"], rec["i0"], rec["i1"], rec["i2"], rec["i3"] = 2, int(ud), int(positive), int(shift), int(wrap)
        if colour is not None:
            rec["f0"], rec["f1"], rec["f2"] = colour
    elif op == "flip":
        rec["op"] = 3
    elif op == "noise":
        if not isinstance(step[2] if len(step) > 2 else None, tuple):
            raise ValueError("noise step carries a host field: plan with Augmenter(fields=False) for the device path")
        rec["op"], rec["f0"], rec["seed_lo"], rec["seed_hi"] = 4, step[1], step[2][0], step[2][1]
    elif op == "exposure":
        rec["op"], rec["f0"] = 5, float(np.asarray(step[1]).reshape(-1)[0])
    elif op == "rotate":
        _, angle, mode, cval, fill = step
        rec["op"], rec["i0"], rec["f0"], rec["f1"] = 6, _ROTATE_MODES[mode], float(angle), float(cval)
        if fill is not None:
            if not isinstance(fill, tuple):
                raise ValueError("rotate step carries a host noise field: plan with Augmenter(fields=False) for the device path")
            rec["i1"], rec["seed_lo"], rec["seed_hi"] = 1, fill[0], fill[1]
    else:
        raise ValueError("unknown augmentation step {!r}".format(op))


def encode_device_ops(recipes, src_idx) -> np.ndarray:
    """[stages][B] array of device records for a mini-batch: recipes[b] = None (keep the original) or the list of planned steps of
    sample b, src_idx[b] = its index among the resident shots.  Stage 0 reads the shots; samples with fewer steps pad with copies."""
    B = len(recipes)
    n = max([1] + [len(r) for r in recipes if r is not None])
    out = np.zeros((n, B), dtype=DEVICE_OP_DTYPE)
    for b, r in enumerate(recipes):
        for k, step in enumerate(r or []):
            _encode_step(step, out[k, b])
    return out
