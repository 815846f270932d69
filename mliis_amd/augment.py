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

Runs on the host (scipy rotation, a few ms per image): it throttles the GPU loop exactly as SURVEY.md 8(f)-4 predicts; an
on-device version is future work.
"""
from __future__ import annotations

import random as _py_random
from typing import List, Optional, Sequence, Tuple

import numpy as np
from scipy import ndimage

BACKGROUND = (1, 0)   # one-hot mask value painted where an operation uncovers pixels

# The reference keeps the operation order in ONE module-level list that every Augmenter instance shuffles in place
# (np_augmenters.py:134,139-141): the training and the evaluation meta-learner of a process therefore share it.  Same here.
_SHARED_ORDER = ["erase", "translate", "flip", "noise", "exposure", "rotate"]
PRISTINE_ORDER = tuple(_SHARED_ORDER)


class Augmenter:
    """`py` / `npr`: generator objects with the `random` / `numpy.random` call surface; default = the global modules (what the
    reference consumes), pass `random.Random(seed)` / `numpy.random.RandomState(seed)` for a private stream."""

    def __init__(self, py=None, npr=None, verbose: bool = True):
        self.py = py if py is not None else _py_random
        self.npr = npr if npr is not None else np.random
        # persistent and process-wide (see _SHARED_ORDER), shuffled in place on every non-trivial call (np_augmenters.py:154);
        # an augmenter with private generators gets a private order
        self.order: List[str] = _SHARED_ORDER if (py is None and npr is None) else list(PRISTINE_ORDER)
        self.default_keep_probability = 1.0 / (len(self.order) + 1)
        if verbose:
            print("Initialized image segmentation augmenter.")

    # ------------------------------------------------------------------------------------------------ operations
    def noise(self, image, mask, mean_sd: float = 5.1):
        sd = np.abs(self.npr.normal(mean_sd, 1, 1))
        field = self.npr.normal(0, sd, image.shape)
        return np.clip(image + field, 0.0, 255.0).astype(np.float32), mask.astype(np.float32)

    def exposure(self, image, mask, mean_sd: float = 12.75):
        sd = np.abs(self.npr.normal(mean_sd, 1, 1))
        offset = self.npr.normal(0, sd, 1)
        return np.clip(image + offset, 0.0, 255.0).astype(np.float32), mask.astype(np.float32)

    def erase(self, image, mask, area=(0.02, 0.10), aspect=(0.3, 1 / 0.3), grey=(0, 255)):
        rows, cols = image.shape[:2]
        a = self.npr.uniform(area[0], area[1]) * rows * cols
        r = self.npr.uniform(aspect[0], aspect[1])
        bw, bh = int(np.sqrt(a / r)), int(np.sqrt(a * r))
        y0 = self.npr.randint(0, rows)
        x0 = self.npr.randint(0, cols)
        value = self.npr.uniform(grey[0], grey[1])
        image[y0:y0 + bh, x0:x0 + bw, :] = value
        mask[y0:y0 + bh, x0:x0 + bw, :] = BACKGROUND
        return image.astype(np.float32), mask.astype(np.float32)

    @staticmethod
    def flip(image, mask):
        return np.fliplr(image).astype(np.float32), np.fliplr(mask).astype(np.float32)

    def _shift(self, a, shift: int, wrap, positive, along_rows: bool, fill: Optional[Sequence[float]]):
        """along_rows: roll axis 0 and paint a band of COLUMNS (the reference's 'lr'); else roll axis 1 and paint ROWS ('ud')."""
        a = np.roll(a, shift if positive else -shift, 0 if along_rows else 1)
        if not wrap:
            colour = fill if fill is not None else self.npr.uniform(0, 255, a.shape[2])
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

    def translate(self, image, mask, max_shift: int = 23):
        ud = self.py.getrandbits(1)
        positive = self.py.getrandbits(1)
        shift = self.npr.randint(1, max_shift + 1, 1)[0]
        wrap = self.py.getrandbits(1)
        image = self._shift(image, shift, wrap, positive, not ud, None)
        mask = self._shift(mask, shift, wrap, positive, not ud, list(BACKGROUND))
        return image.astype(np.float32), mask.astype(np.float32)

    def rotate(self, image, mask, max_angle: int = 45):
        angle = self.npr.randint(-max_angle, max_angle)
        mode = self.py.sample(["reflect", "constant", "mirror", "wrap"], 1)[0]
        noise_fill, cval = False, 0
        if mode == "constant":
            if self.py.getrandbits(1):
                cval, noise_fill = -256, True
            else:
                cval = self.npr.randint(0, 256)
        image = ndimage.rotate(image, angle=angle, reshape=False, mode=mode, cval=cval)
        if noise_fill:
            hole = image == -256
            image[hole] = self.npr.randint(0, 256, size=image.shape)[hole]
        mask = ndimage.rotate(mask, angle=angle, reshape=False, mode=mode, cval=-256, order=0)
        if mode == "constant":
            mask[mask[:, :, 0] == -256] = BACKGROUND
        return image, mask

    # ------------------------------------------------------------------------------------------------ driver
    def apply_augmentations(self, image, mask, prob_to_return_original: Optional[float] = 0.0, return_image_mask_in_list: bool = True):
        """With probability `prob_to_return_original` (None -> 1/7) the inputs come back untouched (as a tuple); otherwise a random
        number of the shuffled operations is applied to copies and [image, mask] (a list, like the reference) is returned."""
        keep = self.default_keep_probability if prob_to_return_original is None else prob_to_return_original
        if self.npr.rand() <= keep:
            return image, mask
        image, mask = image.copy(), mask.copy()
        self.py.shuffle(self.order)
        count = self.npr.randint(1, len(self.order) + 1)
        for name in self.order[:count]:
            image, mask = getattr(self, name)(image, mask)
        return [image, mask] if return_image_mask_in_list else (image, mask)
