"""Checkpoint directory layout of the reference:

    <dir>/checkpoint                 text; first line  model_checkpoint_path: "model.ckpt-<step>"
    <dir>/model.ckpt-<step>.npz      all global variables under their TF names (trainables + BN moving statistics)

`tf.train.Saver(max_to_keep=2).save(sess, save_dir/model.ckpt, global_step=i)` (train.py:54,129-131) and
`latest_checkpoint` (utils/util.py:42-50: regex on the first line of `checkpoint`) are mirrored.  Two tensor containers under the
same names and shapes: numpy's .npz (default) and the TensorFlow TensorBundle (`fmt="tf"`: model.ckpt-<step>.index +
.data-00000-of-00001, mliis_amd/tfbundle.py) that tf.train.Saver itself writes; `load` takes whichever is there.
"""
from __future__ import annotations

import os
import re
from typing import Dict, List, Optional

import numpy as np

CKPT_PREFIX = "model.ckpt"

# ---- Adam(beta1 = 0) step count in a checkpoint.  TF keeps it only as beta2_power = beta2^(t+1) in float32, which underflows to
#      exactly 0 after ~103k steps (0.999^t < 1.4e-45) and is already wrong by hundreds of steps in the denormal range -- a default-
#      optimizer run passes that after a few hundred meta-iterations.  Our own checkpoints therefore ALSO carry the count itself
#      (`adam_step`, int64); beta2_power stays for TensorFlow's sake.
ADAM_STEP_KEY = "adam_step"
ADAM_SATURATED_STEPS = 1000000   # beta2_power too small to invert: the bias correction 1 / (1 - beta2^t) is 1 to fp32 from ~16k steps on


def adam_step_entries(t: int, beta2: float) -> Dict[str, np.ndarray]:
    """The optimizer-state scalars saved beside the `<var>/Adam_1` slots after t applied steps."""
    return {"beta1_power": np.float32(0.0), "beta2_power": np.float32(float(beta2) ** (int(t) + 1.0)), ADAM_STEP_KEY: np.int64(int(t))}


def adam_step_from(values, beta2: float) -> Optional[int]:
    """Steps applied so far according to a checkpoint: the explicit count when present, else beta2_power inverted (TF-written
    checkpoints), else None (no Adam state stored).  A beta2_power that has underflowed (or is not a valid power) means "many"."""
    import math
    if ADAM_STEP_KEY in values:
        return max(int(np.asarray(values[ADAM_STEP_KEY]).reshape(-1)[0]), 0)
    if "beta2_power" not in values:
        return None
    b2p = float(np.asarray(values["beta2_power"]).reshape(-1)[0])
    if not (b2p > 1e-30):                    # underflowed, in the denormal range (or NaN): too many steps to invert
        return ADAM_SATURATED_STEPS
    if b2p > 1.0:                            # not a power of beta2: a fresh optimizer
        return 0
    return max(int(round(math.log(b2p) / math.log(float(beta2)))) - 1, 0)


def latest_checkpoint(checkpoint_dir: str, ckpt_prefix: str = CKPT_PREFIX) -> str:
    with open(os.path.join(checkpoint_dir, "checkpoint")) as f:
        first = f.readline()
    m = re.findall(re.escape(ckpt_prefix + "-") + r"[0-9]+", first)
    if not m:
        raise FileNotFoundError("no {}-<step> entry in {}/checkpoint".format(ckpt_prefix, checkpoint_dir))
    return os.path.join(checkpoint_dir, m[0])


FORMATS = ("npz", "tf")


def _files_of(path: str) -> List[str]:
    return [path + ".npz", path + ".index", path + ".data-00000-of-00001"]


class Saver:
    def __init__(self, max_to_keep: int = 2, fmt: str = "npz"):
        if fmt not in FORMATS:
            raise ValueError("checkpoint format must be one of {} but is {}".format(FORMATS, fmt))
        self.max_to_keep = max_to_keep
        self.fmt = fmt
        self.kept: List[str] = []

    def save(self, values: Dict[str, np.ndarray], save_dir: str, global_step: int, prefix: str = CKPT_PREFIX) -> str:
        os.makedirs(save_dir, exist_ok=True)
        base = "{}-{}".format(prefix, global_step)
        path = os.path.join(save_dir, base)
        if self.fmt == "tf":
            from . import tfbundle
            tfbundle.write_bundle(path, values)
        else:
            np.savez(path + ".npz", **{k.replace("/", "|"): v for k, v in values.items()})
        if base in self.kept:
            self.kept.remove(base)
        self.kept.append(base)
        while self.max_to_keep and len(self.kept) > self.max_to_keep:
            old = self.kept.pop(0)
            for f in _files_of(os.path.join(save_dir, old)):
                try:
                    os.remove(f)
                except OSError:
                    pass
        with open(os.path.join(save_dir, "checkpoint"), "w") as f:
            f.write('model_checkpoint_path: "{}"\n'.format(base))
            for b in self.kept:
                f.write('all_model_checkpoint_paths: "{}"\n'.format(b))
        return path


def load(path: str) -> Dict[str, np.ndarray]:
    """path: <dir>/model.ckpt-<step> (as returned by latest_checkpoint); a TensorBundle (.index) or the .npz container."""
    from . import tfbundle
    if tfbundle.is_bundle(path):
        return tfbundle.read_bundle(path)
    with np.load(path + ".npz") as z:
        return {k.replace("|", "/"): z[k] for k in z.files}


def save_fine_tuned_checkpoint(values, save_dir: str, task_name: str, eval_sample_num: int, step: int, fmt: str = "npz") -> str:
    """<save_dir>/<task_name>/<eval_sample_num>/model.ckpt-<step>  (utils/util.py:72-81)."""
    d = os.path.join(save_dir, str(task_name), str(eval_sample_num))
    return Saver(max_to_keep=1, fmt=fmt).save(values, d, step)
