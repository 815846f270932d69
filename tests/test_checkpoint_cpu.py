"""Adam(beta1 = 0) step count through a checkpoint (mliis_amd/checkpoint.py: adam_step_entries / adam_step_from).  TensorFlow stores
the count only as beta2_power = 0.999^(t+1) in float32 (the reference saves every global variable, train.py:54,131; the default inner
optimizer is Adam, meta_learners/args.py:151-154), which underflows to 0 after ~103k steps: the explicit `adam_step` entry must
survive both tensor containers, and a TF-style checkpoint with a saturated beta2_power must load instead of raising."""
import numpy as np
import pytest

from mliis_amd import checkpoint as ck

BETA2 = 0.999


@pytest.mark.parametrize("fmt", ["npz", "tf"])
@pytest.mark.parametrize("t", [0, 7, 295, 16000, 200000, 5000000])
def test_adam_step_round_trips_through_both_containers(tmp_path, fmt, t):
    vals = {"w": np.arange(6, dtype=np.float32).reshape(2, 3), "w/Adam_1": np.ones((2, 3), np.float32)}
    vals.update(ck.adam_step_entries(t, BETA2))
    path = ck.Saver(fmt=fmt).save(vals, str(tmp_path), global_step=3)
    back = ck.load(ck.latest_checkpoint(str(tmp_path)))
    assert path.endswith("model.ckpt-3") and set(back) == set(vals)
    assert back[ck.ADAM_STEP_KEY].dtype == np.int64 and ck.adam_step_from(back, BETA2) == t
    if t >= 200000:   # what TensorFlow alone would have kept: underflowed (or denormal) -- the explicit count is what restores t
        assert float(back["beta2_power"]) < 1e-37


def test_tensorflow_style_checkpoints_invert_beta2_power_and_saturate_instead_of_raising():
    for t in (0, 1, 58, 295, 5000, 40000):
        tf_vals = {"beta1_power": np.float32(0.0), "beta2_power": np.float32(BETA2 ** (t + 1.0))}
        assert ck.adam_step_from(tf_vals, BETA2) == t
    for b2p in (0.0, 1e-45, 1e-39, 3e-31):    # underflowed / denormal / below the invertible range
        assert ck.adam_step_from({"beta2_power": np.float32(b2p)}, BETA2) == ck.ADAM_SATURATED_STEPS
    assert ck.adam_step_from({"beta2_power": np.float32(np.nan)}, BETA2) == ck.ADAM_SATURATED_STEPS
    assert ck.adam_step_from({"w": np.zeros(1)}, BETA2) is None              # no optimizer state in the checkpoint
    # the bias correction at the saturation value is 1 to fp32, as it is for every t beyond ~16k
    assert np.float32(1.0 - BETA2 ** ck.ADAM_SATURATED_STEPS) == np.float32(1.0)
