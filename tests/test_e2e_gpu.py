"""The whole program on the device: `run_metasegnet.main` with the HIP learner (64x64 synthetic tasks) -- training with the on-device
augmenter, the reference's checkpoint cadence, evaluation, `meta-test_results.json`, resume (incl. the Adam slot variables), the
TensorBundle checkpoint format, evaluation-only determinism.  tests/test_e2e_cpu.py drives the same program through the CPU oracle
learner; this one proves the product path (libmliis_hip.so) runs it end to end."""
import contextlib
import io
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BASE = ["--image_size", "64", "--rsd", "2", "4", "--shots", "3", "--inner-batch", "4", "--inner-iters", "3", "--meta-batch", "2",
        "--eval-samples", "2", "--eval-iters", "2", "--eval-batch", "3", "--synthetic-tasks", "6", "--meta-step", "0.5",
        "--learning-rate", "0.005", "--skip-train-task-eval"]


def _run(argv):
    import run_metasegnet
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        run_metasegnet.main(argv)
    return buf.getvalue()


@pytest.mark.parametrize("fmt", ["npz", "tf"])
def test_main_on_the_device(tmp_path, fmt):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mliis_amd import checkpoint as ckpt
    d1, d2 = str(tmp_path / "a"), str(tmp_path / "b")
    extra = ["--checkpoint-format", fmt] if fmt == "tf" else []
    # the reference's run.sh flavour: FOMAML, Adam (default optimizer), augmentation on
    out = _run(BASE + extra + ["--foml", "--foml-tail", "2", "--train-shots", "5", "--meta-iters", "3", "--eval-interval", "2", "--augment",
                               "--aug_rate", "0.7", "--checkpoint", d1])
    assert "Mean IoU over all meta-test tasks:" in out
    latest = ckpt.latest_checkpoint(d1)
    assert latest.endswith("model.ckpt-2")
    vals = ckpt.load(latest)
    assert any(k.endswith("/Adam_1") for k in vals) and "beta2_power" in vals          # optimizer slots travel with the checkpoint
    assert all(np.isfinite(v).all() for v in vals.values())
    res = json.load(open(os.path.join(d1, "meta-test_results.json")))
    assert len(res) >= 1 and all(0.0 <= float(x) <= 1.0 for v in res.values() for x in v)
    for split in ("train", "test"):
        rows = [json.loads(l) for l in open(os.path.join(d1, split, "scalars.jsonl"))]
        assert [r["step"] for r in rows] == [0, 2]
    # resume with a zero meta-step size: trainables stay where the checkpoint left them
    out = _run(BASE + extra + ["--meta-iters", "1", "--eval-interval", "0", "--checkpoint", d2, "--continue_training_from_checkpoint", d1,
                               "--meta-step", "0.0", "--meta-step-final", "0.0"])
    assert "Continuing meta-training from checkpoint" in out
    v2 = ckpt.load(ckpt.latest_checkpoint(d2))
    key = "decode/final_layer_weights/kernel"
    np.testing.assert_array_equal(v2[key], vals[key])
    # evaluation only: seeded, deterministic, trains nothing
    outs = []
    for _ in range(2):
        os.remove(os.path.join(d1, "meta-test_results.json"))
        o = _run(BASE + extra + ["--pretrained", "--checkpoint", d1])
        assert "Meta-training..." not in o
        outs.append(json.load(open(os.path.join(d1, "meta-test_results.json"))))
    assert outs[0] == outs[1]


def test_main_with_every_optional_decoder(tmp_path):
    """`--spatial_pyramid_pooling --skip_decoding --l1 --darc1` through the command line: the ASPP, the DeepLabv3+-style decoder and
    both extra regularisers train, checkpoint (their variables under the reference's scope names) and evaluate."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mliis_amd import checkpoint as ckpt
    d1 = str(tmp_path / "a")
    out = _run(BASE + ["--sgd", "--spatial_pyramid_pooling", "--skip_decoding", "--l1", "--darc1", "--meta-iters", "2", "--eval-interval", "0",
                       "--checkpoint", d1])
    assert "Mean IoU over all meta-test tasks:" in out
    vals = ckpt.load(ckpt.latest_checkpoint(d1))
    for k in ("decode/spatial_pyramid_pooling/branch_1/conv2d/kernel", "decode/decode_skip_connections/depthwise_conv2d_1/depthwise_kernel",
              "decode/decode_skip_connections/batch_normalization_4/moving_variance", "decode/decode_skip_connections_3/conv2d_3/kernel"):
        assert k in vals and np.isfinite(vals[k]).all(), k
    assert vals["decode/decode_skip_connections_3/conv2d/kernel"].shape == (1, 1, 168, 112)     # RSD(4)'s residual 1x1 branch
