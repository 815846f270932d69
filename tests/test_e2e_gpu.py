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


def _write_fss_shards(data_dir, H, n_tasks=6, examples=8, seed=0):
    """FSS-1000-style shards `<task>.tfrecord.gzip` (data/fss_1000_image_to_tfrecord.py:80,99-178: one tf.train.Example per image with
    raw uint8 `image` [H,H,3] and `mask` [H,H] in {0,255}); two of the task names are on the FSS test list."""
    from mliis_amd import tfrecord
    os.makedirs(data_dir, exist_ok=True)
    rng = np.random.default_rng(seed)
    names = tfrecord.fss_test_task_ids()[:2] + ["zz_train_task_%d" % i for i in range(n_tasks - 2)]
    arrays = {}
    for name in names:
        imgs = rng.integers(0, 256, size=(examples, H, H, 3), dtype=np.uint8)
        blocks = rng.random((examples, H // 8, H // 8)) < 0.3
        masks = (np.kron(blocks, np.ones((1, 8, 8))) * 255).astype(np.uint8)
        tfrecord.write_records(os.path.join(data_dir, name + ".tfrecord.gzip"), [tfrecord.make_example_bytes(i, m) for i, m in zip(imgs, masks)])
        arrays[name] = (imgs, masks)
    return names, arrays


def test_tfrecord_shards_drive_the_hip_path(tmp_path):
    """SURVEY 8(f)-2 on the device: FSS-style TFRecord-GZIP shards -> mliis_amd.tfrecord (framing, Example protobuf, gzip, the 0..255
    image / two-channel label semantics of data/input_fn.py:28-65) -> the HIP learner.  (a) the first inner-step loss on a decoded
    task equals the float64 oracle's on arrays built straight from the source pixels (not through the reader); (b) `run_metasegnet.main
    --data-dir` trains, checkpoints and evaluates from the shards (the two tasks on the FSS test list form the meta-test split)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mliis_amd import tfrecord
    from mliis_amd.learner import Learner
    from oracle import efficientlab_ref as R
    H = 64
    data_dir = str(tmp_path / "fss")
    names, arrays = _write_fss_shards(data_dir, H)
    train, val, test, ntr, _, nte = tfrecord.read_fss_1000_dataset(data_dir, image_size=H)
    assert sorted(nte) == sorted(n + ".tfrecord.gzip" for n in names[:2]) and len(train) == 4 and not val
    task = [t for t in train if t.name.startswith("zz_train_task_0")][0]
    assert task.batch_size == 8
    x, y = task.sample(5)
    imgs, masks = arrays["zz_train_task_0"]
    x_ref = imgs[:5].astype(np.float64)
    y_ref = np.stack([255 - masks[:5].astype(np.float64), masks[:5].astype(np.float64)], axis=-1) / 255.0
    assert x.dtype == np.float32 and np.array_equal(x, x_ref.astype(np.float32)) and np.array_equal(y, y_ref.astype(np.float32))
    O = R.OracleLearner(image_size=H, seed=0, dtype=torch.float64, lr=1e-3, drop_connect=False)
    L = Learner(image_size=H, seed=9, use_graph=False, drop_connect=False)
    L.load_named({k: v.numpy() for k, v in O.params.items()}, strict=False)
    L.load_task(x, y)                                    # what the reader decoded goes to the device
    idx = [0, 1, 2, 3, 4, 0, 1, 2]
    lo = O.inner_step(torch.tensor(x_ref)[idx], torch.tensor(y_ref)[idx])
    L.inner_step(idx)
    assert abs(L.loss_value() - lo) <= 1e-4 * max(1.0, abs(lo)), (L.loss_value(), lo)
    L.close()
    d1 = str(tmp_path / "ckpt")
    out = _run(["--image_size", str(H), "--rsd", "2", "4", "--fss_1000", "--data-dir", data_dir, "--sgd", "--shots", "3", "--inner-batch", "4",
                "--inner-iters", "2", "--meta-batch", "2", "--meta-iters", "2", "--eval-interval", "0", "--eval-samples", "2", "--eval-iters", "2",
                "--eval-batch", "3", "--meta-step", "0.5", "--learning-rate", "0.005", "--skip-train-task-eval", "--checkpoint", d1])
    assert "4 training tasks, 0 val tasks, 2 test tasks." in out and "Mean IoU over all meta-test tasks:" in out
    res = json.load(open(os.path.join(d1, "meta-test_results.json")))
    assert sorted(res) == sorted(n + ".tfrecord.gzip" for n in names[:2])


def test_restore_from_a_bundle_the_product_did_not_write(tmp_path):
    """SURVEY 8(f)-3 on the device: a TensorBundle checkpoint produced by tests/independent_bundle.py (an independent writer: own
    CRC-32C, no prefix compression, multi-block index) restores into the HIP learner -- `--pretrained` evaluation from it gives exactly
    the results of the evaluation from the checkpoint the product wrote."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from mliis_amd import checkpoint as ckpt
    from tests.independent_bundle import write_bundle_independent
    d1, d2 = str(tmp_path / "a"), str(tmp_path / "b")
    _run(BASE + ["--sgd", "--meta-iters", "2", "--eval-interval", "0", "--checkpoint", d1])
    ref = json.load(open(os.path.join(d1, "meta-test_results.json")))
    latest = ckpt.latest_checkpoint(d1)
    vals = ckpt.load(latest)
    os.makedirs(d2)
    write_bundle_independent(os.path.join(d2, os.path.basename(latest)), vals, entries_per_block=7)
    with open(os.path.join(d2, "checkpoint"), "w") as f:
        f.write('model_checkpoint_path: "%s"\n' % os.path.basename(latest))
    outs = []
    for d in (d1, d2):
        if os.path.exists(os.path.join(d, "meta-test_results.json")):
            os.remove(os.path.join(d, "meta-test_results.json"))
        o = _run(BASE + ["--sgd", "--pretrained", "--checkpoint", d])
        assert "Meta-training..." not in o
        outs.append(json.load(open(os.path.join(d, "meta-test_results.json"))))
    assert outs[0] == outs[1] and set(outs[0]) == set(ref)
