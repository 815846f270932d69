"""TFRecord-GZIP / tf.train.Example reader-writer (no TensorFlow) and the FSS-1000 task grouping."""
import gzip
import os
import struct

import numpy as np
import pytest

from mliis_amd import tfrecord as T


def test_crc32c_known_answers():
    # RFC 3720 test vectors for CRC-32C (Castagnoli)
    assert T.crc32c(b"") == 0x00000000
    assert T.crc32c(b"\x00" * 32) == 0x8A9136AA
    assert T.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert T.crc32c(bytes(range(32))) == 0x46DD794E
    assert T.crc32c(b"123456789") == 0xE3069283
    # TFRecord mask: ((crc >> 15) | (crc << 17)) + 0xa282ead8
    assert T.masked_crc(b"123456789") == ((((0xE3069283 >> 15) | (0xE3069283 << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _shard(path, n, size, seed):
    g = np.random.default_rng(seed)
    imgs = g.integers(0, 256, (n, size, size, 3), dtype=np.uint8)
    msks = (g.random((n, size, size)) < 0.4).astype(np.uint8) * 255
    msks[0, 0, 0] = 128  # a fractional label
    T.write_records(path, [T.make_example_bytes(i, m) for i, m in zip(imgs, msks)])
    return imgs, msks


def test_roundtrip_and_parse_semantics(tmp_path):
    p = str(tmp_path / "apple.tfrecord.gzip")
    imgs, msks = _shard(p, 7, 16, 0)
    assert open(p, "rb").read(2) == b"\x1f\x8b"
    recs = list(T.read_records(p, verify_data_crc=True))
    assert len(recs) == 7
    f = T.parse_example_bytes(recs[3])
    assert set(f) == {"image", "mask"} and f["image"] == imgs[3].tobytes() and f["mask"] == msks[3].tobytes()
    image, label = T.parse_example(recs[0], 16)
    assert image.dtype == np.float32 and image.shape == (16, 16, 3) and np.array_equal(image, imgs[0].astype(np.float32))
    assert label.shape == (16, 16, 2) and np.allclose(label.sum(-1), 1.0)
    assert label[0, 0, 1] == pytest.approx(128 / 255) and label[0, 0, 0] == pytest.approx(127 / 255)   # (255-m, m)/255
    with pytest.raises(ValueError):
        T.parse_example(recs[0], 32)      # wrong image size
    # uncompressed file is accepted too
    q = str(tmp_path / "plain.tfrecord")
    T.write_records(q, recs[:2], compress=False)
    assert len(list(T.read_records(q, verify_data_crc=True))) == 2


def test_corruption_is_detected(tmp_path):
    p = str(tmp_path / "x.tfrecord.gzip")
    _shard(p, 2, 8, 1)
    raw = bytearray(gzip.open(p, "rb").read())
    raw[20] ^= 0xFF
    bad = str(tmp_path / "bad.tfrecord")
    open(bad, "wb").write(bytes(raw))
    with pytest.raises(ValueError):
        list(T.read_records(bad, verify_data_crc=True))
    raw2 = bytearray(gzip.open(p, "rb").read())
    raw2[0] ^= 0x01          # length field
    open(bad, "wb").write(bytes(raw2))
    with pytest.raises(ValueError):
        list(T.read_records(bad))
    open(bad, "wb").write(bytes(gzip.open(p, "rb").read()[:30]))
    with pytest.raises(ValueError):
        list(T.read_records(bad))


def test_task_semantics_and_split(tmp_path, golden):
    names = ["ab_wheel", "zebra", golden["fss_test_tasks"][0], golden["fss_test_tasks"][5], "mango"]
    for i, n in enumerate(names):
        _shard(str(tmp_path / (n + ".tfrecord.gzip")), 4 + i, 8, 10 + i)
    tr, va, te, trn, van, ten = T.read_fss_1000_dataset(str(tmp_path), num_val_tasks=1, image_size=8)
    assert sorted(ten) == sorted(n + ".tfrecord.gzip" for n in names[2:4])           # canonical test classes are held out
    assert van == ["zebra.tfrecord.gzip"] and sorted(trn) == ["ab_wheel.tfrecord.gzip", "mango.tfrecord.gzip"]   # reproducible val split pops the last sorted shard
    t = [x for x in tr if x.name.startswith("mango")][0]
    assert t.batch_size == 8
    x3, y3 = t.sample(3)
    x8, _ = t.sample(8)
    assert x3.shape == (3, 8, 8, 3) and y3.shape == (3, 8, 8, 2) and np.array_equal(x3, x8[:3])       # always the FIRST n examples
    with pytest.raises(ValueError):
        t.sample(9)
    assert len(T.fss_test_task_ids()) == 240 and T.fp_k_test_task_ids() == golden["fp_k_test_tasks"]
    assert T.fss_test_task_ids() == golden["fss_test_tasks"]


def test_fp_k_shot_reader_pools_the_shards_of_a_task(tmp_path):
    """read_fp_k_shot_dataset (metaseg.py:124-179): one task per synonym set, pooling every shard whose name contains a synonym."""
    sizes = {"aeroplane": 3, "airliner": 2, "bus": 4, "motorbike": 1, "potted_plant": 2, "television": 3, "tvmonitor": 1, "zebra": 5}
    for i, (n, k) in enumerate(sizes.items()):
        _shard(str(tmp_path / (n + ".tfrecord.gzip")), k, 8, 30 + i)
    tasks, names = T.read_fp_k_shot_dataset(str(tmp_path), image_size=8)
    assert len(tasks) == len(names) == 5
    by = {frozenset(s): t for s, t in zip(T.DEFAULT_K_SHOT_SET, tasks)}
    assert by[frozenset({"airliner", "aeroplane"})].batch_size == 5 and by[frozenset({"bus"})].batch_size == 4
    assert by[frozenset({"television", "tvmonitor"})].batch_size == 4 and by[frozenset({"potted_plant", "potted plant"})].batch_size == 2
    t = by[frozenset({"airliner", "aeroplane"})]
    assert t.name in ("airliner", "aeroplane")
    x, y = t.sample(5)
    assert x.shape == (5, 8, 8, 3) and y.shape == (5, 8, 8, 2)
    with pytest.raises(ValueError):
        t.sample(6)
    with pytest.raises(ValueError):
        T.read_fp_k_shot_dataset(str(tmp_path), all_task_names=[{"unicorn"}], image_size=8)


def test_k_shot_experiment_sizes_the_resident_task():
    from mliis_amd.args import argument_parser, model_kwargs
    a = argument_parser().parse_args(["--run_k_shot_learning_curves_experiment", "--checkpoint", "c"])
    assert model_kwargs(a)["max_shots"] == 420                      # 400-shot pool + 20 held-out examples
    a = argument_parser().parse_args(["--run_k_shot_learning_curves_experiment", "--k-shot-range", "1", "5", "--k-shot-test-samples", "4", "--checkpoint", "c"])
    assert model_kwargs(a)["max_shots"] == 16 and model_kwargs(a)["matmul_precision"] == "fp32"
