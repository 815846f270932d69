"""TensorBundle (TensorFlow checkpoint) reader/writer: known-answer bytes derived by hand from the published wire formats
(protobuf encoding rules, leveldb table_format.md, crc32c RFC 3720 vectors), round trips across block / restart boundaries, corruption
detection, and the checkpoint-directory integration.  No TF-written file exists in the reference: parity with TF itself is unpinned."""
import os
import struct

import numpy as np
import pytest

from mliis_amd import checkpoint as ckpt
from mliis_amd import tfbundle as tb
from mliis_amd.tfrecord import crc32c


def test_crc32c_known_answers():
    assert crc32c(b"123456789") == 0xE3069283                      # CRC-32C check value
    assert crc32c(bytes(32)) == 0x8A9136AA                          # RFC 3720 B.4: 32 bytes of zeros
    assert crc32c(bytes([0xFF] * 32)) == 0x62A8AB43                 # RFC 3720 B.4: 32 bytes of ones
    assert crc32c(bytes(range(32))) == 0x46DD794E                   # RFC 3720 B.4: ascending
    assert tb._mask(0) == 0xA282EAD8 and tb._mask(0xFFFFFFFF) == (0xFFFFFFFF + 0xA282EAD8) & 0xFFFFFFFF


def test_entry_and_header_wire_bytes():
    # BundleEntryProto{dtype=DT_FLOAT(1), shape{dim{size:2} dim{size:3}}, offset=24, size=24, crc32c=0x01020304}
    got = tb._encode_entry(1, (2, 3), 0, 24, 24, 0x01020304)
    want = bytes([0x08, 0x01, 0x12, 0x08, 0x12, 0x02, 0x08, 0x02, 0x12, 0x02, 0x08, 0x03, 0x20, 0x18, 0x28, 0x18, 0x35, 0x04, 0x03, 0x02, 0x01])
    assert got == want
    e = tb._decode_entry(want)
    assert (e["dtype"], e["shape"], e["shard"], e["offset"], e["size"], e["crc"]) == (1, (2, 3), 0, 24, 24, 0x01020304)
    # scalar: empty shape message; offset 0 omitted (proto3 default)
    assert tb._encode_entry(9, (), 0, 0, 8, 7) == bytes([0x08, 0x09, 0x12, 0x00, 0x28, 0x08, 0x35, 0x07, 0, 0, 0])
    # BundleHeaderProto{num_shards=1, version{producer=1}}
    assert tb._encode_header(1) == bytes([0x08, 0x01, 0x1A, 0x02, 0x08, 0x01])
    assert tb._decode_header(bytes([0x08, 0x01, 0x1A, 0x02, 0x08, 0x01])) == dict(num_shards=1, endianness=0, producer=1)


def test_table_layout_small(tmp_path):
    p = str(tmp_path / "t.sst")
    tb.write_table(p, [(b"", b"H"), (b"ab", b"1"), (b"abc", b"22")])
    data = open(p, "rb").read()
    # one data block: entries (shared, non_shared, vlen, key suffix, value), restart array [0], count 1
    block = bytes([0, 0, 1]) + b"H" + bytes([0, 2, 1]) + b"ab1" + bytes([2, 1, 2]) + b"c22" + struct.pack("<II", 0, 1)
    assert data[:len(block)] == block
    assert data[len(block)] == 0                                                          # kNoCompression
    assert struct.unpack("<I", data[len(block) + 1:len(block) + 5])[0] == tb._mask(crc32c(block + b"\x00"))
    assert struct.unpack("<Q", data[-8:])[0] == 0xDB4775248B80FB57 and len(data[-48:]) == 48
    assert tb.read_table(p) == [(b"", b"H"), (b"ab", b"1"), (b"abc", b"22")]
    with pytest.raises(ValueError):
        tb.write_table(p, [(b"b", b"1"), (b"a", b"2")])


def test_table_many_blocks_and_restarts(tmp_path):
    p = str(tmp_path / "t.sst")
    items = [(("var/%05d/kernel" % i).encode(), os.urandom(1 + i % 37)) for i in range(700)]
    tb.write_table(p, items, block_size=512, restart_interval=4)
    assert tb.read_table(p) == items
    raw = bytearray(open(p, "rb").read())
    raw[10] ^= 0x40
    open(p, "wb").write(raw)
    with pytest.raises(ValueError):
        tb.read_table(p)


def test_bundle_round_trip_and_corruption(tmp_path):
    rng = np.random.default_rng(0)
    tensors = {"efficientnet-b0/stem/conv2d/kernel": rng.standard_normal((3, 3, 3, 32)).astype(np.float32),
               "efficientnet-b0/blocks_0/tpu_batch_normalization/moving_variance": rng.random(32).astype(np.float32),
               "global_step": np.array(1234, dtype=np.int64), "beta1_power": np.array(0.5, dtype=np.float32),
               "flags": np.array([True, False]), "big": rng.standard_normal((70, 1000)).astype(np.float32), "empty": np.zeros((0, 4), np.float32)}
    prefix = str(tmp_path / "model.ckpt-7")
    tb.write_bundle(prefix, tensors)
    assert sorted(os.listdir(tmp_path)) == ["model.ckpt-7.data-00000-of-00001", "model.ckpt-7.index"]
    back = tb.read_bundle(prefix)
    assert set(back) == set(tensors)
    for k, v in tensors.items():
        assert back[k].dtype == v.dtype and back[k].shape == v.shape and np.array_equal(back[k], v), k
    info = tb.list_bundle(prefix)
    names = sorted(tensors, key=lambda s: s.encode())
    assert [info[n]["offset"] for n in names] == list(np.cumsum([0] + [tensors[n].nbytes for n in names])[:-1])   # back to back, key order
    assert tb.read_bundle(prefix, ["global_step"])["global_step"] == 1234
    with pytest.raises(KeyError):
        tb.read_bundle(prefix, ["nope"])
    d = prefix + ".data-00000-of-00001"
    raw = bytearray(open(d, "rb").read())
    raw[info["big"]["offset"] + 5] ^= 1
    open(d, "wb").write(raw)
    with pytest.raises(ValueError, match="checksum"):
        tb.read_bundle(prefix)
    assert tb.read_bundle(prefix, ["global_step"])["global_step"] == 1234      # untouched tensors still verify


def test_checkpoint_directory_with_bundles(tmp_path):
    vals = {"a/kernel": np.arange(6, dtype=np.float32).reshape(2, 3), "a/bias": np.ones(3, np.float32)}
    s = ckpt.Saver(max_to_keep=2, fmt="tf")
    for step in (1, 2, 3):
        s.save({k: v + step for k, v in vals.items()}, str(tmp_path), step)
    files = sorted(os.listdir(tmp_path))
    assert files == ["checkpoint", "model.ckpt-2.data-00000-of-00001", "model.ckpt-2.index", "model.ckpt-3.data-00000-of-00001",
                     "model.ckpt-3.index"]
    path = ckpt.latest_checkpoint(str(tmp_path))
    assert path.endswith("model.ckpt-3")
    back = ckpt.load(path)
    assert np.array_equal(back["a/kernel"], vals["a/kernel"] + 3)
    with pytest.raises(ValueError):
        ckpt.Saver(fmt="hdf5")


def test_bundle_written_by_an_independent_writer_reads_back(tmp_path):
    """tests/independent_bundle.py shares no code with mliis_amd.tfbundle (own CRC-32C, no prefix compression, a restart point per
    entry, small data blocks -> a multi-entry index block, explicit zero-valued proto fields): the product READER must restore its
    files, dtype / shape / values and checksums."""
    from tests.independent_bundle import write_bundle_independent, _crc32c
    assert _crc32c(b"123456789") == 0xE3069283 == crc32c(b"123456789")
    rng = np.random.default_rng(3)
    tensors = {"efficientnet-b0/stem/conv2d/kernel": rng.standard_normal((3, 3, 3, 32)).astype(np.float32),
               "efficientnet-b0/stem/tpu_batch_normalization/gamma": rng.random(32).astype(np.float32),
               "decode/final_layer_weights/bias": np.array([0.5, -0.25], np.float32), "global_step": np.array(77, dtype=np.int64),
               "adam_step": np.array(200000, dtype=np.int64), "beta2_power": np.array(0.0, dtype=np.float32),
               "wide": rng.standard_normal((5, 700)).astype(np.float64), "ints": np.arange(12, dtype=np.int32).reshape(3, 4)}
    for epb in (1, 3, 100):
        prefix = str(tmp_path / ("model.ckpt-%d" % epb))
        write_bundle_independent(prefix, tensors, entries_per_block=epb)
        assert tb.is_bundle(prefix)
        back = tb.read_bundle(prefix)
        assert set(back) == set(tensors)
        for k, v in tensors.items():
            assert back[k].dtype == v.dtype and back[k].shape == v.shape and np.array_equal(back[k], v), (epb, k)
        assert tb.list_bundle(prefix)["global_step"]["shape"] == ()
    # and a flipped payload bit is caught through the independent writer's checksum
    d = prefix + ".data-00000-of-00001"
    raw = bytearray(open(d, "rb").read())
    raw[3] ^= 2
    open(d, "wb").write(raw)
    with pytest.raises(ValueError, match="checksum"):
        tb.read_bundle(prefix)
