"""FSS-1000 / FP-k task shards without TensorFlow: TFRecord-GZIP framing + the two-feature `tf.train.Example` the reference's
converter writes (data/fss_1000_image_to_tfrecord.py:99-178: features `image` and `mask`, raw uint8 bytes), parsed with the
semantics of data/input_fn.py:28-65 (image -> float32 0..255 [H,W,3]; mask -> 2-channel (255-m, m)/255) and grouped into tasks
like meta_learners/metaseg.py:24-121 (one task per shard, canonical 240-task test split, `sample(n)` = first n examples).

Wire formats (public, stable): a TFRecord is  u64 length | u32 masked-crc32c(length) | data | u32 masked-crc32c(data);  an Example
is protobuf  Example{1: Features{1: map<string, Feature{1: BytesList{1: bytes}}>}}.  A writer is included so tests and users can
produce shards without TensorFlow.  Host-side integer/byte work only.
"""
from __future__ import annotations

import glob
import gzip
import os
import random
import struct
from typing import Dict, Iterator, List, Optional, Sequence, Tuple

import numpy as np

# ------------------------------------------------------------------------------------------------ crc32c (Castagnoli), masked
_POLY = 0x82F63B78
_TABLE = []
for _i in range(256):
    _c = _i
    for _ in range(8):
        _c = (_c >> 1) ^ (_POLY if _c & 1 else 0)
    _TABLE.append(_c)
_TABLE_NP = np.array(_TABLE, dtype=np.uint32)


# slicing-by-8 tables: _T8[k][b] = crc of byte b followed by k zero bytes (checkpoint tensors are megabytes: 8 bytes per iteration)
_T8 = [_TABLE]
for _k in range(1, 8):
    _prev = _T8[-1]
    _T8.append([_TABLE[_prev[_b] & 0xFF] ^ (_prev[_b] >> 8) for _b in range(256)])


def crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    n8 = len(data) // 8
    if n8:
        t0, t1, t2, t3, t4, t5, t6, t7 = _T8
        for (q,) in struct.iter_unpack("<Q", memoryview(data)[:n8 * 8]):
            q ^= c
            c = (t7[q & 0xFF] ^ t6[(q >> 8) & 0xFF] ^ t5[(q >> 16) & 0xFF] ^ t4[(q >> 24) & 0xFF] ^ t3[(q >> 32) & 0xFF] ^
                 t2[(q >> 40) & 0xFF] ^ t1[(q >> 48) & 0xFF] ^ t0[q >> 56])
    for b in memoryview(data)[n8 * 8:]:
        c = _TABLE[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc(data: bytes) -> int:
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------ record framing
def read_records(path: str, verify_data_crc: bool = False) -> Iterator[bytes]:
    opener = gzip.open if _is_gzip(path) else open
    with opener(path, "rb") as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) < 12:
                raise ValueError("{}: truncated record header".format(path))
            (length,), (lcrc,) = struct.unpack("<Q", head[:8]), struct.unpack("<I", head[8:])
            if masked_crc(head[:8]) != lcrc:
                raise ValueError("{}: corrupt record length (crc mismatch)".format(path))
            data = f.read(length)
            tail = f.read(4)
            if len(data) < length or len(tail) < 4:
                raise ValueError("{}: truncated record".format(path))
            if verify_data_crc and masked_crc(data) != struct.unpack("<I", tail)[0]:
                raise ValueError("{}: corrupt record data (crc mismatch)".format(path))
            yield data


def _is_gzip(path: str) -> bool:
    with open(path, "rb") as f:
        return f.read(2) == b"\x1f\x8b"


def write_records(path: str, records: Sequence[bytes], compress: bool = True):
    opener = gzip.open if compress else open
    with opener(path, "wb") as f:
        for data in records:
            head = struct.pack("<Q", len(data))
            f.write(head + struct.pack("<I", masked_crc(head)) + data + struct.pack("<I", masked_crc(data)))


# ------------------------------------------------------------------------------------------------ minimal protobuf
def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    shift = val = 0
    while True:
        b = buf[pos]
        pos += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, pos
        shift += 7


def _fields(buf: bytes) -> Iterator[Tuple[int, int, bytes]]:
    pos = 0
    while pos < len(buf):
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 2:
            ln, pos = _varint(buf, pos)
            yield fno, wt, buf[pos:pos + ln]
            pos += ln
        elif wt == 0:
            v, pos = _varint(buf, pos)
            yield fno, wt, v
        elif wt == 5:
            yield fno, wt, buf[pos:pos + 4]
            pos += 4
        elif wt == 1:
            yield fno, wt, buf[pos:pos + 8]
            pos += 8
        else:
            raise ValueError("unsupported protobuf wire type {}".format(wt))


def parse_example_bytes(example: bytes) -> Dict[str, bytes]:
    """tf.train.Example -> {feature name: first bytes_list value}."""
    out: Dict[str, bytes] = {}
    for fno, _, features in _fields(example):
        if fno != 1:
            continue
        for f2, _, entry in _fields(features):
            if f2 != 1:
                continue
            key, val = None, None
            for f3, _, v in _fields(entry):
                if f3 == 1:
                    key = v.decode()
                elif f3 == 2:
                    for f4, _, lst in _fields(v):
                        if f4 == 1:  # BytesList
                            for f5, _, item in _fields(lst):
                                if f5 == 1 and val is None:
                                    val = item
            if key is not None and val is not None:
                out[key] = val
    return out


def _enc_varint(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _ld(fno: int, payload: bytes) -> bytes:
    return _enc_varint((fno << 3) | 2) + _enc_varint(len(payload)) + payload


def make_example_bytes(image_u8: np.ndarray, mask_u8: np.ndarray) -> bytes:
    feats = b""
    for name, arr in (("image", image_u8), ("mask", mask_u8)):
        feature = _ld(1, _ld(1, np.ascontiguousarray(arr, dtype=np.uint8).tobytes()))   # Feature{bytes_list{value}}
        feats += _ld(1, _ld(1, name.encode()) + _ld(2, feature))                        # map entry
    return _ld(1, feats)                                                                 # Example{features}


# ------------------------------------------------------------------------------------------------ parse semantics (input_fn.py:28-65)
def parse_example(example: bytes, image_width: int = 224) -> Tuple[np.ndarray, np.ndarray]:
    f = parse_example_bytes(example)
    if "image" not in f or "mask" not in f:
        raise ValueError("example lacks the 'image' / 'mask' features")
    img = np.frombuffer(f["image"], dtype=np.uint8)
    msk = np.frombuffer(f["mask"], dtype=np.uint8)
    if img.size != image_width * image_width * 3 or msk.size != image_width * image_width:
        raise ValueError("example is not {0}x{0}: image {1} bytes, mask {2} bytes".format(image_width, img.size, msk.size))
    image = img.reshape(image_width, image_width, 3).astype(np.float32)
    m = msk.reshape(image_width, image_width)
    label = np.stack([255 - m, m], axis=2).astype(np.float32) / 255.0
    return image, label


class ShardTask:
    """One binary segmentation task = one shard (meta_learners/metaseg.py:181-230).  Examples are decoded once, lazily."""

    def __init__(self, path, image_size: int = 224, name: Optional[str] = None):
        """`path`: one shard, or a list of shards read one after another (the FP-k tasks pool the shards of several synonyms)."""
        self.tfrecord_paths = path
        self._paths = [path] if isinstance(path, str) else list(path)
        self.name = name or os.path.basename(self._paths[0])
        self.image_size = image_size
        self._images = self._labels = None
        self.batch_size = sum(1 for p in self._paths for _ in read_records(p))   # count_examples_in_tfrecords

    def _load(self):
        if self._images is None:
            pairs = [parse_example(r, self.image_size) for p in self._paths for r in read_records(p)]
            self._images = np.stack([p[0] for p in pairs])
            self._labels = np.stack([p[1] for p in pairs])

    def sample(self, num_images: int):
        if num_images > self.batch_size:
            raise ValueError("Tried to sample {} examples.Cannot sample more than {} examples that generator was initialized with.".format(
                num_images, self.batch_size))
        self._load()
        return self._images[:num_images], self._labels[:num_images]


def _list(name: str) -> List[str]:
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", name)) as f:
        return [ln.rstrip("\n") for ln in f if ln.strip()]


def fss_test_task_ids() -> List[str]:
    return _list("fss_test_set.txt")


def fp_k_test_task_ids() -> List[str]:
    return _list("fp-k_test_set.txt")


def split_train_test_tasks(all_tasks: List[str], n_test: int, reproducible_splits: bool = False):
    """data/fss_1000_utils.py:8-19."""
    all_tasks = sorted(all_tasks) if reproducible_splits else list(all_tasks)
    if not reproducible_splits:
        random.shuffle(all_tasks)
    test = [all_tasks.pop() for _ in range(n_test)]
    assert not set(test) & set(all_tasks), "train-test leakage"
    return all_tasks, test


def read_fss_1000_dataset(data_dir: str, num_val_tasks: int = 0, num_test_tasks: int = 240, test_task_ids: Optional[List[str]] = "fss",
                          image_size: int = 224):
    """(train_tasks, val_tasks, test_tasks, train_names, val_names, test_names) -- meta_learners/metaseg.py:24-121."""
    if test_task_ids == "fss":
        test_task_ids = fss_test_task_ids()
    shards = glob.glob(os.path.join(data_dir, "*.tfrecord*"))
    if test_task_ids is None:
        train_shards, test_shards = split_train_test_tasks(shards, num_test_tasks)
    else:
        ids = set(test_task_ids)
        key = lambda p: os.path.basename(p).replace(".tfrecord.gzip", "")  # noqa: E731
        test_shards = [p for p in shards if key(p) in ids]
        train_shards = [p for p in shards if key(p) not in ids]
    train_shards, val_shards = split_train_test_tasks(train_shards, num_val_tasks, reproducible_splits=True)
    print("{} training tasks, {} val tasks, {} test tasks.".format(len(train_shards), len(val_shards), len(test_shards)))
    mk = lambda ps: [ShardTask(p, image_size) for p in ps]  # noqa: E731
    tr, va, te = mk(train_shards), mk(val_shards), mk(test_shards)
    return tr, va, te, [t.name for t in tr], [t.name for t in va], [t.name for t in te]


DEFAULT_K_SHOT_SET = [{"airliner", "aeroplane"}, {"bus"}, {"motorbike"}, {"potted_plant", "potted plant"}, {"television", "tvmonitor"}]


def read_fp_k_shot_dataset(data_dir: str, all_task_names=DEFAULT_K_SHOT_SET, image_size: int = 224):
    """(tasks, task names) of the FP-k-shot set -- meta_learners/metaseg.py:124-179: one task per synonym set, pooling every shard
    whose file name contains one of the synonyms (blanks removed); the task is named after the first synonym the set iterates to
    (Python set order, as in the reference).  Shards are read in sorted-path order (the reference hands TensorFlow a list of globs)."""
    all_tasks = glob.glob(os.path.join(data_dir, "*.tfrecord*"))
    print("{} tasks found.".format(len(all_tasks)))
    print("Building k-shot-FSS-1000 test task samplers...")
    tasks, names = [], []
    for synonyms in all_task_names:
        shards, task_name = [], None
        for i, synonym in enumerate(synonyms):
            synonym = synonym.replace(" ", "")
            if i == 0:
                task_name = synonym
                print("Processing task: {}".format(task_name))
            shards.extend(sorted(x for x in all_tasks if synonym in os.path.basename(x) and x not in shards))
        print("task shards: {}".format(shards))
        if not shards:
            raise ValueError("no shards for task {} under {}".format(task_name, data_dir))
        t = ShardTask(shards, image_size, name=task_name)
        print("{} examples in task {}".format(t.batch_size, task_name))
        tasks.append(t)
        names.append(task_name)
    return tasks, names
