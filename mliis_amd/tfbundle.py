"""TensorFlow checkpoint files (TensorBundle V2) without TensorFlow: the container `tf.train.Saver` writes and restores in the
reference (train.py:54,129-131; models/efficientlab.py:398-443; utils/util.py:42-50,72-81) -- SURVEY.md 8(f)-3.

    <prefix>.index                    an SSTable (LevelDB table format): key "" -> BundleHeaderProto, key <variable name> ->
                                      BundleEntryProto {dtype, shape, shard_id, offset, size, crc32c}; keys sorted bytewise
    <prefix>.data-0000i-of-0000N      raw little-endian tensor bytes at [offset, offset + size) of shard i

tensorflow==1.15.4 is not vendored in the reference and cannot be installed here, and the reference ships no checkpoint file, so
this module restates the published formats (tensorflow/core/util/tensor_bundle/*, tensorflow/core/lib/io/{format,block,table}*,
leveldb table_format.md) and is exercised by round trips and hand-assembled files only: parity with a TF-written file is UNPINNED.

Table format: a sequence of blocks, each followed by a 5-byte trailer (compression type, masked crc32c of block + type); a block is
prefix-compressed entries (varint shared, varint non_shared, varint value_len, key suffix, value) + uint32 restart offsets + uint32
restart count; the index block maps a separator key (here: the last key of a data block) to the block's (offset, size); the 48-byte
footer holds the metaindex and index handles and the magic 0xdb4775248b80fb57.  TensorBundle writes its index uncompressed; a
Snappy-compressed block raises (no decompressor is available here).  Host-side integer/byte work only.
"""
from __future__ import annotations

import os
import struct
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np

from .tfrecord import _enc_varint, _fields, _ld, _varint, crc32c

TABLE_MAGIC = 0xDB4775248B80FB57
BLOCK_SIZE = 262144          # table::Options().block_size in TensorFlow
RESTART_INTERVAL = 16
_MASK_DELTA = 0xA282EAD8

# tensorflow/core/framework/types.proto
_DT = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 4: np.dtype("u1"), 5: np.dtype("<i2"), 6: np.dtype("i1"),
       9: np.dtype("<i8"), 10: np.dtype("?"), 17: np.dtype("<u2"), 19: np.dtype("<f2"), 22: np.dtype("<u4"), 23: np.dtype("<u8")}
_DT_REV = {v: k for k, v in _DT.items()}


def _mask(c: int) -> int:
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + _MASK_DELTA) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------ protobuf (hand-rolled, 3 messages)
def _vfield(fno: int, v: int) -> bytes:
    return _enc_varint(fno << 3) + _enc_varint(v & 0xFFFFFFFFFFFFFFFF)


def _encode_header(num_shards: int) -> bytes:
    # BundleHeaderProto {1: num_shards, 2: endianness (LITTLE = 0, omitted), 3: VersionDef {1: producer = 1}}
    return _vfield(1, num_shards) + _ld(3, _vfield(1, 1))


def _encode_entry(dtype: int, shape: Tuple[int, ...], shard: int, offset: int, size: int, crc_masked: int) -> bytes:
    out = _vfield(1, dtype)
    dims = b"".join(_ld(2, _vfield(1, d) if d else b"") for d in shape)       # TensorShapeProto {2: repeated Dim {1: size}}
    out += _ld(2, dims)
    if shard:
        out += _vfield(3, shard)
    if offset:
        out += _vfield(4, offset)
    if size:
        out += _vfield(5, size)
    out += _enc_varint((6 << 3) | 5) + struct.pack("<I", crc_masked)        # fixed32
    return out


def _decode_entry(buf: bytes) -> dict:
    e = dict(dtype=0, shape=(), shard=0, offset=0, size=0, crc=None, sliced=False)
    for fno, wt, v in _fields(buf):
        if fno == 1:
            e["dtype"] = v
        elif fno == 2:
            dims = []
            for f2, _, d in _fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _, s in _fields(d):
                        if f3 == 1:
                            size = s if s < (1 << 63) else s - (1 << 64)
                    dims.append(size)
            e["shape"] = tuple(dims)
        elif fno == 3:
            e["shard"] = v
        elif fno == 4:
            e["offset"] = v
        elif fno == 5:
            e["size"] = v
        elif fno == 6:
            e["crc"] = struct.unpack("<I", v)[0]
        elif fno == 7:
            e["sliced"] = True
    return e


def _decode_header(buf: bytes) -> dict:
    h = dict(num_shards=1, endianness=0, producer=0)
    for fno, _, v in _fields(buf):
        if fno == 1:
            h["num_shards"] = v
        elif fno == 2:
            h["endianness"] = v
        elif fno == 3:
            for f2, _, p in _fields(v):
                if f2 == 1:
                    h["producer"] = p
    return h


# ------------------------------------------------------------------------------------------------ LevelDB table: writer
class _BlockBuilder:
    def __init__(self, restart_interval: int):
        self.ri = restart_interval
        self.buf = bytearray()
        self.restarts = [0]
        self.count = 0
        self.last_key = b""
        self.n = 0

    def add(self, key: bytes, value: bytes):
        shared = 0
        if self.count < self.ri:
            m = min(len(key), len(self.last_key))
            while shared < m and key[shared] == self.last_key[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.count = 0
        self.buf += _enc_varint(shared) + _enc_varint(len(key) - shared) + _enc_varint(len(value)) + key[shared:] + value
        self.last_key = key
        self.count += 1
        self.n += 1

    def size_estimate(self) -> int:
        return len(self.buf) + 4 * len(self.restarts) + 4

    def finish(self) -> bytes:
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + struct.pack("<I", len(self.restarts))


def _handle(offset: int, size: int) -> bytes:
    return _enc_varint(offset) + _enc_varint(size)


def write_table(path: str, items: List[Tuple[bytes, bytes]], block_size: int = BLOCK_SIZE, restart_interval: int = RESTART_INTERVAL):
    """items: (key, value) pairs in strictly increasing bytewise key order."""
    out = bytearray()
    index = _BlockBuilder(1)

    def emit(contents: bytes) -> Tuple[int, int]:
        off = len(out)
        out.extend(contents)
        out.append(0)                                                        # kNoCompression
        out.extend(struct.pack("<I", _mask(crc32c(contents + b"\x00"))))
        return off, len(contents)

    blk = _BlockBuilder(restart_interval)
    prev = None
    for k, v in items:
        if prev is not None and not k > prev:
            raise ValueError("table keys must be strictly increasing")
        prev = k
        blk.add(k, v)
        if blk.size_estimate() >= block_size:
            off, size = emit(blk.finish())
            index.add(blk.last_key, _handle(off, size))
            blk = _BlockBuilder(restart_interval)
    if blk.n:
        off, size = emit(blk.finish())
        index.add(blk.last_key, _handle(off, size))
    meta = emit(_BlockBuilder(restart_interval).finish())
    idx = emit(index.finish())
    footer = _handle(*meta) + _handle(*idx)
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC)
    out.extend(footer)
    with open(path, "wb") as f:
        f.write(out)


# ------------------------------------------------------------------------------------------------ LevelDB table: reader
def _read_block(data: bytes, off: int, size: int, verify: bool) -> bytes:
    contents, trailer = data[off:off + size], data[off + size:off + size + 5]
    if len(contents) < size or len(trailer) < 5:
        raise ValueError("table block out of range")
    if trailer[0] != 0:
        raise ValueError("compressed table block (type {}): not supported".format(trailer[0]))
    if verify and _mask(crc32c(contents + trailer[:1])) != struct.unpack("<I", trailer[1:])[0]:
        raise ValueError("table block checksum mismatch")
    return contents


def _block_entries(block: bytes) -> Iterator[Tuple[bytes, bytes]]:
    if len(block) < 4:
        raise ValueError("table block too short")
    nrestarts = struct.unpack("<I", block[-4:])[0]
    limit = len(block) - 4 - 4 * nrestarts
    if limit < 0:
        raise ValueError("corrupt table block (restart array)")
    pos, key = 0, b""
    while pos < limit:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        if shared > len(key) or pos + non_shared + vlen > limit:
            raise ValueError("corrupt table block (entry)")
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def read_table(path: str, verify: bool = True) -> List[Tuple[bytes, bytes]]:
    with open(path, "rb") as f:
        data = f.read()
    if len(data) < 48 or struct.unpack("<Q", data[-8:])[0] != TABLE_MAGIC:
        raise ValueError("{}: not an SSTable (bad magic)".format(path))
    footer = data[-48:]
    pos = 0
    _, pos = _varint(footer, pos)
    _, pos = _varint(footer, pos)
    ioff, pos = _varint(footer, pos)
    isize, pos = _varint(footer, pos)
    out = []
    for _, h in _block_entries(_read_block(data, ioff, isize, verify)):
        boff, p = _varint(h, 0)
        bsize, p = _varint(h, p)
        out.extend(_block_entries(_read_block(data, boff, bsize, verify)))
    return out


# ------------------------------------------------------------------------------------------------ bundle
def _data_path(prefix: str, shard: int, num_shards: int) -> str:
    return "{}.data-{:05d}-of-{:05d}".format(prefix, shard, num_shards)


def write_bundle(prefix: str, tensors: Dict[str, np.ndarray]):
    """One shard, tensors back to back in key order (what a single-device tf.train.Saver produces)."""
    items = [(b"", _encode_header(1))]
    offset = 0
    with open(_data_path(prefix, 0, 1), "wb") as f:
        for name in sorted(tensors, key=lambda s: s.encode()):
            if not name:
                raise ValueError("empty variable name")
            a = np.asarray(tensors[name])
            dt = a.dtype.newbyteorder("<") if a.dtype.byteorder == ">" else a.dtype
            if np.dtype(dt) not in _DT_REV:
                raise ValueError("{}: dtype {} not supported".format(name, a.dtype))
            raw = np.ascontiguousarray(a, dtype=dt).tobytes()
            f.write(raw)
            items.append((name.encode(), _encode_entry(_DT_REV[np.dtype(dt)], a.shape, 0, offset, len(raw), _mask(crc32c(raw)))))
            offset += len(raw)
    write_table(prefix + ".index", items)


def list_bundle(prefix: str) -> Dict[str, dict]:
    """name -> {dtype (numpy), shape, shard, offset, size, crc}; raises on a non-bundle."""
    table = read_table(prefix + ".index")
    if not table or table[0][0] != b"":
        raise ValueError("{}.index: missing bundle header".format(prefix))
    header = _decode_header(table[0][1])
    if header["endianness"] != 0:
        raise ValueError("big-endian bundles are not supported")
    out = {}
    for k, v in table[1:]:
        e = _decode_entry(v)
        if e["sliced"]:
            raise ValueError("{}: partitioned (sliced) variables are not supported".format(k.decode()))
        if e["dtype"] not in _DT:
            raise ValueError("{}: DataType {} not supported".format(k.decode(), e["dtype"]))
        e["dtype"] = _DT[e["dtype"]]
        e["num_shards"] = header["num_shards"]
        out[k.decode()] = e
    return out


def read_bundle(prefix: str, names: Optional[List[str]] = None, verify: bool = True) -> Dict[str, np.ndarray]:
    entries = list_bundle(prefix)
    want = list(entries) if names is None else names
    files: Dict[int, object] = {}
    out = {}
    try:
        for n in want:
            if n not in entries:
                raise KeyError("variable {} not in checkpoint {}".format(n, prefix))
            e = entries[n]
            f = files.get(e["shard"])
            if f is None:
                f = files[e["shard"]] = open(_data_path(prefix, e["shard"], e["num_shards"]), "rb")
            f.seek(e["offset"])
            raw = f.read(e["size"])
            count = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
            if len(raw) != e["size"] or e["size"] != count * e["dtype"].itemsize:
                raise ValueError("{}: size mismatch (entry {} bytes, shape {} of {})".format(n, e["size"], e["shape"], e["dtype"]))
            if verify and e["crc"] is not None and _mask(crc32c(raw)) != e["crc"]:
                raise ValueError("{}: tensor checksum mismatch".format(n))
            out[n] = np.frombuffer(raw, dtype=e["dtype"]).reshape(e["shape"]).copy()
    finally:
        for f in files.values():
            f.close()
    return out


def is_bundle(prefix: str) -> bool:
    return os.path.exists(prefix + ".index")
