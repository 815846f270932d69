"""A second, independent writer of TensorFlow V2 checkpoints (TensorBundle), used by the tests only, so that
`mliis_amd.tfbundle.read_bundle` and the learner restore from a file that `tfbundle.write_bundle` did NOT produce.

Written from the published formats alone (leveldb doc/table_format.md; tensorflow/core/protobuf/tensor_bundle.proto;
tensorflow/core/lib/hash/crc32c.h) and deliberately different from the product writer wherever the format leaves a choice:

    * no prefix compression (every entry has shared = 0) and a restart point at EVERY entry (the product: interval 16, shared prefixes)
    * one data block per few entries (small blocks -> a multi-entry index block), index keys = the last key of each block
    * its own bit-by-bit CRC-32C (reflected polynomial 0x82F63B78), no table lookup
    * protobuf fields written with explicit zero values for offset / shard_id (proto3 allows it; the product omits defaults)

No TensorFlow exists in this environment, so this is still not a TF-written file -- but reader and writer no longer share code."""
import struct

import numpy as np

_DT = {np.dtype("<f4"): 1, np.dtype("<f8"): 2, np.dtype("<i4"): 3, np.dtype("<i8"): 9}


def _crc32c(data: bytes) -> int:
    c = 0xFFFFFFFF
    for b in data:
        c ^= b
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
    return c ^ 0xFFFFFFFF


def _masked(c: int) -> int:
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _vi(v: int) -> bytes:
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        out.append(b | (0x80 if v else 0))
        if not v:
            return bytes(out)


def _block(entries) -> bytes:
    body, restarts = bytearray(), []
    for k, v in entries:
        restarts.append(len(body))
        body += _vi(0) + _vi(len(k)) + _vi(len(v)) + k + v
    for r in restarts:
        body += struct.pack("<I", r)
    body += struct.pack("<I", len(restarts))
    return bytes(body)


def _emit(out: bytearray, block: bytes):
    """append block + trailer {type 0 = uncompressed, masked crc32c(block + type)}; returns its BlockHandle (offset, size)"""
    h = (len(out), len(block))
    out += block + b"\x00" + struct.pack("<I", _masked(_crc32c(block + b"\x00")))
    return h


def write_bundle_independent(prefix: str, tensors: dict, entries_per_block: int = 3):
    names = sorted(tensors, key=lambda s: s.encode())
    data = bytearray()
    items = [(b"", bytes([0x08, 0x01]) + bytes([0x10, 0x00]) + bytes([0x1A, 0x02, 0x08, 0x01]))]   # header: num_shards 1, LITTLE, producer 1
    for n in names:
        a = np.asarray(tensors[n])
        if a.ndim:                       # (np.ascontiguousarray turns a 0-d array into shape (1,))
            a = np.ascontiguousarray(a)
        a = a.astype(a.dtype.newbyteorder("<")) if a.dtype.byteorder == ">" else a
        raw = a.tobytes()
        shape = b"".join(bytes([0x12]) + _vi(len(bytes([0x08]) + _vi(d))) + bytes([0x08]) + _vi(d) for d in a.shape)
        e = bytes([0x08]) + _vi(_DT[a.dtype]) + bytes([0x12]) + _vi(len(shape)) + shape + bytes([0x18, 0x00]) + bytes([0x20]) + _vi(len(data)) + \
            bytes([0x28]) + _vi(len(raw)) + bytes([0x35]) + struct.pack("<I", _masked(_crc32c(raw)))
        items.append((n.encode(), e))
        data += raw
    out, index = bytearray(), []
    for i in range(0, len(items), entries_per_block):
        chunk = items[i:i + entries_per_block]
        off, size = _emit(out, _block(chunk))
        index.append((chunk[-1][0], _vi(off) + _vi(size)))
    meta = _emit(out, _block([]))
    idx = _emit(out, _block(index))
    footer = _vi(meta[0]) + _vi(meta[1]) + _vi(idx[0]) + _vi(idx[1])
    out += footer + bytes(40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57)
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(out))
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        f.write(bytes(data))
