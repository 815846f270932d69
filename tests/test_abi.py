"""The C-ABI library loads, exports every symbol include/mliis_hip.h declares, and the ctypes table matches the header
(argument count and scalar kinds).  No compute calls: runs without a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mliis_hip.h")


def _decls():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(mliis_\w+)\s*\(([^;{]*?)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        params = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        out[name] = (ret, params)
    return out


def _kind(param: str):
    if "*" in param or "hipStream_t" in param:
        return C.c_void_p
    if param.startswith("long long"):
        return C.c_longlong
    if param.startswith("size_t"):
        return C.c_size_t
    if param.startswith("float"):
        return C.c_float
    if param.startswith("int"):
        return C.c_int
    raise AssertionError("unknown param kind: " + param)


def test_header_matches_ctypes_table():
    from mliis_amd._lib import SIGNATURES
    decls = _decls()
    assert set(decls) == set(SIGNATURES), set(decls) ^ set(SIGNATURES)
    for name, (ret, params) in decls.items():
        res, argtypes = SIGNATURES[name]
        assert len(params) == len(argtypes), name
        for p, a in zip(params, argtypes):
            k = _kind(p)
            if k is C.c_void_p:
                assert a is C.c_void_p or issubclass(a, C._Pointer), (name, p)
            else:
                assert a is k, (name, p, a)


def test_library_exports_every_symbol():
    from mliis_amd._lib import LIB_PATH, lib
    if not os.path.exists(LIB_PATH):
        import __graft_entry__ as g
        g.build()
    dll = lib.load()
    for name in _decls():
        assert hasattr(dll, name), name
    assert lib.size("mliis_version") >= 100


def test_workspace_queries_need_no_gpu():
    from mliis_amd._lib import lib
    assert lib.size("mliis_colreduce_workspace_floats", 100352, 32, 1, 2) > 0
    assert lib.size("mliis_dwconv_bwd_filter_workspace_floats", 8, 112, 112, 32, 3, 1) > 0
    assert lib.size("mliis_conv2d_bwd_filter_workspace_floats", 8, 56, 56, 360, 112, 3) >= 9 * 360 * 112
    assert lib.size("mliis_softmax_ce_workspace_floats", 8, 224, 224) > 0
    assert lib.size("mliis_colreduce_workspace_floats", 10, 30, 1, 1) == 0  # C % 4 != 0 -> rejected


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import mliis_amd._lib as L
    fresh = L._Lib()
    monkeypatch.setattr(L, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(L.MliisError):
        fresh.load()
