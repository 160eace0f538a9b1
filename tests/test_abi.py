"""The C-ABI library loads and exports every symbol include/gamer_hip.h declares, and the ctypes
signatures in gamer_amd/_lib.py agree with the header prototypes (no GPU needed)."""
import ctypes as C
import os
import re

import pytest

from gamer_amd import _lib


def _parse_header():
    txt = open(_lib.HEADER_PATH).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int64_t|int|const char\*)\s+(gamer_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S):
        name, args = m.group(2), m.group(3).strip()
        if args == "void":
            protos[name] = []
            continue
        kinds = []
        for a in args.split(","):
            a = " ".join(a.split())
            if "*" in a:
                kinds.append("P")
            elif a.startswith("int64_t"):
                kinds.append("L")
            elif a.startswith("uint64_t"):
                kinds.append("U")
            elif a.startswith("float"):
                kinds.append("F")
            elif a.startswith("int"):
                kinds.append("I")
            else:
                raise AssertionError(f"unparsed argument {a!r} of {name}")
        protos[name] = kinds
    return protos


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        from gamer_amd import build
        build.build()
    return _lib.load()


def test_exports_every_declared_symbol(lib):
    protos = _parse_header()
    assert len(protos) >= 25
    for name in protos:
        assert hasattr(lib, name), f"{name} declared in gamer_hip.h but not exported"
    assert lib.gamer_abi_version() == 9


def test_ctypes_signatures_match_header(lib):
    protos = _parse_header()
    kind_of = {C.c_void_p: "P", C.c_int: "I", C.c_float: "F", C.c_int64: "L", C.c_uint64: "U"}
    for name, kinds in protos.items():
        if name in ("gamer_abi_version", "gamer_last_error"):
            continue
        assert name in _lib._SIGNATURES, f"no ctypes signature for {name}"
        got = [kind_of.get(t, "P") for t in _lib._SIGNATURES[name]]
        assert got == kinds, f"{name}: ctypes {got} != header {kinds}"
    assert set(_lib._SIGNATURES) <= set(protos)


def test_bad_arguments_are_reported_not_crashing(lib):
    # null pointers / bad shapes are rejected on the host before any launch
    rc = lib.gamer_fill_f32(None, 16, 0.0, None)
    assert rc != 0 and b"gamer_fill_f32" in lib.gamer_last_error()
    d = _lib.GemmDesc()
    rc = lib.gamer_gemm_f32(C.byref(d), None)
    assert rc != 0 and b"gamer_gemm_f32" in lib.gamer_last_error()
    d16 = _lib.GemmBf16Desc()
    rc = lib.gamer_gemm_bf16(C.byref(d16), None)
    assert rc != 0 and b"gamer_gemm_bf16" in lib.gamer_last_error()
