"""The C ABI as a contract, checked without a GPU: every prototype in include/coda_hip.h is exported by the built
library and bound by coda_amd/_lib.py with the same arity and the same scalar kind in every slot (the ctypes table
is maintained by hand: one int / long long / float slip would corrupt arguments silently)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_KIND = {ctypes.c_void_p: "ptr", ctypes.c_char_p: "ptr", ctypes.c_int: "int", ctypes.c_longlong: "longlong",
         ctypes.c_ulonglong: "ulonglong", ctypes.c_float: "float"}


def _kind_of_c(decl: str) -> str:
    d = " ".join(decl.replace("const", " ").split())
    if "*" in d:
        return "ptr"
    base = d.rsplit(" ", 1)[0] if " " in d else d          # drop the parameter name
    base = " ".join(base.split())
    return {"int": "int", "long long": "longlong", "unsigned long long": "ulonglong", "float": "float", "void": "void"}[base]


def header_prototypes():
    hdr = open(os.path.join(ROOT, "include", "coda_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    hdr = re.sub(r"^\s*#.*$", "", hdr, flags=re.M)
    protos = {}
    for ret, name, args in re.findall(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(coda_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", hdr):
        args = args.strip()
        kinds = [] if args in ("", "void") else [_kind_of_c(a) for a in args.split(",")]
        rk = "ptr" if "*" in ret else _kind_of_c(ret.strip() + " x")
        protos[name] = (rk, kinds)
    return protos


def test_header_parses():
    protos = header_prototypes()
    assert len(protos) >= 20
    assert protos["coda_version"] == ("int", [])
    assert protos["coda_last_error"][0] == "ptr"
    assert protos["coda_rmsnorm"] == ("int", ["ptr", "longlong", "ptr", "ptr", "longlong", "int", "int", "float", "ptr"])


def test_ctypes_table_matches_header():
    from coda_amd import _lib
    protos = header_prototypes()
    assert set(protos) == set(_lib.SIGNATURES), set(protos) ^ set(_lib.SIGNATURES)
    for name, (rk, kinds) in protos.items():
        res, args = _lib.SIGNATURES[name]
        assert _KIND[res] == rk, (name, "return", _KIND[res], rk)
        assert len(args) == len(kinds), (name, len(args), len(kinds))
        for i, (a, k) in enumerate(zip(args, kinds)):
            assert _KIND[a] == k, f"{name}: argument {i} is {k} in the header but {_KIND[a]} in _lib.SIGNATURES"


def test_library_exports_every_declared_symbol():
    from coda_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = _lib.load()
    for n in header_prototypes():
        assert hasattr(lib, n), n
    hdr = open(os.path.join(ROOT, "include", "coda_hip.h")).read()
    assert lib.coda_version() == int(re.search(r"#define CODA_ABI_VERSION (\d+)", hdr).group(1))
    assert lib.coda_lmhead_workspace_bytes(1024, 151936) > 0      # host-only arithmetic: no GPU needed
