"""The bindings that nothing here can compile or run agree with the C header -- mechanically (hygiene):
integration/bevyray_amd_sys/src/lib.rs (no Rust toolchain in the image) against include/bevyray_amd.h: the same exports, argument
counts, argument and return types (integer widths, const-ness of pointers), constants, brt_stats field order and types and
BRT_ABI_VERSION; and the same for the ctypes binding, the built library and the Cargo manifest's ABI note.  No GPU needed."""
import ctypes
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
import abi_parse  # noqa: E402

HEADER = abi_parse.parse_header(os.path.join(ROOT, "include", "bevyray_amd.h"))
RUST = abi_parse.parse_rust(os.path.join(ROOT, "integration", "bevyray_amd_sys", "src", "lib.rs"))


def test_rust_binding_declares_exactly_the_headers_exports():
    assert sorted(RUST["functions"]) == sorted(HEADER["functions"])
    for name, (ret, args) in HEADER["functions"].items():
        r_ret, r_args = RUST["functions"][name]
        assert r_ret == ret, name
        assert [t for _, t in r_args] == [t for _, t in args], name       # types, in order (names are documentation)
        assert len(r_args) == len(args), name


def test_rust_binding_constants_stats_layout_and_abi_version():
    assert RUST["abi"] == HEADER["abi"] == 6
    for name, value in HEADER["constants"].items():
        assert RUST["constants"].get(name) == value, name
    assert not set(RUST["constants"]) - set(HEADER["constants"])
    assert RUST["stats"] == HEADER["stats"]                                # field order and widths of brt_stats
    manifest = open(os.path.join(ROOT, "integration", "bevyray_amd_sys", "Cargo.toml")).read()
    assert re.search(r"ABI version (\d+)", manifest).group(1) == str(HEADER["abi"])
    librs = open(os.path.join(ROOT, "integration", "bevyray_amd_sys", "src", "lib.rs")).read()
    assert f"BRT_ABI_VERSION {HEADER['abi']}" in librs


def test_ctypes_binding_and_the_built_library_match_the_header():
    from bevyray_amd import _lib
    assert sorted(_lib.EXPORTS) == sorted(HEADER["functions"])
    width = {"i32": ctypes.c_int32, "u32": ctypes.c_uint32, "u64": ctypes.c_uint64, "f32": ctypes.c_float, "f64": ctypes.c_double}
    for name, (ret, args) in HEADER["functions"].items():
        res, argtypes = _lib._PROTOTYPES[name]
        assert len(argtypes) == len(args), name
        for (an, t), ct in zip(args, argtypes):
            if t in width:
                assert ct is width[t], (name, an)
            else:                                                          # pointers: c_void_p, c_char_p or POINTER(...)
                assert ct in (ctypes.c_void_p, ctypes.c_char_p) or issubclass(ct, ctypes._Pointer), (name, an)
        if ret in width:
            assert res is width[ret], name
    assert [(n, {ctypes.c_uint64: "u64", ctypes.c_uint32: "u32", ctypes.c_double: "f64", ctypes.c_float: "f32"}[t]) for n, t in _lib.BrtStats._fields_] == HEADER["stats"]
    # every export is a defined symbol of the built library, and nothing else is exported under the brt_ prefix
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.build()], capture_output=True, text=True, check=True).stdout
    exported = sorted(ln.split()[-1] for ln in out.splitlines() if ln.split()[-1].startswith("brt_") and " T " in ln)
    assert exported == sorted(HEADER["functions"])
    assert _lib.load().brt_abi_version() == HEADER["abi"]
