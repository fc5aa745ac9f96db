"""ctypes binding of libbevyray_amd.so (the C ABI in include/bevyray_amd.h).

The shared object is built in-tree by ``bevyray_amd/csrc/Makefile`` (hipcc, gfx950).  If it
is missing or older than its sources it is rebuilt on import; if that fails the import
fails -- there is no Python or CPU fallback for the render path.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("BRT_LIB_PATH") or os.path.join(_HERE, "libbevyray_amd.so")   # BRT_LIB_PATH: A/B of builds
_SOURCES = ["brt_api.cpp", "brt_interop.cpp", "brt_ctx.h", "brt_host.cpp", "brt_kernels.hip", "brt_trace_prod.hip", "brt_trace_tune.hip", "brt_trace.h",
            "brt_host.h", "brt_kernels.h", "brt_layout.h", "brt_device.h", "brt_ploc.h", "brt_sah.h", "brt_srgb_table.h", "brt_bvh.hip", "brt_sah.hip", "brt_order.hip", "Makefile"]

_lock = threading.Lock()
_lib = None


class BrtStats(C.Structure):
    _fields_ = [
        ("rays", C.c_uint64), ("node_pops", C.c_uint64), ("interior_visits", C.c_uint64),
        ("sphere_tests", C.c_uint64), ("hits", C.c_uint64), ("paths", C.c_uint64),
        ("kernel_ms", C.c_double), ("gather_ms", C.c_double), ("total_ms", C.c_double),
        ("lds_bytes", C.c_uint32), ("scene_in_lds", C.c_uint32), ("n_workgroups", C.c_uint32),
        ("threads_per_workgroup", C.c_uint32), ("prepass_ms", C.c_double),
        ("kernel_variant", C.c_uint32), ("measured_tile_costs", C.c_uint32),
        ("tree_rebuilt", C.c_uint32), ("tree_reach", C.c_float), ("forwarded_bytes", C.c_uint64),
        ("hot_records", C.c_uint32), ("reserved", C.c_uint32),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    paths = [os.path.join(_CSRC, s) for s in _SOURCES] + [os.path.join(_HERE, "..", "include", "bevyray_amd.h")]
    return any(os.path.exists(p) and os.path.getmtime(p) > t for p in paths)


def build(force: bool = False) -> str:
    """Compile the HIP extension for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if os.environ.get("BRT_LIB_PATH"):
        return LIB_PATH          # an explicitly chosen build is used as it is
    with _lock:
        if force or _stale():
            cmd = ["make", "-C", _CSRC, "-j", str(min(6, os.cpu_count() or 1))] + (["-B"] if force else [])
            proc = subprocess.run(cmd, capture_output=True, text=True)
            if proc.returncode != 0:
                raise RuntimeError("building libbevyray_amd.so failed:\n" + proc.stdout + proc.stderr)
    return LIB_PATH


_VP, _U32, _I32, _F = C.c_void_p, C.c_uint32, C.c_int32, C.c_float
_PROTOTYPES = {
    # name: (restype, argtypes) -- must list every export of include/bevyray_amd.h
    "brt_abi_version": (_U32, []),
    "brt_last_error": (C.c_char_p, [_VP]),
    "brt_create": (_I32, [C.POINTER(_I32), _I32, C.POINTER(_VP)]),
    "brt_destroy": (_I32, [_VP]),
    "brt_set_policy": (_I32, [_VP, _U32]),
    "brt_set_tuning": (_I32, [_VP, C.c_char_p, _U32]),
    "brt_get_tuning": (_I32, [_VP, C.c_char_p, C.POINTER(_U32), C.POINTER(_U32)]),
    "brt_upload_scene": (_I32, [_VP, _VP, _U32, _VP, _U32, _VP, _U32]),
    "brt_render": (_I32, [_VP, _VP, _VP, _U32, _U32, _U32, _VP, _VP, _VP, _U32, C.POINTER(BrtStats)]),
    "brt_host_alloc": (_I32, [_VP, C.c_uint64, C.POINTER(_VP)]),
    "brt_host_free": (_I32, [_VP, _VP]),
    "brt_render_part_device": (_I32, [_VP, _VP, _VP, _U32, _U32, _U32, _U32, _U32, _VP, _VP, _VP, _VP, _U32,
                                      C.POINTER(BrtStats)]),
    "brt_render_device": (_I32, [_VP, _VP, _VP, _U32, _U32, _U32, _VP, _VP, _VP, _VP, _U32, C.POINTER(BrtStats)]),
    "brt_tile_rows": (_U32, [_U32, _U32]),
    "brt_set_strip_table": (_I32, [_VP, _U32, _U32, _VP]),
    "brt_plan_strips": (_I32, [_VP, _VP, _VP, _U32, _U32, _U32, _U32, _U32, _VP]),
    "brt_host_plan_strips": (_I32, [_VP, _U32, _U32, _VP]),
    "brt_deinterleave_device": (_I32, [_VP, _VP, _U32, _U32, _U32, _VP, _VP, _U32]),
    "brt_rccl_unique_id": (_I32, [_VP]),
    "brt_rccl_comm_create": (_I32, [_VP, _VP, _I32, _I32, C.POINTER(_VP)]),
    "brt_rccl_comm_destroy": (_I32, [_VP, _VP]),
    "brt_gather_rccl": (_I32, [_VP, _VP, _I32, _I32, _VP, _VP, _U32, _U32, _VP, _VP, _U32]),
    "brt_import_frame_fd": (_I32, [_VP, _I32, C.c_uint64, _U32, C.POINTER(_VP)]),
    "brt_release_frame": (_I32, [_VP, _VP]),
    "brt_debug_export_frame_fd": (_I32, [_VP, C.c_uint64, C.POINTER(_I32), C.POINTER(_VP)]),
    "brt_debug_copy_to_host": (_I32, [_VP, _VP, _VP, C.c_uint64]),
    "brt_debug_eval": (_I32, [_VP, _U32, _VP, _VP, _U32]),
    "brt_debug_profile": (_I32, [_VP, C.POINTER(C.c_uint64)]),
    "brt_debug_tile_order": (_I32, [_VP, _VP, _VP, _U32, _U32, C.c_uint64, _U32, _U32, _U32, _VP, _VP]),
    "brt_build_bvh": (_I32, [_VP, _U32, _VP, _U32, C.POINTER(_U32)]),
    "brt_build_bvh_sah": (_I32, [_VP, _U32, _F, _VP, _U32, C.POINTER(_U32)]),
    "brt_host_srgb_thresholds": (_I32, [C.POINTER(_F)]),
    "brt_host_tree_reach": (_I32, [_VP, _U32, _VP, C.POINTER(_F), C.POINTER(_U32), C.POINTER(_F)]),
    "brt_build_bvh_device": (_I32, [_VP, _VP, _U32, _VP, _U32, C.POINTER(_U32), C.POINTER(C.c_double)]),
    "brt_build_bvh_sah_device": (_I32, [_VP, _VP, _U32, _F, _VP, _U32, C.POINTER(_U32), C.POINTER(C.c_double)]),
    "brt_validate_scene": (_I32, [_VP, _U32, _VP, _U32, _VP, _U32, C.POINTER(_U32)]),
    "brt_scene_generate": (_I32, [_U32, C.c_uint64, _VP, _VP, _U32, C.POINTER(_U32)]),
    "brt_host_camera_extract": (_I32, [C.POINTER(_F), C.POINTER(_F), C.POINTER(_F), _F, _F, _F, _F, _U32, _U32, _VP]),
    "brt_host_window_extract": (_I32, [_F, _U32, _VP]),
    "brt_host_material": (_I32, [C.POINTER(_F), _F, _F, _F, _F, _F, _VP]),
    "brt_host_tile_order": (_I32, [_VP, _VP, _U32, _U32, C.c_uint64, _U32, _U32, _U32, _U32, _U32, _VP, _VP]),
}
EXPORTS = tuple(_PROTOTYPES)


def _share_hip_runtime_with_torch() -> None:
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7, the same
    SONAME this library links against from /opt/rocm).  Two HIP runtimes in one process cannot
    both own the GPU ("No HIP GPUs are available"), and the dynamic loader shares by SONAME only
    with whatever was loaded FIRST.  So when torch is installed it is imported before
    libbevyray_amd.so is opened; both then run on torch's copy.  Processes without torch (the
    Rust/C++ hosts) simply get /opt/rocm's runtime.  BRT_NO_TORCH=1 skips this."""
    import importlib.util
    import sys
    if os.environ.get("BRT_NO_TORCH") == "1" or "torch" in sys.modules:
        return
    if importlib.util.find_spec("torch") is not None:
        import torch  # noqa: F401


def load() -> C.CDLL:
    """Load (building first if needed) the shared library and declare every prototype."""
    global _lib
    if _lib is not None:
        return _lib
    build()
    _share_hip_runtime_with_torch()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.brt_abi_version() != 6:
        raise RuntimeError("libbevyray_amd.so ABI version mismatch")
    _lib = lib
    return lib


def kernel_code_hash(path: str = LIB_PATH) -> str:
    """sha256 (first 16 hex digits) of the `.hip_fatbin` section of the shared library, i.e. of the gfx950
    code objects only: changes iff the device code changes (the build is deterministic), not when host
    code does.  profiles/*.json record it next to the PMC counters they were measured on, and bench.py
    drops file-sourced counters whose hash differs from the library it is timing."""
    import hashlib
    import struct
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"\x7fELF" or data[4] != 2:
        raise RuntimeError(f"{path}: not an ELF64 file")
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
    def sec(i):
        name, _type, _flags, _addr, off, size = struct.unpack_from("<IIQQQQ", data, shoff + i * shentsize)
        return name, off, size
    _, stroff, strsize = sec(shstrndx)
    strtab = data[stroff:stroff + strsize]
    for i in range(shnum):
        name, off, size = sec(i)
        if strtab[name:strtab.index(b"\0", name)] == b".hip_fatbin":
            return hashlib.sha256(data[off:off + size]).hexdigest()[:16]
    raise RuntimeError(f"{path}: no .hip_fatbin section")


class BrtError(RuntimeError):
    def __init__(self, code: int, text: str):
        super().__init__(f"bevyray_amd error {code}: {text}")
        self.code = code
        self.text = text


def check(rc: int, ctx=None) -> None:
    if rc != 0:
        msg = load().brt_last_error(ctx)
        raise BrtError(rc, msg.decode("utf-8", "replace") if msg else "")
