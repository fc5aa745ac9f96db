"""Loads the CPU oracle (oracle/libbevyray_oracle.so).  TEST INFRASTRUCTURE ONLY: this module
lives under tests/ and is imported by tests, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the bevyray_amd package."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "libbevyray_oracle.so")


def usable_cores():
    """CPU cores this process may actually use: the affinity mask, capped by the cgroup CPU quota
    (the GPU boxes report 256 logical CPUs but grant e.g. 16 via cpu.max; more threads than
    that only get throttled)."""
    import math
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, math.ceil(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, math.ceil(q / period)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)



def build(force=False):
    src = os.path.join(ORACLE_DIR, "bevyray_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        proc = subprocess.run(["make", "-C", ORACLE_DIR] + (["-B"] if force else []), capture_output=True, text=True)
        if proc.returncode != 0:
            raise RuntimeError("oracle build failed:\n" + proc.stdout + proc.stderr)
    return LIB


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        VP, U32, F = C.c_void_p, C.c_uint32, C.c_float
        lib.oracle_render.restype = C.c_int
        lib.oracle_render.argtypes = [VP, U32, VP, U32, VP, U32, VP, VP, U32, U32, U32, U32, U32, VP, VP, VP, VP, C.c_int]
        lib.oracle_render_strided.restype = C.c_int
        lib.oracle_render_strided.argtypes = [VP, U32, VP, U32, VP, U32, VP, VP, U32, U32, U32, U32, U32, U32, VP, VP, VP, VP,
                                              C.c_int]
        lib.oracle_set_policy.restype = None
        lib.oracle_set_policy.argtypes = [C.c_int, C.c_int, C.c_int]
        lib.oracle_tan_half_fov.restype = F
        lib.oracle_tan_half_fov.argtypes = [F]
        lib.oracle_rng_next.restype = U32
        lib.oracle_rng_next.argtypes = [U32]
        lib.oracle_rng_float.restype = F
        lib.oracle_rng_float.argtypes = [C.POINTER(U32)]
        lib.oracle_seed.restype = U32
        lib.oracle_seed.argtypes = [F, U32, U32, U32, U32]
        lib.oracle_unit_ball.restype = None
        lib.oracle_unit_ball.argtypes = [C.POINTER(U32), C.POINTER(F)]
        lib.oracle_min.restype = F
        lib.oracle_min.argtypes = [F, F]
        lib.oracle_max.restype = F
        lib.oracle_max.argtypes = [F, F]
        lib.oracle_ray_bounding_dst.restype = F
        lib.oracle_ray_bounding_dst.argtypes = [C.POINTER(F)] * 4
        lib.oracle_hit_sphere.restype = F
        lib.oracle_hit_sphere.argtypes = [C.POINTER(F)] * 3 + [F]
        lib.oracle_encode_frame.restype = C.c_int
        lib.oracle_encode_frame.argtypes = [VP, C.c_uint64, C.c_int, VP]
        lib.oracle_raycast.restype = C.c_int
        lib.oracle_raycast.argtypes = [VP, U32, VP, U32, C.POINTER(F), C.POINTER(F), C.POINTER(F), C.POINTER(U32),
                                       C.POINTER(C.c_int)]

    def render(self, buffers, level, camera, window, width, height, raster_rgba=None, raster_depth=None,
               rows=None, threads=None, row_step=1):
        """Full frame (or rows r0, r0+row_step, ... < r1) -> (frame (H,W,4) f32, counters dict)."""
        models = np.ascontiguousarray(buffers.models)
        materials = np.ascontiguousarray(buffers.materials)
        bvh = np.ascontiguousarray(buffers.bvh)
        # (numpy re-packs a padded record dtype in concatenate & co.: 20-byte "models" would be walked as 32-byte ones)
        if (models.dtype.itemsize, materials.dtype.itemsize, bvh.dtype.itemsize) != (32, 32, 48):
            raise ValueError(f"oracle.render: buffers are not in the wire layout (record sizes {models.dtype.itemsize}, "
                             f"{materials.dtype.itemsize}, {bvh.dtype.itemsize}; expected 32, 32, 48)")
        out = np.zeros((height, width, 4), np.float32)
        cnt = (C.c_uint64 * 5)()
        r0, r1 = (0, height) if rows is None else rows
        rr = None if raster_rgba is None else np.ascontiguousarray(raster_rgba, np.float32)
        rd = None if raster_depth is None else np.ascontiguousarray(raster_depth, np.float32)
        if threads is None:
            threads = usable_cores()
        rc = self.lib.oracle_render_strided(models.ctypes.data, len(models), materials.ctypes.data, len(materials),
                                    bvh.ctypes.data, len(bvh), camera.ctypes.data, window.ctypes.data,
                                    int(level["level"][0]), width, height, r0, r1, row_step,
                                    None if rr is None else rr.ctypes.data, None if rd is None else rd.ctypes.data,
                                    out.ctypes.data, cnt, threads)
        if rc != 0:
            raise RuntimeError(f"oracle_render failed: {rc}")
        names = ["rays", "node_pops", "interior_visits", "sphere_tests", "hits"]
        return out, dict(zip(names, [int(x) for x in cnt]))

    def encode_frame(self, frame, fmt):
        """The colour target's store conversion of an (H, W, 4) f32 frame: fmt "srgb8" / "unorm8" -> (H, W, 4) u8, "f16" -> (H, W, 4) u16 bits."""
        frame = np.ascontiguousarray(frame, np.float32)
        code = {"srgb8": 1, "f16": 2, "unorm8": 3}[fmt]
        out = np.zeros(frame.shape, np.uint16 if fmt == "f16" else np.uint8)
        rc = self.lib.oracle_encode_frame(frame.ctypes.data, frame.size // 4, code, out.ctypes.data)
        if rc != 0:
            raise RuntimeError(f"oracle_encode_frame failed: {rc}")
        return out

    def policy(self, or_short_circuit=False, minmax="minnum", pow="mul"):
        """Context manager: render under an alternative policy (same names as numpy_restatement.DEFAULT_POLICY)."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            self.lib.oracle_set_policy(int(bool(or_short_circuit)), int(minmax == "select"), int(pow == "exp2log2"))
            try:
                yield self
            finally:
                self.lib.oracle_set_policy(0, 0, 0)
        return cm()

    def rng_floats(self, state, n):
        s = C.c_uint32(state)
        vals = [self.lib.oracle_rng_float(C.byref(s)) for _ in range(n)]
        return np.array(vals, np.float32), s.value

    def unit_ball(self, state):
        s = C.c_uint32(state)
        out = (C.c_float * 3)()
        self.lib.oracle_unit_ball(C.byref(s), out)
        return np.array(list(out), np.float32), s.value


_oracle = None


def load():
    global _oracle
    if _oracle is None:
        _oracle = Oracle(C.CDLL(build()))
    return _oracle
