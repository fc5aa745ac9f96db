"""Host-side mirror of bevyray's plugin surface for the ray-tracing pass.

Names, fields and error behaviour follow the reference's Rust (reference
src/raytracing/mod.rs, extract.rs, pipeline.rs) so that tests read like tests of the
reference would; the arithmetic lives in the C++/HIP library behind the C ABI
(include/bevyray_amd.h).  Nothing here traces rays.

    RaytracePlugin            mod.rs:24-84      owns the GPU context (RaytracingPipeline::from_world)
    RaytracedCamera           mod.rs:86-91      {level, sample_count, bounces}
    Raytracing                mod.rs:94-101     Skip/FallbackRaster/FallbackRaytraced/Pure = 0..3
    RaytracedSphere           mod.rs:103-106    {radius}
    StandardMaterial          bevy 0.14 defaults of the fields extract.rs:200-207 reads
    CameraExtract / WindowExtract / RaytraceLevelExtract / RaytraceMaterial / Model / BVHNode
                              extract.rs:56-237 byte layouts (numpy structured dtypes)
    prepare_buffers           extract.rs:280-337
    RayTracingNode.run        pipeline.rs:58-220
"""
from __future__ import annotations

import ctypes as C
import enum
import math
from dataclasses import dataclass, field
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import BrtError, BrtStats

# ---- wire formats (extract.rs:56-61, 83-104, 181-189, 213-218, 229-237) ---------------------

MODEL_DTYPE = np.dtype({"names": ["position", "radius", "material_id"],
                        "formats": [("<f4", 3), "<f4", "<u4"], "offsets": [0, 12, 16], "itemsize": 32})
MATERIAL_DTYPE = np.dtype({"names": ["base_color", "metallic", "roughness", "reflectance", "ior", "specular_transmission"],
                           "formats": [("<f4", 3), "<f4", "<f4", "<f4", "<f4", "<f4"],
                           "offsets": [0, 12, 16, 20, 24, 28], "itemsize": 32})
BVH_NODE_DTYPE = np.dtype({"names": ["bounds_min", "bounds_max", "index", "model_count"],
                           "formats": [("<f4", 3), ("<f4", 3), "<u4", "<u4"], "offsets": [0, 16, 28, 32], "itemsize": 48})
CAMERA_DTYPE = np.dtype({"names": ["sample_count", "bounce_count", "projection", "near", "far", "fov", "aspect",
                                   "position", "direction", "up"],
                         "formats": ["<u4", "<u4", "<u4", "<f4", "<f4", "<f4", "<f4", ("<f4", 3), ("<f4", 3), ("<f4", 3)],
                         "offsets": [0, 4, 8, 12, 16, 20, 24, 32, 48, 64], "itemsize": 80})
WINDOW_DTYPE = np.dtype({"names": ["random_seed", "height"], "formats": ["<f4", "<u4"], "offsets": [0, 4], "itemsize": 16})
LEVEL_DTYPE = np.dtype({"names": ["level"], "formats": ["<u4"], "offsets": [0], "itemsize": 32})

STRIP_ROWS = 8
FLAG_COUNTERS = 1
FLAG_KERNEL_SIMPLE = 2
POLICY_OR_SHORT_CIRCUIT = 1   # brt_set_policy: the WGSL-spec reading of `||` in raytrace.wgsl:269 (default: both operands evaluated)
POLICY_MINMAX_SELECT = 2      # ... min / max by compare-select (default: minNum / maxNum)
POLICY_POW_EXP2_LOG2 = 4      # ... pow(x, 5) as exp2(5 log2 x) (default: multiplies)
EXTMEM_OPAQUE_FD, EXTMEM_DMABUF_FD = 1, 2   # brt_import_frame_fd handle types
FLAG_CALLER_STREAM = 4   # device entry points: `stream` is the caller's stream even when its handle is 0
# format of an assembled DEVICE frame (render_device, gather_rccl, deinterleave_device): the colour target's own (pipeline.rs:311-315)
FLAG_OUT_RGBA32F, FLAG_OUT_RGBA8_UNORM_SRGB, FLAG_OUT_RGBA16F, FLAG_OUT_RGBA8_UNORM = 0, 8, 16, 24
OUT_PIXEL_BYTES = {FLAG_OUT_RGBA32F: 16, FLAG_OUT_RGBA8_UNORM_SRGB: 4, FLAG_OUT_RGBA16F: 8, FLAG_OUT_RGBA8_UNORM: 4}

SCENE_COVER, SCENE_RTIOW_FINAL, SCENE_STRESS_GRID = 0, 1, 2


class Raytracing(enum.IntEnum):
    """mod.rs:94-101, #[repr(u32)]"""
    Skip = 0
    FallbackRaster = 1
    FallbackRaytraced = 2
    Pure = 3


@dataclass
class RaytracedCamera:
    """mod.rs:86-91"""
    level: Raytracing = Raytracing.FallbackRaytraced
    sample_count: int = 4
    bounces: int = 4


@dataclass
class RaytracedSphere:
    """mod.rs:103-106"""
    radius: float = 1.0


@dataclass
class StandardMaterial:
    """The StandardMaterial fields extract.rs:200-207 reads, with bevy 0.14's defaults.
    base_color is sRGB (Color::srgb), decoded to linear by RaytraceMaterial.prepare_asset."""
    base_color: Tuple[float, float, float] = (1.0, 1.0, 1.0)
    metallic: float = 0.0
    perceptual_roughness: float = 0.5
    reflectance: float = 0.5
    ior: float = 1.5
    specular_transmission: float = 0.0


@dataclass
class PerspectiveProjection:
    fov: float = math.pi / 4.0
    aspect_ratio: float = 1.0
    near: float = 0.1
    far: float = 1000.0


@dataclass
class OrthographicProjection:
    """Unsupported by the reference: CameraExtract returns None (extract.rs:148)."""
    scale: float = 1.0


@dataclass
class Transform:
    """Transform::from_translation(t).looking_at(target, up)"""
    translation: Tuple[float, float, float] = (0.0, 0.0, 5.0)
    target: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    up: Tuple[float, float, float] = (0.0, 1.0, 0.0)


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def _trim(arr: np.ndarray, n: int) -> np.ndarray:
    """First n records as an owned array, copied bytewise (padding bytes stay zero)."""
    raw = arr.view(np.uint8).reshape(len(arr), arr.dtype.itemsize)[:n].copy()
    return raw.view(arr.dtype).reshape(n)


class CameraExtract:
    """extract.rs:83-158"""

    @staticmethod
    def extract_component(camera: RaytracedCamera, transform: Transform, projection):
        if not isinstance(projection, PerspectiveProjection):
            return None  # extract.rs:148
        lib = _lib.load()
        cam = np.zeros(1, CAMERA_DTYPE)
        _lib.check(lib.brt_host_camera_extract(_f3(transform.translation), _f3(transform.target), _f3(transform.up),
                                               projection.fov, projection.aspect_ratio, projection.near, projection.far,
                                               int(camera.sample_count), int(camera.bounces), cam.ctypes.data))
        level = np.zeros(1, LEVEL_DTYPE)
        level["level"] = int(camera.level)
        return level, cam


class WindowExtract:
    """extract.rs:56-81.  The reference draws random_seed from thread_rng every frame; here
    it is an explicit input."""

    @staticmethod
    def extract_component(physical_height: int, random_seed: float):
        lib = _lib.load()
        win = np.zeros(1, WINDOW_DTYPE)
        _lib.check(lib.brt_host_window_extract(float(random_seed), int(physical_height), win.ctypes.data))
        return win


class RaytraceMaterial:
    """extract.rs:181-209"""

    @staticmethod
    def prepare_asset(source: StandardMaterial) -> np.ndarray:
        lib = _lib.load()
        out = np.zeros(1, MATERIAL_DTYPE)
        _lib.check(lib.brt_host_material(_f3(source.base_color), source.metallic, source.perceptual_roughness,
                                         source.reflectance, source.ior, source.specular_transmission, out.ctypes.data))
        return out


def build_bvh(models: np.ndarray) -> np.ndarray:
    """The build_ploc call + flatten of extract.rs:315-332 (native PLOC builder)."""
    lib = _lib.load()
    models = np.ascontiguousarray(models, MODEL_DTYPE)
    n = len(models)
    cap = max(1, 2 * n)
    nodes = np.zeros(cap, BVH_NODE_DTYPE)
    out_n = C.c_uint32(0)
    _lib.check(lib.brt_build_bvh(models.ctypes.data, n, nodes.ctypes.data, cap, C.byref(out_n)))
    return _trim(nodes, out_n.value)


def build_bvh_sah(models: np.ndarray, reach: float = 0.0) -> np.ndarray:
    """The binned-SAH tree that brt_upload_scene builds when the caller passes no BVH (brt_build_bvh_sah); reach: what the
    leaf pads cover (0 = the scene's own extent; tree_reach(models, camera) for a camera further out)."""
    lib = _lib.load()
    models = np.ascontiguousarray(models, MODEL_DTYPE)
    n = len(models)
    cap = max(1, 2 * n)
    nodes = np.zeros(cap, BVH_NODE_DTYPE)
    out_n = C.c_uint32(0)
    _lib.check(lib.brt_build_bvh_sah(models.ctypes.data, n, float(reach), nodes.ctypes.data, cap, C.byref(out_n)))
    return _trim(nodes, out_n.value)


def tree_reach(models: np.ndarray, camera: np.ndarray):
    """brt_host_tree_reach: (scene scale S, level, reach) the callee-built SAH tree needs for this camera."""
    lib = _lib.load()
    models = np.ascontiguousarray(models, MODEL_DTYPE)
    s, lvl, r = C.c_float(0), C.c_uint32(0), C.c_float(0)
    _lib.check(lib.brt_host_tree_reach(models.ctypes.data, len(models), camera.ctypes.data, C.byref(s), C.byref(lvl), C.byref(r)))
    return float(s.value), int(lvl.value), float(r.value)


def srgb_thresholds() -> np.ndarray:
    """brt_host_srgb_thresholds: the 255 f32 decision thresholds of the exact 8-bit sRGB encode (code = thresholds <= c)."""
    out = np.zeros(255, np.float32)
    _lib.check(_lib.load().brt_host_srgb_thresholds(out.ctypes.data_as(C.POINTER(C.c_float))))
    return out


def validate_scene(models, materials, bvh) -> int:
    """Returns the maximum leaf depth; raises BrtError for what brt_upload_scene would reject."""
    lib = _lib.load()
    models = np.ascontiguousarray(models, MODEL_DTYPE)
    materials = np.ascontiguousarray(materials, MATERIAL_DTYPE)
    bvh = np.ascontiguousarray(bvh, BVH_NODE_DTYPE)
    depth = C.c_uint32(0)
    _lib.check(lib.brt_validate_scene(models.ctypes.data, len(models), materials.ctypes.data, len(materials),
                                      bvh.ctypes.data, len(bvh), C.byref(depth)))
    return depth.value


@dataclass
class Buffers:
    """ModelBuffer / MaterialBuffer / BVHBuffer (extract.rs:252-262)."""
    models: np.ndarray
    materials: np.ndarray
    bvh: np.ndarray


def prepare_buffers(data: Sequence[Tuple[Tuple[float, float, float], RaytracedSphere, StandardMaterial]]) -> Buffers:
    """extract.rs:280-337: one Model + one material entry per sphere (material_id = enumerate
    index), AABBs padded by 0.1, PLOC BVH, flattened nodes."""
    n = len(data)
    models = np.zeros(n, MODEL_DTYPE)
    materials = np.zeros(n, MATERIAL_DTYPE)
    for i, (position, sphere, material) in enumerate(data):
        materials[i] = RaytraceMaterial.prepare_asset(material)[0]
        models[i]["position"] = position
        models[i]["radius"] = sphere.radius
        models[i]["material_id"] = i
    return Buffers(models, materials, build_bvh(models))


def generate_scene(kind: int, seed: int = 1) -> Buffers:
    """Seeded version of the demo scene setup (main.rs:49-240) and the other benchmark scenes."""
    lib = _lib.load()
    cap = 16384
    models = np.zeros(cap, MODEL_DTYPE)
    materials = np.zeros(cap, MATERIAL_DTYPE)
    n = C.c_uint32(0)
    _lib.check(lib.brt_scene_generate(kind, seed, models.ctypes.data, materials.ctypes.data, cap, C.byref(n)))
    models, materials = _trim(models, n.value), _trim(materials, n.value)
    return Buffers(models, materials, build_bvh(models))


def cover_camera(width: int, height: int, sample_count: int, bounces: int,
                 level: Raytracing = Raytracing.Pure, seed: float = 0.5):
    """The 'cover' view: position (13,2,3) looking at the origin, fov 0.4 rad,
    near 0.1, far 1000.  Returns (level, camera, window) extracts."""
    cam = RaytracedCamera(level=level, sample_count=sample_count, bounces=bounces)
    proj = PerspectiveProjection(fov=0.4, aspect_ratio=width / height, near=0.1, far=1000.0)
    lvl, cex = CameraExtract.extract_component(cam, Transform((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0)), proj)
    return lvl, cex, WindowExtract.extract_component(height, seed)


def rtiow_camera(width: int, height: int, sample_count: int, bounces: int,
                 level: Raytracing = Raytracing.Pure, seed: float = 0.5):
    """The book's final-scene view for BASELINE.json configs 3 and 4: lookfrom (13,2,3),
    lookat the origin, vfov 20 degrees = 0.34906585 rad, no defocus (the shader has none)."""
    cam = RaytracedCamera(level=level, sample_count=sample_count, bounces=bounces)
    proj = PerspectiveProjection(fov=0.34906585, aspect_ratio=width / height, near=0.1, far=1000.0)
    lvl, cex = CameraExtract.extract_component(cam, Transform((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0)), proj)
    return lvl, cex, WindowExtract.extract_component(height, seed)


def tile_rows(height: int, n_parts: int) -> int:
    return int(_lib.load().brt_tile_rows(height, n_parts))


class RaytracePlugin:
    """mod.rs:24-84.  build()/finish() create the GPU context (the reference queues the
    render pipeline in RaytracingPipeline::from_world, pipeline.rs:233-331)."""

    def __init__(self, device_ids: Sequence[int] = (0,)):
        lib = _lib.load()
        ids = (C.c_int32 * len(device_ids))(*device_ids)
        ctx = C.c_void_p()
        _lib.check(lib.brt_create(ids, len(device_ids), C.byref(ctx)))
        self._lib = lib
        self._ctx = ctx
        self.node = RayTracingNode(self)

    def close(self):
        if self._ctx:
            self._lib.brt_destroy(self._ctx)
            self._ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_policy(self, flags: int) -> None:
        """brt_set_policy: POLICY_OR_SHORT_CIRCUIT or 0 (the reading of `||` in raytrace.wgsl:269; changes pixels)."""
        _lib.check(self._lib.brt_set_policy(self._ctx, flags), self._ctx)

    def set_tuning(self, name: str, value: int) -> None:
        """brt_set_tuning: a scheduling / launch-shape knob of this context (never changes a pixel)."""
        _lib.check(self._lib.brt_set_tuning(self._ctx, name.encode(), int(value)), self._ctx)

    def get_tuning(self, name: str):
        """(value, default) of a knob."""
        v, d = C.c_uint32(0), C.c_uint32(0)
        _lib.check(self._lib.brt_get_tuning(self._ctx, name.encode(), C.byref(v), C.byref(d)), self._ctx)
        return int(v.value), int(d.value)

    def tuning(self, **knobs):
        """Context manager: `with plugin.tuning(BRT_LEAF_VOTE=8): ...` sets knobs and restores the previous values."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            old = {k: self.get_tuning(k)[0] for k in knobs}
            for k, v in knobs.items():
                self.set_tuning(k, v)
            try:
                yield self
            finally:
                for k, v in old.items():
                    self.set_tuning(k, v)
        return cm()

    def alloc_frame(self, width: int, height: int) -> np.ndarray:
        """Page-locked (height, width, 4) f32 frame owned by the context (brt_host_alloc): passing it
        as `out` to RayTracingNode.run lets the library DMA straight into it."""
        nbytes = width * height * 16
        ptr = C.c_void_p()
        _lib.check(self._lib.brt_host_alloc(self._ctx, nbytes, C.byref(ptr)), self._ctx)
        buf = (C.c_float * (width * height * 4)).from_address(ptr.value)
        return np.frombuffer(buf, np.float32).reshape(height, width, 4)

    def build_bvh(self, models: np.ndarray):
        """GPU PLOC build (brt_build_bvh_device): returns (nodes, kernel ms); same bytes as build_bvh()."""
        models = np.ascontiguousarray(models, MODEL_DTYPE)
        n = len(models)
        cap = max(1, 2 * n)
        nodes = np.zeros(cap, BVH_NODE_DTYPE)
        out_n = C.c_uint32(0)
        ms = C.c_double(0.0)
        _lib.check(self._lib.brt_build_bvh_device(self._ctx, models.ctypes.data, n, nodes.ctypes.data, cap, C.byref(out_n),
                                                  C.byref(ms)), self._ctx)
        return _trim(nodes, out_n.value), ms.value

    def build_bvh_sah(self, models: np.ndarray, reach: float = 0.0):
        """GPU binned-SAH build (brt_build_bvh_sah_device): returns (nodes, kernel ms); same bytes as build_bvh_sah()."""
        models = np.ascontiguousarray(models, MODEL_DTYPE)
        n = len(models)
        cap = max(1, 2 * n)
        nodes = np.zeros(cap, BVH_NODE_DTYPE)
        out_n = C.c_uint32(0)
        ms = C.c_double(0.0)
        _lib.check(self._lib.brt_build_bvh_sah_device(self._ctx, models.ctypes.data, n, float(reach), nodes.ctypes.data, cap, C.byref(out_n),
                                                      C.byref(ms)), self._ctx)
        return _trim(nodes, out_n.value), ms.value

    # -- frame targets in another API's memory (brt_import_frame_fd) --------------------------------------------------
    def import_frame_fd(self, fd: int, nbytes: int, handle_type: int = 2) -> int:
        """brt_import_frame_fd: maps the memory behind a file descriptor (EXTMEM_OPAQUE_FD = 1: a Vulkan opaque-fd export;
        EXTMEM_DMABUF_FD = 2: a dma-buf of a HIP virtual-memory allocation) and returns the device pointer."""
        ptr = C.c_void_p()
        _lib.check(self._lib.brt_import_frame_fd(self._ctx, int(fd), int(nbytes), int(handle_type), C.byref(ptr)), self._ctx)
        return int(ptr.value)

    def release_frame(self, d_frame: int) -> None:
        _lib.check(self._lib.brt_release_frame(self._ctx, d_frame), self._ctx)

    def debug_export_frame_fd(self, nbytes: int):
        """(fd, device pointer) of a fresh exportable allocation on the first device (the other side of import_frame_fd in tests)."""
        fd, ptr = C.c_int32(-1), C.c_void_p()
        _lib.check(self._lib.brt_debug_export_frame_fd(self._ctx, int(nbytes), C.byref(fd), C.byref(ptr)), self._ctx)
        return int(fd.value), int(ptr.value)

    def debug_copy_to_host(self, d_src: int, shape, dtype=np.float32) -> np.ndarray:
        out = np.empty(shape, dtype)
        _lib.check(self._lib.brt_debug_copy_to_host(self._ctx, d_src, out.ctypes.data, out.nbytes), self._ctx)
        return out

    # -- strips by measured cost (one process per GPU) --------------------------------------------------------------------
    def set_strip_table(self, n_parts: int, part_of_strip) -> None:
        """brt_set_strip_table: part_of_strip[s] = the part that renders frame strip s (a permutation of the parts inside every group of
        n_parts consecutive strips); None: back to s % n_parts.  Every rank sets the same table."""
        if part_of_strip is None:
            _lib.check(self._lib.brt_set_strip_table(self._ctx, int(n_parts), 0, None), self._ctx)
            return
        t = np.ascontiguousarray(part_of_strip, np.uint32)
        _lib.check(self._lib.brt_set_strip_table(self._ctx, int(n_parts), len(t), t.ctypes.data), self._ctx)

    def plan_strips(self, level, camera, window, width: int, height: int, n_parts: int, probe_spp: int = 4) -> np.ndarray:
        """brt_plan_strips: renders the frame once at probe_spp samples per pixel with the per-tile ray counts on, deals the strips of
        every group of n_parts out by cost, installs the table and returns it (deterministic: every rank gets the same one)."""
        t = np.zeros((height + STRIP_ROWS - 1) // STRIP_ROWS, np.uint32)
        _lib.check(self._lib.brt_plan_strips(self._ctx, camera.ctypes.data, window.ctypes.data, int(level["level"][0]), width, height,
                                             int(n_parts), int(probe_spp), t.ctypes.data), self._ctx)
        return t

    # -- the RCCL gather behind the C ABI (one process per GPU) -----------------------------------------------------------
    @staticmethod
    def rccl_unique_id() -> bytes:
        """brt_rccl_unique_id (ncclGetUniqueId): on one rank; the 128 bytes go to the other ranks by the host's own means."""
        buf = (C.c_char * 128)()
        _lib.check(_lib.load().brt_rccl_unique_id(buf))
        return bytes(buf.raw)

    def rccl_comm_create(self, unique_id: bytes, rank: int, world: int) -> int:
        assert len(unique_id) == 128
        comm = C.c_void_p()
        _lib.check(self._lib.brt_rccl_comm_create(self._ctx, unique_id, rank, world, C.byref(comm)), self._ctx)
        return int(comm.value)

    def rccl_comm_destroy(self, comm: int) -> None:
        _lib.check(self._lib.brt_rccl_comm_destroy(self._ctx, comm), self._ctx)

    def debug_profile(self) -> dict:
        """Lane-utilisation profile of the last FLAG_COUNTERS launch: section -> (executions, lanes)."""
        raw = (C.c_uint64 * 64)()
        _lib.check(self._lib.brt_debug_profile(self._ctx, raw), self._ctx)
        self.last_raw = [int(x) for x in raw]
        # -DBRT_ASM_COUNT builds: executions / active lanes of the hand-written loops of the last production launch (else zeros)
        self.last_asm_counts = {"interior": (int(raw[33]), int(raw[34])), "leaf": (int(raw[35]), int(raw[36])), "ball": (int(raw[37]), int(raw[38])),
                                "repairing_loop_lanes": (int(raw[39]), int(raw[43])), "rows_interior_exec": int(raw[44]), "rows_calls": int(raw[5]),
                                "rows_cycles": int(raw[6]), "wide_cycles": int(raw[7])}
        # sampler stage (knob BRT_BALL_SERVERS): what the server waves executed, pick-ups that ran into their bound (must be 0)
        self.last_sampler_stage = {"iterations": int(raw[37]), "lanes": int(raw[38]), "gave_up": int(raw[46]), "door_polls": int(raw[47]),
                                   "idle_polls_at_end_max": int(raw[48]), "pickup_lane_polls": int(raw[49])}
        names = ["interior", "leaf", "camera", "scatter", "sky", "ball", "camera_top", "round"]
        prof = {n: (int(raw[8 + 2 * k]), int(raw[9 + 2 * k])) for k, n in enumerate(names)}
        # wave time stamps (100 MHz wall clock): first start, first / last "pixel queue empty", last end
        self.last_order_meta = {"critical_tiles": int(raw[40]), "longest_pixel_rays": int(raw[41]), "split_tiles": int(raw[42]),
                                "second_halves_taken": int(raw[62]), "second_halves_left": int(raw[63])}
        t0 = (~int(raw[24])) & (2**64 - 1)
        if raw[29]:
            first_empty = ((~int(raw[25])) & (2**64 - 1)) if raw[25] else 0
            self.last_timeline = {
                "waves": int(raw[29]),
                "first_empty_ms": (first_empty - t0) / 1e5 if first_empty else None,
                "last_empty_ms": (int(raw[26]) - t0) / 1e5 if raw[26] else None,
                "end_ms": (int(raw[27]) - t0) / 1e5,
                "mean_wave_drain_ms": int(raw[28]) / 1e5 / int(raw[29]),
                "mean_wave_life_ms": int(raw[45]) / 1e5 / int(raw[29]),       # against end_ms: what the tail of the launch leaves idle
                "wave_life_hist_0.33ms": [(int(raw[46 + (b >> 2)]) >> (16 * (b & 3))) & 0xffff for b in range(64)],
                "drain_rounds": int(raw[31]),
                "drain_live_lanes_per_round": int(raw[30]) / max(1, int(raw[31])),
                # mean per wave, ms: pixel refill | walk loop | shading (of which the rejection-sampler loop) | drain
                # logic + camera ray + walk begin
                "wave_ms_refill_walk_shade_ball_pre": [int(raw[5]) / 1e5 / int(raw[29]), int(raw[6]) / 1e5 / int(raw[29]),
                                                       int(raw[7]) / 1e5 / int(raw[29]), int(raw[44]) / 1e5 / int(raw[29]),
                                                       int(raw[43]) / 1e5 / int(raw[29])],
            }
        return prof

    def debug_eval(self, op: int, inputs: np.ndarray) -> np.ndarray:
        inputs = np.ascontiguousarray(inputs, np.float32)
        assert inputs.ndim == 2 and inputs.shape[1] == 16
        out = np.zeros((inputs.shape[0], 8), np.float32)
        _lib.check(self._lib.brt_debug_eval(self._ctx, op, inputs.ctypes.data, out.ctypes.data, inputs.shape[0]), self._ctx)
        return out


class RayTracingNode:
    """pipeline.rs:29-221.  `run` uploads the three storage buffers and draws the frame."""

    def __init__(self, plugin: RaytracePlugin):
        self._p = plugin
        self.last_stats: Optional[dict] = None

    def write_buffers(self, buffers: Buffers) -> None:
        """pipeline.rs:136-138"""
        p = self._p
        models = np.ascontiguousarray(buffers.models, MODEL_DTYPE)
        materials = np.ascontiguousarray(buffers.materials, MATERIAL_DTYPE)
        bvh = None if buffers.bvh is None else np.ascontiguousarray(buffers.bvh, BVH_NODE_DTYPE)
        _lib.check(p._lib.brt_upload_scene(p._ctx, models.ctypes.data, len(models), materials.ctypes.data, len(materials),
                                           None if bvh is None else bvh.ctypes.data, 0 if bvh is None else len(bvh)), p._ctx)

    def run(self, level, camera, window, width: int, height: int, buffers: Optional[Buffers] = None,
            raster_rgba: Optional[np.ndarray] = None, raster_depth: Optional[np.ndarray] = None,
            flags: int = 0, out: Optional[np.ndarray] = None) -> Optional[np.ndarray]:
        """Returns the RGBA f32 frame (height, width, 4), or None when the pass is skipped the
        way the reference skips it (missing camera extract, empty buffers)."""
        if camera is None or window is None or level is None:
            return None  # pipeline.rs:88-102
        p = self._p
        if buffers is not None:
            try:
                self.write_buffers(buffers)
            except BrtError as e:
                if e.code == -6:  # BRT_ERR_EMPTY_SCENE: no binding -> skip (pipeline.rs:141-151)
                    return None
                raise
        if out is None:
            out = np.empty((height, width, 4), np.float32)
        assert out.shape == (height, width, 4) and out.dtype == np.float32 and out.flags["C_CONTIGUOUS"]
        stats = BrtStats()
        rr = None if raster_rgba is None else np.ascontiguousarray(raster_rgba, np.float32)
        rd = None if raster_depth is None else np.ascontiguousarray(raster_depth, np.float32)
        if rr is not None:
            assert rr.shape == (height, width, 4)
        if rd is not None:
            assert rd.shape == (height, width)
        _lib.check(p._lib.brt_render(p._ctx, camera.ctypes.data, window.ctypes.data, int(level["level"][0]), width, height,
                                     None if rr is None else rr.ctypes.data, None if rd is None else rd.ctypes.data,
                                     out.ctypes.data, flags, C.byref(stats)), p._ctx)
        self.last_stats = stats.as_dict()
        return out

    # -- device-pointer entry points (used by bench.py with torch tensors) ------------------------

    def render_part_device(self, level, camera, window, width: int, height: int, part: int, n_parts: int,
                           d_out_tile: int, d_raster_rgba: int = 0, d_raster_depth: int = 0, stream: Optional[int] = None,
                           flags: int = 0) -> dict:
        """stream=None: the context's own stream, synchronous, full stats.  stream=<hipStream_t handle>
        (0 = the default stream): asynchronous on that stream (FLAG_CALLER_STREAM is added)."""
        p = self._p
        stats = BrtStats()
        if stream is not None:
            flags |= FLAG_CALLER_STREAM
        _lib.check(p._lib.brt_render_part_device(p._ctx, camera.ctypes.data, window.ctypes.data, int(level["level"][0]),
                                                 width, height, part, n_parts, d_raster_rgba or None,
                                                 d_raster_depth or None, d_out_tile, stream or None, flags,
                                                 C.byref(stats)), p._ctx)
        return stats.as_dict()

    def render_device(self, level, camera, window, width: int, height: int, d_frame: int, d_raster_rgba: int = 0,
                      d_raster_depth: int = 0, stream: Optional[int] = None, flags: int = 0) -> dict:
        """brt_render_device: the frame of an N-device context assembled on its first device (tiles by peer copy, then
        the de-interleave kernel).  d_frame / d_raster_*: device pointers on the first device.  Stream rule as for
        render_part_device."""
        p = self._p
        stats = BrtStats()
        if stream is not None:
            flags |= FLAG_CALLER_STREAM
        _lib.check(p._lib.brt_render_device(p._ctx, camera.ctypes.data, window.ctypes.data, int(level["level"][0]), width, height,
                                            d_raster_rgba or None, d_raster_depth or None, d_frame, stream or None, flags,
                                            C.byref(stats)), p._ctx)
        self.last_stats = stats.as_dict()
        return self.last_stats

    def deinterleave_device(self, d_tiles: int, n_parts: int, width: int, height: int, d_frame: int,
                            stream: Optional[int] = None, out_format: int = FLAG_OUT_RGBA32F):
        """stream=None: own stream, synchronous.  stream=<handle> (0 = default stream): asynchronous there --
        pass the stream the gather was enqueued on so that the copy kernel runs behind it.  out_format: FLAG_OUT_*."""
        p = self._p
        _lib.check(p._lib.brt_deinterleave_device(p._ctx, d_tiles, n_parts, width, height, d_frame, stream or None,
                                                  (0 if stream is None else FLAG_CALLER_STREAM) | out_format), p._ctx)

    def gather_rccl(self, comm: int, rank: int, world: int, d_tile: int, d_tiles_on_root: int, width: int, height: int,
                    d_frame_on_root: int = 0, stream: Optional[int] = None, out_format: int = FLAG_OUT_RGBA32F):
        """brt_gather_rccl: ONE ncclGather of every rank's tile to rank 0 and, there, the de-interleave kernel behind it on the
        same stream.  Stream rule as for render_part_device."""
        p = self._p
        _lib.check(p._lib.brt_gather_rccl(p._ctx, comm, rank, world, d_tile, d_tiles_on_root or None, width, height,
                                          d_frame_on_root or None, stream or None,
                                          (0 if stream is None else FLAG_CALLER_STREAM) | out_format), p._ctx)
