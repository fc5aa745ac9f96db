"""bevyray_amd -- MI355X-native path-tracing render node behind bevyray's plugin surface.

Only the hot path of GrandmasterB42/bevyray is here: the per-pixel ray loop
(assets/shaders/*.wgsl) as hand-written HIP kernels for gfx950, behind the C ABI declared in
include/bevyray_amd.h, plus the host-side mirror of the plugin types that feed it.
Importing the package builds/loads the HIP library; without it nothing renders.
"""
from . import _lib
from ._lib import BrtError, build, load
from .raytracing import (  # noqa: F401
    BVH_NODE_DTYPE, CAMERA_DTYPE, EXTMEM_DMABUF_FD, EXTMEM_OPAQUE_FD, FLAG_CALLER_STREAM, FLAG_COUNTERS, FLAG_KERNEL_SIMPLE, FLAG_OUT_RGBA16F, FLAG_OUT_RGBA32F, FLAG_OUT_RGBA8_UNORM, FLAG_OUT_RGBA8_UNORM_SRGB, OUT_PIXEL_BYTES, POLICY_MINMAX_SELECT, POLICY_OR_SHORT_CIRCUIT, POLICY_POW_EXP2_LOG2, LEVEL_DTYPE, MATERIAL_DTYPE, MODEL_DTYPE,
    SCENE_COVER, SCENE_RTIOW_FINAL, SCENE_STRESS_GRID, STRIP_ROWS, WINDOW_DTYPE, Buffers, CameraExtract,
    OrthographicProjection, PerspectiveProjection, RaytracedCamera, RaytracedSphere, RaytraceMaterial, RaytracePlugin,
    Raytracing, RayTracingNode, StandardMaterial, Transform, WindowExtract, build_bvh, build_bvh_sah, cover_camera, generate_scene, tree_reach,
    prepare_buffers, rtiow_camera, srgb_thresholds, tile_rows, validate_scene,
)

__all__ = [n for n in dir() if not n.startswith("_")]
