//! UNVERIFIED SOURCE: written against `include/bevyray_amd.h` (BRT_ABI_VERSION 4), never compiled (no Rust
//! toolchain in the build environment).  What IS checked mechanically: `tests/test_abi_binding.py` parses this file and the C
//! header and compares every export (name, argument count, argument and return types), every constant, the field order and
//! types of `brt_stats` and the ABI version.  What is compiled and tested against the same ABI: the ctypes binding
//! `bevyray_amd/_lib.py` (every GPU test goes through it) and the C++ host `bevyray_amd/host/raytracing.hpp`.
//!
//! One declaration per export of the header; the doc comment of each names the reference code it replaces.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_void};

pub const BRT_ABI_VERSION: u32 = 4;
pub const BRT_STRIP_ROWS: u32 = 8;

pub const BRT_OK: i32 = 0;
pub const BRT_ERR_INVALID_ARGUMENT: i32 = -1;
pub const BRT_ERR_NO_DEVICE: i32 = -2;
pub const BRT_ERR_HIP: i32 = -3;
pub const BRT_ERR_INVALID_BVH: i32 = -4;
pub const BRT_ERR_INVALID_SCENE: i32 = -5;
pub const BRT_ERR_EMPTY_SCENE: i32 = -6;
pub const BRT_ERR_NO_SCENE: i32 = -7;
pub const BRT_ERR_UNSUPPORTED: i32 = -8;
pub const BRT_ERR_CAPACITY: i32 = -9;
pub const BRT_ERR_RCCL: i32 = -10;

/// Raytracing level, reference src/raytracing/mod.rs:94-101 (#[repr(u32)])
pub const BRT_LEVEL_SKIP: u32 = 0;
pub const BRT_LEVEL_FALLBACK_RASTER: u32 = 1;
pub const BRT_LEVEL_FALLBACK_RAYTRACED: u32 = 2;
pub const BRT_LEVEL_PURE: u32 = 3;

/// brt_render* flags
pub const BRT_FLAG_COUNTERS: u32 = 1;
pub const BRT_FLAG_KERNEL_SIMPLE: u32 = 2;
pub const BRT_FLAG_CALLER_STREAM: u32 = 4;

/// brt_set_policy: the WGSL-spec (short-circuit) reading of `||` in raytrace.wgsl:269; default 0 = both operands evaluated
pub const BRT_POLICY_OR_SHORT_CIRCUIT: u32 = 1;

/// brt_import_frame_fd handle types
pub const BRT_EXTMEM_OPAQUE_FD: u32 = 1;
pub const BRT_EXTMEM_DMABUF_FD: u32 = 2;

/// brt_scene_generate kinds
pub const BRT_SCENE_COVER: u32 = 0;
pub const BRT_SCENE_RTIOW_FINAL: u32 = 1;
pub const BRT_SCENE_STRESS_GRID: u32 = 2;

/// brt_debug_eval op codes
pub const BRT_DBG_MINMAX: u32 = 0;
pub const BRT_DBG_SQRT_DIV: u32 = 1;
pub const BRT_DBG_RNG: u32 = 2;
pub const BRT_DBG_SLAB: u32 = 3;
pub const BRT_DBG_SPHERE: u32 = 4;
pub const BRT_DBG_SEED: u32 = 5;
pub const BRT_DBG_DIV: u32 = 6;
pub const BRT_DBG_DIV_SWEEP: u32 = 7;
pub const BRT_DBG_SQRT_SWEEP: u32 = 8;

#[repr(C)]
pub struct brt_ctx { _private: [u8; 0] }

#[repr(C)]
#[derive(Default, Debug, Clone, Copy)]
pub struct brt_stats {
    pub rays: u64,
    pub node_pops: u64,
    pub interior_visits: u64,
    pub sphere_tests: u64,
    pub hits: u64,
    pub paths: u64,
    pub kernel_ms: f64,
    pub gather_ms: f64,
    pub total_ms: f64,
    pub lds_bytes: u32,
    pub scene_in_lds: u32,
    pub n_workgroups: u32,
    pub threads_per_workgroup: u32,
    pub prepass_ms: f64,
    pub kernel_variant: u32,
    pub measured_tile_costs: u32,
}

extern "C" {
    pub fn brt_abi_version() -> u32;
    pub fn brt_last_error(ctx: *const brt_ctx) -> *const c_char;
    /// replaces RaytracingPipeline::from_world (pipeline.rs:233-331)
    pub fn brt_create(device_ids: *const i32, n_devices: i32, out_ctx: *mut *mut brt_ctx) -> i32;
    pub fn brt_destroy(ctx: *mut brt_ctx) -> i32;
    pub fn brt_set_policy(ctx: *mut brt_ctx, flags: u32) -> i32;
    pub fn brt_set_tuning(ctx: *mut brt_ctx, name: *const c_char, value: u32) -> i32;
    pub fn brt_get_tuning(ctx: *const brt_ctx, name: *const c_char, out_value: *mut u32, out_default: *mut u32) -> i32;
    /// replaces model_buffer / material_buffer / bvh_buffer .write_buffer (pipeline.rs:136-138); bvh_nodes null: the callee builds the tree on the GPU (extract.rs:315-332 is then not needed)
    pub fn brt_upload_scene(ctx: *mut brt_ctx, models: *const c_void, n_models: u32, materials: *const c_void,
                            n_materials: u32, bvh_nodes: *const c_void, n_nodes: u32) -> i32;
    /// replaces set_bind_group x2 + draw(0..3, 0..1) (pipeline.rs:160-217) and the fragment() grid (raytrace.wgsl:93-123)
    pub fn brt_render(ctx: *mut brt_ctx, camera80: *const c_void, window16: *const c_void, level: u32, width: u32,
                      height: u32, raster_rgba: *const f32, raster_depth: *const f32, out_rgba: *mut f32,
                      flags: u32, stats_or_null: *mut brt_stats) -> i32;
    pub fn brt_host_alloc(ctx: *mut brt_ctx, bytes: u64, out_ptr: *mut *mut c_void) -> i32;
    pub fn brt_host_free(ctx: *mut brt_ctx, ptr: *mut c_void) -> i32;
    /// one rank's strips into a device tile (row tiling over N GPUs)
    pub fn brt_render_part_device(ctx: *mut brt_ctx, camera80: *const c_void, window16: *const c_void, level: u32,
                                  width: u32, height: u32, part: u32, n_parts: u32, d_raster_rgba: *const f32,
                                  d_raster_depth: *const f32, d_out_tile: *mut f32, hip_stream: *mut c_void,
                                  flags: u32, stats_or_null: *mut brt_stats) -> i32;
    /// the whole frame of an N-device context, assembled on its first device (peer copies over xGMI + one copy kernel); replaces the pass on post_process.destination (pipeline.rs:191-217) for a single-process node
    pub fn brt_render_device(ctx: *mut brt_ctx, camera80: *const c_void, window16: *const c_void, level: u32,
                             width: u32, height: u32, d_raster_rgba: *const f32, d_raster_depth: *const f32,
                             d_frame: *mut f32, hip_stream: *mut c_void, flags: u32, stats_or_null: *mut brt_stats) -> i32;
    pub fn brt_tile_rows(height: u32, n_parts: u32) -> u32;
    pub fn brt_deinterleave_device(ctx: *mut brt_ctx, d_tiles: *const f32, n_parts: u32, width: u32, height: u32,
                                   d_frame: *mut f32, hip_stream: *mut c_void, flags: u32) -> i32;
    /// ncclGetUniqueId through the library's own librccl (dlopen): on one rank, the 128 bytes go to the others by the host's means
    pub fn brt_rccl_unique_id(out_id128: *mut c_void) -> i32;
    /// ncclCommInitRank on the context's first device; every rank calls it
    pub fn brt_rccl_comm_create(ctx: *mut brt_ctx, id128: *const c_void, rank: i32, world: i32,
                                out_comm: *mut *mut c_void) -> i32;
    pub fn brt_rccl_comm_destroy(ctx: *mut brt_ctx, comm: *mut c_void) -> i32;
    /// the ONE collective of the path for a host that runs one process per GPU: ncclGather of the tiles to rank 0 (rccl.h:745) + the de-interleave kernel
    pub fn brt_gather_rccl(ctx: *mut brt_ctx, nccl_comm: *mut c_void, rank: i32, world: i32, d_tile: *const f32,
                           d_tiles_on_root: *mut f32, width: u32, height: u32, d_frame_on_root: *mut f32,
                           hip_stream: *mut c_void, flags: u32) -> i32;
    /// maps the memory behind the host's colour target (a Vulkan opaque-fd export, or a dma-buf) so that the frame is written where the next pass reads it (pipeline.rs:191-203)
    pub fn brt_import_frame_fd(ctx: *mut brt_ctx, fd: i32, bytes: u64, handle_type: u32, out_d_frame: *mut *mut f32) -> i32;
    pub fn brt_release_frame(ctx: *mut brt_ctx, d_frame: *mut f32) -> i32;
    pub fn brt_debug_export_frame_fd(ctx: *mut brt_ctx, bytes: u64, out_fd: *mut i32, out_d_ptr: *mut *mut f32) -> i32;
    pub fn brt_debug_copy_to_host(ctx: *mut brt_ctx, d_src: *const c_void, h_dst: *mut c_void, bytes: u64) -> i32;
    pub fn brt_debug_eval(ctx: *mut brt_ctx, op: u32, in16: *const f32, out8: *mut f32, n: u32) -> i32;
    pub fn brt_debug_profile(ctx: *mut brt_ctx, out64: *mut u64) -> i32;
    pub fn brt_debug_tile_order(ctx: *mut brt_ctx, ray_sum: *const u32, longest_pixel: *const u32, n_tiles: u32,
                                sample_count: u32, grid_lanes: u64, out_order: *mut u32, out_info2: *mut u32) -> i32;
    pub fn brt_host_tile_order(ray_sum: *const u32, longest_pixel: *const u32, n_tiles: u32, sample_count: u32,
                               grid_lanes: u64, sorted: u32, lane_permille: u32, out_order: *mut u32,
                               out_info3: *mut u32) -> i32;
    /// replaces obvhs::ploc::build_ploc::<24> + flatten (extract.rs:315-332)
    pub fn brt_build_bvh(models: *const c_void, n_models: u32, out_nodes: *mut c_void, capacity: u32,
                         out_n_nodes: *mut u32) -> i32;
    /// binned-SAH tree for the same node contract, CPU statement of what brt_upload_scene builds on the GPU when bvh_nodes is null
    pub fn brt_build_bvh_sah(models: *const c_void, n_models: u32, out_nodes: *mut c_void, capacity: u32,
                             out_n_nodes: *mut u32) -> i32;
    pub fn brt_build_bvh_device(ctx: *mut brt_ctx, models: *const c_void, n_models: u32, out_nodes: *mut c_void,
                                capacity: u32, out_n_nodes: *mut u32, out_build_ms: *mut f64) -> i32;
    /// the same tree built on the GPU (byte-identical), with the kernel time
    pub fn brt_build_bvh_sah_device(ctx: *mut brt_ctx, models: *const c_void, n_models: u32,
                                    out_nodes: *mut c_void, capacity: u32, out_n_nodes: *mut u32,
                                    out_build_ms: *mut f64) -> i32;
    pub fn brt_validate_scene(models: *const c_void, n_models: u32, materials: *const c_void, n_materials: u32,
                              bvh_nodes: *const c_void, n_nodes: u32, out_max_depth: *mut u32) -> i32;
    pub fn brt_scene_generate(kind: u32, seed: u64, out_models: *mut c_void, out_materials: *mut c_void,
                              capacity: u32, out_n_models: *mut u32) -> i32;
    pub fn brt_host_camera_extract(translation3: *const f32, target3: *const f32, up3: *const f32, fov: f32,
                                   aspect_ratio: f32, near_: f32, far_: f32, sample_count: u32, bounces: u32,
                                   out_camera80: *mut c_void) -> i32;
    pub fn brt_host_window_extract(random_seed: f32, physical_height: u32, out_window16: *mut c_void) -> i32;
    pub fn brt_host_material(base_color_srgb3: *const f32, metallic: f32, perceptual_roughness: f32,
                             reflectance: f32, ior: f32, specular_transmission: f32, out_material32: *mut c_void) -> i32;
}

/// Text of the last error as an owned String (ctx may be null).
pub unsafe fn last_error(ctx: *const brt_ctx) -> String {
    let p = brt_last_error(ctx);
    if p.is_null() { String::new() } else { std::ffi::CStr::from_ptr(p).to_string_lossy().into_owned() }
}
