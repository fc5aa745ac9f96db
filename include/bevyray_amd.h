/*
 * bevyray_amd.h -- C ABI of the MI355X path-tracing render node.
 *
 * This is the drop-in boundary for ONE path of GrandmasterB42/bevyray: the wgpu
 * fragment pass that `RayTracingNode::run` encodes
 * (reference src/raytracing/pipeline.rs:58-220) together with the per-pixel ray
 * loop it dispatches (reference assets/shaders/raytrace.wgsl:93-421,
 * random.wgsl:3-30, const.wgsl:1-2).  The Rust node keeps its registration,
 * ViewQuery and extract stage; the body of `run` calls these functions instead
 * of `write_buffer` x3 + bind groups + `draw(0..3, 0..1)`.  INTEGRATION.md shows
 * the Rust `extern "C"` block and the replacement body.
 *
 * Conventions
 *   - Every function returns int32: BRT_OK (0) or a negative BRT_ERR_* code; none
 *     throws or aborts across the boundary (every export body runs inside an exception
 *     barrier: a failed host allocation is BRT_ERR_OUT_OF_MEMORY, anything else
 *     BRT_ERR_INTERNAL).  brt_last_error() gives the text.
 *     (Reference: every "not ready" condition returns Ok(()) and skips the pass,
 *     pipeline.rs:82-85,89-102,113-115,141-151; the Rust side maps non-zero to a
 *     logged warning + Ok(()).)
 *   - The caller owns every pointer it passes; the callee copies during the call
 *     and retains nothing.  All device memory lives in the opaque context.
 *   - A context is single-caller (not re-entrant), like a render-graph node that
 *     the runner executes sequentially.
 *   - Wire formats are the encase/WGSL layouts the reference's extract stage
 *     already produces (extract.rs:56-61, 83-104, 181-189, 213-218, 229-237):
 *       Model            32 B  position vec3 @0, radius f32 @12, material_id u32 @16
 *       RaytraceMaterial 32 B  base_color vec3 @0, metallic @12, roughness @16,
 *                              reflectance @20, ior @24, specular_transmission @28
 *       BVHNode          48 B  bounds_min vec3 @0, bounds_max vec3 @16, index u32 @28,
 *                              model_count u32 @32
 *       CameraExtract    80 B  sample_count u32 @0, bounce_count @4, projection @8,
 *                              near f32 @12, far @16, fov @20, aspect @24,
 *                              position vec3 @32, direction vec3 @48, up vec3 @64
 *       WindowExtract    16 B  random_seed f32 @0, height u32 @4
 *   - Frames are RGBA f32, row-major, top row first, width*height*16 bytes (device frames optionally in the colour
 *     target's own format: BRT_FLAG_OUT_*).
 */
#ifndef BEVYRAY_AMD_H
#define BEVYRAY_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BRT_ABI_VERSION 6u

/* Rows per strip of the interleaved row tiling. */
#define BRT_STRIP_ROWS 8u

enum {
    BRT_OK = 0,
    BRT_ERR_INVALID_ARGUMENT = -1,
    BRT_ERR_NO_DEVICE = -2,        /* no HIP device / HIP runtime failure at create */
    BRT_ERR_HIP = -3,              /* a HIP call failed; text in brt_last_error */
    BRT_ERR_INVALID_BVH = -4,      /* node index out of range, cycle, leaf range out of range */
    BRT_ERR_INVALID_SCENE = -5,    /* material_id out of range (non-finite spheres are accepted, as in the reference) */
    BRT_ERR_EMPTY_SCENE = -6,      /* zero spheres: the reference skips the pass (pipeline.rs:141-151) */
    BRT_ERR_NO_SCENE = -7,         /* render before upload */
    BRT_ERR_UNSUPPORTED = -8,      /* e.g. orthographic projection (extract.rs:148) */
    BRT_ERR_CAPACITY = -9,         /* caller buffer too small */
    BRT_ERR_RCCL = -10,            /* librccl could not be loaded, or an RCCL call failed; text in brt_last_error */
    BRT_ERR_OUT_OF_MEMORY = -11,   /* a host allocation failed (std::bad_alloc caught at the boundary); the context stays usable */
    BRT_ERR_INTERNAL = -12         /* any other exception caught at the boundary; text in brt_last_error */
};

/* Raytracing level, reference src/raytracing/mod.rs:94-101 (#[repr(u32)]). */
enum {
    BRT_LEVEL_SKIP = 0,
    BRT_LEVEL_FALLBACK_RASTER = 1,
    BRT_LEVEL_FALLBACK_RAYTRACED = 2,
    BRT_LEVEL_PURE = 3
};

/* brt_render* flags */
enum {
    BRT_FLAG_COUNTERS = 1u,        /* also count node pops / interior visits / sphere tests / hits */
    BRT_FLAG_KERNEL_SIMPLE = 2u,   /* one-thread-per-pixel bring-up kernel instead of the persistent one */
    BRT_FLAG_CALLER_STREAM = 4u,   /* device entry points: `hip_stream` is the caller's stream even when it is NULL
                                      (NULL is then the legacy default stream, not "the context's own stream"):
                                      the work is enqueued there, ordered with whatever the caller enqueued before
                                      (e.g. an RCCL gather), and the call does not synchronise */
    /* Format of the assembled DEVICE frame (brt_render_device, brt_gather_rccl, brt_deinterleave_device): the reference's pass writes
     * into post_process.destination, whose format is TextureFormat::bevy_default() (pipeline.rs:311-315) -- an 8-bit sRGB target, or
     * Rgba16Float under HDR -- so with brt_import_frame_fd the frame can be stored where the next pass reads it, in that format.
     * The pass's result IS the RGBA f32 frame (parity is stated on it); the other formats are its store conversion, exactly:
     *   RGBA8_UNORM_SRGB  r, g, b: round(255 * OETF(clamp(c, 0, 1))), OETF the sRGB encode (12.92 c below 0.0031308, else
     *                     1.055 c^(1/2.4) - 0.055), evaluated exactly (as the number of decision thresholds <= c, see
     *                     bevyray_amd/csrc/brt_srgb_table.h); alpha: the RGBA8_UNORM rule.  NaN stores 0.  4 bytes per pixel, r first.
     *   RGBA8_UNORM       round-half-even(255 * clamp(c, 0, 1)), exact.  4 bytes per pixel.
     *   RGBA16F           f32 -> f16, round to nearest even.  8 bytes per pixel.
     * brt_render / brt_render_part_device always write RGBA f32 (host frame / a rank's tile: the tiles travel as f32). */
    BRT_FLAG_OUT_RGBA32F = 0u,
    BRT_FLAG_OUT_RGBA8_UNORM_SRGB = 8u,
    BRT_FLAG_OUT_RGBA16F = 16u,
    BRT_FLAG_OUT_RGBA8_UNORM = 24u,
    BRT_FLAG_OUT_MASK = 24u
};

typedef struct brt_ctx brt_ctx;

/* Per-call statistics.  `rays` (one per raycast() call, raytrace.wgsl:190) is always
 * counted; the other four only with BRT_FLAG_COUNTERS (else 0). */
typedef struct brt_stats {
    uint64_t rays;
    uint64_t node_pops;        /* raytrace.wgsl:321-323 */
    uint64_t interior_visits;  /* raytrace.wgsl:327-341 */
    uint64_t sphere_tests;     /* raytrace.wgsl:350-352 */
    uint64_t hits;             /* scatter() calls, raytrace.wgsl:204 */
    uint64_t paths;            /* pixels * sample_count rendered by this call */
    double   kernel_ms;        /* HIP-event time of the trace kernel on its own stream (max over devices) */
    double   gather_ms;        /* tile copy-out / gather time */
    double   total_ms;         /* host wall time of the call */
    uint32_t lds_bytes;        /* dynamic LDS per workgroup of the trace kernel */
    uint32_t scene_in_lds;     /* 1: BVH + spheres LDS-resident; 2: the top levels of the BVH in LDS, the rest from L2; 0: all from L2 */
    uint32_t n_workgroups;
    uint32_t threads_per_workgroup;
    double   prepass_ms;       /* kernel time of the dispatch-order pre-pass that ran before this frame (first frame
                                  of a view on the synchronous paths; else 0).  Not part of kernel_ms. */
    uint32_t kernel_variant;   /* instantiation of the trace kernel that ran: 0 general, 1 / 2 the LEAN ones (Pure level; 2: no
                                  pixel chain can be critical), + 16: knobs live (a tuning knob off its default) */
    uint32_t measured_tile_costs; /* 1: this frame measured the per-tile ray counts for the dispatch order of the next ones
                                  (first frames of a view, every frame while the camera moves, after scene uploads) */
    uint32_t tree_rebuilt;     /* 1: the callee-built tree was rebuilt for this call's camera before the launch (see brt_upload_scene) */
    float    tree_reach;       /* the `reach` the resident callee-built SAH tree was built with (what brt_build_bvh_sah takes: 0 = the
                                  scene's own extent); 0 for a caller's tree */
    uint64_t forwarded_bytes;  /* brt_render_device: bytes of the raster inputs forwarded to the other devices of the context */
    uint32_t hot_records;      /* a scene walked from an LDS tile + L2 (scene_in_lds == 2): the tree's records are numbered by how often this
                                  view visits them (measured by the pre-pass of a first frame), so that the tile holds the ones the walk
                                  uses; this many of them were visited at all.  0: breadth-first numbering (no pre-pass yet / another mode) */
    uint32_t reserved;
} brt_stats;

uint32_t brt_abi_version(void);

/* Text of the last error on this context (ctx may be NULL: last error of brt_create /
 * the host-only helpers on this thread).  The pointer stays valid until the next call. */
const char* brt_last_error(const brt_ctx* ctx);

/* Replaces: RaytracingPipeline::from_world (pipeline.rs:233-331) -- one-time GPU setup.
 * device_ids[n_devices] are HIP ordinals; the frame is row-tiled over them in strips of
 * BRT_STRIP_ROWS rows (strip s -> device s % n_devices).  An ordinal may repeat.
 * Threads: all state lives in the context; calls on ONE context must not overlap, different contexts may be driven from different
 * host threads at the same time (tests/test_parity_gpu.py::test_two_contexts_driven_from_two_host_threads). */
int32_t brt_create(const int32_t* device_ids, int32_t n_devices, brt_ctx** out_ctx);
int32_t brt_destroy(brt_ctx* ctx);

/* The readings of the shader that WGSL leaves to the implementation and that nothing in the reference pins (naga /
 * the driver decide them; no wgpu run is possible where this library was built).  They change pixels, so they are an API option,
 * not environment variables.  Default (flags = 0) is what the parity tests are stated on; each flag switches ONE reading to its
 * alternative (the frames run in the knobs-live instantiation of the kernel then), so that the day a real wgpu frame can be compared
 * (scripts/compare_wgpu_frame.py names the combination it matches) the product can follow it:
 *   BRT_POLICY_OR_SHORT_CIRCUIT  `if cannot_refract || reflectance(cos_theta, ri) > rngNextFloat(state)` (raytrace.wgsl:269): default =
 *                                both operands evaluated, the RNG draw always happens; flag = the WGSL-spec reading, no draw when
 *                                cannot_refract.  Differs only for glass with ior < 1.
 *   BRT_POLICY_MINMAX_SELECT     min / max (raytrace.wgsl:391-394, :263, :405): default = IEEE minNum / maxNum (a NaN operand yields the
 *                                other one); flag = compare-select, min(a, b) = b < a ? b : a, max(a, b) = a < b ? b : a.  Differs
 *                                only when a bound or a ray component is a NaN.
 *   BRT_POLICY_POW_EXP2_LOG2     pow(x, 5.0) (raytrace.wgsl:415): default = (x x)(x x) x; flag = exp2(5 log2 x) evaluated in f64 and
 *                                rounded to f32 once.  Differs in the last bits of Schlick's reflectance.
 * Applies to the frames rendered after the call.  The bring-up kernel (BRT_FLAG_KERNEL_SIMPLE) implements the default only. */
enum { BRT_POLICY_OR_SHORT_CIRCUIT = 1u, BRT_POLICY_MINMAX_SELECT = 2u, BRT_POLICY_POW_EXP2_LOG2 = 4u };
int32_t brt_set_policy(brt_ctx* ctx, uint32_t flags);

/* Scheduling / launch-shape knobs of the trace path (names and meaning: DESIGN.md section 6, "Tuning aids").  None
 * changes a pixel.  They live in the context; nothing reads the environment per frame.  brt_create takes initial values
 * from the environment variables of the same names, once, and only when BRT_ENABLE_TUNING=1 is set.  No reference
 * counterpart (the reference has one fullscreen draw and nothing to tune, pipeline.rs:206-217). */
int32_t brt_set_tuning(brt_ctx* ctx, const char* name, uint32_t value);
int32_t brt_get_tuning(const brt_ctx* ctx, const char* name, uint32_t* out_value, uint32_t* out_default);

/* Replaces: model_buffer / material_buffer / bvh_buffer .write_buffer (pipeline.rs:136-138).
 * Takes the three CPU vectors that prepare_buffers builds (extract.rs:299-336), validates
 * them (indices in range, BVH reachable from node 0 without cycles) and copies them to
 * every device of the context.  If bvh_nodes == NULL / n_nodes == 0 the callee builds the
 * BVH itself (see brt_build_bvh_sah / brt_build_bvh_device): recommended, the ray loop runs faster in that tree than in
 * the caller's PLOC tree and the caller saves its own per-frame build (extract.rs:315-332).
 * The callee's tree pads a sphere's box by what the f32 arithmetic of the two intersection tests needs for the distances rays travel
 * (0.01 ... 0.1) instead of the reference's flat 0.1 (Model::aabb, extract.rs:220-227).  Those distances depend on the camera, which
 * an upload does not know: the tree is built for the scene's own extent, and every brt_render* call checks its camera first -- one
 * that is further out than the resident tree covers has the tree rebuilt on the GPU (same scene bytes, larger pads, up to the
 * reference's 0.1; 0.3-1 ms, brt_stats::tree_rebuilt) before its frame is launched.  The rule is brt_host_tree_reach.  A caller's
 * tree is used as it comes. */
int32_t brt_upload_scene(brt_ctx* ctx,
                         const void* models, uint32_t n_models,
                         const void* materials, uint32_t n_materials,
                         const void* bvh_nodes, uint32_t n_nodes);

/* Replaces: set_bind_group x2 + draw(0..3, 0..1) (pipeline.rs:160-217) and the whole
 * fragment() invocation grid (raytrace.wgsl:93-123).  Renders a width x height frame into
 * out_rgba (HOST pointer).  raster_rgba / raster_depth (host, width*height*4 / width*height
 * floats, reverse-Z depth) are the `screen_texture` and depth prepass the shader samples
 * (raytrace.wgsl:98,106,116); either may be NULL = cleared to 0.  Synchronous. */
int32_t brt_render(brt_ctx* ctx, const void* camera80, const void* window16, uint32_t level,
                   uint32_t width, uint32_t height,
                   const float* raster_rgba, const float* raster_depth,
                   float* out_rgba, uint32_t flags, brt_stats* stats_or_null);

/* Optional fast path for brt_render: page-locked host memory owned by the context.  When
 * out_rgba (and/or raster_rgba / raster_depth) lies inside such an allocation the frame is DMA'd
 * straight into it (no staging copy on the CPU: ~2 ms less per 1080p frame).  The memory stays
 * valid until brt_host_free / brt_destroy.  Any other pointer keeps working through staging. */
int32_t brt_host_alloc(brt_ctx* ctx, uint64_t bytes, void** out_ptr);
int32_t brt_host_free(brt_ctx* ctx, void* ptr);

/* Same frame, but only the strips of `part` out of `n_parts` (strip s belongs to part
 * s % n_parts), written densely into a DEVICE tile buffer of brt_tile_rows() rows on the
 * context's first device: tile row (k*BRT_STRIP_ROWS + r) is frame row
 * ((k*n_parts + part)*BRT_STRIP_ROWS + r).  d_raster_* are optional DEVICE full-frame
 * buffers.  `hip_stream` is a hipStream_t.  NULL without BRT_FLAG_CALLER_STREAM = the context's own
 * (non-blocking) stream: the call then SYNCHRONISES before it returns, fills every field of stats and
 * refreshes the dispatch-order history.  A non-NULL stream, or BRT_FLAG_CALLER_STREAM: asynchronous on that
 * stream; only the launch-shape fields and `paths` of stats are filled.
 * One render per context is in flight at a time: the control block (counters, tile queue) is per context, so
 * a call first makes its stream wait (hipStreamWaitEvent) for the previous call's kernel, whichever stream
 * that ran on.  d_out_tile must not be read or overwritten by other streams before this call's work is done. */
int32_t brt_render_part_device(brt_ctx* ctx, const void* camera80, const void* window16, uint32_t level,
                               uint32_t width, uint32_t height, uint32_t part, uint32_t n_parts,
                               const float* d_raster_rgba, const float* d_raster_depth,
                               float* d_out_tile, void* hip_stream, uint32_t flags,
                               brt_stats* stats_or_null);

/* The whole frame of an N-device context (brt_create with N ordinals), assembled ON ITS FIRST DEVICE: every device traces its
 * strips, the tiles of devices 1..N-1 travel to the first device by peer copy (hipMemcpyPeerAsync: xGMI inside a node) and a
 * copy kernel de-interleaves them into d_frame (DEVICE pointer on the first device, width*height*4 floats) -- the single
 * gather of tile buffers at frame end, for a single-process host.  Replaces, together with brt_upload_scene, the pass that
 * RayTracingNode::run encodes on post_process.destination (pipeline.rs:191-217); the caller copies or maps d_frame into its
 * colour target.  d_raster_rgba / d_raster_depth: optional full-frame DEVICE buffers on the first device (the callee forwards
 * them to the other devices).  Stream rule as for brt_render_part_device: NULL without BRT_FLAG_CALLER_STREAM = the context's
 * own stream, synchronous, every field of stats filled (gather_ms = from the end of the first device's trace to the
 * assembled frame); otherwise the de-interleave is enqueued on `hip_stream` (a stream of the first device) behind the
 * tiles' arrival and the call does not synchronise.  One frame per context in flight. */
int32_t brt_render_device(brt_ctx* ctx, const void* camera80, const void* window16, uint32_t level,
                          uint32_t width, uint32_t height,
                          const float* d_raster_rgba, const float* d_raster_depth,
                          void* d_frame, void* hip_stream, uint32_t flags, brt_stats* stats_or_null);

/* Rows in the dense tile of `part` (same for every part: padded to whole strips). */
uint32_t brt_tile_rows(uint32_t height, uint32_t n_parts);

/* Root side of the gather: d_tiles holds n_parts tiles back to back (each
 * brt_tile_rows()*width*4 floats, i.e. the receive buffer of a gather); writes the
 * de-interleaved width x height frame to d_frame.  Stream rule as for brt_render_part_device: NULL without
 * BRT_FLAG_CALLER_STREAM = the context's own stream, synchronous; otherwise asynchronous on `hip_stream` --
 * pass the stream the gather was enqueued on (with BRT_FLAG_CALLER_STREAM if that is the default stream), or
 * the copy kernel is not ordered behind the gather. */
int32_t brt_deinterleave_device(brt_ctx* ctx, const float* d_tiles, uint32_t n_parts,
                                uint32_t width, uint32_t height, void* d_frame, void* hip_stream, uint32_t flags);

/* ---- strips by measured cost (one process per GPU) ------------------------------------------------------------------------------------
 * By default strip s of BRT_STRIP_ROWS rows belongs to part s % n_parts (interleaved, because bands are badly
 * imbalanced).  A STRIP TABLE assigns the strips by their cost instead: part_of_strip[s] = the part that renders frame strip s, a
 * permutation of the parts inside every group of n_parts consecutive strips -- so every part still has exactly one strip per group (its
 * k-th local strip lies in group k): tile size and layout, brt_tile_rows and the ONE gather stay as they are, only the two lookups
 * "local strip -> frame strip" (trace kernel) and "frame strip -> part" (assembly) go through the table.  Pixels cannot change: a
 * pixel's seed depends on its frame coordinates only (raytrace.wgsl:95).
 *   brt_set_strip_table   installs the table for frames of n_strips = ceil(height / 8) strips split n_parts ways (NULL: back to s % n_parts);
 *                         brt_render_part_device, brt_deinterleave_device and brt_gather_rccl of such frames use it; EVERY rank must set
 *                         the same table.  BRT_ERR_INVALID_ARGUMENT if a group holds a part twice.  (brt_render / brt_render_device -- one
 *                         context over N devices -- keep s % n_parts.)
 *   brt_plan_strips       makes a table from measured costs and installs it: the frame of this camera is rendered once at `probe_spp`
 *                         samples per pixel on the context's first device with the per-tile ray counts switched on; per group the dearest
 *                         strip goes to the part with the least so far.  Deterministic: every rank of a job computes the same table from
 *                         the same integers, no second collective.  out_part_of_strip (n_strips words) may be NULL. */
int32_t brt_set_strip_table(brt_ctx* ctx, uint32_t n_parts, uint32_t n_strips, const uint32_t* part_of_strip);
int32_t brt_plan_strips(brt_ctx* ctx, const void* camera80, const void* window16, uint32_t level, uint32_t width, uint32_t height,
                        uint32_t n_parts, uint32_t probe_spp, uint32_t* out_part_of_strip);
/* the assignment rule of brt_plan_strips on given costs (host arithmetic, no device: what the CPU tests hold it to) */
int32_t brt_host_plan_strips(const uint64_t* strip_cost, uint32_t n_strips, uint32_t n_parts, uint32_t* out_part_of_strip);

/* ---- the one collective of the path: one process per GPU, one RCCL gather per frame -----------------
 * For a host that runs one process per GPU (instead of one N-device context, brt_render_device): every rank renders its part
 * with brt_render_part_device, then all ranks call brt_gather_rccl -- ONE ncclGather (rccl.h:745) of the tiles to rank 0, and on
 * rank 0 the de-interleave kernel (brt_deinterleave_device) behind it on the same stream.  librccl is resolved with dlopen at
 * the first of these calls (a copy already in the process is used; a single-GPU host needs none); BRT_ERR_RCCL if that fails.
 *   brt_rccl_unique_id    ncclGetUniqueId: on ONE rank; hand the 128 bytes to the others by whatever the host has (a pipe, MPI, a file)
 *   brt_rccl_comm_create  ncclCommInitRank on the context's first device; every rank calls it (it blocks until all have)
 *   brt_gather_rccl       d_tile: this rank's tile (brt_tile_rows(height, world) x width x 4 floats); d_tiles_on_root: `world` such
 *                         tiles on rank 0 (NULL elsewhere); d_frame_on_root: width x height x 4 floats on rank 0, or NULL to skip
 *                         the de-interleave.  Stream rule as for brt_render_part_device (NULL without BRT_FLAG_CALLER_STREAM = the
 *                         context's own stream, synchronous).  The tile must not be rendered into again before the gather is done.
 * A communicator created elsewhere (ncclComm_t of the host's own RCCL binding) may be passed as `nccl_comm` as well. */
int32_t brt_rccl_unique_id(void* out_id128);
int32_t brt_rccl_comm_create(brt_ctx* ctx, const void* id128, int32_t rank, int32_t world, void** out_comm);
int32_t brt_rccl_comm_destroy(brt_ctx* ctx, void* comm);
int32_t brt_gather_rccl(brt_ctx* ctx, void* nccl_comm, int32_t rank, int32_t world, const float* d_tile, float* d_tiles_on_root,
                        uint32_t width, uint32_t height, void* d_frame_on_root, void* hip_stream, uint32_t flags);

/* ---- a frame target in another API's memory -----------------------------------------------------------
 * Replaces: the pass writing straight into post_process.destination (pipeline.rs:191-203).  The host exports the memory behind
 * its colour target (or a buffer it copies from on the GPU) as a file descriptor; brt_import_frame_fd maps it on the context's
 * first device and returns a device pointer that brt_render_device / brt_render_part_device / brt_gather_rccl accept as
 * d_frame.  handle_type: BRT_EXTMEM_OPAQUE_FD = a Vulkan allocation exported with
 * VK_EXTERNAL_MEMORY_HANDLE_TYPE_OPAQUE_FD_BIT (hipImportExternalMemory; the runtime owns the descriptor on success);
 * BRT_EXTMEM_DMABUF_FD = a dma-buf of a HIP virtual-memory allocation (hipMemImportFromShareableHandle; the caller keeps and
 * closes its descriptor).  brt_release_frame unmaps (after the context's pending work); brt_destroy releases what is left. */
enum { BRT_EXTMEM_OPAQUE_FD = 1, BRT_EXTMEM_DMABUF_FD = 2 };
int32_t brt_import_frame_fd(brt_ctx* ctx, int32_t fd, uint64_t bytes, uint32_t handle_type, void** out_d_frame);
int32_t brt_release_frame(brt_ctx* ctx, void* d_frame);
/* Diagnostic (tests): allocates `bytes` of exportable device memory on the first device (hipMemCreate), maps it and exports it
 * as a dma-buf descriptor -- the other side of brt_import_frame_fd when no Vulkan is at hand.  Release with brt_release_frame;
 * the caller closes the descriptor. */
int32_t brt_debug_export_frame_fd(brt_ctx* ctx, uint64_t bytes, int32_t* out_fd, void** out_d_ptr);
/* Diagnostic: hipMemcpy device -> host on the context's first device (for pointers that are no tensor of the caller's). */
int32_t brt_debug_copy_to_host(brt_ctx* ctx, const void* d_src, void* h_dst, uint64_t bytes);

/* Diagnostic: evaluates one device function of the ray loop on n inputs (16 floats in,
 * 8 floats out per element; op codes BRT_DBG_* below) so that tests can compare single
 * reference functions (rngNextFloat, ray_bounding_dst, hit_sphere, the seed formula,
 * min/max/sqrt/divide) bit for bit.  Host pointers, synchronous. */
enum {
    BRT_DBG_MINMAX = 0,    /* in: a, b            out: min(a,b), max(a,b) */
    BRT_DBG_SQRT_DIV = 1,  /* in: a, b            out: sqrt(a), a / b */
    BRT_DBG_RNG = 2,       /* in: state (bits)    out: float, state', ball.xyz, state'' (bits) */
    BRT_DBG_SLAB = 3,      /* in: o3 d3 bmin3 bmax3 closest   out: child pushed (0/1) */
    BRT_DBG_SPHERE = 4,    /* in: o3 d3 center3 radius        out: accepted t or INF */
    BRT_DBG_SEED = 5,      /* in: seed px py W H (floats)     out: rng seed (bits) */
    BRT_DBG_DIV = 6,       /* in: n, d            out: n / d, shared-reciprocal short form, 1 / d, its short form (brt_device.h) */
    BRT_DBG_DIV_SWEEP = 7, /* in: seed (bits), count          out: mismatches of the short forms over `count` random plain-range pairs,
                              bits of the first mismatching n and d */
    BRT_DBG_SQRT_SWEEP = 8,/* in: first (bits), count         out: mismatches of the short sqrt over `count` consecutive floats per element
                              (element i starts at first + i * count; element 0 also checks +0 and -0), bits of the first mismatching argument */
    BRT_DBG_ENCODE = 9     /* in: c               out (as floats holding integers): the 8-bit sRGB code, the 8-bit unorm code, the f16 bits of c
                              (the store conversions of BRT_FLAG_OUT_*) */
};
int32_t brt_debug_eval(brt_ctx* ctx, uint32_t op, const float* in16, float* out8, uint32_t n);

/* Diagnostic: the 64 raw control words of the last launch on the context's first device: out64[0..4]
 * = the brt_stats counters; after a BRT_FLAG_COUNTERS launch out64[8+2k], out64[9+2k] = how
 * often the waves executed code section k and the sum of active lanes over those executions
 * (k: 0 interior step, 1 leaf step, 2 camera ray made where a sample ended, 3 scatter, 4 sky, 5 rejection-sampler iteration,
 * 6 camera ray made at the top of a round (first sample of a pixel, late sample ends), 7 ray round); wave timeline in 100 MHz ticks: [24] ~first start, [25] ~first / [26] last "lane queue empty",
 * [27] last end, [28] sum of (end - empty) over waves, [29] waves, [30]/[31] live lanes and rounds after "empty";
 * wave time summed over waves: [5] pixel refill, [6] walk loop, [7] shading, [43] drain logic + camera ray +
 * walk begin, [44] rejection-sampler loop; [40] critical tiles and [41] longest pixel (rays) of the view's last measured frame, [42] tiles its
 * order hands out as two half-sample jobs, [62] / [63] pixels of the last launch whose second-half lane took the first half's state over /
 * left the pixel to the first-half lane (whose state was not there yet).
 * out64 must hold 64 words. */
int32_t brt_debug_profile(brt_ctx* ctx, uint64_t* out64);

/* Diagnostic: the dispatch order as the GPU builds it (bevyray_amd/csrc/brt_order.hip, used behind every measuring
 * frame with default settings) for given per-tile ray counts: out_order as brt_host_tile_order's (n_tiles + split_tail words),
 * out_info4 = {critical tiles at the front, longest pixel, non-sky tiles, tiles handed out as two half-sample jobs}.  Tests
 * compare it with brt_host_tile_order.  Host pointers, synchronous. */
int32_t brt_debug_tile_order(brt_ctx* ctx, const uint32_t* ray_sum, const uint32_t* longest_pixel, uint32_t n_tiles,
                             uint32_t sample_count, uint64_t grid_lanes, uint32_t tiles_x, uint32_t dilate, uint32_t split_tail,
                             uint32_t* out_order, uint32_t* out_info4);

/* ---- host-only helpers (no GPU needed) --------------------------------------------- */

/* The dispatch order brt_render derives from one frame's per-tile ray counts (sum and longest pixel of each
 * 8x8 tile): out_order[k] = k-th tile to hand out -- the non-sky tiles (longest pixel
 * first when `sorted`), then the one-ray-per-sample "sky" tiles; out_info5[0..2] = {tiles at the front that go to the
 * lane queue (pixel by pixel to single lanes; the rest are handed out as whole tiles), critical tiles at the
 * front, longest pixel}.  grid_lanes = CUs x threads per workgroup.  dilate > 0: a tile is ranked by the longest pixel of the
 * (2 dilate + 1)^2 tiles around it in the tiles_x-wide tile grid, and is "sky" only if all of them were (what brt_render does by
 * default with radius 2, and with the radius the motion covers when the camera has moved since the costs were measured).
 * split_tail > 0: the last min(split_tail, non-sky tiles) non-sky tiles are in the order TWICE -- [other non-sky tiles | those, first
 * half of the samples | the same, second half | sky tiles] -- so that the jobs ahead of the cheap sky tiles are half as long and the
 * end of a launch is balanced (brt_render does this for frames of at least 6 tiles per wave slot); out_order then holds
 * n_tiles + that many words (room for n_tiles + split_tail), out_info5 = {lane-queue tiles, critical tiles, longest pixel,
 * non-sky tiles, tiles that are split}.  No reference counterpart: the reference draws one fullscreen triangle
 * (pipeline.rs:206-215). */
int32_t brt_host_tile_order(const uint32_t* ray_sum, const uint32_t* longest_pixel, uint32_t n_tiles, uint32_t sample_count,
                            uint64_t grid_lanes, uint32_t sorted, uint32_t lane_permille, uint32_t tiles_x, uint32_t dilate,
                            uint32_t split_tail, uint32_t* out_order, uint32_t* out_info5);

/* Replaces: obvhs::ploc::build_ploc::<24>(aabbs, identity, SortPrecision::U64, 0) and the
 * flatten into BVHNode (extract.rs:315-332), including Model::aabb's 0.1 pad
 * (extract.rs:220-227).  Writes up to `capacity` 48-byte nodes; contract: node 0 is the
 * root, an interior node's children are `index` and `index+1`, leaf iff model_count > 0 and
 * then `index` addresses the model buffer directly.  2*n_models-1 nodes for n_models >= 1. */
int32_t brt_build_bvh(const void* models, uint32_t n_models,
                      void* out_nodes, uint32_t capacity, uint32_t* out_n_nodes);

/* A better tree for the same contract: top-down binned SAH (16 bins, single-sphere leaves, depth capped below the
 * shader's 32-entry stack).  The reference rebuilds PLOC on the CPU every frame (extract.rs:315-321, "TODO" at
 * extract.rs:264-267); the shader only needs the node contract above, and an SAH tree costs the ray loop fewer node
 * visits (10 004-sphere grid: 23.6 -> 19.1 interior visits per ray).  This is the CPU statement of the tree
 * brt_upload_scene builds ON THE GPU (brt_build_bvh_sah_device, the same bytes) when the caller passes no BVH (up to 65 536
 * spheres; above that, or with the knob BRT_BVH_QUALITY=0: PLOC on the GPU).
 * reach: the longest distance a ray has travelled when it reaches a sphere, camera included (it sizes the leaf pads: brt_upload_scene
 * above); 0 = the scene's own extent (what an upload builds); otherwise what brt_host_tree_reach returns for a camera. */
int32_t brt_build_bvh_sah(const void* models, uint32_t n_models, float reach,
                          void* out_nodes, uint32_t capacity, uint32_t* out_n_nodes);

/* The reach rule of the callee-built SAH tree (host arithmetic; brt_render* applies it before every launch): out_scene_scale = S, the
 * largest |centre|_1 + radius over the scene's ordinary spheres (radius <= 100); the camera needs |position|_1 + S + L (L: the longest
 * tangent from the camera to a sphere of radius > 100, i.e. how far away a primary ray can land on the ground); out_level = 0 when
 * that is within 2 S, else the smallest k with 2 S * 2^(k/4) >= it (at most 80); out_reach = the `reach` of that level for
 * brt_build_bvh_sah (0 at level 0).  The resident tree is rebuilt when a camera needs a higher level than it has, or at least two
 * levels less.  camera80: a CameraExtract.  No reference counterpart (the reference pads by 0.1 whatever the camera). */
int32_t brt_host_tree_reach(const void* models, uint32_t n_models, const void* camera80, float* out_scene_scale, uint32_t* out_level,
                            float* out_reach);

/* The same build on the GPU (PLOC in one workgroup, bevyray_amd/csrc/brt_bvh.hip): takes the
 * host model vector, returns byte-identical nodes to brt_build_bvh plus the kernel time.
 * brt_upload_scene uses it when the caller passes no BVH and the scene has more than 65 536 spheres (or BRT_BVH_QUALITY=0).
 * Needs a context (a GPU). */
int32_t brt_build_bvh_device(brt_ctx* ctx, const void* models, uint32_t n_models,
                             void* out_nodes, uint32_t capacity, uint32_t* out_n_nodes, double* out_build_ms);

/* The binned-SAH tree of brt_build_bvh_sah built on the GPU (bevyray_amd/csrc/brt_sah.hip: one workgroup for the nodes of more
 * than 1024 spheres, then a workgroup per subtree, a wave per node): byte-identical nodes plus the kernel time.  This is what
 * brt_upload_scene runs when the caller passes no BVH (up to 65 536 spheres), so that a scene that changes every frame --
 * the reference rebuilds and re-uploads per frame, extract.rs:299-336 -- costs no host-side build.  Needs a context (a GPU). */
int32_t brt_build_bvh_sah_device(brt_ctx* ctx, const void* models, uint32_t n_models, float reach,
                                 void* out_nodes, uint32_t capacity, uint32_t* out_n_nodes, double* out_build_ms);

/* The 255 decision thresholds of the exact 8-bit sRGB encode (BRT_FLAG_OUT_RGBA8_UNORM_SRGB): out255[k - 1] = the smallest f32 >=
 * EOTF((k - 0.5) / 255); the code of a linear value c is the number of thresholds <= c.  For hosts / tests that want to state the
 * same encode on the CPU. */
int32_t brt_host_srgb_thresholds(float* out255);

/* Checks what brt_upload_scene checks, without a context. */
int32_t brt_validate_scene(const void* models, uint32_t n_models,
                           const void* materials, uint32_t n_materials,
                           const void* bvh_nodes, uint32_t n_nodes, uint32_t* out_max_depth);

/* Seeded, deterministic versions of the demo scene (reference src/main.rs:49-240 uses an
 * unseeded RNG) and of the other BASELINE.json scenes.  Writes 32-byte Models and one
 * 32-byte RaytraceMaterial per model (material_id = index, extract.rs:301-310). */
enum {
    BRT_SCENE_COVER = 0,       /* main.rs:87-239, sRGB colours decoded like extract.rs:201 */
    BRT_SCENE_RTIOW_FINAL = 1, /* "Ray Tracing in One Weekend" final scene, linear albedos */
    BRT_SCENE_STRESS_GRID = 2  /* ground + 100 x 100 grid of r=0.2 spheres + 3 big spheres */
};
int32_t brt_scene_generate(uint32_t kind, uint64_t seed,
                           void* out_models, void* out_materials, uint32_t capacity,
                           uint32_t* out_n_models);

/* Host-side mirror of the reference's extract stage, so that a non-Rust host (the C++ /
 * Python harnesses in this repo) produces the same bytes the Rust plugin would:
 *   brt_host_camera_extract  = CameraExtract::extract_component (extract.rs:118-157) for a
 *                              Transform::from_translation(t).looking_at(target, up)
 *   brt_host_window_extract  = WindowExtract::extract_component (extract.rs:70-80), seed explicit
 *   brt_host_material        = RaytraceMaterial::prepare_asset (extract.rs:196-208) */
int32_t brt_host_camera_extract(const float* translation3, const float* target3, const float* up3,
                                float fov, float aspect_ratio, float near_, float far_,
                                uint32_t sample_count, uint32_t bounces, void* out_camera80);
int32_t brt_host_window_extract(float random_seed, uint32_t physical_height, void* out_window16);
int32_t brt_host_material(const float* base_color_srgb3, float metallic, float perceptual_roughness,
                          float reflectance, float ior, float specular_transmission,
                          void* out_material32);

#ifdef __cplusplus
}
#endif
#endif /* BEVYRAY_AMD_H */
