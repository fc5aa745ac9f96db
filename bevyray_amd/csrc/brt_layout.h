// brt_layout.h -- wire formats of the reference's extract stage and the device-side
// scene encoding.  Shared by the host code and the HIP kernels.
//
// Wire structs follow the WGSL declarations (reference assets/shaders/raytrace.wgsl:30-87)
// and their Rust twins (reference src/raytracing/extract.rs:56-61, 83-104, 181-189,
// 213-218, 229-237): vec3 aligns to 16 bytes and has size 12; a struct's size rounds up to
// its alignment.
#pragma once
#include <cstddef>
#include <cstdint>

namespace brt {

struct Model {             // raytrace.wgsl:57-61
    float position[3];
    float radius;
    uint32_t material_id;
    uint32_t _pad[3];
};
struct Material {          // raytrace.wgsl:64-77
    float base_color[3];
    float metallic;
    float roughness;
    float reflectance;     // never read by the shader (raytrace.wgsl:72)
    float ior;
    float specular_transmission;
};
struct BVHNode {           // raytrace.wgsl:80-87
    float bounds_min[3];
    float _pad0;
    float bounds_max[3];
    uint32_t index;
    uint32_t model_count;
    uint32_t _pad1[3];
};
struct Camera {            // raytrace.wgsl:35-47
    uint32_t sample_count;
    uint32_t bounce_count;
    uint32_t projection_type;
    float near_;
    float far_;
    float fov;
    float aspect;
    float _pad0;
    float position[3];
    float _pad1;
    float direction[3];
    float _pad2;
    float up[3];
    float _pad3;
};
struct Window {            // raytrace.wgsl:50-54
    float random_seed;
    uint32_t height;
    float _pad[2];
};

static_assert(sizeof(Model) == 32 && offsetof(Model, radius) == 12 && offsetof(Model, material_id) == 16, "Model");
static_assert(sizeof(Material) == 32 && offsetof(Material, metallic) == 12 &&
              offsetof(Material, specular_transmission) == 28, "Material");
static_assert(sizeof(BVHNode) == 48 && offsetof(BVHNode, bounds_max) == 16 && offsetof(BVHNode, index) == 28 &&
              offsetof(BVHNode, model_count) == 32, "BVHNode");
static_assert(sizeof(Camera) == 80 && offsetof(Camera, near_) == 12 && offsetof(Camera, aspect) == 24 &&
              offsetof(Camera, position) == 32 && offsetof(Camera, direction) == 48 && offsetof(Camera, up) == 64, "Camera");
static_assert(sizeof(Window) == 16 && offsetof(Window, height) == 4, "Window");

// ---- device-side scene encoding ------------------------------------------------------
//
// The reference re-reads a 48-byte node on every pop and then fetches both children
// (raytrace.wgsl:323,329,336): three dependent fetches per interior visit.  At upload the
// tree is re-encoded so that one interior visit is ONE record fetch and a pop needs no
// fetch at all: the traversal stack holds 32-bit child descriptors instead of node ids, and
// each interior node becomes a "pair record" with both children's boxes and descriptors.
// The push/pop ORDER is the reference's, so results (ties, stack-overflow rule) are the same.
//
// Descriptor (32-bit form; scenes of more than 16 382 spheres):
//   bit31 = 0                 interior: bits[30:0] = offset of the pair record in 16-byte units (index * 7), so that
//                             a granule address is one shift-add
//   bit31 = 1, bit30 = 1      leaf with exactly one sphere: bits[29:0] = model index
//   bit31 = 1, bit30 = 0      general leaf: bits[29:0] = index into the leaf table {first, count}
// 16-bit form (every index < 16383, always the case for an LDS-resident scene): the same three
// cases with the flags in bits 15/14 and a 14-bit index, so that a traversal-stack entry is 16 bits
// and 16 waves' stacks fit beside the scene (or its top levels) in a CU's LDS.  The interior index is
// the pair record's INDEX here (byte offset = index * 112, one v_mul_u32_u24 -- the same instruction
// count as the shift of the 32-bit form), so that the form covers trees of up to 16 382 interior nodes.
// REGISTER form (what the kernels compare, what pair records and root_desc hold): the 16-bit form
// SIGN-EXTENDED to 32 bits, the 32-bit form as it is.  All-ones (-1) is the "walk finished" marker,
// so as signed integers:   interior >= 0,   leaf < -1,   finished == -1   -- one compare each, and a
// finished lane fails both body tests without a separate guard.  A 16-bit stack entry is read
// back with a sign-extending load (ds_read_i16).
template <bool D16>
struct Desc {
    static constexpr uint32_t LEAF = D16 ? 0xFFFF8000u : 0x80000000u;   // OR-mask that makes a leaf descriptor
    static constexpr uint32_t LEAF1 = D16 ? 0x4000u : 0x40000000u;      // set together with LEAF
    static constexpr uint32_t INDEX_MASK = D16 ? 0x3FFFu : 0x3FFFFFFFu;
    static constexpr uint32_t DONE = 0xFFFFFFFFu;                       // "walk finished" marker, never a real descriptor
    static constexpr bool is_interior(uint32_t d) { return (int32_t)d >= 0; }
    static constexpr bool is_leaf(uint32_t d) { return (int32_t)d < -1; }
    // interior descriptor of pair record `index`, and back to the record's byte offset
    static constexpr uint32_t interior(uint32_t index) { return D16 ? index : index * 7u; }
    static constexpr uint32_t record_offset(uint32_t d) { return D16 ? d * 112u : d << 4; }
};
constexpr uint32_t DESC32_MAX_INDEX = 0x3FFFFFFEu;   // largest encodable index (all-ones is DONE)
constexpr uint32_t DESC16_MAX_INDEX = 0x3FFEu;

// Pair record: 112 bytes (28 words) per interior node, children L = node `index`, R = `index + 1`.
// The slab test needs, per axis, the plane the ray ENTERS through and the plane it LEAVES through;
// which of {min, max} that is depends only on the sign of the ray direction on that axis.  Each axis
// block therefore holds the four bounds twice, as two 16-byte granules
//   G0 = { min L, min R, max L, max R }      G1 = { max L, max R, min L, min R }
// and a lane reads ONE granule -- G0 for a direction >= 0, G1 for a direction < 0 -- as
// { near L, near R, far L, far R }: the reference's min()/max() per axis (raytrace.wgsl:391-392)
// become an address chosen once per ray, and the read is an aligned ds_read_b128 (4 LDS cycles per
// wave; two 8-byte reads at an unaligned offset would go through ds_read2_b64 at half the LDS rate).
//   bytes   0.. 31   x block (G0, G1)
//   bytes  32.. 63   y block
//   bytes  64.. 95   z block
//   bytes  96..103   descL, descR (register form); 104..111 padding
// The stride of 28 words spreads the granules of different records over 16 bank positions.
// That choice is exact when the ray is "safe" (origin finite, 1/direction finite and non-zero) and the
// boxes are finite with min <= max (checked at upload, `boxes_ordered`): then (b - o) * inv is monotone in b
// and no NaN can arise.  Otherwise the kernels apply min/max to the two values they read, which is the
// reference's expression whichever granule was read.
constexpr uint32_t PAIR_WORDS = 28;
constexpr uint32_t PAIR_BYTES = 112;
constexpr uint32_t PAIR_UNITS = PAIR_BYTES / 16;   // an interior descriptor counts records in 16-byte units
constexpr uint32_t PAIR_X = 0, PAIR_Y = 32, PAIR_Z = 64, PAIR_DESC = 96;   // byte offsets in a record
constexpr size_t pair_array_bytes(uint32_t n_pairs) { return (size_t)n_pairs * PAIR_BYTES; }
// Spheres: { center.x, center.y, center.z, radius*radius } (hit_sphere only uses r*r,
// raytrace.wgsl:375), material ids in a parallel u32 array.
// Materials: two float4 per material, as on the wire.

// Where the kernel reads pair records / spheres from (chosen per launch, brt_api.cpp plan_launch):
enum SceneMode : int {
    SCENE_GLOBAL = 0,    // everything from global memory (L2)
    SCENE_LDS = 1,       // pair records, spheres, material ids and leaf table copied to LDS by every workgroup
    SCENE_LDS_TOP = 2    // the first `lds_pairs` pair records (breadth-first order: the top of the tree) in LDS,
                         // deeper records, spheres and material ids from global memory
};

struct DeviceSceneView {
    const float* pairs;      // n_pairs records of PAIR_BYTES
    const float* spheres;    // float4[n_models]
    const uint32_t* sphere_material;  // u32[n_models]
    const float* materials;  // float4[2*n_materials]
    const float* sphere_mats;         // float4[2*n_models]: the material of every sphere (materials[sphere_material[i]])
    const uint32_t* leaf_table;       // uint2[n_leaf_table]
    uint32_t n_pairs, n_models, n_materials, n_leaf_table;
    uint32_t root_desc;
    uint32_t stack_entries;  // min(32, max leaf depth + 1)
    uint32_t desc16;         // 1: descriptors are in the 16-bit form
    uint32_t simple_tree;    // 1: every leaf holds one sphere and max leaf depth + 1 < 31
    uint32_t boxes_ordered;  // 1: every child box is finite with min <= max
    uint32_t lds_pairs;      // SCENE_LDS_TOP: pair records [0, lds_pairs) are staged in LDS (set per launch)
};

// Defaults of the tuning knobs in FrameParams.  The production kernel (TUNABLE = false) has them folded in as
// constants and the lane queue compiled out; brt_api.cpp picks the TUNABLE instantiation when the environment
// asks for anything else.
constexpr uint32_t kRefillMin = 1, kWalkExitLanes = 12, kLeafVote = 12, kDrainDonate = 40, kPoolAdopt = 56;

// Frame-uniform values, evaluated once on the host with the reference's expressions
// (raytrace.wgsl:95,141-153,177-182).
struct FrameParams {
    uint32_t width, height;          // render target size (uv = (p + 0.5) / size)
    uint32_t level;
    uint32_t sample_count, bounce_count;
    float seed_scaled;               // window.random_seed * 10000.0
    float inv_width, inv_height;     // 1.0 / (f32(window.height) * aspect), 1.0 / f32(window.height)
    float aspect, tan_half_fov;
    float cam_pos[3], cam_dir[3], cam_up[3], cam_right[3];  // right = cross(direction, up)
    float near_, far_, fallback_far;
    float spp_f;                     // f32(sample_count)
    // work decomposition
    uint32_t part, n_parts;          // interleaved strips: strip s -> part s % n_parts, unless ...
    const uint32_t* strip_of;        // ... a strip table is set (brt_set_strip_table): frame strip of this part's k-th local strip (local_strips entries;
                                     // a strip index past the frame: padding), or null
    uint32_t tiles_x;                // ceil(width / 8)
    uint32_t local_strips;           // strips owned by this part (padded count)
    uint32_t queue_size;             // local_strips * tiles_x * 64
    // Slots [0, queue_lane) are the lane queue (any free lane takes the next pixel); slots [queue_lane, queue_size)
    // are the tile queue: whole 8x8 tiles, taken one at a time by a wave that has nothing left.  Default:
    // queue_lane == 0, every tile goes through the tile queue.
    uint32_t queue_lane;
    // Pixels are independent but each is ONE sequential chain of sample_count paths (the
    // reference threads one RNG state through them), so the frame ends when the slowest pixels
    // end: hand out the (usually) expensive ones first.  With the usual "up = +Y" camera the
    // lower rows see ground and objects, the upper rows sky (one ray per sample).
    uint32_t bottom_up;
    uint32_t refill_min;             // lanes that must be free before a wave takes new pixels (1..64)
    uint32_t walk_exit_lanes;        // a wave leaves the walk loop when <= this many lanes still walk (0: never)
    uint32_t leaf_vote;              // lanes that must wait at a leaf before the wave runs the leaf body (0/1: always)
    // Drain (pixel queue empty): a wave with <= drain_donate live paths hands them to the workgroup's LDS pool
    // (pool_cap records) and ends; waves with idle lanes take them over.  pool_cap == 0: off.
    uint32_t drain_donate, pool_cap;
    uint32_t pool_adopt;             // a wave takes paths from the pool when it has at most this many live ones
    // Queue slots [crit_begin, crit_end) (the tiles with the longest pixel chains, when tile_order is set) are
    // CRITICAL: a wave that holds one of their pixels raises its issue priority (s_setprio), because the
    // frame cannot end before its longest sequential chain has.
    uint32_t crit_begin, crit_end;
    uint32_t wgq_batch;              // queue slots a workgroup takes at a time (multiple of 64, <= 512; chosen by launch_part)
    // Dispatch order: tile_order[k] = k-th tile to hand out (from the ray counts of the previous frame
    // of the same view, brt_api.cpp); this frame's measurement for the next one: tile_cost[tile] += rays
    // of each finished pixel, tile_cost[n_tiles + tile] = max of them.  Either may be null.
    const uint32_t* tile_order;
    uint32_t* tile_cost;
    const uint32_t* order_meta;      // order built on the GPU (brt_order.hip): [0] = critical tiles at its front (replaces crit_end),
                                     // [2], [3] = split_nonsky, split_tiles below
    // Half-sample jobs (brt_host.cpp build_tile_order): order positions [split_nonsky - split_tiles, split_nonsky) are FIRST halves (the
    // lane ends after sample_count / 2 samples and leaves its pixel state in slice_state), the next split_tiles positions the SECOND
    // halves of the same tiles; the tile queue then has split_tiles * 64 more slots than queue_size.  slice_state: 9 words per queue slot
    // (tile * 64 + position in the tile) in three planes -- {rng, sum.x, sum.y, sum.z} | {depth sum, rays so far, -, -} | flag -- a
    // record is valid when its flag carries slice_serial (brt_trace.h slice_slot).
    uint32_t split_nonsky, split_tiles;
    uint32_t* slice_state;
    uint32_t slice_serial;
    // Pre-pass of a scene walked from the LDS tile + global memory (SCENE_LDS_TOP): interior visits per pair record are counted (a
    // histogram in each workgroup's LDS, added here at the end of the launch); the host then re-numbers the records by how often THIS
    // view visits them, so that the tile holds the records the walk actually uses instead of the breadth-first top (brt_api.cpp
    // apply_hot_order).  n_pairs words, or null.  TUNABLE instantiation only.
    uint32_t* record_hits;
    uint32_t raster_dense;           // 1: the raster inputs hold this part's strips only, in the tile's own layout (row k * 8 + r = frame row
                                     // (k * n_parts + part) * 8 + r): what a device of an N-device context is sent; 0: the full frame
    uint32_t rows_on;                // 1: the launch's LDS holds the scratch of the row-mode walk of thin waves (brt_device.h walk_rows_asm; plan_launch leaves it
                                     // out when the scene only fits the LDS without it)
    uint32_t tunable;                // 1: some knob above differs from its default -> the TUNABLE kernel instantiation
    uint32_t policy_flags;           // TUNABLE only; bit 0: `||` of raytrace.wgsl:269 short-circuits (alternative policy)
};

}  // namespace brt
