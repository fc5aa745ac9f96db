// brt_ctx.h -- the opaque context of the C ABI (include/bevyray_amd.h) and the helpers its translation units share:
// brt_api.cpp (upload, render, builds) and brt_interop.cpp (RCCL gather, external-memory frames).  Internal.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "brt_host.h"
#include "brt_kernels.h"

namespace brt {

struct DeviceCtx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_last = nullptr;   // end of the last launch that used the control block (any stream)
    hipEvent_t ev_p0 = nullptr, ev_p1 = nullptr;   // around the dispatch-order pre-pass of a first frame
    int num_cus = 0;
    size_t max_lds = 0;
    // scene
    char* d_scene = nullptr;
    size_t scene_cap = 0;
    DeviceSceneView view{};
    // control block: 32 x u64 counters @0 (5 stats + section profile @8..23), queue counter @256
    char* d_ctrl = nullptr;
    // frame-sized buffers owned by the context (brt_render)
    float* d_tile = nullptr;
    size_t tile_cap = 0;
    float* d_raster_rgba = nullptr;
    size_t raster_rgba_cap = 0;
    float* d_raster_depth = nullptr;
    size_t raster_depth_cap = 0;
    float* d_gather = nullptr;   // first device only: the tiles of all devices back to back (brt_render_device)
    size_t gather_cap = 0;
    float* d_pack = nullptr;     // first device only: the raster inputs' strips of parts 1 .. N-1, packed per part (brt_render_device)
    size_t pack_cap = 0;
    hipEvent_t ev_pack = nullptr;   // first device: the strips are packed (the other devices' peer copies start behind it)
    hipEvent_t ev_copy = nullptr;   // this device's tile has arrived in the first device's gather buffer
    hipEvent_t ev_asm = nullptr;    // first device: the frame of the last brt_render_device call is assembled (the gather buffer is free)
    hipEvent_t ev_in = nullptr;     // first device: the caller's stream at the start of a brt_render_device call
    hipEvent_t ev_g0 = nullptr, ev_g1 = nullptr;   // first device: around waiting for the tiles + de-interleave
    float* h_stage = nullptr;  // pinned
    size_t stage_cap = 0;
    // longest-first dispatch (see plan_tile_order): this frame's per-tile ray counts and the
    // order derived from the previous frame of the same view
    uint32_t* d_tile_cost = nullptr;
    size_t tile_cost_cap = 0;
    uint32_t* d_tile_order = nullptr;
    uint32_t order_nonsky = 0, order_split = 0;          // host-built order: TileOrder::n_nonsky, n_split (a GPU-built one has them in d_order_meta)
    uint32_t* d_slice_state = nullptr;                   // FrameParams::slice_state
    size_t slice_state_cap = 0;
    uint32_t slice_serial = 0;
    uint32_t order_lane = 0;                             // tiles of the lane queue; the rest of d_tile_order is the tile queue
    uint32_t order_crit = 0;                             // d_tile_order[0 .. crit) are the CRITICAL tiles
    size_t tile_order_cap = 0;
    uint32_t* d_order_meta = nullptr;                    // order built on the GPU: [0] critical tiles, [1] longest pixel
    // strip table on this device (brt_set_strip_table): part_of_strip[n_strips], then strip_of[local_strips] of `strip_part`
    char* d_strip_table = nullptr;
    size_t strip_table_cap = 0;
    uint32_t strip_epoch = 0, strip_part = 0xffffffffu;  // the ctx->strip_epoch / part the device copy was made for
    char* d_order_scratch = nullptr;
    size_t order_scratch_cap = 0;
    bool order_on_device = false;                        // d_tile_order / d_order_meta were written by brt_order.hip
    uint64_t view_rays = 0;                              // rays of the last completed frame of the view `view_key` (0: unknown)
    uint32_t view_key[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // order key + sample_count, bounce_count
    bool order_valid = false;
    uint32_t remeasure_in = 0;                    // frames until the costs are measured again (0: nothing pending); set by scene uploads
    uint64_t order_cam = 0;                       // hash of the camera the costs were last measured with
    // the last measurement itself stays in d_tile_cost until the next one: a frame whose camera has moved since re-ranks the
    // tiles from it with a neighbourhood radius that covers the motion (attach_tile_order, brt_order.hip)
    bool costs_valid = false;
    float cost_cam_pos[3] = {0, 0, 0}, cost_cam_dir[3] = {0, 0, 0};
    uint32_t cost_spp = 0;
    uint32_t order_key[6] = {0, 0, 0, 0, 0, 0};   // width, height, part, n_parts, scene epoch, n_tiles
    std::vector<uint32_t> h_cost;
    std::vector<uint32_t> h_order;
    // pair records re-numbered by how often the view visits them (brt_api.cpp apply_hot_order)
    uint32_t* d_record_hits = nullptr;                   // FrameParams::record_hits of a pre-pass
    size_t record_hits_cap = 0;
    uint32_t hot_tree = 0;                               // brt_ctx::tree_epoch of the tree whose records are in hot order on this device (0: none)
    uint32_t hot_records = 0;                            // ... and how many of them the measuring pre-pass visited at all
    float hot_cam_pos[3] = {0, 0, 0}, hot_cam_dir[3] = {0, 0, 1};   // the camera they were counted with
    uint32_t frames_since_upload = 0;                    // frames this device has rendered since the last scene upload / tree rebuild
    std::vector<uint32_t> h_hits, h_rank;
    std::vector<uint32_t> h_total_rank, h_total_srank;   // the numbering of the records / spheres on this device as a map from the encoder's numbering
    uint64_t hot_shape = 0;                              // tree_shape_hash of the tree that numbering was counted on (0: none)
    std::vector<float> h_spheres_cur, h_sphmats_cur;
    std::vector<uint32_t> h_sphmat_cur;
    std::vector<float> h_pairs_hot, h_pairs_cur;          // scratch / the records as they are on the device (when hot_tree matches)
    // GPU BVH build
    char* d_bvh_scratch = nullptr;
    size_t bvh_scratch_cap = 0;
    char* d_bvh_models = nullptr;
    size_t bvh_models_cap = 0;
};


// ---- tuning knobs -----------------------------------------------------------------------------------------------------
// Scheduling / launch-shape knobs of the trace path.  None of them changes a pixel (every one has a test that says so).
// They live in the context: brt_set_tuning(ctx, name, value) sets one; brt_create reads the environment variables of the
// same names ONCE, and only when BRT_ENABLE_TUNING=1 is set -- nothing reads the environment per frame, and the one
// switch that does change pixels (the reading of `||` in raytrace.wgsl:269) is not a knob at all: brt_set_policy.
enum Knob : int {
    K_BOTTOM_UP, K_REFILL_MIN, K_WALK_EXIT, K_LEAF_VOTE, K_DRAIN_DONATE, K_POOL_ADOPT, K_WGQ_BATCH, K_LPT_LANE_PERMILLE, K_TUNABLE,
    K_FORCE_GLOBAL_SCENE, K_FORCE_LDS_TOP, K_BLOCK_THREADS, K_WG_PER_CU, K_POOL_CAP, K_LPT, K_LPT_SORT, K_LPT_SKY_SLACK, K_CRIT,
    K_ORDER_ON_HOST, K_NO_LEAN, K_PREPASS_SPP, K_NO_DIRTY_TRACKING, K_CPU_BVH, K_PLOC_ONE_BLOCK_MAX, K_BVH_QUALITY, K_POOL_FORCE, K_LPT_REFRESH_EVERY, K_LEAN_MEASURE, K_LPT_DILATE, K_SPLIT_TAIL, K_SPLIT_FORCE, K_HOT_RECORDS, K_TEST_THROW, K_BALL_SERVERS, K_COUNT
};
struct KnobDef { const char* name; uint32_t dflt; };
constexpr KnobDef kKnobs[K_COUNT] = {
    {"BRT_BOTTOM_UP", 0}, {"BRT_REFILL_MIN", kRefillMin}, {"BRT_WALK_EXIT", kWalkExitLanes}, {"BRT_LEAF_VOTE", kLeafVote},
    {"BRT_DRAIN_DONATE", kDrainDonate}, {"BRT_POOL_ADOPT", kPoolAdopt}, {"BRT_WGQ_BATCH", 0}, {"BRT_LPT_LANE_PERMILLE", 0},
    {"BRT_TUNABLE", 0}, {"BRT_FORCE_GLOBAL_SCENE", 0}, {"BRT_FORCE_LDS_TOP", 0}, {"BRT_BLOCK_THREADS", 0}, {"BRT_WG_PER_CU", 0},
    {"BRT_POOL_CAP", 384}, {"BRT_LPT", 1}, {"BRT_LPT_SORT", 1}, {"BRT_LPT_SKY_SLACK", 20}, {"BRT_CRIT", 1}, {"BRT_ORDER_ON_HOST", 0},
    {"BRT_NO_LEAN", 0}, {"BRT_PREPASS_SPP", 4}, {"BRT_NO_DIRTY_TRACKING", 0}, {"BRT_CPU_BVH", 0},
    {"BRT_PLOC_ONE_BLOCK_MAX", kPlocOneBlockMax}, {"BRT_BVH_QUALITY", 1}, {"BRT_POOL_FORCE", 0}, {"BRT_LPT_REFRESH_EVERY", 0}, {"BRT_LEAN_MEASURE", 1}, {"BRT_LPT_DILATE", 3}, {"BRT_SPLIT_TAIL", 16}, {"BRT_SPLIT_FORCE", 0}, {"BRT_HOT_RECORDS", 1}, {"BRT_TEST_THROW", 0}, {"BRT_BALL_SERVERS", 0}};
struct Knobs {
    uint32_t v[K_COUNT];
    Knobs() { for (int i = 0; i < K_COUNT; i++) v[i] = kKnobs[i].dflt; }
    uint32_t operator[](Knob k) const { return v[k]; }
};

}  // namespace brt

namespace brt {
// a frame target that lives in memory allocated elsewhere (brt_import_frame_fd, brt_interop.cpp)
struct ExternalFrame {
    void* ptr = nullptr;
    size_t bytes = 0, mapped = 0;
    uint32_t type = 0;                                   // BRT_EXTMEM_*
    hipExternalMemory_t ext = nullptr;                   // BRT_EXTMEM_OPAQUE_FD
    hipMemGenericAllocationHandle_t vmm{};               // BRT_EXTMEM_DMABUF_FD (and allocations exported by brt_debug_export_frame_fd)
};
}  // namespace brt

struct brt_ctx {
    std::vector<brt::ExternalFrame> external;   // imported / exported frame targets, released by brt_release_frame / brt_destroy
    std::vector<brt::DeviceCtx> devs;
    brt::EncodedScene enc;
    bool has_scene = false;
    uint32_t scene_epoch = 0;   // bumped by every upload
    // Strip table (brt_set_strip_table): strip_part[s] = the part that renders frame strip s, a permutation of the parts inside every
    // group of n_parts consecutive strips (so that a part's k-th local strip lies in group k: tile layout and gather stay as they are).
    // Empty: strip s belongs to part s % n_parts.
    std::vector<uint32_t> strip_part;
    uint32_t strip_n_parts = 0, strip_epoch = 0;
    uint32_t tree_epoch = 0;    // bumped whenever the encoded tree on the devices is rewritten (uploads, rebuilds for the camera's reach)
    uint32_t last_n_models = 0; // spheres of the last successful upload
    std::vector<std::pair<char*, size_t>> pinned;   // brt_host_alloc blocks
    // bytes of the last successful upload (dirty tracking: an unchanged scene is not re-sent)
    std::vector<char> last_models, last_materials, last_bvh;
    float scene_centre[3] = {0, 0, 0};   // mean centre of the scene's ordinary spheres (radius <= 100): what a camera translation is judged against
    // the callee-built SAH tree and the camera (brt_sah.h "leaf boxes", brt_api.cpp ensure_tree_reach): the leaf pads of the resident
    // tree cover rays of up to tree_reach; a camera that needs more has the tree rebuilt before its frame is launched
    bool tree_callee_sah = false;        // the resident tree was built here with the binned-SAH builder (a caller's tree is honoured as it comes)
    brt::TreeScene tree_scene;           // scale, radii and big spheres of the resident scene (brt_host.h)
    uint32_t tree_level = 0;             // reach of the resident tree = 2 S * 2^(level / 4); level 0: the scene's own extent
    float tree_reach = 0.0f;             // ... as passed to the builder (0 at level 0)
    uint32_t tree_rebuilds = 0;          // rebuilds since brt_create (diagnostic)
    brt::Knobs knobs;           // tuning knobs (brt_set_tuning; environment once at brt_create under BRT_ENABLE_TUNING=1)
    uint32_t policy_flags = 0;  // brt_set_policy
    std::string last_error;
};

namespace brt {

inline int32_t ctx_fail(brt_ctx* ctx, int32_t code, const std::string& msg) {
    if (ctx) ctx->last_error = msg;
    g_last_error = msg;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                              \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
            return ctx_fail(ctx, BRT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));       \
    } while (0)

template <typename T>
int32_t ensure(brt_ctx* ctx, T** ptr, size_t* cap, size_t bytes) {
    if (*cap >= bytes && *ptr) return BRT_OK;
    if (*ptr) HIP_TRY(ctx, hipFree(*ptr));
    *ptr = nullptr;
    *cap = 0;
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(ptr), bytes));
    *cap = bytes;
    return BRT_OK;
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

void release_external_frames(brt_ctx* ctx);   // brt_interop.cpp; called by brt_destroy

}  // namespace brt

namespace brt {
// strip table (brt_api.cpp): the device copies for fp->part on dc; *part_of_strip (frame strip -> part) for the assembly, fp->strip_of
// (this part's k-th local strip -> frame strip) for the kernel.  Nothing is attached when the context holds no table for fp's frame and split.
bool strip_table_valid(const uint32_t* part_of_strip, uint32_t n_strips, uint32_t n_parts);
int32_t strip_table_attach(brt_ctx* ctx, DeviceCtx& dc, FrameParams* fp, const uint32_t** part_of_strip, hipStream_t stream);
uint64_t part_pixels_table(const brt_ctx* ctx, const FrameParams& fp);
}  // namespace brt
