// brt_api.cpp -- the extern "C" boundary (include/bevyray_amd.h): context, scene upload,
// frame parameters, kernel launches, strip tiling over devices.
//
// What each export replaces in the reference is cited in the header.  This file holds no ray
// arithmetic: rays are traced only by the HIP kernels (brt_kernels.hip).  Without a usable
// HIP device brt_create fails with BRT_ERR_NO_DEVICE -- there is no CPU fallback.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "brt_ctx.h"
#include "brt_sah.h"

using namespace brt;

namespace {

constexpr uint32_t kPolicyMask = BRT_POLICY_OR_SHORT_CIRCUIT | BRT_POLICY_MINMAX_SELECT | BRT_POLICY_POW_EXP2_LOG2;

uint32_t env_u32(const char* name, uint32_t dflt) {
    const char* v = std::getenv(name);
    if (!v || !*v) return dflt;
    return (uint32_t)std::strtoul(v, nullptr, 10);
}

// Frame-uniform values with the reference's own expressions (raytrace.wgsl:95,141-153,177-182).
int32_t make_frame_params(brt_ctx* ctx, const void* camera80, const void* window16, uint32_t level, uint32_t width,
                          uint32_t height, uint32_t part, uint32_t n_parts, FrameParams* out) {
    if (const uint32_t k = ctx->knobs[K_TEST_THROW]) { ctx->knobs.v[K_TEST_THROW] = 0u; throw_for_test(k); }   // (tests: the exception barrier; every brt_render* comes through here)
    if (!camera80 || !window16) return ctx_fail(ctx, BRT_ERR_INVALID_ARGUMENT, "camera/window is null");
    if (width == 0 || height == 0 || width > 32768u || height > 32768u)
        return ctx_fail(ctx, BRT_ERR_INVALID_ARGUMENT, "width/height must be in [1, 32768]");
    if (n_parts == 0 || part >= n_parts) return ctx_fail(ctx, BRT_ERR_INVALID_ARGUMENT, "part >= n_parts");
    Camera cam;
    Window win;
    std::memcpy(&cam, camera80, sizeof cam);
    std::memcpy(&win, window16, sizeof win);
    if (cam.projection_type != 0)
        return ctx_fail(ctx, BRT_ERR_UNSUPPORTED, "only perspective projection (0) is supported (extract.rs:148)");
    if (cam.bounce_count >= 0x7fffffffu) return ctx_fail(ctx, BRT_ERR_INVALID_ARGUMENT, "bounce_count too large");
    FrameParams fp;
    std::memset(&fp, 0, sizeof fp);
    fp.width = width;
    fp.height = height;
    fp.level = level;
    fp.sample_count = cam.sample_count;
    fp.bounce_count = cam.bounce_count;
    fp.seed_scaled = win.random_seed * 10000.0f;
    const float heightf = (float)win.height;
    const float widthf = (float)win.height * cam.aspect;
    fp.inv_width = 1.0f / widthf;
    fp.inv_height = 1.0f / heightf;
    fp.aspect = cam.aspect;
    fp.tan_half_fov = tan_half_fov(cam.fov);
    for (int k = 0; k < 3; k++) {
        fp.cam_pos[k] = cam.position[k];
        fp.cam_dir[k] = cam.direction[k];
        fp.cam_up[k] = cam.up[k];
    }
    const float* a = cam.direction;
    const float* b = cam.up;
    fp.cam_right[0] = a[1] * b[2] - a[2] * b[1];
    fp.cam_right[1] = a[2] * b[0] - a[0] * b[2];
    fp.cam_right[2] = a[0] * b[1] - a[1] * b[0];
    fp.near_ = cam.near_;
    fp.far_ = cam.far_;
    fp.fallback_far = (level == 1u) ? cam.far_ + 10.0f : cam.far_ - 1.0f;
    fp.spp_f = (float)cam.sample_count;
    fp.part = part;
    fp.n_parts = n_parts;
    fp.tiles_x = (width + 7u) / 8u;
    const uint32_t strips = (height + BRT_STRIP_ROWS - 1u) / BRT_STRIP_ROWS;
    fp.local_strips = (strips + n_parts - 1u) / n_parts;
    fp.queue_size = fp.local_strips * fp.tiles_x * 64u;
    fp.queue_lane = 0u;              // whole tiles for idle waves; attach_tile_order may give the front of the order to the lane queue
    fp.bottom_up = ctx->knobs[K_BOTTOM_UP];
    fp.refill_min = ctx->knobs[K_REFILL_MIN];
    if (fp.refill_min < 1u) fp.refill_min = 1u;
    if (fp.refill_min > 64u) fp.refill_min = 64u;
    fp.walk_exit_lanes = ctx->knobs[K_WALK_EXIT];
    if (fp.walk_exit_lanes > 63u) fp.walk_exit_lanes = 63u;
    fp.leaf_vote = ctx->knobs[K_LEAF_VOTE];
    if (fp.leaf_vote > 64u) fp.leaf_vote = 64u;
    fp.drain_donate = ctx->knobs[K_DRAIN_DONATE];
    if (fp.drain_donate > 56u) fp.drain_donate = 56u;
    fp.pool_cap = 0;               // set by launch_part from the launch plan
    fp.pool_adopt = ctx->knobs[K_POOL_ADOPT];
    if (fp.pool_adopt > 63u) fp.pool_adopt = 63u;
    fp.crit_begin = fp.crit_end = 0;   // set by attach_tile_order
    fp.split_nonsky = fp.split_tiles = 0;
    fp.slice_state = nullptr;
    fp.slice_serial = 0;
    fp.wgq_batch = ctx->knobs[K_WGQ_BATCH] & ~63u;
    if (fp.wgq_batch > 512u) fp.wgq_batch = 512u;
    fp.policy_flags = ctx->policy_flags & kPolicyMask;
    // any knob off its default (or the lane queue asked for) -> the TUNABLE instantiation of the kernel
    fp.tunable = (fp.policy_flags != 0u || fp.bottom_up != 0u || fp.refill_min != kRefillMin || fp.walk_exit_lanes != kWalkExitLanes ||
                  fp.leaf_vote != kLeafVote || fp.drain_donate != kDrainDonate || fp.pool_adopt != kPoolAdopt ||
                  fp.wgq_batch != 0u || ctx->knobs[K_LPT_LANE_PERMILLE] != 0u || ctx->knobs[K_TUNABLE] != 0u)
                     ? 1u : 0u;
    *out = fp;
    return BRT_OK;
}

struct LaunchPlan {
    int scene_mode;              // SceneMode (brt_layout.h)
    uint32_t lds_pairs;          // SCENE_LDS_TOP: pair records staged in LDS
    uint32_t block, grid, wg_per_cu;
    uint32_t pool_cap;           // records of the drain pool per workgroup (0: none)
    uint32_t rows;               // 1: with the scratch of the row-mode walk (SCENE_LDS)
    size_t lds_bytes;
    uint32_t variant;            // brt_stats::kernel_variant of the launch
    uint32_t measured;           // the launch measured the tile costs
};

// Choose the kernel variant and grid.
//   SCENE_LDS      the whole encoded scene (pair records, spheres, material ids) fits a workgroup's LDS beside the
//                  stacks and the drain pool: one 1024-thread workgroup per CU (measured on the cover scene,
//                  bound by VALU pipe time; 4, 6 and 8 waves/SIMD run within 4 % of each other);
//   SCENE_LDS_TOP  it does not fit, but descriptors are 16-bit (<= 16 382 spheres): the LDS left beside stacks and
//                  pool holds the top of the tree (pair records are in breadth-first order), the rest comes from L2;
//   SCENE_GLOBAL   larger scenes: 256-thread workgroups, as many per CU as their stacks allow.
// The 32-byte materials always stay in global memory (read once per hit; measured: no difference).
// The knobs BRT_FORCE_GLOBAL_SCENE / BRT_FORCE_LDS_TOP=<records> / BRT_BLOCK_THREADS / BRT_WG_PER_CU override (tests, tuning).
LaunchPlan plan_launch(const Knobs& kn, const DeviceCtx& dc, const FrameParams& fp) {
    LaunchPlan lp{};
    const uint32_t hist = fp.record_hits ? dc.view.n_pairs : 0u;      // a pre-pass that counts record visits keeps a histogram in LDS (SCENE_LDS_TOP only)
    const bool force_global = kn[K_FORCE_GLOBAL_SCENE] != 0;
    const uint32_t force_top = kn[K_FORCE_LDS_TOP];
    const uint32_t block_env = kn[K_BLOCK_THREADS];
    const uint32_t wg_env = kn[K_WG_PER_CU];
    const uint32_t max_waves_cu = 32;
    lp.scene_mode = SCENE_GLOBAL;
    // drain pool: every wave but one may hand over up to drain_donate paths, but the takers empty the pool
    // while the donors fill it: 384 records (36 KB) are enough in practice, and a donation that does not
    // fit is simply retried a round later
    // ... and no pool at all when the launch has at most ~2.5 tiles per wave slot of the chip (a rank's share of a frame split 4 to 16
    // ways, a small frame): then every wave is in its last tiles from the start, the thinning waves would do nothing but pass paths
    // around, and a path that waits in the pool is a chain that stands still.  Measured, slowest share of config 2 in 2 / 4 / 8 / 16 parts
    // with / without the pool: 6.33 / 5.43 / 5.45 / 5.29 against 6.89 / 5.10 / 5.15 / 5.04 ms; config 4 in 8 / 16 parts: 106.8 / 91.0
    // against 107.9 / 86.4 ms (whole frames: 10.8 against 12.7 ms, 674 against 721 ms).
    const uint64_t n_tiles = fp.queue_size / 64u, wave_slots = (uint64_t)dc.num_cus * (BRT_BLOCK / 64u);
    // (BRT_POOL_FORCE=1 keeps the pool whatever the frame size: the pool's hand-over / take-over paths are then exercised by
    //  the small frames of the parity tests too)
    const uint32_t pool_max = (2u * n_tiles <= 5u * wave_slots && kn[K_POOL_FORCE] == 0u) ? 0u : kn[K_POOL_CAP];
    auto pool_of = [&](uint32_t block) {
        const uint32_t want = fp.drain_donate * (block / 64u - 1u);
        return want < pool_max ? want : pool_max;
    };
    if (!force_global && !force_top && dc.view.desc16) {
        struct Cand { uint32_t block, per_cu; };
        const Cand cands[] = {{1024, 1}, {512, 2}, {512, 3}, {1024, 2}, {512, 1}, {256, 1}};
        // (what is given up first when the scene is large: the 5 KB of the thin waves' row-mode scratch, then the drain pool -- never the
        //  LDS-resident scene itself for either of them)
        for (int opt = 0; opt < 4 && lp.scene_mode != SCENE_LDS; opt++) {
            const int with_pool = opt < 2 ? 1 : 0, with_rows = (opt & 1) == 0 ? 1 : 0;
            for (const Cand& c : cands) {
                if (block_env && c.block != block_env) continue;
                if (wg_env && c.per_cu != wg_env) continue;
                const uint32_t pool = with_pool ? pool_of(c.block) : 0u;
                const size_t need = trace_lds_bytes(dc.view, SCENE_LDS, c.block, pool, 0, with_rows != 0);
                if (need * c.per_cu <= dc.max_lds && c.per_cu * (c.block / 64) <= max_waves_cu) {
                    lp.scene_mode = SCENE_LDS;
                    lp.block = c.block;
                    lp.wg_per_cu = c.per_cu;
                    lp.lds_bytes = need;
                    lp.pool_cap = pool;
                    lp.rows = (uint32_t)with_rows;
                    break;
                }
            }
        }
    }
    if (lp.scene_mode != SCENE_LDS && !force_global && dc.view.desc16) {
        // top of the tree in LDS: everything that is left of a workgroup's LDS share for pair records
        DeviceSceneView v = dc.view;
        v.lds_pairs = 0;
        const uint32_t block = block_env ? block_env : BRT_BLOCK;
        const uint32_t per_cu = wg_env ? wg_env : 1u;
        const size_t share = dc.max_lds / per_cu;
        const uint32_t pool = pool_of(block) < 192u / per_cu ? pool_of(block) : 192u / per_cu;   // half the pool: the tile is worth more
        const size_t fixed = trace_lds_bytes(v, SCENE_LDS_TOP, block, pool, hist);
        if (fixed + 64 * PAIR_BYTES <= share && per_cu * (block / 64u) <= max_waves_cu) {
            uint32_t k = (uint32_t)((share - fixed) / PAIR_BYTES);
            if (k > v.n_pairs) k = v.n_pairs;
            if (force_top && force_top < k) k = force_top;
            v.lds_pairs = k;
            lp.scene_mode = SCENE_LDS_TOP;
            lp.lds_pairs = k;
            lp.block = block;
            lp.wg_per_cu = per_cu;
            lp.pool_cap = pool;
            lp.lds_bytes = trace_lds_bytes(v, SCENE_LDS_TOP, block, pool, hist);
        }
    }
    if (lp.scene_mode == SCENE_GLOBAL) {
        lp.block = block_env ? block_env : 256u;
        lp.pool_cap = pool_of(lp.block);
        lp.lds_bytes = trace_lds_bytes(dc.view, SCENE_GLOBAL, lp.block, lp.pool_cap);
        uint32_t per_cu = (uint32_t)(dc.max_lds / (lp.lds_bytes ? lp.lds_bytes : 1));
        const uint32_t by_waves = max_waves_cu / (lp.block / 64u);
        if (per_cu > by_waves) per_cu = by_waves;
        if (per_cu < 1) per_cu = 1;
        if (wg_env) per_cu = wg_env;
        lp.wg_per_cu = per_cu;
    }
    lp.grid = (uint32_t)dc.num_cus * lp.wg_per_cu;
    const uint32_t useful = (fp.queue_size + lp.block - 1u) / lp.block;
    if (lp.grid > useful) lp.grid = useful;
    if (lp.grid < 1) lp.grid = 1;
    return lp;
}

bool is_pinned(const brt_ctx* ctx, const void* p, size_t bytes) {
    const char* c = static_cast<const char*>(p);
    for (const auto& b : ctx->pinned)
        if (c >= b.first && c + bytes <= b.first + b.second) return true;
    return false;
}

// Dispatch order of the 8x8 tiles (brt_host.cpp build_tile_order has the rule).  The cost of a tile is not
// known in advance, but a renderer draws nearly the same frame again and again: the kernel measures the rays
// each tile needed (sum and longest pixel: two atomics per finished pixel, on the frames named below) and the next
// frames use the order built from that.  Pixels never change, only the queue order does.
// BRT_LPT=0 disables (raster order); BRT_LPT_SORT, BRT_LPT_LANE_PERMILLE, BRT_LPT_SKY_SLACK, BRT_CRIT: see
// update_tile_order.
// When the costs are measured again (round 4; the measurement itself runs in the LEAN instantiations now and costs a frame ~1 %):
//   * never for a view whose camera and scene do not change: there is nothing new to measure (until round 3: every 64th frame);
//   * EVERY frame while the camera moves (the reference's demo is a fly-camera app, src/main.rs:40): the order a frame runs in
//     is then the one measured on the frame before it -- one frame stale instead of up to 64;
//   * within kLptAfterUpload frames of a scene upload (an animated scene re-uploads every frame, extract.rs:299-336).
// The order is a hint: a stale one costs speed, never a pixel.
constexpr uint32_t kLptAfterUpload = 4;
bool lpt_enabled(const brt_ctx* ctx) { return ctx->knobs[K_LPT] != 0; }
uint64_t camera_hash(const FrameParams& fp) {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const float* f, int n) {
        for (int i = 0; i < n; i++) { uint32_t u; std::memcpy(&u, f + i, 4); h = (h ^ u) * 1099511628211ull; }
    };
    mix(fp.cam_pos, 3); mix(fp.cam_dir, 3); mix(fp.cam_up, 3); mix(&fp.tan_half_fov, 1); mix(&fp.aspect, 1);
    return h | 1ull;    // never 0 (= "not measured yet")
}

// The dispatch order only depends on which tiles hold long pixels: it survives a scene upload (an animated scene
// re-uploads every frame, extract.rs:299-336, and moves little between two frames) and is measured again soon after one
// (brt_upload_scene ages it); it never affects pixels.  The ray count of a view (LEAN = 2: "no pixel chain can be
// critical") additionally depends on sample and bounce counts; a wrong guess after a scene change only costs speed.
void order_key_of(const brt_ctx* ctx, const FrameParams& fp, uint32_t key[6]) {
    key[0] = fp.width; key[1] = fp.height; key[2] = fp.part; key[3] = fp.n_parts;
    key[4] = fp.strip_of ? ctx->strip_epoch : 0u;      // (another strip table: other strips, other costs)
    key[5] = fp.local_strips * fp.tiles_x;
}
void view_key_of(const brt_ctx* ctx, const FrameParams& fp, uint32_t key[8]) {
    order_key_of(ctx, fp, key);
    key[6] = fp.sample_count; key[7] = fp.bounce_count;
}

// How far, in 8-pixel tiles, the picture has moved since the costs were measured: the angle between the two viewing directions
// and a translation as seen at the distance of the scene's centre, in pixels of this frame, plus one tile.
constexpr uint32_t kMaxDilate = 8;      // beyond this the old costs say nothing about the new view: a first frame again (pre-pass)
// how far, in pixels of this frame, the picture has moved between a remembered camera and this frame's (NaN-safe: "very far")
double camera_motion_px(const brt_ctx* ctx, const float* pos0, const float* dir0, const FrameParams& fp) {
    const float* a = dir0;
    const float* b = fp.cam_dir;
    const double la = std::sqrt((double)a[0] * a[0] + (double)a[1] * a[1] + (double)a[2] * a[2]);
    const double lb = std::sqrt((double)b[0] * b[0] + (double)b[1] * b[1] + (double)b[2] * b[2]);
    double c = ((double)a[0] * b[0] + (double)a[1] * b[1] + (double)a[2] * b[2]) / (la * lb);
    if (!(c == c)) return 1e30;
    c = c > 1.0 ? 1.0 : (c < -1.0 ? -1.0 : c);
    const double px_per_rad = 0.5 * (double)fp.height / (double)fp.tan_half_fov;
    double dp = 0.0, dist = 0.0;
    for (int k = 0; k < 3; k++) {
        dp += ((double)fp.cam_pos[k] - pos0[k]) * ((double)fp.cam_pos[k] - pos0[k]);
        dist += ((double)fp.cam_pos[k] - ctx->scene_centre[k]) * ((double)fp.cam_pos[k] - ctx->scene_centre[k]);
    }
    const double rot_px = std::acos(c) * px_per_rad, trans_px = std::sqrt(dp) / std::max(std::sqrt(dist), 1e-3) * px_per_rad;
    const double px = std::max(rot_px, trans_px);
    return px == px ? px : 1e30;
}
uint32_t dilation_tiles(const brt_ctx* ctx, const DeviceCtx& dc, const FrameParams& fp) {
    const double px = camera_motion_px(ctx, dc.cost_cam_pos, dc.cost_cam_dir, fp);
    if (!(px < 8.0 * kMaxDilate)) return kMaxDilate + 1u;
    return (uint32_t)std::ceil(px / 8.0) + 1u;
}
// the camera has moved further than the old costs can follow
bool camera_jumped(const brt_ctx* ctx, const DeviceCtx& dc, const FrameParams& fp) {
    return dc.costs_valid && camera_hash(fp) != dc.order_cam && dilation_tiles(ctx, dc, fp) > kMaxDilate;
}

// Tiles at the end of the order that are handed out as two half-sample jobs (brt_host.cpp build_tile_order): BRT_SPLIT_TAIL quarters of
// the wave slots (default 16: four tiles per wave slot), but at least half the launch's tiles.  The second halves must come up late
// enough behind the first ones to find their states: on the headline frame 0 / 4 / 8 / 12 / 16 / 24 / every non-sky tile give 9.49 /
// 9.53 / 9.40 / 9.33 / 9.21 / 9.28 / 9.25 ms -- with 8, 1 % of the second halves come too early, leave the pixel to its first-half
// lane, and those lanes are the new stragglers; and where the jobs are longer the gap must be wider: the 4K / 1024 spp frame (129 600
// tiles) 589 ms without, 596 with 16, 587 with 64 or all -- hence the half (profiles/r04/split_tail.txt).  The builders cap the number
// at the non-sky tiles that are not critical.  Only for launches of at least 6 tiles per wave slot, like the neighbourhood ranking: a rank's share
// of a frame split 2 or 4 ways has nothing to balance at its end.
uint32_t split_tail_of(const brt_ctx* ctx, const DeviceCtx& dc, uint32_t n_tiles) {
    const uint64_t wave_slots = (uint64_t)dc.num_cus * (BRT_BLOCK / 64u);
    if (ctx->knobs[K_SPLIT_FORCE] != 0u) return ctx->knobs[K_SPLIT_FORCE] < n_tiles ? ctx->knobs[K_SPLIT_FORCE] : n_tiles;   // (tests: that many tiles, whatever the frame)
    if ((uint64_t)n_tiles < 6u * wave_slots) return 0u;
    uint64_t r = (uint64_t)ctx->knobs[K_SPLIT_TAIL] * wave_slots / 4u;
    if (r != 0u && r < n_tiles / 2u) r = n_tiles / 2u;
    return (uint32_t)(r < n_tiles ? r : n_tiles);
}

// the order of brt_order.hip from the costs in d_tile_cost (measured at `spp` samples per pixel), on `stream`
int32_t build_order_on_device(brt_ctx* ctx, DeviceCtx& dc, uint32_t n_tiles, uint32_t tiles_x, uint32_t spp, uint32_t rx, uint32_t ry,
                              hipStream_t stream) {
    if (!dc.d_order_meta) HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&dc.d_order_meta), 256));
    int32_t rc = ensure(ctx, &dc.d_order_scratch, &dc.order_scratch_cap, order_scratch_bytes(n_tiles));
    if (rc != BRT_OK) return rc;
    const uint32_t split_tail = split_tail_of(ctx, dc, n_tiles);
    rc = ensure(ctx, &dc.d_tile_order, &dc.tile_order_cap, ((size_t)n_tiles + split_tail) * 4);
    if (rc != BRT_OK) return rc;
    const uint64_t sky_cost = (uint64_t)64 * spp * (1000 + ctx->knobs[K_LPT_SKY_SLACK]) / 1000;   // as build_tile_order
    HIP_TRY(ctx, launch_build_order(dc.d_tile_cost, dc.d_tile_cost + n_tiles, n_tiles, sky_cost, (uint64_t)dc.num_cus * BRT_BLOCK, tiles_x, rx,
                                    ry, split_tail, dc.d_tile_order, dc.d_order_meta, dc.d_order_scratch, stream));
    HIP_TRY(ctx, hipEventRecord(dc.ev_last, stream));   // a later launch on another stream starts behind the order build
    return BRT_OK;
}

// before the launch: attach the order table if the history matches this view, and -- when the
// history is missing, the camera has moved or a scene upload asks for it -- the (zeroed) cost buffer to measure again
int32_t attach_tile_order(brt_ctx* ctx, DeviceCtx& dc, FrameParams& fp, hipStream_t stream, bool may_measure, uint32_t flags) {
    fp.tile_order = nullptr;
    fp.tile_cost = nullptr;
    // (the bring-up kernel takes slot q = tile q / 64 of a plain order: an order with half-sample jobs -- n_tiles + n_split entries,
    //  [non-sky | second halves | sky] -- would have it render the split tiles twice and the last sky tiles never: raster order there)
    if (!lpt_enabled(ctx) || fp.level == 0u || (flags & BRT_FLAG_KERNEL_SIMPLE)) return BRT_OK;
    const uint32_t n_tiles = fp.local_strips * fp.tiles_x;
    uint32_t key[6];
    order_key_of(ctx, fp, key);
    const bool match = dc.order_valid && std::memcmp(key, dc.order_key, sizeof key) == 0;
    if (match && dc.costs_valid && dc.order_on_device && camera_hash(fp) != dc.order_cam && ctx->knobs[K_LPT_DILATE] != 0u) {
        // the camera has moved since the costs were measured (one frame ago, while it keeps moving): rank every tile by its
        // neighbourhood of the radius the motion covers.  Behind the previous frame's work, ahead of this frame's launch.
        uint32_t r = dilation_tiles(ctx, dc, fp);
        if (r <= kMaxDilate && ctx->knobs[K_LPT_DILATE] >= 2u && r < ctx->knobs[K_LPT_DILATE] - 1u &&
            (uint64_t)n_tiles >= 6u * (uint64_t)dc.num_cus * (BRT_BLOCK / 64u))
            r = ctx->knobs[K_LPT_DILATE] - 1u;
        if (r <= kMaxDilate) {
            HIP_TRY(ctx, hipStreamWaitEvent(stream, dc.ev_last, 0));
            const int32_t rc = build_order_on_device(ctx, dc, n_tiles, fp.tiles_x, dc.cost_spp, r, (r + fp.n_parts - 1u) / fp.n_parts, stream);
            if (rc != BRT_OK) return rc;
        }
    }
    if (match) {
        fp.tile_order = dc.d_tile_order;
        fp.queue_lane = dc.order_lane * 64u;
        fp.crit_begin = 0u;
        fp.crit_end = dc.order_crit * 64u;
        fp.order_meta = dc.order_on_device ? dc.d_order_meta : nullptr;   // then the kernel reads the critical count there
        // half-sample jobs at the end of the order: where they are (a GPU-built order says so in d_order_meta) and where the pixel
        // states wait between a tile's two jobs
        if (split_tail_of(ctx, dc, n_tiles) != 0u) {
            const size_t bytes = (size_t)fp.queue_size * 36u;       // three planes of 16 + 16 + 4 bytes per queue slot (brt_trace.h slice_slot)
            const bool fresh = bytes > dc.slice_state_cap;
            int32_t rc = ensure(ctx, &dc.d_slice_state, &dc.slice_state_cap, bytes);
            if (rc != BRT_OK) return rc;
            if (fresh || dc.slice_serial >= 0x3fffffffu) {      // flags of a new buffer (or after 2^30 launches: a flag holds serial << 2) must not look valid
                HIP_TRY(ctx, hipMemsetAsync(dc.d_slice_state, 0, dc.slice_state_cap, stream));
                dc.slice_serial = 0u;
            }
            fp.slice_state = dc.d_slice_state;
            fp.slice_serial = ++dc.slice_serial;
            if (!dc.order_on_device) { fp.split_nonsky = dc.order_nonsky; fp.split_tiles = dc.order_split; }
        }
    }
    bool due = !match || camera_hash(fp) != dc.order_cam || ctx->knobs[K_LPT_REFRESH_EVERY] == 1u;
    if (may_measure && match && dc.remeasure_in != 0u && --dc.remeasure_in == 0u) due = true;
    if (may_measure && due) {
        int32_t rc = ensure(ctx, &dc.d_tile_cost, &dc.tile_cost_cap, (size_t)n_tiles * 8);   // sums, then maxima
        if (rc != BRT_OK) return rc;
        HIP_TRY(ctx, hipMemsetAsync(dc.d_tile_cost, 0, (size_t)n_tiles * 8, stream));
        fp.tile_cost = dc.d_tile_cost;
    }
    return BRT_OK;
}

// after a measuring frame has been enqueued on `stream`: build the order of the next frames.  Default settings: on the
// GPU, on the same stream, no host round trip (brt_order.hip).  BRT_ORDER_ON_HOST=1, or any non-default ordering knob
// (BRT_LPT_SORT, BRT_LPT_SKY_SLACK, BRT_LPT_LANE_PERMILLE, BRT_CRIT): counts to the host, build_tile_order, order back up.
int32_t update_tile_order(brt_ctx* ctx, DeviceCtx& dc, const FrameParams& fp, hipStream_t stream) {
    if (!fp.tile_cost) return BRT_OK;
    const uint32_t n_tiles = fp.local_strips * fp.tiles_x;
    TileOrderParams tp{};
    tp.sample_count = fp.sample_count;
    tp.grid_lanes = (uint64_t)dc.num_cus * BRT_BLOCK;
    tp.sorted = ctx->knobs[K_LPT_SORT];
    tp.sky_slack_permille = ctx->knobs[K_LPT_SKY_SLACK];
    tp.lane_permille = ctx->knobs[K_LPT_LANE_PERMILLE];
    tp.critical = ctx->knobs[K_CRIT];
    tp.tiles_x = fp.tiles_x;
    tp.split_tail = split_tail_of(ctx, dc, n_tiles);
    int32_t rc = ensure(ctx, &dc.d_tile_order, &dc.tile_order_cap, ((size_t)n_tiles + tp.split_tail) * 4);
    if (rc != BRT_OK) return rc;
    const bool on_device = tp.sorted == 1u && tp.critical == 1u && tp.lane_permille == 0u && ctx->knobs[K_ORDER_ON_HOST] == 0u;
    dc.costs_valid = false;
    if (on_device) {
        // BRT_LPT_DILATE (default 3): 0 = every tile ranked by itself; 1 = by a neighbourhood only when the camera has moved since the
        // measurement (attach_tile_order); v >= 2 = a neighbourhood of radius at least v - 1 always.  Radius 2 is the default also for
        // the view the costs were measured in: same box A/B, config 2 10.88 -> 10.64 ms, config 3 68.4 -> 66.7, config 5 25.4 -> 24.3
        // (profiles/r04/static_dilate_configs.txt) -- expensive pixels come in clusters (glass and metal silhouettes), and handing out
        // a cluster's tiles together starts all of its long chains early instead of ranking each tile on one noisy maximum.
        // Only for launches of at least 6 tiles per wave slot: a rank's share of a frame split 2 or 4 ways hands every wave 2-4 tiles, there
        // the truly longest tiles must be in the first hand-out and neighbours ranked up beside them push some of them into the second
        // (slowest share of config 2 in 2 / 4 parts 6.24 / 5.13 ms without, 6.60 / 5.31 with; profiles/r04/parts_dilate.txt).
        const uint64_t wave_slots = (uint64_t)dc.num_cus * (BRT_BLOCK / 64u);
        const uint32_t r0 = (ctx->knobs[K_LPT_DILATE] >= 2u && (uint64_t)n_tiles >= 6u * wave_slots) ? ctx->knobs[K_LPT_DILATE] - 1u : 0u;
        // (the same radius for the costs of a first frame's pre-pass: radius 0 / 1 / 2 / 3 / 4 / 6 there gave 11.2 / 11.6 / 11.6 / 12.0 /
        //  12.5 / 13.2 ms for pre-pass + frame, profiles/r04/first_frame_prepass.txt)
        tp.dilate_x = r0;
        tp.dilate_y = r0 ? (r0 + fp.n_parts - 1u) / fp.n_parts : 0u;
        rc = build_order_on_device(ctx, dc, n_tiles, fp.tiles_x, tp.sample_count, tp.dilate_x, tp.dilate_y, stream);
        if (rc != BRT_OK) return rc;
        dc.order_lane = 0;
        dc.order_crit = 0;
        dc.order_on_device = true;
        dc.costs_valid = true;
        dc.cost_spp = tp.sample_count;
        for (int k = 0; k < 3; k++) { dc.cost_cam_pos[k] = fp.cam_pos[k]; dc.cost_cam_dir[k] = fp.cam_dir[k]; }
    } else {
        dc.h_cost.resize(2 * (size_t)n_tiles);
        HIP_TRY(ctx, hipMemcpyAsync(dc.h_cost.data(), dc.d_tile_cost, (size_t)n_tiles * 8, hipMemcpyDeviceToHost, stream));
        HIP_TRY(ctx, hipStreamSynchronize(stream));
        TileOrder to;
        build_tile_order(dc.h_cost.data(), dc.h_cost.data() + n_tiles, n_tiles, tp, &to);   // sums, then longest pixels (brt_host.cpp)
        dc.h_order.swap(to.order);
        dc.order_lane = to.n_lane;
        dc.order_crit = to.n_critical;
        dc.order_on_device = false;
        dc.order_nonsky = to.n_nonsky;
        dc.order_split = to.n_split;
        HIP_TRY(ctx, hipMemcpyAsync(dc.d_tile_order, dc.h_order.data(), dc.h_order.size() * 4, hipMemcpyHostToDevice, stream));
        HIP_TRY(ctx, hipStreamSynchronize(stream));   // h_order may be reused by the next call
    }
    order_key_of(ctx, fp, dc.order_key);
    dc.order_valid = true;
    dc.remeasure_in = 0;
    dc.order_cam = camera_hash(fp);
    return BRT_OK;
}

// Launch the trace of one part on one device into d_out_tile.  Asynchronous on `stream`.
int32_t launch_part(brt_ctx* ctx, DeviceCtx& dc, const FrameParams& fp, const float* d_raster_rgba,
                    const float* d_raster_depth, float* d_out_tile, hipStream_t stream, uint32_t flags, bool timed,
                    LaunchPlan* plan_out) {
    // the control block and the order table are per context: whatever stream the previous launch ran on, this
    // one starts after it (include/bevyray_amd.h: one render per context in flight)
    HIP_TRY(ctx, hipStreamWaitEvent(stream, dc.ev_last, 0));
    HIP_TRY(ctx, hipMemsetAsync(dc.d_ctrl, 0, 512, stream));
    if (timed) HIP_TRY(ctx, hipEventRecord(dc.ev0, stream));
    LaunchPlan lp{};
    if (fp.level == 0u) {
        HIP_TRY(ctx, launch_passthrough(fp, d_out_tile, d_raster_rgba, stream));
    } else {
        TraceLaunch tl{};
        tl.scene = dc.view;
        tl.frame = fp;
        tl.queue_counter = reinterpret_cast<uint32_t*>(dc.d_ctrl + 256);
        tl.out_tile = d_out_tile;
        tl.raster_rgba = d_raster_rgba;
        tl.raster_depth = d_raster_depth;
        tl.counters = reinterpret_cast<unsigned long long*>(dc.d_ctrl);
        tl.counters_on = (flags & BRT_FLAG_COUNTERS) != 0;
        tl.stream = stream;
        if (flags & BRT_FLAG_KERNEL_SIMPLE) {
            if (fp.policy_flags != 0u) return ctx_fail(ctx, BRT_ERR_UNSUPPORTED, "the bring-up kernel implements the default policy only (brt_set_policy)");
            HIP_TRY(ctx, launch_trace_simple(tl));
            lp.block = 256;
            lp.grid = (fp.queue_size + 255u) / 256u;
        } else {
            lp = plan_launch(ctx->knobs, dc, fp);
            // LEAN instantiation (brt_trace.h): Pure level, not a measuring frame, and no critical tile possible -- a tile is
            // critical only if its longest pixel needs at least half a lane's share of the frame's rays (build_tile_order),
            // and no pixel needs more than sample_count * (bounce_count + 1); the frame's rays are those of the last
            // completed frame of this view.  (A wrong guess would only cost speed: critical tiles are a scheduling hint.)
            {
                uint32_t key[8];
                view_key_of(ctx, fp, key);
                const bool known = dc.view_rays != 0 && std::memcmp(key, dc.view_key, sizeof key) == 0;
                const uint64_t per_lane = known ? dc.view_rays / ((uint64_t)dc.num_cus * BRT_BLOCK) : 0;
                const uint64_t longest_bound = (uint64_t)fp.sample_count * ((uint64_t)fp.bounce_count + 1u);
                const bool lean1 = !tl.frame.tunable && fp.level == 3u && !tl.counters_on && ctx->knobs[K_NO_LEAN] == 0u &&
                                   (fp.tile_cost == nullptr || ctx->knobs[K_LEAN_MEASURE] != 0u);
                tl.lean = !lean1 ? 0 : ((known && longest_bound < per_lane / 2) ? 2 : 1);
                lp.variant = (uint32_t)tl.lean | (tl.frame.tunable ? 16u : 0u);
                lp.measured = fp.tile_cost != nullptr ? 1u : 0u;
            }
            // sampler stage (knob BRT_BALL_SERVERS, brt_trace.h SRV): the steady-state instantiation of an LDS-resident scene with two of the
            // workgroup's sixteen waves serving the rejection sampler of the other fourteen; its mailboxes come out of the drain pool's LDS
            tl.srv = 0u;
            if (ctx->knobs[K_BALL_SERVERS] != 0u && lp.scene_mode == SCENE_LDS && tl.lean == 2 && dc.view.simple_tree && lp.block == BRT_BLOCK &&
                lp.wg_per_cu == 1u) {
                const size_t pool_bytes = lp.pool_cap ? 16u + (size_t)lp.pool_cap * POOL_RECORD_BYTES : 0u, rest = lp.lds_bytes - pool_bytes;
                const size_t need = srv_lds_bytes(lp.block);
                if (rest + need <= dc.max_lds) {
                    size_t room = dc.max_lds - rest - need;
                    uint32_t pool = room > 16u ? (uint32_t)((room - 16u) / POOL_RECORD_BYTES) : 0u;
                    if (pool > lp.pool_cap) pool = lp.pool_cap;
                    if (lp.pool_cap != 0u && pool < 64u) pool = 0u;
                    lp.pool_cap = pool;
                    lp.lds_bytes = rest + (pool ? 16u + (size_t)pool * POOL_RECORD_BYTES : 0u) + need;
                    tl.srv = SRV_WAVES;
                    lp.variant |= 32u;
                }
            }
            tl.scene_mode = lp.scene_mode;
            tl.scene.lds_pairs = lp.lds_pairs;
            tl.grid = lp.grid;
            tl.block = lp.block;
            tl.lds_bytes = lp.lds_bytes;
            tl.frame.pool_cap = lp.pool_cap;
            tl.frame.rows_on = lp.scene_mode == SCENE_LDS ? lp.rows : 0u;
            if (tl.frame.wgq_batch == 0u) {
                // queue slots a workgroup takes at a time: at most half a pixel per lane, and small enough that every
                // workgroup comes back for at least 8 batches -- a rank that renders 1/8 of the frame has one tile
                // per wave and must hand them out one by one
                uint32_t b = (fp.queue_size / (lp.grid * 8u)) & ~63u;
                const uint32_t cap = (lp.block / 2u) & ~63u;
                if (b > cap) b = cap;
                if (b > 512u) b = 512u;
                if (b < 64u) b = 64u;
                tl.frame.wgq_batch = b;
            }
            HIP_TRY(ctx, launch_trace_persistent(tl));
        }
    }
    if (timed) HIP_TRY(ctx, hipEventRecord(dc.ev1, stream));
    HIP_TRY(ctx, hipEventRecord(dc.ev_last, stream));
    if (plan_out) *plan_out = lp;
    return BRT_OK;
}

// ---- pair records in the order of their use ------------------------------------------------------------------------------------
// A scene whose tree does not fit the LDS is walked from a tile of the first K records in LDS and the rest from L2 (SCENE_LDS_TOP).
// validate_and_encode numbers the records breadth first, so the tile is the top ~9.8 levels of the tree, everywhere in the scene; a
// view, though, walks a small part of the tree over and over: of the 10 004-sphere frame's interior visits the breadth-first tile
// of 879 records serves 78 %, the 879 records this view visits most would serve 98 % (3.7 -> 0.3 global steps per ray;
// profiles/r05/visit_hist_config5.txt), and the global steps are what the walk waits on (40 % of the wave cycles parked, the texture
// addresser 71 % busy: profiles/r05/config5_memory_side_before.json).  So the pre-pass of a first frame counts the visits per record
// (FrameParams::record_hits), and here the records are re-numbered by them, most visited first (ties: breadth-first order), the child
// descriptors and the root re-written, the records re-sent -- 1.1 MB and a host sort of 10 003 counts per pre-pass and device.
// The spheres are re-numbered with them (an internal numbering: nothing outside sees a sphere's index), see below.
// Only the NUMBERING of the records changes: the walk visits the same nodes in the same order, pixels and all five counters stay.
// the shape of the encoded tree: both child descriptors of every record, in the encoder's (breadth-first) numbering.  Two uploads with
// the same hash have the same records in the same places holding the same spheres -- only boxes and centres may have moved
uint64_t tree_shape_hash(const EncodedScene& e) {
    uint64_t h = 1469598103934665603ull ^ ((uint64_t)e.n_pairs << 32) ^ e.n_models;
    constexpr uint32_t kDescWord = PAIR_DESC / 4u;
    for (uint32_t r = 0; r < e.n_pairs; r++)
        for (uint32_t k = 0; k < 2u; k++) {
            uint32_t d;
            std::memcpy(&d, e.pairs.data() + (size_t)r * PAIR_WORDS + kDescWord + k, 4);
            h = (h ^ d) * 1099511628211ull;
        }
    return h | 1ull;
}

// records and spheres of `e` in the numbering rank / srank (old index -> new index) into the four host arrays; the root's descriptor
void permute_scene(const std::vector<float>& pairs, const std::vector<float>& spheres, const std::vector<uint32_t>& sphmat,
                   const std::vector<float>& sphmats, uint32_t root, const std::vector<uint32_t>& rank, const std::vector<uint32_t>& srank,
                   std::vector<float>* out_pairs, std::vector<float>* out_spheres, std::vector<uint32_t>* out_sphmat,
                   std::vector<float>* out_sphmats, uint32_t* out_root) {
    using D = Desc<true>;
    constexpr uint32_t kDescWord = PAIR_DESC / 4u;
    const uint32_t n = (uint32_t)rank.size(), m = (uint32_t)srank.size();
    auto remap = [&](uint32_t d) {
        if ((int32_t)d >= 0 && d < n) return rank[d];                                                  // interior (16-bit form: the record's index)
        if ((int32_t)d < -1 && (d & D::LEAF1) && (d & D::INDEX_MASK) < m) return (d & ~D::INDEX_MASK) | srank[d & D::INDEX_MASK];   // single-sphere leaf
        return d;
    };
    out_pairs->resize(pairs.size());
    for (uint32_t i = 0; i < n; i++) {
        const float* src = pairs.data() + (size_t)i * PAIR_WORDS;
        float* dst = out_pairs->data() + (size_t)rank[i] * PAIR_WORDS;
        std::memcpy(dst, src, PAIR_BYTES);
        for (uint32_t k = 0; k < 2u; k++) {
            uint32_t d;
            std::memcpy(&d, src + kDescWord + k, 4);
            d = remap(d);
            std::memcpy(dst + kDescWord + k, &d, 4);
        }
    }
    out_spheres->resize(spheres.size());
    out_sphmats->resize(sphmats.size());
    out_sphmat->resize(m);
    for (uint32_t i = 0; i < m; i++) {
        std::memcpy(out_spheres->data() + (size_t)srank[i] * 4, spheres.data() + (size_t)i * 4, 16);
        std::memcpy(out_sphmats->data() + (size_t)srank[i] * 8, sphmats.data() + (size_t)i * 8, 32);
        (*out_sphmat)[srank[i]] = sphmat[i];
    }
    *out_root = remap(root);
}

int32_t apply_hot_order(brt_ctx* ctx, DeviceCtx& dc, hipStream_t stream) {
    const uint32_t n = dc.view.n_pairs, m = dc.view.n_models;
    const EncodedScene& e = ctx->enc;
    if (n == 0u || e.pairs.size() != (size_t)n * PAIR_WORDS || e.spheres.size() != (size_t)m * 4u || e.sphere_material.size() != m || e.sphere_mats.size() != (size_t)m * 8u) return BRT_OK;
    // the records (and spheres) as they are on the device now: as encoded after an upload, else in the order of this device's last
    // pre-pass (the counts are indexed by THAT numbering)
    if (dc.hot_tree != ctx->tree_epoch) { dc.h_pairs_cur = e.pairs; dc.h_spheres_cur = e.spheres; dc.h_sphmat_cur = e.sphere_material; dc.h_sphmats_cur = e.sphere_mats; }
    const std::vector<float>& cur = dc.h_pairs_cur;
    const uint32_t cur_root = dc.hot_tree != ctx->tree_epoch ? e.root_desc : dc.view.root_desc;
    dc.h_hits.resize(n);
    HIP_TRY(ctx, hipMemcpyAsync(dc.h_hits.data(), dc.d_record_hits, (size_t)n * 4, hipMemcpyDeviceToHost, stream));
    HIP_TRY(ctx, hipStreamSynchronize(stream));
    std::vector<uint32_t>& rank = dc.h_rank;        // rank[old index] = new index
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return dc.h_hits[a] > dc.h_hits[b]; });
    rank.resize(n);
    uint32_t visited = 0;
    for (uint32_t i = 0; i < n; i++) { rank[order[i]] = i; visited += dc.h_hits[order[i]] != 0u ? 1u : 0u; }
    // the spheres likewise, by the visits of the record they hang under (a sphere is tested when its parent is visited and its box is
    // hit: the parent's count ranks it well enough -- the first 1 024 take 97 % of the 10 004-sphere frame's tests -- and needs no
    // second histogram): the leaf steps' sphere reads then fall on a few hot lines of the L1 (staging the first 512 / 1 024 / 2 048 of them
    // in LDS for the leaf step of the hand-written loop was built and measured: 13.88 / 13.97 / 14.2 ms against 13.92 without, not kept:
    // profiles/r05/config5_hot_records_ab.txt)
    using D = Desc<true>;
    constexpr uint32_t kDescWord = PAIR_DESC / 4u;
    std::vector<uint32_t> score(m, 0u), sorder(m), srank(m);
    auto leaf_sphere = [&](uint32_t d, uint32_t* id) {
        if ((int32_t)d < -1 && (d & D::LEAF1)) { *id = d & D::INDEX_MASK; return *id < m; }
        return false;
    };
    for (uint32_t r = 0; r < n; r++)
        for (uint32_t k = 0; k < 2u; k++) {
            uint32_t d, id;
            std::memcpy(&d, cur.data() + (size_t)r * PAIR_WORDS + kDescWord + k, 4);
            if (leaf_sphere(d, &id) && dc.h_hits[r] > score[id]) score[id] = dc.h_hits[r];
        }
    for (uint32_t i = 0; i < m; i++) sorder[i] = i;
    std::stable_sort(sorder.begin(), sorder.end(), [&](uint32_t a, uint32_t b) { return score[a] > score[b]; });
    for (uint32_t i = 0; i < m; i++) srank[sorder[i]] = i;
    std::vector<float> sph, mats;
    std::vector<uint32_t> mat;
    uint32_t new_root = 0;
    permute_scene(cur, dc.h_spheres_cur, dc.h_sphmat_cur, dc.h_sphmats_cur, cur_root, rank, srank, &dc.h_pairs_hot, &sph, &mat, &mats, &new_root);
    HIP_TRY(ctx, hipMemcpyAsync(const_cast<float*>(dc.view.pairs), dc.h_pairs_hot.data(), dc.h_pairs_hot.size() * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(ctx, hipMemcpyAsync(const_cast<float*>(dc.view.spheres), sph.data(), sph.size() * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(ctx, hipMemcpyAsync(const_cast<uint32_t*>(dc.view.sphere_material), mat.data(), mat.size() * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(ctx, hipMemcpyAsync(const_cast<float*>(dc.view.sphere_mats), mats.data(), mats.size() * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(ctx, hipStreamSynchronize(stream));       // (the vectors are reused by the next call)
    // the numbering as a map from the ENCODER's numbering (what the next upload of a tree of the same shape starts from): composed with
    // the one the counts were indexed by
    if (dc.hot_tree == ctx->tree_epoch && dc.h_total_rank.size() == n && dc.h_total_srank.size() == m) {
        for (uint32_t i = 0; i < n; i++) dc.h_total_rank[i] = rank[dc.h_total_rank[i]];
        for (uint32_t i = 0; i < m; i++) dc.h_total_srank[i] = srank[dc.h_total_srank[i]];
    } else {
        dc.h_total_rank = rank;
        dc.h_total_srank = srank;
    }
    dc.hot_shape = tree_shape_hash(e);
    dc.h_pairs_cur.swap(dc.h_pairs_hot);
    dc.h_spheres_cur.swap(sph);
    dc.h_sphmat_cur.swap(mat);
    dc.h_sphmats_cur.swap(mats);
    dc.view.root_desc = new_root;
    dc.hot_tree = ctx->tree_epoch;
    dc.hot_records = visited;
    return BRT_OK;
}

// The first frame of a view has no measured dispatch order (raster order: 12.9 instead of 9.6 ms on the headline
// frame).  A pre-pass of the same view at a few samples per pixel measures the tile costs first -- pixels are sequential
// chains of samples, so k samples predict the chain lengths of the full frame -- and the frame itself then runs in that
// order (and measures again, for the frames that follow).  k = min(BRT_PREPASS_SPP, samples / 16): default 4, 0 = off;
// measured on the headline frame (pre-pass + frame, ms): none 12.9, k = 1 14.2 (worse than raster order: one sample ranks
// the wrong tiles first), 2 11.5, 3 11.3, 4 11.0, 5 11.2, 6 11.3, 8 11.7 (profiles/r04/first_frame_prepass.txt) -- so no
// pre-pass where the rule would leave k = 1.  Enqueued on the context's own stream ahead of the frame (the order is built
// on the GPU behind it: no host round trip); only on the entry points that own their stream.  The pre-pass renders into
// the frame's own tile buffer; the frame overwrites every pixel of it afterwards.
int32_t prepass_order(brt_ctx* ctx, DeviceCtx& dc, const FrameParams& fp, const float* d_raster_rgba,
                      const float* d_raster_depth, float* d_out_tile, hipStream_t stream, uint32_t flags, bool* ran) {
    *ran = false;
    const uint32_t knob = ctx->knobs[K_PREPASS_SPP];
    const uint32_t k = knob < fp.sample_count / 16u ? knob : fp.sample_count / 16u;
    if (k == 0u || (k == 1u && knob != 1u) || !lpt_enabled(ctx) || fp.level == 0u || (flags & BRT_FLAG_KERNEL_SIMPLE)) return BRT_OK;
    uint32_t key[6];
    order_key_of(ctx, fp, key);
    // the records of a scene walked from an LDS tile were numbered for another view: once the picture has moved a quarter of the
    // frame's height since they were counted (a tile that holds another view's records serves fewer visits than the breadth-first one
    // would), this frame is a first frame again -- a pre-pass of ~2 ms every dozen frames of a steady orbit
    bool hot_stale = ctx->knobs[K_HOT_RECORDS] != 0u && dc.hot_tree == ctx->tree_epoch && dc.hot_records != 0u &&
                     camera_motion_px(ctx, dc.hot_cam_pos, dc.hot_cam_dir, fp) > std::max(64.0, 0.25 * (double)fp.height);
    // ... or were never numbered for this tree although the scene has stood still for two frames (an upload of a scene with the same
    // number of spheres keeps the dispatch order and runs no pre-pass -- right for a scene that is re-uploaded every frame, whose
    // tree changes under any count; a scene that then stays gets its records counted on its second frame)
    if (dc.frames_since_upload < 1000u) dc.frames_since_upload++;
    if (!hot_stale && ctx->knobs[K_HOT_RECORDS] != 0u && dc.hot_tree != ctx->tree_epoch && dc.frames_since_upload == 2u && dc.view.desc16 &&
        dc.view.simple_tree && dc.view.n_pairs > 64u && plan_launch(ctx->knobs, dc, fp).scene_mode == SCENE_LDS_TOP)
        hot_stale = true;
    // history matches and the camera has not jumped out of its reach: nothing to do
    if (dc.order_valid && std::memcmp(key, dc.order_key, sizeof key) == 0 && !(ctx->knobs[K_LPT_DILATE] != 0u && camera_jumped(ctx, dc, fp)) && !hot_stale)
        return BRT_OK;
    FrameParams pp = fp;
    pp.sample_count = k;
    pp.spp_f = (float)k;
    pp.tile_order = nullptr;
    pp.order_meta = nullptr;
    pp.slice_state = nullptr;     // (raster order: nothing is split)
    pp.split_nonsky = pp.split_tiles = 0u;
    pp.tunable = 1u;   // the knobs-live instantiation (all knobs at their defaults): the pre-pass then shows up under its own
                       // kernel name in rocprofv3 --stats instead of pulling down the average of the frame kernel
    pp.queue_lane = 0u;
    pp.crit_begin = pp.crit_end = 0u;
    const uint32_t n_tiles = pp.local_strips * pp.tiles_x;
    int32_t rc = ensure(ctx, &dc.d_tile_cost, &dc.tile_cost_cap, (size_t)n_tiles * 8);
    if (rc != BRT_OK) return rc;
    HIP_TRY(ctx, hipMemsetAsync(dc.d_tile_cost, 0, (size_t)n_tiles * 8, stream));
    pp.tile_cost = dc.d_tile_cost;
    // a scene that is walked from the LDS tile + global memory: the pre-pass also counts the interior visits per pair record, and the
    // records are then re-numbered by them (apply_hot_order): every pre-pass does (first frame of a view, camera jump)
    bool count_hits = false;
    if (ctx->knobs[K_HOT_RECORDS] != 0u && dc.view.desc16 && dc.view.simple_tree && dc.view.n_pairs > 64u &&
        plan_launch(ctx->knobs, dc, pp).scene_mode == SCENE_LDS_TOP) {
        rc = ensure(ctx, &dc.d_record_hits, &dc.record_hits_cap, (size_t)dc.view.n_pairs * 4);
        if (rc != BRT_OK) return rc;
        HIP_TRY(ctx, hipMemsetAsync(dc.d_record_hits, 0, (size_t)dc.view.n_pairs * 4, stream));
        pp.record_hits = dc.d_record_hits;
        count_hits = plan_launch(ctx->knobs, dc, pp).scene_mode == SCENE_LDS_TOP;      // (the histogram must leave room for a tile)
        if (!count_hits) pp.record_hits = nullptr;
    }
    HIP_TRY(ctx, hipEventRecord(dc.ev_p0, stream));
    rc = launch_part(ctx, dc, pp, d_raster_rgba, d_raster_depth, d_out_tile, stream, flags & ~(uint32_t)BRT_FLAG_COUNTERS, false, nullptr);
    if (rc != BRT_OK) return rc;
    HIP_TRY(ctx, hipEventRecord(dc.ev_p1, stream));
    rc = update_tile_order(ctx, dc, pp, stream);      // on the GPU, behind the pre-pass, no host round trip (default settings)
    if (rc != BRT_OK) return rc;
    if (count_hits) {
        rc = apply_hot_order(ctx, dc, stream);
        if (rc != BRT_OK) return rc;
        for (int q = 0; q < 3; q++) { dc.hot_cam_pos[q] = fp.cam_pos[q]; dc.hot_cam_dir[q] = fp.cam_dir[q]; }
    }
    dc.remeasure_in = 1u;                             // the frame that follows measures again, at full sample count
    *ran = true;
    return BRT_OK;
}

// kernel time of the pre-pass that prepass_order enqueued (its events have completed once the frame behind it has)
int32_t prepass_elapsed(brt_ctx* ctx, DeviceCtx& dc, bool ran, double* ms_out) {
    *ms_out = 0.0;
    if (!ran) return BRT_OK;
    float ms = 0.0f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, dc.ev_p0, dc.ev_p1));
    *ms_out = ms;
    return BRT_OK;
}

int32_t read_counters(brt_ctx* ctx, DeviceCtx& dc, hipStream_t stream, brt_stats* st) {
    unsigned long long c[5];
    HIP_TRY(ctx, hipMemcpyAsync(c, dc.d_ctrl, sizeof c, hipMemcpyDeviceToHost, stream));
    HIP_TRY(ctx, hipStreamSynchronize(stream));
    st->rays += c[0];
    st->node_pops += c[1];
    st->interior_visits += c[2];
    st->sphere_tests += c[3];
    st->hits += c[4];
    return BRT_OK;
}

uint64_t part_pixels(const FrameParams& fp) {
    uint64_t rows = 0;
    const uint32_t strips = (fp.height + BRT_STRIP_ROWS - 1u) / BRT_STRIP_ROWS;
    // (a part has one strip in every group of n_parts strips whatever the strip table says -- except in the last, partial group:
    //  callers with a table correct that: strip_table_attach)
    for (uint32_t s = fp.part; s < strips; s += fp.n_parts) {
        const uint32_t r0 = s * BRT_STRIP_ROWS;
        rows += (r0 + BRT_STRIP_ROWS <= fp.height) ? BRT_STRIP_ROWS : (fp.height - r0);
    }
    return rows * fp.width;
}

void free_device(DeviceCtx& dc) {
    if (hipSetDevice(dc.device) != hipSuccess) return;
    if (dc.d_scene) (void)hipFree(dc.d_scene);
    if (dc.d_ctrl) (void)hipFree(dc.d_ctrl);
    if (dc.d_strip_table) (void)hipFree(dc.d_strip_table);
    if (dc.d_tile) (void)hipFree(dc.d_tile);
    if (dc.d_gather) (void)hipFree(dc.d_gather);
    if (dc.d_pack) (void)hipFree(dc.d_pack);
    for (hipEvent_t e : {dc.ev_copy, dc.ev_asm, dc.ev_in, dc.ev_g0, dc.ev_g1, dc.ev_pack})
        if (e) (void)hipEventDestroy(e);
    if (dc.d_raster_rgba) (void)hipFree(dc.d_raster_rgba);
    if (dc.d_raster_depth) (void)hipFree(dc.d_raster_depth);
    if (dc.h_stage) (void)hipHostFree(dc.h_stage);
    if (dc.d_bvh_scratch) (void)hipFree(dc.d_bvh_scratch);
    if (dc.d_tile_cost) (void)hipFree(dc.d_tile_cost);
    if (dc.d_tile_order) (void)hipFree(dc.d_tile_order);
    if (dc.d_order_meta) (void)hipFree(dc.d_order_meta);
    if (dc.d_order_scratch) (void)hipFree(dc.d_order_scratch);
    if (dc.d_slice_state) (void)hipFree(dc.d_slice_state);
    if (dc.d_record_hits) (void)hipFree(dc.d_record_hits);
    if (dc.d_bvh_models) (void)hipFree(dc.d_bvh_models);
    if (dc.ev0) (void)hipEventDestroy(dc.ev0);
    if (dc.ev1) (void)hipEventDestroy(dc.ev1);
    if (dc.ev_last) (void)hipEventDestroy(dc.ev_last);
    if (dc.ev_p0) (void)hipEventDestroy(dc.ev_p0);
    if (dc.ev_p1) (void)hipEventDestroy(dc.ev_p1);
    if (dc.stream) (void)hipStreamDestroy(dc.stream);
    dc = DeviceCtx();
}

constexpr uint32_t kMaxSahModels = 1u << 16;         // host-side binned SAH for callee-built trees up to here
constexpr uint32_t kMaxGpuBuildModels = 1u << 24;   // scratch ~ 250 B per sphere; the grid version of the builder has no structural limit

// A BVH built on the context's first device: models (host) -> nodes (host vector), kernel time in ms.  sah: the binned-SAH tree of
// brt_sah.h (brt_sah.hip), else PLOC (brt_bvh.hip); each byte-identical to its CPU twin.
int32_t build_bvh_on_device(brt_ctx* ctx, const Model* models, uint32_t n, bool sah, float reach, std::vector<BVHNode>* out, double* build_ms) {
    out->clear();
    if (n == 0) return BRT_OK;
    if (n > (sah ? kMaxSahModels : kMaxGpuBuildModels))
        return ctx_fail(ctx, BRT_ERR_UNSUPPORTED, sah ? "GPU SAH build supports up to 65 536 spheres" : "GPU BVH build supports up to 2^24 spheres");
    DeviceCtx& dc = ctx->devs[0];
    HIP_TRY(ctx, hipSetDevice(dc.device));
    int32_t rc = ensure(ctx, &dc.d_bvh_scratch, &dc.bvh_scratch_cap,
                        sah ? sah_scratch_bytes(n) : ploc_scratch_bytes(n, nullptr, ctx->knobs[K_PLOC_ONE_BLOCK_MAX]));
    if (rc != BRT_OK) return rc;
    rc = ensure(ctx, &dc.d_bvh_models, &dc.bvh_models_cap, (size_t)n * sizeof(Model));
    if (rc != BRT_OK) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(dc.d_bvh_models, models, (size_t)n * sizeof(Model), hipMemcpyHostToDevice, dc.stream));
    BVHNode* d_out = nullptr;
    uint32_t* d_info = nullptr;
    HIP_TRY(ctx, hipEventRecord(dc.ev0, dc.stream));
    if (sah) HIP_TRY(ctx, launch_build_sah(reinterpret_cast<const Model*>(dc.d_bvh_models), n, reach, dc.d_bvh_scratch, &d_out, &d_info, dc.stream));
    else HIP_TRY(ctx, launch_build_ploc(reinterpret_cast<const Model*>(dc.d_bvh_models), n, dc.d_bvh_scratch, &d_out, &d_info, ctx->knobs[K_PLOC_ONE_BLOCK_MAX], dc.stream));
    HIP_TRY(ctx, hipEventRecord(dc.ev1, dc.stream));
    out->resize(2 * (size_t)n - 1);
    HIP_TRY(ctx, hipMemcpyAsync(out->data(), d_out, out->size() * sizeof(BVHNode), hipMemcpyDeviceToHost, dc.stream));
    uint32_t info[2] = {0u, 0u};
    HIP_TRY(ctx, hipMemcpyAsync(info, d_info, sizeof info, hipMemcpyDeviceToHost, dc.stream));
    HIP_TRY(ctx, hipStreamSynchronize(dc.stream));
    if (info[0] != 2u * n - 1u) {
        out->clear();
        return ctx_fail(ctx, BRT_ERR_HIP, "GPU BVH build made no progress after " + std::to_string(info[1]) + " rounds");
    }
    if (sah) sah_giant_leaves_first(out->data(), (uint32_t)out->size(), models, n);      // (the rule's last step: brt_sah.h)
    float ms = 0.0f;
    HIP_TRY(ctx, hipEventElapsedTime(&ms, dc.ev0, dc.ev1));
    if (build_ms) *build_ms = ms;
    return BRT_OK;
}

// ---- the callee-built SAH tree and the camera (brt_sah.h "leaf boxes of the tree the CALLEE builds") -------------------------------
// The leaf pads of that tree cover the rounding of the sphere test for rays of up to `reach`; at upload the camera is unknown and the
// tree is built for the scene's own extent, reach = 2 S.  Every render call checks its camera: need = |camera|_1 + S + L (L: the longest
// tangent from the camera to a big sphere -- how far from the camera a primary ray can land on the ground, from where it bounces back
// into the scene).  Reaches come in steps of 2^(1/4) (level k: 2 S * 2^(k / 4), the pads grow by 2^(1/2) per step): a camera that needs
// a higher level than the resident tree has -- or at least two levels less: it has come back -- gets the tree rebuilt on the GPU before
// its frame is launched (brt_sah.hip: 0.2-0.5 ms + the re-encode); never a tree whose pads are below what the camera needs.  A
// caller's tree (and the callee's PLOC tree: the reference's flat 0.1) is honoured as it comes.
// (the rule itself -- tree_scene_of, tree_level_for, tree_reach_of, tree_pads_equal -- is host arithmetic: brt_host.cpp, exported as brt_host_tree_reach)
int32_t upload_scene(brt_ctx* ctx, const void* models, uint32_t n_models, const void* materials, uint32_t n_materials,
                     const void* bvh_nodes, uint32_t n_nodes, uint32_t level, bool rebuild);
// before a frame is launched: *rebuilt = the tree was rebuilt for this camera
int32_t ensure_tree_reach(brt_ctx* ctx, const void* camera80, uint32_t* rebuilt) {
    *rebuilt = 0u;
    if (!ctx->has_scene || !ctx->tree_callee_sah || !camera80) return BRT_OK;
    Camera cam;
    std::memcpy(&cam, camera80, sizeof cam);
    uint32_t need = tree_level_for(ctx->tree_scene.scale, ctx->tree_scene.big, cam.position);
    if (tree_pads_equal(ctx->tree_scene, 0.0f, tree_reach_of(ctx->tree_scene.scale, need))) need = 0u;   // the tree of the scene's own extent is that tree
    if (need <= ctx->tree_level && need + 2u > ctx->tree_level) return BRT_OK;
    // (the cover camera at its usual place already "needs" level 5 -- with every pad still at the 0.01 floor: same bytes, level 0)
    if (tree_pads_equal(ctx->tree_scene, ctx->tree_reach, tree_reach_of(ctx->tree_scene.scale, need))) return BRT_OK;
    *rebuilt = 1u;
    return upload_scene(ctx, ctx->last_models.data(), (uint32_t)(ctx->last_models.size() / sizeof(Model)), ctx->last_materials.data(),
                        (uint32_t)(ctx->last_materials.size() / sizeof(Material)), nullptr, 0u, need, true);
}
void tree_stats(const brt_ctx* ctx, uint32_t rebuilt, brt_stats* stats) {
    if (!stats) return;
    stats->tree_rebuilt = rebuilt;
    stats->tree_reach = ctx->tree_callee_sah ? ctx->tree_reach : 0.0f;
}

}  // namespace

extern "C" {

uint32_t brt_abi_version(void) { return BRT_ABI_VERSION; }

const char* brt_last_error(const brt_ctx* ctx) { return ctx ? ctx->last_error.c_str() : g_last_error.c_str(); }

int32_t brt_create(const int32_t* device_ids, int32_t n_devices, brt_ctx** out_ctx) {
    return guard(nullptr, [&]() -> int32_t {
    if (!out_ctx) return fail(BRT_ERR_INVALID_ARGUMENT, "out_ctx is null");
    *out_ctx = nullptr;
    if (!device_ids || n_devices < 1 || n_devices > 64) return fail(BRT_ERR_INVALID_ARGUMENT, "need 1..64 device ids");
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count < 1)
        return fail(BRT_ERR_NO_DEVICE, std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "count 0") +
                                           " (this library has no CPU path)");
    brt_ctx* ctx = new (std::nothrow) brt_ctx();
    if (!ctx) return guard_fail(nullptr, BRT_ERR_OUT_OF_MEMORY, "out of memory");
    // (anything below that throws -- the vector, a std::string of an error text -- must not leak the context and its device objects)
    struct Cleanup {
        brt_ctx* c;
        ~Cleanup() { if (c) { for (auto& d : c->devs) free_device(d); delete c; } }
    } cleanup{ctx};
    ctx->devs.resize((size_t)n_devices);
    for (int i = 0; i < n_devices; i++) {
        DeviceCtx& dc = ctx->devs[(size_t)i];
        dc.device = device_ids[i];
        int32_t rc = BRT_OK;
        auto body = [&]() -> int32_t {
            if (dc.device < 0 || dc.device >= count)
                return ctx_fail(ctx, BRT_ERR_INVALID_ARGUMENT, "device id " + std::to_string(dc.device) + " out of range");
            HIP_TRY(ctx, hipSetDevice(dc.device));
            hipDeviceProp_t prop;
            HIP_TRY(ctx, hipGetDeviceProperties(&prop, dc.device));
            if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
                return ctx_fail(ctx, BRT_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950");
            dc.num_cus = prop.multiProcessorCount;
            int lds = 0;
            HIP_TRY(ctx, hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dc.device));
            dc.max_lds = (size_t)lds;
            HIP_TRY(ctx, hipStreamCreateWithFlags(&dc.stream, hipStreamNonBlocking));
            HIP_TRY(ctx, hipEventCreate(&dc.ev0));
            HIP_TRY(ctx, hipEventCreate(&dc.ev1));
            HIP_TRY(ctx, hipEventCreate(&dc.ev_p0));
            HIP_TRY(ctx, hipEventCreate(&dc.ev_p1));
            HIP_TRY(ctx, hipEventCreateWithFlags(&dc.ev_last, hipEventDisableTiming));
            HIP_TRY(ctx, hipEventCreateWithFlags(&dc.ev_copy, hipEventDisableTiming));
            HIP_TRY(ctx, hipEventCreateWithFlags(&dc.ev_asm, hipEventDisableTiming));
            HIP_TRY(ctx, hipEventCreateWithFlags(&dc.ev_in, hipEventDisableTiming));
            HIP_TRY(ctx, hipEventCreateWithFlags(&dc.ev_pack, hipEventDisableTiming));
            HIP_TRY(ctx, hipEventCreate(&dc.ev_g0));
            HIP_TRY(ctx, hipEventCreate(&dc.ev_g1));
            HIP_TRY(ctx, hipEventRecord(dc.ev_asm, dc.stream));
            HIP_TRY(ctx, hipEventRecord(dc.ev_last, dc.stream));
            HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&dc.d_ctrl), 512));
            return BRT_OK;
        };
        rc = body();
        if (rc != BRT_OK) {
            g_last_error = ctx->last_error;
            return rc;          // (cleanup frees the devices and the context)
        }
    }
    // tuning knobs from the environment: once, here, and only on request (BRT_ENABLE_TUNING=1)
    if (env_u32("BRT_ENABLE_TUNING", 0) != 0u)
        for (int k = 0; k < K_COUNT; k++) ctx->knobs.v[k] = env_u32(kKnobs[k].name, kKnobs[k].dflt);
    cleanup.c = nullptr;
    *out_ctx = ctx;
    return BRT_OK;
    });
}

int32_t brt_set_policy(brt_ctx* ctx, uint32_t flags) {
    return guard(ctx ? &ctx->last_error : nullptr, [&]() -> int32_t {
    if (!ctx) return fail(BRT_ERR_INVALID_ARGUMENT, "ctx is null");
    if (flags & ~kPolicyMask) return ctx_fail(ctx, BRT_ERR_INVALID_ARGUMENT, "unknown policy flag");
    ctx->policy_flags = flags;
    return BRT_OK;
    });
}

int32_t brt_set_tuning(brt_ctx* ctx, const char* name, uint32_t value) {
    return guard(ctx ? &ctx->last_error : nullptr, [&]() -> int32_t {
    if (!ctx || !name) return fail(BRT_ERR_INVALID_ARGUMENT, "null pointer");
    for (int k = 0; k < K_COUNT; k++)
        if (std::strcmp(name, kKnobs[k].name) == 0) {
            ctx->knobs.v[k] = value;
            // a knob may change how the dispatch order is built or used: forget the history of every view (the next frame of
            // a view is a "first frame" again: pre-pass, measuring frame)
            for (auto& dc : ctx->devs) { dc.order_valid = false; dc.view_rays = 0; }
            // ... and a knob of the callee's BVH build changes what the same scene bytes upload to: no dirty-tracking shortcut
            if (k == K_BVH_QUALITY || k == K_CPU_BVH || k == K_PLOC_ONE_BLOCK_MAX) ctx->last_models.clear();
            return BRT_OK;
        }
    return ctx_fail(ctx, BRT_ERR_INVALID_ARGUMENT, std::string("unknown tuning knob ") + name);
    });
}

int32_t brt_get_tuning(const brt_ctx* ctx, const char* name, uint32_t* out_value, uint32_t* out_default) {
    return guard(nullptr, [&]() -> int32_t {
    if (!ctx || !name) return fail(BRT_ERR_INVALID_ARGUMENT, "null pointer");
    for (int k = 0; k < K_COUNT; k++)
        if (std::strcmp(name, kKnobs[k].name) == 0) {
            if (out_value) *out_value = ctx->knobs.v[k];
            if (out_default) *out_default = kKnobs[k].dflt;
            return BRT_OK;
        }
    return fail(BRT_ERR_INVALID_ARGUMENT, std::string("unknown tuning knob ") + name);
    });
}

int32_t brt_host_alloc(brt_ctx* ctx, uint64_t bytes, void** out_ptr) {
    return guard(ctx ? &ctx->last_error : nullptr, [&]() -> int32_t {
    if (!ctx || !out_ptr || bytes == 0) return fail(BRT_ERR_INVALID_ARGUMENT, "null pointer / zero size");
    *out_ptr = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->devs[0].device));
    void* p = nullptr;
    HIP_TRY(ctx, hipHostMalloc(&p, (size_t)bytes, hipHostMallocPortable));
    ctx->pinned.emplace_back(static_cast<char*>(p), (size_t)bytes);
    *out_ptr = p;
    return BRT_OK;
    });
}

int32_t brt_host_free(brt_ctx* ctx, void* ptr) {
    return guard(ctx ? &ctx->last_error : nullptr, [&]() -> int32_t {
    if (!ctx) return fail(BRT_ERR_INVALID_ARGUMENT, "ctx is null");
    for (size_t i = 0; i < ctx->pinned.size(); i++)
        if (ctx->pinned[i].first == ptr) {
            HIP_TRY(ctx, hipHostFree(ptr));
            ctx->pinned.erase(ctx->pinned.begin() + (long)i);
            return BRT_OK;
        }
    return ctx_fail(ctx, BRT_ERR_INVALID_ARGUMENT, "pointer was not allocated by brt_host_alloc");
    });
}

int32_t brt_destroy(brt_ctx* ctx) {
    return guard(ctx ? &ctx->last_error : nullptr, [&]() -> int32_t {
    if (!ctx) return BRT_OK;
    release_external_frames(ctx);
    for (auto& b : ctx->pinned) (void)hipHostFree(b.first);
    for (auto& d : ctx->devs) free_device(d);
    delete ctx;
    return BRT_OK;
    });
}

int32_t brt_upload_scene(brt_ctx* ctx, const void* models, uint32_t n_models, const void* materials, uint32_t n_materials,
                         const void* bvh_nodes, uint32_t n_nodes) {
    return guard(ctx ? &ctx->last_error : nullptr, [&]() -> int32_t {
    if (!ctx) return fail(BRT_ERR_INVALID_ARGUMENT, "ctx is null");
    return upload_scene(ctx, models, n_models, materials, n_materials, bvh_nodes, n_nodes, 0u, false);
    });
}

}  // extern "C"

namespace {

// brt_upload_scene; and, with rebuild_level != 0 / `rebuild`, the same scene bytes again (ctx->last_*) in a callee-built SAH tree whose
// leaf pads cover a longer reach (ensure_tree_reach): the dispatch-order history and the dirty-tracking state stay as they are.
int32_t upload_scene(brt_ctx* ctx, const void* models, uint32_t n_models, const void* materials, uint32_t n_materials,
                     const void* bvh_nodes, uint32_t n_nodes, uint32_t level, bool rebuild) {
    if (const uint32_t k = ctx->knobs[K_TEST_THROW]) { ctx->knobs.v[K_TEST_THROW] = 0u; throw_for_test(k); }   // (tests: the exception barrier)
    // Dirty tracking (the reference re-uploads everything every frame, README.md:17 lists that as
    // future work): identical bytes as the last successful upload -> nothing to do, and the tile-cost
    // history stays valid.  BRT_NO_DIRTY_TRACKING=1 disables.
    const size_t mb = (size_t)n_models * sizeof(Model), tb = (size_t)n_materials * sizeof(Material),
                 bb = (bvh_nodes ? (size_t)n_nodes : 0) * sizeof(BVHNode);
    auto same = [](const std::vector<char>& v, const void* p, size_t n) {
        return v.size() == n && (n == 0 || (p && std::memcmp(v.data(), p, n) == 0));
    };
    if (!rebuild && ctx->has_scene && ctx->knobs[K_NO_DIRTY_TRACKING] == 0 && same(ctx->last_models, models, mb) &&
        same(ctx->last_materials, materials, tb) && same(ctx->last_bvh, bvh_nodes, bb))
        return BRT_OK;
    // a frame of an asynchronous entry point (caller's stream) may still be reading the resident scene, or the build scratch: the
    // scene buffers are rewritten only once every device of the context has drained
    for (auto& dc : ctx->devs) {
        HIP_TRY(ctx, hipSetDevice(dc.device));
        for (hipEvent_t e : {dc.ev_last, dc.ev_copy, dc.ev_asm}) HIP_TRY(ctx, hipEventSynchronize(e));
    }
    ctx->has_scene = false;
    // the scale of the scene and its big spheres, for the reach a camera needs (brt_sah.h; ensure_tree_reach)
    float reach = 0.0f;
    if (!rebuild) {
        ctx->tree_scene = tree_scene_of(static_cast<const Model*>(models), models ? n_models : 0u);
        ctx->tree_callee_sah = false;
        ctx->tree_level = 0;
        ctx->tree_reach = 0.0f;
    } else {
        reach = tree_reach_of(ctx->tree_scene.scale, level);
    }
    std::vector<BVHNode> built;
    const BVHNode* nodes = static_cast<const BVHNode*>(bvh_nodes);
    if (n_models > 0 && models && (!bvh_nodes || n_nodes == 0)) {
        // no BVH from the caller: build it here, on the GPU.  Default: the binned-SAH tree of brt_sah.h up to kMaxSahModels spheres
        // (fewer node visits per ray than PLOC: DESIGN.md section 9), PLOC above that or with the knob BRT_BVH_QUALITY=0.
        // BRT_CPU_BVH=1: the CPU twin of either (the same bytes).
        const bool sah = ctx->knobs[K_BVH_QUALITY] != 0u && n_models <= kMaxSahModels;
        const bool on_gpu = ctx->knobs[K_CPU_BVH] == 0u && n_models <= kMaxGpuBuildModels;
        int32_t rc;
        if (on_gpu) rc = build_bvh_on_device(ctx, static_cast<const Model*>(models), n_models, sah, reach, &built, nullptr);
        else rc = sah ? build_bvh_sah(static_cast<const Model*>(models), n_models, reach, &built)
                      : build_bvh_ploc(static_cast<const Model*>(models), n_models, &built);
        if (rc != BRT_OK) return rc;
        ctx->tree_callee_sah = sah;
        ctx->tree_level = level;
        ctx->tree_reach = reach;
        nodes = built.data();
        n_nodes = (uint32_t)built.size();
    }
    std::string err;
    int32_t rc = validate_and_encode(static_cast<const Model*>(models), n_models, static_cast<const Material*>(materials),
                                     n_materials, nodes, n_nodes, &ctx->enc, &err);
    if (rc != BRT_OK) return ctx_fail(ctx, rc, err);
    const EncodedScene& e = ctx->enc;

    // one blob per device, sections 256-byte aligned
    struct Sec { const void* src; size_t bytes; size_t off; };
    Sec secs[6] = {
        {e.pairs.data(), e.pairs.size() * 4, 0}, {e.spheres.data(), e.spheres.size() * 4, 0},
        {e.sphere_material.data(), e.sphere_material.size() * 4, 0}, {e.materials.data(), e.materials.size() * 4, 0},
        {e.leaf_table.data(), e.leaf_table.size() * 4, 0}, {e.sphere_mats.data(), e.sphere_mats.size() * 4, 0}};
    size_t total = 0;
    for (auto& s : secs) { s.off = total; total += align256(s.bytes ? s.bytes : 16); }

    // A tree of the same shape as the one a device last counted the record visits of (an animated scene: the reference re-extracts
    // and re-uploads every frame, extract.rs:299-336, and the spheres move a little) goes up in THAT numbering straight away: the
    // records the view uses stay the ones the LDS tile holds (apply_hot_order), without a pre-pass per upload
    const uint64_t shape = (ctx->knobs[K_HOT_RECORDS] != 0u && !rebuild && e.desc16 && e.simple_tree && e.n_pairs > 64u) ? tree_shape_hash(e) : 0ull;
    ctx->tree_epoch++;          // (the numbering on the devices is the encoder's again unless a device re-applies its own below)
    for (auto& dc : ctx->devs) {
        HIP_TRY(ctx, hipSetDevice(dc.device));
        int32_t r2 = ensure(ctx, &dc.d_scene, &dc.scene_cap, total);
        if (r2 != BRT_OK) return r2;
        const bool reuse = shape != 0ull && dc.hot_shape == shape && dc.hot_records != 0u && dc.h_total_rank.size() == e.n_pairs &&
                           dc.h_total_srank.size() == e.n_models;
        uint32_t root = e.root_desc;
        Sec mine[6];
        for (int q = 0; q < 6; q++) mine[q] = secs[q];
        if (reuse) {
            permute_scene(e.pairs, e.spheres, e.sphere_material, e.sphere_mats, e.root_desc, dc.h_total_rank, dc.h_total_srank, &dc.h_pairs_cur,
                          &dc.h_spheres_cur, &dc.h_sphmat_cur, &dc.h_sphmats_cur, &root);
            mine[0].src = dc.h_pairs_cur.data(); mine[1].src = dc.h_spheres_cur.data(); mine[2].src = dc.h_sphmat_cur.data(); mine[5].src = dc.h_sphmats_cur.data();
        }
        for (auto& s : mine)
            if (s.bytes) HIP_TRY(ctx, hipMemcpyAsync(dc.d_scene + s.off, s.src, s.bytes, hipMemcpyHostToDevice, dc.stream));
        const uint32_t kept_hot = reuse ? dc.hot_records : 0u;
        DeviceSceneView v{};
        v.pairs = reinterpret_cast<const float*>(dc.d_scene + secs[0].off);
        v.spheres = reinterpret_cast<const float*>(dc.d_scene + secs[1].off);
        v.sphere_material = reinterpret_cast<const uint32_t*>(dc.d_scene + secs[2].off);
        v.materials = reinterpret_cast<const float*>(dc.d_scene + secs[3].off);
        v.leaf_table = reinterpret_cast<const uint32_t*>(dc.d_scene + secs[4].off);
        v.sphere_mats = reinterpret_cast<const float*>(dc.d_scene + secs[5].off);
        v.n_pairs = e.n_pairs;
        v.n_models = e.n_models;
        v.n_materials = e.n_materials;
        v.n_leaf_table = (uint32_t)(e.leaf_table.size() / 2);
        v.root_desc = root;
        v.stack_entries = e.stack_entries;
        v.desc16 = e.desc16 ? 1u : 0u;
        v.simple_tree = e.simple_tree ? 1u : 0u;
        v.boxes_ordered = e.boxes_ordered ? 1u : 0u;
        dc.view = v;
        dc.hot_records = kept_hot;
        if (reuse) dc.hot_tree = ctx->tree_epoch;
        dc.frames_since_upload = 0u;
    }
    for (auto& dc : ctx->devs) {
        HIP_TRY(ctx, hipSetDevice(dc.device));
        HIP_TRY(ctx, hipStreamSynchronize(dc.stream));  // the caller's vectors are no longer referenced
    }
    ctx->has_scene = true;
    if (rebuild) {                                            // (the bytes are ctx->last_* themselves; centre, dirty tracking: unchanged)
        ctx->tree_rebuilds++;
        // another tree: its records have no visit counts yet -- the next frame of a view is a first frame again (pre-pass), as after
        // the camera jump that usually comes with a rebuild
        if (ctx->enc.n_models != 0u && !ctx->enc.pairs.empty())
            for (auto& dc : ctx->devs) if (dc.hot_tree != 0u) { dc.order_valid = false; dc.view_rays = 0; }
        return BRT_OK;
    }
    ctx->last_models.assign(static_cast<const char*>(models), static_cast<const char*>(models) + mb);
    ctx->last_materials.assign(static_cast<const char*>(materials), static_cast<const char*>(materials) + tb);
    if (bb) ctx->last_bvh.assign(static_cast<const char*>(bvh_nodes), static_cast<const char*>(bvh_nodes) + bb);
    else ctx->last_bvh.clear();
    ctx->scene_epoch++;
    {
        double c[3] = {0, 0, 0};
        uint32_t cnt = 0;
        const Model* m = static_cast<const Model*>(models);
        for (uint32_t i = 0; i < n_models; i++) {
            const float* q = m[i].position;
            if (!(std::fabs(m[i].radius) <= 100.0f) || !std::isfinite(q[0]) || !std::isfinite(q[1]) || !std::isfinite(q[2])) continue;
            c[0] += q[0]; c[1] += q[1]; c[2] += q[2];
            cnt++;
        }
        for (int k = 0; k < 3; k++) ctx->scene_centre[k] = cnt ? (float)(c[k] / cnt) : 0.0f;
    }
    // the dispatch order of the views rendered so far stays in use as a hint, but is measured again within kLptAfterUpload
    // frames (a scene that changes every frame: every kLptAfterUpload-th frame is a measuring frame)
    // (a scene with a different number of spheres is a different scene, not the next frame of an animation: its views start
    //  from scratch, with a pre-pass)
    const bool same_shape = ctx->last_n_models == n_models;
    ctx->last_n_models = n_models;
    for (auto& dc : ctx->devs) {
        if (!same_shape) { dc.order_valid = false; dc.view_rays = 0; }
        else if (dc.order_valid && (dc.remeasure_in == 0u || dc.remeasure_in > kLptAfterUpload)) dc.remeasure_in = kLptAfterUpload;
    }
    return BRT_OK;
}

}  // namespace

namespace brt {

// ---- strip table (brt_set_strip_table) ------------------------------------------------------------------------------
// Strips go to parts by `s mod N` unless the caller sets a table.  A table permutes the parts INSIDE every group of N consecutive strips,
// so a part still has exactly one strip per group: its k-th local strip lies in group k, tiles keep their size and layout, the ONE gather
// stays as it is, and only two lookups change -- the kernel's "local strip -> frame strip" (FrameParams::strip_of) and the assembly's
// "frame strip -> part" (k_deinterleave).  Pixels cannot change: a pixel's seed depends on its frame coordinates (raytrace.wgsl:95).
// brt_plan_strips makes such a table from measured costs: the frame is rendered once at a few samples per pixel on THIS device with the
// per-tile ray counts switched on, and the strips of every group are dealt out, dearest strip to the part with the least so far.  Every
// rank computes the same table from the same integers (the kernel is deterministic), so no second collective is needed.
bool strip_table_valid(const uint32_t* t, uint32_t n_strips, uint32_t n_parts) {
    if (n_parts == 0u || n_parts > 64u) return false;
    for (uint32_t g = 0; g * n_parts < n_strips; g++) {
        uint64_t seen = 0;
        for (uint32_t s = g * n_parts; s < n_strips && s < (g + 1u) * n_parts; s++) {
            if (t[s] >= n_parts || (seen >> t[s]) & 1ull) return false;
            seen |= 1ull << t[s];
        }
    }
    return true;
}
// the device copies for `part` on device dc (made once per table and part): *part_of_strip for the assembly, fp->strip_of for the kernel
int32_t strip_table_attach(brt_ctx* ctx, DeviceCtx& dc, FrameParams* fp, const uint32_t** part_of_strip, hipStream_t stream) {
    if (part_of_strip) *part_of_strip = nullptr;
    const uint32_t strips = fp ? (fp->height + BRT_STRIP_ROWS - 1u) / BRT_STRIP_ROWS : (uint32_t)ctx->strip_part.size();
    const uint32_t n_parts = fp ? fp->n_parts : ctx->strip_n_parts;
    if (ctx->strip_part.empty() || ctx->strip_n_parts != n_parts || ctx->strip_part.size() != strips || n_parts < 2u) return BRT_OK;
    const uint32_t part = fp ? fp->part : 0u, local = (strips + n_parts - 1u) / n_parts;
    if (dc.strip_epoch != ctx->strip_epoch || dc.strip_part != part || !dc.d_strip_table) {
        std::vector<uint32_t> h(ctx->strip_part);
        h.resize((size_t)strips + local, 0xffffffu);                           // strip_of: a group without a strip of this part: padding
        for (uint32_t s = 0; s < strips; s++)
            if (ctx->strip_part[s] == part) h[(size_t)strips + s / n_parts] = s;
        int32_t rc = ensure(ctx, &dc.d_strip_table, &dc.strip_table_cap, h.size() * 4u);
        if (rc != BRT_OK) return rc;
        HIP_TRY(ctx, hipMemcpyAsync(dc.d_strip_table, h.data(), h.size() * 4u, hipMemcpyHostToDevice, stream));
        HIP_TRY(ctx, hipStreamSynchronize(stream));                            // (h is a local; once per table)
        dc.strip_epoch = ctx->strip_epoch;
        dc.strip_part = part;
    }
    const uint32_t* d = reinterpret_cast<const uint32_t*>(dc.d_strip_table);
    if (part_of_strip) *part_of_strip = d;
    if (fp) fp->strip_of = d + strips;
    return BRT_OK;
}
uint64_t part_pixels_table(const brt_ctx* ctx, const FrameParams& fp) {
    uint64_t rows = 0;
    for (uint32_t s = 0; s < (uint32_t)ctx->strip_part.size(); s++)
        if (ctx->strip_part[s] == fp.part) {
            const uint32_t r0 = s * BRT_STRIP_ROWS;
            rows += (r0 + BRT_STRIP_ROWS <= fp.height) ? BRT_STRIP_ROWS : (fp.height - r0);
        }
    return rows * fp.width;
}

}  // namespace brt

extern "C" {

uint32_t brt_tile_rows(uint32_t height, uint32_t n_parts) {
    if (n_parts == 0) return 0;
    const uint32_t strips = (height + BRT_STRIP_ROWS - 1u) / BRT_STRIP_ROWS;
    return ((strips + n_parts - 1u) / n_parts) * BRT_STRIP_ROWS;
}

}  // extern "C"

namespace {

// After a failure inside brt_render some devices may still be tracing, or copying into the caller's
// (possibly page-locked) frame: wait for every stream of the context before the error is returned, so that
// nothing of this call is in flight when the caller gets its buffers back.  The first error message stays.
void drain_all_streams(brt_ctx* ctx) {
    const std::string keep = ctx->last_error;
    for (auto& dc : ctx->devs) {
        if (!dc.stream) continue;
        if (hipSetDevice(dc.device) != hipSuccess) continue;
        (void)hipStreamSynchronize(dc.stream);
    }
    (void)hipGetLastError();
    ctx->last_error = keep;
    g_last_error = keep;
}

int32_t render_part_device(brt_ctx* ctx, const void* camera80, const void* window16, uint32_t level, uint32_t width,
                               uint32_t height, uint32_t part, uint32_t n_parts, const float* d_raster_rgba,
                               const float* d_raster_depth, float* d_out_tile, void* hip_stream, uint32_t flags,
                               brt_stats* stats) {
    const auto t0 = std::chrono::steady_clock::now();
    FrameParams fp;
    int32_t rc = make_frame_params(ctx, camera80, window16, level, width, height, part, n_parts, &fp);
    if (rc != BRT_OK) return rc;
    DeviceCtx& dc = ctx->devs[0];
    HIP_TRY(ctx, hipSetDevice(dc.device));
    const bool own_stream = (hip_stream == nullptr) && !(flags & BRT_FLAG_CALLER_STREAM);
    hipStream_t stream = own_stream ? dc.stream : static_cast<hipStream_t>(hip_stream);
    rc = strip_table_attach(ctx, dc, &fp, nullptr, stream);        // the context's strip table, if it is one for this frame and split
    if (rc != BRT_OK) return rc;
    bool prepass_ran = false;
    if (own_stream) {
        rc = prepass_order(ctx, dc, fp, d_raster_rgba, d_raster_depth, d_out_tile, stream, flags, &prepass_ran);
        if (rc != BRT_OK) return rc;
    }
    rc = attach_tile_order(ctx, dc, fp, stream, own_stream, flags);
    if (rc != BRT_OK) return rc;
    LaunchPlan lp{};
    rc = launch_part(ctx, dc, fp, d_raster_rgba, d_raster_depth, d_out_tile, stream, flags, own_stream, &lp);
    if (rc != BRT_OK) return rc;
    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        stats->paths = (fp.strip_of ? part_pixels_table(ctx, fp) : part_pixels(fp)) * (uint64_t)fp.sample_count;
        stats->lds_bytes = (uint32_t)lp.lds_bytes;
        stats->scene_in_lds = lp.scene_mode == SCENE_LDS ? 1u : (lp.scene_mode == SCENE_LDS_TOP ? 2u : 0u);
        stats->n_workgroups = lp.grid;
        stats->threads_per_workgroup = lp.block;
        stats->kernel_variant = lp.variant;
        stats->measured_tile_costs = lp.measured;
        stats->hot_records = dc.hot_tree == ctx->tree_epoch ? dc.hot_records : 0u;
    }
    if (own_stream) {
        // the order of the next frames is built on the same stream behind the frame, BEFORE the one synchronisation of this
        // call (read_counters): nothing of this call is in flight when it returns
        rc = update_tile_order(ctx, dc, fp, stream);
        if (rc != BRT_OK) return rc;
        brt_stats tmp{};
        rc = read_counters(ctx, dc, stream, &tmp);  // synchronises
        if (rc != BRT_OK) return rc;
        dc.view_rays = tmp.rays;
        view_key_of(ctx, fp, dc.view_key);
        float ms = 0.0f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, dc.ev0, dc.ev1));
        if (stats) {
            stats->rays = tmp.rays; stats->node_pops = tmp.node_pops; stats->interior_visits = tmp.interior_visits;
            stats->sphere_tests = tmp.sphere_tests; stats->hits = tmp.hits;
            stats->kernel_ms = ms;
            rc = prepass_elapsed(ctx, dc, prepass_ran, &stats->prepass_ms);
            if (rc != BRT_OK) return rc;
            stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
    }
    return BRT_OK;
}

int32_t render_frame(brt_ctx* ctx, const void* camera80, const void* window16, uint32_t level, uint32_t width, uint32_t height,
                     const float* raster_rgba, const float* raster_depth, float* out_rgba, uint32_t flags, brt_stats* stats) {
    const auto t0 = std::chrono::steady_clock::now();
    const uint32_t n_parts = (uint32_t)ctx->devs.size();
    std::vector<FrameParams> fps(n_parts);
    for (uint32_t p = 0; p < n_parts; p++) {
        int32_t rc = make_frame_params(ctx, camera80, window16, level, width, height, p, n_parts, &fps[p]);
        if (rc != BRT_OK) return rc;
    }
    const uint32_t tile_rows = brt_tile_rows(height, n_parts);
    const size_t tile_bytes = (size_t)tile_rows * width * 16;
    const size_t frame_px = (size_t)width * height;
    brt_stats st{};
    LaunchPlan lp{};
    double prepass_ms = 0.0;
    std::vector<char> prepass_ran(n_parts, 0);
    const bool direct = is_pinned(ctx, out_rgba, frame_px * 16);

    // launch every device, then collect: the devices trace their strips concurrently
    for (uint32_t p = 0; p < n_parts; p++) {
        DeviceCtx& dc = ctx->devs[p];
        HIP_TRY(ctx, hipSetDevice(dc.device));
        int32_t rc = ensure(ctx, &dc.d_tile, &dc.tile_cap, tile_bytes);
        if (rc != BRT_OK) return rc;
        const float* d_rgba = nullptr;
        const float* d_depth = nullptr;
        // raster inputs: the whole frame for a one-device context; else this device's strips only, densely in the tile's own layout
        // (FrameParams::raster_dense) -- a strided 2-D copy, 1 / n_parts of the bytes over PCIe per device
        auto send = [&](const float* src, float** d_buf, size_t* cap, uint32_t fpp) -> int32_t {
            const size_t px_bytes = (size_t)fpp * 4u;
            if (n_parts == 1) {
                int32_t r = ensure(ctx, d_buf, cap, frame_px * px_bytes);
                if (r != BRT_OK) return r;
                HIP_TRY(ctx, hipMemcpyAsync(*d_buf, src, frame_px * px_bytes, hipMemcpyHostToDevice, dc.stream));
                return BRT_OK;
            }
            int32_t r = ensure(ctx, d_buf, cap, (size_t)tile_rows * width * px_bytes);
            if (r != BRT_OK) return r;
            const uint32_t strips = (height + BRT_STRIP_ROWS - 1u) / BRT_STRIP_ROWS, full = height / BRT_STRIP_ROWS;
            const size_t strip_bytes = (size_t)BRT_STRIP_ROWS * width * px_bytes;
            const uint32_t n_full = full > p ? (full - p + n_parts - 1u) / n_parts : 0u;      // whole strips p, p + n, ... < full
            if (n_full)
                HIP_TRY(ctx, hipMemcpy2DAsync(*d_buf, strip_bytes, reinterpret_cast<const char*>(src) + (size_t)p * strip_bytes,
                                              strip_bytes * n_parts, strip_bytes, n_full, hipMemcpyHostToDevice, dc.stream));
            if (strips > full && full % n_parts == p)                                       // the frame's last, partial strip is this part's
                HIP_TRY(ctx, hipMemcpyAsync(reinterpret_cast<char*>(*d_buf) + (size_t)n_full * strip_bytes,
                                            reinterpret_cast<const char*>(src) + (size_t)full * strip_bytes,
                                            (size_t)(height - full * BRT_STRIP_ROWS) * width * px_bytes, hipMemcpyHostToDevice, dc.stream));
            return BRT_OK;
        };
        if (raster_rgba) {
            rc = send(raster_rgba, &dc.d_raster_rgba, &dc.raster_rgba_cap, 4u);
            if (rc != BRT_OK) return rc;
            d_rgba = dc.d_raster_rgba;
        }
        if (raster_depth) {
            rc = send(raster_depth, &dc.d_raster_depth, &dc.raster_depth_cap, 1u);
            if (rc != BRT_OK) return rc;
            d_depth = dc.d_raster_depth;
        }
        fps[p].raster_dense = n_parts > 1 ? 1u : 0u;
        if (!direct && dc.stage_cap < tile_bytes) {
            if (dc.h_stage) HIP_TRY(ctx, hipHostFree(dc.h_stage));
            dc.h_stage = nullptr;
            dc.stage_cap = 0;
            HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void**>(&dc.h_stage), tile_bytes, hipHostMallocDefault));
            dc.stage_cap = tile_bytes;
        }
        bool ran = false;
        rc = prepass_order(ctx, dc, fps[p], d_rgba, d_depth, dc.d_tile, dc.stream, flags, &ran);
        if (rc != BRT_OK) return rc;
        prepass_ran[p] = ran;
        rc = attach_tile_order(ctx, dc, fps[p], dc.stream, true, flags);
        if (rc != BRT_OK) return rc;
        rc = launch_part(ctx, dc, fps[p], d_rgba, d_depth, dc.d_tile, dc.stream, flags, true, &lp);
        if (rc != BRT_OK) return rc;
        if (direct) {
            // page-locked destination: DMA every strip to its place in the frame, no CPU copy
            const uint32_t strips = (height + BRT_STRIP_ROWS - 1u) / BRT_STRIP_ROWS;
            for (uint32_t s = p, k = 0; s < strips; s += n_parts, k++) {
                const uint32_t r0 = s * BRT_STRIP_ROWS;
                const uint32_t rows = (r0 + BRT_STRIP_ROWS <= height) ? BRT_STRIP_ROWS : (height - r0);
                if (n_parts == 1) {   // the tile IS the frame: one copy
                    HIP_TRY(ctx, hipMemcpyAsync(out_rgba, dc.d_tile, frame_px * 16, hipMemcpyDeviceToHost, dc.stream));
                    break;
                }
                HIP_TRY(ctx, hipMemcpyAsync(out_rgba + (size_t)r0 * width * 4, dc.d_tile + (size_t)k * BRT_STRIP_ROWS * width * 4,
                                            (size_t)rows * width * 16, hipMemcpyDeviceToHost, dc.stream));
            }
        } else {
            HIP_TRY(ctx, hipMemcpyAsync(dc.h_stage, dc.d_tile, tile_bytes, hipMemcpyDeviceToHost, dc.stream));
        }
    }
    double kernel_ms = 0.0, gather_ms = 0.0;
    for (uint32_t p = 0; p < n_parts; p++) {
        DeviceCtx& dc = ctx->devs[p];
        HIP_TRY(ctx, hipSetDevice(dc.device));
        const uint64_t rays_before = st.rays;
        int32_t rc = update_tile_order(ctx, dc, fps[p], dc.stream);   // behind the frame, ahead of the synchronisation
        if (rc != BRT_OK) return rc;
        rc = read_counters(ctx, dc, dc.stream, &st);  // synchronises the stream
        if (rc != BRT_OK) return rc;
        dc.view_rays = st.rays - rays_before;
        view_key_of(ctx, fps[p], dc.view_key);
        float ms = 0.0f;
        HIP_TRY(ctx, hipEventElapsedTime(&ms, dc.ev0, dc.ev1));
        if (ms > kernel_ms) kernel_ms = ms;
        double pp_ms = 0.0;
        rc = prepass_elapsed(ctx, dc, prepass_ran[p] != 0, &pp_ms);
        if (rc != BRT_OK) return rc;
        if (pp_ms > prepass_ms) prepass_ms = pp_ms;
        const auto g0 = std::chrono::steady_clock::now();
        const uint32_t strips = direct ? 0u : (height + BRT_STRIP_ROWS - 1u) / BRT_STRIP_ROWS;
        for (uint32_t s = p, k = 0; s < strips; s += n_parts, k++) {
            const uint32_t r0 = s * BRT_STRIP_ROWS;
            const uint32_t rows = (r0 + BRT_STRIP_ROWS <= height) ? BRT_STRIP_ROWS : (height - r0);
            std::memcpy(out_rgba + (size_t)r0 * width * 4, dc.h_stage + (size_t)k * BRT_STRIP_ROWS * width * 4,
                        (size_t)rows * width * 16);
        }
        gather_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - g0).count();
        st.paths += part_pixels(fps[p]) * (uint64_t)fps[p].sample_count;
    }
    if (stats) {
        *stats = st;
        stats->kernel_ms = kernel_ms;
        stats->prepass_ms = prepass_ms;
        stats->gather_ms = gather_ms;
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        stats->lds_bytes = (uint32_t)lp.lds_bytes;
        stats->scene_in_lds = lp.scene_mode == SCENE_LDS ? 1u : (lp.scene_mode == SCENE_LDS_TOP ? 2u : 0u);
        stats->n_workgroups = lp.grid;
        stats->threads_per_workgroup = lp.block;
        stats->kernel_variant = lp.variant;
        stats->measured_tile_costs = lp.measured;
        stats->hot_records = ctx->devs[0].hot_tree == ctx->tree_epoch ? ctx->devs[0].hot_records : 0u;
    }
    return BRT_OK;
}


// The frame of an N-device context assembled on its FIRST device: every device traces its strips, the tiles of the
// others travel to the first device's gather buffer by peer copy (xGMI between the GPUs of a node; a plain device copy
// when an ordinal repeats), and k_deinterleave writes the frame -- what bevyray_amd/parallel.py does with one process per
// GPU and an RCCL gather, for a single-process host (the Rust node).
int32_t render_frame_device(brt_ctx* ctx, const void* camera80, const void* window16, uint32_t level, uint32_t width, uint32_t height,
                            const float* d_raster_rgba, const float* d_raster_depth, void* d_frame, void* hip_stream, uint32_t flags,
                            brt_stats* stats) {
    const auto t0 = std::chrono::steady_clock::now();
    const uint32_t n_parts = (uint32_t)ctx->devs.size();
    std::vector<FrameParams> fps(n_parts);
    for (uint32_t p = 0; p < n_parts; p++) {
        int32_t rc = make_frame_params(ctx, camera80, window16, level, width, height, p, n_parts, &fps[p]);
        if (rc != BRT_OK) return rc;
    }
    const uint32_t tile_rows = brt_tile_rows(height, n_parts);
    const size_t tile_floats = (size_t)tile_rows * width * 4, tile_bytes = tile_floats * 4;
    DeviceCtx& d0 = ctx->devs[0];
    HIP_TRY(ctx, hipSetDevice(d0.device));
    const bool own_stream = (hip_stream == nullptr) && !(flags & BRT_FLAG_CALLER_STREAM);
    hipStream_t stream0 = own_stream ? d0.stream : static_cast<hipStream_t>(hip_stream);
    // A previous asynchronous frame (caller's stream) may still be copying into the gather buffer or reading a tile / raster copy:
    // hipFree only synchronises the current device, so buffers grow only once every device of the context has drained.
    {
        bool grow = d0.gather_cap < tile_bytes * n_parts;
        for (uint32_t p = 1; p < n_parts; p++) {
            const DeviceCtx& dc = ctx->devs[p];
            grow = grow || dc.tile_cap < tile_bytes || (d_raster_rgba && dc.raster_rgba_cap < tile_bytes) ||
                   (d_raster_depth && level != 0u && dc.raster_depth_cap < tile_bytes / 4);
        }
        grow = grow || ((d_raster_rgba || d_raster_depth) && d0.pack_cap < (tile_bytes + tile_bytes / 4) * (n_parts - 1u));
        if (grow)
            for (auto& dc : ctx->devs) {
                HIP_TRY(ctx, hipSetDevice(dc.device));
                for (hipEvent_t e : {dc.ev_last, dc.ev_copy, dc.ev_asm}) HIP_TRY(ctx, hipEventSynchronize(e));
            }
        HIP_TRY(ctx, hipSetDevice(d0.device));
    }
    int32_t rc = ensure(ctx, &d0.d_gather, &d0.gather_cap, tile_bytes * n_parts);
    if (rc != BRT_OK) return rc;
    // the other devices start behind whatever the caller enqueued before this call (its raster inputs)
    HIP_TRY(ctx, hipEventRecord(d0.ev_in, stream0));
    // Raster inputs of the other devices: a device reads only its own strips, so only those travel -- packed per part on the first
    // device (k_pack_strips, the tile's own layout: FrameParams::raster_dense), one peer copy per device and input: 1 / n_parts of
    // the frame each instead of the whole frame (round 4: 41 MB at 1080p, 166 MB at 4K, x 7 devices, every frame at levels 1 / 2)
    const bool fwd_rgba = n_parts > 1 && d_raster_rgba != nullptr, fwd_depth = n_parts > 1 && d_raster_depth != nullptr && level != 0u;
    uint64_t forwarded = 0;
    float* pack_rgba = nullptr;
    float* pack_depth = nullptr;
    if (fwd_rgba || fwd_depth) {
        rc = ensure(ctx, &d0.d_pack, &d0.pack_cap, (tile_bytes + tile_bytes / 4) * (n_parts - 1u));
        if (rc != BRT_OK) return rc;
        pack_rgba = d0.d_pack;
        pack_depth = d0.d_pack + tile_floats * (n_parts - 1u);
        HIP_TRY(ctx, hipStreamWaitEvent(stream0, d0.ev_asm, 0));    // (the previous frame's devices have read the pack buffer: ev_copy sits behind their reads)
        for (uint32_t q = 1; q < n_parts; q++) HIP_TRY(ctx, hipStreamWaitEvent(stream0, ctx->devs[q].ev_last, 0));
        if (fwd_rgba) HIP_TRY(ctx, launch_pack_strips(d_raster_rgba, pack_rgba, width, height, n_parts, tile_rows, 4u, stream0));
        if (fwd_depth) HIP_TRY(ctx, launch_pack_strips(d_raster_depth, pack_depth, width, height, n_parts, tile_rows, 1u, stream0));
        HIP_TRY(ctx, hipEventRecord(d0.ev_pack, stream0));
    }
    LaunchPlan lp{};
    std::vector<char> prepass_ran(n_parts, 0);
    for (uint32_t p = 0; p < n_parts; p++) {
        DeviceCtx& dc = ctx->devs[p];
        HIP_TRY(ctx, hipSetDevice(dc.device));
        hipStream_t sp = p == 0 ? stream0 : dc.stream;
        float* out = d0.d_gather + (size_t)p * tile_floats;
        const float* d_rgba = d_raster_rgba;
        const float* d_depth = d_raster_depth;
        if (p != 0) {
            rc = ensure(ctx, &dc.d_tile, &dc.tile_cap, tile_bytes);
            if (rc != BRT_OK) return rc;
            out = dc.d_tile;
            HIP_TRY(ctx, hipStreamWaitEvent(sp, d0.ev_in, 0));
            HIP_TRY(ctx, hipStreamWaitEvent(sp, d0.ev_asm, 0));     // the gather buffer is free again (previous frame assembled)
            // (level 0 reads the raster colour too -- k_passthrough, raytrace.wgsl:97-99 -- so it is forwarded at every level: a
            //  device must never be handed a pointer into another device's memory, peer access is not enabled)
            if (fwd_rgba || fwd_depth) HIP_TRY(ctx, hipStreamWaitEvent(sp, d0.ev_pack, 0));
            if (fwd_rgba) {
                rc = ensure(ctx, &dc.d_raster_rgba, &dc.raster_rgba_cap, tile_bytes);
                if (rc != BRT_OK) return rc;
                HIP_TRY(ctx, hipMemcpyPeerAsync(dc.d_raster_rgba, dc.device, pack_rgba + (size_t)(p - 1u) * tile_floats, d0.device, tile_bytes, sp));
                d_rgba = dc.d_raster_rgba;
                forwarded += tile_bytes;
            }
            if (fwd_depth) {
                rc = ensure(ctx, &dc.d_raster_depth, &dc.raster_depth_cap, tile_bytes / 4);
                if (rc != BRT_OK) return rc;
                HIP_TRY(ctx, hipMemcpyPeerAsync(dc.d_raster_depth, dc.device, pack_depth + (size_t)(p - 1u) * (tile_floats / 4), d0.device, tile_bytes / 4, sp));
                d_depth = dc.d_raster_depth;
                forwarded += tile_bytes / 4;
            } else if (level == 0u) {
                d_depth = nullptr;
            }
            fps[p].raster_dense = 1u;
        } else {
            HIP_TRY(ctx, hipStreamWaitEvent(sp, d0.ev_asm, 0));
        }
        if (own_stream) {
            bool ran = false;
            rc = prepass_order(ctx, dc, fps[p], d_rgba, d_depth, out, sp, flags, &ran);
            if (rc != BRT_OK) return rc;
            prepass_ran[p] = ran;
        }
        rc = attach_tile_order(ctx, dc, fps[p], sp, own_stream, flags);
        if (rc != BRT_OK) return rc;
        rc = launch_part(ctx, dc, fps[p], d_rgba, d_depth, out, sp, flags, true, &lp);
        if (rc != BRT_OK) return rc;
        if (p != 0) {
            HIP_TRY(ctx, hipMemcpyPeerAsync(d0.d_gather + (size_t)p * tile_floats, d0.device, dc.d_tile, dc.device, tile_bytes, sp));
            HIP_TRY(ctx, hipEventRecord(dc.ev_copy, sp));
            HIP_TRY(ctx, hipEventRecord(dc.ev_last, sp));
        }
    }
    HIP_TRY(ctx, hipSetDevice(d0.device));
    HIP_TRY(ctx, hipEventRecord(d0.ev_g0, stream0));
    for (uint32_t p = 1; p < n_parts; p++) HIP_TRY(ctx, hipStreamWaitEvent(stream0, ctx->devs[p].ev_copy, 0));
    HIP_TRY(ctx, launch_deinterleave(d0.d_gather, d_frame, width, height, n_parts, tile_rows, flags & BRT_FLAG_OUT_MASK, stream0));
    HIP_TRY(ctx, hipEventRecord(d0.ev_g1, stream0));
    HIP_TRY(ctx, hipEventRecord(d0.ev_asm, stream0));
    HIP_TRY(ctx, hipEventRecord(d0.ev_last, stream0));
    brt_stats st{};
    st.forwarded_bytes = forwarded;
    for (uint32_t p = 0; p < n_parts; p++) st.paths += part_pixels(fps[p]) * (uint64_t)fps[p].sample_count;
    if (own_stream) {
        double kernel_ms = 0.0, prepass_ms = 0.0;
        for (uint32_t p = 0; p < n_parts; p++) {
            DeviceCtx& dc = ctx->devs[p];
            HIP_TRY(ctx, hipSetDevice(dc.device));
            hipStream_t sp = p == 0 ? stream0 : dc.stream;
            rc = update_tile_order(ctx, dc, fps[p], sp);
            if (rc != BRT_OK) return rc;
            const uint64_t before = st.rays;
            rc = read_counters(ctx, dc, sp, &st);       // synchronises this device's stream
            if (rc != BRT_OK) return rc;
            dc.view_rays = st.rays - before;
            view_key_of(ctx, fps[p], dc.view_key);
            float ms = 0.0f;
            HIP_TRY(ctx, hipEventElapsedTime(&ms, dc.ev0, dc.ev1));
            if (ms > kernel_ms) kernel_ms = ms;
            double pp = 0.0;
            rc = prepass_elapsed(ctx, dc, prepass_ran[p] != 0, &pp);
            if (rc != BRT_OK) return rc;
            if (pp > prepass_ms) prepass_ms = pp;
        }
        float gms = 0.0f;
        HIP_TRY(ctx, hipEventElapsedTime(&gms, d0.ev_g0, d0.ev_g1));
        st.kernel_ms = kernel_ms;
        st.prepass_ms = prepass_ms;
        st.gather_ms = gms;        // from the end of the first device's trace to the assembled frame (waits for the slowest device)
    }
    if (stats) {
        *stats = st;
        stats->total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        stats->lds_bytes = (uint32_t)lp.lds_bytes;
        stats->scene_in_lds = lp.scene_mode == SCENE_LDS ? 1u : (lp.scene_mode == SCENE_LDS_TOP ? 2u : 0u);
        stats->n_workgroups = lp.grid;
        stats->threads_per_workgroup = lp.block;
        stats->kernel_variant = lp.variant;
        stats->measured_tile_costs = lp.measured;
        stats->hot_records = ctx->devs[0].hot_tree == ctx->tree_epoch ? ctx->devs[0].hot_records : 0u;
    }
    return BRT_OK;
}

}  // namespace

extern "C" {

int32_t brt_render(brt_ctx* ctx, const void* camera80, const void* window16, uint32_t level, uint32_t width, uint32_t height,
                   const float* raster_rgba, const float* raster_depth, float* out_rgba, uint32_t flags, brt_stats* stats) {
    return guard(ctx ? &ctx->last_error : nullptr, [&]() -> int32_t {
    if (!ctx) return fail(BRT_ERR_INVALID_ARGUMENT, "ctx is null");
    if (!out_rgba) return ctx_fail(ctx, BRT_ERR_INVALID_ARGUMENT, "out_rgba is null");
    if (!ctx->has_scene && level != 0u) return ctx_fail(ctx, BRT_ERR_NO_SCENE, "brt_upload_scene has not succeeded yet");
    if (flags & BRT_FLAG_OUT_MASK) return ctx_fail(ctx, BRT_ERR_UNSUPPORTED, "brt_render writes RGBA f32 (BRT_FLAG_OUT_* apply to the device frame of brt_render_device / brt_gather_rccl / brt_deinterleave_device)");
    uint32_t rebuilt = 0u;
    int32_t rc = level != 0u ? ensure_tree_reach(ctx, camera80, &rebuilt) : BRT_OK;
    if (rc == BRT_OK) rc = render_frame(ctx, camera80, window16, level, width, height, raster_rgba, raster_depth, out_rgba, flags, stats);
    if (rc != BRT_OK) drain_all_streams(ctx);
    else tree_stats(ctx, rebuilt, stats);
    return rc;
    });
}

int32_t brt_render_part_device(brt_ctx* ctx, const void* camera80, const void* window16, uint32_t level, uint32_t width,
                               uint32_t height, uint32_t part, uint32_t n_parts, const float* d_raster_rgba,
                               const float* d_raster_depth, float* d_out_tile, void* hip_stream, uint32_t flags,
                               brt_stats* stats) {
    return guard(ctx ? &ctx->last_error : nullptr, [&]() -> int32_t {
    if (!ctx) return fail(BRT_ERR_INVALID_ARGUMENT, "ctx is null");
    if (!d_out_tile) return ctx_fail(ctx, BRT_ERR_INVALID_ARGUMENT, "d_out_tile is null");
    if (!ctx->has_scene && level != 0u) return ctx_fail(ctx, BRT_ERR_NO_SCENE, "brt_upload_scene has not succeeded yet");
    if (flags & BRT_FLAG_OUT_MASK) return ctx_fail(ctx, BRT_ERR_UNSUPPORTED, "a rank's tile is RGBA f32 (the format is applied where the frame is assembled: brt_gather_rccl / brt_deinterleave_device)");
    uint32_t rebuilt = 0u;
    int32_t rc = level != 0u ? ensure_tree_reach(ctx, camera80, &rebuilt) : BRT_OK;
    if (rc == BRT_OK) rc = render_part_device(ctx, camera80, window16, level, width, height, part, n_parts, d_raster_rgba, d_raster_depth,
                                              d_out_tile, hip_stream, flags, stats);
    // a failed call leaves nothing in flight on the context's own streams (a caller's stream is the caller's to drain)
    if (rc != BRT_OK) drain_all_streams(ctx);
    else tree_stats(ctx, rebuilt, stats);
    return rc;
    });
}

int32_t brt_render_device(brt_ctx* ctx, const void* camera80, const void* window16, uint32_t level, uint32_t width, uint32_t height,
                          const float* d_raster_rgba, const float* d_raster_depth, void* d_frame, void* hip_stream, uint32_t flags,
                          brt_stats* stats) {
    return guard(ctx ? &ctx->last_error : nullptr, [&]() -> int32_t {
    if (!ctx) return fail(BRT_ERR_INVALID_ARGUMENT, "ctx is null");
    if (!d_frame) return ctx_fail(ctx, BRT_ERR_INVALID_ARGUMENT, "d_frame is null");
    if (!ctx->has_scene && level != 0u) return ctx_fail(ctx, BRT_ERR_NO_SCENE, "brt_upload_scene has not succeeded yet");
    if (flags & BRT_FLAG_KERNEL_SIMPLE) return ctx_fail(ctx, BRT_ERR_UNSUPPORTED, "brt_render_device runs the persistent kernel only");
    uint32_t rebuilt = 0u;
    int32_t rc = level != 0u ? ensure_tree_reach(ctx, camera80, &rebuilt) : BRT_OK;
    if (rc == BRT_OK) rc = render_frame_device(ctx, camera80, window16, level, width, height, d_raster_rgba, d_raster_depth, d_frame, hip_stream,
                                               flags, stats);
    if (rc != BRT_OK) drain_all_streams(ctx);
    else tree_stats(ctx, rebuilt, stats);
    return rc;
    });
}

int32_t brt_set_strip_table(brt_ctx* ctx, uint32_t n_parts, uint32_t n_strips, const uint32_t* part_of_strip) {
    return guard(ctx ? &ctx->last_error : nullptr, [&]() -> int32_t {
    if (!ctx) return fail(BRT_ERR_INVALID_ARGUMENT, "ctx is null");
    ctx->strip_epoch++;
    if (!part_of_strip) { ctx->strip_part.clear(); ctx->strip_n_parts = 0u; return BRT_OK; }
    if (n_parts < 1u || n_strips < 1u || n_strips > 4096u) return ctx_fail(ctx, BRT_ERR_INVALID_ARGUMENT, "n_parts / n_strips out of range");
    if (!strip_table_valid(part_of_strip, n_strips, n_parts))
        return ctx_fail(ctx, BRT_ERR_INVALID_ARGUMENT, "strip table: every group of n_parts consecutive strips must hold each part at most once");
    ctx->strip_part.assign(part_of_strip, part_of_strip + n_strips);
    ctx->strip_n_parts = n_parts;
    return BRT_OK;
    });
}

int32_t brt_plan_strips(brt_ctx* ctx, const void* camera80, const void* window16, uint32_t level, uint32_t width, uint32_t height,
                        uint32_t n_parts, uint32_t probe_spp, uint32_t* out_part_of_strip) {
    return guard(ctx ? &ctx->last_error : nullptr, [&]() -> int32_t {
    if (!ctx) return fail(BRT_ERR_INVALID_ARGUMENT, "ctx is null");
    if (!ctx->has_scene) return ctx_fail(ctx, BRT_ERR_NO_SCENE, "no scene uploaded");
    if (n_parts < 1u || n_parts > 64u || probe_spp < 1u) return ctx_fail(ctx, BRT_ERR_INVALID_ARGUMENT, "n_parts must be 1..64, probe_spp >= 1");
    FrameParams fp;
    int32_t rc = make_frame_params(ctx, camera80, window16, level == 0u ? 3u : level, width, height, 0u, 1u, &fp);
    if (rc != BRT_OK) return rc;
    fp.sample_count = probe_spp;
    fp.spp_f = (float)probe_spp;
    DeviceCtx& dc = ctx->devs[0];
    HIP_TRY(ctx, hipSetDevice(dc.device));
    const uint32_t n_tiles = fp.local_strips * fp.tiles_x, strips = fp.local_strips;
    rc = ensure(ctx, &dc.d_tile, &dc.tile_cap, (size_t)strips * BRT_STRIP_ROWS * width * 16u);
    if (rc != BRT_OK) return rc;
    rc = ensure(ctx, &dc.d_tile_cost, &dc.tile_cost_cap, (size_t)n_tiles * 8u);
    if (rc != BRT_OK) return rc;
    HIP_TRY(ctx, hipStreamWaitEvent(dc.stream, dc.ev_last, 0));
    HIP_TRY(ctx, hipMemsetAsync(dc.d_tile_cost, 0, (size_t)n_tiles * 8u, dc.stream));
    fp.tile_cost = dc.d_tile_cost;                              // ray sums per tile (then maxima)
    rc = launch_part(ctx, dc, fp, nullptr, nullptr, dc.d_tile, dc.stream, 0u, false, nullptr);
    if (rc != BRT_OK) return rc;
    std::vector<uint32_t> cost(n_tiles);
    HIP_TRY(ctx, hipMemcpyAsync(cost.data(), dc.d_tile_cost, (size_t)n_tiles * 4u, hipMemcpyDeviceToHost, dc.stream));
    HIP_TRY(ctx, hipStreamSynchronize(dc.stream));
    dc.costs_valid = false;                                     // (the cost buffer no longer holds a view's measurement)
    std::vector<uint64_t> strip_cost(strips, 0);
    for (uint32_t t = 0; t < n_tiles; t++) strip_cost[t / fp.tiles_x] += cost[t];
    std::vector<uint32_t> table(strips, 0u);
    plan_strip_table(strip_cost.data(), strips, n_parts, table.data());          // the rule: brt_host.cpp
    if (out_part_of_strip) std::memcpy(out_part_of_strip, table.data(), (size_t)strips * 4u);
    ctx->strip_epoch++;
    ctx->strip_part = table;
    ctx->strip_n_parts = n_parts;
    return BRT_OK;
    });
}

int32_t brt_deinterleave_device(brt_ctx* ctx, const float* d_tiles, uint32_t n_parts, uint32_t width, uint32_t height,
                                void* d_frame, void* hip_stream, uint32_t flags) {
    return guard(ctx ? &ctx->last_error : nullptr, [&]() -> int32_t {
    if (!ctx) return fail(BRT_ERR_INVALID_ARGUMENT, "ctx is null");
    if (!d_tiles || !d_frame || n_parts == 0) return ctx_fail(ctx, BRT_ERR_INVALID_ARGUMENT, "null buffer / n_parts == 0");
    DeviceCtx& dc = ctx->devs[0];
    HIP_TRY(ctx, hipSetDevice(dc.device));
    const bool own_stream = (hip_stream == nullptr) && !(flags & BRT_FLAG_CALLER_STREAM);
    hipStream_t stream = own_stream ? dc.stream : static_cast<hipStream_t>(hip_stream);
    const uint32_t* part_of_strip = nullptr;
    {
        FrameParams key{};                                        // (which frame and split: the table must be one for them)
        key.height = height; key.n_parts = n_parts; key.part = 0u;
        int32_t rc = strip_table_attach(ctx, ctx->devs[0], &key, &part_of_strip, stream);
        if (rc != BRT_OK) return rc;
    }
    HIP_TRY(ctx, launch_deinterleave(d_tiles, d_frame, width, height, n_parts, brt_tile_rows(height, n_parts), flags & BRT_FLAG_OUT_MASK, stream, part_of_strip));
    if (own_stream) HIP_TRY(ctx, hipStreamSynchronize(stream));
    return BRT_OK;
    });
}

int32_t brt_build_bvh_device(brt_ctx* ctx, const void* models, uint32_t n_models, void* out_nodes, uint32_t capacity,
                             uint32_t* out_n_nodes, double* out_build_ms) {
    return guard(ctx ? &ctx->last_error : nullptr, [&]() -> int32_t {
    if (!ctx) return fail(BRT_ERR_INVALID_ARGUMENT, "ctx is null");
    if (!out_n_nodes) return ctx_fail(ctx, BRT_ERR_INVALID_ARGUMENT, "out_n_nodes is null");
    *out_n_nodes = 0;
    if (n_models == 0) return BRT_OK;
    if (!models) return ctx_fail(ctx, BRT_ERR_INVALID_ARGUMENT, "models is null");
    std::vector<BVHNode> nodes;
    int32_t rc = build_bvh_on_device(ctx, static_cast<const Model*>(models), n_models, false, 0.0f, &nodes, out_build_ms);
    if (rc != BRT_OK) return rc;
    *out_n_nodes = (uint32_t)nodes.size();
    if (nodes.size() > capacity || !out_nodes)
        return ctx_fail(ctx, BRT_ERR_CAPACITY, "BVH needs " + std::to_string(nodes.size()) + " nodes, capacity " + std::to_string(capacity));
    std::memcpy(out_nodes, nodes.data(), nodes.size() * sizeof(BVHNode));
    return BRT_OK;
    });
}

int32_t brt_build_bvh_sah_device(brt_ctx* ctx, const void* models, uint32_t n_models, float reach, void* out_nodes, uint32_t capacity,
                                 uint32_t* out_n_nodes, double* out_build_ms) {
    return guard(ctx ? &ctx->last_error : nullptr, [&]() -> int32_t {
    if (!ctx) return fail(BRT_ERR_INVALID_ARGUMENT, "ctx is null");
    if (!out_n_nodes) return ctx_fail(ctx, BRT_ERR_INVALID_ARGUMENT, "out_n_nodes is null");
    *out_n_nodes = 0;
    if (n_models == 0) return BRT_OK;
    if (!models) return ctx_fail(ctx, BRT_ERR_INVALID_ARGUMENT, "models is null");
    std::vector<BVHNode> nodes;
    int32_t rc = build_bvh_on_device(ctx, static_cast<const Model*>(models), n_models, true, reach, &nodes, out_build_ms);
    if (rc != BRT_OK) return rc;
    *out_n_nodes = (uint32_t)nodes.size();
    if (nodes.size() > capacity || !out_nodes)
        return ctx_fail(ctx, BRT_ERR_CAPACITY, "BVH needs " + std::to_string(nodes.size()) + " nodes, capacity " + std::to_string(capacity));
    std::memcpy(out_nodes, nodes.data(), nodes.size() * sizeof(BVHNode));
    return BRT_OK;
    });
}

int32_t brt_debug_profile(brt_ctx* ctx, uint64_t* out64) {
    return guard(ctx ? &ctx->last_error : nullptr, [&]() -> int32_t {
    if (!ctx || !out64) return fail(BRT_ERR_INVALID_ARGUMENT, "null pointer");
    DeviceCtx& dc = ctx->devs[0];
    HIP_TRY(ctx, hipSetDevice(dc.device));
    HIP_TRY(ctx, hipMemcpy(out64, dc.d_ctrl, 512, hipMemcpyDeviceToHost));
    // [40], [41]: critical tiles and longest pixel (rays) of the view's last MEASURED frame, when the order was built on the GPU
    out64[40] = out64[41] = 0;
    out64[42] = (dc.order_valid && !dc.order_on_device) ? dc.order_split : 0u;      // [42]: tiles handed out as two half-sample jobs
    if (dc.order_valid && dc.order_on_device && dc.d_order_meta) {
        uint32_t meta[4] = {0u, 0u, 0u, 0u};
        HIP_TRY(ctx, hipStreamSynchronize(dc.stream));
        HIP_TRY(ctx, hipMemcpy(meta, dc.d_order_meta, sizeof meta, hipMemcpyDeviceToHost));
        out64[40] = meta[0];
        out64[41] = meta[1];
        out64[42] = meta[3];
    }
    return BRT_OK;
    });
}

int32_t brt_debug_tile_order(brt_ctx* ctx, const uint32_t* ray_sum, const uint32_t* longest_pixel, uint32_t n_tiles,
                             uint32_t sample_count, uint64_t grid_lanes, uint32_t tiles_x, uint32_t dilate, uint32_t split_tail,
                             uint32_t* out_order, uint32_t* out_info4) {
    return guard(ctx ? &ctx->last_error : nullptr, [&]() -> int32_t {
    if (!ctx) return fail(BRT_ERR_INVALID_ARGUMENT, "ctx is null");
    if (!ray_sum || !longest_pixel || !out_order || !out_info4 || n_tiles == 0) return ctx_fail(ctx, BRT_ERR_INVALID_ARGUMENT, "null buffer / no tiles");
    if (dilate != 0u && (tiles_x == 0u || n_tiles % tiles_x != 0u)) return ctx_fail(ctx, BRT_ERR_INVALID_ARGUMENT, "dilate needs a tiles_x that divides n_tiles");
    DeviceCtx& dc = ctx->devs[0];
    HIP_TRY(ctx, hipSetDevice(dc.device));
    uint32_t* d_cost = nullptr;
    uint32_t* d_order = nullptr;
    uint32_t* d_meta = nullptr;
    char* d_scratch = nullptr;
    auto body = [&]() -> int32_t {
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&d_cost), (size_t)n_tiles * 8));
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&d_order), ((size_t)n_tiles + split_tail) * 4));
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&d_meta), 256));
        HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&d_scratch), order_scratch_bytes(n_tiles)));
        HIP_TRY(ctx, hipMemcpyAsync(d_cost, ray_sum, (size_t)n_tiles * 4, hipMemcpyHostToDevice, dc.stream));
        HIP_TRY(ctx, hipMemcpyAsync(d_cost + n_tiles, longest_pixel, (size_t)n_tiles * 4, hipMemcpyHostToDevice, dc.stream));
        const uint64_t sky_cost = (uint64_t)64 * sample_count * (1000 + 20) / 1000;
        HIP_TRY(ctx, launch_build_order(d_cost, d_cost + n_tiles, n_tiles, sky_cost, grid_lanes, tiles_x, dilate, dilate, split_tail, d_order, d_meta, d_scratch, dc.stream));
        HIP_TRY(ctx, hipMemcpyAsync(out_info4, d_meta, 16, hipMemcpyDeviceToHost, dc.stream));
        HIP_TRY(ctx, hipStreamSynchronize(dc.stream));
        HIP_TRY(ctx, hipMemcpy(out_order, d_order, ((size_t)n_tiles + out_info4[3]) * 4, hipMemcpyDeviceToHost));
        return BRT_OK;
    };
    const int32_t rc = body();
    if (d_cost) (void)hipFree(d_cost);
    if (d_order) (void)hipFree(d_order);
    if (d_meta) (void)hipFree(d_meta);
    if (d_scratch) (void)hipFree(d_scratch);
    return rc;
    });
}

int32_t brt_debug_eval(brt_ctx* ctx, uint32_t op, const float* in16, float* out8, uint32_t n) {
    return guard(ctx ? &ctx->last_error : nullptr, [&]() -> int32_t {
    if (!ctx) return fail(BRT_ERR_INVALID_ARGUMENT, "ctx is null");
    if (!in16 || !out8) return ctx_fail(ctx, BRT_ERR_INVALID_ARGUMENT, "null buffer");
    if (n == 0) return BRT_OK;
    DeviceCtx& dc = ctx->devs[0];
    HIP_TRY(ctx, hipSetDevice(dc.device));
    float* d_in = nullptr;
    float* d_out = nullptr;
    HIP_TRY(ctx, hipMalloc(reinterpret_cast<void**>(&d_in), (size_t)n * 64));
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d_out), (size_t)n * 32);
    if (e != hipSuccess) { (void)hipFree(d_in); return ctx_fail(ctx, BRT_ERR_HIP, hipGetErrorString(e)); }
    int32_t rc = BRT_OK;
    auto body = [&]() -> int32_t {
        HIP_TRY(ctx, hipMemcpyAsync(d_in, in16, (size_t)n * 64, hipMemcpyHostToDevice, dc.stream));
        HIP_TRY(ctx, launch_debug_eval(op, d_in, d_out, n, dc.stream));
        HIP_TRY(ctx, hipMemcpyAsync(out8, d_out, (size_t)n * 32, hipMemcpyDeviceToHost, dc.stream));
        HIP_TRY(ctx, hipStreamSynchronize(dc.stream));
        return BRT_OK;
    };
    rc = body();
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    return rc;
    });
}

}  // extern "C"
