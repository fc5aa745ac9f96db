// brt_host.h -- internal interface of the host-side (CPU) parts; see brt_host.cpp.
#pragma once
#include <cstdint>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/bevyray_amd.h"
#include "brt_layout.h"

namespace brt {

extern thread_local std::string g_last_error;
int32_t fail(int32_t code, const std::string& msg);

// Exception barrier of the C ABI (include/bevyray_amd.h: "none throws or aborts across the boundary"; the reference's node returns
// Ok(()) and skips the pass on every failure, pipeline.rs:82-85).  Every extern "C" body runs inside guard(): an exception -- in
// practice std::bad_alloc from a std::vector / std::string of the host side -- becomes an error code and a brt_last_error text
// instead of unwinding into the (Rust) caller, which would be undefined behaviour.  guard_fail itself cannot throw: the texts
// it stores fit std::string's in-place buffer or are truncated to it when even that assignment fails.
int32_t guard_fail(std::string* ctx_error, int32_t code, const char* what) noexcept;
template <class F>
inline int32_t guard(std::string* ctx_error, F&& body) noexcept {
    try {
        return body();
    } catch (const std::bad_alloc&) {
        return guard_fail(ctx_error, BRT_ERR_OUT_OF_MEMORY, "out of memory");
    } catch (const std::exception& e) {
        return guard_fail(ctx_error, BRT_ERR_INTERNAL, e.what());
    } catch (...) {
        return guard_fail(ctx_error, BRT_ERR_INTERNAL, "unknown error");
    }
}
// test hook: throws what `kind` names (1 std::bad_alloc, 2 std::logic_error, 3 a non-standard type); brt_api.cpp calls it from
// brt_upload_scene / brt_render* when the tuning knob BRT_TEST_THROW is set (and clears the knob), so that the GPU suite can
// force an exception through the barrier of exports that own a context
[[noreturn]] void throw_for_test(uint32_t kind);

// Scene in the device encoding (brt_layout.h), still in host vectors.
struct EncodedScene {
    std::vector<float> pairs;            // PAIR_WORDS per pair record (brt_layout.h)
    std::vector<float> spheres;          // 4 per model
    std::vector<uint32_t> sphere_material;
    std::vector<float> sphere_mats;      // 8 per model: the model's own material (materials[material_id]) -- one read per hit instead of two dependent ones
    std::vector<float> materials;        // 8 per material
    std::vector<uint32_t> leaf_table;    // 2 per general leaf
    uint32_t n_pairs = 0, n_models = 0, n_materials = 0;
    uint32_t root_desc = 0;
    uint32_t max_leaf_depth = 0;
    uint32_t stack_entries = 2;
    bool desc16 = false;                 // descriptors in the 16-bit form (brt_layout.h)
    bool simple_tree = false;            // single-sphere leaves only, depth below the stack-overflow rule
    bool boxes_ordered = false;          // every child box finite with min <= max
};

int32_t validate_and_encode(const Model* models, uint32_t n_models, const Material* materials, uint32_t n_materials,
                            const BVHNode* nodes, uint32_t n_nodes, EncodedScene* out, std::string* err);
int32_t build_bvh_ploc(const Model* models, uint32_t n_models, std::vector<BVHNode>* out);
int32_t build_bvh_sah(const Model* models, uint32_t n_models, float reach, std::vector<BVHNode>* out);   // binned SAH, same node contract; reach: brt_sah.h
// last step of the SAH rule (brt_sah.h "giant spheres first"), on the host for both builders: the CPU one calls it itself, the GPU
// build (brt_api.cpp build_bvh_on_device) on the nodes it has read back
void sah_giant_leaves_first(BVHNode* nodes, uint32_t n_nodes, const Model* models, uint32_t n_models);
// the reach a camera needs of the callee-built SAH tree (rule: brt_sah.h "leaf boxes"; used by brt_api.cpp ensure_tree_reach)
constexpr uint32_t kTreeLevelMax = 80;      // 2 S * 2^20: every pad has long been the reference's 0.1
struct TreeScene {
    float scale = 0.0f;                    // S of brt_sah.h (NaN: no ordinary sphere)
    float rmin = 3.4e38f, rmax = 0.0f;     // radii of the ordinary spheres (rmin > rmax: none)
    std::vector<float> big;                // {centre, radius} of the spheres of radius > 100 (the ground): tangent lengths from the camera
};
TreeScene tree_scene_of(const Model* models, uint32_t n_models);
float tree_scale_used(float scene_scale, float reach);
bool tree_pads_equal(const TreeScene& t, float reach_a, float reach_b);   // the trees of both reaches are the same bytes
uint32_t tree_level_for(float scene_scale, const std::vector<float>& big_spheres, const float cam_pos[3]);
float tree_reach_of(float scene_scale, uint32_t level);      // the builders' `reach` of a level (0 at level 0: the scene's own extent)
int32_t scene_generate(uint32_t kind, uint64_t seed, std::vector<Model>* models, std::vector<Material>* materials);
float tan_half_fov(float fov);

// Dispatch order of the 8x8 tiles from one frame's measurement (brt_host.cpp).
struct TileOrderParams {
    uint32_t sample_count;       // samples per pixel of the measured frame
    uint64_t grid_lanes;         // lanes of the launch grid (CUs x threads per workgroup)
    uint32_t sorted;             // 1: non-sky tiles by their longest pixel, longest first; 0: raster order
    uint32_t sky_slack_permille; // a tile is "sky" when it needed <= 64 * spp * (1 + slack) rays
    uint32_t lane_permille;      // share of the non-sky tiles (the front of the order) that goes to the lane queue
    uint32_t critical;           // 1: mark critical tiles
    // a tile is ranked by the longest pixel of its (2 dilate_x + 1) x (2 dilate_y + 1) neighbourhood in the tiles_x-wide tile grid and
    // is "sky" only if that whole neighbourhood was (0, 0: by itself); brt_order.hip has the why
    uint32_t tiles_x = 0, dilate_x = 0, dilate_y = 0;
    // the last min(split_tail, non-sky tiles) non-sky tiles are in the order TWICE, as two half-sample jobs (see build_tile_order)
    uint32_t split_tail = 0;
};
struct TileOrder {
    std::vector<uint32_t> order; // order[k] = k-th tile to hand out
    uint32_t n_lane = 0;         // order[0 .. n_lane) is the lane queue, the rest the tile queue
    uint32_t n_critical = 0;     // order[0 .. n_critical) are the critical tiles
    uint32_t longest_pixel = 0;
    uint32_t n_nonsky = 0;       // order[0 .. n_nonsky) are the tiles that are not sky (before the second halves are inserted)
    uint32_t n_split = 0;        // order[n_nonsky - n_split .. n_nonsky) = first halves, order[n_nonsky .. n_nonsky + n_split) = the same tiles, second halves
};
void build_tile_order(const uint32_t* ray_sum, const uint32_t* longest, uint32_t n_tiles, const TileOrderParams& p, TileOrder* out);

// Strips to parts by measured cost (brt_plan_strips; exported for tests as brt_host_plan_strips): inside every group of n_parts consecutive
// strips a permutation of the parts -- groups from the dearest down, in a group the dearest strip to the part with the least so far, ties by
// index: a function of the integers alone, so every rank of a job computes the same table.
void plan_strip_table(const uint64_t* strip_cost, uint32_t n_strips, uint32_t n_parts, uint32_t* out_part_of_strip);

}  // namespace brt
