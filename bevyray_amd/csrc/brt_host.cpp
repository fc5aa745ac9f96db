// brt_host.cpp -- host-side (CPU) parts of the render node that run before the kernels:
//   * scene validation + re-encoding into the device format   (brt_layout.h)
//   * native PLOC BVH builder                                  (replaces the obvhs call,
//                                                               reference extract.rs:315-332)
//   * seeded scene generators                                  (reference main.rs:49-240)
//   * mirror of the extract stage for non-Rust hosts           (reference extract.rs:63-209)
// None of this traces rays; there is no CPU rendering path in the product.
#include "brt_host.h"
#include "brt_ploc.h"
#include "brt_sah.h"
#include "brt_srgb_table.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace brt {

thread_local std::string g_last_error;

int32_t fail(int32_t code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

int32_t guard_fail(std::string* ctx_error, int32_t code, const char* what) noexcept {
    // (an assignment that fits the string's capacity -- at least the in-place buffer of 15 characters -- does not allocate)
    auto put = [&](std::string& dst) noexcept {
        try {
            dst.assign(what ? what : "?");
        } catch (...) {
            try { dst.assign(code == BRT_ERR_OUT_OF_MEMORY ? "out of memory" : "internal error"); } catch (...) {}
        }
    };
    put(g_last_error);
    if (ctx_error) put(*ctx_error);
    return code;
}

void throw_for_test(uint32_t kind) {
    if (kind == 1u) throw std::bad_alloc();
    if (kind == 2u) throw std::logic_error("BRT_TEST_THROW: logic_error");
    throw 42;
}

// ---------------------------------------------------------------------------------------
// Validation + encoding
// ---------------------------------------------------------------------------------------

int32_t validate_and_encode(const Model* models, uint32_t n_models, const Material* materials, uint32_t n_materials,
                            const BVHNode* nodes, uint32_t n_nodes, EncodedScene* out, std::string* err) {
    if (n_models == 0) { *err = "scene has no spheres (the reference skips the pass)"; return BRT_ERR_EMPTY_SCENE; }
    if (!models || !materials || n_materials == 0) { *err = "null models/materials"; return BRT_ERR_INVALID_ARGUMENT; }
    if (!nodes || n_nodes == 0) { *err = "null/empty BVH"; return BRT_ERR_INVALID_BVH; }
    if (n_models > DESC32_MAX_INDEX || n_nodes > DESC32_MAX_INDEX) { *err = "scene too large for 30-bit descriptors"; return BRT_ERR_UNSUPPORTED; }
    for (uint32_t i = 0; i < n_models; i++) {
        if (models[i].material_id >= n_materials) {
            *err = "model " + std::to_string(i) + ": material_id " + std::to_string(models[i].material_id) +
                   " >= n_materials " + std::to_string(n_materials);
            return BRT_ERR_INVALID_SCENE;
        }
    }

    // Breadth-first walk from node 0.  Interior nodes get pair-record ids in BFS order, so
    // the top of the tree has the lowest record ids.
    std::vector<uint32_t> pair_id(n_nodes, 0xffffffffu);
    std::vector<uint8_t> seen(n_nodes, 0);
    std::vector<uint32_t> depth(n_nodes, 0);
    std::vector<uint32_t> order;  // interior nodes in BFS order
    order.reserve(n_nodes / 2 + 1);
    uint32_t max_leaf_depth = 0;
    uint32_t n_general_leaves = 0;   // leaves of more than one sphere (they go through the leaf table)

    auto leaf_check = [&](uint32_t n) -> bool {
        uint64_t end = (uint64_t)nodes[n].index + (uint64_t)nodes[n].model_count;
        return end <= (uint64_t)n_models;
    };

    std::vector<uint32_t> frontier{0};
    seen[0] = 1;
    size_t head = 0;
    while (head < frontier.size()) {
        uint32_t n = frontier[head++];
        const BVHNode& nd = nodes[n];
        if (nd.model_count > 0) {
            if (!leaf_check(n)) {
                *err = "BVH leaf " + std::to_string(n) + ": models [" + std::to_string(nd.index) + ", +" +
                       std::to_string(nd.model_count) + ") exceed n_models " + std::to_string(n_models);
                return BRT_ERR_INVALID_BVH;
            }
            max_leaf_depth = std::max(max_leaf_depth, depth[n]);
            if (nd.model_count > 1) n_general_leaves++;
            continue;
        }
        uint64_t c0 = nd.index, c1 = (uint64_t)nd.index + 1;
        if (c1 >= n_nodes) {
            *err = "BVH node " + std::to_string(n) + ": child index " + std::to_string(c1) + " >= n_nodes " + std::to_string(n_nodes);
            return BRT_ERR_INVALID_BVH;
        }
        if (seen[c0] || seen[c1]) {
            *err = "BVH node " + std::to_string(n) + ": child " + std::to_string(seen[c0] ? c0 : c1) +
                   " is reachable twice (cycle or shared subtree)";
            return BRT_ERR_INVALID_BVH;
        }
        seen[c0] = seen[c1] = 1;
        depth[c0] = depth[c1] = depth[n] + 1;
        pair_id[n] = (uint32_t)order.size();
        order.push_back(n);
        frontier.push_back((uint32_t)c0);
        frontier.push_back((uint32_t)c1);
    }

    EncodedScene& e = *out;
    e = EncodedScene();
    e.n_pairs = (uint32_t)order.size();
    if ((uint64_t)e.n_pairs * PAIR_UNITS > 0x7FFFFFFFull) { *err = "scene too large for 31-bit record offsets"; return BRT_ERR_UNSUPPORTED; }
    e.pairs.assign(pair_array_bytes(e.n_pairs) / 4, 0.0f);

    // 16-bit descriptors when every index fits 14 bits: sphere ids, pair-record ids, leaf-table ids
    e.desc16 = (n_models <= DESC16_MAX_INDEX) && (e.n_pairs <= DESC16_MAX_INDEX) && (n_general_leaves <= DESC16_MAX_INDEX);
    const uint32_t LEAF = e.desc16 ? Desc<true>::LEAF : Desc<false>::LEAF;
    const uint32_t LEAF1 = e.desc16 ? Desc<true>::LEAF1 : Desc<false>::LEAF1;
    auto desc_of = [&](uint32_t n) -> uint32_t {
        const BVHNode& nd = nodes[n];
        if (nd.model_count == 0) return e.desc16 ? Desc<true>::interior(pair_id[n]) : Desc<false>::interior(pair_id[n]);
        if (nd.model_count == 1) return LEAF | LEAF1 | nd.index;
        uint32_t id = (uint32_t)(e.leaf_table.size() / 2);
        e.leaf_table.push_back(nd.index);
        e.leaf_table.push_back(nd.model_count);
        return LEAF | id;
    };

    // (for the 16-bit form LEAF is 0xFFFF8000: the register form, sign-extended -- brt_layout.h)
    e.root_desc = desc_of(0);
    e.boxes_ordered = true;
    for (uint32_t i = 0; i < e.n_pairs; i++) {
        const BVHNode& nd = nodes[order[i]];
        const BVHNode& L = nodes[nd.index];
        const BVHNode& R = nodes[nd.index + 1];
        float* rec = &e.pairs[(size_t)PAIR_WORDS * i];
        const uint32_t block[3] = {PAIR_X / 4, PAIR_Y / 4, PAIR_Z / 4};
        for (int k = 0; k < 3; k++) {
            float* g = rec + block[k];
            g[0] = L.bounds_min[k]; g[1] = R.bounds_min[k]; g[2] = L.bounds_max[k]; g[3] = R.bounds_max[k];   // G0
            g[4] = L.bounds_max[k]; g[5] = R.bounds_max[k]; g[6] = L.bounds_min[k]; g[7] = R.bounds_min[k];   // G1
            for (const BVHNode* c : {&L, &R})
                if (!(std::isfinite(c->bounds_min[k]) && std::isfinite(c->bounds_max[k]) && c->bounds_min[k] <= c->bounds_max[k]))
                    e.boxes_ordered = false;
        }
        const uint32_t dd[2] = {desc_of(nd.index), desc_of(nd.index + 1)};
        std::memcpy(rec + PAIR_DESC / 4, dd, sizeof dd);
    }

    e.n_models = n_models;
    e.spheres.resize(4 * (size_t)n_models);
    e.sphere_material.resize(n_models);
    for (uint32_t i = 0; i < n_models; i++) {
        e.spheres[4 * (size_t)i + 0] = models[i].position[0];
        e.spheres[4 * (size_t)i + 1] = models[i].position[1];
        e.spheres[4 * (size_t)i + 2] = models[i].position[2];
        e.spheres[4 * (size_t)i + 3] = models[i].radius * models[i].radius;  // raytrace.wgsl:375
        e.sphere_material[i] = models[i].material_id;
    }
    e.n_materials = n_materials;
    e.materials.resize(8 * (size_t)n_materials);
    std::memcpy(e.materials.data(), materials, 32 * (size_t)n_materials);
    // the material of every sphere beside it (material ids were validated above): the reference's extract stage writes one material
    // entry per sphere anyway (extract.rs:301-310), and a hit then needs ONE read that does not wait for the id
    e.sphere_mats.resize(8 * (size_t)n_models);
    for (uint32_t i = 0; i < n_models; i++) std::memcpy(e.sphere_mats.data() + 8 * (size_t)i, &materials[models[i].material_id], 32);
    e.max_leaf_depth = max_leaf_depth;
    // After popping a node of depth k the stack holds at most k entries and receives at
    // most two pushes, so the deepest write index is max_leaf_depth (entries needed: +1).
    e.stack_entries = std::min<uint32_t>(32u, max_leaf_depth + 1u);
    if (e.stack_entries < 2) e.stack_entries = 2;
    e.simple_tree = e.leaf_table.empty() && (max_leaf_depth + 1u < 31u);
    return BRT_OK;
}

// ---------------------------------------------------------------------------------------
// PLOC builder, CPU version (the GPU version in brt_bvh.hip produces the same bytes; both use
// the arithmetic and the numbering rule of brt_ploc.h).
// ---------------------------------------------------------------------------------------

int32_t build_bvh_ploc(const Model* models, uint32_t n_models, std::vector<BVHNode>* out) {
    out->clear();
    if (n_models == 0) return BRT_OK;
    const uint32_t n = n_models;
    std::vector<PlocBox> box(2 * (size_t)n - 1);
    std::vector<int32_t> left(2 * (size_t)n - 1, -1), right(2 * (size_t)n - 1, -1);
    PlocBox scene = ploc_model_box(models[0].position, models[0].radius);
    for (uint32_t i = 0; i < n; i++) {
        box[i] = ploc_model_box(models[i].position, models[i].radius);
        scene = ploc_merge(scene, box[i]);
    }
    std::vector<std::pair<uint64_t, uint32_t>> keys(n);
    for (uint32_t i = 0; i < n; i++) keys[i] = {ploc_morton(box[i], scene), i};
    std::sort(keys.begin(), keys.end());   // (code, index): a strict total order

    std::vector<int32_t> cur(n), next, nn;
    for (uint32_t i = 0; i < n; i++) cur[i] = (int32_t)keys[i].second;
    uint32_t created = n;
    while (cur.size() > 1) {
        const int32_t m = (int32_t)cur.size();
        nn.assign(m, -1);
        for (int32_t i = 0; i < m; i++) {
            float best = 0.0f;
            int32_t bj = -1;
            const int32_t lo = std::max(0, i - PLOC_SEARCH), hi = std::min(m - 1, i + PLOC_SEARCH);
            for (int32_t j = lo; j <= hi; j++) {
                if (j == i) continue;
                const float a = ploc_pair_cost(box[cur[std::min(i, j)]], box[cur[std::max(i, j)]]);
                if (ploc_better(a, i, j, best, bj)) { best = a; bj = j; }
            }
            nn[i] = bj;
        }
        next.clear();
        for (int32_t i = 0; i < m; i++) {
            const int32_t j = nn[i];
            if (nn[j] == i) {
                if (i < j) {
                    box[created] = ploc_merge(box[cur[i]], box[cur[j]]);
                    left[created] = cur[i];
                    right[created] = cur[j];
                    next.push_back((int32_t)created);
                    created++;
                }
            } else {
                next.push_back(cur[i]);
            }
        }
        if (next.size() == cur.size()) return BRT_ERR_INVALID_SCENE;   // cannot happen (brt_ploc.h: the cheapest pair is mutual)
        cur.swap(next);
    }

    // numbering rule of brt_ploc.h: root = the last cluster; children of rank r at 1+2r, 2+2r
    out->resize(2 * (size_t)n - 1);
    auto write = [&](uint32_t slot, int32_t t) {
        BVHNode& o = (*out)[slot];
        std::memset(&o, 0, sizeof o);
        for (int k = 0; k < 3; k++) { o.bounds_min[k] = box[t].mn[k]; o.bounds_max[k] = box[t].mx[k]; }
        if (left[t] < 0) { o.index = (uint32_t)t; o.model_count = 1; }          // leaf: model id (extract.rs:318,329)
        else { o.index = 1u + 2u * ((2u * n - 2u) - (uint32_t)t); o.model_count = 0; }
    };
    write(0, (int32_t)(2 * n - 2));
    for (uint32_t t = n; t < 2 * n - 1; t++) {
        const uint32_t r = (2 * n - 2) - t;
        write(1 + 2 * r, left[t]);
        write(2 + 2 * r, right[t]);
    }
    return BRT_OK;
}

// ---------------------------------------------------------------------------------------
// Binned-SAH builder (top down), for trees the CALLEE builds (brt_upload_scene without a BVH).
// The reference builds PLOC (obvhs, extract.rs:315-321) because it rebuilds on the CPU every frame; the shader only
// asks for the node contract (root at 0, children adjacent, single-sphere leaves that address the model buffer,
// raytrace.wgsl:325-341), and pixels do not depend on the topology except through exact ties and the stack-overflow
// rule (both reproduced by the oracle on whatever tree it is given).  The ray loop visits fewer nodes in an SAH tree:
// 10 004-sphere grid 23.6 -> 19.1 interior visits per ray (tests/tools/exp_traversal.c), cover scene 13.1 -> 12.0.
// Deterministic (plain loops, f64 costs, index tie-breaks), NaN / inf spheres included: a subset whose centroids do not
// split falls back to halves in index order.  Depth is capped below the simple-tree limit (31 levels) by switching to
// halves when a subtree's remaining depth budget only just covers a balanced split.
// ---------------------------------------------------------------------------------------

int32_t build_bvh_sah(const Model* models, uint32_t n_models, float reach, std::vector<BVHNode>* out) {
    out->clear();
    if (n_models == 0) return BRT_OK;
    const uint32_t n = n_models;
    std::vector<SahKeyBox> box(n);
    std::vector<double> cen(3 * (size_t)n);
    uint32_t scale_key = sah_reach_key(reach);    // the scene's scale: max over its ordinary spheres (an order-independent integer max), not below reach / 2
    for (uint32_t i = 0; i < n; i++) {
        const uint32_t k = sah_key_max(sah_scale_term(models[i].position, models[i].radius));
        scale_key = k > scale_key ? k : scale_key;
    }
    const float scale = sah_unkey_max(scale_key);
    for (uint32_t i = 0; i < n; i++) {
        const PlocBox b = sah_model_box(models[i].position, models[i].radius, scale);
        box[i] = sah_keybox(b);
        for (int k = 0; k < 3; k++) cen[3 * (size_t)i + k] = sah_centroid(b, k);
    }
    std::vector<uint32_t> idx(n), tmp(n);
    for (uint32_t i = 0; i < n; i++) idx[i] = i;
    out->resize(2 * (size_t)n - 1);
    auto write_node = [&](uint32_t slot, const SahKeyBox& kb, uint32_t index, uint32_t count) {
        BVHNode& o = (*out)[slot];
        std::memset(&o, 0, sizeof o);
        const PlocBox b = sah_unkeybox(kb);
        for (int k = 0; k < 3; k++) { o.bounds_min[k] = b.mn[k]; o.bounds_max[k] = b.mx[k]; }
        o.index = index;
        o.model_count = count;
    };
    // (the rule, the cost and the numbering: brt_sah.h; brt_sah.hip runs the same nodes in another order and writes the same bytes)
    struct Job { uint32_t slot, begin, end, depth, rank; };
    std::vector<Job> todo;
    todo.push_back({0u, 0u, n, 0u, 0u});
    while (!todo.empty()) {
        const Job j = todo.back();
        todo.pop_back();
        const uint32_t count = j.end - j.begin;
        SahKeyBox nb = sah_keybox_empty();
        for (uint32_t i = j.begin; i < j.end; i++) sah_keybox_merge(nb, box[idx[i]]);
        if (count == 1) {
            write_node(j.slot, nb, idx[j.begin], 1u);          // leaf: the model id itself (extract.rs:318,329)
            continue;
        }
        uint32_t mid = j.begin + count / 2;                    // fallback: halves in the current order
        const bool balanced_only = j.depth + sah_ceil_log2(count) >= kSahMaxDepth;
        if (!balanced_only && count > 2) {
            double cmin[3], cmax[3];
            for (int k = 0; k < 3; k++) { cmin[k] = kSahDblMax; cmax[k] = -kSahDblMax; }
            for (uint32_t i = j.begin; i < j.end; i++)
                for (int k = 0; k < 3; k++) {
                    const double c = cen[3 * (size_t)idx[i] + k];
                    cmin[k] = c < cmin[k] ? c : cmin[k];
                    cmax[k] = c > cmax[k] ? c : cmax[k];
                }
            double best_cost = kSahDblMax;
            int best_axis = -1, best_bin = -1;
            for (int k = 0; k < 3; k++) {
                if (!sah_axis_usable(cmin[k], cmax[k])) continue;
                SahKeyBox bb[kSahBins];
                uint32_t bc[kSahBins] = {};
                for (int b = 0; b < kSahBins; b++) bb[b] = sah_keybox_empty();
                const double scale = (double)kSahBins / (cmax[k] - cmin[k]);
                for (uint32_t i = j.begin; i < j.end; i++) {
                    const int b = sah_bin(cen[3 * (size_t)idx[i] + k], cmin[k], scale);
                    sah_keybox_merge(bb[b], box[idx[i]]);
                    bc[b]++;
                }
                // cost of splitting after bin s = area(left) * w(n_left) + area(right) * w(n_right), w = sah_side_weight
                for (int s = 0; s + 1 < kSahBins; s++) {
                    SahKeyBox L = sah_keybox_empty(), R = sah_keybox_empty();
                    uint32_t nl = 0, nr = 0;
                    for (int b = 0; b < kSahBins; b++) {
                        if (b <= s) { sah_keybox_merge(L, bb[b]); nl += bc[b]; }
                        else { sah_keybox_merge(R, bb[b]); nr += bc[b]; }
                    }
                    if (nl == 0 || nr == 0) continue;
                    const double cost = sah_half_area(L) * sah_side_weight(nl, count) + sah_half_area(R) * sah_side_weight(nr, count);
                    if (cost < best_cost) { best_cost = cost; best_axis = k; best_bin = s; }   // strict <: first axis / bin wins ties
                }
            }
            if (best_axis >= 0) {
                const double scale = (double)kSahBins / (cmax[best_axis] - cmin[best_axis]);
                uint32_t nl = 0, nr = 0;                       // stable partition by "bin <= best_bin"
                for (uint32_t i = j.begin; i < j.end; i++) {
                    const uint32_t m = idx[i];
                    if (sah_bin(cen[3 * (size_t)m + best_axis], cmin[best_axis], scale) <= best_bin) idx[j.begin + nl++] = m;
                    else tmp[nr++] = m;
                }
                for (uint32_t i = 0; i < nr; i++) idx[j.begin + nl + i] = tmp[i];
                const uint32_t m = j.begin + nl;
                // a lopsided split must leave both sides inside the depth budget; else halves (of the new order)
                if (m > j.begin && m < j.end) {
                    const uint32_t big = std::max(m - j.begin, j.end - m);
                    if (j.depth + 1 + sah_ceil_log2(big) <= kSahMaxDepth) mid = m;
                }
            }
        }
        const uint32_t child = 1u + 2u * j.rank;
        write_node(j.slot, nb, child, 0u);
        // the reference pops child `index + 1` first (raytrace.wgsl:329-341): no preference is encoded here
        todo.push_back({child + 1, mid, j.end, j.depth + 1, j.rank + (mid - j.begin)});
        todo.push_back({child, j.begin, mid, j.depth + 1, j.rank + 1u});
    }
    sah_giant_leaves_first(out->data(), (uint32_t)out->size(), models, n_models);
    return BRT_OK;
}

#ifndef BRT_GIANT_RULE
#define BRT_GIANT_RULE 1   // 1: by the boxes (below); 0: round 5's rule, a sphere of radius > 100 (A/B: scripts/exp_tree_rules.py)
#endif
void sah_giant_leaves_first(BVHNode* nodes, uint32_t n_nodes, const Model* models, uint32_t n_models) {
    // A LEAF whose box takes at least half of its parent's surface, beside a sibling that is a subtree: nearly every ray that enters
    // the parent enters that leaf -- the ground under a scene, a big sphere beside a cluster of small ones.  Decided on the boxes the
    // builder made (half areas in f64, as in the split cost), not on a radius: a ground of radius 50 or 80 is found like one of
    // radius 1000, a scene without such a sphere is left alone.
    auto half_area = [](const BVHNode& nd) {
        const double dx = (double)nd.bounds_max[0] - (double)nd.bounds_min[0], dy = (double)nd.bounds_max[1] - (double)nd.bounds_min[1],
                     dz = (double)nd.bounds_max[2] - (double)nd.bounds_min[2];
        return (dx * dy + dy * dz) + dz * dx;
    };
    auto giant_leaf = [&](const BVHNode& parent, const BVHNode& nd, const BVHNode& sibling) {
        if (nd.model_count != 1u || nd.index >= n_models) return false;
#if BRT_GIANT_RULE == 0
        (void)parent; (void)sibling;
        return models[nd.index].radius > 100.0f && std::isfinite(models[nd.index].radius);
#else
        const double a = half_area(nd), ap = half_area(parent);
        return sibling.model_count == 0u && std::isfinite(a) && std::isfinite(ap) && ap > 0.0 && a >= 0.5 * ap;
#endif
    };
    for (uint32_t i = 0; i < n_nodes; i++) {
        if (nodes[i].model_count != 0u) continue;
        const uint32_t a = nodes[i].index;
        if (a + 1u >= n_nodes || a + 1u == 0u) continue;
        if (giant_leaf(nodes[i], nodes[a], nodes[a + 1u]) && !giant_leaf(nodes[i], nodes[a + 1u], nodes[a])) std::swap(nodes[a], nodes[a + 1u]);
    }
}

// ---------------------------------------------------------------------------------------
// Seeded scenes
// ---------------------------------------------------------------------------------------

namespace {

struct SplitMix64 {
    uint64_t s;
    explicit SplitMix64(uint64_t seed) : s(seed) {}
    uint64_t next() {
        uint64_t z = (s += 0x9e3779b97f4a7c15ull);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    }
    float uniform() { return (float)(next() >> 40) * (1.0f / 16777216.0f); }  // [0,1), like rand::random::<f32>()
    float uniform(float lo, float hi) { return lo + (hi - lo) * uniform(); }
};

// bevy_color Srgba -> LinearRgba per channel (Color::to_linear, extract.rs:201)
inline float srgb_to_linear(float c) {
    if (c <= 0.0f) return c;
    if (c <= 0.04045f) return c / 12.92f;
    return std::pow((c + 0.055f) / 1.055f, 2.4f);
}

// bevy 0.14 StandardMaterial::default(): base_color WHITE, metallic 0, perceptual_roughness
// 0.5, reflectance 0.5, ior 1.5, specular_transmission 0.
struct StdMat {
    float base[3] = {1.0f, 1.0f, 1.0f};
    bool srgb = true;
    float metallic = 0.0f, perceptual_roughness = 0.5f, reflectance = 0.5f, ior = 1.5f, specular_transmission = 0.0f;
};

inline Material prepare_asset(const StdMat& m) {  // RaytraceMaterial::prepare_asset, extract.rs:196-208
    Material r;
    for (int k = 0; k < 3; k++) r.base_color[k] = m.srgb ? srgb_to_linear(m.base[k]) : m.base[k];
    r.metallic = m.metallic;
    r.roughness = m.perceptual_roughness;
    r.reflectance = m.reflectance;
    r.ior = m.ior;
    r.specular_transmission = m.specular_transmission;
    return r;
}

struct SceneOut {
    std::vector<Model> models;
    std::vector<Material> materials;
    void add(float x, float y, float z, float radius, const StdMat& m) {
        Model md;
        std::memset(&md, 0, sizeof md);
        md.position[0] = x; md.position[1] = y; md.position[2] = z;
        md.radius = radius;
        md.material_id = (uint32_t)models.size();  // one material entry per sphere, extract.rs:301-310
        models.push_back(md);
        materials.push_back(prepare_asset(m));
    }
};

// main.rs:105-182: small spheres over a grid, material lottery on choose_mat
void small_spheres(SceneOut& s, SplitMix64& rng, int a0, int a1, int b0, int b1, bool srgb, bool rtiow, bool skip_rule) {
    for (int a = a0; a < a1; a++) {
        for (int b = b0; b < b1; b++) {
            const float choose_mat = rng.uniform();
            const float cx = (float)a + 0.9f * rng.uniform();
            const float cy = 0.2f;
            const float cz = (float)b + 0.9f * rng.uniform();
            const float dx = cx - 4.0f, dy = cy - 0.2f, dz = cz - 0.0f;
            if (skip_rule && !(std::sqrt(dx * dx + dy * dy + dz * dz) > 0.9f)) continue;  // main.rs:115
            StdMat m;
            m.srgb = srgb;
            if (choose_mat < 0.8f) {            // diffuse, main.rs:116-124
                float r0 = rng.uniform(), g0 = rng.uniform(), b0_ = rng.uniform();
                float r1 = rng.uniform(), g1 = rng.uniform(), b1_ = rng.uniform();
                m.base[0] = r0 * r1; m.base[1] = g0 * g1; m.base[2] = b0_ * b1_;
                m.metallic = 0.0f;
                if (rtiow) m.perceptual_roughness = 0.0f;
            } else if (choose_mat < 0.95f) {    // metal, main.rs:137-146
                if (rtiow) {
                    m.base[0] = rng.uniform(0.5f, 1.0f); m.base[1] = rng.uniform(0.5f, 1.0f); m.base[2] = rng.uniform(0.5f, 1.0f);
                    m.perceptual_roughness = rng.uniform(0.0f, 0.5f);
                } else {
                    m.base[0] = rng.uniform(); m.base[1] = rng.uniform(); m.base[2] = rng.uniform();
                    m.perceptual_roughness = rng.uniform();
                }
                m.metallic = 1.0f;
            } else {                            // glass, main.rs:159-166
                m.metallic = 0.0f;
                m.ior = 1.5f;
                m.specular_transmission = 1.0f;
                if (rtiow) m.perceptual_roughness = 0.0f;
            }
            s.add(cx, cy, cz, 0.2f, m);
        }
    }
}

void big_spheres(SceneOut& s, bool srgb, bool rtiow) {  // main.rs:184-239
    StdMat glass; glass.srgb = srgb; glass.metallic = 0.0f; glass.ior = 1.5f; glass.specular_transmission = 1.0f;
    if (rtiow) glass.perceptual_roughness = 0.0f;
    s.add(0.0f, 1.0f, 0.0f, 1.0f, glass);
    StdMat diffuse; diffuse.srgb = srgb; diffuse.base[0] = 0.4f; diffuse.base[1] = 0.2f; diffuse.base[2] = 0.1f; diffuse.metallic = 0.0f;
    if (rtiow) diffuse.perceptual_roughness = 0.0f;
    s.add(-4.0f, 1.0f, 0.0f, 1.0f, diffuse);
    StdMat metal; metal.srgb = srgb; metal.base[0] = 0.7f; metal.base[1] = 0.6f; metal.base[2] = 0.5f; metal.metallic = 1.0f;
    metal.perceptual_roughness = 0.0f;
    s.add(4.0f, 1.0f, 0.0f, 1.0f, metal);
}

}  // namespace

int32_t scene_generate(uint32_t kind, uint64_t seed, std::vector<Model>* models, std::vector<Material>* materials) {
    SceneOut s;
    SplitMix64 rng(seed);
    StdMat ground;
    ground.base[0] = ground.base[1] = ground.base[2] = 0.5f;
    ground.metallic = 0.0f;
    switch (kind) {
        case BRT_SCENE_COVER:   // main.rs:87-239; grid a in -11..=11 (23), b in -11..11 (22)
            ground.srgb = true;
            s.add(0.0f, -1000.0f, 0.0f, 1000.0f, ground);
            small_spheres(s, rng, -11, 12, -11, 11, true, false, true);
            big_spheres(s, true, false);
            break;
        case BRT_SCENE_RTIOW_FINAL:  // the book's final scene: 22 x 22, linear albedos, no extra roughness on diffuse
            ground.srgb = false;
            ground.perceptual_roughness = 0.0f;
            s.add(0.0f, -1000.0f, 0.0f, 1000.0f, ground);
            small_spheres(s, rng, -11, 11, -11, 11, false, true, true);
            big_spheres(s, false, true);
            break;
        case BRT_SCENE_STRESS_GRID:  // 100 x 100 small spheres, cover-scene lottery
            ground.srgb = true;
            s.add(0.0f, -1000.0f, 0.0f, 1000.0f, ground);
            small_spheres(s, rng, -50, 50, -50, 50, true, false, false);
            big_spheres(s, true, false);
            break;
        default:
            return fail(BRT_ERR_INVALID_ARGUMENT, "unknown scene kind");
    }
    models->swap(s.models);
    materials->swap(s.materials);
    return BRT_OK;
}

// ---------------------------------------------------------------------------------------
// Extract-stage mirror
// ---------------------------------------------------------------------------------------

float tan_half_fov(float fov) { return (float)std::tan((double)(fov * 0.5f)); }

// ---------------------------------------------------------------------------------------
// Dispatch order of the 8x8 tiles (used by brt_api.cpp for k_trace_persistent's queues).
//
// A pixel is one sequential chain of samples (the reference threads one RNG state through them), so a frame
// ends when its slowest pixels end: a tile with a long pixel must not be handed out late.  From the rays each
// tile needed in a measured frame (sum and longest pixel):
//   first   the non-sky tiles by their longest pixel, longest first (`sorted`; raster order otherwise);
//   last    the "sky" tiles (one ray per sample: every path leaves the scene at once), in raster order.
// The kernel hands out WHOLE tiles to waves that have nothing left (the tile queue); `lane_permille` of the
// non-sky tiles, from the front, can instead be handed out pixel by pixel to single free lanes (the lane queue).
// CRITICAL tiles: a pixel whose chain alone takes half of what a lane works through in the whole frame
// (sum of rays / lanes of the grid) bounds the frame time by itself -- RTIOW at 256 spp and 50 bounces has
// pixels of > 10 000 sequential rays in a frame of ~5 000 rays per lane.  Tiles holding such a pixel (and at
// least half the frame's longest pixel) are flagged; in the sorted order they are its front.
// HALF-SAMPLE JOBS (split_tail): a launch ends in a tail -- when the queue runs dry most waves are in sky tiles (0.2 ms) and end
// together, the others are in their last non-sky tile (a pixel chain of >= ~130 rays: 1-2 ms) and end that much later: 6 % of the
// wave time on the headline frame.  The sky tiles at the end can only absorb a stagger of twice their own volume.
// So the LAST split_tail non-sky tiles are handed out twice: once for the first half of the samples (the lane leaves its pixel's
// state -- rng, sums, ray count -- in HBM), once, behind all the first halves, for the second half.  The jobs ahead of the sky tiles
// are then half as long, their stagger is what the sky tiles can absorb.  order = [non-sky, longest first ... | first halves |
// second halves | sky]; the order has n_tiles + n_split entries.
// ---------------------------------------------------------------------------------------
void build_tile_order(const uint32_t* ray_sum, const uint32_t* longest, uint32_t n_tiles, const TileOrderParams& p, TileOrder* out) {
    TileOrder& o = *out;
    o = TileOrder();
    o.order.resize(n_tiles);
    const uint64_t sky_cost = (uint64_t)64 * p.sample_count * (1000 + p.sky_slack_permille) / 1000;
    // what a tile is ranked by: itself, or its neighbourhood (brt_order.hip tile_cost_at)
    std::vector<uint32_t> rank(n_tiles);
    std::vector<uint8_t> is_sky(n_tiles);
    const bool dilated = (p.dilate_x | p.dilate_y) != 0u && p.tiles_x != 0u;
    const uint32_t tiles_y = dilated ? n_tiles / p.tiles_x : 0u;
    for (uint32_t i = 0; i < n_tiles; i++) {
        uint32_t lp = longest[i];
        bool sky = ray_sum[i] <= sky_cost;
        if (dilated) {
            const uint32_t ty = i / p.tiles_x, tx = i - ty * p.tiles_x;
            const uint32_t x0 = tx > p.dilate_x ? tx - p.dilate_x : 0u, x1 = std::min(tx + p.dilate_x, p.tiles_x - 1u);
            const uint32_t y0 = ty > p.dilate_y ? ty - p.dilate_y : 0u, y1 = std::min(ty + p.dilate_y, tiles_y - 1u);
            for (uint32_t y = y0; y <= y1 && y < tiles_y; y++)
                for (uint32_t x = x0; x <= x1; x++) {
                    const uint32_t j = y * p.tiles_x + x;
                    lp = std::max(lp, longest[j]);
                    sky = sky && ray_sum[j] <= sky_cost;
                }
        }
        rank[i] = lp;
        is_sky[i] = sky ? 1 : 0;
    }
    std::vector<uint64_t> keys;                                                // non-sky tiles: (~rank, index)
    uint64_t sum = 0;
    for (uint32_t i = 0; i < n_tiles; i++) {
        sum += ray_sum[i];
        o.longest_pixel = longest[i] > o.longest_pixel ? longest[i] : o.longest_pixel;
        if (!is_sky[i]) keys.push_back(((uint64_t)(~rank[i]) << 32) | i);
    }
    if (p.sorted) std::sort(keys.begin(), keys.end());                         // longest first, then by index
    uint32_t k = 0;
    o.n_nonsky = (uint32_t)keys.size();
    o.order.resize(n_tiles);
    for (uint64_t key : keys) o.order[k++] = (uint32_t)(key & 0xffffffffu);
    o.n_lane = (uint32_t)((uint64_t)keys.size() * p.lane_permille / 1000u);
    if (o.n_lane > keys.size()) o.n_lane = (uint32_t)keys.size();
    if (p.sorted && p.critical && p.grid_lanes != 0) {
        const uint64_t per_lane = sum / p.grid_lanes;
        const uint64_t thr = per_lane / 2 > o.longest_pixel / 2 ? per_lane / 2 : o.longest_pixel / 2;
        if (o.longest_pixel >= per_lane / 2)
            while (o.n_critical < keys.size() && rank[o.order[o.n_critical]] >= thr) o.n_critical++;
    }
    // the second halves of the last n_split non-sky tiles (never of a critical tile: its wave's priority goes with the queue slot), then the sky tiles in raster order
    o.n_split = p.sorted ? std::min(p.split_tail, o.n_nonsky - o.n_critical) : 0u;
    o.order.resize((size_t)n_tiles + o.n_split);
    for (uint32_t i = 0; i < o.n_split; i++) o.order[k++] = (uint32_t)(keys[keys.size() - o.n_split + i] & 0xffffffffu);
    for (uint32_t tile = 0; tile < n_tiles; tile++)
        if (is_sky[tile]) o.order[k++] = tile;
}

// ---- the reach a camera needs of the callee-built SAH tree (brt_sah.h "leaf boxes"; brt_api.cpp ensure_tree_reach) ---------------
TreeScene tree_scene_of(const Model* models, uint32_t n_models) {
    TreeScene t;
    uint32_t scale_key = kSahKeyMaxIdentity;
    for (uint32_t i = 0; i < n_models; i++) {
        const Model& m = models[i];
        const float term = sah_scale_term(m.position, m.radius);
        const uint32_t k = sah_key_max(term);
        scale_key = k > scale_key ? k : scale_key;
        if (term == term) {                                   // an ordinary sphere (the ones sah_model_pad sizes by the scale)
            t.rmin = m.radius < t.rmin ? m.radius : t.rmin;
            t.rmax = m.radius > t.rmax ? m.radius : t.rmax;
        }
        if (m.radius > 100.0f && std::isfinite(m.radius) && std::isfinite(m.position[0]) && std::isfinite(m.position[1]) &&
            std::isfinite(m.position[2]))
            t.big.insert(t.big.end(), {m.position[0], m.position[1], m.position[2], m.radius});
    }
    t.scale = sah_unkey_max(scale_key);
    return t;
}
// the scale the builders use for a `reach` (brt_sah.h sah_reach_key: max(S, reach / 2))
float tree_scale_used(float scene_scale, float reach) {
    const float half = 0.5f * reach;
    return (half > scene_scale) ? half : scene_scale;
}
// every leaf pad is the same under both reaches (all at the floor under the larger, or all at the ceiling under the smaller): the
// trees are the same bytes, a rebuild would change nothing
bool tree_pads_equal(const TreeScene& t, float reach_a, float reach_b) {
    const float sa = tree_scale_used(t.scale, reach_a), sb = tree_scale_used(t.scale, reach_b);
    if (sa == sb || !(t.rmin <= t.rmax)) return true;
    const float lo = sa < sb ? sa : sb, hi = sa < sb ? sb : sa;
    return sah_model_pad(t.rmin, hi) <= 0.01f || sah_model_pad(t.rmax, lo) >= 0.1f;   // (the pad falls with the radius)
}
float tree_reach_of(float scene_scale, uint32_t level) {
    if (level == 0u) return 0.0f;
    const double r = 2.0 * (double)scene_scale * std::exp2(0.25 * (double)level);
    return r < 3.0e38 ? (float)r : 3.0e38f;      // (finite: an infinite reach would read as "no floor", brt_sah.h sah_reach_key)
}
uint32_t tree_level_for(float scene_scale, const std::vector<float>& big_spheres, const float cam_pos[3]) {
    const double S = scene_scale;
    if (!(S > 0.0) || !std::isfinite(S)) return 0u;                 // no ordinary sphere: every pad is 0.1 as it is
    const double l1 = (std::fabs((double)cam_pos[0]) + std::fabs((double)cam_pos[1])) + std::fabs((double)cam_pos[2]);
    double L = 0.0;
    for (size_t i = 0; i + 3 < big_spheres.size(); i += 4) {
        const double dx = (double)cam_pos[0] - big_spheres[i], dy = (double)cam_pos[1] - big_spheres[i + 1], dz = (double)cam_pos[2] - big_spheres[i + 2];
        const double r = big_spheres[i + 3], h = std::sqrt(dx * dx + dy * dy + dz * dz) - r;
        const double t = h > 0.0 ? std::sqrt(h * (2.0 * r + h)) : 2.0 * r;      // (a camera inside the sphere: a chord)
        if (t > L) L = t;
    }
    const double need = l1 + S + L;
    if (!std::isfinite(need)) return kTreeLevelMax;
    if (need <= 2.0 * S) return 0u;
    const double k = std::ceil(4.0 * std::log2(need / (2.0 * S)));
    return k < 1.0 ? 1u : (k > (double)kTreeLevelMax ? kTreeLevelMax : (uint32_t)k);
}

void plan_strip_table(const uint64_t* strip_cost, uint32_t n_strips, uint32_t n_parts, uint32_t* out) {
    std::vector<uint64_t> load(n_parts, 0), gcost((n_strips + n_parts - 1u) / n_parts, 0);
    std::vector<uint32_t> groups(gcost.size());
    for (uint32_t s = 0; s < n_strips; s++) gcost[s / n_parts] += strip_cost[s];
    for (uint32_t g = 0; g < groups.size(); g++) groups[g] = g;
    std::stable_sort(groups.begin(), groups.end(), [&](uint32_t a, uint32_t b) { return gcost[a] > gcost[b]; });
    for (uint32_t g : groups) {
        std::vector<uint32_t> ss, ps(n_parts);
        for (uint32_t s = g * n_parts; s < n_strips && s < (g + 1u) * n_parts; s++) ss.push_back(s);
        for (uint32_t p = 0; p < n_parts; p++) ps[p] = p;
        std::stable_sort(ss.begin(), ss.end(), [&](uint32_t a, uint32_t b) { return strip_cost[a] > strip_cost[b]; });
        std::stable_sort(ps.begin(), ps.end(), [&](uint32_t a, uint32_t b) { return load[a] < load[b]; });
        for (size_t i = 0; i < ss.size(); i++) { out[ss[i]] = ps[i]; load[ps[i]] += strip_cost[ss[i]]; }
    }
}

}  // namespace brt

using namespace brt;

extern "C" {

int32_t brt_host_plan_strips(const uint64_t* strip_cost, uint32_t n_strips, uint32_t n_parts, uint32_t* out_part_of_strip) {
    return guard(nullptr, [&]() -> int32_t {
    if (!strip_cost || !out_part_of_strip || n_strips == 0u || n_parts == 0u || n_parts > 64u) return fail(BRT_ERR_INVALID_ARGUMENT, "null pointer / n_parts must be 1..64");
    plan_strip_table(strip_cost, n_strips, n_parts, out_part_of_strip);
    return BRT_OK;
    });
}

int32_t brt_build_bvh(const void* models, uint32_t n_models, void* out_nodes, uint32_t capacity, uint32_t* out_n_nodes) {
    return guard(nullptr, [&]() -> int32_t {
    if (!out_n_nodes) return fail(BRT_ERR_INVALID_ARGUMENT, "out_n_nodes is null");
    *out_n_nodes = 0;
    if (n_models == 0) return BRT_OK;
    if (!models) return fail(BRT_ERR_INVALID_ARGUMENT, "models is null");
    std::vector<BVHNode> nodes;
    int32_t rc = build_bvh_ploc((const Model*)models, n_models, &nodes);
    if (rc != BRT_OK) return rc;
    *out_n_nodes = (uint32_t)nodes.size();
    if (nodes.size() > capacity || !out_nodes)
        return fail(BRT_ERR_CAPACITY, "BVH needs " + std::to_string(nodes.size()) + " nodes, capacity " + std::to_string(capacity));
    std::memcpy(out_nodes, nodes.data(), nodes.size() * sizeof(BVHNode));
    return BRT_OK;
    });
}

int32_t brt_build_bvh_sah(const void* models, uint32_t n_models, float reach, void* out_nodes, uint32_t capacity, uint32_t* out_n_nodes) {
    return guard(nullptr, [&]() -> int32_t {
    if (!out_n_nodes) return fail(BRT_ERR_INVALID_ARGUMENT, "out_n_nodes is null");
    *out_n_nodes = 0;
    if (n_models == 0) return BRT_OK;
    if (!models) return fail(BRT_ERR_INVALID_ARGUMENT, "models is null");
    std::vector<BVHNode> nodes;
    int32_t rc = build_bvh_sah((const Model*)models, n_models, reach, &nodes);
    if (rc != BRT_OK) return rc;
    *out_n_nodes = (uint32_t)nodes.size();
    if (nodes.size() > capacity || !out_nodes)
        return fail(BRT_ERR_CAPACITY, "BVH needs " + std::to_string(nodes.size()) + " nodes, capacity " + std::to_string(capacity));
    std::memcpy(out_nodes, nodes.data(), nodes.size() * sizeof(BVHNode));
    return BRT_OK;
    });
}

int32_t brt_host_srgb_thresholds(float* out255) {
    return guard(nullptr, [&]() -> int32_t {
    if (!out255) return fail(BRT_ERR_INVALID_ARGUMENT, "null pointer");
    std::memcpy(out255, kSrgbThreshold, sizeof kSrgbThreshold);
    return BRT_OK;
    });
}

int32_t brt_host_tree_reach(const void* models, uint32_t n_models, const void* camera80, float* out_scene_scale, uint32_t* out_level,
                            float* out_reach) {
    return guard(nullptr, [&]() -> int32_t {
    if ((!models && n_models != 0u) || !camera80) return fail(BRT_ERR_INVALID_ARGUMENT, "null pointer");
    const TreeScene t = tree_scene_of((const Model*)models, n_models);
    Camera cam;
    std::memcpy(&cam, camera80, sizeof cam);
    uint32_t level = tree_level_for(t.scale, t.big, cam.position);
    if (tree_pads_equal(t, 0.0f, tree_reach_of(t.scale, level))) level = 0u;      // the tree of the scene's own extent already is that tree
    if (out_scene_scale) *out_scene_scale = t.scale;
    if (out_level) *out_level = level;
    if (out_reach) *out_reach = tree_reach_of(t.scale, level);
    return BRT_OK;
    });
}

int32_t brt_validate_scene(const void* models, uint32_t n_models, const void* materials, uint32_t n_materials,
                           const void* bvh_nodes, uint32_t n_nodes, uint32_t* out_max_depth) {
    return guard(nullptr, [&]() -> int32_t {
    EncodedScene e;
    std::string err;
    int32_t rc = validate_and_encode((const Model*)models, n_models, (const Material*)materials, n_materials,
                                     (const BVHNode*)bvh_nodes, n_nodes, &e, &err);
    if (rc != BRT_OK) return fail(rc, err);
    if (out_max_depth) *out_max_depth = e.max_leaf_depth;
    return BRT_OK;
    });
}

int32_t brt_scene_generate(uint32_t kind, uint64_t seed, void* out_models, void* out_materials, uint32_t capacity,
                           uint32_t* out_n_models) {
    return guard(nullptr, [&]() -> int32_t {
    if (!out_n_models) return fail(BRT_ERR_INVALID_ARGUMENT, "out_n_models is null");
    std::vector<Model> models;
    std::vector<Material> materials;
    int32_t rc = scene_generate(kind, seed, &models, &materials);
    if (rc != BRT_OK) return rc;
    *out_n_models = (uint32_t)models.size();
    if (models.size() > capacity || !out_models || !out_materials)
        return fail(BRT_ERR_CAPACITY, "scene has " + std::to_string(models.size()) + " spheres, capacity " + std::to_string(capacity));
    std::memcpy(out_models, models.data(), models.size() * sizeof(Model));
    std::memcpy(out_materials, materials.data(), materials.size() * sizeof(Material));
    return BRT_OK;
    });
}

// CameraExtract::extract_component (extract.rs:118-157) for
// Transform::from_translation(t).looking_at(target, up): direction = forward(), up = up().
int32_t brt_host_camera_extract(const float* t, const float* target, const float* up, float fov, float aspect_ratio,
                                float near_, float far_, uint32_t sample_count, uint32_t bounces, void* out_camera80) {
    return guard(nullptr, [&]() -> int32_t {
    if (!t || !target || !up || !out_camera80) return fail(BRT_ERR_INVALID_ARGUMENT, "null pointer");
    auto norm = [](float* v) { float l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); v[0] /= l; v[1] /= l; v[2] /= l; };
    float back[3] = {t[0] - target[0], t[1] - target[1], t[2] - target[2]};
    norm(back);
    float right[3] = {up[1] * back[2] - up[2] * back[1], up[2] * back[0] - up[0] * back[2], up[0] * back[1] - up[1] * back[0]};
    norm(right);
    float up2[3] = {back[1] * right[2] - back[2] * right[1], back[2] * right[0] - back[0] * right[2], back[0] * right[1] - back[1] * right[0]};
    Camera c;
    std::memset(&c, 0, sizeof c);
    c.sample_count = sample_count;
    c.bounce_count = bounces;
    c.projection_type = 0;
    c.near_ = near_; c.far_ = far_; c.fov = fov; c.aspect = aspect_ratio;
    for (int k = 0; k < 3; k++) { c.position[k] = t[k]; c.direction[k] = -back[k]; c.up[k] = up2[k]; }
    std::memcpy(out_camera80, &c, sizeof c);
    return BRT_OK;
    });
}

int32_t brt_host_window_extract(float random_seed, uint32_t physical_height, void* out_window16) {
    return guard(nullptr, [&]() -> int32_t {
    if (!out_window16) return fail(BRT_ERR_INVALID_ARGUMENT, "null pointer");
    Window w;
    std::memset(&w, 0, sizeof w);
    w.random_seed = random_seed;
    w.height = physical_height;
    std::memcpy(out_window16, &w, sizeof w);
    return BRT_OK;
    });
}

int32_t brt_host_material(const float* base_color_srgb3, float metallic, float perceptual_roughness, float reflectance,
                          float ior, float specular_transmission, void* out_material32) {
    return guard(nullptr, [&]() -> int32_t {
    if (!base_color_srgb3 || !out_material32) return fail(BRT_ERR_INVALID_ARGUMENT, "null pointer");
    StdMat m;
    for (int k = 0; k < 3; k++) m.base[k] = base_color_srgb3[k];
    m.metallic = metallic; m.perceptual_roughness = perceptual_roughness; m.reflectance = reflectance;
    m.ior = ior; m.specular_transmission = specular_transmission;
    Material r = prepare_asset(m);
    std::memcpy(out_material32, &r, sizeof r);
    return BRT_OK;
    });
}

int32_t brt_host_tile_order(const uint32_t* ray_sum, const uint32_t* longest_pixel, uint32_t n_tiles, uint32_t sample_count,
                            uint64_t grid_lanes, uint32_t sorted, uint32_t lane_permille, uint32_t tiles_x, uint32_t dilate,
                            uint32_t split_tail, uint32_t* out_order, uint32_t* out_info5) {
    return guard(nullptr, [&]() -> int32_t {
    if (!ray_sum || !longest_pixel || !out_order || !out_info5) return fail(BRT_ERR_INVALID_ARGUMENT, "null pointer");
    if (dilate != 0u && (tiles_x == 0u || n_tiles % tiles_x != 0u)) return fail(BRT_ERR_INVALID_ARGUMENT, "dilate needs a tiles_x that divides n_tiles");
    TileOrderParams tp{};
    tp.sample_count = sample_count; tp.grid_lanes = grid_lanes; tp.sorted = sorted; tp.sky_slack_permille = 20;
    tp.lane_permille = lane_permille; tp.critical = 1; tp.tiles_x = tiles_x; tp.dilate_x = tp.dilate_y = dilate;
    tp.split_tail = split_tail;
    TileOrder to;
    build_tile_order(ray_sum, longest_pixel, n_tiles, tp, &to);
    std::memcpy(out_order, to.order.data(), to.order.size() * 4);
    out_info5[0] = to.n_lane; out_info5[1] = to.n_critical; out_info5[2] = to.longest_pixel;
    out_info5[3] = to.n_nonsky; out_info5[4] = to.n_split;
    return BRT_OK;
    });
}

}  // extern "C"
