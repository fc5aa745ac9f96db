// brt_ploc.h -- arithmetic shared by the CPU and the GPU PLOC builders (brt_host.cpp,
// brt_bvh.hip) so that both produce byte-identical trees: same f32/f64 operations in the same
// order (-ffp-contract=off on both sides), same total order for nearest-neighbour ties, same
// node numbering.
//
// PLOC = Parallel Locally-Ordered Clustering (Meister & Bittner 2018): sort by Morton code of
// the AABB centre, then repeatedly merge clusters that are mutual nearest neighbours (by merged
// surface area) within a window of +-PLOC_SEARCH in the sorted order.  The reference calls
// obvhs::ploc::build_ploc::<24>(aabbs, identity, SortPrecision::U64, 0) (extract.rs:316-321);
// obvhs is not vendored, so the topology here is this builder's own.
//
// Node numbering (the reference's contract, extract.rs:323-332 + raytrace.wgsl:325-341):
// node 0 is the root, the children of an interior node are adjacent (`index`, `index + 1`), a
// leaf has model_count 1 and `index` = model id.  Clusters created by merging get temporary ids
// n, n+1, ... in creation order; the LAST one (id 2n-2) is the root.  Interior cluster with
// temporary id t has rank r = (2n-2) - t and its children live in slots 1 + 2r and 2 + 2r
// (later merges sit higher in the tree, so the top of the tree gets the low slots).
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define BRT_HD __host__ __device__ inline
#else
#define BRT_HD inline
#endif

namespace brt {

constexpr int PLOC_SEARCH = 24;   // build_ploc::<24>, extract.rs:316

struct PlocBox {
    float mn[3], mx[3];
};
// min/max that ignore a NaN operand (a sphere with a NaN coordinate must not poison the scene box
// or its ancestors, and the result must not depend on the order of a reduction)
BRT_HD float ploc_min(float a, float b) { return (b < a || a != a) ? b : a; }
BRT_HD float ploc_max(float a, float b) { return (a < b || a != a) ? b : a; }
BRT_HD PlocBox ploc_merge(const PlocBox& a, const PlocBox& b) {
    PlocBox r;
    for (int k = 0; k < 3; k++) { r.mn[k] = ploc_min(a.mn[k], b.mn[k]); r.mx[k] = ploc_max(a.mx[k], b.mx[k]); }
    return r;
}
BRT_HD float ploc_half_area(const PlocBox& b) {
    const float dx = b.mx[0] - b.mn[0], dy = b.mx[1] - b.mn[1], dz = b.mx[2] - b.mn[2];
    return (dx * dy + dy * dz) + dz * dx;
}
// Merge cost of two clusters; `first` is the one that comes first in the current cluster order, so
// cost(i, j) == cost(j, i) bit for bit, and NaN (inf - inf, NaN boxes) counts as +inf: together with
// ploc_better's tie rule the costs are strictly totally ordered and the cheapest pair of a round is
// always mutual, i.e. every round merges at least once whatever the input.
BRT_HD float ploc_pair_cost(const PlocBox& first, const PlocBox& second) {
    const float h = ploc_half_area(ploc_merge(first, second));
    return (h == h) ? h : __builtin_inff();
}
// Model::aabb, extract.rs:220-227: centre -+ (radius + 0.1)
BRT_HD PlocBox ploc_model_box(const float* position, float radius) {
    const float pad = radius + 0.1f;
    PlocBox b;
    for (int k = 0; k < 3; k++) {
        const float lo = position[k] - pad, hi = position[k] + pad;
        b.mn[k] = (lo == lo) ? lo : __builtin_nanf("");   // one NaN bit pattern on every machine (x86 and gfx950
        b.mx[k] = (hi == hi) ? hi : __builtin_nanf("");   // differ in the sign/payload they generate)
    }
    return b;
}
BRT_HD uint64_t ploc_spread21(uint64_t x) {
    x &= 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
// 63-bit Morton code of the box centre inside the scene box, 21 bits per axis, in f64
BRT_HD uint64_t ploc_morton(const PlocBox& b, const PlocBox& scene) {
    uint64_t q[3];
    for (int k = 0; k < 3; k++) {
        double ext = (double)scene.mx[k] - (double)scene.mn[k];
        if (!(ext > 1e-30)) ext = 1e-30;
        const double c = 0.5 * ((double)b.mn[k] + (double)b.mx[k]);
        double u = (c - (double)scene.mn[k]) / ext;
        if (!(u > 0.0)) u = 0.0;
        if (u > 1.0) u = 1.0;
        q[k] = (uint64_t)(u * 2097151.0);
    }
    return ploc_spread21(q[0]) | (ploc_spread21(q[1]) << 1) | (ploc_spread21(q[2]) << 2);
}
// Is (area a, pair {i,j}) strictly better than (area best, pair {i,bj})?  Strict total order
// (area, min, max): the globally smallest pair is then always mutual, so every round merges.
BRT_HD bool ploc_better(float a, int i, int j, float best, int bj) {
    if (bj < 0) return true;
    if (a < best) return true;
    if (!(a == best)) return false;
    const int p0 = i < j ? i : j, p1 = i < j ? j : i, q0 = i < bj ? i : bj, q1 = i < bj ? bj : i;
    return p0 < q0 || (p0 == q0 && p1 < q1);
}

}  // namespace brt
