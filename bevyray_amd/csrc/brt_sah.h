// brt_sah.h -- arithmetic and rules shared by the two binned-SAH builders: the CPU one (brt_host.cpp build_bvh_sah, the
// statement of the rule) and the GPU one (brt_sah.hip, what brt_upload_scene runs when the caller passes no BVH).  Both
// produce BYTE-IDENTICAL trees (tested with memcmp): every quantity below is either an integer, a min / max over a SET
// (order independent by construction, see the keys), or a fixed expression of such values in f64 / f32 with no contraction.
//
// The tree replaces what the reference builds per frame with obvhs' PLOC (extract.rs:315-332, its own to-do at
// extract.rs:264-267 asks for a GPU builder); the shader only needs the node contract (raytrace.wgsl:325-341): node 0 is the
// root, an interior node's children are `index`, `index + 1`, a leaf has model_count 1 and `index` = model id.
//
// The rule (top down, one node = one range [begin, end) of the index list, depth d, preorder rank r among interior nodes):
//   * node box = union of the sphere boxes of the range; a sphere's box is its bounds padded by sah_model_pad (below; the
//     reference's Model::aabb, extract.rs:220-227, pads by a flat 0.1);
//   * count == 1: leaf.  Else the children live in slots 1 + 2r and 2 + 2r (what a depth-first builder that allocates two
//     slots per interior node in preorder hands out); left child: rank r + 1, right child: rank r + (left count);
//   * split position `mid`: halves of the current order, unless a binned-SAH split applies: count > 2, the depth budget is
//     not exhausted (d + ceil_log2(count) < kSahMaxDepth), and some axis has a finite, positive centroid extent.  16 bins
//     per axis over the centroid extent of the range; cost of splitting after bin b = area(left) * w(n_left) + area(right) *
//     w(n_right) in f64, w = sah_side_weight below: the count itself (the textbook surface area heuristic) in a node of up to
//     kSahDepthWeightMin spheres, a piecewise-linear log2 of it in the larger nodes at the top of the tree (both sides of such a node
//     are split on many more times, and what a side costs a ray that enters it grows with its DEPTH, not with its count);
//     the first (axis, bin) in axis-major order with the smallest cost < DBL_MAX wins; the range is
//     STABLY partitioned by "bin <= b" -- also when the split is then refused because the larger side would not fit the
//     depth budget (d + 1 + ceil_log2(larger) > kSahMaxDepth), in which case `mid` stays at the halves of the new order.
//   * giant leaves first (a last pass over the finished nodes, on the host for both builders: brt_host.cpp sah_giant_leaves_first):
//     where one child of a node is a LEAF whose box takes at least half of the node's surface and the other child is a subtree -- the
//     ground, a child of the root; a big sphere beside a cluster of small ones; decided on the boxes, not on a radius (round 5: "radius >
//     100") -- that leaf goes to slot `index + 1`, which the reference pops FIRST (raytrace.wgsl:329-341).  Every ray enters the ground's box;
//     popped last, its sphere test comes at the end of each lane's own walk, a few lanes at a time; popped first, all lanes of a wave
//     make it together in one full-width leaf step right after the root (config 2 -0.9 %, config 3 -1.5 ... -4 %, config 5 -0 ... -1.7 %;
//     same frames: profiles/r05/ground_first.txt).
#pragma once
#include <cstdint>

#include "brt_ploc.h"   // BRT_HD, PlocBox, ploc_model_box

namespace brt {

#ifndef BRT_SAH_BINS
#define BRT_SAH_BINS 16   // (the CPU twin takes any count; measured on the oracle: 24 / 32 / 64 bins walk 3 % fewer interior nodes per ray -- docs/experiments.md)
#endif
constexpr int kSahBins = BRT_SAH_BINS;
constexpr uint32_t kSahMaxDepth = 28;     // leaves at depth <= 28: stack_entries <= 29 < 31 (simple tree, brt_host.cpp)
constexpr double kSahDblMax = 1.7976931348623157e308;

BRT_HD uint32_t sah_f32_bits(float f) { union { float f; uint32_t u; } c; c.f = f; return c.u; }
BRT_HD float sah_bits_f32(uint32_t u) { union { float f; uint32_t u; } c; c.u = u; return c.f; }

// ---- order-independent, bit-exact min / max over sets of f32 ------------------------------------------------------------
// A float maps to a u32 KEY that is monotone in the float's value with -0 < +0; the minimum (maximum) of a set is the
// smallest (largest) key.  A NaN is ignored (a sphere with a NaN coordinate must not poison its ancestors): it maps to the
// key that loses every comparison, and a set of NaNs only gives the one canonical NaN of ploc_model_box back.  As keys the
// reductions are integer min / max: associative, commutative, the same bits in whatever order 1 or 1024 threads take them
// (ploc_min / ploc_max, brt_ploc.h, keep the first of -0 / +0 they see).
BRT_HD uint32_t sah_key(float f) {
    const uint32_t u = sah_f32_bits(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
BRT_HD float sah_unkey(uint32_t k) { return sah_bits_f32((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }
constexpr uint32_t kSahKeyMinIdentity = 0xffffffffu, kSahKeyMaxIdentity = 0u;   // never the key of a number (of a NaN only)
BRT_HD uint32_t sah_key_min(float f) { return (f != f) ? kSahKeyMinIdentity : sah_key(f); }
BRT_HD uint32_t sah_key_max(float f) { return (f != f) ? kSahKeyMaxIdentity : sah_key(f); }
BRT_HD float sah_unkey_min(uint32_t k) { return k == kSahKeyMinIdentity ? __builtin_nanf("") : sah_unkey(k); }
BRT_HD float sah_unkey_max(uint32_t k) { return k == kSahKeyMaxIdentity ? __builtin_nanf("") : sah_unkey(k); }

// a box as keys: mn[k] = key_min, mx[k] = key_max
struct SahKeyBox {
    uint32_t mn[3], mx[3];
};
BRT_HD SahKeyBox sah_keybox_empty() {
    SahKeyBox b;
    for (int k = 0; k < 3; k++) { b.mn[k] = kSahKeyMinIdentity; b.mx[k] = kSahKeyMaxIdentity; }
    return b;
}
BRT_HD SahKeyBox sah_keybox(const PlocBox& p) {
    SahKeyBox b;
    for (int k = 0; k < 3; k++) { b.mn[k] = sah_key_min(p.mn[k]); b.mx[k] = sah_key_max(p.mx[k]); }
    return b;
}
BRT_HD void sah_keybox_merge(SahKeyBox& a, const SahKeyBox& b) {
    for (int k = 0; k < 3; k++) {
        a.mn[k] = b.mn[k] < a.mn[k] ? b.mn[k] : a.mn[k];
        a.mx[k] = b.mx[k] > a.mx[k] ? b.mx[k] : a.mx[k];
    }
}
BRT_HD PlocBox sah_unkeybox(const SahKeyBox& b) {
    PlocBox p;
    for (int k = 0; k < 3; k++) { p.mn[k] = sah_unkey_min(b.mn[k]); p.mx[k] = sah_unkey_max(b.mx[k]); }
    return p;
}

BRT_HD bool sah_finite(double x) { return x - x == 0.0; }   // false for +-inf and NaN

// ---- leaf boxes of the tree the CALLEE builds -----------------------------------------------------------------------------
// The reference pads a sphere's box by a flat 0.1 (Model::aabb, extract.rs:220-227) before it hands the boxes to obvhs.  Boxes only
// cull: a pixel is the closest accepted sphere test among the leaves the walk reaches, and it reaches every sphere a ray can be
// accepted by as long as the box is CONSERVATIVE for the float arithmetic of the two tests -- a ray that the f32 sphere test accepts
// must pass the f32 slab test of the sphere's box.  A caller's tree is honoured as it comes; the tree this library builds itself pads
// by what that needs instead of by 0.1, because the pad is most of a small sphere's box (r = 0.2: half-width 0.3 -> 0.21, half the
// surface) and the walk pays for it: interior visits per ray 12.7 -> 11.2 (cover), 19.5 -> 17.1 (10 k grid), sphere tests 2.3 -> 1.8.
// What it needs: the sphere test's discriminant h*h - a*c carries a rounding error of ~2^-24 of its terms, dist^2 * a, i.e. it can
// accept a ray whose true distance from the centre exceeds r by up to ~2^-24 * dist^2 / (2 r), dist = the distance the ray has
// travelled to the sphere; the slab arithmetic's own error, ~2^-22 * dist, is small beside it.  dist is bounded by the REACH of the
// frame: rays start at the camera or on a sphere, so reach = max(2 S, |camera|_1 + S + L) with S the scene's scale (the largest
// |c|_1 + r over its ordinary spheres, r <= 100: a ground sphere of radius 1000 is not a distance rays travel between its own
// points -- it is convex) and L the longest tangent from the camera to such a big sphere (how far a primary ray can land on the ground
// away from the camera).  pad = clamp(2^-24 * reach^2 / r, 0.01, 0.1): twice the model's bound, the reference's own 0.1 where the
// model asks for more (there the reference's culling is as marginal as any), never less than 0.01.  The camera is not known at
// upload: brt_upload_scene builds for reach = 2 S, and a render call whose camera needs more rebuilds the tree on the GPU first
// (brt_api.cpp ensure_tree_reach; round 4 ignored the camera and lost pixels from a distance of ~260 on: VERDICT r4).
// Measured: on the 10 k grid (S = 100, pad 0.012) the first pixel of a 640 x 360 x 16 spp frame differs from the brute-force frame at
// a pad of 1e-4, none at 1e-3 (docs/experiments.md); the -m gpu suite compares the frames of configs 2 and 5 in this tree with the
// oracle's frames in the caller's 0.1-padded PLOC tree at full size, and far-camera frames (x 20 ... x 60 the cover distance).
// `reach` enters the builders as a FLOOR on the scale (scale = max(S, reach / 2)), as the max's initial key: still an integer max.
BRT_HD uint32_t sah_reach_key(float reach) {      // initial value of the scale's key-max (kSahKeyMaxIdentity: no floor)
    const float half = 0.5f * reach;
    return (half > 0.0f && half - half == 0.0f) ? sah_key_max(half) : kSahKeyMaxIdentity;
}
BRT_HD float sah_scale_term(const float* position, float radius) {      // |c|_1 + r of an ordinary sphere, else NaN (ignored by the max)
    const float s = ((__builtin_fabsf(position[0]) + __builtin_fabsf(position[1])) + __builtin_fabsf(position[2])) + radius;
    return (radius > 0.0f && radius <= 100.0f && s - s == 0.0f) ? s : __builtin_nanf("");
}
BRT_HD float sah_model_pad(float radius, float scale) {
    if (!(radius > 0.0f) || !(radius <= 100.0f) || !(scale - scale == 0.0f)) return 0.1f;
    const float d = 2.0f * scale;
    float p = (5.9604645e-8f * (d * d)) / radius;      // 2^-24
    p = p > 0.01f ? p : 0.01f;
    return p < 0.1f ? p : 0.1f;                         // (a NaN stays out by the tests above)
}
BRT_HD PlocBox sah_model_box(const float* position, float radius, float scale) {
    const float pad = radius + sah_model_pad(radius, scale);
    PlocBox b;
    for (int k = 0; k < 3; k++) {
        const float lo = position[k] - pad, hi = position[k] + pad;
        b.mn[k] = (lo == lo) ? lo : __builtin_nanf("");
        b.mx[k] = (hi == hi) ? hi : __builtin_nanf("");
    }
    return b;
}

// half the surface area in f64; a box that is not finite costs "everything" (never chosen: the winner must be < DBL_MAX)
BRT_HD double sah_half_area(const SahKeyBox& kb) {
    const PlocBox b = sah_unkeybox(kb);
    const double dx = (double)b.mx[0] - (double)b.mn[0], dy = (double)b.mx[1] - (double)b.mn[1], dz = (double)b.mx[2] - (double)b.mn[2];
    const double a = (dx * dy + dy * dz) + dz * dx;
    return sah_finite(a) ? a : kSahDblMax;
}

// centroid of a padded sphere box on one axis, f64; a box that is not finite sits at 0
BRT_HD double sah_centroid(const PlocBox& b, int k) {
    const double c = 0.5 * ((double)b.mn[k] + (double)b.mx[k]);
    return sah_finite(c) ? c : 0.0;
}

// bin of a centroid on an axis whose centroid extent is [cmin, cmin + ext], scale = kSahBins / ext
BRT_HD int sah_bin(double c, double cmin, double scale) {
    int b = (int)((c - cmin) * scale);
    return b < 0 ? 0 : (b >= kSahBins ? kSahBins - 1 : b);
}
// weight of a side of n >= 1 spheres in the split cost of a node of `node_count` spheres.  Large nodes: log2(n) + 2 with log2
// interpolated linearly between the powers of two -- e + (n / 2^e - 1), every step exact in f64 (integers, a division by a power of
// two), so the CPU and the GPU builder agree to the bit.  Measured on the GPU over six seeds of the cover and RTIOW scenes and two of
// the 10 004-sphere grid (1080p, 64 spp, 8 bounces; scripts/exp_tree_cost_seeds.py, profiles/r05/tree_cost_seeds.txt): kernel time
// -1.3 % / -2.0 % / -2.0 % against the count everywhere; the depth weight in EVERY node -0.5 % / -0.9 % / -1.6 % and config 3 (one
// 11 064-ray pixel long, in a leaf that ends up one level deeper) +7 %; with the threshold at 256 config 3 +2 %.
constexpr uint32_t kSahDepthWeightMin = 256;
BRT_HD double sah_side_weight(uint32_t n, uint32_t node_count) {
    if (node_count <= kSahDepthWeightMin) return (double)n;
    uint32_t e = 0;
    while ((n >> (e + 1u)) != 0u) e++;
    return (double)e + ((double)n / (double)(1u << e) - 1.0) + 2.0;
}
BRT_HD bool sah_axis_usable(double cmin, double cmax) {
    const double ext = cmax - cmin;
    return ext > 0.0 && sah_finite(ext);
}

BRT_HD uint32_t sah_ceil_log2(uint32_t c) {
    uint32_t d = 0;
    while ((1u << d) < c) d++;
    return d;
}

}  // namespace brt
