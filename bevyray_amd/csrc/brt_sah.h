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
//   * node box = union of the padded sphere boxes (Model::aabb, extract.rs:220-227) of the range;
//   * count == 1: leaf.  Else the children live in slots 1 + 2r and 2 + 2r (what a depth-first builder that allocates two
//     slots per interior node in preorder hands out); left child: rank r + 1, right child: rank r + (left count);
//   * split position `mid`: halves of the current order, unless a binned-SAH split applies: count > 2, the depth budget is
//     not exhausted (d + ceil_log2(count) < kSahMaxDepth), and some axis has a finite, positive centroid extent.  16 bins
//     per axis over the centroid extent of the range; cost of splitting after bin b = area(left) * n_left + area(right) *
//     n_right in f64; the first (axis, bin) in axis-major order with the smallest cost < DBL_MAX wins; the range is
//     STABLY partitioned by "bin <= b" -- also when the split is then refused because the larger side would not fit the
//     depth budget (d + 1 + ceil_log2(larger) > kSahMaxDepth), in which case `mid` stays at the halves of the new order.
#pragma once
#include <cstdint>

#include "brt_ploc.h"   // BRT_HD, PlocBox, ploc_model_box

namespace brt {

constexpr int kSahBins = 16;
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
