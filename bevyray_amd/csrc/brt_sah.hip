// brt_sah.hip -- the binned-SAH BVH of brt_sah.h built on the GPU (gfx950): what brt_upload_scene runs when the caller
// passes no BVH (up to kMaxSahModels spheres), so that a scene that is re-uploaded every frame -- what the reference does,
// extract.rs:299-336, README.md:17 -- costs no host-side tree build (round 3's host builder: 8-9 ms on one core at 10 004
// spheres against a 21 ms frame).  Replaces extract.rs:315-332 for callee-built trees; byte-identical to the CPU statement of
// the rule (brt_host.cpp build_bvh_sah; tests compare with memcmp).
//
// A top-down build is a tree of dependent splits whose sizes fall from n to 2, so the work is cut where its shape changes:
//   k_sah_scale  grid      the scene's scale (an integer max over keys) for the leaf boxes' padding (brt_sah.h sah_model_pad);
//   k_sah_prep   grid      sphere boxes as keys (brt_sah.h), f64 centroids, identity index list, the root task;
//   k_sah_top    per level (scenes of more than kSubMax spheres) A WORKGROUP PER NODE of more than kSubMax spheres, the nodes of a
//                          level side by side: each split a few barrier-separated passes of all 1024 threads over the node's range
//                          in global memory; children of more than kSubMax spheres go to the next level's list, smaller ones become
//                          subtree tasks.  ceil(log2(n / kSubMax)) + 2 levels are launched, then ONE workgroup finishes whatever a
//                          lopsided top has left (depth first);
//   k_sah_sub    a workgroup per subtree task (<= kSubMax spheres), the subtree's boxes, centroids and index lists STAGED IN LDS:
//                          level by level, A WAVE PER NODE for nodes of more than kLaneMax spheres (no barrier inside a split: a wave
//                          is its own team) and A LANE PER NODE below that -- seven of eight interior nodes of a tree hold at most 8
//                          spheres, and as wave splits each is ~2.5 us of dependent latency.
// No host round trip, no grid-wide synchronisation; subtrees are independent, so the bottom of the tree -- where nearly all the
// n - 1 splits are -- runs on as many CUs as there are subtrees.  10 004 spheres: scale + prep 8 us, top 5 levels 76 / 74 / 50 / 38 /
// 30 us, subtrees 190 us (profiles/r04/sah_build_kernel_breakdown.txt).
//
// One split by a team (sah_split): (1) node box and centroid extent of the range -- lanes stride over the range, wave shuffles (and
// LDS across waves for a block team) reduce; (2) binning: a lane per sphere, LDS atomics (integer min / max on keys, add on the counts)
// into 3 x 16 bins, each bound LOOKED AT before the atomic (a bound only moves one way: clustered spheres then cost a broadcast read
// instead of a 64-way serialised atomic); (3) 48 lanes evaluate the 3 x 15 split candidates (left / right box and count of candidate
// (axis, bin) from the bins), a wave argmin over (cost, axis, bin) picks the CPU loop's winner; (4) stable partition of the range into
// the other index buffer by ballot + mbcnt ranks.  Every quantity is an integer or a min / max over a set, so the order in which
// lanes arrive does not show in the result.
#include <hip/hip_runtime.h>

#include "brt_kernels.h"
#include "brt_sah.h"

namespace brt {

namespace {

constexpr uint32_t SB = 1024, SW = SB / 64;
constexpr uint32_t kSubMax = 1024;      // subtrees of at most this many spheres are built by one workgroup each
constexpr uint32_t kListCap = kSubMax / 2;   // nodes of >= 2 spheres in one level of a subtree
constexpr uint32_t kNBins = 3 * kSahBins;
static_assert(kNBins <= 64, "one lane per split candidate (step 3 of split_range): more bins need a loop there");
constexpr uint32_t kTopCounters = 32;

struct SahTask {
    uint32_t slot, begin, end, rank;
    uint32_t depth_buf;      // depth | index buffer << 8
};

struct SahBins {             // identity: count 0, mn = kSahKeyMinIdentity, mx = kSahKeyMaxIdentity
    uint32_t count[kNBins];
    uint32_t mn[3][kNBins], mx[3][kNBins];
};

// Where a team reads spheres from.  GlobalStore: the scene's arrays in global memory, handles = model ids (k_sah_top).
// LocalStore: ONE subtree staged in LDS by its workgroup (k_sah_sub) -- boxes, centroids and both index buffers, structure of
// arrays, handles and positions relative to the subtree: a split is a chain of dependent passes over its range (index -> box /
// centroid), and from L2 each link of that chain costs ~0.7 us against ~0.05 us from LDS.
struct GlobalStore {
    const SahKeyBox* kbox;   // per sphere
    const double* cen;       // 3 per sphere
    uint32_t* idx[2];        // index list, two buffers (a partition writes the other one)
    __device__ __forceinline__ uint32_t get(uint32_t buf, uint32_t i) const { return idx[buf][i]; }
    __device__ __forceinline__ void put(uint32_t buf, uint32_t i, uint32_t m) const { idx[buf][i] = m; }
    __device__ __forceinline__ SahKeyBox box(uint32_t m) const { return kbox[m]; }
    __device__ __forceinline__ double centroid(uint32_t m, int k) const { return cen[3 * (size_t)m + k]; }
    __device__ __forceinline__ uint32_t model(uint32_t m) const { return m; }
};
struct LocalStore {
    uint32_t* kb;            // [6][kSubMax]: min keys x, y, z, max keys x, y, z
    double* cen;             // [3][kSubMax]
    uint32_t* id;            // [kSubMax]: model id of a handle
    uint16_t* idx;           // [2][kSubMax]
    __device__ __forceinline__ uint32_t get(uint32_t buf, uint32_t i) const { return idx[buf * kSubMax + i]; }
    __device__ __forceinline__ void put(uint32_t buf, uint32_t i, uint32_t m) const { idx[buf * kSubMax + i] = (uint16_t)m; }
    __device__ __forceinline__ SahKeyBox box(uint32_t m) const {
        SahKeyBox b;
        for (int k = 0; k < 3; k++) { b.mn[k] = kb[k * kSubMax + m]; b.mx[k] = kb[(3 + k) * kSubMax + m]; }
        return b;
    }
    __device__ __forceinline__ double centroid(uint32_t m, int k) const { return cen[k * kSubMax + m]; }
    __device__ __forceinline__ uint32_t model(uint32_t m) const { return id[m]; }
};
constexpr size_t kLocalStoreBytes = (size_t)kSubMax * (3 * 8 + 6 * 4 + 4 + 2 * 2);

struct SahGlobals {
    GlobalStore st;
    BVHNode* out;
    SahTask* tasks;          // subtree tasks (k_sah_top -> k_sah_sub), ranges in global positions
    uint32_t* n_tasks;
};

struct SahShared {           // block-team scratch
    SahKeyBox part_box[SW];
    double part_c[SW][6];
    uint32_t part_n[SW];
    int best_axis, best_bin;
    uint32_t n_left;
};

__device__ __forceinline__ uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
__device__ __forceinline__ uint32_t mbcnt64(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}
// memory written by this wave (LDS or global) is visible to its other lanes / the team's other waves after this
__device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); }
template <bool COOP>
__device__ __forceinline__ void team_sync() {
    if (COOP) __syncthreads();
    else wave_fence();
}

__device__ __forceinline__ void write_node(BVHNode* out, uint32_t slot, const SahKeyBox& kb, uint32_t index, uint32_t count) {
    const PlocBox b = sah_unkeybox(kb);
    BVHNode o;
    o._pad0 = 0.0f; o._pad1[0] = o._pad1[1] = o._pad1[2] = 0u;
    for (int k = 0; k < 3; k++) { o.bounds_min[k] = b.mn[k]; o.bounds_max[k] = b.mx[k]; }
    o.index = index;
    o.model_count = count;
    out[slot] = o;
}

// element k of a 3-array held in registers (a dynamic index would send the array to scratch memory)
template <typename T>
__device__ __forceinline__ T pick3(const T (&a)[3], int k) { return k == 0 ? a[0] : (k == 1 ? a[1] : a[2]); }

struct SplitResult {
    uint32_t mid;
    uint32_t buf;            // index buffer that holds the children's ranges
};

// One node of the rule of brt_sah.h, by a team: the whole block (COOP) or one wave.  `t` is team-uniform (ranges in the
// store's positions).  Writes the node (interior) and returns where its range was cut; the caller deals with the children.
template <bool COOP, typename Store>
__device__ SplitResult sah_split(const Store& st, BVHNode* out, const SahTask& t, SahBins* bins, SahShared* sh) {
    const uint32_t lane = lane_id();
    const uint32_t tid = COOP ? threadIdx.x : lane, team = COOP ? SB : 64u, wave = threadIdx.x >> 6;
    const uint32_t begin = t.begin, end = t.end, count = end - begin, depth = t.depth_buf & 0xffu, buf = t.depth_buf >> 8;

    // (1) node box, centroid extent
    SahKeyBox nb = sah_keybox_empty();
    double cmin[3] = {kSahDblMax, kSahDblMax, kSahDblMax}, cmax[3] = {-kSahDblMax, -kSahDblMax, -kSahDblMax};
#pragma unroll 1
    for (uint32_t i = begin + tid; i < end; i += team) {
        const uint32_t m = st.get(buf, i);
        sah_keybox_merge(nb, st.box(m));
        for (int k = 0; k < 3; k++) {
            const double c = st.centroid(m, k);
            cmin[k] = c < cmin[k] ? c : cmin[k];
            cmax[k] = c > cmax[k] ? c : cmax[k];
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        SahKeyBox o;
        for (int k = 0; k < 3; k++) { o.mn[k] = __shfl_xor(nb.mn[k], off, 64); o.mx[k] = __shfl_xor(nb.mx[k], off, 64); }
        sah_keybox_merge(nb, o);
        for (int k = 0; k < 3; k++) {
            const double a = __shfl_xor(cmin[k], off, 64), b = __shfl_xor(cmax[k], off, 64);
            cmin[k] = a < cmin[k] ? a : cmin[k];
            cmax[k] = b > cmax[k] ? b : cmax[k];
        }
    }
    if (COOP) {
        if (lane == 0) {
            sh->part_box[wave] = nb;
            for (int k = 0; k < 3; k++) { sh->part_c[wave][k] = cmin[k]; sh->part_c[wave][3 + k] = cmax[k]; }
        }
        __syncthreads();
        nb = sh->part_box[0];
        for (int k = 0; k < 3; k++) { cmin[k] = sh->part_c[0][k]; cmax[k] = sh->part_c[0][3 + k]; }
#pragma unroll 1
        for (uint32_t w = 1; w < SW; w++) {
            sah_keybox_merge(nb, sh->part_box[w]);
            for (int k = 0; k < 3; k++) {
                const double a = sh->part_c[w][k], b = sh->part_c[w][3 + k];
                cmin[k] = a < cmin[k] ? a : cmin[k];
                cmax[k] = b > cmax[k] ? b : cmax[k];
            }
        }
    }

    SplitResult res;
    res.mid = begin + count / 2;                      // halves of the current order
    res.buf = buf;
    const bool balanced_only = depth + sah_ceil_log2(count) >= kSahMaxDepth;
    bool usable[3];
    double scale[3];
    bool any = false;
    for (int k = 0; k < 3; k++) {
        usable[k] = sah_axis_usable(cmin[k], cmax[k]);
        scale[k] = usable[k] ? (double)kSahBins / (cmax[k] - cmin[k]) : 0.0;
        any = any || usable[k];
    }
    if (!balanced_only && count > 2 && any) {
        // (2) binning
        uint32_t* words = reinterpret_cast<uint32_t*>(bins);
        for (uint32_t w = tid; w < sizeof(SahBins) / 4u; w += team)
            words[w] = w < kNBins ? 0u : (w < 4u * kNBins ? kSahKeyMinIdentity : kSahKeyMaxIdentity);
        team_sync<COOP>();
#pragma unroll 1
        for (uint32_t i = begin + tid; i < end; i += team) {
            const uint32_t m = st.get(buf, i);
            const SahKeyBox kb = st.box(m);
            for (int k = 0; k < 3; k++) {
                if (!usable[k]) continue;
                const uint32_t b = (uint32_t)k * kSahBins + (uint32_t)sah_bin(st.centroid(m, k), cmin[k], scale[k]);
                atomicAdd(&bins->count[b], 1u);
                // (look before the atomic: a bound only ever moves one way, so a value that would not move what the bin holds NOW
                //  never will -- clustered spheres, e.g. 10 000 of one height in one bin, then cost a broadcast read instead of a
                //  64-way serialised atomic each)
                for (int j = 0; j < 3; j++) {
                    if (kb.mn[j] < __atomic_load_n(&bins->mn[j][b], __ATOMIC_RELAXED)) atomicMin(&bins->mn[j][b], kb.mn[j]);
                    if (kb.mx[j] > __atomic_load_n(&bins->mx[j][b], __ATOMIC_RELAXED)) atomicMax(&bins->mx[j][b], kb.mx[j]);
                }
            }
        }
        team_sync<COOP>();
        // (3) the 3 x 15 candidates: lane l = axis l / 16, split after bin l % 16
        int best_axis = -1, best_bin = -1;
        uint32_t n_left = 0;
        if (!COOP || wave == 0) {
            double cost = __builtin_inf();
            uint32_t nl = 0;
            if (lane < kNBins) {
                const int k = (int)(lane / kSahBins), s = (int)(lane % kSahBins);
                if (pick3(usable, k) && s + 1 < kSahBins) {
                    SahKeyBox L = sah_keybox_empty(), R = sah_keybox_empty();
                    uint32_t nr = 0;
#pragma unroll 2
                    for (int b = 0; b < kSahBins; b++) {
                        const uint32_t w = (uint32_t)k * kSahBins + (uint32_t)b;
                        SahKeyBox bb;
                        for (int j = 0; j < 3; j++) { bb.mn[j] = bins->mn[j][w]; bb.mx[j] = bins->mx[j][w]; }
                        if (b <= s) { sah_keybox_merge(L, bb); nl += bins->count[w]; }
                        else { sah_keybox_merge(R, bb); nr += bins->count[w]; }
                    }
                    if (nl != 0u && nr != 0u) {
                        const double c = sah_half_area(L) * sah_side_weight(nl, count) + sah_half_area(R) * sah_side_weight(nr, count);
                        if (c < kSahDblMax) cost = c;          // the winner must beat DBL_MAX (strict), as in the CPU loop
                    }
                }
            }
            // argmin over (cost, lane): the first (axis, bin) in axis-major order among the cheapest
            double bc = cost;
            uint32_t bl = lane;
            for (int off = 32; off > 0; off >>= 1) {
                const double oc = __shfl_xor(bc, off, 64);
                const uint32_t ol = __shfl_xor(bl, off, 64);
                if (oc < bc || (oc == bc && ol < bl)) { bc = oc; bl = ol; }
            }
            if (bc < __builtin_inf()) {
                best_axis = (int)(bl / kSahBins);
                best_bin = (int)(bl % kSahBins);
                n_left = __shfl(nl, (int)bl, 64);
            }
            if (COOP && lane == 0) { sh->best_axis = best_axis; sh->best_bin = best_bin; sh->n_left = n_left; }
        }
        if (COOP) {
            __syncthreads();
            best_axis = sh->best_axis; best_bin = sh->best_bin; n_left = sh->n_left;
        }
        if (best_axis >= 0) {
            // (4) stable partition by "bin <= best_bin" into the other buffer
            const double cm = pick3(cmin, best_axis), sc = pick3(scale, best_axis);
            uint32_t w_begin = begin, w_end = end, lo = begin, ro = begin + n_left;
            if (COOP) {
                const uint32_t chunk = ((count + SW - 1u) / SW + 63u) & ~63u;
                w_begin = begin + wave * chunk < end ? begin + wave * chunk : end;
                w_end = w_begin + chunk < end ? w_begin + chunk : end;
                uint32_t cl = 0;
                for (uint32_t i0 = w_begin; i0 < w_end; i0 += 64u) {
                    const uint32_t i = i0 + lane;
                    const bool left = i < w_end && sah_bin(st.centroid(st.get(buf, i), best_axis), cm, sc) <= best_bin;
                    cl += (uint32_t)__popcll(__ballot(left));
                }
                if (lane == 0) sh->part_n[wave] = cl;
                __syncthreads();
#pragma unroll 1
                for (uint32_t w = 0; w < wave; w++) {
                    const uint32_t wb = begin + w * chunk < end ? begin + w * chunk : end, we = wb + chunk < end ? wb + chunk : end;
                    lo += sh->part_n[w];
                    ro += (we - wb) - sh->part_n[w];
                }
            }
            for (uint32_t i0 = w_begin; i0 < w_end; i0 += 64u) {
                const uint32_t i = i0 + lane;
                const bool valid = i < w_end;
                const uint32_t m = valid ? st.get(buf, i) : 0u;
                const bool left = valid && sah_bin(st.centroid(m, best_axis), cm, sc) <= best_bin;
                const bool right = valid && !left;
                const uint64_t ml = __ballot(left), mr = __ballot(right);
                if (left) st.put(buf ^ 1u, lo + mbcnt64(ml), m);
                if (right) st.put(buf ^ 1u, ro + mbcnt64(mr), m);
                lo += (uint32_t)__popcll(ml);
                ro += (uint32_t)__popcll(mr);
            }
            team_sync<COOP>();
            res.buf = buf ^ 1u;
            const uint32_t m = begin + n_left;
            // a lopsided split must leave both sides inside the depth budget; else halves (of the new order)
            if (m > begin && m < end) {
                const uint32_t big = (m - begin) > (end - m) ? (m - begin) : (end - m);
                if (depth + 1u + sah_ceil_log2(big) <= kSahMaxDepth) res.mid = m;
            }
        }
    }
    if (tid == 0) write_node(out, t.slot, nb, 1u + 2u * t.rank, 0u);
    return res;
}

// The same node by ONE LANE, for nodes of at most kLaneMax spheres -- seven of eight interior nodes of a tree, and as wave
// splits each a chain of ~2.5 us of dependent latency.  64 such nodes per wave at once instead.  The spheres' boxes sit in
// registers, their bins per axis packed four bits each in one word.  The candidates are the rule's: cutting after bin s gives
// the same two sets for every s from one occupied bin up to the next, and the rule takes the first s of the cheapest, so only
// s = an occupied bin (a sphere's own bin) can win: at most `count` candidates per axis instead of 15.
constexpr uint32_t kLaneMax = 8;
template <typename Store>
__device__ SplitResult sah_split_lane(const Store& st, BVHNode* out, const SahTask& t) {
    const uint32_t begin = t.begin, count = t.end - t.begin, depth = t.depth_buf & 0xffu, buf = t.depth_buf >> 8;
    SahKeyBox bx[kLaneMax], nb = sah_keybox_empty();
    uint32_t h[kLaneMax];
    double cmin[3] = {kSahDblMax, kSahDblMax, kSahDblMax}, cmax[3] = {-kSahDblMax, -kSahDblMax, -kSahDblMax};
#pragma unroll
    for (uint32_t q = 0; q < kLaneMax; q++) {
        bx[q] = sah_keybox_empty();
        h[q] = 0u;
        if (q < count) {
            h[q] = st.get(buf, begin + q);
            bx[q] = st.box(h[q]);
            sah_keybox_merge(nb, bx[q]);
            for (int k = 0; k < 3; k++) {
                const double c = st.centroid(h[q], k);
                cmin[k] = c < cmin[k] ? c : cmin[k];
                cmax[k] = c > cmax[k] ? c : cmax[k];
            }
        }
    }
    SplitResult res;
    res.mid = begin + count / 2;
    res.buf = buf;
    const bool balanced_only = depth + sah_ceil_log2(count) >= kSahMaxDepth;
    if (!balanced_only && count > 2) {
        double best_cost = kSahDblMax;
        int best_axis = -1, best_bin = -1;
        uint32_t best_nl = 0, best_packed = 0;
#pragma unroll 1
        for (int k = 0; k < 3; k++) {
            const double cm = pick3(cmin, k), cx = pick3(cmax, k);
            if (!sah_axis_usable(cm, cx)) continue;
            const double scale = (double)kSahBins / (cx - cm);
            uint32_t packed = 0;
#pragma unroll
            for (uint32_t q = 0; q < kLaneMax; q++)
                if (q < count) packed |= (uint32_t)sah_bin(st.centroid(h[q], k), cm, scale) << (4u * q);
#pragma unroll 1
            for (uint32_t p = 0; p < count; p++) {
                const int s = (int)((packed >> (4u * p)) & 15u);
                SahKeyBox L = sah_keybox_empty(), R = sah_keybox_empty();
                uint32_t nl = 0, nr = 0;
#pragma unroll
                for (uint32_t q = 0; q < kLaneMax; q++) {
                    if (q < count) {
                        if ((int)((packed >> (4u * q)) & 15u) <= s) { sah_keybox_merge(L, bx[q]); nl++; }
                        else { sah_keybox_merge(R, bx[q]); nr++; }
                    }
                }
                if (nl == 0u || nr == 0u) continue;
                const double cost = sah_half_area(L) * sah_side_weight(nl, count) + sah_half_area(R) * sah_side_weight(nr, count);
                // the CPU loop's order is (axis, bin) ascending with a strict <: the axes come in that order here, the bins do not
                if (cost < best_cost || (cost == best_cost && k == best_axis && s < best_bin)) {
                    best_cost = cost; best_axis = k; best_bin = s; best_nl = nl; best_packed = packed;
                }
            }
        }
        if (best_axis >= 0) {
            uint32_t lo = begin, ro = begin + best_nl;
#pragma unroll
            for (uint32_t q = 0; q < kLaneMax; q++)
                if (q < count) {
                    const bool left = (int)((best_packed >> (4u * q)) & 15u) <= best_bin;
                    st.put(buf ^ 1u, left ? lo : ro, h[q]);
                    lo += left ? 1u : 0u;
                    ro += left ? 0u : 1u;
                }
            res.buf = buf ^ 1u;
            const uint32_t m = begin + best_nl, end = t.end;
            if (m > begin && m < end) {
                const uint32_t big = (m - begin) > (end - m) ? (m - begin) : (end - m);
                if (depth + 1u + sah_ceil_log2(big) <= kSahMaxDepth) res.mid = m;
            }
        }
    }
    write_node(out, t.slot, nb, 1u + 2u * t.rank, 0u);
    return res;
}

// the children of a split: tasks for ranges of >= 2 spheres (handed to `emit`), leaves written at once
template <typename Store, typename Emit>
__device__ __forceinline__ void sah_children(const Store& st, BVHNode* out, const SahTask& t, const SplitResult& r, bool writer, Emit emit) {
    const uint32_t depth = t.depth_buf & 0xffu, child = 1u + 2u * t.rank;
    SahTask c[2];
    c[0].slot = child;      c[0].begin = t.begin; c[0].end = r.mid; c[0].rank = t.rank + 1u;
    c[1].slot = child + 1u; c[1].begin = r.mid;   c[1].end = t.end; c[1].rank = t.rank + (r.mid - t.begin);
    for (int s = 0; s < 2; s++) {
        c[s].depth_buf = (depth + 1u) | (r.buf << 8);
        if (c[s].end - c[s].begin == 1u) {
            if (writer) {
                const uint32_t m = st.get(r.buf, c[s].begin);
                write_node(out, c[s].slot, st.box(m), st.model(m), 1u);     // leaf: the model id itself (extract.rs:318,329)
            }
        } else {
            emit(c[s]);
        }
    }
}

// Splits by the whole block.  Children of more than `coop_min` spheres: with `lifo`, split by this block in turn (depth first: at
// most depth + 1 stack entries), else handed to `big(task)`; smaller ones (>= 2 spheres) go to `small(task)` (both called by
// thread 0 only).
template <typename Store, typename Big, typename Small>
__device__ void sah_coop_phase(const Store& st, BVHNode* out, const SahTask& root, uint32_t coop_min, bool lifo, SahBins* bins,
                               SahShared* sh, SahTask* stack, uint32_t* stack_n, Big big, Small small) {
    if (threadIdx.x == 0) { stack[0] = root; *stack_n = 1u; }
    for (;;) {
        __syncthreads();
        const uint32_t n = *stack_n;
        if (n == 0u) break;
        const SahTask t = stack[n - 1u];
        __syncthreads();
        if (threadIdx.x == 0) *stack_n = n - 1u;
        const SplitResult r = sah_split<true>(st, out, t, bins, sh);
        __syncthreads();
        sah_children(st, out, t, r, threadIdx.x == 0, [&](const SahTask& c) {
            if (threadIdx.x != 0) return;
            if (c.end - c.begin <= coop_min) small(c);
            else if (lifo) stack[(*stack_n)++] = c;
            else big(c);
        });
    }
}

}  // namespace

// the scene's scale (brt_sah.h sah_scale_term): an integer max over keys, the same bits in whatever order the atomics land
__global__ __launch_bounds__(256) void k_sah_scale(const Model* __restrict__ models, uint32_t n, uint32_t* scale_key) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t k = i < n ? sah_key_max(sah_scale_term(models[i].position, models[i].radius)) : kSahKeyMaxIdentity;
    for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = __shfl_xor(k, off, 64);
        k = o > k ? o : k;
    }
    if ((threadIdx.x & 63u) == 0u && k != kSahKeyMaxIdentity) atomicMax(scale_key, k);
}

__global__ __launch_bounds__(256) void k_sah_prep(const Model* __restrict__ models, uint32_t n, const uint32_t* scale_key, SahKeyBox* kbox, double* cen,
                                                  uint32_t* idx0, SahTask* tasks, uint32_t* n_tasks, SahTask* top_list, uint32_t* top_n,
                                                  uint32_t* info) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        SahTask r;
        r.slot = 0u; r.begin = 0u; r.end = n; r.rank = 0u; r.depth_buf = 0u;
        tasks[0] = r;
        *n_tasks = n <= kSubMax ? 1u : 0u;      // a small scene is one subtree task; else k_sah_top emits them
        top_list[0] = r;                        // level 0 of the top of the tree
        for (uint32_t l = 0; l < kTopCounters; l++) top_n[l] = l == 0u ? 1u : 0u;
        info[0] = 2u * n - 1u;
        info[1] = 0u;
    }
    if (i >= n) return;
    const PlocBox b = sah_model_box(models[i].position, models[i].radius, sah_unkey_max(*scale_key));
    kbox[i] = sah_keybox(b);
    for (int k = 0; k < 3; k++) cen[3 * (size_t)i + k] = sah_centroid(b, k);
    idx0[i] = i;
}

// One level of the top of the tree: a block per node of more than kSubMax spheres (in[0 .. *n_in)); children of more than kSubMax
// spheres go to the next level's list, smaller ones become subtree tasks.  `lifo`: ONE block finishes whatever is left below
// the levels that were launched (a lopsided top; the depth budget bounds it).
__global__ __launch_bounds__(SB) void k_sah_top(SahGlobals g, const SahTask* in, const uint32_t* n_in, SahTask* next, uint32_t* n_next,
                                                uint32_t lifo) {
    __shared__ SahBins bins;
    __shared__ SahShared sh;
    __shared__ SahTask stack[64];
    __shared__ uint32_t stack_n;
    const uint32_t n = *n_in;
    for (uint32_t ti = blockIdx.x; ti < n; ti += gridDim.x) {
        const SahTask root = in[ti];
        sah_coop_phase(g.st, g.out, root, kSubMax, lifo != 0u, &bins, &sh, stack, &stack_n,
                       [&](const SahTask& c) { next[atomicAdd(n_next, 1u)] = c; },
                       [&](const SahTask& c) { g.tasks[atomicAdd(g.n_tasks, 1u)] = c; });
        __syncthreads();
    }
}

__global__ __launch_bounds__(SB) void k_sah_sub(SahGlobals g) {
    extern __shared__ double sah_dyn[];
    __shared__ SahBins bins[SW];
    __shared__ SahTask list[2][kListCap];       // nodes of more than kLaneMax spheres: a wave each
    __shared__ SahTask lane_list[2][kListCap];  // nodes of 2 .. kLaneMax spheres: a lane each
    __shared__ uint32_t list_n[2], lane_n[2];
    LocalStore st;
    st.cen = sah_dyn;
    st.kb = reinterpret_cast<uint32_t*>(st.cen + 3 * kSubMax);
    st.id = st.kb + 6 * kSubMax;
    st.idx = reinterpret_cast<uint16_t*>(st.id + kSubMax);
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    const uint32_t n_tasks = *g.n_tasks;
    for (uint32_t ti = blockIdx.x; ti < n_tasks; ti += gridDim.x) {
        __syncthreads();
        if (threadIdx.x == 0) { list_n[0] = list_n[1] = lane_n[0] = lane_n[1] = 0u; }
        const SahTask groot = g.tasks[ti];
        const uint32_t count = groot.end - groot.begin;
        // stage the subtree: handle j = the j-th sphere of the task's range
        for (uint32_t j = threadIdx.x; j < count; j += SB) {
            const uint32_t m = g.st.get(groot.depth_buf >> 8, groot.begin + j);
            const SahKeyBox kb = g.st.box(m);
            for (int k = 0; k < 3; k++) {
                st.kb[k * kSubMax + j] = kb.mn[k];
                st.kb[(3 + k) * kSubMax + j] = kb.mx[k];
                st.cen[k * kSubMax + j] = g.st.centroid(m, k);
            }
            st.id[j] = m;
            st.idx[j] = (uint16_t)j;
        }
        __syncthreads();
        SahTask root = groot;                     // the same node in the store's positions
        root.begin = 0u; root.end = count; root.depth_buf = groot.depth_buf & 0xffu;
        if (count == 1u) {                        // a scene of one sphere: the root is a leaf (emitted tasks hold >= 2 spheres)
            if (threadIdx.x == 0) write_node(g.out, root.slot, st.box(0u), st.model(0u), 1u);
            continue;
        }
        if (threadIdx.x == 0) {
            if (count > kLaneMax) { list[0][0] = root; list_n[0] = 1u; }
            else { lane_list[0][0] = root; lane_n[0] = 1u; }
        }
        // level by level: the level's larger nodes a wave each, then its small ones a lane each
        uint32_t cur = 0u;
        for (;;) {
            __syncthreads();
            const uint32_t n_cur = list_n[cur], l_cur = lane_n[cur];
            __syncthreads();
            if (n_cur == 0u && l_cur == 0u) break;
            if (threadIdx.x == 0) { list_n[cur] = 0u; lane_n[cur] = 0u; }       // the lists after next
            auto emit = [&](const SahTask& c) {
                if (c.end - c.begin > kLaneMax) list[cur ^ 1u][atomicAdd(&list_n[cur ^ 1u], 1u)] = c;
                else lane_list[cur ^ 1u][atomicAdd(&lane_n[cur ^ 1u], 1u)] = c;
            };
            for (uint32_t e = wave; e < n_cur; e += SW) {
                const SahTask t = list[cur][e];
                const SplitResult r = sah_split<false>(st, g.out, t, &bins[wave], nullptr);
                sah_children(st, g.out, t, r, lane == 0, [&](const SahTask& c) { if (lane == 0) emit(c); });
            }
            for (uint32_t e = threadIdx.x; e < l_cur; e += SB) {
                const SahTask t = lane_list[cur][e];
                const SplitResult r = sah_split_lane(st, g.out, t);
                sah_children(st, g.out, t, r, true, emit);
            }
            cur ^= 1u;
        }
    }
}

// levels of k_sah_top launched side by side before the one-block finish: enough for a balanced top plus two
static uint32_t sah_top_levels(uint32_t n) {
    uint32_t l = 0;
    while (((size_t)kSubMax << l) < n) l++;
    return l + 2u;
}
static size_t sah_top_list_tasks(uint32_t n) { return (size_t)n / kSubMax + 2; }   // nodes of > kSubMax spheres in one level

size_t sah_scratch_bytes(uint32_t n) {
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    return al((size_t)n * sizeof(SahKeyBox)) + al((size_t)n * 3 * sizeof(double)) + 2 * al((size_t)n * 4) +
           al(((size_t)n / 2 + 2) * sizeof(SahTask)) + al(256) + 2 * al(sah_top_list_tasks(n) * sizeof(SahTask)) + al(256) +
           al(2 * (size_t)n * sizeof(BVHNode)) + al(256);
}

hipError_t launch_build_sah(const Model* d_models, uint32_t n, float reach, char* d_scratch, BVHNode** d_out, uint32_t** d_info, hipStream_t stream) {
    auto take = [&](size_t bytes) { char* r = d_scratch; d_scratch += (bytes + 255) & ~(size_t)255; return r; };
    SahGlobals g;
    SahKeyBox* kbox = reinterpret_cast<SahKeyBox*>(take((size_t)n * sizeof(SahKeyBox)));
    double* cen = reinterpret_cast<double*>(take((size_t)n * 3 * sizeof(double)));
    g.st.kbox = kbox;
    g.st.cen = cen;
    g.st.idx[0] = reinterpret_cast<uint32_t*>(take((size_t)n * 4));
    g.st.idx[1] = reinterpret_cast<uint32_t*>(take((size_t)n * 4));
    g.tasks = reinterpret_cast<SahTask*>(take(((size_t)n / 2 + 2) * sizeof(SahTask)));
    g.n_tasks = reinterpret_cast<uint32_t*>(take(256));
    SahTask* top_list[2];
    top_list[0] = reinterpret_cast<SahTask*>(take(sah_top_list_tasks(n) * sizeof(SahTask)));
    top_list[1] = reinterpret_cast<SahTask*>(take(sah_top_list_tasks(n) * sizeof(SahTask)));
    uint32_t* top_n = reinterpret_cast<uint32_t*>(take(256));          // kTopCounters level counters
    g.out = reinterpret_cast<BVHNode*>(take(2 * (size_t)n * sizeof(BVHNode)));
    uint32_t* info = reinterpret_cast<uint32_t*>(take(256));
    *d_out = g.out;
    *d_info = info;
    if (n == 0) return hipSuccess;
    // (a node the build failed to write would be a leaf of 2^32 - 1 spheres at sphere 2^32 - 1: brt_upload_scene's validation refuses it)
    hipError_t e = hipMemsetAsync(g.out, 0xff, (2 * (size_t)n - 1) * sizeof(BVHNode), stream);
    if (e != hipSuccess) return e;
    uint32_t* scale_key = top_n + kTopCounters;                      // behind the level counters; starts at the key of reach / 2 (zero = the max's identity)
    e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(scale_key), (int)sah_reach_key(reach), 1, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_sah_scale, dim3((n + 255u) / 256u), dim3(256), 0, stream, d_models, n, scale_key);
    hipLaunchKernelGGL(k_sah_prep, dim3((n + 255u) / 256u), dim3(256), 0, stream, d_models, n, scale_key, kbox, cen, g.st.idx[0], g.tasks,
                       g.n_tasks, top_list[0], top_n, info);
    if (n > kSubMax) {
        // the top of the tree level by level, the nodes of a level side by side (a block each); then one block for the rest
        const uint32_t levels = sah_top_levels(n);
        for (uint32_t l = 0; l < levels; l++) {
            const uint32_t most = (uint32_t)sah_top_list_tasks(n), at_level = l < 20u ? (1u << l) : most;
            hipLaunchKernelGGL(k_sah_top, dim3(at_level < most ? at_level : most), dim3(SB), 0, stream, g, top_list[l & 1u], top_n + l,
                               top_list[(l + 1u) & 1u], top_n + l + 1u, 0u);
        }
        hipLaunchKernelGGL(k_sah_top, dim3(1), dim3(SB), 0, stream, g, top_list[levels & 1u], top_n + levels, top_list[(levels + 1u) & 1u],
                           top_n + levels + 1u, 1u);
    }
    // a block per subtree; with fewer blocks than tasks a block takes several in turn
    const uint32_t max_tasks = n <= kSubMax ? 1u : (n / 2u + 1u);
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sah_sub), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       (int)kLocalStoreBytes);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL(k_sah_sub, dim3(max_tasks < 512u ? max_tasks : 512u), dim3(SB), kLocalStoreBytes, stream, g);
    return hipGetLastError();
}

}  // namespace brt
