// brt_bvh.hip -- GPU-side BVH build: PLOC on gfx950, in one 1024-thread workgroup for scenes of up to
// kPlocOneBlockMax spheres and as a chain of grid-wide kernels above that (end of this file).
//
// Replaces the per-frame CPU rebuild of the reference (extract.rs:315-332, "everything is
// currently copied to storage buffers every frame", README.md:17).  Same algorithm, arithmetic
// and numbering rule as the CPU builder (brt_ploc.h), so the output is byte-identical to
// brt_build_bvh and the tests compare the two with memcmp.
//
// One workgroup because a build is a chain of short dependent phases (sort stages, ~10-15 PLOC
// rounds of search / scan / merge); with all data L2-resident (10k spheres: 0.6 MB of scratch)
// a workgroup barrier per phase is far cheaper than a kernel boundary or a grid barrier per
// phase (MI355X_MICROARCH.md: 1.5-5 us each).  Work per phase is strided over the 1024 threads.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cstdlib>

#include "brt_kernels.h"
#include "brt_ploc.h"

namespace brt {

namespace {

constexpr int BVH_BLOCK = 1024;

__device__ __forceinline__ bool key_less(uint64_t ka, uint32_t ia, uint64_t kb, uint32_t ib) {
    return ka < kb || (ka == kb && ia < ib);
}

// inclusive scan of one u32 per thread over the block (Hillis-Steele in LDS); returns the
// exclusive prefix of this thread and the block total
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* lds, uint32_t& total) {
    const int t = threadIdx.x;
    lds[t] = v;
    __syncthreads();
    for (int off = 1; off < BVH_BLOCK; off <<= 1) {
        const uint32_t add = (t >= off) ? lds[t - off] : 0u;
        __syncthreads();
        lds[t] += add;
        __syncthreads();
    }
    const uint32_t incl = lds[t];
    total = lds[BVH_BLOCK - 1];
    __syncthreads();
    return incl - v;
}

}  // namespace

__global__ __launch_bounds__(BVH_BLOCK) void k_build_ploc(const Model* __restrict__ models, uint32_t n, PlocBox* box,
                                                          int32_t* left, int32_t* right, uint64_t* key, uint32_t* kidx,
                                                          uint32_t n_pow2, int32_t* cur, int32_t* next, int32_t* nn,
                                                          BVHNode* __restrict__ out, uint32_t* __restrict__ info) {
    __shared__ float red[6][BVH_BLOCK];
    __shared__ uint32_t scan_a[BVH_BLOCK], scan_b[BVH_BLOCK];
    __shared__ PlocBox s_scene;
    const uint32_t t = threadIdx.x;

    // 1. padded sphere boxes (Model::aabb) and the scene box
    float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (uint32_t i = t; i < n; i += BVH_BLOCK) {
        const PlocBox b = ploc_model_box(models[i].position, models[i].radius);
        box[i] = b;
        left[i] = -1;
        right[i] = -1;
        for (int k = 0; k < 3; k++) { mn[k] = ploc_min(mn[k], b.mn[k]); mx[k] = ploc_max(mx[k], b.mx[k]); }
    }
    for (int k = 0; k < 3; k++) { red[k][t] = mn[k]; red[3 + k][t] = mx[k]; }
    __syncthreads();
    for (int s = BVH_BLOCK / 2; s > 0; s >>= 1) {
        if ((int)t < s)
            for (int k = 0; k < 3; k++) {
                red[k][t] = ploc_min(red[k][t], red[k][t + s]);
                red[3 + k][t] = ploc_max(red[3 + k][t], red[3 + k][t + s]);
            }
        __syncthreads();
    }
    if (t == 0)
        for (int k = 0; k < 3; k++) { s_scene.mn[k] = red[k][0]; s_scene.mx[k] = red[3 + k][0]; }
    __syncthreads();

    // 2. Morton keys, padded to a power of two with +inf keys
    const PlocBox scene = s_scene;
    for (uint32_t i = t; i < n_pow2; i += BVH_BLOCK) {
        if (i < n) { key[i] = ploc_morton(box[i], scene); kidx[i] = i; }
        else { key[i] = ~0ull; kidx[i] = ~0u; }
    }
    __syncthreads();

    // 3. bitonic sort by (key, index)
    for (uint32_t k = 2; k <= n_pow2; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t i = t; i < n_pow2; i += BVH_BLOCK) {
                const uint32_t l = i ^ j;
                if (l > i) {
                    const uint64_t ka = key[i], kb = key[l];
                    const uint32_t ia = kidx[i], ib = kidx[l];
                    const bool up = (i & k) == 0;
                    if (key_less(kb, ib, ka, ia) == up) { key[i] = kb; kidx[i] = ib; key[l] = ka; kidx[l] = ia; }
                }
            }
            __syncthreads();
        }
    }
    for (uint32_t i = t; i < n; i += BVH_BLOCK) cur[i] = (int32_t)kidx[i];
    __syncthreads();

    // 4. PLOC rounds
    uint32_t m = n, created = n, rounds = 0;
    while (m > 1) {
        for (uint32_t i = t; i < m; i += BVH_BLOCK) {
            const PlocBox bi = box[cur[i]];
            float best = 0.0f;
            int bj = -1;
            const int lo = (int)i - PLOC_SEARCH < 0 ? 0 : (int)i - PLOC_SEARCH;
            const int hi = (int)i + PLOC_SEARCH > (int)m - 1 ? (int)m - 1 : (int)i + PLOC_SEARCH;
            for (int j = lo; j <= hi; j++) {
                if (j == (int)i) continue;
                const PlocBox bj_box = box[cur[j]];
                const float a = j < (int)i ? ploc_pair_cost(bj_box, bi) : ploc_pair_cost(bi, bj_box);
                if (ploc_better(a, (int)i, j, best, bj)) { best = a; bj = j; }
            }
            nn[i] = bj;
        }
        __syncthreads();
        // contiguous chunk per thread so that output order == input order
        const uint32_t chunk = (m + BVH_BLOCK - 1) / BVH_BLOCK;
        const uint32_t c0 = t * chunk, c1 = (c0 + chunk < m) ? c0 + chunk : m;
        uint32_t keep = 0, merge = 0;
        for (uint32_t i = c0; i < c1 && i < m; i++) {
            const int j = nn[i];
            const bool mutual = nn[j] == (int)i;
            if (!mutual) keep++;
            else if ((int)i < j) { keep++; merge++; }
        }
        uint32_t total_keep, total_merge;
        uint32_t pos = block_exclusive_scan(keep, scan_a, total_keep);
        uint32_t mpos = block_exclusive_scan(merge, scan_b, total_merge);
        for (uint32_t i = c0; i < c1 && i < m; i++) {
            const int j = nn[i];
            const bool mutual = nn[j] == (int)i;
            if (!mutual) {
                next[pos++] = cur[i];
            } else if ((int)i < j) {
                const uint32_t id = created + mpos++;
                box[id] = ploc_merge(box[cur[i]], box[cur[j]]);
                left[id] = cur[i];
                right[id] = cur[j];
                next[pos++] = (int32_t)id;
            }
        }
        __syncthreads();
        int32_t* tmp = cur; cur = next; next = tmp;
        m = total_keep;
        created += total_merge;
        rounds++;
        if (total_merge == 0u) break;   // block-uniform; cannot happen (brt_ploc.h), but a build must never spin
    }
    if (m > 1) {
        if (t == 0) { info[0] = 0u; info[1] = rounds; }
        return;
    }

    // 5. numbering rule of brt_ploc.h
    auto write = [&](uint32_t slot, int32_t id) {
        BVHNode o;
        o._pad0 = 0.0f; o._pad1[0] = o._pad1[1] = o._pad1[2] = 0u;
        for (int k = 0; k < 3; k++) { o.bounds_min[k] = box[id].mn[k]; o.bounds_max[k] = box[id].mx[k]; }
        if (left[id] < 0) { o.index = (uint32_t)id; o.model_count = 1u; }
        else { o.index = 1u + 2u * ((2u * n - 2u) - (uint32_t)id); o.model_count = 0u; }
        out[slot] = o;
    };
    if (t == 0) write(0u, (int32_t)(2u * n - 2u));
    for (uint32_t id = n + t; id < 2u * n - 1u; id += BVH_BLOCK) {
        const uint32_t r = (2u * n - 2u) - id;
        write(1u + 2u * r, left[id]);
        write(2u + 2u * r, right[id]);
    }
    if (t == 0) { info[0] = 2u * n - 1u; info[1] = rounds; }
}

// ---- grid version: scenes of more than kPlocOneBlockMax spheres ---------------------------------------------------
//
// The same build as k_build_ploc -- same boxes, same (key, index) order, same nearest-neighbour rule, same
// creation order of the merged clusters, hence the same bytes -- with every phase spread over the whole chip:
//   boxes + per-block scene boxes | scene box | Morton keys | radix sort (hipCUB, stable: ties keep index order)
//   then per PLOC round:  nearest neighbours | per-block keep/merge counts | emit (block prefix + in-block scan)
//                         | advance (one thread: m, created)
// and the numbering pass.  The number of rounds is data dependent (about 1.5 log2 n), so the host launches rounds in
// batches and reads the 16-byte state back between batches; a round that finds m <= 1 does nothing.
namespace {

constexpr uint32_t PB = 1024;          // threads per block = clusters per block in the round kernels

struct PlocState {
    uint32_t m;            // clusters in the current list
    uint32_t created;      // next temporary id
    uint32_t rounds;
    uint32_t stalled;      // a round merged nothing (cannot happen, brt_ploc.h; a build must never spin)
};

__global__ __launch_bounds__(PB) void k_ploc_boxes(const Model* __restrict__ models, uint32_t n, PlocBox* box, int32_t* left,
                                                   int32_t* right, PlocBox* block_box) {
    __shared__ float red[6][PB];
    const uint32_t t = threadIdx.x, i = blockIdx.x * PB + t;
    float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    if (i < n) {
        const PlocBox b = ploc_model_box(models[i].position, models[i].radius);
        box[i] = b;
        left[i] = -1;
        right[i] = -1;
        for (int k = 0; k < 3; k++) { mn[k] = ploc_min(mn[k], b.mn[k]); mx[k] = ploc_max(mx[k], b.mx[k]); }
    }
    for (int k = 0; k < 3; k++) { red[k][t] = mn[k]; red[3 + k][t] = mx[k]; }
    __syncthreads();
    for (int s = PB / 2; s > 0; s >>= 1) {
        if ((int)t < s)
            for (int k = 0; k < 3; k++) {
                red[k][t] = ploc_min(red[k][t], red[k][t + s]);
                red[3 + k][t] = ploc_max(red[3 + k][t], red[3 + k][t + s]);
            }
        __syncthreads();
    }
    if (t == 0) {
        PlocBox b;
        for (int k = 0; k < 3; k++) { b.mn[k] = red[k][0]; b.mx[k] = red[3 + k][0]; }
        block_box[blockIdx.x] = b;
    }
}

// scene box = NaN-ignoring min/max over the block boxes (order independent: brt_ploc.h), then state and keys' scene
__global__ __launch_bounds__(PB) void k_ploc_scene(const PlocBox* block_box, uint32_t n_blocks, uint32_t n, PlocBox* scene,
                                                   PlocState* st) {
    __shared__ float red[6][PB];
    const uint32_t t = threadIdx.x;
    float mn[3] = {3.4e38f, 3.4e38f, 3.4e38f}, mx[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
    for (uint32_t b = t; b < n_blocks; b += PB)
        for (int k = 0; k < 3; k++) { mn[k] = ploc_min(mn[k], block_box[b].mn[k]); mx[k] = ploc_max(mx[k], block_box[b].mx[k]); }
    for (int k = 0; k < 3; k++) { red[k][t] = mn[k]; red[3 + k][t] = mx[k]; }
    __syncthreads();
    for (int s = PB / 2; s > 0; s >>= 1) {
        if ((int)t < s)
            for (int k = 0; k < 3; k++) {
                red[k][t] = ploc_min(red[k][t], red[k][t + s]);
                red[3 + k][t] = ploc_max(red[3 + k][t], red[3 + k][t + s]);
            }
        __syncthreads();
    }
    if (t == 0) {
        for (int k = 0; k < 3; k++) { scene->mn[k] = red[k][0]; scene->mx[k] = red[3 + k][0]; }
        st->m = n; st->created = n; st->rounds = 0u; st->stalled = 0u;
    }
}

__global__ __launch_bounds__(PB) void k_ploc_keys(const PlocBox* box, const PlocBox* scene, uint32_t n, uint64_t* key, uint32_t* kidx) {
    const uint32_t i = blockIdx.x * PB + threadIdx.x;
    if (i < n) { key[i] = ploc_morton(box[i], *scene); kidx[i] = i; }
}

__global__ __launch_bounds__(PB) void k_ploc_cur(const uint32_t* kidx_sorted, uint32_t n, int32_t* cur) {
    const uint32_t i = blockIdx.x * PB + threadIdx.x;
    if (i < n) cur[i] = (int32_t)kidx_sorted[i];
}

__global__ __launch_bounds__(PB) void k_ploc_nn(const PlocState* st, const PlocBox* box, const int32_t* cur, int32_t* nn) {
    const uint32_t m = st->m, i = blockIdx.x * PB + threadIdx.x;
    if (m <= 1u || st->stalled || i >= m) return;
    const PlocBox bi = box[cur[i]];
    float best = 0.0f;
    int bj = -1;
    const int lo = (int)i - PLOC_SEARCH < 0 ? 0 : (int)i - PLOC_SEARCH;
    const int hi = (int)i + PLOC_SEARCH > (int)m - 1 ? (int)m - 1 : (int)i + PLOC_SEARCH;
    for (int j = lo; j <= hi; j++) {
        if (j == (int)i) continue;
        const PlocBox bj_box = box[cur[j]];
        const float a = j < (int)i ? ploc_pair_cost(bj_box, bi) : ploc_pair_cost(bi, bj_box);
        if (ploc_better(a, (int)i, j, best, bj)) { best = a; bj = j; }
    }
    nn[i] = bj;
}

// keep / merge flags of cluster i (keep: it, or the merged cluster in its place, is in the next list)
__device__ __forceinline__ void ploc_flags(const int32_t* nn, uint32_t i, uint32_t m, uint32_t& keep, uint32_t& merge) {
    keep = 0u; merge = 0u;
    if (i >= m) return;
    const int j = nn[i];
    const bool mutual = nn[j] == (int)i;
    if (!mutual) keep = 1u;
    else if ((int)i < j) { keep = 1u; merge = 1u; }
}

__global__ __launch_bounds__(PB) void k_ploc_count(const PlocState* st, const int32_t* nn, uint2* block_count) {
    __shared__ uint32_t sa[PB], sb[PB];
    const uint32_t m = st->m, t = threadIdx.x, i = blockIdx.x * PB + t;
    if (m <= 1u || st->stalled || blockIdx.x * PB >= m) return;
    uint32_t keep, merge, tk, tm;
    ploc_flags(nn, i, m, keep, merge);
    (void)block_exclusive_scan(keep, sa, tk);
    (void)block_exclusive_scan(merge, sb, tm);
    if (t == 0) block_count[blockIdx.x] = make_uint2(tk, tm);
}

__global__ __launch_bounds__(PB) void k_ploc_emit(const PlocState* st, const int32_t* nn, const uint2* block_count, PlocBox* box,
                                                  int32_t* left, int32_t* right, const int32_t* cur, int32_t* next, uint2* totals) {
    __shared__ uint32_t sa[PB], sb[PB];
    const uint32_t m = st->m, t = threadIdx.x, i = blockIdx.x * PB + t;
    if (m <= 1u || st->stalled || blockIdx.x * PB >= m) return;
    // prefix of the blocks before this one (at most 1024 blocks per 2^20 clusters: a strided sum + block reduction)
    const uint32_t n_blocks = (m + PB - 1u) / PB;
    uint32_t pk = 0u, pm = 0u;
    for (uint32_t b = t; b < blockIdx.x; b += PB) { pk += block_count[b].x; pm += block_count[b].y; }
    uint32_t base_keep, base_merge, tk, tm;
    (void)block_exclusive_scan(pk, sa, base_keep);
    (void)block_exclusive_scan(pm, sb, base_merge);
    uint32_t keep, merge;
    ploc_flags(nn, i, m, keep, merge);
    const uint32_t pos = base_keep + block_exclusive_scan(keep, sa, tk);
    const uint32_t mpos = base_merge + block_exclusive_scan(merge, sb, tm);
    if (keep) {
        if (!merge) {
            next[pos] = cur[i];
        } else {
            const int j = nn[i];
            const uint32_t id = st->created + mpos;
            box[id] = ploc_merge(box[cur[i]], box[cur[j]]);
            left[id] = cur[i];
            right[id] = cur[j];
            next[pos] = (int32_t)id;
        }
    }
    if (blockIdx.x == n_blocks - 1u && t == 0) *totals = make_uint2(base_keep + tk, base_merge + tm);
}

__global__ void k_ploc_advance(PlocState* st, const uint2* totals) {
    if (st->m <= 1u || st->stalled) return;
    const uint2 tt = *totals;
    if (tt.y == 0u) { st->stalled = 1u; return; }
    st->m = tt.x;
    st->created += tt.y;
    st->rounds++;
}

__global__ __launch_bounds__(PB) void k_ploc_write(const PlocState* st, uint32_t n, const PlocBox* box, const int32_t* left,
                                                   const int32_t* right, BVHNode* __restrict__ out, uint32_t* __restrict__ info) {
    const uint32_t g = blockIdx.x * PB + threadIdx.x;
    if (st->m != 1u || st->stalled) {
        if (g == 0) { info[0] = 0u; info[1] = st->rounds; }
        return;
    }
    auto write = [&](uint32_t slot, int32_t id) {
        BVHNode o;
        o._pad0 = 0.0f; o._pad1[0] = o._pad1[1] = o._pad1[2] = 0u;
        for (int k = 0; k < 3; k++) { o.bounds_min[k] = box[id].mn[k]; o.bounds_max[k] = box[id].mx[k]; }
        if (left[id] < 0) { o.index = (uint32_t)id; o.model_count = 1u; }
        else { o.index = 1u + 2u * ((2u * n - 2u) - (uint32_t)id); o.model_count = 0u; }
        out[slot] = o;
    };
    if (g == 0) { write(0u, (int32_t)(2u * n - 2u)); info[0] = 2u * n - 1u; info[1] = st->rounds; }
    const uint32_t id = n + g;
    if (id < 2u * n - 1u) {
        const uint32_t r = (2u * n - 2u) - id;
        write(1u + 2u * r, left[id]);
        write(2u + 2u * r, right[id]);
    }
}

size_t radix_temp_bytes(uint32_t n) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const uint64_t*)nullptr, (uint64_t*)nullptr, (const uint32_t*)nullptr,
                                             (uint32_t*)nullptr, (int)n, 0, 63);
    return bytes;
}

}  // namespace

size_t ploc_scratch_bytes(uint32_t n, uint32_t* n_pow2_out, uint32_t one_block_max) {
    uint32_t p = 1;
    while (p < n) p <<= 1;
    if (n_pow2_out) *n_pow2_out = p;
    const size_t nodes = 2 * (size_t)n;
    const bool grid = n > one_block_max;
    const size_t keys = grid ? 2 * (size_t)n : (size_t)p;   // grid version: hipCUB sorts out of place
    size_t b = 0;
    b += nodes * sizeof(PlocBox) + 256;          // box
    b += nodes * 4 * 2 + 512;                    // left, right
    b += keys * 8 + 256 + keys * 4 + 256;        // key, kidx
    b += (size_t)n * 4 * 3 + 768;                // cur, next, nn
    b += nodes * sizeof(BVHNode) + 256;          // out
    b += 256;                                    // info
    if (grid) {
        const size_t blocks = ((size_t)n + PB - 1) / PB;
        b += blocks * sizeof(PlocBox) + 256 + blocks * 8 + 256 + 256 + 256 + 256;   // block boxes, block counts, scene, state, totals
        b += radix_temp_bytes(n) + 256;
    }
    return b;
}

hipError_t launch_build_ploc(const Model* d_models, uint32_t n, char* d_scratch, BVHNode** d_out, uint32_t** d_info,
                             uint32_t one_block_max, hipStream_t stream) {
    uint32_t p2 = 1;
    (void)ploc_scratch_bytes(n, &p2, one_block_max);
    const bool grid = n > one_block_max;
    auto take = [&](size_t bytes) { char* r = d_scratch; d_scratch += (bytes + 255) & ~(size_t)255; return r; };
    const size_t nodes = 2 * (size_t)n;
    const size_t keys = grid ? 2 * (size_t)n : (size_t)p2;
    PlocBox* box = reinterpret_cast<PlocBox*>(take(nodes * sizeof(PlocBox)));
    int32_t* left = reinterpret_cast<int32_t*>(take(nodes * 4));
    int32_t* right = reinterpret_cast<int32_t*>(take(nodes * 4));
    uint64_t* key = reinterpret_cast<uint64_t*>(take(keys * 8));
    uint32_t* kidx = reinterpret_cast<uint32_t*>(take(keys * 4));
    int32_t* cur = reinterpret_cast<int32_t*>(take((size_t)n * 4));
    int32_t* next = reinterpret_cast<int32_t*>(take((size_t)n * 4));
    int32_t* nn = reinterpret_cast<int32_t*>(take((size_t)n * 4));
    BVHNode* out = reinterpret_cast<BVHNode*>(take(nodes * sizeof(BVHNode)));
    uint32_t* info = reinterpret_cast<uint32_t*>(take(256));
    *d_out = out;
    *d_info = info;
    if (!grid) {
        hipLaunchKernelGGL(k_build_ploc, dim3(1), dim3(BVH_BLOCK), 0, stream, d_models, n, box, left, right, key, kidx, p2, cur,
                           next, nn, out, info);
        return hipGetLastError();
    }
    const uint32_t blocks = (n + PB - 1u) / PB;
    PlocBox* block_box = reinterpret_cast<PlocBox*>(take((size_t)blocks * sizeof(PlocBox)));
    uint2* block_count = reinterpret_cast<uint2*>(take((size_t)blocks * 8));
    PlocBox* scene = reinterpret_cast<PlocBox*>(take(256));
    PlocState* st = reinterpret_cast<PlocState*>(take(256));
    uint2* totals = reinterpret_cast<uint2*>(take(256));
    size_t temp_bytes = radix_temp_bytes(n);
    void* temp = take(temp_bytes);
    hipLaunchKernelGGL(k_ploc_boxes, dim3(blocks), dim3(PB), 0, stream, d_models, n, box, left, right, block_box);
    hipLaunchKernelGGL(k_ploc_scene, dim3(1), dim3(PB), 0, stream, block_box, blocks, n, scene, st);
    hipLaunchKernelGGL(k_ploc_keys, dim3(blocks), dim3(PB), 0, stream, box, scene, n, key, kidx);
    hipError_t e = hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, key, key + n, kidx, kidx + n, (int)n, 0, 63, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_ploc_cur, dim3(blocks), dim3(PB), 0, stream, kidx + n, n, cur);
    // rounds in batches; between batches the host reads the state (the only synchronisation of the build)
    for (uint32_t batch = 0; batch < 4096u; batch++) {
        for (int r = 0; r < 8; r++) {
            hipLaunchKernelGGL(k_ploc_nn, dim3(blocks), dim3(PB), 0, stream, st, box, cur, nn);
            hipLaunchKernelGGL(k_ploc_count, dim3(blocks), dim3(PB), 0, stream, st, nn, block_count);
            hipLaunchKernelGGL(k_ploc_emit, dim3(blocks), dim3(PB), 0, stream, st, nn, block_count, box, left, right, cur, next, totals);
            hipLaunchKernelGGL(k_ploc_advance, dim3(1), dim3(1), 0, stream, st, totals);
            int32_t* tmp = cur; cur = next; next = tmp;
        }
        PlocState h{};
        e = hipMemcpyAsync(&h, st, sizeof h, hipMemcpyDeviceToHost, stream);
        if (e != hipSuccess) return e;
        e = hipStreamSynchronize(stream);
        if (e != hipSuccess) return e;
        if (h.m <= 1u || h.stalled) break;
    }
    hipLaunchKernelGGL(k_ploc_write, dim3(blocks), dim3(PB), 0, stream, st, n, box, left, right, out, info);
    return hipGetLastError();
}

}  // namespace brt
