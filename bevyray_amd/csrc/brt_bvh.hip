// brt_bvh.hip -- GPU-side BVH build: PLOC in one 1024-thread workgroup (gfx950).
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

size_t ploc_scratch_bytes(uint32_t n, uint32_t* n_pow2_out) {
    uint32_t p = 1;
    while (p < n) p <<= 1;
    if (n_pow2_out) *n_pow2_out = p;
    const size_t nodes = 2 * (size_t)n;
    size_t b = 0;
    b += nodes * sizeof(PlocBox) + 256;          // box
    b += nodes * 4 * 2 + 512;                    // left, right
    b += (size_t)p * 8 + 256 + (size_t)p * 4 + 256;   // key, kidx
    b += (size_t)n * 4 * 3 + 768;                // cur, next, nn
    b += nodes * sizeof(BVHNode) + 256;          // out
    b += 256;                                    // info
    return b;
}

hipError_t launch_build_ploc(const Model* d_models, uint32_t n, char* d_scratch, BVHNode** d_out, uint32_t** d_info,
                             hipStream_t stream) {
    uint32_t p2 = 1;
    (void)ploc_scratch_bytes(n, &p2);
    auto take = [&](size_t bytes) { char* r = d_scratch; d_scratch += (bytes + 255) & ~(size_t)255; return r; };
    const size_t nodes = 2 * (size_t)n;
    PlocBox* box = reinterpret_cast<PlocBox*>(take(nodes * sizeof(PlocBox)));
    int32_t* left = reinterpret_cast<int32_t*>(take(nodes * 4));
    int32_t* right = reinterpret_cast<int32_t*>(take(nodes * 4));
    uint64_t* key = reinterpret_cast<uint64_t*>(take((size_t)p2 * 8));
    uint32_t* kidx = reinterpret_cast<uint32_t*>(take((size_t)p2 * 4));
    int32_t* cur = reinterpret_cast<int32_t*>(take((size_t)n * 4));
    int32_t* next = reinterpret_cast<int32_t*>(take((size_t)n * 4));
    int32_t* nn = reinterpret_cast<int32_t*>(take((size_t)n * 4));
    BVHNode* out = reinterpret_cast<BVHNode*>(take(nodes * sizeof(BVHNode)));
    uint32_t* info = reinterpret_cast<uint32_t*>(take(256));
    *d_out = out;
    *d_info = info;
    hipLaunchKernelGGL(k_build_ploc, dim3(1), dim3(BVH_BLOCK), 0, stream, d_models, n, box, left, right, key, kidx, p2, cur,
                       next, nn, out, info);
    return hipGetLastError();
}

}  // namespace brt
