// brt_order.hip -- the dispatch order of the 8x8 tiles, built on the GPU from one frame's per-tile ray counts.
//
// Same rule as build_tile_order (brt_host.cpp), which stays the host-side statement of it and the reference in the
// tests: tiles that needed more than a sky tile's rays first, by their longest pixel (longest first, then by index),
// then the "sky" tiles in raster order; the leading tiles whose longest pixel is at least half a lane's share of the
// frame (and half the frame's longest pixel) are CRITICAL.  Building it here keeps the measuring frames and the
// first-frame pre-pass free of a device -> host -> device round trip (1.7 ms of host time per build at 1080p: copy the
// counts back, std::sort, copy the order up, two stream synchronisations).
// Half-sample jobs (split_tail; build_tile_order has the why): the last min(split_tail, non-sky tiles) non-sky tiles appear twice,
// order = [non-sky ... | first halves | second halves | sky], n_tiles + n_split entries; meta[2] = non-sky tiles, meta[3] = n_split.
// One block prepares the 64-bit keys and the frame totals (at most ~130 000 tiles at 4K), hipCUB sorts, a grid kernel
// writes the order.  Default settings only (sorted, critical tiles on, no lane queue); anything else takes the host path.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include "brt_kernels.h"

namespace brt {

namespace {

constexpr int OB = 1024;

// DILATED (rx, ry > 0; round 4): the costs were measured with another camera than the frame that will use the order -- a view that
// moves re-measures every frame, so the order is one frame old and what was expensive has moved a few tiles on.  A tile is then
// ranked by the longest pixel of its (2 rx + 1) x (2 ry + 1) neighbourhood, and is "sky" only if that whole neighbourhood was:
// wherever inside the radius the expensive pixels went, their tile is still handed out early.  (Measured on the headline frame
// with the camera orbiting 0.5 degrees per frame: 13.8 ms with the undilated one-frame-old order, the static view 10.8.)
__device__ __forceinline__ void tile_cost_at(const uint32_t* __restrict__ ray_sum, const uint32_t* __restrict__ longest, uint32_t n_tiles,
                                             uint32_t tiles_x, uint32_t rx, uint32_t ry, unsigned long long sky_cost, uint32_t i,
                                             uint32_t& lp, bool& sky) {
    lp = longest[i];
    sky = (unsigned long long)ray_sum[i] <= sky_cost;
    if ((rx | ry) == 0u || tiles_x == 0u) return;
    const uint32_t tiles_y = n_tiles / tiles_x, ty = i / tiles_x, tx = i - ty * tiles_x;
    const uint32_t x0 = tx > rx ? tx - rx : 0u, x1 = tx + rx < tiles_x ? tx + rx : tiles_x - 1u;
    const uint32_t y0 = ty > ry ? ty - ry : 0u, y1 = ty + ry < tiles_y ? ty + ry : tiles_y - 1u;
    for (uint32_t y = y0; y <= y1; y++)
        for (uint32_t x = x0; x <= x1; x++) {
            const uint32_t j = y * tiles_x + x;
            const uint32_t l = longest[j];
            lp = l > lp ? l : lp;
            sky = sky && (unsigned long long)ray_sum[j] <= sky_cost;
        }
}

__global__ __launch_bounds__(OB) void k_order_prepare(const uint32_t* __restrict__ ray_sum, const uint32_t* __restrict__ longest,
                                                      uint32_t n_tiles, unsigned long long sky_cost, unsigned long long grid_lanes,
                                                      uint32_t tiles_x, uint32_t rx, uint32_t ry, uint32_t split_tail,
                                                      unsigned long long* __restrict__ keys, uint32_t* __restrict__ meta) {
    __shared__ unsigned long long s_sum[OB];
    __shared__ uint32_t s_max[OB], s_cnt[OB], s_ns[OB];
    const uint32_t t = threadIdx.x;
    unsigned long long sum = 0;
    uint32_t mx = 0, nonsky = 0;
    for (uint32_t i = t; i < n_tiles; i += OB) {
        const uint32_t rs = ray_sum[i];
        uint32_t lp;
        bool sky;
        tile_cost_at(ray_sum, longest, n_tiles, tiles_x, rx, ry, sky_cost, i, lp, sky);
        sum += rs;
        mx = longest[i] > mx ? longest[i] : mx;
        keys[i] = ((unsigned long long)(sky ? 0xffffffffu : ~lp) << 32) | i;
        nonsky += sky ? 0u : 1u;
    }
    s_sum[t] = sum; s_max[t] = mx; s_ns[t] = nonsky;
    __syncthreads();
    for (int s = OB / 2; s > 0; s >>= 1) {
        if ((int)t < s) { s_sum[t] += s_sum[t + s]; s_max[t] = s_max[t + s] > s_max[t] ? s_max[t + s] : s_max[t]; s_ns[t] += s_ns[t + s]; }
        __syncthreads();
    }
    const unsigned long long total = s_sum[0];
    const uint32_t longest_pixel = s_max[0];
    const unsigned long long per_lane = grid_lanes ? total / grid_lanes : 0ull;
    const unsigned long long half_lp = longest_pixel / 2u;
    const unsigned long long thr = per_lane / 2 > half_lp ? per_lane / 2 : half_lp;
    const bool any_critical = grid_lanes != 0ull && (unsigned long long)longest_pixel >= per_lane / 2;
    uint32_t cnt = 0;
    if (any_critical)
        for (uint32_t i = t; i < n_tiles; i += OB) {
            uint32_t lp;
            bool sky;
            tile_cost_at(ray_sum, longest, n_tiles, tiles_x, rx, ry, sky_cost, i, lp, sky);
            if (!sky && (unsigned long long)lp >= thr) cnt++;
        }
    s_cnt[t] = cnt;
    __syncthreads();
    for (int s = OB / 2; s > 0; s >>= 1) {
        if ((int)t < s) s_cnt[t] += s_cnt[t + s];
        __syncthreads();
    }
    if (t == 0) {
        meta[0] = s_cnt[0];
        meta[1] = longest_pixel;
        meta[2] = s_ns[0];
        meta[3] = split_tail < s_ns[0] - s_cnt[0] ? split_tail : s_ns[0] - s_cnt[0];      // (never a critical tile: build_tile_order)
    }
}

__global__ void k_order_emit(const unsigned long long* __restrict__ keys_sorted, uint32_t n_tiles, const uint32_t* __restrict__ meta,
                             uint32_t* __restrict__ order) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tiles) return;
    const uint32_t n_nonsky = meta[2], n_split = meta[3];
    const uint32_t tile = (uint32_t)(keys_sorted[i] & 0xffffffffull);
    if (i < n_nonsky) {
        order[i] = tile;
        if (i >= n_nonsky - n_split) order[i + n_split] = tile;      // its second half, behind all the first halves
    } else {
        order[i + n_split] = tile;                                   // sky tiles: behind both
    }
}

size_t order_temp_bytes(uint32_t n_tiles) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, (const unsigned long long*)nullptr, (unsigned long long*)nullptr, (int)n_tiles, 0, 64);
    return bytes;
}

}  // namespace

size_t order_scratch_bytes(uint32_t n_tiles) { return 2 * ((size_t)n_tiles * 8 + 256) + order_temp_bytes(n_tiles) + 256; }

hipError_t launch_build_order(const uint32_t* d_ray_sum, const uint32_t* d_longest, uint32_t n_tiles, uint64_t sky_cost,
                              uint64_t grid_lanes, uint32_t tiles_x, uint32_t dilate_x, uint32_t dilate_y, uint32_t split_tail,
                              uint32_t* d_order, uint32_t* d_meta, char* d_scratch, hipStream_t stream) {
    if (n_tiles == 0) return hipSuccess;
    auto take = [&](size_t bytes) { char* r = d_scratch; d_scratch += (bytes + 255) & ~(size_t)255; return r; };
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(take((size_t)n_tiles * 8));
    unsigned long long* sorted = reinterpret_cast<unsigned long long*>(take((size_t)n_tiles * 8));
    size_t temp_bytes = order_temp_bytes(n_tiles);
    void* temp = take(temp_bytes);
    hipLaunchKernelGGL(k_order_prepare, dim3(1), dim3(OB), 0, stream, d_ray_sum, d_longest, n_tiles, (unsigned long long)sky_cost,
                       (unsigned long long)grid_lanes, tiles_x, dilate_x, dilate_y, split_tail, keys, d_meta);
    hipError_t e = hipcub::DeviceRadixSort::SortKeys(temp, temp_bytes, keys, sorted, (int)n_tiles, 0, 64, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_order_emit, dim3((n_tiles + 255u) / 256u), dim3(256), 0, stream, sorted, n_tiles, d_meta, d_order);
    return hipGetLastError();
}

}  // namespace brt
