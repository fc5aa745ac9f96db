// brt_kernels.h -- host-callable launchers of the HIP kernels (brt_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "brt_layout.h"

#define BRT_BLOCK 1024

#include "../../include/bevyray_amd.h"  // BRT_DBG_* op codes

namespace brt {

struct TraceLaunch {
    DeviceSceneView scene;
    FrameParams frame;
    uint32_t* queue_counter;            // zeroed by the caller on `stream` before the launch
    float* out_tile;
    const float* raster_rgba;
    const float* raster_depth;
    unsigned long long* counters;       // 5 x u64, zeroed by the caller
    int scene_mode;                     // SceneMode (brt_layout.h)
    bool counters_on;
    uint32_t srv;                       // sampler stage (brt_trace.h, SRV): this many of a workgroup's waves serve the rejection sampler of the others (0: none)
    int lean;                           // LEAN instantiation (brt_trace.h): 0 general; 1 level 3 + no tile-cost measurement; 2 + no critical tile possible
    uint32_t grid, block;
    size_t lds_bytes;
    hipStream_t stream;
};

// rows: with the scratch of the row-mode walk of thin waves (SCENE_LDS only; FrameParams::rows_on)
size_t trace_lds_bytes(const DeviceSceneView& sv, int scene_mode, uint32_t block, uint32_t pool_cap, uint32_t hist_words = 0, bool rows = false);
constexpr uint32_t POOL_RECORD_BYTES = 96;   // one path in the drain pool (k_trace_persistent)
// sampler stage (k_trace_persistent<.., SRV>): a 32-byte mailbox entry per lane of every trace wave + 16 control words + a door per wave
constexpr uint32_t SRV_WAVES = 2;
constexpr uint32_t srv_lds_bytes(uint32_t block) { return (block / 64u - SRV_WAVES) * 64u * 32u + 64u + (block / 64u) * 4u; }
constexpr uint32_t WGQ_BYTES = 64;           // a workgroup's share of the pixel queue: 8 control words + 8 tile ids
hipError_t launch_trace_persistent(const TraceLaunch& tl);
hipError_t launch_trace_simple(const TraceLaunch& tl);
hipError_t launch_passthrough(const FrameParams& fp, float* out_tile, const float* raster_rgba, hipStream_t stream);
// out_format: BRT_FLAG_OUT_* (include/bevyray_amd.h): the frame is written in the colour target's own format
// part_of_strip: the context's strip table on this device (frame strip -> part), or null: strip s belongs to part s % n_parts
hipError_t launch_deinterleave(const float* tiles, void* frame, uint32_t width, uint32_t height, uint32_t n_parts,
                               uint32_t tile_rows, uint32_t out_format, hipStream_t stream, const uint32_t* part_of_strip = nullptr);
// the strips of parts 1 .. n_parts-1 of a full-frame input (floats_per_pixel 4: colour, 1: depth), each part densely in its tile layout:
// packed[(part - 1) * tile_rows * width ...]
hipError_t launch_pack_strips(const float* frame, float* packed, uint32_t width, uint32_t height, uint32_t n_parts, uint32_t tile_rows,
                              uint32_t floats_per_pixel, hipStream_t stream);
hipError_t launch_debug_eval(uint32_t op, const float* in, float* out, uint32_t n, hipStream_t stream);

// Dispatch order on the GPU (brt_order.hip): d_meta[0] = critical tiles at the front of the order, d_meta[1] = longest pixel,
// d_meta[2] = non-sky tiles, d_meta[3] = tiles handed out as two half-sample jobs (d_order then has n_tiles + d_meta[3] entries: room for
// n_tiles + split_tail)
size_t order_scratch_bytes(uint32_t n_tiles);
// tiles_x / dilate_x / dilate_y: rank every tile by its neighbourhood of that radius (0, 0: by itself), brt_order.hip
hipError_t launch_build_order(const uint32_t* d_ray_sum, const uint32_t* d_longest, uint32_t n_tiles, uint64_t sky_cost,
                              uint64_t grid_lanes, uint32_t tiles_x, uint32_t dilate_x, uint32_t dilate_y, uint32_t split_tail,
                              uint32_t* d_order, uint32_t* d_meta, char* d_scratch, hipStream_t stream);

// GPU PLOC builder (brt_bvh.hip): scratch size for n models, and the launch; *d_out / *d_info
// point into the scratch (nodes in the reference's 48-byte format; info[0] = node count,
// info[1] = PLOC rounds)
constexpr uint32_t kPlocOneBlockMax = 6000;    // up to here the one-workgroup build (no kernel boundaries) is faster; above: the grid version
// one_block_max: kPlocOneBlockMax, or the knob BRT_PLOC_ONE_BLOCK_MAX (tests force the grid version on small scenes with 0)
size_t ploc_scratch_bytes(uint32_t n, uint32_t* n_pow2_out, uint32_t one_block_max);
hipError_t launch_build_ploc(const Model* d_models, uint32_t n, char* d_scratch, BVHNode** d_out, uint32_t** d_info,
                             uint32_t one_block_max, hipStream_t stream);

// GPU binned-SAH builder (brt_sah.hip; the rule: brt_sah.h; CPU twin: brt_host.cpp build_bvh_sah): *d_out / *d_info point into
// the scratch (info[0] = node count)
size_t sah_scratch_bytes(uint32_t n);
hipError_t launch_build_sah(const Model* d_models, uint32_t n, float reach, char* d_scratch, BVHNode** d_out, uint32_t** d_info, hipStream_t stream);

}  // namespace brt
