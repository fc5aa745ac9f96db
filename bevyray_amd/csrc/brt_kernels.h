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
    bool lds_scene;
    bool counters_on;
    uint32_t grid, block;
    size_t lds_bytes;
    hipStream_t stream;
};

size_t trace_lds_bytes(const DeviceSceneView& sv, bool lds_scene, uint32_t block);
hipError_t launch_trace_persistent(const TraceLaunch& tl);
hipError_t launch_trace_simple(const TraceLaunch& tl);
hipError_t launch_passthrough(const FrameParams& fp, float* out_tile, const float* raster_rgba, hipStream_t stream);
hipError_t launch_deinterleave(const float* tiles, float* frame, uint32_t width, uint32_t height, uint32_t n_parts,
                               uint32_t tile_rows, hipStream_t stream);
hipError_t launch_debug_eval(uint32_t op, const float* in, float* out, uint32_t n, hipStream_t stream);

}  // namespace brt
