// brt_kernels.hip -- HIP kernels of the path-tracing render node, written for gfx950
// (MI355X: 256 CUs, wave64, 160 KB LDS per CU).
//
// k_trace_persistent  the production kernel: persistent threads, one path per lane.
//     * One 1024-thread workgroup per CU (16 waves, 4 per SIMD) stays resident for the whole
//       frame.  When the encoded scene fits, the workgroup first copies pair records, spheres
//       and material ids into LDS (cover scene: 66 KB; the 32-byte materials stay in global
//       memory, one read per hit); the per-lane traversal stacks also live in LDS as
//       [entry][lane] arrays of 16-bit descriptors.
//     * Each lane owns one pixel at a time and walks that pixel's samples and bounces as a
//       flat state machine, one ray segment per outer iteration ("round"), because the
//       reference threads ONE RNG state through all samples of a pixel
//       (raytrace.wgsl:161-163): the samples of a pixel are sequentially dependent, the
//       pixels are not.
//     * A wave that has nothing left takes the next 8x8 tile of the tile queue (one atomicAdd per
//       tile): 64 neighbouring pixels, coherent primary rays.  The ORDER of the tiles comes from the
//       previous frame's ray counts (brt_host.cpp build_tile_order).  Optionally the front of the
//       order is handed out pixel by pixel to single free lanes instead (the lane queue: __ballot of
//       the free lanes, mbcnt ranks, slots taken from a per-workgroup batch in LDS).
//     * A wave that has thinned to `drain_donate` live paths hands them to the other waves of its
//       workgroup through an LDS pool ("drain pool" below) and takes its next tile; waves that hold
//       one of the frame's longest pixel chains run at raised priority and take nothing new.
//     * No ray state ever goes to HBM; the only HBM traffic is the scene load per workgroup
//       and one 16-byte store per pixel.
//     * Bound by instruction issue under divergence, not by memory: the
//       hot bodies are therefore straight-line (selects instead of nested branches).
// k_trace_simple      bring-up kernel: one thread per pixel, scene in global memory, private
//                     stack.  Kept as an independent second implementation for debugging.
// k_deinterleave      root side of the multi-GPU gather.
// k_debug_eval        evaluates single device functions for per-function parity tests.
#include <hip/hip_runtime.h>

#include <hip/hip_fp16.h>

#include "brt_srgb_table.h"
#include "brt_trace.h"

namespace brt {

// ---- bring-up kernel ---------------------------------------------------------------------------

template <bool D16, bool COUNTERS>
__global__ __launch_bounds__(256) void k_trace_simple(DeviceSceneView sv, FrameParams fp, float* __restrict__ out_tile,
                                                      const float* __restrict__ raster_rgba,
                                                      const float* __restrict__ raster_depth,
                                                      unsigned long long* __restrict__ counters) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    ScenePtrs sc;
    sc.pairs = reinterpret_cast<const char*>(sv.pairs);
    sc.pairs_far = sc.pairs;
    sc.near_bytes = 0u;
    sc.near_base = 0u;
    sc.sph_base = 0u;
    sc.rows_scratch = 0u;
    sc.hits = nullptr;
    sc.minmax_select = false;
    sc.boxes_ordered = sv.boxes_ordered != 0u;
    sc.spheres = reinterpret_cast<const float4*>(sv.spheres);
    sc.sphere_material = sv.sphere_material;
    sc.materials = reinterpret_cast<const float4*>(sv.materials);
    sc.sphere_mats = reinterpret_cast<const float4*>(sv.sphere_mats);
    sc.leaf_table = reinterpret_cast<const uint2*>(sv.leaf_table);
    uint32_t n_rays = 0;
    HitCounters hc = {};
    if (q < fp.queue_size) {
        const PixelCoord c = slot_to_pixel(fp, q, slot_tile(fp, q >> 6));
        if (c.inside) {
            PixelState ps;
            pixel_begin(fp, c, ps);
            uint32_t stack[34];   // DONE sentinel + 32 entries + one spare
            for (uint32_t s = 0; s < fp.sample_count; s++) {          // raytrace.wgsl:161
                f3 d = camera_ray_dir(fp, ps.ndc0x, ps.ndc0y, ps.rng);
                f3 o = mk3(fp.cam_pos[0], fp.cam_pos[1], fp.cam_pos[2]);
                f3 tput = mk3(1.0f, 1.0f, 1.0f);
                float first_depth = kInf;
                uint32_t bounce = 0;
                f3 color;
                for (;;) {
                    float t;
                    uint32_t idx;
                    raycast<1, COUNTERS, D16, false>(sc, sv.root_desc, stack, o, d, t, idx, hc);
                    n_rays++;
                    if (shade_segment<COUNTERS>(sc, fp, o, d, tput, bounce, first_depth, t, idx, ps.rng, color, hc)) break;
                }
                ps.sum = ps.sum + color;
                ps.dsum = ps.dsum + (first_depth == kInf ? fp.fallback_far : first_depth);
            }
            pixel_finish(fp, ps, out_tile, raster_rgba, raster_depth);
        }
    }
    const uint32_t lane = lane_id();
    const uint32_t r = wave_sum(n_rays);
    if (lane == 0) atomicAdd(&counters[0], (unsigned long long)r);
    if (COUNTERS) {
        const uint32_t a = wave_sum(hc.node_pops), b = wave_sum(hc.interior), c = wave_sum(hc.sphere_tests),
                       h = wave_sum(hc.hits);
        if (lane == 0) {
            atomicAdd(&counters[1], (unsigned long long)a);
            atomicAdd(&counters[2], (unsigned long long)b);
            atomicAdd(&counters[3], (unsigned long long)c);
            atomicAdd(&counters[4], (unsigned long long)h);
        }
    }
}

// ---- level 0: passthrough of the raster colour (raytrace.wgsl:97-99) -----------------------------

__global__ void k_passthrough(FrameParams fp, float4* __restrict__ out_tile, const float4* __restrict__ raster_rgba) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= fp.queue_size) return;
    const PixelCoord c = slot_to_pixel(fp, q, slot_tile(fp, q >> 6));
    if (!c.inside) return;
    out_tile[c.local_row * fp.width + c.px] =
        raster_rgba ? raster_rgba[fp.raster_dense ? c.local_row * fp.width + c.px : c.py * fp.width + c.px] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// ---- gather root: tiles of all parts -> frame -----------------------------------------------------

// The frame is written in the colour target's own format (reference: the pass renders into post_process.destination, whose format is
// TextureFormat::bevy_default() -- 8-bit sRGB, or Rgba16Float under HDR, pipeline.rs:311-315).  The conversions, exactly:
//   RGBA8 sRGB   colour: v = round(255 * OETF(clamp(c, 0, 1))) as the number of thresholds <= c (brt_srgb_table.h: exact for every f32, a
//                NaN encodes as 0 like the hardware's clamp); alpha: the linear rule below
//   RGBA8        v = round-half-even(255 * clamp(c, 0, 1)), the product exact in f64
//   RGBA16F      f32 -> f16, round to nearest even (v_cvt_f16_f32; overflow to infinity, denormals kept)
BRT_DEV uint32_t encode_srgb8(float c) {
    uint32_t n = 0;                                       // thresholds <= c so far: binary search over the 255 of them
#pragma unroll
    for (uint32_t step = 128u; step != 0u; step >>= 1)
        if (n + step <= 255u && c >= kSrgbThreshold[n + step - 1u]) n += step;
    return n;
}
BRT_DEV uint32_t encode_unorm8(float c) {
    const double x = c > 0.0f ? (c < 1.0f ? (double)c : 1.0) : 0.0;      // (a NaN fails the first test: 0)
    return (uint32_t)__double2int_rn(x * 255.0);
}
template <uint32_t FMT> struct OutPixel;
template <> struct OutPixel<BRT_FLAG_OUT_RGBA32F> {
    typedef float4 type;
    static BRT_DEV float4 make(float4 v) { return v; }
};
template <> struct OutPixel<BRT_FLAG_OUT_RGBA8_UNORM_SRGB> {
    typedef uint32_t type;
    static BRT_DEV uint32_t make(float4 v) { return encode_srgb8(v.x) | (encode_srgb8(v.y) << 8) | (encode_srgb8(v.z) << 16) | (encode_unorm8(v.w) << 24); }
};
template <> struct OutPixel<BRT_FLAG_OUT_RGBA8_UNORM> {
    typedef uint32_t type;
    static BRT_DEV uint32_t make(float4 v) { return encode_unorm8(v.x) | (encode_unorm8(v.y) << 8) | (encode_unorm8(v.z) << 16) | (encode_unorm8(v.w) << 24); }
};
template <> struct OutPixel<BRT_FLAG_OUT_RGBA16F> {
    typedef uint2 type;
    static BRT_DEV uint2 make(float4 v) {
        const uint32_t x = __half_as_ushort(__float2half_rn(v.x)), y = __half_as_ushort(__float2half_rn(v.y));
        const uint32_t z = __half_as_ushort(__float2half_rn(v.z)), w = __half_as_ushort(__float2half_rn(v.w));
        return make_uint2(x | (y << 16), z | (w << 16));
    }
};

template <uint32_t FMT>
__global__ void k_deinterleave(const float4* __restrict__ tiles, typename OutPixel<FMT>::type* __restrict__ frame, uint32_t width,
                               uint32_t height, uint32_t n_parts, uint32_t tile_rows, const uint32_t* __restrict__ part_of_strip) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t y = blockIdx.y;
    if (x >= width || y >= height) return;
    const uint32_t strip = y / 8u, r = y - strip * 8u;
    // (a strip table -- brt_set_strip_table -- permutes the parts inside every group of n_parts strips: the local strip is the group)
    const uint32_t part = part_of_strip ? part_of_strip[strip] : strip % n_parts, k = strip / n_parts;
    frame[(size_t)y * width + x] = OutPixel<FMT>::make(tiles[((size_t)part * tile_rows + (k * 8u + r)) * width + x]);
}

// ---- first device of an N-device context: the strips of parts 1 .. N-1 of a full-frame raster input, each part's densely (the layout
// of its tile buffer) -- what is then sent to that part's device instead of the whole frame --------------------------------------------

template <typename T>
__global__ void k_pack_strips(const T* __restrict__ frame, T* __restrict__ packed, uint32_t width, uint32_t height, uint32_t n_parts,
                              uint32_t tile_rows) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t lr = blockIdx.y, part = blockIdx.z + 1u;
    const uint32_t y = ((lr / 8u) * n_parts + part) * 8u + (lr & 7u);
    if (x >= width || y >= height) return;
    packed[((size_t)(part - 1u) * tile_rows + lr) * width + x] = frame[(size_t)y * width + x];
}

// ---- per-function probes ----------------------------------------------------------------------------

__global__ void k_debug_eval(uint32_t op, const float* __restrict__ in, float* __restrict__ out, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* a = in + (size_t)i * 16;
    float* r = out + (size_t)i * 8;
    for (int k = 0; k < 8; k++) r[k] = 0.0f;
    switch (op) {
        case BRT_DBG_MINMAX: r[0] = min_f(a[0], a[1]); r[1] = max_f(a[0], a[1]); break;
        case BRT_DBG_SQRT_DIV: r[0] = __builtin_sqrtf(a[0]); r[1] = a[0] / a[1]; break;
        case BRT_DBG_RNG: {
            uint32_t s = __float_as_uint(a[0]);
            r[0] = rng_float(s);
            r[1] = __uint_as_float(s);
            const f3 p = rng_unit_ball(s);
            r[2] = p.x; r[3] = p.y; r[4] = p.z; r[5] = __uint_as_float(s);
            break;
        }
        case BRT_DBG_SLAB: {  // o, d, bmin, bmax, closest -> pushed?
            const f3 o = mk3(a[0], a[1], a[2]), d = mk3(a[3], a[4], a[5]);
            const f3 inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
            r[0] = slab_push(o, inv, mk3(a[6], a[7], a[8]), mk3(a[9], a[10], a[11]), a[12]) ? 1.0f : 0.0f;
            break;
        }
        case BRT_DBG_SPHERE: {  // o, d, center, radius -> accepted t (or INF)
            const f3 o = mk3(a[0], a[1], a[2]), d = mk3(a[3], a[4], a[5]);
            float closest = kInf;
            uint32_t idx = 0xffffffffu;
            sphere_test(o, d, dot3(d, d), make_float4(a[6], a[7], a[8], a[9] * a[9]), 0u, closest, idx);
            r[0] = closest;
            break;
        }
        case BRT_DBG_SEED: {  // seed, px, py, W, H (as floats holding integers)
            FrameParams fp;
            fp.seed_scaled = a[0] * 10000.0f;
            const float uvx = (a[1] + 0.5f) / a[3], uvy = (a[2] + 0.5f) / a[4];
            r[0] = __uint_as_float(pixel_seed(fp, uvx, uvy));
            break;
        }
        case BRT_DBG_DIV: {   // n, d -> n / d, div_plain(n, rcp_refined(d)), 1 / d, recip_plain(d)  (the short forms UNGUARDED)
            r[0] = a[0] / a[1];
            r[1] = div_plain(a[0], rcp_refined(a[1]));
            r[2] = 1.0f / a[1];
            r[3] = recip_plain(a[1]);
            break;
        }
        case BRT_DBG_DIV_SWEEP: {   // seed (bits), count (float) -> mismatches of both short forms against `/` over `count`
            // pseudo-random pairs of the plain range: every exponent -40 .. 40, random mantissas and signs
            uint32_t st = __float_as_uint(a[0]) ^ (i * 0x9E3779B9u);
            const uint32_t count = (uint32_t)a[1];
            uint32_t bad = 0, bad_n = 0, bad_d = 0;
            for (uint32_t k = 0; k < count; k++) {
                st = rng_next(st + k);
                const uint32_t en = 127u - 40u + (st % 81u);
                const uint32_t mn = rng_next(st ^ 0x12345u);
                st = rng_next(st);
                const uint32_t ed = 127u - 40u + (st % 81u);
                const uint32_t md = rng_next(st ^ 0xabcdeu);
                float n = __uint_as_float((mn & 0x807fffffu) | (en << 23));
                float d = __uint_as_float((md & 0x807fffffu) | (ed << 23));
                if (en == 167u) n = __uint_as_float((mn & 0x80000000u) | (en << 23));   // 2^40 itself is the last plain value
                if (ed == 167u) d = __uint_as_float((md & 0x80000000u) | (ed << 23));
                const bool ok = __float_as_uint(div_plain(n, rcp_refined(d))) == __float_as_uint(n / d) &&
                                __float_as_uint(recip_plain(d)) == __float_as_uint(1.0f / d);
                if (!ok) { if (!bad) { bad_n = __float_as_uint(n); bad_d = __float_as_uint(d); } bad++; }
            }
            r[0] = (float)bad; r[1] = __uint_as_float(bad_n); r[2] = __uint_as_float(bad_d);
            break;
        }
        case BRT_DBG_SQRT_SWEEP: {   // first bits, count -> mismatches of sqrt_plain against __builtin_sqrtf over `count` consecutive floats
            const uint32_t first = __float_as_uint(a[0]) + i * (uint32_t)a[1];
            const uint32_t count = (uint32_t)a[1];
            uint32_t bad = 0, bad_x = 0;
            for (uint32_t k = 0; k < count; k++) {
                const float x = __uint_as_float(first + k);
                if (!(x >= 0x1p-80f && x <= 0x1p80f)) continue;
                if (__float_as_uint(sqrt_plain(x)) != __float_as_uint(__builtin_sqrtf(x))) { if (!bad) bad_x = first + k; bad++; }
            }
            if (i == 0) {   // and +-0, which sqrt3 (brt_device.h) also sends through the short form
                for (uint32_t z = 0u; z < 2u; z++) {
                    const float x = __uint_as_float(z << 31);
                    if (__float_as_uint(sqrt_plain(x)) != __float_as_uint(__builtin_sqrtf(x))) { if (!bad) bad_x = z << 31; bad++; }
                }
            }
            r[0] = (float)bad; r[1] = __uint_as_float(bad_x);
            break;
        }
        case BRT_DBG_ENCODE: {   // the store conversions of BRT_FLAG_OUT_* on one value
            const uint2 h = OutPixel<BRT_FLAG_OUT_RGBA16F>::make(make_float4(a[0], 0.0f, 0.0f, 0.0f));
            r[0] = (float)encode_srgb8(a[0]); r[1] = (float)encode_unorm8(a[0]); r[2] = (float)(h.x & 0xffffu);
            break;
        }
        default: break;
    }
}

// ---- host-callable launchers ----------------------------------------------------------------------

size_t trace_lds_bytes(const DeviceSceneView& sv, int scene_mode, uint32_t block, uint32_t pool_cap, uint32_t hist_words, bool rows) {
    size_t bytes = 0;
    if (scene_mode == SCENE_LDS) {
        bytes += pair_array_bytes(sv.n_pairs) + (size_t)sv.n_models * 16;
        bytes += (size_t)sv.n_leaf_table * 8 + (BRT_MAT_BY_SPHERE ? 0 : (size_t)sv.n_models * 4);   // (material ids only when a hit still goes through them)
    } else if (scene_mode == SCENE_LDS_TOP) {
        bytes += pair_array_bytes(sv.lds_pairs);
    }
    bytes += (size_t)(block / 64) * (sv.stack_entries + 2) * 64 * (sv.desc16 ? 2 : 4);   // + 2: DONE sentinel, one spare entry
    bytes = (bytes + 15) & ~(size_t)15;
    bytes += WGQ_BYTES;                                                                  // workgroup share of the pixel queue
    if (scene_mode == SCENE_LDS && BRT_WALK_ROWS && rows) bytes += (size_t)(block / 64) * ROWS_SCRATCH_BYTES;   // row-mode walk of thin waves
    if (pool_cap) bytes += 16 + (size_t)pool_cap * POOL_RECORD_BYTES;                    // drain pool: control words + records
    bytes += (size_t)hist_words * 4;                                                     // pre-pass: visits per pair record (FrameParams::record_hits)
    return bytes;
}

// the instantiations live in brt_trace_prod.hip (knobs folded) and brt_trace_tune.hip (knobs live)
hipError_t launch_trace_persistent_prod(const TraceLaunch& tl);
hipError_t launch_trace_persistent_tune(const TraceLaunch& tl);
hipError_t launch_trace_persistent(const TraceLaunch& tl) {
    return tl.frame.tunable ? launch_trace_persistent_tune(tl) : launch_trace_persistent_prod(tl);
}

template <bool D, bool C>
static hipError_t launch_simple_t(const TraceLaunch& tl, uint32_t grid) {
    hipLaunchKernelGGL((k_trace_simple<D, C>), dim3(grid), dim3(256), 0, tl.stream, tl.scene, tl.frame, tl.out_tile,
                       tl.raster_rgba, tl.raster_depth, tl.counters);
    return hipGetLastError();
}

hipError_t launch_trace_simple(const TraceLaunch& tl) {
    const uint32_t grid = (tl.frame.queue_size + 255u) / 256u;
    if (grid == 0) return hipSuccess;
    if (tl.scene.desc16) return tl.counters_on ? launch_simple_t<true, true>(tl, grid) : launch_simple_t<true, false>(tl, grid);
    return tl.counters_on ? launch_simple_t<false, true>(tl, grid) : launch_simple_t<false, false>(tl, grid);
}

hipError_t launch_passthrough(const FrameParams& fp, float* out_tile, const float* raster_rgba, hipStream_t stream) {
    const uint32_t grid = (fp.queue_size + 255u) / 256u;
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL(k_passthrough, dim3(grid), dim3(256), 0, stream, fp, reinterpret_cast<float4*>(out_tile),
                       reinterpret_cast<const float4*>(raster_rgba));
    return hipGetLastError();
}

template <uint32_t FMT>
static void launch_deinterleave_t(const float* tiles, void* frame, uint32_t width, uint32_t height, uint32_t n_parts, uint32_t tile_rows,
                                  const uint32_t* part_of_strip, hipStream_t stream) {
    hipLaunchKernelGGL(k_deinterleave<FMT>, dim3((width + 255u) / 256u, height), dim3(256), 0, stream, reinterpret_cast<const float4*>(tiles),
                       reinterpret_cast<typename OutPixel<FMT>::type*>(frame), width, height, n_parts, tile_rows, part_of_strip);
}
hipError_t launch_deinterleave(const float* tiles, void* frame, uint32_t width, uint32_t height, uint32_t n_parts,
                               uint32_t tile_rows, uint32_t out_format, hipStream_t stream, const uint32_t* part_of_strip) {
    if (width == 0 || height == 0) return hipSuccess;
    switch (out_format) {
        case BRT_FLAG_OUT_RGBA32F: launch_deinterleave_t<BRT_FLAG_OUT_RGBA32F>(tiles, frame, width, height, n_parts, tile_rows, part_of_strip, stream); break;
        case BRT_FLAG_OUT_RGBA8_UNORM_SRGB: launch_deinterleave_t<BRT_FLAG_OUT_RGBA8_UNORM_SRGB>(tiles, frame, width, height, n_parts, tile_rows, part_of_strip, stream); break;
        case BRT_FLAG_OUT_RGBA16F: launch_deinterleave_t<BRT_FLAG_OUT_RGBA16F>(tiles, frame, width, height, n_parts, tile_rows, part_of_strip, stream); break;
        case BRT_FLAG_OUT_RGBA8_UNORM: launch_deinterleave_t<BRT_FLAG_OUT_RGBA8_UNORM>(tiles, frame, width, height, n_parts, tile_rows, part_of_strip, stream); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_pack_strips(const float* frame, float* packed, uint32_t width, uint32_t height, uint32_t n_parts, uint32_t tile_rows,
                              uint32_t floats_per_pixel, hipStream_t stream) {
    if (width == 0 || height == 0 || n_parts < 2 || tile_rows == 0) return hipSuccess;
    const dim3 grid((width + 255u) / 256u, tile_rows, n_parts - 1u);
    if (floats_per_pixel == 4u)
        hipLaunchKernelGGL(k_pack_strips<float4>, grid, dim3(256), 0, stream, reinterpret_cast<const float4*>(frame),
                           reinterpret_cast<float4*>(packed), width, height, n_parts, tile_rows);
    else
        hipLaunchKernelGGL(k_pack_strips<float>, grid, dim3(256), 0, stream, frame, packed, width, height, n_parts, tile_rows);
    return hipGetLastError();
}

hipError_t launch_debug_eval(uint32_t op, const float* in, float* out, uint32_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_debug_eval, dim3((n + 255u) / 256u), dim3(256), 0, stream, op, in, out, n);
    return hipGetLastError();
}

}  // namespace brt
