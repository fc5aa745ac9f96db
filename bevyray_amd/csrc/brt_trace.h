// brt_trace.h -- the persistent trace kernel (template) and its device-side helpers.  Included by
// brt_kernels.hip (bring-up / copy kernels, launch dispatch) and by the two translation units that
// instantiate the kernel: brt_trace_prod.hip (TUNABLE = false: every tuning knob folded to its default,
// the lane queue compiled out) and brt_trace_tune.hip (TUNABLE = true: knobs read from FrameParams).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "brt_device.h"
#include "brt_kernels.h"

#ifndef BRT_PHASE_PRIO_AT
#define BRT_PHASE_PRIO_AT 2   // 0: no phase priorities; 1: raised for the walk; 2: raised from the top of the round (walk begin included)
#endif

namespace brt {

// ---- queue slot -> pixel -------------------------------------------------------------------
// Slot q: tile = q / 64, inside the tile row-major 8x8.  Tiles run along x inside a strip of
// BRT_STRIP_ROWS (= 8) rows; local strip k of part p is frame strip k * n_parts + p.
struct PixelCoord {
    uint32_t px, py;        // frame coordinates
    uint32_t local_row;     // row in the dense tile buffer
    uint32_t tile;          // tile id (strip * tiles_x + tx), < 2^24
    uint32_t t;             // position inside the tile (row-major 8x8), 0 .. 63
    bool inside;
};
BRT_DEV uint32_t slot_tile(const FrameParams& fp, uint32_t slot_tile_index) {
    return fp.tile_order ? fp.tile_order[slot_tile_index] : slot_tile_index;   // dispatch order, if known
}
template <bool TUNABLE = true>
BRT_DEV PixelCoord slot_to_pixel(const FrameParams& fp, uint32_t q, uint32_t tile) {
    const uint32_t t = q & 63u;
    const uint32_t sq = tile / fp.tiles_x, tx = tile - sq * fp.tiles_x;
    // queue order: bottom strips first when fp.bottom_up (a tuning knob: longest-pixels-first heuristic)
    const uint32_t strip = (TUNABLE && fp.bottom_up) ? (fp.local_strips - 1u - sq) : sq;
    PixelCoord c;
    c.px = tx * 8u + (t & 7u);
    const uint32_t r = t >> 3;
    c.local_row = strip * 8u + r;
    c.py = (fp.strip_of ? fp.strip_of[strip] : strip * fp.n_parts + fp.part) * 8u + r;
    c.tile = tile;
    c.t = t;
    c.inside = (c.px < fp.width) && (c.py < fp.height);
    return c;
}

BRT_DEV uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }
BRT_DEV uint32_t mbcnt64(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

BRT_DEV uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Per-pixel state of a lane
struct PixelState {
    float ndc0x, ndc0y;     // uv.x*2-1, 1-uv.y*2   (raytrace.wgsl:146-147)
    f3 sum;                 // running colour sum    (raytrace.wgsl:165)
    float dsum;             // running depth sum     (raytrace.wgsl:166)
    uint32_t rng;
    uint32_t sample;
    uint32_t out_index;     // pixel index in the tile buffer
    uint32_t frame_index;   // pixel index in the frame (raster inputs)
    uint32_t tile;          // bits 0-23: tile id (per-tile cost measurement, slice slot); 24-29: the pixel's position in the tile (slice slot);
                            // 30-31: kSliceFinal / kSliceFirst (what happens at sample_end) / kSliceDone
    uint32_t rays_begin;    // lane's ray counter when the pixel started
    uint32_t sample_end;    // the lane leaves the pixel after this many samples (sample_count, or half of it: a first-half job)
};

// ---- half-sample jobs (FrameParams::split_*; why: brt_host.cpp build_tile_order) ---------------------------------------------------
// A FIRST-half lane stops after sample_count / 2 samples; the SECOND-half lane of the same pixel (another wave, usually on another XCD,
// a millisecond or two later) goes on from there with the state {rng, sums, rays so far} that waits in slice_state.  Nobody waits and
// nothing is computed twice: whoever comes SECOND to the record's flag word carries the pixel on.
//   first half, at its end    writes the state, then flag <- READY;  old flag == GAVE UP: the second half has been here and left,
//                             this lane renders the second half itself (as if the tile had never been split)
//   second half, at its start flag <- GAVE UP;  old flag == READY: the state is there, take it;  anything else: leave the pixel to
//                             the first-half lane (this lane stays idle for the job)
// Both are one atomic exchange on the same word, so exactly one of the two lanes goes on.  The record crosses XCDs, whose L2s are not
// coherent with each other inside a kernel: the flag is an agent-scope atomic (performed at the memory side, like the tile queue's
// counter) with release / acquire semantics around the state words (BRT_SLICE_SYNC below).  A flag belongs to this launch by its serial number.
// A first-half lane does not go to memory at once: the exchanges are a round trip (microseconds) and the lanes of a tile end in ~30
// different rounds -- one by one that cost the headline frame 4 %.  The lane just goes idle (kSliceDone: its registers keep the state)
// and the WAVE settles all of them in one go the next time it runs its management code (slice_settle, top of the kernel's loop).
constexpr uint32_t kSliceFinal = 0u, kSliceFirst = 1u, kSliceDone = 3u, kSliceShift = 30u, kTileMask = 0x00ffffffu, kTileLaneShift = 24u,
                   kSliceFlagsMask = 0xc0000000u;
constexpr uint32_t kSliceReady = 1u, kSliceGaveUp = 2u;      // flag word = serial << 2 | one of these
#ifndef BRT_SLICE_EIGHTHS
#define BRT_SLICE_EIGHTHS 4
#endif
// where a split tile's first job ends (samples).  Shorter last jobs -- 5/8, 6/8, 7/8 -- measured: config 2 9.14 / 9.14 / 9.20 against 9.14-9.19 at
// the half, config 5 16.9-17.5 at 6/8 against 16.7-16.9 (profiles/r04/split_tail.txt)
BRT_DEV uint32_t slice_point(const FrameParams& fp) { return (uint32_t)(((uint64_t)fp.sample_count * BRT_SLICE_EIGHTHS) >> 3); }

// How the 32-byte record crosses from one wave to another (usually another XCD, whose L2 is not coherent with this one's inside a
// kernel).  BRT_SLICE_SYNC:
//   1 (default)  every access to the record is an agent-scope access of its own: the state as two wide stores with sc1 (written
//                through to the memory side: what the memory model lowers an agent-scope atomic store to, 128 and 64 bits wide --
//                tearing does not matter, nobody reads before the flag), `s_waitcnt vmcnt(0)` (their acknowledgements), THEN the
//                flag by an agent-scope exchange; the taker's exchange first, and behind its return two dwordx4 sc1 loads (read at the
//                memory side, never from a stale line of this L2).  This is the release / acquire pair of (2) with the cache-wide
//                operations left out that this record does not need: `buffer_wbl2 sc1` writes back lines that were NOT written
//                through (ours were), `buffer_inv sc1` drops lines that plain loads might hit (ours bypass) -- the form
//                MI355X_MICROARCH.md lists under "Valid forms" (every store of the handed-off bytes sc1 and drained before the
//                flag, every load of them a global sc1 load to registers), which it marks as measured, not architectural; so is
//                this one (stress: scripts/split_stress.py).  2 stores + 1 atomic | 1 atomic + 2 loads per pixel; HBM traffic of
//                the headline launch 342 -> ~100 MB.
//   2            the memory model's recipe verbatim: plain wide stores, the flag exchange with RELEASE semantics (hipcc emits
//                buffer_wbl2 sc1 + s_waitcnt before it), the taker's with ACQUIRE (buffer_inv sc1 behind it), plain wide loads.
//                Correct by the book and measurably slower, because the invalidate drops this XCD's whole L2 every time a wave
//                starts a second-half tile: headline frame 9.22 -> 9.29 ms, the 10 004-sphere frame (whose tree lives in L2)
//                14.89 -> 15.72 ms (+5.6 %); profiles/r05/slice_sync_ab.txt.
//   0            round 4: every word by its own agent-scope atomic (7 + 1 exchanges, 6 fetch_add(0)), the flag behind the returns
//                of the others by a data dependency.  Same speed as (1), 3.4 x the HBM traffic.
// All three render the same frames (forced-split suite, fuzz; scripts/split_stress.py: every tile of a 1080p frame split, 200
// frames, each against the unsplit render).
#ifndef BRT_SLICE_SYNC
#define BRT_SLICE_SYNC 1
#endif
typedef uint32_t slice_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t slice_u2 __attribute__((ext_vector_type(2)));
BRT_DEV uint32_t slice_xchg(uint32_t* p, uint32_t v) { return __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
BRT_DEV uint32_t slice_read(uint32_t* p) { return __hip_atomic_fetch_add(p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Where a pixel's record lives: slot = tile * 64 + position in the tile, in three planes -- A {rng, sum.xyz} 16 B, B {depth sum, rays
// so far, -, -} 16 B, F the flag word -- so that the 64 lanes of a tile, which settle together (slice_settle: the wave's management
// code), write and read WHOLE 128-byte lines with one instruction each (a wave's dwordx4 = 1 KB contiguous), and the flags sit in
// lines of their own.  (Round 4 / the first form of round 5 kept 32-byte records in pixel order: half-filled lines, 240 MB of HBM
// traffic per headline launch.)  Plane B is only needed by frames that average depth (levels 1, 2) or measure tile costs: the
// steady-state frame of a Pure-level view moves 16 + 16 bytes per split pixel.
struct SliceSlot { uint32_t* a; uint32_t* b; uint32_t* f; };
BRT_DEV SliceSlot slice_slot(const FrameParams& fp, const PixelState& ps) {
    const size_t slot = (size_t)(ps.tile & kTileMask) * 64u + ((ps.tile >> kTileLaneShift) & 63u), n = fp.queue_size;   // n = tiles * 64
    SliceSlot r;
    // (flags first: a frame of another size in the same buffer then finds, where its flags are, only flags of earlier launches --
    //  never a state word that might look like one; the buffer only grows with a fresh, zeroed allocation: attach_tile_order)
    r.f = fp.slice_state + slot;
    r.a = fp.slice_state + n + 4u * slot;
    r.b = fp.slice_state + 5u * n + 4u * slot;
    return r;
}

// first half: true when the second-half lane has already given up on this pixel (the caller then carries on with it)
BRT_DEV bool slice_store(const FrameParams& fp, const PixelState& ps, uint32_t rays, bool with_b) {
    const SliceSlot rec = slice_slot(fp, ps);
    const uint32_t ready = (fp.slice_serial << 2) | kSliceReady, gave_up = (fp.slice_serial << 2) | kSliceGaveUp;
    const slice_u4 a = {ps.rng, __float_as_uint(ps.sum.x), __float_as_uint(ps.sum.y), __float_as_uint(ps.sum.z)};
    const slice_u4 b = {__float_as_uint(ps.dsum), rays, 0u, 0u};
#if BRT_SLICE_SYNC == 2
    *reinterpret_cast<slice_u4*>(rec.a) = a;
    if (with_b) *reinterpret_cast<slice_u4*>(rec.b) = b;
    return __hip_atomic_exchange(rec.f, ready, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT) == gave_up;
#elif BRT_SLICE_SYNC == 1
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(rec.a), "v"(a) : "memory");
    if (with_b) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(rec.b), "v"(b) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the stores are acknowledged (written through) before the flag goes out
    return slice_xchg(rec.f, ready) == gave_up;
#else
    uint32_t seen = slice_xchg(rec.a + 0, a.x);
    seen |= slice_xchg(rec.a + 1, a.y);
    seen |= slice_xchg(rec.a + 2, a.z);
    seen |= slice_xchg(rec.a + 3, a.w);
    seen |= slice_xchg(rec.b + 0, b.x);
    seen |= slice_xchg(rec.b + 1, b.y);
    // the flag goes out when the six exchanges have RETURNED (performed): its value is made to depend on what they returned
    uint32_t flag = ready;
    asm volatile("v_and_b32 %1, 0, %1\n\tv_or_b32 %0, %0, %1" : "+v"(flag), "+v"(seen));
    return slice_xchg(rec.f, flag) == gave_up;
#endif
}

// second half: true when the state was there (ps continues at sample_count / 2; *rays_before = rays of the first half)
BRT_DEV bool slice_load(const FrameParams& fp, PixelState& ps, uint32_t* rays_before, bool with_b) {
    const SliceSlot rec = slice_slot(fp, ps);
    const uint32_t ready = (fp.slice_serial << 2) | kSliceReady, gave_up = (fp.slice_serial << 2) | kSliceGaveUp;
    slice_u4 a, b = {0u, 0u, 0u, 0u};
#if BRT_SLICE_SYNC == 2
    if (__hip_atomic_exchange(rec.f, gave_up, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != ready) return false;
    a = *reinterpret_cast<const slice_u4*>(rec.a);
    if (with_b) b = *reinterpret_cast<const slice_u4*>(rec.b);
#elif BRT_SLICE_SYNC == 1
    if (slice_xchg(rec.f, gave_up) != ready) return false;
    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(a) : "v"(rec.a) : "memory");
    if (with_b) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "+&v"(b) : "v"(rec.b) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a), "+v"(b) :: "memory");      // (the loads' results are not the compiler's to wait for: it did not issue them)
#else
    if (slice_xchg(rec.f, gave_up) != ready) return false;
    a.x = slice_read(rec.a + 0); a.y = slice_read(rec.a + 1); a.z = slice_read(rec.a + 2); a.w = slice_read(rec.a + 3);
    b.x = slice_read(rec.b + 0); b.y = slice_read(rec.b + 1);
#endif
    ps.rng = a.x;
    ps.sum = mk3(__uint_as_float(a.y), __uint_as_float(a.z), __uint_as_float(a.w));
    ps.dsum = __uint_as_float(b.x);
    *rays_before = b.y;
    ps.sample = slice_point(fp);
    return true;
}

BRT_DEV void pixel_begin(const FrameParams& fp, const PixelCoord& c, PixelState& ps) {
    const float uvx = ((float)c.px + 0.5f) / (float)fp.width;
    const float uvy = ((float)c.py + 0.5f) / (float)fp.height;
    ps.rng = pixel_seed(fp, uvx, uvy);
    ps.ndc0x = uvx * 2.0f - 1.0f;
    ps.ndc0y = 1.0f - uvy * 2.0f;
    ps.sum = mk3(0.0f, 0.0f, 0.0f);
    ps.dsum = 0.0f;
    ps.sample = 0;
    ps.out_index = c.local_row * fp.width + c.px;
    ps.frame_index = fp.raster_dense ? ps.out_index : c.py * fp.width + c.px;
    ps.tile = c.tile | (c.t << kTileLaneShift);
}

// PURE_LEVEL: the launch is known to be level 3 (Raytracing::Pure): no depth average, no raster inputs
template <bool PURE_LEVEL = false>
BRT_DEV void pixel_finish(const FrameParams& fp, const PixelState& ps, float* out_tile, const float* raster_rgba,
                          const float* raster_depth) {
    const f3 avg = mk3(ps.sum.x / fp.spp_f, ps.sum.y / fp.spp_f, ps.sum.z / fp.spp_f);   // :169
    if (PURE_LEVEL) {
        reinterpret_cast<float4*>(out_tile)[ps.out_index] = make_float4(avg.x, avg.y, avg.z, 1.0f);      // :122
        return;
    }
    const float avg_depth = ps.dsum / fp.spp_f;                                           // :170
    reinterpret_cast<float4*>(out_tile)[ps.out_index] =
        resolve_pixel(fp, avg, avg_depth, raster_rgba, raster_depth, ps.frame_index);
}

// One ray segment of the bounce loop, raytrace.wgsl:189-212, after raycast returned (t, idx).
// Returns true when the sample has ended; then `color` is its gamma-encoded colour (:223).
template <bool COUNTERS>
BRT_DEV bool shade_segment(const ScenePtrs& sc, const FrameParams& fp, f3& o, f3& d, f3& tput, uint32_t& bounce,
                           float& first_depth, float t, uint32_t idx, uint32_t& rng, f3& color, HitCounters& hc,
                           bool or_short_circuit = false) {
    if (bounce == 0) first_depth = t;                               // :193-195
    f3 light = mk3(0.0f, 0.0f, 0.0f);
    bool ended;
    if (t == kInf) {                                                // :198-201
        light = background_gradient(d);
        ended = true;
    } else {
        hc.hits++;
        f3 att;
        const bool absorbed = scatter<COUNTERS>(sc, o, d, t, idx, rng, att, hc, or_short_circuit);  // :204
        if (absorbed) {
            ended = true;                                           // :207-209, light stays 0
        } else {
            tput = tput * att;                                      // :211
            bounce++;
            ended = bounce > fp.bounce_count;                       // loop exit, :189
            if (ended) tput = mk3(0.0f, 0.0f, 0.0f);                // :215-217
        }
    }
    if (ended) {
        const f3 c = tput * light;
        color = mk3(__builtin_sqrtf(c.x), __builtin_sqrtf(c.y), __builtin_sqrtf(c.z));  // :223
    }
    return ended;
}

// ---- the landed ray segments of a ROUND, for a divergent wave ------------------------------------------------
// What shade_segment + scatter do for one ray (raytrace.wgsl:189-223, :231-299), then -- for the lanes whose sample ended
// here and whose pixel has samples left -- the next sample's camera ray (:162, :175-186), arranged by what a wave pays:
// a round of the persistent kernel has lanes in every state (on the cover frame ~25 of 64 end at the sky, ~38 scatter,
// ~25 start a sample), and each divergent section costs the wave its full instruction count.  The shader has five
// normalize() calls on a sample's way (camera ray, sky gradient, hit normal, reflected / glass direction); a lane is in
// at most two of them per round, so TWO bodies serve all five:
//   N1  normalize(hit ? position - centre : direction)        hit normal | sky gradient
//   N2  normalize(metal ? reflect(d, n) : glass ? d : camera)  metal | glass | the next sample's camera ray
// and the two RNG draws of the camera jitter follow the sample's last draw in the lane's own sequence, wherever the
// sample ended.  Per lane: the shader's operations, operands and order -- only the company in the wave changes.
//
// Samples end EARLY (before N2: sky; diffuse absorbed or at the bounce limit; glass at the bounce limit, whose draw of
// :269 is made first) or LATE (metal absorbed or at the limit -- `absorbed` needs N2 -- and glass at the limit under the
// or-short-circuit policy): early ones get their camera ray in N2, late ones are flagged need_cam and get it at the top
// of the next round (1 in ~40 samples on the cover frame).
// SRV: the rejection sampler is a stage of the workgroup (ball_server_asm, brt_device.h): the hit lanes POST {rng, points needed} behind the
// material lottery, the round's shading that does not need the points runs, and they PICK the points UP behind the second normalize.
// What changes per lane is nothing -- the same draws, operations and operands in the same order --; what changes in the wave: a diffuse
// sample that ends here (absorbed, or at the bounce limit) is found out behind the pick-up, i.e. LATE (its camera ray is made at the top
// of the next round), and a metal lane's `u + fuzz` moves behind the pick-up too.  srv_serial: the round's serial number (wave-uniform).
typedef uint32_t lds_u4 __attribute__((ext_vector_type(4)));
typedef uint32_t lds_u2 __attribute__((ext_vector_type(2)));
template <bool COUNTERS, int LEAN, bool TUNABLE, bool SRV = false>
BRT_DEV void shade_landed(const ScenePtrs& sc, const FrameParams& fp, bool landed, float t, uint32_t idx, PixelState& ps,
                          f3& o, f3& d, f3& tput, uint32_t& bounce, float& first_depth, bool& active, bool& need_cam,
                          uint32_t& n_rays, HitCounters& hc, float* out_tile, const float* raster_rgba,
                          const float* raster_depth, uint32_t srv_serial = 0u) {
    const bool osc = TUNABLE && (fp.policy_flags & 1u);      // alternative reading of :269 (fixtures only)
    const bool sel = TUNABLE && (fp.policy_flags & 2u);      // ... of min / max (:263, :405): compare-select
    const bool pw5 = TUNABLE && (fp.policy_flags & 4u);      // ... of pow (:415): exp2(5 log2 x)
    const bool hit = landed && t != kInf, sky = landed && t == kInf;
    prof_section<COUNTERS>(hc, SEC_SCATTER, hit);
    prof_section<COUNTERS>(hc, SEC_SKY, sky);
    // a sample ends with linear colour c: :223, :165-166, then the pixel if that was its last sample
    auto end_sample = [&](f3 c) {
        ps.sum = ps.sum + sqrt3(c);                                                                   // :223, :165
        if (!LEAN) ps.dsum = ps.dsum + (first_depth == kInf ? fp.fallback_far : first_depth);        // :166,219-221 (levels 1, 2 only)
        ps.sample++;
        bounce = 0;
        need_cam = true;
        if (ps.sample == ps.sample_end) {
            const uint32_t at_end = ps.tile >> kSliceShift, tile = ps.tile & kTileMask;
            if (at_end == kSliceFirst) {
                ps.tile |= kSliceDone << kSliceShift;               // settled later, by the wave (slice_settle)
            } else {
                const uint32_t rays = n_rays - ps.rays_begin;       // (a second half: of the whole pixel)
                pixel_finish<LEAN != 0>(fp, ps, out_tile, raster_rgba, raster_depth);
                // (in the LEAN instantiations too since round 4: a view whose camera moves measures every frame -- launch_part -- and
                //  must not fall back to the general instantiation for it)
                if (fp.tile_cost) {
                    atomicAdd(&fp.tile_cost[tile], rays);
                    atomicMax(&fp.tile_cost[fp.local_strips * fp.tiles_x + tile], rays);
                }
            }
            active = false;
        }
    };
    if (!landed) return;
    n_rays++;
    if (bounce == 0) first_depth = t;                                              // :193-195
    // ---- N1 ----
    // (state is updated IN PLACE and as early as its old value is dead -- o here, for every hit lane, whether its sample
    // goes on or not: a conditional assignment at the end costs the wave a copy of every such value at each join)
    f3 v1 = d;
    if (hit) {
        hc.hits++;
        const float4 s = sc.spheres[idx];
        o = mk3(o.x + t * d.x, o.y + t * d.y, o.z + t * d.z);                      // ray_at, :130-132: the next segment's origin
        v1 = mk3(o.x - s.x, o.y - s.y, o.z - s.z);                                 // :356
    }
    const f3 n1 = normalize3(v1);
    // ---- hit: the material lottery and the balls (scatter(), brt_device.h) ----
    // (kind, att, acc, ior, draw: only ever read for hit lanes -- deliberately left undefined for the others)
    bool metal = false, glass = false, diffuse = false, absorbed = false;
    f3 att, acc;
    float ior, draw, m1x_of_hit = 0.0f;
    if (hit) {
#if BRT_MAT_BY_SPHERE
        const float4 m0 = sc.sphere_mats[2 * idx];      // base_color.rgb, metallic
        const float4 m1 = sc.sphere_mats[2 * idx + 1];  // roughness, reflectance, ior, specular_transmission
#else
        const uint32_t mid = sc.sphere_material[idx];
        const float4 m0 = sc.materials[2 * mid];
        const float4 m1 = sc.materials[2 * mid + 1];
#endif
        metal = rng_float(ps.rng) < m0.w;                                          // :234
        glass = !metal && (rng_float(ps.rng) < m1.w);                              // :249 (drawn only when not metal)
        diffuse = !metal && !glass;
        if (glass && !osc) draw = rng_float(ps.rng);                               // :269, default policy: always drawn
        ior = m1.z;
        m1x_of_hit = m1.x;
        att = glass ? mk3(1.0f, 1.0f, 1.0f) : mk3(m0.x, m0.y, m0.z);
        uint32_t need = metal ? 1u : (diffuse ? 2u : 0u);
        acc = diffuse ? n1 : mk3(-0.0f, -0.0f, -0.0f);
        float scale = diffuse ? 1.0f : m1.x;
        unsigned long long t_ball = 0;
        if (COUNTERS) t_ball = wall_clock64();
        if (SRV) need = 0u;                                                       // (the points come from the servers: below)
        else if (BRT_BALL_ASM && !COUNTERS) {
            ball_loop_asm(ps.rng, acc, m1.x, __builtin_amdgcn_ballot_w64(diffuse), __builtin_amdgcn_ballot_w64(metal), &hc.asm_counts);
            need = 0u;
        }
        while (need != 0u) {                                                      // random.wgsl:19-24
            if (COUNTERS) prof_section<COUNTERS>(hc, SEC_BALL, true);
            const float px = rng_ball_coord(ps.rng);
            const float py = rng_ball_coord(ps.rng);
            const float pz = rng_ball_coord(ps.rng);
            const f3 p = mk3(px, py, pz);
            const bool ok = dot3(p, p) <= 1.0f;
            const f3 cand = acc + scale * p;
            acc = mk3(ok ? cand.x : acc.x, ok ? cand.y : acc.y, ok ? cand.z : acc.z);
            scale = ok ? m1.x : scale;
            need -= ok ? 1u : 0u;
        }
        if (COUNTERS) {   // booked by the first lane of the section, like prof_section
            const uint64_t m = __ballot(true);
            if (mbcnt64(m) == 0u) hc.ticks_ball += wall_clock64() - t_ball;
        }
        if (!SRV && diffuse) {                                                    // :281-297
            const float eps = 1e-8f;
            if (__builtin_fabsf(acc.x) < eps && __builtin_fabsf(acc.y) < eps && __builtin_fabsf(acc.z) < eps) acc = n1;
            absorbed = dot3(acc, n1) < 0.0f;
        }
    }
    // ---- sampler stage: post this round's requests (compacted: request k of the wave in entry k of its mailbox) ----
    const bool srv_need = SRV && hit && (metal || diffuse);
    uint32_t srv_slot = 0u;
    if (SRV) {
        const uint64_t nm = __ballot(srv_need);
        if (nm != 0ull) {
            srv_slot = sc.srv_mbox + 32u * mbcnt64(nm);
            if (srv_need) {
                const lds_u2 rq = {ps.rng, (metal ? 1u : 2u) | (srv_serial << 2)};
                *reinterpret_cast<__attribute__((address_space(3))) lds_u2*>((uintptr_t)srv_slot) = rq;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");           // the entries before the door
            if (mbcnt64(nm) == 0u && srv_need)                                // (the first posting lane rings)
                *reinterpret_cast<volatile __attribute__((address_space(3))) uint32_t*>((uintptr_t)sc.srv_door) = (srv_serial << 8) | (uint32_t)__popcll(nm);
        }
    }
    // ---- early ends ----
    const bool limit = hit && bounce >= fp.bounce_count;        // :189: this was the last segment the loop allows
    const bool early = sky || (!SRV && diffuse && (absorbed || limit)) || (glass && limit && !osc);
    bool cam = false;
    f3 cdir;
    if (early) {
        f3 c;
        if (sky) {
            const float a = 0.5f * (n1.y + 1.0f);                                  // :364-369
            const float b = 1.0f - a;
            c = tput * mk3(b * 1.0f + a * 0.5f, b * 1.0f + a * 0.7f, b * 1.0f + a * 1.0f);
        } else if (absorbed) {
            c = tput * mk3(0.0f, 0.0f, 0.0f);                                      // :207-209: light stays 0
        } else {
            c = mk3(0.0f, 0.0f, 0.0f);                                             // :215-217: throughput 0 (x light 0)
        }
        end_sample(c);
        cam = active;
        if (cam) {
            prof_section<COUNTERS>(hc, SEC_CAMERA, true);
            cdir = camera_dir_raw(fp, ps.ndc0x, ps.ndc0y, ps.rng);                 // :162 + :175-186
        }
    }
    // ---- N2 ----
    const bool second = (metal || glass) && !early;
    if (second || cam) {
        const float dn = dot3(d, n1);                                             // :358 front_face (incoming direction); reflect3
        f3 v2 = metal ? d - (2.0f * dn) * n1 : d;                                 // :238 reflect() / :261
        if (cam) v2 = cdir;
        const f3 u = normalize3(v2);
        if (cam) {
            d = u;                      // (origin, throughput, first depth of the new sample: top of the next round)
            need_cam = false;
        } else if (metal) {                                                       // :234-245
            if (SRV) d = u;                                                        // (+ the fuzz behind the pick-up)
            else { d = u + acc; absorbed = dot3(d, n1) < 0.0f; }
        } else {                                                                  // glass, :249-280
            const float ri = dn < 0.0f ? (1.0f / ior) : ior;
            const float cos_theta = sel ? min_sel(dot3(neg3(u), n1), 1.0f) : min_f(dot3(neg3(u), n1), 1.0f);
            const float sin_theta = __builtin_sqrtf(1.0f - cos_theta * cos_theta);
            const bool cannot_refract = ri * sin_theta > 1.0f;
            const float refl = schlick(cos_theta, ri, pw5);
            bool reflects = cannot_refract;
            if (osc) {
                if (!cannot_refract) reflects = refl > rng_float(ps.rng);
            } else {
                reflects = reflects || (refl > draw);
            }
            d = reflects ? reflect3(u, n1) : refract3(u, n1, ri, sel);
        }
    }
    // ---- sampler stage: pick the points up (valid when the entry's last word is this round's serial) ----
    if (SRV && __ballot(srv_need) != 0ull) {
        bool waiting = srv_need;
        for (uint32_t spin = 0u; ; spin++) {
            if (waiting) {
                const uint32_t flag = *reinterpret_cast<volatile __attribute__((address_space(3))) uint32_t*>((uintptr_t)(srv_slot + 28u));
                waiting = flag != srv_serial;
            }
            if (__ballot(waiting) == 0ull) break;
            // (a bound, not a protocol step: brt_device.h ball_server_asm.  Whoever runs into it poisons the workgroup -- control word 1 --
            //  and nobody of it polls again: a wrong frame, counted, instead of a kernel that stands still)
            volatile __attribute__((address_space(3))) uint32_t* poison = reinterpret_cast<volatile __attribute__((address_space(3))) uint32_t*>((uintptr_t)(sc.srv_ctl + 4u));
            if (spin > (1u << 20) || ((spin & 255u) == 255u && *poison != 0u)) { *poison = 1u; hc.hits |= 0x80000000u; break; }
            if (BRT_ASM_COUNT) hc.asm_counts.srv_wait_polls++;
            __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (srv_need) {
            const lds_u4 r0 = *reinterpret_cast<__attribute__((address_space(3))) lds_u4*>((uintptr_t)srv_slot);
            const lds_u4 r1 = *reinterpret_cast<__attribute__((address_space(3))) lds_u4*>((uintptr_t)(srv_slot + 16u));
            ps.rng = r0.x;
            if (diffuse) acc = acc + mk3(__uint_as_float(r0.y), __uint_as_float(r0.z), __uint_as_float(r0.w));      // normal + 1.0 * p1 (:285)
            acc = acc + m1x_of_hit * mk3(__uint_as_float(r1.x), __uint_as_float(r1.y), __uint_as_float(r1.z));     // + roughness * p (:238, :285)
            if (diffuse) {                                                        // :281-297
                const float eps = 1e-8f;
                if (__builtin_fabsf(acc.x) < eps && __builtin_fabsf(acc.y) < eps && __builtin_fabsf(acc.z) < eps) acc = n1;
                absorbed = dot3(acc, n1) < 0.0f;
            } else {                                                              // metal, :238-245
                d = d + acc;
                absorbed = dot3(d, n1) < 0.0f;
            }
        }
    }
    // ---- the paths that go on: next segment from the hit point; late ends ----
    if (hit && !early) {
        if (diffuse) d = acc;
        if (absorbed || limit) {                                                  // metal; glass under the or-short-circuit policy
            f3 c = mk3(0.0f, 0.0f, 0.0f);                                         // loop exit, :189 + :215-217: throughput 0 (x light 0)
            if (absorbed) c = tput * c;                                           // :207-209: light stays 0
            end_sample(c);
        } else {
            tput = tput * att;                                                    // :211
            bounce++;
        }
    }
}

// ---- drain pool ------------------------------------------------------------------------------
// When the pixel queue is empty a wave's lanes run out of pixels one by one, but a round costs the
// wave the same instructions with 5 live lanes as with 60: on the cover frame 18 % of all rounds were
// executed in that phase with 22 live lanes on average.  So the waves of a workgroup CONSOLIDATE:
// a wave that is down to `drain_donate` live paths finishes the walks in flight, writes its paths
// (pixel state + next ray segment, 23 words) to a pool in LDS and ends; waves with idle lanes take
// them over and continue them.  A path is the same sequence of operations whichever lane runs it, so
// pixels and counters do not change.
// Control words {lock, count, alive waves}; all three only change under the lock.  Invariants: the
// last alive wave never donates, and a wave only leaves when the pool is empty -- so every pooled
// path is picked up.  The lock holder runs straight-line code (no waiting inside).
BRT_DEV void pool_lock(uint32_t* ctl, uint32_t lane) {
    if (lane == 0)
        while (atomicCAS(&ctl[0], 0u, 1u) != 0u) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
BRT_DEV void pool_unlock(uint32_t* ctl, uint32_t lane) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) atomicExch(&ctl[0], 0u);
}
BRT_DEV uint32_t pool_peek(const uint32_t* ctl, int i) { return __atomic_load_n(&ctl[i], __ATOMIC_RELAXED); }

// ---- persistent kernel -----------------------------------------------------------------------

// MODE (brt_layout.h SceneMode): where pair records and spheres are read from.  SCENE_LDS: everything in LDS
// (the 32-byte materials stay in global memory, one read per hit).  SCENE_LDS_TOP: the first sv.lds_pairs pair
// records -- breadth-first order, i.e. the top levels of the tree, where every ray starts and most node visits
// happen -- are staged in LDS, deeper records and the spheres come from global memory (L2).  SCENE_GLOBAL: no
// staging.  D16: 16-bit descriptors and u16 stack entries (always with the LDS modes).  SIMPLE: see raycast
// (brt_device.h).  TUNABLE: tuning knobs live (FrameParams) instead of folded to their defaults, lane queue built in.
// LEAN: what the steady-state frame of a Pure-level view needs and nothing else, so that the other checks, registers
// and kernel arguments leave the round loop.  1: level 3 (no raster inputs, no depth average); 2: also no critical tiles (the host can rule them out: launch_part).  (A/B at the time it went in, round 2:
// headline frame 12.87 (0) -> 12.73 (1) -> 12.59 ms (2); current timings: docs/experiments.md.)
#ifndef BRT_PRIO_TOP
#define BRT_PRIO_TOP 1   // the phase priorities also for scenes walked from the LDS tile + global memory (k_trace_persistent, kPhasePrio)
#endif
// SRV: the rejection sampler as a stage of the workgroup (brt_device.h ball_server_asm): the last SRV_WAVES waves serve, the others trace.
template <int MODE, bool D16, bool SIMPLE, bool COUNTERS, bool TUNABLE, int LEAN, bool SRV = false>
__global__ __launch_bounds__(BRT_BLOCK) void k_trace_persistent(DeviceSceneView sv, FrameParams fp,
                                                                uint32_t* __restrict__ queue_counter,
                                                                float* __restrict__ out_tile,
                                                                const float* __restrict__ raster_rgba,
                                                                const float* __restrict__ raster_depth,
                                                                unsigned long long* __restrict__ counters) {
    static_assert(MODE == SCENE_GLOBAL || D16, "a scene staged in LDS always uses 16-bit descriptors");
    static_assert(!LEAN || (!COUNTERS && !TUNABLE), "LEAN is a specialisation of the production timing kernel");
    static_assert(!SRV || (LEAN == 2 && MODE == SCENE_LDS && BRT_HAND_ASM), "the sampler stage exists for the steady-state instantiation of an LDS-resident scene");
    using StackT = typename std::conditional<D16, int16_t, int32_t>::type;   // sign-extending loads: brt_layout.h
    extern __shared__ uint4 smem[];
    ScenePtrs sc;
    StackT* stacks;
    sc.materials = reinterpret_cast<const float4*>(sv.materials);
    sc.sphere_mats = reinterpret_cast<const float4*>(sv.sphere_mats);
    sc.boxes_ordered = sv.boxes_ordered != 0u;
    sc.pairs_far = reinterpret_cast<const char*>(sv.pairs);
    sc.near_bytes = 0u;
    sc.near_base = 0u;
    sc.sph_base = 0u;
    sc.rows_scratch = 0u;
    sc.hits = nullptr;
    sc.minmax_select = TUNABLE && (fp.policy_flags & 2u) != 0u;
    if (MODE == SCENE_LDS) {
        // carve: pair records | spheres | leaf_table | (material ids, when a hit still goes through them) | stacks
        const uint32_t pair_granules = (uint32_t)(pair_array_bytes(sv.n_pairs) / 16);
        float4* p = reinterpret_cast<float4*>(smem);
        float4* l_pairs = p; p += pair_granules;
        float4* l_sp = p; p += sv.n_models;
        uint2* p2 = reinterpret_cast<uint2*>(p);
        uint2* l_lt = p2; p2 += sv.n_leaf_table;
        uint32_t* p1 = reinterpret_cast<uint32_t*>(p2);
        uint32_t* l_sm = p1; p1 += BRT_MAT_BY_SPHERE ? 0u : sv.n_models;
        stacks = reinterpret_cast<StackT*>(p1);
        const float4* g_pairs = reinterpret_cast<const float4*>(sv.pairs);
        const float4* g_sp = reinterpret_cast<const float4*>(sv.spheres);
        const uint2* g_lt = reinterpret_cast<const uint2*>(sv.leaf_table);
        for (uint32_t i = threadIdx.x; i < pair_granules; i += blockDim.x) l_pairs[i] = g_pairs[i];
        for (uint32_t i = threadIdx.x; i < sv.n_models; i += blockDim.x) { l_sp[i] = g_sp[i]; if (!BRT_MAT_BY_SPHERE) l_sm[i] = sv.sphere_material[i]; }
        for (uint32_t i = threadIdx.x; i < sv.n_leaf_table; i += blockDim.x) l_lt[i] = g_lt[i];
        sc.pairs = reinterpret_cast<const char*>(l_pairs);
        sc.near_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(l_pairs);   // LDS byte address (walk_loop_wave_lds)
        sc.sph_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(l_sp);
        sc.spheres = l_sp; sc.sphere_material = BRT_MAT_BY_SPHERE ? sv.sphere_material : l_sm; sc.leaf_table = l_lt;
    } else {
        sc.spheres = reinterpret_cast<const float4*>(sv.spheres);
        sc.sphere_material = sv.sphere_material;
        sc.leaf_table = reinterpret_cast<const uint2*>(sv.leaf_table);
        if (MODE == SCENE_LDS_TOP) {
            // carve: the first lds_pairs pair records | stacks
            const uint32_t pair_granules = sv.lds_pairs * PAIR_UNITS;
            float4* l_pairs = reinterpret_cast<float4*>(smem);
            const float4* g_pairs = reinterpret_cast<const float4*>(sv.pairs);
            for (uint32_t i = threadIdx.x; i < pair_granules; i += blockDim.x) l_pairs[i] = g_pairs[i];
            sc.pairs = reinterpret_cast<const char*>(l_pairs);
            sc.near_bytes = sv.lds_pairs * PAIR_BYTES;
            sc.near_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(l_pairs);
            stacks = reinterpret_cast<StackT*>(l_pairs + pair_granules);
        } else {
            sc.pairs = reinterpret_cast<const char*>(sv.pairs);
            stacks = reinterpret_cast<StackT*>(smem);
        }
    }
    const uint32_t lane = lane_id();
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t n_waves = blockDim.x >> 6;
    const uint32_t n_trace = SRV ? n_waves - SRV_WAVES : n_waves;        // waves that trace (the rest serve the sampler)
    // This lane's column of the wave's [entry][64] stack array.  16-bit entries: lanes l and l + 32 share a dword (column 2 (l mod 32)
    // + l / 32) instead of lanes 2k and 2k + 1: the LDS serves a wave's access 32 lanes at a time, and two lanes of one half that sit at
    // different stack depths would hit the same bank at different addresses -- a 2-way conflict on every pop and push of the walk.
    const uint32_t stack_col = D16 ? ((lane & 31u) * 2u + (lane >> 5)) : lane;
    StackT* stk = stacks + wave * ((sv.stack_entries + 2u) * 64u) + stack_col;   // + 2: DONE sentinel (entry 0) and one spare entry
    // behind the stacks (same sums as trace_lds_bytes): the workgroup's share of the pixel queue, then the
    // drain pool (4 control words, then the records)
    char* const lds = reinterpret_cast<char*>(smem);
    uint32_t off = (uint32_t)(reinterpret_cast<char*>(stacks + n_waves * ((sv.stack_entries + 2u) * 64u)) - lds);
    off = (off + 15u) & ~15u;
    uint32_t* const wgq = reinterpret_cast<uint32_t*>(lds + off);
    off += WGQ_BYTES;
    if (MODE == SCENE_LDS && BRT_WALK_ROWS && fp.rows_on != 0u) {      // every wave's scratch for the row-mode walk of a thin wave (walk_rows_asm)
        sc.rows_scratch = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(lds + off) + wave * ROWS_SCRATCH_BYTES;
        off += n_waves * ROWS_SCRATCH_BYTES;
    }
    uint32_t* pool_ctl = nullptr;
    float4* pool = nullptr;
    if (fp.pool_cap != 0u) {
        pool_ctl = reinterpret_cast<uint32_t*>(lds + off);
        pool = reinterpret_cast<float4*>(lds + off + 16u);
        if (threadIdx.x == 0) { pool_ctl[0] = 0u; pool_ctl[1] = 0u; pool_ctl[2] = n_trace; pool_ctl[3] = 0u; }
    }
    // sampler stage: behind the pool {16 control words: [0] trace waves alive | a door word per wave | 2 KB of mailbox per trace wave}
    uint32_t srv_ctl = 0u;
    if (SRV) {
        const uint32_t srv_off = off + (fp.pool_cap != 0u ? 16u + fp.pool_cap * POOL_RECORD_BYTES : 0u);
        uint32_t* const sw = reinterpret_cast<uint32_t*>(lds + srv_off);
        for (uint32_t i = threadIdx.x; i < srv_lds_bytes(blockDim.x) / 4u; i += blockDim.x) sw[i] = i == 0u ? n_trace : 0u;
        srv_ctl = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)(lds + srv_off);
        sc.srv_ctl = srv_ctl;
        sc.srv_door = srv_ctl + 64u + 4u * wave;
        sc.srv_mbox = srv_ctl + 64u + 4u * n_waves + 2048u * wave;
    }
    // pre-pass of a scene walked from the tile + global memory: this workgroup's histogram of interior visits per record, behind the pool
    constexpr bool kHits = TUNABLE && MODE == SCENE_LDS_TOP;
    uint32_t* hist = nullptr;
    if (kHits && fp.record_hits != nullptr) {
        if (fp.pool_cap != 0u) off += 16u + fp.pool_cap * POOL_RECORD_BYTES;
        hist = reinterpret_cast<uint32_t*>(lds + off);
        for (uint32_t i = threadIdx.x; i < sv.n_pairs; i += blockDim.x) hist[i] = 0u;
        sc.hits = hist;
    }
    if (threadIdx.x < WGQ_BYTES / 4u) wgq[threadIdx.x] = 0u;
    __syncthreads();
    if (SRV && wave >= n_trace) {
        // a server: trace waves (wave - n_trace), (wave - n_trace) + SRV_WAVES, .. are its own
        const uint32_t s_id = (uint32_t)__builtin_amdgcn_readfirstlane((int)(wave - n_trace));
        uint32_t c8[8];
        ball_server_asm((uint32_t)__builtin_amdgcn_readfirstlane((int)(srv_ctl + 64u + 4u * n_waves)), (uint32_t)__builtin_amdgcn_readfirstlane((int)(srv_ctl + 64u)),
                        (uint32_t)__builtin_amdgcn_readfirstlane((int)srv_ctl), s_id, SRV_WAVES, (n_trace - s_id + SRV_WAVES - 1u) / SRV_WAVES, c8);
        if (lane == 0) {      // what the stage executed (brt_debug_profile): iterations, lanes in them, door polls, idle polls at the end
            atomicAdd(&counters[37], (unsigned long long)c8[0]);
            atomicAdd(&counters[38], (unsigned long long)c8[1]);
            atomicAdd(&counters[47], (unsigned long long)c8[2]);
            atomicMax(&counters[48], (unsigned long long)c8[3]);
        }
        return;
    }
    // slots a workgroup takes from the global queue at a time (chosen by the host, brt_api.cpp launch_part)
    const uint32_t wgq_batch = fp.wgq_batch >= 64u ? fp.wgq_batch : 64u;
    // tuning knobs: live in the TUNABLE instantiation, constants (brt_layout.h) in the production one
    const uint32_t queue_lane = TUNABLE ? fp.queue_lane : 0u;
    const uint32_t refill_min = TUNABLE ? fp.refill_min : kRefillMin;
    const uint32_t walk_exit_lanes = TUNABLE ? fp.walk_exit_lanes : kWalkExitLanes;
    const uint32_t leaf_vote = TUNABLE ? fp.leaf_vote : kLeafVote;
    const uint32_t drain_donate = TUNABLE ? fp.drain_donate : kDrainDonate;
    const uint32_t pool_adopt = TUNABLE ? fp.pool_adopt : kPoolAdopt;
    // queue slots [crit_begin, crit_end) hold the CRITICAL tiles; when the order was built on the GPU its count lives there too
    const uint32_t crit_end = LEAN == 2 ? 0u : (fp.order_meta ? fp.order_meta[0] * 64u : fp.crit_end);
    // half-sample jobs at the end of the order (FrameParams::split_*): slots [split_lo, split_mid) first halves, [split_mid, split_hi) second
    // halves.  The counting instantiations (and frames of very few samples) take a first half as the whole tile and skip the second.
    constexpr bool kSlices = !COUNTERS;
    const uint32_t split_tiles = fp.slice_state ? (fp.order_meta ? fp.order_meta[3] : fp.split_tiles) : 0u;
    const uint32_t split_mid = (fp.order_meta ? fp.order_meta[2] : fp.split_nonsky) * 64u;
    const uint32_t split_lo = split_mid - split_tiles * 64u, split_hi = split_mid + split_tiles * 64u;
    const bool slices = kSlices && fp.sample_count >= 16u;
    const bool slice_b = LEAN == 0 || fp.tile_cost != nullptr;      // the hand-over carries {depth sum, rays so far} too (slice_store)
    const uint32_t queue_size = fp.queue_size + split_tiles * 64u;

    PixelState ps;
    ps.sample = 0; ps.rng = 0; ps.out_index = 0; ps.frame_index = 0; ps.tile = 0; ps.rays_begin = 0;
    ps.ndc0x = ps.ndc0y = 0.0f; ps.sum = mk3(0.0f, 0.0f, 0.0f); ps.dsum = 0.0f; ps.sample_end = fp.sample_count;
    f3 o = mk3(0.0f, 0.0f, 0.0f), d = mk3(0.0f, 0.0f, 1.0f), tput = mk3(1.0f, 1.0f, 1.0f);
    uint32_t bounce = 0;
    float first_depth = kInf;
    bool active = false;
    bool exhausted = !TUNABLE;        // the lane queue has nothing (more) for this lane; without a lane queue: from the start
    bool in_flight = false;           // this lane's walk was suspended by walk_run's early exit
    bool need_cam = false;            // this lane's next segment is a camera ray that shade_landed has not made yet (new pixel, late end)
    bool crit = false;                // this lane's pixel is one of the frame's longest chains (FrameParams::crit_*)
    bool wave_crit = false;
    WalkState<StackT> walk;
    walk.a = 0.0f; walk.inv = mk3(0.0f, 0.0f, 0.0f); walk.closest = kInf; walk.closest_idx = 0xffffffffu;
    walk.cur = Desc<D16>::DONE; walk.sp = stk; walk.n = 0;
    walk.ox = walk.oy = walk.oz = 0u;
    uint32_t n_rays = 0;
    HitCounters hc = {};
    // COUNTERS build: when this wave started, when it first found the pixel queue empty, when it ended
    // (100 MHz wall clock; read by brt_debug_profile as words 24..29)
    unsigned long long t_start = 0, t_empty = 0, drain_lane_rounds = 0;
    unsigned long long t_mark = 0, ticks_refill = 0, ticks_walk = 0, ticks_shade = 0, ticks_pre = 0;   // phase times of this wave
#ifdef BRT_LIFE_HIST       // diagnostic build: wave lifetimes in the production instantiations too (finer bins: 2^13 ticks = 0.082 ms)
    unsigned long long lh_start = wall_clock64();
#endif
    if (COUNTERS) t_start = t_mark = wall_clock64();

    uint32_t srv_serial = 0u;         // SRV: number of this wave's rounds so far (the serial of its sampler requests)
    bool tiles_done = false;          // the tile queue (slots FrameParams::queue_lane .. queue_size) is empty
    bool again_mark = false;          // COUNTERS: the round that starts follows another one directly (phase times)
    // lane takes queue slot q of `tile`
    // first-half lanes that have ended (kSliceDone) hand their pixel over -- or carry on with it, if its second-half lane has come and
    // gone: slice_store.  Before anything reuses a lane: at the top of the management code.
    auto slice_settle = [&]() {
        const bool done = kSlices && !active && (ps.tile >> kSliceShift) == kSliceDone;
        if (__ballot(done) == 0ull) return;
        if (done) {
            ps.tile &= ~kSliceFlagsMask;
            if (slice_store(fp, ps, n_rays - ps.rays_begin, slice_b)) {
                ps.sample_end = fp.sample_count;        // (sample, rng, sums: as the first half left them; need_cam is set, bounce is 0)
                active = true;
            }
        }
    };
    auto begin_pixel = [&](uint32_t q, uint32_t tile) {
        const PixelCoord c = slot_to_pixel<TUNABLE>(fp, q, tile);
        const bool first_half = q >= split_lo && q < split_mid, second_half = q >= split_mid && q < split_hi;
        if (c.inside && !(second_half && !slices)) {
            pixel_begin(fp, c, ps);
            crit = q >= fp.crit_begin && q < crit_end;
            ps.rays_begin = n_rays;
            ps.sample_end = fp.sample_count;
            bool taken = true;                          // (a second-half lane whose first half is not there yet leaves the pixel alone)
            if (slices && first_half) {
                ps.sample_end = slice_point(fp);
                ps.tile |= kSliceFirst << kSliceShift;
            } else if (slices && second_half) {
                uint32_t rays_before = 0u;
                taken = slice_load(fp, ps, &rays_before, slice_b);
                ps.rays_begin = n_rays - rays_before;
                // (diagnostic, brt_debug_profile [62], [63]: second halves that took the pixel over / that left it to the first-half lane)
                const uint64_t tm = __ballot(taken), lm = __ballot(!taken);
                if (mbcnt64(tm | lm) == 0u) {
                    if (tm) atomicAdd(&counters[62], (unsigned long long)__popcll(tm));
                    if (lm) atomicAdd(&counters[63], (unsigned long long)__popcll(lm));
                }
            }
            if (fp.sample_count == 0) {
                // 0/0 per channel.  The sums are compile-time zeros here; keep them opaque: hipcc 7.2
                // otherwise folds the four divisions into one and then drops two channels of the
                // level-1/2 result (found by scripts/fuzz_parity.py; tests: sample_count 0).
                asm volatile("" : "+v"(ps.sum.x), "+v"(ps.sum.y), "+v"(ps.sum.z), "+v"(ps.dsum));
                pixel_finish<LEAN != 0>(fp, ps, out_tile, raster_rgba, raster_depth);
            } else if (taken) { active = true; bounce = 0; need_cam = true; }
        }
    };

    for (;;) {
        if (split_tiles != 0u) slice_settle();
        // A wave that carries one of the frame's CRITICAL pixels (FrameParams::crit_*) issues ahead of its SIMD
        // mates and takes no new pixels: its rounds get shorter as its other pixels end, and the frame cannot
        // end before that chain has.
        if (crit_end != 0u) {
            const bool wc = __ballot(active && crit) != 0ull;
            if (wc != wave_crit) {
                wave_crit = wc;
                if (wc) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
            }
        }
        if (COUNTERS) { const unsigned long long now = wall_clock64(); ticks_shade += now - t_mark; t_mark = now; }
        // ---- refill empty lanes from the lane queue (wave-aggregated; TUNABLE builds only: the default order hands
        //      out whole tiles, below) ----
        while (TUNABLE) {
            const bool need = !active && !exhausted && !wave_crit;
            const uint64_t m = __ballot(need);
            if (m == 0) break;
            // keep a wave's pixels of one cost class: take new ones only in batches of refill_min
            // (a round costs the max over its lanes, so a cheap pixel dropped among expensive ones
            // pays their price for each of its samples); always refill when nothing else runs
            if ((uint32_t)__popcll(m) < refill_min && __ballot(active) != 0ull) break;
            // The workgroup takes slots from the global queue a batch at a time (one atomic on the hot address
            // and one read of the order table per batch instead of per refill) and its waves share the batch
            // through LDS: {lock, lo, hi, batch base, queue empty} + the batch's tile ids.
            const uint32_t n_need = (uint32_t)__popcll(m);
            const uint32_t rank = mbcnt64(m);
            uint32_t q = 0xffffffffu, tile = 0;
            bool none_left;
            {
                pool_lock(wgq, lane);
                uint32_t lo = pool_peek(wgq, 1), hi = pool_peek(wgq, 2), bbase = pool_peek(wgq, 3);
                bool done = pool_peek(wgq, 4) != 0u;
                if (lo == hi && !done) {
                    // guided: the batches shrink with what is left of the queue (judged from this workgroup's last
                    // batch), down to single tiles, so that no workgroup sits on a big share when the queue runs dry
                    const uint32_t left = queue_lane > bbase ? queue_lane - bbase : 0u;
                    uint32_t batch = (left / (gridDim.x * 4u)) & ~63u;
                    batch = batch > wgq_batch ? wgq_batch : (batch < 64u ? 64u : batch);
                    // single tiles until the queue is past the CRITICAL tiles: the waves that carry them run at raised
                    // priority and should sit on different CUs, not eight to a workgroup
                    if (bbase < crit_end) batch = 64u;
                    uint32_t b = 0;
                    if (lane == 0) b = atomicAdd(queue_counter, batch);
                    b = (uint32_t)__shfl((int)b, 0, 64);
                    bbase = b;
                    lo = b < queue_lane ? b : queue_lane;
                    hi = b + batch < queue_lane ? b + batch : queue_lane;
                    if (hi < lo) hi = lo;
                    done = lo == hi;
                    const uint32_t ti = (b >> 6) + lane;
                    if (lane < (batch >> 6) && ti < (queue_lane >> 6)) wgq[8u + lane] = slot_tile(fp, ti);
                    if (lane == 0) { wgq[2] = hi; wgq[3] = bbase; wgq[4] = done ? 1u : 0u; }
                }
                const uint32_t avail = hi - lo;
                const uint32_t take = n_need < avail ? n_need : avail;
                if (need && rank < take) {
                    q = lo + rank;
                    tile = wgq[8u + ((q - bbase) >> 6)];
                }
                if (lane == 0) wgq[1] = lo + take;
                none_left = done && take == 0u;
                pool_unlock(wgq, lane);
            }
            if (need) {
                if (none_left) {
                    exhausted = true;
                    if (COUNTERS && t_empty == 0) t_empty = wall_clock64();
                } else if (q != 0xffffffffu) {
                    begin_pixel(q, tile);
                }
            }
        }
        // ---- nothing left to do (and the lane queue, if any, is empty): the next whole tile ----
        if (queue_lane != queue_size && !tiles_done && !wave_crit && __ballot(active) == 0ull &&
            __ballot(!exhausted) == 0ull) {
            uint32_t b = 0;
            if (lane == 0) b = atomicAdd(queue_counter + 1, 64u);
            b = queue_lane + (uint32_t)__shfl((int)b, 0, 64);
            if (b < queue_size) begin_pixel(b + lane, slot_tile(fp, b >> 6));
            else {
                tiles_done = true;
                if (COUNTERS && t_empty == 0) t_empty = wall_clock64();
            }
        }
        if (COUNTERS) { const unsigned long long now = wall_clock64(); ticks_refill += now - t_mark; t_mark = now; }
        // ---- drain: hand the paths over / take paths over / leave (see "drain pool" above) ----
        bool finish_walks = false;    // this round runs every walk to its end so that the wave can hand over next round
        if (fp.pool_cap != 0u && __ballot(exhausted) != 0ull && !wave_crit) {
            const uint64_t am = __ballot(active);
            uint32_t live = (uint32_t)__popcll(am);
            asm volatile("" : "+s"(live));   // a 32-bit scalar (see wave_count, brt_device.h)
            const bool quiet = __ballot(in_flight) == 0ull;              // no suspended walk: every live path is between two rays
            // (hand over only while the tile queue has a next tile for this wave: once it is empty every wave is in its last pixels, a
            //  handed-over path would only wait in the pool -- a chain standing still -- and join a wave that is thinning itself.
            //  Measured: whole frames -0.4 %, one eighth of config 4 109.4 -> 107.1 ms.)
            const bool thin = live != 0u && live <= drain_donate && !tiles_done;
            bool leave = false;
            if (live == 0u || (thin && quiet && pool_peek(pool_ctl, 2) > 1u) || (live <= pool_adopt && pool_peek(pool_ctl, 1) != 0u)) {
                pool_lock(pool_ctl, lane);
                const uint32_t count = pool_peek(pool_ctl, 1), alive = pool_peek(pool_ctl, 2);
                if (thin && quiet && alive > 1u && count + live <= fp.pool_cap) {
                    // hand over: live lane r writes record count + r
                    if (active) {
                        float4* rec = pool + 6u * (count + mbcnt64(am));
                        rec[0] = make_float4(ps.ndc0x, ps.ndc0y, ps.sum.x, ps.sum.y);
                        rec[1] = make_float4(ps.sum.z, ps.dsum, __uint_as_float(ps.rng), __uint_as_float(ps.sample));
                        rec[2] = make_float4(__uint_as_float(ps.out_index), __uint_as_float(ps.frame_index),
                                             __uint_as_float(ps.tile), __uint_as_float(n_rays - ps.rays_begin));
                        rec[3] = make_float4(o.x, o.y, o.z, d.x);
                        rec[4] = make_float4(d.y, d.z, tput.x, tput.y);
                        rec[5] = make_float4(tput.z, __uint_as_float(bounce), first_depth, __uint_as_float((crit ? 1u : 0u) | (need_cam ? 2u : 0u)));
                        active = false;
                    }
                    // (while the tile queue has tiles the wave stays: it takes one next round)
                    const bool stay = queue_lane != queue_size && !tiles_done;
                    if (lane == 0) { pool_ctl[1] = count + live; if (!stay) pool_ctl[2] = alive - 1u; }
                    leave = !stay;
                } else if (count != 0u && live < 64u) {
                    // take over: idle lane r of k takes record count - k + r
                    const uint64_t im = ~am;
                    const uint32_t idle = 64u - live;
                    const uint32_t k = idle < count ? idle : count;
                    const uint32_t r = mbcnt64(im);
                    if (!active && r < k) {
                        const float4* rec = pool + 6u * (count - k + r);
                        const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3], r4 = rec[4], r5 = rec[5];
                        ps.ndc0x = r0.x; ps.ndc0y = r0.y; ps.sum = mk3(r0.z, r0.w, r1.x); ps.dsum = r1.y;
                        ps.rng = __float_as_uint(r1.z); ps.sample = __float_as_uint(r1.w);
                        ps.out_index = __float_as_uint(r2.x); ps.frame_index = __float_as_uint(r2.y);
                        ps.tile = __float_as_uint(r2.z); ps.rays_begin = n_rays - __float_as_uint(r2.w);
                        ps.sample_end = (ps.tile >> kSliceShift) == kSliceFirst ? slice_point(fp) : fp.sample_count;
                        o = mk3(r3.x, r3.y, r3.z); d = mk3(r3.w, r4.x, r4.y); tput = mk3(r4.z, r4.w, r5.x);
                        bounce = __float_as_uint(r5.y); first_depth = r5.z;
                        crit = (__float_as_uint(r5.w) & 1u) != 0u; need_cam = (__float_as_uint(r5.w) & 2u) != 0u;
                        active = true; in_flight = false; exhausted = true;
                    }
                    if (lane == 0) pool_ctl[1] = count - k;
                } else if (live == 0u && count == 0u && (queue_lane == queue_size || tiles_done)) {
                    if (lane == 0) pool_ctl[2] = alive - 1u;
                    leave = true;
                }
                pool_unlock(pool_ctl, lane);
            }
            if (leave) break;
            finish_walks = wave_count(active) <= drain_donate && !tiles_done;
        } else if (__ballot(active) == 0 && (queue_lane == queue_size || tiles_done || __ballot(!exhausted) != 0ull)) {
            break;
        }
        if (__ballot(active) == 0) continue;      // (pool on) nothing live but paths may still arrive

        // ---- rounds.  While nothing above would apply (the test at the bottom) -- no tile, no hand-over, no take-over, no leave; no lane
        //      queue refill; the same critical pixels -- the rounds run in a loop of their own: the management code redefines every state variable on some path (a path taken over from the pool),
        //      and in ONE loop with it the compiler gave all 24 of them a second home and copied them there and back every round.
        bool again;
        do {
        if (COUNTERS && again_mark) { const unsigned long long now = wall_clock64(); ticks_shade += now - t_mark; t_mark = now; }
        prof_section<COUNTERS>(hc, SEC_ROUND, active);
        if (COUNTERS && __ballot(exhausted) != 0ull) drain_lane_rounds += (unsigned long long)__popcll(__ballot(active)) | (1ull << 32);
        const bool fresh = active && !in_flight;       // starts a ray segment in this round
        // Issue priority by phase (LDS-resident scenes): from here to the end of the walk a wave goes ahead of its SIMD mates that are
        // shading.  The walk is where the LDS round trips are (a step cannot start before the one before it has chosen its node), the
        // shading code is arithmetic that fills the slots those leave; without it the four waves of a SIMD take turns by age and a
        // walking wave waits behind a shading one with its next read not even issued.  (Critical waves stay at 3, FrameParams::crit_*.
        // Scenes walked from L2 -- config 5 -- lost 2 % with it under the compiler's walk loop; under the hand-written one they gain 2-3 %:
        // 15.1 -> 14.8 ms, profiles/r04/ab_walk_top.txt.)
        constexpr int kPhasePrio = (MODE == SCENE_LDS || BRT_PRIO_TOP) ? BRT_PHASE_PRIO_AT : 0;
        if (kPhasePrio == 2 && !wave_crit) __builtin_amdgcn_s_setprio(1);
        prof_section<COUNTERS>(hc, SEC_CAMERA_TOP, fresh && need_cam);
        if (fresh && bounce == 0) {
            // the first segment of a sample, raytrace.wgsl:175-186; its direction (:162) was made by shade_landed when the previous
            // sample ended, unless this is the pixel's first sample or that one ended late
            o = mk3(fp.cam_pos[0], fp.cam_pos[1], fp.cam_pos[2]);
            tput = mk3(1.0f, 1.0f, 1.0f);
            first_depth = kInf;
            if (need_cam) {
                d = camera_ray_dir(fp, ps.ndc0x, ps.ndc0y, ps.rng);
                need_cam = false;
            }
        }
        if (fresh) walk_begin<D16>(walk, sc, sv.root_desc, stk, d);
        if (COUNTERS) { const unsigned long long now = wall_clock64(); ticks_pre += now - t_mark; t_mark = now; }
        if (kPhasePrio == 1 && !wave_crit) __builtin_amdgcn_s_setprio(1);
        if (active) walk_run<64, COUNTERS, D16, SIMPLE, MODE, StackT, kHits, TUNABLE, LEAN != 2>(sc, walk, stk, o, d, finish_walks ? 0u : walk_exit_lanes, leaf_vote, hc);
        in_flight = active && walk_pending<D16, SIMPLE>(walk);
        if (COUNTERS) { const unsigned long long now = wall_clock64(); ticks_walk += now - t_mark; t_mark = now; }
        const bool landed = active && !in_flight;      // walk finished: shade this segment now
        const float t = walk.closest;
        const uint32_t idx = walk.closest_idx;
        if (kPhasePrio != 0 && !wave_crit) __builtin_amdgcn_s_setprio(0);
        if (SRV) srv_serial = (srv_serial + 1u) & 0xffffffu;
        shade_landed<COUNTERS, LEAN, TUNABLE, SRV>(sc, fp, landed, t, idx, ps, o, d, tput, bounce, first_depth, active, need_cam, n_rays, hc,
                                                   out_tile, raster_rgba, raster_depth, srv_serial);
        again_mark = true;
        {
            // another round at once unless the management code would do something: no live path (next tile / leave), a lane queue, other
            // critical pixels than before, or -- with a drain pool, for a wave that is not critical -- paths to hand over (thin, no walk in
            // flight, not the last wave alive) or to take over (the same tests as above, on what this round left)
            const uint32_t live_now = wave_count(active);
            bool mgmt = live_now == 0u || (crit_end != 0u && (__ballot(active && crit) != 0ull) != wave_crit) ||
                        (TUNABLE && __ballot(!active && !exhausted) != 0ull);     // (a lane that may still take a pixel from the lane queue)
            const bool pooled = fp.pool_cap != 0u && !wave_crit && (!TUNABLE || __ballot(exhausted) != 0ull);
            if (!mgmt && pooled && live_now <= pool_adopt)
                mgmt = (live_now <= drain_donate && !tiles_done && __ballot(in_flight) == 0ull && pool_peek(pool_ctl, 2) > 1u) ||
                       pool_peek(pool_ctl, 1) != 0u;
            finish_walks = pooled && live_now <= drain_donate && !tiles_done;
            again = __builtin_amdgcn_readfirstlane(mgmt ? 0 : 1) != 0;      // (wave-uniform by construction; the LDS peeks hide that from the compiler)
        }
        } while (again);
        again_mark = false;
    }

    // ---- pre-pass: the workgroup's histogram of record visits into the launch's (every wave of the workgroup ends up here) ----
    if (kHits && hist != nullptr) {
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < sv.n_pairs; i += blockDim.x) {
            const uint32_t c = hist[i];
            if (c != 0u) atomicAdd(&fp.record_hits[i], c);
        }
    }
    if (SRV) {
        // this trace wave has ended (no request of it is outstanding): the servers end behind the last one
        if (lane == 0) __hip_atomic_fetch_add(reinterpret_cast<__attribute__((address_space(3))) uint32_t*>((uintptr_t)srv_ctl), 0xffffffffu, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t gave_up = wave_sum(hc.hits >> 31);
        if (lane == 0 && gave_up != 0u) atomicAdd(&counters[46], (unsigned long long)gave_up);      // (never: a pick-up that ran into its bound)
    }
    // ---- counters: one atomic per wave ----
    const uint32_t r = wave_sum(n_rays);
    if (lane == 0) atomicAdd(&counters[0], (unsigned long long)r);
#if BRT_ASM_COUNT
    if (!COUNTERS) {      // what the hand-written loops of this wave executed (each call booked by one lane: sum over the lanes)
        const AsmCounts& c = hc.asm_counts;
        const uint32_t v[12] = {wave_sum(c.int_exec), wave_sum(c.int_lanes), wave_sum(c.leaf_exec), wave_sum(c.leaf_lanes), wave_sum(c.ball_exec), wave_sum(c.ball_lanes),
                                wave_sum(c.fix_int_lanes), wave_sum(c.fix_leaf_lanes), wave_sum(c.rows_int_exec), wave_sum(c.rows_calls), wave_sum(c.rows_cycles), wave_sum(c.wide_cycles)};
        if (lane == 0) {            // (word 32 is the tile queue's counter)
            for (int k = 0; k < 6; k++) atomicAdd(&counters[33 + k], (unsigned long long)v[k]);
            atomicAdd(&counters[39], (unsigned long long)v[6]);
            atomicAdd(&counters[43], (unsigned long long)v[7]);
            atomicAdd(&counters[44], (unsigned long long)v[8]);       // of the interior executions: in the row-mode walk of thin waves (walk_rows_asm)
            atomicAdd(&counters[5], (unsigned long long)v[9]);        // calls of that walk  (words 44 and 5 .. 7 are phase times in the COUNTERS build)
            atomicAdd(&counters[6], (unsigned long long)v[10]);       // shader clocks in the row-mode walk ...
            atomicAdd(&counters[7], (unsigned long long)v[11]);       // ... and in the wide hand-written walk (first active lane's view, summed over the waves)
            const uint32_t wp = wave_sum(c.srv_wait_polls);
            atomicAdd(&counters[49], (unsigned long long)wp);         // sampler stage: lane-polls of the pick-up
        }
    }
#endif
    if (COUNTERS) {
        const uint32_t a = wave_sum(hc.node_pops), b = wave_sum(hc.interior), c = wave_sum(hc.sphere_tests);
        if (lane == 0) {
            atomicAdd(&counters[1], (unsigned long long)a);
            atomicAdd(&counters[2], (unsigned long long)b);
            atomicAdd(&counters[3], (unsigned long long)c);
        }
    }
#ifdef BRT_LIFE_HIST
    if (!COUNTERS && lane == 0) {
        const unsigned long long lh_end = wall_clock64();
        atomicMax(&counters[24], ~lh_start);
        atomicMax(&counters[27], lh_end);
        atomicAdd(&counters[29], 1ull);
        atomicAdd(&counters[45], lh_end - lh_start);
        const unsigned long long bin = (lh_end - lh_start) >> 13;
        // bins 88 .. 151 (7.2 .. 12.4 ms)
        if (bin >= 88ull && bin < 152ull) atomicAdd(&counters[46 + ((bin - 88ull) >> 2)], 1ull << (16 * ((bin - 88ull) & 3ull)));
    }
#endif
    if (COUNTERS) {
        const unsigned long long t_end = wall_clock64();
        unsigned long long te = 0;   // earliest "queue empty" seen by a lane of this wave
        for (int l = 0; l < 64; l++) {
            const unsigned long long v = __shfl(t_empty, l, 64);
            if (v != 0 && (te == 0 || v < te)) te = v;
        }
        if (lane == 0) {
            atomicMax(&counters[24], ~t_start);                 // min start
            if (te) { atomicMax(&counters[25], ~te); atomicMax(&counters[26], te); atomicAdd(&counters[28], t_end - te); }
            atomicMax(&counters[27], t_end);
            atomicAdd(&counters[29], 1ull);
            atomicAdd(&counters[45], t_end - t_start);                       // wave lifetimes (their mean against the kernel's span: the tail)
            {   // histogram of the lifetimes: bins of 2^15 ticks (0.33 ms), four 16-bit counts to a word, words 46..61
                const unsigned long long bin = (t_end - t_start) >> 15;
                if (bin < 64ull) atomicAdd(&counters[46 + (bin >> 2)], 1ull << (16 * (bin & 3ull)));
            }
            atomicAdd(&counters[30], drain_lane_rounds & 0xffffffffull);   // live lanes summed over the rounds after "empty"
            atomicAdd(&counters[31], drain_lane_rounds >> 32);              // those rounds
            atomicAdd(&counters[5], ticks_refill);    // wave time in: the pixel refill loop (queue atomic, tile order, pixel_begin)
            atomicAdd(&counters[6], ticks_walk);      //               the walk loop
            atomicAdd(&counters[7], ticks_shade + (t_end - t_mark));   // shading of the landed rays, pixel finish
            atomicAdd(&counters[43], ticks_pre);                       // drain logic, camera ray, walk_begin
        }
    }
    if (COUNTERS) {
        unsigned long long tb = hc.ticks_ball;   // sum over the lanes that booked it
        for (int off = 32; off > 0; off >>= 1) tb += __shfl_down(tb, off, 64);
        if (lane == 0) atomicAdd(&counters[44], tb);
    }
    if (COUNTERS) {
        const uint32_t h = wave_sum(hc.hits);
        if (lane == 0) atomicAdd(&counters[4], (unsigned long long)h);
        for (int k = 0; k < 8; k++) {   // each execution was booked by ONE lane of the wave: sum over lanes
            const uint32_t e = wave_sum(hc.sec_exec[k]), l = wave_sum(hc.sec_lanes[k]);
            if (lane == 0) {
                atomicAdd(&counters[8 + 2 * k], (unsigned long long)e);
                atomicAdd(&counters[9 + 2 * k], (unsigned long long)l);
            }
        }
    }
}


template <int MODE, bool D, bool S, bool C, bool T, int LEAN = 0, bool SRV = false>
static hipError_t launch_persistent_t(const TraceLaunch& tl) {
    auto kern = k_trace_persistent<MODE, D, S, C, T, LEAN, SRV>;
    if (MODE == SCENE_LDS || MODE == SCENE_LDS_TOP) {
        // the hand-written walk loops (walk_wave_lds_asm, walk_wave_top_asm) address the pair records from LDS address 0: the dynamic LDS must start there
        static const size_t static_lds = [&] {
            hipFuncAttributes at{};
            return hipFuncGetAttributes(&at, reinterpret_cast<const void*>(kern)) == hipSuccess ? at.sharedSizeBytes : (size_t)1;
        }();
        if (static_lds != 0) return hipErrorInvalidConfiguration;
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)tl.lds_bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(tl.grid), dim3(tl.block), tl.lds_bytes, tl.stream, tl.scene, tl.frame, tl.queue_counter,
                       tl.out_tile, tl.raster_rgba, tl.raster_depth, tl.counters);
    return hipGetLastError();
}

template <int MODE, bool D, bool T>
static hipError_t launch_persistent_md(const TraceLaunch& tl) {
    // LEAN instantiations only where they matter: the production timing kernel on a simple (PLOC-shaped) tree staged in LDS
    if (!T && MODE == SCENE_LDS && tl.lean == 2 && tl.srv != 0u && tl.scene.simple_tree && !tl.counters_on)
        return launch_persistent_t<MODE, D, true, false, false, (!T && MODE == SCENE_LDS) ? 2 : 0, (!T && MODE == SCENE_LDS && BRT_HAND_ASM)>(tl);
    if (!T && MODE != SCENE_GLOBAL && tl.lean && tl.scene.simple_tree && !tl.counters_on)
        return tl.lean == 2 ? launch_persistent_t<MODE, D, true, false, false, (!T && MODE != SCENE_GLOBAL) ? 2 : 0>(tl)
                            : launch_persistent_t<MODE, D, true, false, false, (!T && MODE != SCENE_GLOBAL) ? 1 : 0>(tl);
    if (tl.scene.simple_tree)
        return tl.counters_on ? launch_persistent_t<MODE, D, true, true, T>(tl) : launch_persistent_t<MODE, D, true, false, T>(tl);
    return tl.counters_on ? launch_persistent_t<MODE, D, false, true, T>(tl) : launch_persistent_t<MODE, D, false, false, T>(tl);
}

// every (scene mode, descriptor width) combination that plan_launch can choose
template <bool T>
static hipError_t launch_persistent_all(const TraceLaunch& tl) {
    switch (tl.scene_mode) {
        case SCENE_LDS:
            if (!tl.scene.desc16) return hipErrorInvalidValue;
            return launch_persistent_md<SCENE_LDS, true, T>(tl);
        case SCENE_LDS_TOP:
            if (!tl.scene.desc16) return hipErrorInvalidValue;
            return launch_persistent_md<SCENE_LDS_TOP, true, T>(tl);
        default:
            return tl.scene.desc16 ? launch_persistent_md<SCENE_GLOBAL, true, T>(tl) : launch_persistent_md<SCENE_GLOBAL, false, T>(tl);
    }
}

}  // namespace brt
