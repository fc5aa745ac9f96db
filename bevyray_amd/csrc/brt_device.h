// brt_device.h -- device-side arithmetic of the ray loop for gfx950.
//
// Every function cites the reference lines whose RESULT it reproduces (reference
// assets/shaders/raytrace.wgsl, random.wgsl).  The structure is not the shader's: invariants
// are hoisted per ray / per frame, the BVH is walked through re-encoded pair records
// (brt_layout.h) with the top of the stack in a register, and hits carry (t, sphere id) until
// the walk ends.  Hoisting never changes a value: each hoisted expression is the same tree of
// separately rounded f32 operations (build flag -ffp-contract=off; correctly rounded
// divide/sqrt are hipcc's default).
//
// Numeric policy (same as oracle/bevyray_oracle.c):
//   dot = (x*x' + y*y') + z*z';  normalize(v) = v / sqrt(dot(v,v));
//   min/max = v_min_f32 / v_max_f32 (IEEE minNum/maxNum, -0 < +0);
//   pow(x,5) = (x*x)*(x*x)*x;  u32(f32) truncates and saturates.
#pragma once
#include <hip/hip_runtime.h>

#include "brt_layout.h"

namespace brt {

#define BRT_DEV __device__ __forceinline__

// The hand-written loops (walk_wave_lds_asm, ball_loop_asm) name gfx950 registers and count its wait states and its lgkmcnt by
// hand: a device pass for any other target (the Makefile's ARCH can be overridden) compiles them out and takes the
// compiler-built loops (walk_loop_wave, the `while (need)` sampler of shade_landed) instead.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#define BRT_HAND_ASM 0
#else
#define BRT_HAND_ASM 1
#endif

// Diagnostic build (-DBRT_ASM_COUNT, scripts/asm_count.py): the hand-written loops count their own executions and active lanes --
// interior steps, leaf steps, rejection-sampler iterations -- in scalar registers (s_add_u32 beside the vector work), summed per
// wave into control words 33..38 (brt_debug_profile); the steps of the rare waves that carry an unsafe ray and walk in the compiler's
// repairing loop first (walk_run) go into words 39 and 43 (lanes).  The COUNTERS instantiation counts the same quantities in the COMPILER's loops;
// this closes the gap between "the work of the frame" and "what the timed kernel executed" (What's weak #8).
#ifndef BRT_ASM_COUNT
#define BRT_ASM_COUNT 0
#endif
struct AsmCounts { uint32_t int_exec, int_lanes, leaf_exec, leaf_lanes, ball_exec, ball_lanes, fix_int_lanes, fix_leaf_lanes, rows_int_exec, rows_calls, rows_cycles, wide_cycles, srv_wait_polls; };
// (a call's counts are wave-uniform scalars; they are booked by the first lane that is active at the call, like prof_section, and
//  summed over the lanes at the end of the kernel)
__device__ __forceinline__ bool first_active_lane() {
    const uint64_t m = __ballot(true);
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) == 0u;
}

#ifndef BRT_MAT_BY_SPHERE
#define BRT_MAT_BY_SPHERE 1   // a hit reads its material beside the sphere's index (one read) instead of through the material id (two dependent reads; 0)
#endif
#ifndef BRT_EXEC_MOVES
#define BRT_EXEC_MOVES 0   // bit 0: sphere test, bit 1: ball loop -- `if` bodies of v_mov under EXEC instead of v_cndmask selects; measured: -0.0 ... +0.8 % (the compiler adds a branch per `if`), off
#endif

constexpr float kInf = 3.40282347e+38f;  // const.wgsl:2 (FLT_MAX, compared with ==)

struct f3 {
    float x, y, z;
};
BRT_DEV f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
BRT_DEV f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
BRT_DEV f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
BRT_DEV f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
BRT_DEV f3 operator*(float s, f3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
BRT_DEV f3 neg3(f3 a) { return mk3(-a.x, -a.y, -a.z); }
BRT_DEV float dot3(f3 a, f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
BRT_DEV float min_f(float a, float b) { return __builtin_fminf(a, b); }
BRT_DEV float max_f(float a, float b) { return __builtin_fmaxf(a, b); }
// the other reading of WGSL's min / max (implementation defined for a NaN operand): compare-select, min(a, b) = b < a ? b : a and
// max(a, b) = a < b ? b : a -- what brt_set_policy(BRT_POLICY_MINMAX_SELECT) switches the knobs-live instantiation to (the oracle's
// minmax_select policy); they differ from minNum / maxNum only when an operand is a NaN (and in the sign of a zero)
BRT_DEV float min_sel(float a, float b) { return b < a ? b : a; }
BRT_DEV float max_sel(float a, float b) { return a < b ? b : a; }

// ---- correctly rounded division with a shared reciprocal ----------------------------------------------------
// hipcc expands the correctly rounded f32 quotient n / d (-fhip-fp32-correctly-rounded-divide-sqrt) to
//     ds = v_div_scale(d), ns = v_div_scale(n)            operands scaled by 2^+-64 when a result would leave the normal range
//     r0 = v_rcp_f32(ds);  r1 = fma(fma(-ds, r0, 1), r0, r0)
//     q0 = ns * r1;  q1 = fma(fma(-ds, q0, ns), r1, q0);  q2 = v_div_fmas(fma(-ds, q1, ns), r1, q1)   (an fma, rescaled)
//     v_div_fixup(q2, d, n)                               zeros, infinities, NaNs, overflow / underflow
// (11 instructions, one of them quarter rate).  With |d| and |n| both in [2^-40, 2^40] ("plain"): nothing is scaled,
// v_div_fmas is the fma, v_div_fixup returns q2 -- so div_plain(n, rcp_refined(d)) below IS that expansion,
// instruction for instruction, without its three no-ops; and r1 depends on d alone, so the three quotients of a
// normalize() or the sphere tests of one ray share it.  Outside the plain range the callers use `/`.  Whether a
// WAVE takes the short form is decided by ballot (all its active lanes plain), so there is one straight-line body
// either way.  tests: BRT_DBG_DIV / BRT_DBG_DIV_SWEEP compare both forms bit for bit (2^-40 .. 2^40 x all mantissas).
#ifndef BRT_SHARED_RCP
#define BRT_SHARED_RCP 3   // bit 0: normalize(), bit 1: 1 / direction.  (A/B on one MI355X, headline frame: 13.33 ms without,
                           // 13.18 with bit 0, 13.08 with both; the sphere test's `/ a` with a per-ray reciprocal did NOT pay --
                           // its range guard and wave vote cost the leaf step what the shorter quotient saves: 13.16 ms)
#endif
constexpr float kPlainLo = 0x1p-40f, kPlainHi = 0x1p40f;
struct RcpRef { float d, r1; };
BRT_DEV RcpRef rcp_refined(float d) {
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e = __builtin_fmaf(-d, r0, 1.0f);
    RcpRef R;
    R.d = d;
    R.r1 = __builtin_fmaf(e, r0, r0);
    return R;
}
BRT_DEV float div_plain(float n, RcpRef R) {
    float q = n * R.r1;
    float e = __builtin_fmaf(-R.d, q, n);
    q = __builtin_fmaf(e, R.r1, q);
    e = __builtin_fmaf(-R.d, q, n);
    return __builtin_fmaf(e, R.r1, q);
}
// 1.0f / d in the plain range: q0 = 1 * r1 is exact
BRT_DEV float recip_plain(float d) {
    const RcpRef R = rcp_refined(d);
    float e = __builtin_fmaf(-d, R.r1, 1.0f);
    const float q = __builtin_fmaf(e, R.r1, R.r1);
    e = __builtin_fmaf(-d, q, 1.0f);
    return __builtin_fmaf(e, R.r1, q);
}
// all three components of v have a magnitude in [lo, hi] (a NaN or zero component fails)
BRT_DEV bool all_within(f3 v, float lo, float hi) {
    const float ax = __builtin_fabsf(v.x), ay = __builtin_fabsf(v.y), az = __builtin_fabsf(v.z);
    return min_f(min_f(ax, ay), az) >= lo && max_f(max_f(ax, ay), az) <= hi;
}
BRT_DEV bool wave_all(bool p) { return __ballot(!p) == 0ull; }   // over the lanes that are active at the call
// Number of active lanes for which p holds, as a 32-bit wave-uniform SCALAR (readfirstlane of the already scalar popcount
// folds away and leaves an i32 that the compiler knows is uniform: compares and loop exits on it stay on the scalar
// unit; without it LLVM folds the truncation of the 64-bit popcount into the comparison that follows, and a 64-bit
// compare of a scalar only exists on the vector ALU).
BRT_DEV uint32_t wave_count(bool p) {
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)__popcll(__ballot(p)));
}

// The correctly rounded sqrt likewise: hipcc emits v_sqrt_f32 (1 ulp) and picks the best of {s - 1 ulp, s, s + 1 ulp} by
// the sign of two exact residuals, wrapped in a 2^32 pre-scale for tiny arguments and a class test for 0 / inf / NaN
// (16 instructions).  For x in [2^-80, 2^80] the wrapping does nothing: these 9 instructions are the rest, verbatim.
// (BRT_DBG_SQRT_SWEEP: every float of that range, bit for bit against __builtin_sqrtf.)
BRT_DEV float sqrt_plain(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float s_dn = __uint_as_float(__float_as_uint(s) - 1u), s_up = __uint_as_float(__float_as_uint(s) + 1u);
    const float r_dn = __builtin_fmaf(-s_dn, s, x), r_up = __builtin_fmaf(-s_up, s, x);
    float r = (0.0f >= r_dn) ? s_dn : s;
    r = (0.0f < r_up) ? s_up : r;
    return r;
}

// sqrt of three values at once (the sample colour, raytrace.wgsl:223): the short form when every active lane's
// components are zero (sqrt_plain returns +-0 for +-0: v_sqrt_f32 does, and both residual tests are false) or in the plain range
BRT_DEV f3 sqrt3(f3 c) {
    const float ax = __builtin_fabsf(c.x), ay = __builtin_fabsf(c.y), az = __builtin_fabsf(c.z);
    const float mx = max_f(max_f(ax, ay), az);
    const bool plain = (min_f(min_f(ax, ay), az) >= 0x1p-80f && mx <= 0x1p80f) || mx == 0.0f;
    if (wave_all(plain)) return mk3(sqrt_plain(c.x), sqrt_plain(c.y), sqrt_plain(c.z));
    return mk3(__builtin_sqrtf(c.x), __builtin_sqrtf(c.y), __builtin_sqrtf(c.z));
}

BRT_DEV f3 normalize3(f3 v) {
#if BRT_SHARED_RCP & 1
    // components in [2^-40, 2^39]  =>  dot in [2^-80, 2^80) and len in [max |v_i|, 2 max |v_i|): both plain
    // (products and sums of positive terms are monotone, sqrt is monotone, the sum has 3 terms)
    if (wave_all(all_within(v, kPlainLo, 0x1p39f))) {
        const RcpRef R = rcp_refined(sqrt_plain(dot3(v, v)));
        return mk3(div_plain(v.x, R), div_plain(v.y, R), div_plain(v.z, R));
    }
#endif
    const float len = __builtin_sqrtf(dot3(v, v));
    return mk3(v.x / len, v.y / len, v.z / len);
}

BRT_DEV uint32_t f32_to_u32_sat(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)f;
}

// random.wgsl:8-15
BRT_DEV uint32_t rng_next(uint32_t state) {
    const uint32_t old = state + 747796405u + 2891336453u;
    const uint32_t word = ((old >> ((old >> 28u) + 4u)) ^ old) * 277803737u;
    return (word >> 22u) ^ word;
}
// random.wgsl:3-6; f32(0xffffffffu) == 2^32, and x / 2^32 == x * 2^-32 exactly
BRT_DEV float rng_float(uint32_t& state) {
    state = rng_next(state);
    return (float)state * 2.3283064365386963e-10f;
}
// 2x - 1 of random.wgsl:21-23: 2x is exact in binary floating point, so the single rounding of
// fma(x, 2, -1) is bit for bit the rounding of the shader's (2*x) - 1 -- one instruction instead of two
// without touching the "no contraction" policy (nothing is rounded differently).
BRT_DEV float two_x_minus_one(float x) { return __builtin_fmaf(x, 2.0f, -1.0f); }
// 2 * rngNextFloat() - 1 in one step: f32(state) * 2^-32 and the doubling are both exact scalings, so
// fma(f32(state), 2^-31, -1) rounds exactly once, where the shader's expression rounds.
BRT_DEV float rng_ball_coord(uint32_t& state) {
    state = rng_next(state);
    return __builtin_fmaf((float)state, 4.656612873077392578125e-10f, -1.0f);
}
// random.wgsl:17-30 (a point INSIDE the unit ball, not normalised)
BRT_DEV f3 rng_unit_ball(uint32_t& state) {
    f3 p;
    for (;;) {
        const float x = rng_float(state);
        const float y = rng_float(state);
        const float z = rng_float(state);
        p = mk3(two_x_minus_one(x), two_x_minus_one(y), two_x_minus_one(z));
        if (dot3(p, p) <= 1.0f) break;
    }
    return p;
}

// The rejection sampler of a ROUND by hand (random.wgsl:19-24 + raytrace.wgsl:238 / :285), for the lanes active at the call:
// lanes of m2 need two balls (diffuse: acc = normal + ball, then + roughness * ball), lanes of m1 one (metal: acc = -0 + roughness * ball).
// Every iteration draws a candidate p for each lane that still needs one; the loop itself only KEEPS the accepted points: a lane that is
// done drops out of EXEC, so its last candidate -- the accepted one -- stays in x, y, z, and the first point of a lane that needs two is
// moved aside (three v_mov under EXEC = "two needed -> one").  The sums are made once behind the loop: acc += p1 (the shader's 1.0 * p1
// is p1), acc += roughness * p.  What the compiler's version (shade_landed, COUNTERS builds) spends per iteration beside the three
// draws -- four v_cndmask, a counter with borrow, a compare on it, acc + scale * p -- is here those three moves and the bookkeeping of
// who needs how many as two scalar masks (the scalar unit runs beside the vector pipe: tests/tools/issue_bench.hip): 39 vector
// instructions per iteration instead of 50 (43 with the sums inside the loop: a round runs ~8 iterations, the sums are 9 instructions),
// none of them a 4-cycle select.  Per lane: the same draws and the same mul / add, in the same order.
#ifndef BRT_BALL_ASM
#define BRT_BALL_ASM BRT_HAND_ASM
#endif
BRT_DEV void ball_loop_asm(uint32_t& rng, f3& acc, float rough, uint64_t m2, uint64_t m1, AsmCounts* ac = nullptr) {
#if BRT_HAND_ASM
#if BRT_ASM_COUNT
    uint32_t c_exec, c_lanes, c_tmp;
#define BRT_COUNT_BALL "s_bcnt1_i32_b64 %[c_tmp], exec\n s_add_u32 %[c_lanes], %[c_lanes], %[c_tmp]\n s_add_u32 %[c_exec], %[c_exec], 1\n"
#else
#define BRT_COUNT_BALL
#endif
    uint32_t t;
    float x, y, z, x1, y1, z1, q, r;
    uint64_t s_all, s_up, s_two, s_part;
    const uint32_t c_mul = 277803737u, c_2m31 = 0x30000000u /* 2^-31 */;
#define BRT_RNG_DRAW(dst)                                                                                                   \
    "v_add_u32_e32 %[rng], 0xd8e8c2ba, %[rng]\n"          /* random.wgsl:9: state + 747796405 + 2891336453 */             \
    "v_lshrrev_b32_e32 %[t], 28, %[rng]\n"                                                                                  \
    "v_add_u32_e32 %[t], 4, %[t]\n"                                                                                         \
    "v_lshrrev_b32_e32 %[t], %[t], %[rng]\n"                                                                                \
    "v_xor_b32_e32 %[rng], %[t], %[rng]\n"                                                                                  \
    "v_mul_lo_u32 %[rng], %[rng], %[c_mul]\n"                                                                               \
    "v_lshrrev_b32_e32 %[t], 22, %[rng]\n"                                                                                  \
    "v_xor_b32_e32 %[rng], %[t], %[rng]\n"                                                                                  \
    "v_cvt_f32_u32_e32 " dst ", %[rng]\n"                                                                                  \
    "v_fma_f32 " dst ", " dst ", %[c_2m31], -1.0\n"       /* rng_ball_coord: 2 * (state * 2^-32) - 1, rounded once */
#ifdef BRT_EXP_BALL_FREE       /* experiment (wrong pixels): every candidate is accepted -- what a rejection sampler that costs nothing would buy */
#define BRT_BALL_ACCEPT "v_cmp_ge_f32_e32 vcc, 1.0, %[q]\n s_mov_b64 vcc, exec\n"
#else
#define BRT_BALL_ACCEPT "v_cmp_ge_f32_e32 vcc, 1.0, %[q]\n"                 /* accepted (inactive lanes: 0) */
#endif
#define BRT_BALL_ITERATION                                                                                                  \
        BRT_RNG_DRAW("%[x]")                                                                                                \
        BRT_RNG_DRAW("%[y]")                                                                                                \
        BRT_RNG_DRAW("%[z]")                                                                                                \
        "v_mul_f32_e32 %[q], %[x], %[x]\n"                  /* dot3(p, p) = (x*x + y*y) + z*z */                           \
        "v_mul_f32_e32 %[r], %[y], %[y]\n"                                                                                  \
        "v_add_f32_e32 %[q], %[q], %[r]\n"                                                                                  \
        "v_mul_f32_e32 %[r], %[z], %[z]\n"                                                                                  \
        "v_add_f32_e32 %[q], %[q], %[r]\n"                                                                                  \
        BRT_BALL_ACCEPT                                                                                                     \
        "s_and_b64 %[s_up], %[m2], vcc\n"                   /* two needed -> one */                                        \
        "s_andn2_b64 %[m1], %[m1], vcc\n"                   /* one needed -> none */                                       \
        "s_andn2_b64 %[m2], %[m2], vcc\n"                                                                                   \
        "s_or_b64 %[m1], %[m1], %[s_up]\n"                                                                                  \
        "s_mov_b64 exec, %[s_up]\n"                                                                                         \
        "v_mov_b32_e32 %[x1], %[x]\n"                       /* the first of two points */                                  \
        "v_mov_b32_e32 %[y1], %[y]\n"                                                                                       \
        "v_mov_b32_e32 %[z1], %[z]\n"                                                                                       \
        "s_or_b64 exec, %[m2], %[m1]\n"                     /* SCC: somebody still needs one */
    asm volatile(
#if BRT_ASM_COUNT
        "s_mov_b32 %[c_exec], 0\n s_mov_b32 %[c_lanes], 0\n"
#endif
        "s_mov_b64 %[s_all], exec\n"
        "s_mov_b64 %[s_two], %[m2]\n"
        "s_or_b64 %[s_part], %[m2], %[m1]\n"
        "s_mov_b64 exec, %[s_part]\n"
        "s_cbranch_execz 2f\n"
        "1:\n"
        BRT_COUNT_BALL
        BRT_BALL_ITERATION
        "s_cbranch_scc1 1b\n"                               // (two iterations per trip with a fall-through exit between them: measured, no gain)
        "s_mov_b64 exec, %[s_two]\n"                        // diffuse: acc = normal + 1.0 * p1 ...
        "v_add_f32_e32 %[ax], %[ax], %[x1]\n"
        "v_add_f32_e32 %[ay], %[ay], %[y1]\n"
        "v_add_f32_e32 %[az], %[az], %[z1]\n"
        "s_mov_b64 exec, %[s_part]\n"                       // ... + roughness * p2; metal: -0 + roughness * p
        "v_mul_f32_e32 %[x], %[rough], %[x]\n"
        "v_mul_f32_e32 %[y], %[rough], %[y]\n"
        "v_mul_f32_e32 %[z], %[rough], %[z]\n"
        "v_add_f32_e32 %[ax], %[ax], %[x]\n"
        "v_add_f32_e32 %[ay], %[ay], %[y]\n"
        "v_add_f32_e32 %[az], %[az], %[z]\n"
        "2:\n"
        "s_mov_b64 exec, %[s_all]\n"
        : [rng] "+v"(rng), [ax] "+v"(acc.x), [ay] "+v"(acc.y), [az] "+v"(acc.z), [m2] "+s"(m2), [m1] "+s"(m1),
          [t] "=&v"(t), [x] "=&v"(x), [y] "=&v"(y), [z] "=&v"(z), [x1] "=&v"(x1), [y1] "=&v"(y1), [z1] "=&v"(z1), [q] "=&v"(q), [r] "=&v"(r),
          [s_all] "=&s"(s_all), [s_up] "=&s"(s_up), [s_two] "=&s"(s_two), [s_part] "=&s"(s_part)
#if BRT_ASM_COUNT
          , [c_exec] "=&s"(c_exec), [c_lanes] "=&s"(c_lanes), [c_tmp] "=&s"(c_tmp)
#endif
        : [rough] "v"(rough), [c_mul] "s"(c_mul), [c_2m31] "s"(c_2m31)
        : "vcc", "scc", "memory");
#if BRT_ASM_COUNT
    if (first_active_lane()) { ac->ball_exec += c_exec; ac->ball_lanes += c_lanes; }
#endif
#undef BRT_COUNT_BALL
#undef BRT_BALL_ITERATION
#undef BRT_BALL_ACCEPT
#undef BRT_RNG_DRAW
#endif
}

// ---- the rejection sampler as a STAGE of the workgroup (k_trace_persistent<.., SRV>) ------------------------------
// In the wave-level loop above a round's ~35 hit lanes wait for the unluckiest of them: 7.7 iterations at 14.6 of 64 lanes, 13 % of the
// headline frame (sampler made free: 8.90 -> 7.74 ms, profiles/r06/sampler_stage.txt).  The sampler's state is the smallest of all stages --
// {rng, points needed} in, {rng, p1, p2} out -- so here it leaves the path's lane: of the workgroup's sixteen waves fourteen trace and
// SRV_WAVES = 2 SERVE.  A trace wave posts its round's requests into its mailbox in LDS (32 bytes per lane: {rng, need | serial << 2}
// compacted by mbcnt, then its door word {serial, count}), goes on with the part of the round's shading that does not need the points
// (sky, sample ends, camera rays, the second normalize, the glass branch) and picks the results up behind it: {rng', p1} {p2, serial} in
// the request's own entry, valid when the last word is the round's serial.  A server wave keeps 64 requests in flight: every iteration
// draws a candidate for each of its lanes (the iteration of ball_loop_asm, on fixed registers), the lanes that are done write their
// entry back and free lanes take the next requests of the server's current batch (a trace wave's door it has not served yet).  A lane's
// draws are the same hash chain whichever wave runs them: pixels cannot change.  A server owns every other trace wave (static: no shared
// queue, no atomics); it ends when the workgroup's last trace wave has ended (control word 0).  Nobody can wait for ever: the requests
// of a posted batch are taken whenever a server lane is free, a request is done with probability 1, and both loops give up after a
// bounded number of polls (a wrong frame instead of a hang; the give-ups are counted).
//   mbox: LDS byte address of wave 0's mailbox (2 KB per trace wave); door: of door[0]; ctl: control words {alive trace waves, ..}
//   first, stride, n_mine: this server serves trace waves first, first + stride, .. (n_mine of them)
#ifndef BRT_SRV_LOW
#define BRT_SRV_LOW 0
#endif
#ifndef BRT_SRV_PATIENCE
#define BRT_SRV_PATIENCE 8
#endif
#ifndef BRT_SRV_PRIO
#define BRT_SRV_PRIO "s_setprio 3\n"
#endif
BRT_DEV void ball_server_asm(uint32_t mbox, uint32_t door, uint32_t ctl, uint32_t first, uint32_t stride, uint32_t n_mine, uint32_t* counts8) {
    const uint32_t low = BRT_SRV_LOW, patience = BRT_SRV_PATIENCE;
#if BRT_HAND_ASM
    uint32_t pos, cnt, base, i0, d0, nfree, take, idle, c_iter, c_lanes, c_polls, c_tmp, polled, stall;
    uint64_t m2, m1, busy, s_up, s_mine, s_tmp, s_take;
    const uint32_t c_mul = 277803737u, c_2m31 = 0x30000000u /* 2^-31 */;
#define BRT_SRV_DRAW(dst)                                                                                                   \
    "v_add_u32_e32 v100, 0xd8e8c2ba, v100\n"              /* random.wgsl:9 */                                              \
    "v_lshrrev_b32_e32 v108, 28, v100\n"                                                                                    \
    "v_add_u32_e32 v108, 4, v108\n"                                                                                         \
    "v_lshrrev_b32_e32 v108, v108, v100\n"                                                                                  \
    "v_xor_b32_e32 v100, v108, v100\n"                                                                                      \
    "v_mul_lo_u32 v100, v100, %[c_mul]\n"                                                                                   \
    "v_lshrrev_b32_e32 v108, 22, v100\n"                                                                                    \
    "v_xor_b32_e32 v100, v108, v100\n"                                                                                      \
    "v_cvt_f32_u32_e32 " dst ", v100\n"                                                                                    \
    "v_fma_f32 " dst ", " dst ", %[c_2m31], -1.0\n"
    asm volatile(
        BRT_SRV_PRIO
        "s_mov_b64 %[m2], 0\n s_mov_b64 %[m1], 0\n s_mov_b32 %[pos], 0\n s_mov_b32 %[cnt], 0\n s_mov_b32 %[idle], 0\n"
        "s_mov_b32 %[c_iter], 0\n s_mov_b32 %[c_lanes], 0\n s_mov_b32 %[c_polls], 0\n s_mov_b32 %[base], 0\n s_mov_b32 %[stall], 0\n"
        "v_mbcnt_lo_u32_b32 v112, -1, 0\n"
        "v_mbcnt_hi_u32_b32 v112, -1, v112\n"                   // lane
        "v_cmp_gt_u32_e32 vcc, %[n_mine], v112\n"               // lanes that watch a door
        "s_mov_b64 %[s_mine], vcc\n"
        "v_mul_lo_u32 v109, v112, %[stride]\n"
        "v_add_u32_e32 v109, %[first], v109\n"                  // the trace wave this lane watches
        "v_lshl_add_u32 v109, v109, 2, %[door]\n"               // ... its door word
        "v_mov_b32_e32 v110, 0\n"                               // the door value this lane has served
        "v_mov_b32_e32 v111, 0\n"
        // ---- a pass: (A) the LDS reads of the refill go out -- the next requests of the current batch for the free lanes, or, with no batch
        //      at hand, the doors --, (B) one candidate for every lane that needs one (ball_loop_asm's iteration) runs under their round trip,
        //      lanes that are done write their entry back, (C) the reads are taken in: the takers start, or a new batch is opened.
        "1:\n"
        "s_or_b64 %[busy], %[m2], %[m1]\n"
        "s_mov_b64 %[s_take], 0\n"
        "s_mov_b32 %[polled], 0\n"
        "s_cmp_lt_u32 %[pos], %[cnt]\n"
        "s_cbranch_scc0 2f\n"
        "s_not_b64 %[s_tmp], %[busy]\n"                         // (A) free lanes take requests pos .. of the batch
        "s_bcnt1_i32_b64 %[nfree], %[s_tmp]\n"
        "s_cbranch_scc0 4f\n"
        "s_sub_u32 %[take], %[cnt], %[pos]\n"
        "s_min_u32 %[take], %[take], %[nfree]\n"
        "s_mov_b64 exec, %[s_tmp]\n"
        "v_mbcnt_lo_u32_b32 v112, exec_lo, 0\n"
        "v_mbcnt_hi_u32_b32 v112, exec_hi, v112\n"              // rank among the free lanes
        "v_cmp_gt_u32_e32 vcc, %[take], v112\n"
        "s_mov_b64 %[s_take], vcc\n"
        "s_mov_b64 exec, vcc\n"
        "v_add_u32_e32 v112, %[pos], v112\n"
        "v_lshl_add_u32 v111, v112, 5, %[base]\n"               // the request's entry
        "ds_read_b64 v[114:115], v111\n"                        // { rng, need | serial << 2 }
        "s_add_u32 %[pos], %[pos], %[take]\n"
        "s_branch 4f\n"
        "2:\n"                                                  // (A) no batch at hand: the doors (only worth a look with room for a batch)
        "s_not_b64 %[s_tmp], %[busy]\n"
        "s_bcnt1_i32_b64 %[nfree], %[s_tmp]\n"
        "s_cmp_lt_u32 %[nfree], 12\n"
        "s_cbranch_scc1 4f\n"
        "s_mov_b64 exec, %[s_mine]\n"
        "ds_read_b32 v112, v109\n"
        "s_mov_b32 %[polled], 1\n"
        "s_add_u32 %[c_polls], %[c_polls], 1\n"
        // ---- (B) ------------------------------------------------------------------------------------------------------------------------
        "4:\n"
        "s_mov_b64 exec, %[busy]\n"
        "s_cbranch_execz 6f\n"
        // (a server that iterates with a few lanes costs its SIMD what one with 64 does: below %[low] lanes it lets requests gather for up to
        //  %[patience] passes -- each a look at the doors and a short sleep -- before it goes on with what it has)
        "s_bcnt1_i32_b64 %[c_tmp], exec\n"
        "s_cmp_ge_u32 %[c_tmp], %[low]\n"
        "s_cbranch_scc1 5f\n"
        "s_add_u32 %[stall], %[stall], 1\n"
        "s_cmp_ge_u32 %[stall], %[patience]\n"
        "s_cbranch_scc1 5f\n"
        "s_sleep 1\n"
        "s_branch 6f\n"
        "5:\n"
        "s_mov_b32 %[stall], 0\n"
        "s_add_u32 %[c_lanes], %[c_lanes], %[c_tmp]\n s_add_u32 %[c_iter], %[c_iter], 1\n"
        BRT_SRV_DRAW("v104")
        BRT_SRV_DRAW("v105")
        BRT_SRV_DRAW("v106")
        "v_mul_f32_e32 v108, v104, v104\n"                      // dot3(p, p) = (x*x + y*y) + z*z
        "v_mul_f32_e32 v113, v105, v105\n"
        "v_add_f32_e32 v108, v108, v113\n"
        "v_mul_f32_e32 v113, v106, v106\n"
        "v_add_f32_e32 v108, v108, v113\n"
        "v_cmp_ge_f32_e32 vcc, 1.0, v108\n"                     // accepted
        "s_and_b64 %[s_up], %[m2], vcc\n"                       // two needed -> one
        "s_andn2_b64 %[m1], %[m1], vcc\n"                       // one needed -> none
        "s_andn2_b64 %[m2], %[m2], vcc\n"
        "s_or_b64 %[m1], %[m1], %[s_up]\n"
        "s_mov_b64 exec, %[s_up]\n"
        "v_mov_b32_e32 v101, v104\n"                            // the first of two points
        "v_mov_b32_e32 v102, v105\n"
        "v_mov_b32_e32 v103, v106\n"
        "s_or_b64 %[s_tmp], %[m2], %[m1]\n"
        "s_andn2_b64 exec, %[busy], %[s_tmp]\n"                 // done: the entry goes back {rng', p1} {p2, serial}
        "s_cbranch_execz 6f\n"
        "ds_write_b128 v111, v[100:103]\n"
        "ds_write_b128 v111, v[104:107] offset:16\n"
        // ---- (C) ------------------------------------------------------------------------------------------------------------------------
        "6:\n"
        "s_mov_b64 exec, %[s_take]\n"
        "s_cbranch_execz 7f\n"
        "s_waitcnt lgkmcnt(0)\n"                                // (the takers' requests; any write-back above is behind them in the queue: waited for too)
        "v_mov_b32_e32 v100, v114\n"
        "v_and_b32_e32 v112, 3, v115\n"
        "v_lshrrev_b32_e32 v107, 2, v115\n"                     // the serial goes back with the result
        "v_cmp_eq_u32_e32 vcc, 2, v112\n"
        "s_or_b64 %[m2], %[m2], vcc\n"
        "v_cmp_eq_u32_e32 vcc, 1, v112\n"
        "s_or_b64 %[m1], %[m1], vcc\n"
        "s_mov_b64 exec, -1\n"
        "s_branch 1b\n"
        "7:\n"
        "s_mov_b64 exec, -1\n"
        "s_cmp_eq_u32 %[polled], 0\n"
        "s_cbranch_scc1 1b\n"
        "s_waitcnt lgkmcnt(0)\n"                                // the doors
        "s_mov_b64 exec, %[s_mine]\n"
        "v_cmp_ne_u32_e32 vcc, v112, v110\n"                    // a value this lane has not served
        "s_mov_b64 exec, -1\n"
        "s_cbranch_vccz 8f\n"
        "s_ff1_i32_b64 %[i0], vcc\n"                            // the first door with a new batch
        "s_lshl_b64 %[s_tmp], 1, %[i0]\n"
        "s_mov_b64 exec, %[s_tmp]\n"
        "v_mov_b32_e32 v110, v112\n"                            // served (from now on)
        "s_mov_b64 exec, -1\n"
        "v_readlane_b32 %[d0], v112, %[i0]\n"
        "s_and_b32 %[cnt], %[d0], 0xff\n"                       // requests of the batch
        "s_mov_b32 %[pos], 0\n"
        "s_mul_i32 %[base], %[i0], %[stride]\n"
        "s_add_u32 %[base], %[base], %[first]\n"
        "s_lshl_b32 %[base], %[base], 11\n"                     // 2 KB per trace wave
        "s_add_u32 %[base], %[base], %[mbox]\n"
        "s_mov_b32 %[idle], 0\n"
        "s_branch 1b\n"
        "8:\n"                                                  // no new batch
        "s_or_b64 %[busy], %[m2], %[m1]\n"
        "s_cmp_lg_u64 %[busy], 0\n"
        "s_cbranch_scc1 1b\n"
        "v_mov_b32_e32 v112, %[ctl]\n"
        "ds_read_b32 v112, v112\n"                              // nothing in flight: has the last trace wave ended?
        "s_waitcnt lgkmcnt(0)\n"
        "v_readfirstlane_b32 %[d0], v112\n"
        "s_cmp_eq_u32 %[d0], 0\n"
        "s_cbranch_scc1 9f\n"
        "s_sleep 1\n"
        "s_add_u32 %[idle], %[idle], 1\n"
        "s_cmp_gt_u32 %[idle], 0x1000000\n"                     // (a bound, not a protocol step)
        "s_cbranch_scc1 9f\n"
        "s_branch 1b\n"
        "9:\n"
        "s_mov_b64 exec, -1\n"
        "s_waitcnt lgkmcnt(0)\n"
        "s_setprio 0\n"
        : [pos] "=&s"(pos), [cnt] "=&s"(cnt), [base] "=&s"(base), [i0] "=&s"(i0), [d0] "=&s"(d0), [nfree] "=&s"(nfree), [take] "=&s"(take),
          [idle] "=&s"(idle), [c_iter] "=&s"(c_iter), [c_lanes] "=&s"(c_lanes), [c_polls] "=&s"(c_polls), [c_tmp] "=&s"(c_tmp), [m2] "=&s"(m2),
          [m1] "=&s"(m1), [busy] "=&s"(busy), [s_up] "=&s"(s_up), [s_mine] "=&s"(s_mine), [s_tmp] "=&s"(s_tmp), [s_take] "=&s"(s_take), [polled] "=&s"(polled), [stall] "=&s"(stall)
        : [low] "s"(low), [patience] "s"(patience), [mbox] "s"(mbox), [door] "s"(door), [ctl] "s"(ctl), [first] "s"(first), [stride] "s"(stride), [n_mine] "s"(n_mine), [c_mul] "s"(c_mul),
          [c_2m31] "s"(c_2m31)
        : "vcc", "scc", "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115");
    counts8[0] = c_iter; counts8[1] = c_lanes; counts8[2] = c_polls; counts8[3] = idle;
#undef BRT_SRV_DRAW
#endif
}

// raytrace.wgsl:387-398 with 1/d hoisted per ray.  Returns whether the child is pushed
// (raytrace.wgsl:331,338: dst != INF && dst < closest.distance).
BRT_DEV bool slab_push(f3 o, f3 inv, f3 bmin, f3 bmax, float closest) {
    const float tminx = (bmin.x - o.x) * inv.x, tmaxx = (bmax.x - o.x) * inv.x;
    const float tminy = (bmin.y - o.y) * inv.y, tmaxy = (bmax.y - o.y) * inv.y;
    const float tminz = (bmin.z - o.z) * inv.z, tmaxz = (bmax.z - o.z) * inv.z;
    const float t_near = max_f(max_f(min_f(tminx, tmaxx), min_f(tminy, tmaxy)), min_f(tminz, tmaxz));
    const float t_far = min_f(min_f(max_f(tminx, tmaxx), max_f(tminy, tmaxy)), max_f(tminz, tmaxz));
    const bool hit = (t_far >= t_near) && (t_far > 0.0f);
    // dst = hit ? (t_near > 0 ? t_near : 0) : INF;  pushed iff dst != INF && dst < closest.
    // closest <= INF always, so `dst < closest` already implies `dst != INF`.  closest is INF or an
    // accepted t > 0.001 (sphere_test), i.e. always > 0, and for a hit t_near is not NaN, so
    // `max(t_near, 0) < closest` == `t_near < closest`: the clamp never changes the outcome.
    return hit && (t_near < closest);
}

// raytrace.wgsl:371-383 + the accept test of :353-354.  `a` = dot(d,d) hoisted per ray,
// s.w = radius*radius.  Branch-free: discriminant < 0 gives sqrt -> NaN -> t NaN -> rejected
// by `t > 0.001`, as the shader's -1.0 is (and t != -1.0 is implied by t > 0.001).
BRT_DEV void sphere_test(f3 o, f3 d, float a, float4 s, uint32_t idx, float& closest, uint32_t& closest_idx) {
    const f3 oc = mk3(s.x - o.x, s.y - o.y, s.z - o.z);
    const float h = dot3(d, oc);
    const float c = dot3(oc, oc) - s.w;
    const float disc = h * h - a * c;
    const float t = (h - __builtin_sqrtf(disc)) / a;
    const bool accept = (t > 0.001f) && (t < closest);
#if BRT_EXEC_MOVES & 1
    // two v_mov under EXEC instead of two v_cndmask (the empty asm keeps the compiler from turning the branch back into
    // selects: on gfx950 a v_cndmask costs a SIMD 4 cycles, a v_mov 2 -- tests/tools/issue_bench.hip)
    if (accept) {
        closest = t;
        closest_idx = idx;
        asm volatile("" : "+v"(closest), "+v"(closest_idx));
    }
#else
    closest = accept ? t : closest;
    closest_idx = accept ? idx : closest_idx;
#endif
}

// closest is INF (FLT_MAX) or an accepted t > 0.001: a positive normal float, whose predecessor is its
// bit pattern minus one
BRT_DEV float float_below(float closest) { return __uint_as_float(__float_as_uint(closest) - 1u); }

// Scene accessors.  The persistent kernel instantiates with LDS pointers, the bring-up
// kernel and the large-scene variant with global pointers.
struct ScenePtrs {
    const char* pairs;       // pair records of PAIR_BYTES (brt_layout.h): near/far planes by granule, descriptors
    // SCENE_LDS_TOP: `pairs` is the LDS tile holding the records below byte offset `near_bytes` (the top of the
    // tree in breadth-first order), `pairs_far` the whole array in global memory
    const char* pairs_far;
    uint32_t near_bytes;
    uint32_t near_base;      // LDS byte address of the tile (SCENE_LDS: of the pair records)
    uint32_t sph_base;       // SCENE_LDS: LDS byte address of the spheres
    uint32_t rows_scratch;   // SCENE_LDS: LDS byte address of THIS WAVE's scratch for the row-mode walk (walk_rows_asm), 0: none
    uint32_t srv_ctl;              // sampler stage (SRV): LDS byte address of its control words
    uint32_t srv_mbox, srv_door;   // sampler stage (SRV): LDS byte addresses of THIS trace wave's mailbox (64 entries of 32 bytes) and door word
    bool boxes_ordered;      // every child box finite with min <= max (decided at upload)
    const float4* spheres;
    const uint32_t* sphere_material;
    const float4* materials;
    const float4* sphere_mats;     // the material of every sphere, two float4 each
    const uint2* leaf_table;
    bool minmax_select;      // BRT_POLICY_MINMAX_SELECT (knobs-live instantiation only): min / max by compare-select, in the shader's operand order
    uint32_t* hits;          // pre-pass of a SCENE_LDS_TOP scene: interior visits per pair record, a histogram in LDS (else null)
};

struct HitCounters {
    uint32_t node_pops, interior, sphere_tests, hits;
    // COUNTERS builds only: per code section, how often the wave executed it and with how many
    // lanes (lane-utilisation profile; read by brt_debug_profile)
    uint32_t sec_exec[8], sec_lanes[8];
    unsigned long long ticks_ball;   // COUNTERS builds: wave time in the rejection-sampler loop (100 MHz ticks)
    AsmCounts asm_counts;            // -DBRT_ASM_COUNT builds: what the hand-written loops executed (else untouched)
};
enum { SEC_INTERIOR = 0, SEC_LEAF, SEC_CAMERA, SEC_SCATTER, SEC_SKY, SEC_BALL, SEC_CAMERA_TOP, SEC_ROUND };   // SEC_CAMERA: camera rays made in shade_landed; _TOP: at the top of a round

// Counts one execution of a code section and its active lanes.  May be called under divergent
// control flow: the ballot only sees the lanes that reached the call; the first of them books it.
template <bool COUNTERS>
BRT_DEV void prof_section(HitCounters& hc, int sec, bool pred) {
    if (COUNTERS) {
        const uint64_t m = __ballot(pred);
        if (pred && __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) == 0u) {
            hc.sec_exec[sec]++;
            hc.sec_lanes[sec] += (uint32_t)__popcll(m);
        }
    }
}

// raytrace.wgsl:313-362: closest hit of one ray.  `stk` points at this lane's column of a
// [entries][STRIDE] array (LDS: 16-bit entries when descriptors are 16-bit; bring-up kernel:
// a private array, STRIDE 1).
// Reference bookkeeping: stack_index == (entries in stk) + (cur valid ? 1 : 0); the loop
// condition stack_index > 0 && stack_index < 32 (raytrace.wgsl:320) is `cur != DONE && n < 31`.
//
// The kernel is bound by TOTAL instruction issue (VALU + scalar + branches), and a
// divergent if/else nest costs scalar exec-mask bookkeeping on every level.  So both bodies
// are straight-line: the child selection, push and pop of raytrace.wgsl:331-341 become
// selects, the push is an unconditional LDS store (above the top when nothing is pushed, where
// it is dead), the pop an unconditional LDS load.  hit_sphere's `discriminant < 0 -> -1`
// (raytrace.wgsl:378-380) needs no branch either: sqrt of a negative is NaN and a NaN t
// fails `t > 0.001`, exactly like -1 does.
// SIMPLE_TREE (decided at upload): every leaf holds one sphere and the tree is shallower than
// 31 levels, so neither the leaf table nor the stack-overflow rule can come into play and
// both checks are compiled out.
//
// The walk is SUSPENDABLE: its state (WalkState) survives walk_run returning early.  With
// `exit_lanes` > 0 a wave stops iterating as soon as no more than min(exit_lanes, half of the
// lanes that entered) are still walking; the finished lanes are shaded and given their next ray
// by the caller, and the stragglers continue next to those new rays in the next call instead of
// keeping the whole wave in the loop with a handful of live lanes.  Per lane nothing changes:
// the same steps in the same order.
template <typename StackT>
struct WalkState {
    float a;                 // dot(d, d)
    f3 inv;                  // 1 / d
    float closest;
    uint32_t closest_idx;
    uint32_t cur;            // DONE when the walk has ended
    StackT* sp;
    uint32_t n;              // entries in use (overflow rule of general trees only)
    // byte offset, inside a pair record, of the granule this ray's direction selects on each axis
    // (brt_layout.h): G0 reads {min, max} = {near, far} for a direction >= 0, G1 {max, min} for a direction < 0
    uint32_t ox, oy, oz;
};

template <bool D16, typename StackT>
BRT_DEV void walk_begin(WalkState<StackT>& w, const ScenePtrs& sc, uint32_t root_desc, StackT* stk, f3 d) {
    w.a = dot3(d, d);
#if BRT_SHARED_RCP & 2
    if (wave_all(all_within(d, kPlainLo, kPlainHi))) w.inv = mk3(recip_plain(d.x), recip_plain(d.y), recip_plain(d.z));
    else
#endif
        w.inv = mk3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    // (under the compare-select policy the walk applies min / max itself, to {t_min, t_max} in the shader's operand order: granule G0)
    w.ox = PAIR_X + ((w.inv.x < 0.0f && !sc.minmax_select) ? 16u : 0u);
    w.oy = PAIR_Y + ((w.inv.y < 0.0f && !sc.minmax_select) ? 16u : 0u);
    w.oz = PAIR_Z + ((w.inv.z < 0.0f && !sc.minmax_select) ? 16u : 0u);
    w.closest = kInf;
    w.closest_idx = 0xffffffffu;
    w.cur = root_desc;
    // Stack convention: entry 0 holds DONE for ever, pushed nodes live in entries 1..n, `sp` points at
    // entry n.  A pop is then `cur = *sp; sp -= STRIDE` with no emptiness test (the empty stack pops
    // DONE and the walk ends), a push is a store to sp[STRIDE]; the address is carried instead of n.
    stk[0] = (StackT)-1;                   // Desc::DONE in stack form
    w.sp = stk;
    w.n = 0;
}

// true while this lane's walk has not ended (general trees: also ended by the overflow rule)
template <bool D16, bool SIMPLE_TREE, typename StackT>
BRT_DEV bool walk_pending(const WalkState<StackT>& w) {
    return w.cur != Desc<D16>::DONE && (SIMPLE_TREE || w.n < 31u);
}

// Leaf step of the walk (raytrace.wgsl:325-326, 348-362) for a lane whose `cur` is a leaf: test its
// sphere(s), then pop.
template <int STRIDE, bool COUNTERS, bool D16, bool SIMPLE_TREE, typename StackT>
BRT_DEV void walk_leaf_step(const ScenePtrs& sc, f3 o, f3 d, float a, float& closest, uint32_t& closest_idx, uint32_t& cur,
                            StackT*& sp, uint32_t& n, HitCounters& hc) {
    using DS = Desc<D16>;
    if (COUNTERS) hc.node_pops++;
    if (BRT_ASM_COUNT && !COUNTERS && STRIDE == 64) hc.asm_counts.fix_leaf_lanes++;      // (single-sphere leaves: one test per step)
    if (STRIDE == 64) prof_section<COUNTERS>(hc, SEC_LEAF, true);
    const uint32_t first = cur & DS::INDEX_MASK;
    const uint32_t popped = (uint32_t)(int32_t)*sp;  // issued before the sphere arithmetic (sign-extending load)
    if (SIMPLE_TREE || (cur & DS::LEAF1)) {           // one sphere (what PLOC produces)
        if (COUNTERS) hc.sphere_tests++;
        sphere_test(o, d, a, sc.spheres[first], first, closest, closest_idx);
    } else {                                          // general leaf: {first, count} from the leaf table
        const uint2 lt = sc.leaf_table[first];
        for (uint32_t i = lt.x; i < lt.x + lt.y; i++) {
            if (COUNTERS) hc.sphere_tests++;
            sphere_test(o, d, a, sc.spheres[i], i, closest, closest_idx);
        }
    }
    cur = popped;
    sp -= STRIDE;                                     // below entry 0 only after DONE was popped: never used again
    n--;                                              // wraps with it; only read while cur != DONE
}

// A ray is "safe" for the near/far read offsets (brt_layout.h) when its origin is finite and every
// component of 1/direction is finite and non-zero: then, for a finite box with min <= max,
// (b - o) * inv is monotone in b and never NaN, so the value read as "near" IS min(t_min, t_max) and
// the value read as "far" IS max(t_min, t_max) of raytrace.wgsl:391-392 (up to the sign of a zero,
// which no comparison below can see).
BRT_DEV bool ray_is_safe(f3 o, f3 inv) {
    constexpr int kFinite = 0x1F8;          // -normal, -denormal, -0, +0, +denormal, +normal
    constexpr int kFiniteNonZero = 0x198;   // -normal, -denormal, +denormal, +normal
    return __builtin_amdgcn_classf(o.x, kFinite) && __builtin_amdgcn_classf(o.y, kFinite) &&
           __builtin_amdgcn_classf(o.z, kFinite) && __builtin_amdgcn_classf(inv.x, kFiniteNonZero) &&
           __builtin_amdgcn_classf(inv.y, kFiniteNonZero) && __builtin_amdgcn_classf(inv.z, kFiniteNonZero);
}

// Interior step (raytrace.wgsl:327-342, 387-398) for a lane whose `cur` is a pair record: both slab
// tests, push/pop as selects.  FIX: apply min/max to the {near, far} values read (needed when some ray
// of the wave is not safe or the boxes are not ordered); without it the read offset has already made
// that choice.
template <int STRIDE, bool COUNTERS, bool FIX, bool D16, int MODE, typename StackT, bool HITS = false, bool SEL = false>
BRT_DEV void walk_interior_step(const ScenePtrs& sc, f3 o, f3 inv, uint32_t ox, uint32_t oy, uint32_t oz, float below,
                                uint32_t& cur, StackT*& sp, uint32_t& n, HitCounters& hc) {
    if (COUNTERS) { hc.node_pops++; hc.interior++; }
    if (BRT_ASM_COUNT && !COUNTERS && STRIDE == 64) hc.asm_counts.fix_int_lanes++;
    if (HITS) atomicAdd(sc.hits + cur, 1u);              // (16-bit descriptors: an interior descriptor is the record's index)
    if (STRIDE == 64) prof_section<COUNTERS>(hc, SEC_INTERIOR, true);
    const uint32_t ro = Desc<D16>::record_offset(cur);   // byte offset of the pair record
    float4 gx, gy, gz;                                   // per axis { near L, near R, far L, far R }
    uint2 D;
    if (MODE == SCENE_LDS_TOP) {
        // Lanes of one wave take either side.  The two sides use pointers of DIFFERENT address spaces on purpose:
        // with generic pointers the optimiser merges the branches into one set of flat_load instructions, which
        // send every lane through the texture addresser (measured on the 10k-sphere scene: TA busy 74 % with or
        // without the tile); ds_read_b128 for the lanes in the tile leaves the vector-memory path to the others.
        typedef float vf4 __attribute__((ext_vector_type(4)));
        typedef uint32_t vu2 __attribute__((ext_vector_type(2)));
        typedef const __attribute__((address_space(3))) vf4 lds_f4;
        typedef const __attribute__((address_space(3))) vu2 lds_u2;
        typedef const __attribute__((address_space(1))) vf4 glb_f4;
        typedef const __attribute__((address_space(1))) vu2 glb_u2;
        vf4 vx, vy, vz;
        vu2 vd;
        if (ro < sc.near_bytes) {
            const uint32_t rec = sc.near_base + ro;      // LDS byte address of the record
            vx = *reinterpret_cast<lds_f4*>((uintptr_t)(rec + ox));
            vy = *reinterpret_cast<lds_f4*>((uintptr_t)(rec + oy));
            vz = *reinterpret_cast<lds_f4*>((uintptr_t)(rec + oz));
            vd = *reinterpret_cast<lds_u2*>((uintptr_t)(rec + PAIR_DESC));
        } else {
            const char* rec = sc.pairs_far + ro;
            vx = *(glb_f4*)(rec + ox);
            vy = *(glb_f4*)(rec + oy);
            vz = *(glb_f4*)(rec + oz);
            vd = *(glb_u2*)(rec + PAIR_DESC);
        }
        gx = make_float4(vx.x, vx.y, vx.z, vx.w);
        gy = make_float4(vy.x, vy.y, vy.z, vy.w);
        gz = make_float4(vz.x, vz.y, vz.z, vz.w);
        D = make_uint2(vd.x, vd.y);
    } else {
        const char* rec = sc.pairs + ro;
        gx = *reinterpret_cast<const float4*>(rec + ox);
        gy = *reinterpret_cast<const float4*>(rec + oy);
        gz = *reinterpret_cast<const float4*>(rec + oz);
        D = *reinterpret_cast<const uint2*>(rec + PAIR_DESC);
    }
    // the would-be pop: its LDS latency hides behind the slab arithmetic; the store below goes to the entry
    // above it, never to it
    const uint32_t popped = (uint32_t)(int32_t)*sp;
    // .x = child L (`index`), .y = child R (`index + 1`); (b - o) * (1/d) as in raytrace.wgsl:388-390
    float nLx = (gx.x - o.x) * inv.x, nRx = (gx.y - o.x) * inv.x, fLx = (gx.z - o.x) * inv.x, fRx = (gx.w - o.x) * inv.x;
    float nLy = (gy.x - o.y) * inv.y, nRy = (gy.y - o.y) * inv.y, fLy = (gy.z - o.y) * inv.y, fRy = (gy.w - o.y) * inv.y;
    float nLz = (gz.x - o.z) * inv.z, nRz = (gz.y - o.z) * inv.z, fLz = (gz.z - o.z) * inv.z, fRz = (gz.w - o.z) * inv.z;
    // (SEL: the values read are {t_min, t_max} themselves -- granule G0, walk_begin -- and min / max are the compare-select forms in the
    //  shader's operand order, raytrace.wgsl:391-394)
    auto MIN = [](float a, float b) { return SEL ? min_sel(a, b) : min_f(a, b); };
    auto MAX = [](float a, float b) { return SEL ? max_sel(a, b) : max_f(a, b); };
    if (FIX) {
        float t;
        t = MIN(nLx, fLx); fLx = MAX(nLx, fLx); nLx = t;
        t = MIN(nRx, fRx); fRx = MAX(nRx, fRx); nRx = t;
        t = MIN(nLy, fLy); fLy = MAX(nLy, fLy); nLy = t;
        t = MIN(nRy, fRy); fRy = MAX(nRy, fRy); nRy = t;
        t = MIN(nLz, fLz); fLz = MAX(nLz, fLz); nLz = t;
        t = MIN(nRz, fRz); fRz = MAX(nRz, fRz); nRz = t;
    }
    const float tnL = MAX(MAX(nLx, nLy), nLz), tfL = MIN(MIN(fLx, fLy), fLz);   // :393-394
    const float tnR = MAX(MAX(nRx, nRy), nRz), tfR = MIN(MIN(fRx, fRy), fRz);
    // pushed iff hit && dst < closest (raytrace.wgsl:331,338); see slab_push for why t_near serves as dst.
    // `below` is the largest float under closest, so t_near < closest == t_near <= below.  Without NaNs
    // (FIX off) the three compares fold into one: t_far > 0 == t_far >= the smallest denormal, hence
    //   t_far >= t_near && t_far > 0 && t_near <= below   ==   max(t_near, denorm_min) <= min(t_far, below)
    // (below >= 0.001 > denorm_min always; f32 denormals are not flushed in this build).
    bool p1, p2;
    if (FIX) {
        p1 = (tfL >= tnL) && (tfL > 0.0f) && (tnL <= below);
        p2 = (tfR >= tnR) && (tfR > 0.0f) && (tnR <= below);
    } else {
        const float dmin = __uint_as_float(1u);
        p1 = max_f(tnL, dmin) <= min_f(tfL, below);
        p2 = max_f(tnR, dmin) <= min_f(tfR, below);
    }
    // reference: push `index` (D.x) then `index+1` (D.y); the later push is popped first
    const bool both = p1 && p2, none = !p1 && !p2;
    // stored even when it is not pushed: the entry above the top is dead, and it exists (an interior node
    // of depth k sees at most k entries below it; brt_host.cpp sizes the stack by the deepest LEAF)
    sp[STRIDE] = (StackT)D.x;
    cur = p2 ? D.y : (p1 ? D.x : popped);
    const int step = both ? 1 : (none ? -1 : 0);
    sp += step * STRIDE;
    n += (uint32_t)step;
}

// The wave-level walk loop (see walk_run).
template <bool COUNTERS, bool D16, bool SIMPLE_TREE, bool FIX, int MODE, typename StackT, bool HITS = false, bool SEL = false>
BRT_DEV void walk_loop_wave(const ScenePtrs& sc, f3 o, f3 d, float a, f3 inv, uint32_t ox, uint32_t oy, uint32_t oz,
                            float& closest, uint32_t& closest_idx, uint32_t& cur, StackT*& sp, uint32_t& n,
                            uint32_t exit_at, uint32_t vote, HitCounters& hc) {
    using DS = Desc<D16>;
    float below = float_below(closest);
    for (;;) {
        for (;;) {
            const bool interior = DS::is_interior(cur) && (SIMPLE_TREE || n < 31u);
            if (__ballot(interior) == 0ull) break;
            if (interior) walk_interior_step<64, COUNTERS, FIX, D16, MODE, StackT, HITS, SEL>(sc, o, inv, ox, oy, oz, below, cur, sp, n, hc);
            if (wave_count(DS::is_leaf(cur) && (SIMPLE_TREE || n < 31u)) >= vote) break;
        }
        if (DS::is_leaf(cur) && (SIMPLE_TREE || n < 31u)) {
            walk_leaf_step<64, COUNTERS, D16, SIMPLE_TREE>(sc, o, d, a, closest, closest_idx, cur, sp, n, hc);
            below = float_below(closest);
        }
        if (wave_count(cur != DS::DONE && (SIMPLE_TREE || n < 31u)) <= exit_at) break;
    }
}

// ---- the walk loop of an LDS-resident simple tree, written against the MEASURED issue costs of gfx950 --------------------
// tests/tools/issue_bench.hip (profiles/r03/issue_bench.txt), cycles of a SIMD per wave64 instruction with >= 2 waves resident:
//   2   v_add/sub/mul/fma/fmac_f32, v_mov, v_and/xor, v_add_u32, v_lshrrev (the two VALU decoders co-execute these)
//   4   v_min/max (f32, i32, u32), v_min3/max3/med3, every v_cmp, v_cndmask, v_cvt, v_mul_lo_u32, v_mul/mad_u32_u24,
//       v_lshl_add, v_add3, v_bfe, v_and_or, v_div_scale/fmas/fixup        8   v_rcp, v_sqrt
//   SALU: 4 on the scalar unit, hidden while that unit has spare time; s_cbranch not taken ~5, taken ~12 -- and those
//   ADD to the SIMD's time (the issue stalls), they do not hide behind other waves' VALU work.
// The compiler's version of the interior step (walk_interior_step under `if (interior)` in walk_loop_wave) spends, per
// step and SIMD: 48 cycles on the 24 sub/mul, 32 on the 8 min/max, 8 on the two compares -- and 24 on six v_cndmask /
// v_lshl_add that move the results into `cur` and the stack pointer, 14 on the record address, 8 on two compares for the
// loop control and ~27 on four branch instructions (one taken): ~160.  Here the selects become v_mov / v_add_u32 under
// EXEC masks built on the scalar unit (11 cycles instead of 24), the loop has ONE branch per iteration (the taken
// back edge) and one compare: ~134.  Same per-lane steps in the same order as walk_loop_wave: pixels and counters do
// not change.  (Only for SIMPLE trees, without COUNTERS, for waves without unsafe rays; everything else takes walk_loop_wave.)
#ifndef BRT_WALK_FAST
#define BRT_WALK_FAST BRT_HAND_ASM
#endif
#ifndef BRT_WALK_FAST_TOP
#define BRT_WALK_FAST_TOP 1   // the hand-written loop also for scenes walked from the LDS tile + global memory (walk_wave_top_asm)
#endif
// The whole wave-level walk loop (walk_loop_wave's nest: interior steps until `vote` lanes wait at a leaf, one leaf step,
// until at most exit_at lanes still walk) as ONE block of hand-scheduled code.
//   cur / spa     descriptor of the lane's current node (sign-extended 16-bit form: interior >= 0, leaf < -1, DONE = -1) and
//                 LDS byte address of its stack top (16-bit entries, 128 bytes apart)
//   gofs          the ray's granule offsets {x, y, z} inside a pair record; the records start at LDS address 0 (they are the
//                 first thing in the kernel's dynamic LDS and the kernel has no static LDS: launch_persistent_t, brt_trace.h,
//                 checks that on the HOST and fails the launch with hipErrorInvalidConfiguration otherwise)
//   sph           LDS byte address of the spheres {centre, r^2}
// The record and the sphere live in FIXED registers v[100:113]: inline asm cannot name the single registers of a 128-bit
// operand, and the arithmetic works on them in place.
// Interior step: raytrace.wgsl:327-342 + 387-398 (see walk_interior_step for the forms used: near / far planes by granule,
// the three compares of the push rule as one).  Leaf step: raytrace.wgsl:348-362 + 371-383 -- the discriminant, then
// hipcc's own correctly rounded sqrt and divide expansions, instruction for instruction as it emits them for sphere_test
// (2^32 pre-scale below 2^-96, v_sqrt_f32 + two residual corrections, class fix-up; v_div_scale x2, v_rcp_f32 + one Newton step,
// three fma steps, v_div_fmas, v_div_fixup), with the wait states its hazard recogniser would insert; the accept rule
// t > 0.001 && t < closest moves t and the sphere id under EXEC.
BRT_DEV void walk_wave_lds_asm(uint32_t& cur, uint32_t& spa, float& closest, uint32_t& closest_idx, uint32_t gofs_x, uint32_t gofs_y,
                               uint32_t gofs_z, f3 o, f3 inv, f3 d, float a, uint32_t sph, uint32_t exit_at, uint32_t vote, AsmCounts* ac = nullptr) {
#if BRT_HAND_ASM
#if BRT_ASM_COUNT
    uint32_t c_ie, c_il, c_le, c_ll, c_tmp;
#define BRT_COUNT_INT "s_bcnt1_i32_b64 %[c_tmp], exec\n s_add_u32 %[c_il], %[c_il], %[c_tmp]\n s_add_u32 %[c_ie], %[c_ie], 1\n"
#define BRT_COUNT_LEAF "s_bcnt1_i32_b64 %[c_tmp], exec\n s_add_u32 %[c_ll], %[c_ll], %[c_tmp]\n s_add_u32 %[c_le], %[c_le], 1\n"
#else
#define BRT_COUNT_INT
#define BRT_COUNT_LEAF
#endif
    uint32_t t0, tx, ty, tz, pop, cnt, nw, thr;
    float below;
    uint64_t s_all, s_take, s_p2, s_any, s_both;
    const uint32_t rec_bytes = PAIR_BYTES, c_tiny = 0x0f800000u /* 2^-96 */, c_eps = 0x3a83126fu /* 0.001f */, c_cls = 0x260u /* +-0, +inf */;
    asm volatile(
#if BRT_ASM_COUNT
        "s_mov_b32 %[c_ie], 0\n s_mov_b32 %[c_il], 0\n s_mov_b32 %[c_le], 0\n s_mov_b32 %[c_ll], 0\n"
#endif
        "s_waitcnt lgkmcnt(0)\n"                                // nothing of the compiler's in flight: the counted waits below are exact
        "s_mov_b64 %[s_all], exec\n"
        "v_add_u32_e32 %[below], -1, %[closest]\n"              // the largest float below closest (closest is FLT_MAX or an accepted t > 0)
        // ---- outer loop: leave when at most exit_at lanes still walk -------------------------------------------------------
        "3:\n"
        "v_cmp_ne_u32_e32 vcc, -1, %[cur]\n"
        "s_bcnt1_i32_b64 %[nw], vcc\n"
        "s_cmp_le_u32 %[nw], %[exit_at]\n"
        "s_cbranch_scc1 9f\n"
        "s_max_u32 %[thr], %[nw], %[vote]\n"                    // interior steps while more than max(walking, vote) - vote lanes are at one
        "s_sub_u32 %[thr], %[thr], %[vote]\n"
        "v_cmp_lt_i32_e32 vcc, -1, %[cur]\n"                    // interior descriptors are >= 0
        "s_bcnt1_i32_b64 %[cnt], vcc\n"
        "s_cmp_le_u32 %[cnt], %[thr]\n"
        "s_cbranch_scc1 5f\n"
        // ---- interior steps ----------------------------------------------------------------------------------------------------
        "1:\n"
        "s_mov_b64 %[s_take], vcc\n"
        "s_mov_b64 exec, vcc\n"
        BRT_COUNT_INT
        "ds_read_i16 %[pop], %[spa]\n"                          // the would-be pop: needs no address arithmetic, goes out first
        "v_mul_lo_u32 %[t0], %[cur], %[rec_bytes]\n"
        "v_add_u32_e32 %[tx], %[t0], %[gofs_x]\n"
        "ds_read_b128 v[100:103], %[tx]\n"                      // { near L, near R, far L, far R } per axis
        "v_add_u32_e32 %[ty], %[t0], %[gofs_y]\n"
        "ds_read_b128 v[104:107], %[ty]\n"
        "v_add_u32_e32 %[tz], %[t0], %[gofs_z]\n"
        "ds_read_b128 v[108:111], %[tz]\n"
        "ds_read_b64 v[112:113], %[t0] offset:96\n"             // descriptors of L and R
        "s_waitcnt lgkmcnt(3)\n"
        "v_sub_f32_e32 v100, v100, %[ox]\n v_sub_f32_e32 v101, v101, %[ox]\n v_sub_f32_e32 v102, v102, %[ox]\n v_sub_f32_e32 v103, v103, %[ox]\n"
        "v_mul_f32_e32 v100, v100, %[ix]\n v_mul_f32_e32 v101, v101, %[ix]\n v_mul_f32_e32 v102, v102, %[ix]\n v_mul_f32_e32 v103, v103, %[ix]\n"
        "s_waitcnt lgkmcnt(2)\n"
        "v_sub_f32_e32 v104, v104, %[oy]\n v_sub_f32_e32 v105, v105, %[oy]\n v_sub_f32_e32 v106, v106, %[oy]\n v_sub_f32_e32 v107, v107, %[oy]\n"
        "v_mul_f32_e32 v104, v104, %[iy]\n v_mul_f32_e32 v105, v105, %[iy]\n v_mul_f32_e32 v106, v106, %[iy]\n v_mul_f32_e32 v107, v107, %[iy]\n"
        "s_waitcnt lgkmcnt(1)\n"
        "v_sub_f32_e32 v108, v108, %[oz]\n v_sub_f32_e32 v109, v109, %[oz]\n v_sub_f32_e32 v110, v110, %[oz]\n v_sub_f32_e32 v111, v111, %[oz]\n"
        "v_mul_f32_e32 v108, v108, %[iz]\n v_mul_f32_e32 v109, v109, %[iz]\n v_mul_f32_e32 v110, v110, %[iz]\n v_mul_f32_e32 v111, v111, %[iz]\n"
        "v_max_f32_e32 v100, v100, v104\n"
        "v_max_f32_e32 v101, v101, v105\n"
        "v_min_f32_e32 v102, v102, v106\n"
        "v_min_f32_e32 v103, v103, v107\n"
        "v_max3_f32 v100, v100, v108, 1\n"                      // t_near = max(.., denorm_min)
        "v_max3_f32 v101, v101, v109, 1\n"
        "v_min3_f32 v102, v102, v110, %[below]\n"               // t_far = min(.., below(closest))
        "v_min3_f32 v103, v103, v111, %[below]\n"
        "s_waitcnt lgkmcnt(0)\n"                                // descriptors (and the pop, which went out first) are here
        "ds_write_b16 %[spa], v112 offset:128\n"                // child L above the top: dead unless both are pushed
        "v_cmp_le_f32_e32 vcc, v100, v102\n"                    // p1: child L is pushed (raytrace.wgsl:331)
        "v_cmp_le_f32_e64 %[s_p2], v101, v103\n"                // p2: child R is pushed (raytrace.wgsl:338)
        "s_or_b64 %[s_any], vcc, %[s_p2]\n"
        "s_and_b64 %[s_both], vcc, %[s_p2]\n"
        "s_andn2_b64 exec, %[s_take], %[s_any]\n"               // no child pushed: pop
        "v_mov_b32_e32 %[cur], %[pop]\n"
        "v_add_u32_e32 %[spa], 0xffffff80, %[spa]\n"
        "s_andn2_b64 exec, vcc, %[s_p2]\n"                      // only L
        "v_mov_b32_e32 %[cur], v112\n"
        "s_mov_b64 exec, %[s_p2]\n"                             // R (pushed last, popped first)
        "v_mov_b32_e32 %[cur], v113\n"
        "s_mov_b64 exec, %[s_both]\n"                           // both: L stays on the stack
        "v_add_u32_e32 %[spa], 0x80, %[spa]\n"
        "s_mov_b64 exec, %[s_all]\n"
        "v_cmp_lt_i32_e32 vcc, -1, %[cur]\n"
        "s_bcnt1_i32_b64 %[cnt], vcc\n"
        "s_cmp_gt_u32 %[cnt], %[thr]\n"
        "s_cbranch_scc1 1b\n"
        // ---- one leaf step for every lane that waits at a leaf --------------------------------------------------------------
        "5:\n"
        "v_cmp_gt_i32_e32 vcc, -1, %[cur]\n"                    // leaf descriptors are < -1
        "s_and_b64 exec, %[s_all], vcc\n"
        BRT_COUNT_LEAF
        "v_and_b32_e32 v112, 0x3fff, %[cur]\n"                  // the leaf's sphere
        "v_lshl_add_u32 %[t0], v112, 4, %[sph]\n"
        "ds_read_b128 v[100:103], %[t0]\n"                      // { centre, r^2 }
        "ds_read_i16 %[cur], %[spa]\n"                          // pop
        "s_waitcnt lgkmcnt(1)\n"
        "v_sub_f32_e32 v104, v100, %[ox]\n"                     // oc = centre - origin
        "v_sub_f32_e32 v105, v101, %[oy]\n"
        "v_sub_f32_e32 v106, v102, %[oz]\n"
        "v_mul_f32_e32 v107, %[dx], v104\n"                     // h = dot(d, oc) = (dx ocx + dy ocy) + dz ocz
        "v_mul_f32_e32 v108, %[dy], v105\n"
        "v_mul_f32_e32 v104, v104, v104\n"                      // dot(oc, oc)
        "v_mul_f32_e32 v105, v105, v105\n"
        "v_add_f32_e32 v104, v104, v105\n"
        "v_mul_f32_e32 v105, v106, v106\n"
        "v_mul_f32_e32 v109, %[dz], v106\n"
        "v_add_f32_e32 v107, v107, v108\n"
        "v_add_f32_e32 v104, v105, v104\n"
        "v_add_f32_e32 v107, v109, v107\n"                      // h
        "v_sub_f32_e32 v104, v104, v103\n"                      // c = dot(oc, oc) - r^2
        "v_mul_f32_e32 v105, v107, v107\n"                      // h h
        "v_mul_f32_e32 v104, %[a], v104\n"                      // a c
        "v_sub_f32_e32 v104, v105, v104\n"                      // discriminant
        // sqrt(discriminant), correctly rounded: hipcc's expansion (a negative argument gives NaN, rejected below like the
        // shader's -1).  Same operations on the same values as the compiler emits; the order is chosen so that every wait
        // state its hazard recogniser fills with s_nop (VALU -> vcc / SGPR -> v_cndmask: 2, v_sqrt / v_rcp -> use: 1) holds
        // an instruction that is needed anyway.
        "v_cmp_gt_f32_e32 vcc, %[c_tiny], v104\n"               // below 2^-96: scale by 2^32 (vcc stays until the result is scaled back)
        "v_mul_f32_e32 v105, 0x4f800000, v104\n"
        "v_add_u32_e32 %[spa], 0xffffff80, %[spa]\n"            // (the pop's pointer move)
        "v_cndmask_b32_e32 v104, v104, v105, vcc\n"
        "v_sqrt_f32_e32 v105, v104\n"                           // s
        "v_cmp_class_f32_e64 %[s_both], v104, %[c_cls]\n"       // +-0 and +inf are their own roots
        "v_add_u32_e32 v106, -1, v105\n"                        // s - 1 ulp
        "v_add_u32_e32 v109, 1, v105\n"                         // s + 1 ulp
        "v_fma_f32 v108, -v106, v105, v104\n"                   // x - (s - 1 ulp) s
        "v_fma_f32 v110, -v109, v105, v104\n"                   // x - (s + 1 ulp) s
        "v_cmp_ge_f32_e64 %[s_p2], 0, v108\n"
        "v_cmp_lt_f32_e64 %[s_any], 0, v110\n"
        "s_nop 0\n"                                             // (the one wait state left over)
        "v_cndmask_b32_e64 v106, v105, v106, %[s_p2]\n"
        "v_cndmask_b32_e64 v105, v106, v109, %[s_any]\n"
        "v_mul_f32_e32 v106, 0x37800000, v105\n"
        "v_cndmask_b32_e32 v105, v105, v106, vcc\n"
        "v_cndmask_b32_e64 v104, v105, v104, %[s_both]\n"
        "v_sub_f32_e32 v104, v107, v104\n"                      // h - sqrt(discriminant)
        // ... / a, correctly rounded: hipcc's expansion
        "v_div_scale_f32 v105, %[s_p2], %[a], %[a], v104\n"
        "v_rcp_f32_e32 v106, v105\n"
        "v_div_scale_f32 v107, vcc, v104, %[a], v104\n"
        "v_fma_f32 v109, -v105, v106, 1.0\n"
        "v_fmac_f32_e32 v106, v109, v106\n"
        "v_mul_f32_e32 v108, v107, v106\n"
        "v_fma_f32 v109, -v105, v108, v107\n"
        "v_fmac_f32_e32 v108, v109, v106\n"
        "v_fma_f32 v105, -v105, v108, v107\n"
        "v_div_fmas_f32 v105, v105, v106, v108\n"
        "v_div_fixup_f32 v104, v105, %[a], v104\n"              // t
        // accepted iff t > 0.001 && t < closest (raytrace.wgsl:353-354; strict: the first sphere reached wins ties)
        "v_cmp_lt_f32_e32 vcc, %[c_eps], v104\n"
        "v_cmp_lt_f32_e64 %[s_p2], v104, %[closest]\n"
        "s_and_b64 exec, vcc, %[s_p2]\n"
        "v_mov_b32_e32 %[closest], v104\n"
        "v_mov_b32_e32 %[cidx], v112\n"
        "v_add_u32_e32 %[below], -1, v104\n"
        "s_mov_b64 exec, %[s_all]\n"
        "s_waitcnt lgkmcnt(0)\n"                                // the pop has long arrived; the next test reads it
        "s_branch 3b\n"
        "9:\n"
        "s_waitcnt lgkmcnt(0)\n"
        : [cur] "+v"(cur), [spa] "+v"(spa), [closest] "+v"(closest), [cidx] "+v"(closest_idx), [below] "=&v"(below), [t0] "=&v"(t0),
          [tx] "=&v"(tx), [ty] "=&v"(ty), [tz] "=&v"(tz), [pop] "=&v"(pop), [cnt] "=&s"(cnt), [nw] "=&s"(nw), [thr] "=&s"(thr),
          [s_all] "=&s"(s_all), [s_take] "=&s"(s_take), [s_p2] "=&s"(s_p2), [s_any] "=&s"(s_any), [s_both] "=&s"(s_both)
#if BRT_ASM_COUNT
          , [c_ie] "=&s"(c_ie), [c_il] "=&s"(c_il), [c_le] "=&s"(c_le), [c_ll] "=&s"(c_ll), [c_tmp] "=&s"(c_tmp)
#endif
        : [gofs_x] "v"(gofs_x), [gofs_y] "v"(gofs_y), [gofs_z] "v"(gofs_z), [ox] "v"(o.x), [oy] "v"(o.y), [oz] "v"(o.z), [ix] "v"(inv.x),
          [iy] "v"(inv.y), [iz] "v"(inv.z), [dx] "v"(d.x), [dy] "v"(d.y), [dz] "v"(d.z), [a] "v"(a), [sph] "s"(sph), [exit_at] "s"(exit_at),
          [vote] "s"(vote), [rec_bytes] "s"(rec_bytes), [c_tiny] "s"(c_tiny), [c_eps] "s"(c_eps), [c_cls] "s"(c_cls)
        : "vcc", "scc", "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112",
          "v113");
#if BRT_ASM_COUNT
    if (first_active_lane()) { ac->int_exec += c_ie; ac->int_lanes += c_il; ac->leaf_exec += c_le; ac->leaf_lanes += c_ll; }
#endif
#undef BRT_COUNT_INT
#undef BRT_COUNT_LEAF
#endif
}

// The same loop for scenes that do not fit the LDS (SCENE_LDS_TOP: the top of the tree -- the records below byte offset near_bytes -- in the
// LDS tile at address 0, the rest and the spheres in global memory; SCENE_GLOBAL: near_bytes = 0).  An interior step reads its record
// from the tile or from the global array, lane by lane (two EXEC masks, one destination); the loads are waited for together, the
// arithmetic is walk_wave_lds_asm's.  The compiler's loop for these modes (walk_loop_wave) spends ~8 branch instructions and six selects
// per step: 1.03 G branches on the 10 004-sphere frame against 0.38 G on the cover frame.
BRT_DEV void walk_wave_top_asm(uint32_t& cur, uint32_t& spa, float& closest, uint32_t& closest_idx, uint32_t gofs_x, uint32_t gofs_y,
                               uint32_t gofs_z, f3 o, f3 inv, f3 d, float a, uint32_t near_bytes, uint64_t far, uint64_t sphg,
                               uint32_t exit_at, uint32_t vote, AsmCounts* ac = nullptr) {
#if BRT_HAND_ASM
#if BRT_ASM_COUNT
    uint32_t c_ie, c_il, c_le, c_ll, c_tmp;
#define BRT_COUNT_INT "s_bcnt1_i32_b64 %[c_tmp], exec\n s_add_u32 %[c_il], %[c_il], %[c_tmp]\n s_add_u32 %[c_ie], %[c_ie], 1\n"
#define BRT_COUNT_LEAF "s_bcnt1_i32_b64 %[c_tmp], exec\n s_add_u32 %[c_ll], %[c_ll], %[c_tmp]\n s_add_u32 %[c_le], %[c_le], 1\n"
#else
#define BRT_COUNT_INT
#define BRT_COUNT_LEAF
#endif
    uint32_t t0, tx, ty, tz, pop, cnt, nw, thr;
    float below;
    uint64_t s_all, s_take, s_p2, s_any, s_both;
    const uint32_t rec_bytes = PAIR_BYTES, c_tiny = 0x0f800000u /* 2^-96 */, c_eps = 0x3a83126fu /* 0.001f */, c_cls = 0x260u /* +-0, +inf */;
    asm volatile(
#if BRT_ASM_COUNT
        "s_mov_b32 %[c_ie], 0\n s_mov_b32 %[c_il], 0\n s_mov_b32 %[c_le], 0\n s_mov_b32 %[c_ll], 0\n"
#endif
        "s_waitcnt lgkmcnt(0)\n"                                // nothing of the compiler's in flight: the counted waits below are exact
        "s_mov_b64 %[s_all], exec\n"
        "v_add_u32_e32 %[below], -1, %[closest]\n"              // the largest float below closest (closest is FLT_MAX or an accepted t > 0)
        // ---- outer loop: leave when at most exit_at lanes still walk -------------------------------------------------------
        "3:\n"
        "v_cmp_ne_u32_e32 vcc, -1, %[cur]\n"
        "s_bcnt1_i32_b64 %[nw], vcc\n"
        "s_cmp_le_u32 %[nw], %[exit_at]\n"
        "s_cbranch_scc1 9f\n"
        "s_max_u32 %[thr], %[nw], %[vote]\n"                    // interior steps while more than max(walking, vote) - vote lanes are at one
        "s_sub_u32 %[thr], %[thr], %[vote]\n"
        "v_cmp_lt_i32_e32 vcc, -1, %[cur]\n"                    // interior descriptors are >= 0
        "s_bcnt1_i32_b64 %[cnt], vcc\n"
        "s_cmp_le_u32 %[cnt], %[thr]\n"
        "s_cbranch_scc1 5f\n"
        // ---- interior steps ----------------------------------------------------------------------------------------------------
        "1:\n"
        "s_mov_b64 %[s_take], vcc\n"
        "s_mov_b64 exec, vcc\n"
        BRT_COUNT_INT
        "ds_read_i16 %[pop], %[spa]\n"                          // the would-be pop: needs no address arithmetic, goes out first
        "v_mul_lo_u32 %[t0], %[cur], %[rec_bytes]\n"
        "v_cmp_le_u32_e32 vcc, %[near_bytes], %[t0]\n"          // lanes whose record is only in the global array (L2)
        "v_add_u32_e32 %[tx], %[t0], %[gofs_x]\n"
        "s_cbranch_vccnz 8f\n"                                  // ... are rare once the records are numbered by their use (apply_hot_order):
        "ds_read_b128 v[100:103], %[tx]\n"                      // the common step reads the tile only, behind ONE branch that is not taken
        "v_add_u32_e32 %[ty], %[t0], %[gofs_y]\n"
        "ds_read_b128 v[104:107], %[ty]\n"
        "v_add_u32_e32 %[tz], %[t0], %[gofs_z]\n"
        "ds_read_b128 v[108:111], %[tz]\n"
        "ds_read_b64 v[112:113], %[t0] offset:96\n"             // descriptors of L and R
        "6:\n"
        // (counted waits that hold for every mix: the LDS queue is {pop, x, y, z, descriptors} or just {pop}, the vector-memory queue {x, y, z,
        //  descriptors} or empty -- a lane's x granule has arrived when at most 3 are outstanding in BOTH)
        "s_waitcnt vmcnt(3) lgkmcnt(3)\n"
        "v_sub_f32_e32 v100, v100, %[ox]\n v_sub_f32_e32 v101, v101, %[ox]\n v_sub_f32_e32 v102, v102, %[ox]\n v_sub_f32_e32 v103, v103, %[ox]\n"
        "v_mul_f32_e32 v100, v100, %[ix]\n v_mul_f32_e32 v101, v101, %[ix]\n v_mul_f32_e32 v102, v102, %[ix]\n v_mul_f32_e32 v103, v103, %[ix]\n"
        "s_waitcnt vmcnt(2) lgkmcnt(2)\n"
        "v_sub_f32_e32 v104, v104, %[oy]\n v_sub_f32_e32 v105, v105, %[oy]\n v_sub_f32_e32 v106, v106, %[oy]\n v_sub_f32_e32 v107, v107, %[oy]\n"
        "v_mul_f32_e32 v104, v104, %[iy]\n v_mul_f32_e32 v105, v105, %[iy]\n v_mul_f32_e32 v106, v106, %[iy]\n v_mul_f32_e32 v107, v107, %[iy]\n"
        "s_waitcnt vmcnt(1) lgkmcnt(1)\n"
        "v_sub_f32_e32 v108, v108, %[oz]\n v_sub_f32_e32 v109, v109, %[oz]\n v_sub_f32_e32 v110, v110, %[oz]\n v_sub_f32_e32 v111, v111, %[oz]\n"
        "v_mul_f32_e32 v108, v108, %[iz]\n v_mul_f32_e32 v109, v109, %[iz]\n v_mul_f32_e32 v110, v110, %[iz]\n v_mul_f32_e32 v111, v111, %[iz]\n"
        "v_max_f32_e32 v100, v100, v104\n"
        "v_max_f32_e32 v101, v101, v105\n"
        "v_min_f32_e32 v102, v102, v106\n"
        "v_min_f32_e32 v103, v103, v107\n"
        "v_max3_f32 v100, v100, v108, 1\n"                      // t_near = max(.., denorm_min)
        "v_max3_f32 v101, v101, v109, 1\n"
        "v_min3_f32 v102, v102, v110, %[below]\n"               // t_far = min(.., below(closest))
        "v_min3_f32 v103, v103, v111, %[below]\n"
        "s_waitcnt vmcnt(0) lgkmcnt(0)\n"                       // descriptors (and the pop, which went out first) are here
        "ds_write_b16 %[spa], v112 offset:128\n"                // child L above the top: dead unless both are pushed
        "v_cmp_le_f32_e32 vcc, v100, v102\n"                    // p1: child L is pushed (raytrace.wgsl:331)
        "v_cmp_le_f32_e64 %[s_p2], v101, v103\n"                // p2: child R is pushed (raytrace.wgsl:338)
        "s_or_b64 %[s_any], vcc, %[s_p2]\n"
        "s_and_b64 %[s_both], vcc, %[s_p2]\n"
        "s_andn2_b64 exec, %[s_take], %[s_any]\n"               // no child pushed: pop
        "v_mov_b32_e32 %[cur], %[pop]\n"
        "v_add_u32_e32 %[spa], 0xffffff80, %[spa]\n"
        "s_andn2_b64 exec, vcc, %[s_p2]\n"                      // only L
        "v_mov_b32_e32 %[cur], v112\n"
        "s_mov_b64 exec, %[s_p2]\n"                             // R (pushed last, popped first)
        "v_mov_b32_e32 %[cur], v113\n"
        "s_mov_b64 exec, %[s_both]\n"                           // both: L stays on the stack
        "v_add_u32_e32 %[spa], 0x80, %[spa]\n"
        "s_mov_b64 exec, %[s_all]\n"
        "v_cmp_lt_i32_e32 vcc, -1, %[cur]\n"
        "s_bcnt1_i32_b64 %[cnt], vcc\n"
        "s_cmp_gt_u32 %[cnt], %[thr]\n"
        "s_cbranch_scc1 1b\n"
        // ---- one leaf step for every lane that waits at a leaf --------------------------------------------------------------
        "5:\n"
        "v_cmp_gt_i32_e32 vcc, -1, %[cur]\n"                    // leaf descriptors are < -1
        "s_and_b64 exec, %[s_all], vcc\n"
        BRT_COUNT_LEAF
        "v_and_b32_e32 v112, 0x3fff, %[cur]\n"                  // the leaf's sphere
        "v_lshlrev_b32_e32 %[t0], 4, v112\n"
        "global_load_dwordx4 v[100:103], %[t0], %[sphg]\n"      // { centre, r^2 }
        "ds_read_i16 %[cur], %[spa]\n"                          // pop
        "s_waitcnt vmcnt(0)\n"
        "v_sub_f32_e32 v104, v100, %[ox]\n"                     // oc = centre - origin
        "v_sub_f32_e32 v105, v101, %[oy]\n"
        "v_sub_f32_e32 v106, v102, %[oz]\n"
        "v_mul_f32_e32 v107, %[dx], v104\n"                     // h = dot(d, oc) = (dx ocx + dy ocy) + dz ocz
        "v_mul_f32_e32 v108, %[dy], v105\n"
        "v_mul_f32_e32 v104, v104, v104\n"                      // dot(oc, oc)
        "v_mul_f32_e32 v105, v105, v105\n"
        "v_add_f32_e32 v104, v104, v105\n"
        "v_mul_f32_e32 v105, v106, v106\n"
        "v_mul_f32_e32 v109, %[dz], v106\n"
        "v_add_f32_e32 v107, v107, v108\n"
        "v_add_f32_e32 v104, v105, v104\n"
        "v_add_f32_e32 v107, v109, v107\n"                      // h
        "v_sub_f32_e32 v104, v104, v103\n"                      // c = dot(oc, oc) - r^2
        "v_mul_f32_e32 v105, v107, v107\n"                      // h h
        "v_mul_f32_e32 v104, %[a], v104\n"                      // a c
        "v_sub_f32_e32 v104, v105, v104\n"                      // discriminant
        // sqrt(discriminant), correctly rounded: hipcc's expansion (a negative argument gives NaN, rejected below like the
        // shader's -1).  Same operations on the same values as the compiler emits; the order is chosen so that every wait
        // state its hazard recogniser fills with s_nop (VALU -> vcc / SGPR -> v_cndmask: 2, v_sqrt / v_rcp -> use: 1) holds
        // an instruction that is needed anyway.
        "v_cmp_gt_f32_e32 vcc, %[c_tiny], v104\n"               // below 2^-96: scale by 2^32 (vcc stays until the result is scaled back)
        "v_mul_f32_e32 v105, 0x4f800000, v104\n"
        "v_add_u32_e32 %[spa], 0xffffff80, %[spa]\n"            // (the pop's pointer move)
        "v_cndmask_b32_e32 v104, v104, v105, vcc\n"
        "v_sqrt_f32_e32 v105, v104\n"                           // s
        "v_cmp_class_f32_e64 %[s_both], v104, %[c_cls]\n"       // +-0 and +inf are their own roots
        "v_add_u32_e32 v106, -1, v105\n"                        // s - 1 ulp
        "v_add_u32_e32 v109, 1, v105\n"                         // s + 1 ulp
        "v_fma_f32 v108, -v106, v105, v104\n"                   // x - (s - 1 ulp) s
        "v_fma_f32 v110, -v109, v105, v104\n"                   // x - (s + 1 ulp) s
        "v_cmp_ge_f32_e64 %[s_p2], 0, v108\n"
        "v_cmp_lt_f32_e64 %[s_any], 0, v110\n"
        "s_nop 0\n"                                             // (the one wait state left over)
        "v_cndmask_b32_e64 v106, v105, v106, %[s_p2]\n"
        "v_cndmask_b32_e64 v105, v106, v109, %[s_any]\n"
        "v_mul_f32_e32 v106, 0x37800000, v105\n"
        "v_cndmask_b32_e32 v105, v105, v106, vcc\n"
        "v_cndmask_b32_e64 v104, v105, v104, %[s_both]\n"
        "v_sub_f32_e32 v104, v107, v104\n"                      // h - sqrt(discriminant)
        // ... / a, correctly rounded: hipcc's expansion
        "v_div_scale_f32 v105, %[s_p2], %[a], %[a], v104\n"
        "v_rcp_f32_e32 v106, v105\n"
        "v_div_scale_f32 v107, vcc, v104, %[a], v104\n"
        "v_fma_f32 v109, -v105, v106, 1.0\n"
        "v_fmac_f32_e32 v106, v109, v106\n"
        "v_mul_f32_e32 v108, v107, v106\n"
        "v_fma_f32 v109, -v105, v108, v107\n"
        "v_fmac_f32_e32 v108, v109, v106\n"
        "v_fma_f32 v105, -v105, v108, v107\n"
        "v_div_fmas_f32 v105, v105, v106, v108\n"
        "v_div_fixup_f32 v104, v105, %[a], v104\n"              // t
        // accepted iff t > 0.001 && t < closest (raytrace.wgsl:353-354; strict: the first sphere reached wins ties)
        "v_cmp_lt_f32_e32 vcc, %[c_eps], v104\n"
        "v_cmp_lt_f32_e64 %[s_p2], v104, %[closest]\n"
        "s_and_b64 exec, vcc, %[s_p2]\n"
        "v_mov_b32_e32 %[closest], v104\n"
        "v_mov_b32_e32 %[cidx], v112\n"
        "v_add_u32_e32 %[below], -1, v104\n"
        "s_mov_b64 exec, %[s_all]\n"
        "s_waitcnt lgkmcnt(0)\n"                                // the pop has long arrived; the next test reads it
        "s_branch 3b\n"
        // ---- out of line: an interior step with lanes on both sides (two EXEC masks, one destination) ---------------------------
        "8:\n"
        "v_add_u32_e32 %[ty], %[t0], %[gofs_y]\n"
        "v_add_u32_e32 %[tz], %[t0], %[gofs_z]\n"
        "s_and_b64 exec, %[s_take], vcc\n"                      // the global loads go out first ...
        "global_load_dwordx4 v[100:103], %[tx], %[far]\n"       // { near L, near R, far L, far R } per axis
        "global_load_dwordx4 v[104:107], %[ty], %[far]\n"
        "global_load_dwordx4 v[108:111], %[tz], %[far]\n"
        "global_load_dwordx2 v[112:113], %[t0], %[far] offset:96\n"
        "s_andn2_b64 exec, %[s_take], vcc\n"                    // ... then the lanes in the LDS tile
        "s_cbranch_execz 7f\n"
        "ds_read_b128 v[100:103], %[tx]\n"
        "ds_read_b128 v[104:107], %[ty]\n"
        "ds_read_b128 v[108:111], %[tz]\n"
        "ds_read_b64 v[112:113], %[t0] offset:96\n"
        "7:\n"
        "s_mov_b64 exec, %[s_take]\n"
        "s_branch 6b\n"
        "9:\n"
        "s_waitcnt lgkmcnt(0)\n"
        : [cur] "+v"(cur), [spa] "+v"(spa), [closest] "+v"(closest), [cidx] "+v"(closest_idx), [below] "=&v"(below), [t0] "=&v"(t0),
          [tx] "=&v"(tx), [ty] "=&v"(ty), [tz] "=&v"(tz), [pop] "=&v"(pop), [cnt] "=&s"(cnt), [nw] "=&s"(nw), [thr] "=&s"(thr),
          [s_all] "=&s"(s_all), [s_take] "=&s"(s_take), [s_p2] "=&s"(s_p2), [s_any] "=&s"(s_any), [s_both] "=&s"(s_both)
#if BRT_ASM_COUNT
          , [c_ie] "=&s"(c_ie), [c_il] "=&s"(c_il), [c_le] "=&s"(c_le), [c_ll] "=&s"(c_ll), [c_tmp] "=&s"(c_tmp)
#endif
        : [gofs_x] "v"(gofs_x), [gofs_y] "v"(gofs_y), [gofs_z] "v"(gofs_z), [ox] "v"(o.x), [oy] "v"(o.y), [oz] "v"(o.z), [ix] "v"(inv.x),
          [iy] "v"(inv.y), [iz] "v"(inv.z), [dx] "v"(d.x), [dy] "v"(d.y), [dz] "v"(d.z), [a] "v"(a), [near_bytes] "s"(near_bytes), [far] "s"(far), [sphg] "s"(sphg), [exit_at] "s"(exit_at),
          [vote] "s"(vote), [rec_bytes] "s"(rec_bytes), [c_tiny] "s"(c_tiny), [c_eps] "s"(c_eps), [c_cls] "s"(c_cls)
        : "vcc", "scc", "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112",
          "v113");
#if BRT_ASM_COUNT
    if (first_active_lane()) { ac->int_exec += c_ie; ac->int_lanes += c_il; ac->leaf_exec += c_le; ac->leaf_lanes += c_ll; }
#endif
#undef BRT_COUNT_INT
#undef BRT_COUNT_LEAF
#endif
}

// ---- thin waves: ONE RAY PER ROW OF 16 LANES ("row mode", walk_rows_asm) -----------------------------------------------------
// A wave that walks at most four rays -- the tail of a pixel chain: config 3 ends in ONE 11 064-ray pixel, a rank's share of a
// frame split over 8 GPUs ends in its longest pixel -- is bound by its own instruction issue: a lone wave issues one instruction
// every ~5 cycles whatever the instruction does (tests/tools/issue_bench.hip), and an interior step of walk_wave_lds_asm is ~66 of
// them for a single useful lane.  Here a ray gets a row of 16 lanes and the step is spread over them:
//     quad 0 = { near L x, y, z | clamp }   quad 1 = { far L x, y, z | clamp, child L's descriptor }
//     quad 2 = { near R x, y, z | clamp }   quad 3 = { far R x, y, z | clamp, child R's descriptor }
// every lane reads ONE word of the pair record (its plane, chosen by the ray's sign like the granule offsets of the wide loop),
// computes (plane - o) * (1 / d) once -- the far quads multiply by -(1 / d): -(x y) is x (-y) exactly --, the fourth lane of a quad
// holds the clamp (denorm_min for the near side, -below(closest) for the far side), two quad-permute max steps leave
// max(near planes, denorm_min) resp. -min(far planes, below) in every lane of the quad (v_max_f32 is a total order on non-NaN values:
// any association gives the same bits; rays that can produce a NaN never come here), four row broadcasts bring them to every lane
// and each lane makes the reference's decision (raytrace.wgsl:331, :338-341) for itself: ~38 instructions per step instead of ~66.
// The leaf step forms the two dot products of hit_sphere (raytrace.wgsl:371-383) across three lanes in the shader's order,
// (x x' + y y') + z z', and then runs hipcc's sqrt and divide expansions exactly as walk_wave_lds_asm does.  Same nodes in the same
// order, same ties, same t: pixels and counters do not change.  The rays' stacks stay where they are (the row reads and writes its
// source lane's column of the wave's stack array), so a walk that was suspended by the wide loop is finished here as it stands.
// Rays travel through 80 bytes of LDS scratch per row: { o, a | 1/d, closest | d, closest id | granule offsets, cur | stack top }.
// EXEC: unlike the other hand-written loops this one WIDENS the execution mask (to the 16 n lanes of its rows, whatever lanes were active at the
// call) and restores it at the end.  What it writes there are only its own temporaries -- asm outputs and the clobbered v100 - v113 --, which by
// the compiler's own liveness hold no value across this statement; the call sits in the round loop of the kernel, where no other side of a
// divergent branch is pending whose values could live in those registers for the lanes that are inactive here (suite, fuzz and sequence soaks
// run through it: profiles/r06/fuzz_soak.txt).
#ifndef BRT_WALK_ROWS
#define BRT_WALK_ROWS BRT_HAND_ASM
#endif
constexpr uint32_t ROWS_SCRATCH_BYTES = 4u * 80u;      // per wave
BRT_DEV void walk_rows_asm(uint32_t& cur, uint32_t spa, float& closest, uint32_t& closest_idx, uint32_t gofs_x, uint32_t gofs_y,
                           uint32_t gofs_z, f3 o, f3 inv, f3 d, float a, uint32_t sph, uint32_t scratch, AsmCounts* ac = nullptr) {
#if BRT_HAND_ASM
#if BRT_ASM_COUNT
    uint32_t c_ie, c_il, c_le, c_ll, c_tmp;
#define BRT_COUNT_INT "s_bcnt1_i32_b64 %[c_tmp], %[s_take]\n s_lshr_b32 %[c_tmp], %[c_tmp], 4\n s_add_u32 %[c_il], %[c_il], %[c_tmp]\n s_add_u32 %[c_ie], %[c_ie], 1\n"
#define BRT_COUNT_LEAF "s_bcnt1_i32_b64 %[c_tmp], %[s_take]\n s_lshr_b32 %[c_tmp], %[c_tmp], 4\n s_add_u32 %[c_ll], %[c_ll], %[c_tmp]\n s_add_u32 %[c_le], %[c_le], 1\n"
#define BRT_COUNT_INT1 "s_add_u32 %[c_il], %[c_il], 1\n s_add_u32 %[c_ie], %[c_ie], 1\n"
#else
#define BRT_COUNT_INT
#define BRT_COUNT_LEAF
#define BRT_COUNT_INT1
#endif
    uint32_t wa, rowb, ofs, sphk, rcur, rspa, rcidx, below, lane;
    float oj, mj, dj, cj, ra, rclosest, mj1, cj1;
    uint32_t n4, ofs1;
    uint64_t s_all, s_walk, s_rows, s_take, s_p2, s_any, s_both, s_pad, s_farpad;
    const uint32_t rec_bytes = PAIR_BYTES, c_tiny = 0x0f800000u /* 2^-96 */, c_eps = 0x3a83126fu /* 0.001f */, c_cls = 0x260u /* +-0, +inf */,
                   c80 = 80u;
    // (the leaf step of a row: sphere test across three lanes, hipcc's sqrt / divide expansions, accept, pop; EXEC = the rows of %[s_take])
#define BRT_ROWS_LEAF_STEP \
        "v_and_b32_e32 v112, 0x3fff, %[rcur]\n" \
        "v_lshl_add_u32 v111, v112, 4, %[sphk]\n" \
        "ds_read_b32 v100, v111\n" \
        "ds_read_i16 %[rcur], %[rspa]\n" \
        "s_waitcnt lgkmcnt(1)\n" \
        "v_sub_f32_e32 v101, v100, %[oj]\n" \
        "v_add_u32_e32 %[rspa], 0xffffff80, %[rspa]\n" \
        "v_mul_f32_e32 v102, %[dj], v101\n" \
        "v_mul_f32_e32 v103, v101, v101\n" \
        "s_nop 1\n" \
        "v_add_f32_dpp v106, v102, v102 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" \
        "v_add_f32_dpp v108, v103, v103 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n" \
        "s_nop 1\n" \
        "v_add_f32_dpp v107, v102, v106 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n" \
        "v_add_f32_dpp v104, v103, v108 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n" \
        "s_nop 1\n" \
        "v_subrev_f32_dpp v104, v100, v104 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n" \
        "v_mul_f32_e32 v105, v107, v107\n" \
        "v_mul_f32_e32 v104, %[ra], v104\n" \
        "v_sub_f32_e32 v104, v105, v104\n" \
        "v_cmp_gt_f32_e32 vcc, %[c_tiny], v104\n" \
        "v_mul_f32_e32 v105, 0x4f800000, v104\n" \
        "s_nop 0\n" \
        "v_cndmask_b32_e32 v104, v104, v105, vcc\n" \
        "v_sqrt_f32_e32 v105, v104\n" \
        "v_cmp_class_f32_e64 %[s_both], v104, %[c_cls]\n" \
        "v_add_u32_e32 v106, -1, v105\n" \
        "v_add_u32_e32 v109, 1, v105\n" \
        "v_fma_f32 v108, -v106, v105, v104\n" \
        "v_fma_f32 v110, -v109, v105, v104\n" \
        "v_cmp_ge_f32_e64 %[s_p2], 0, v108\n" \
        "v_cmp_lt_f32_e64 %[s_any], 0, v110\n" \
        "s_nop 0\n" \
        "v_cndmask_b32_e64 v106, v105, v106, %[s_p2]\n" \
        "v_cndmask_b32_e64 v105, v106, v109, %[s_any]\n" \
        "v_mul_f32_e32 v106, 0x37800000, v105\n" \
        "v_cndmask_b32_e32 v105, v105, v106, vcc\n" \
        "v_cndmask_b32_e64 v104, v105, v104, %[s_both]\n" \
        "v_sub_f32_e32 v104, v107, v104\n" \
        "v_div_scale_f32 v105, %[s_p2], %[ra], %[ra], v104\n" \
        "v_rcp_f32_e32 v106, v105\n" \
        "v_div_scale_f32 v107, vcc, v104, %[ra], v104\n" \
        "v_fma_f32 v109, -v105, v106, 1.0\n" \
        "v_fmac_f32_e32 v106, v109, v106\n" \
        "v_mul_f32_e32 v108, v107, v106\n" \
        "v_fma_f32 v109, -v105, v108, v107\n" \
        "v_fmac_f32_e32 v108, v109, v106\n" \
        "v_fma_f32 v105, -v105, v108, v107\n" \
        "v_div_fmas_f32 v105, v105, v106, v108\n" \
        "v_div_fixup_f32 v104, v105, %[ra], v104\n" \
        "s_nop 1\n" \
        "v_mov_b32_dpp v105, v104 row_newbcast:0 row_mask:0xf bank_mask:0xf\n" \
        "v_cmp_lt_f32_e32 vcc, %[c_eps], v105\n" \
        "v_cmp_lt_f32_e64 %[s_p2], v105, %[rclosest]\n" \
        "s_and_b64 exec, vcc, %[s_p2]\n" \
        "v_mov_b32_e32 %[rclosest], v105\n" \
        "v_mov_b32_e32 %[rcidx], v112\n" \
        "v_add_u32_e32 %[below], -1, v105\n" \
        "v_xor_b32_e32 v106, 0x80000000, %[below]\n" \
        "v_cndmask_b32_e64 %[cj], %[cj], v106, %[s_farpad]\n" \
        "s_mov_b64 exec, %[s_rows]\n" \
        "s_waitcnt lgkmcnt(0)\n"
    asm volatile(
#if BRT_ASM_COUNT
        "s_mov_b32 %[c_ie], 0\n s_mov_b32 %[c_il], 0\n s_mov_b32 %[c_le], 0\n s_mov_b32 %[c_ll], 0\n"
#endif
        "s_waitcnt lgkmcnt(0)\n"
        "s_mov_b64 %[s_all], exec\n"
        // ---- the walking lanes publish their rays: lane of rank r among them -> scratch row r ----------------------------------
        "v_cmp_ne_u32_e32 vcc, -1, %[cur]\n"
        "s_mov_b64 %[s_walk], vcc\n"
        "s_mov_b64 exec, vcc\n"
        "v_mbcnt_lo_u32_b32 %[wa], exec_lo, 0\n"               // rank among the walking lanes (EXEC is their mask)
        "v_mbcnt_hi_u32_b32 %[wa], exec_hi, %[wa]\n"
        "v_mul_u32_u24_e32 %[wa], %[c80], %[wa]\n"
        "v_add_u32_e32 %[wa], %[scratch], %[wa]\n"
        "ds_write2_b32 %[wa], %[ox], %[oy] offset0:0 offset1:1\n"
        "ds_write2_b32 %[wa], %[oz], %[a] offset0:2 offset1:3\n"
        "ds_write2_b32 %[wa], %[ix], %[iy] offset0:4 offset1:5\n"
        "ds_write2_b32 %[wa], %[iz], %[closest] offset0:6 offset1:7\n"
        "ds_write2_b32 %[wa], %[dx], %[dy] offset0:8 offset1:9\n"
        "ds_write2_b32 %[wa], %[dz], %[cidx] offset0:10 offset1:11\n"
        "ds_write2_b32 %[wa], %[gofs_x], %[gofs_y] offset0:12 offset1:13\n"
        "ds_write2_b32 %[wa], %[gofs_z], %[cur] offset0:14 offset1:15\n"
        "ds_write_b32 %[wa], %[spa] offset:64\n"
        // ---- rows [0, walking lanes): EXEC = their 16 n lanes ---------------------------------------------------------------------
        "s_bcnt1_i32_b64 %[n4], %[s_walk]\n"
        "s_lshl_b32 %[n4], %[n4], 4\n"
        "s_bfm_b64 %[s_rows], %[n4], 0\n"                       // the low 16 n bits (n < 4)
        "s_cmp_eq_u32 %[n4], 64\n"
        "s_cselect_b64 %[s_rows], -1, %[s_rows]\n"
        "s_mov_b64 exec, %[s_rows]\n"
        "s_mov_b32 vcc_lo, 0x88888888\n s_mov_b32 vcc_hi, 0x88888888\n s_mov_b64 %[s_pad], vcc\n"        // fourth lane of every quad
        "s_mov_b32 vcc_lo, 0x80808080\n s_mov_b32 vcc_hi, 0x80808080\n s_mov_b64 %[s_farpad], vcc\n"     // ... of the far quads (1, 3)
        // lane roles: j = lane & 15, k = j & 3 (axis; 3 = the clamp lane), q = j >> 2 (near L, far L, near R, far R)
        "v_mbcnt_lo_u32_b32 %[lane], -1, 0\n"
        "v_mbcnt_hi_u32_b32 %[lane], -1, %[lane]\n"
        "v_lshrrev_b32_e32 %[rowb], 4, %[lane]\n"
        "v_mul_u32_u24_e32 %[rowb], %[c80], %[rowb]\n"
        "v_add_u32_e32 %[rowb], %[scratch], %[rowb]\n"          // this row's scratch
        "v_and_b32_e32 v100, 3, %[lane]\n"                      // k
        "v_lshlrev_b32_e32 v101, 2, v100\n"                     // 4 k
        "v_add_u32_e32 %[sphk], %[sph], v101\n"                 // leaf step: this lane's word of a sphere { centre, r^2 }
        "v_cndmask_b32_e64 v101, v101, 0, %[s_pad]\n"           // (a clamp lane reads axis x: its products are never used)
        "v_add_u32_e32 v101, %[rowb], v101\n"
        "ds_read_b32 %[oj], v101\n"
        "ds_read_b32 %[mj], v101 offset:16\n"
        "ds_read_b32 %[dj], v101 offset:32\n"
        "ds_read_b32 %[ofs], v101 offset:48\n"
        "ds_read_b32 %[ra], %[rowb] offset:12\n"
        "ds_read_b32 %[rclosest], %[rowb] offset:28\n"
        "ds_read_b32 %[rcidx], %[rowb] offset:44\n"
        "ds_read_b32 %[rcur], %[rowb] offset:60\n"
        "ds_read_b32 %[rspa], %[rowb] offset:64\n"
        "v_bfe_u32 v102, %[lane], 2, 2\n"                       // q
        "v_and_b32_e32 v103, 1, v102\n"                         // far side?
        "v_lshlrev_b32_e32 v104, 31, v103\n"                    // its sign bit
        "v_lshlrev_b32_e32 v103, 3, v103\n"                     // word of the granule { near L, near R, far L, far R }: far + 8 bytes,
        "v_lshrrev_b32_e32 v105, 1, v102\n"                     //                                                      R + 4 bytes
        "v_lshl_add_u32 v103, v105, 2, v103\n"
        "v_lshl_add_u32 v105, v105, 2, %[c96]\n"                // a clamp lane reads a descriptor instead: L (quads 0, 1), R (quads 2, 3)
        "s_waitcnt lgkmcnt(0)\n"
        "v_xor_b32_e32 %[mj], v104, %[mj]\n"                    // far quads: -(1 / d)
        "v_add_u32_e32 %[ofs], %[ofs], v103\n"
        "v_cndmask_b32_e64 %[ofs], %[ofs], v105, %[s_pad]\n"
        "v_add_u32_e32 %[below], -1, %[rclosest]\n"
        "v_xor_b32_e32 v104, 0x80000000, %[below]\n"
        "v_mov_b32_e32 %[cj], 1\n"                              // denorm_min
        "v_cndmask_b32_e64 %[cj], %[cj], v104, %[s_farpad]\n"   // -below
        "v_cndmask_b32_e64 %[mj1], %[mj], 0, %[s_pad]\n"        // the one-ray form below: a clamp lane reads the plane of the lane before it, ...
        "v_mov_b32_e32 %[cj1], 0x80000000\n"
        "v_cndmask_b32_e64 %[cj1], %[cj1], %[cj], %[s_pad]\n"   // ... multiplies by 0 and adds its clamp; a plane lane adds -0
        "v_mov_b32_dpp %[ofs1], %[ofs] row_shr:1 row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_cndmask_b32_e64 %[ofs1], %[ofs], %[ofs1], %[s_pad]\n"
        // ---- ONE ray (the tail of a pixel chain): its steps are one dependent chain, not an issue budget -- a lone wave issues an instruction
        //      every ~5 cycles but a step of the loop below takes ~450, most of it the hand-offs between the vector pipe, the scalar unit
        //      and EXEC in its decision part.  This form keeps EXEC fixed (the row) and decides with selects: the three records the walk can
        //      go to next (child L, child R, the stack top) have their LDS addresses ready before the slab test is done, the push rule picks
        //      one and the next reads go out; the only scalar instruction of a step is its back edge.
        "s_cmp_lg_u32 %[n4], 16\n"
        "s_cbranch_scc1 3f\n"
        "v_cmp_lt_i32_e32 vcc, -1, %[rcur]\n"
        "s_cbranch_vccz 14f\n"
        "10:\n"                                                  // ---- interior steps while the ray stays at interior nodes
        "v_mul_u32_u24_e32 v113, %[rec_bytes], %[rcur]\n"       // byte address of the record (they start at LDS address 0)
        "v_mov_b32_e32 v112, 0xffffffc0\n"                      // -64
        "v_add_u32_e32 v111, v113, %[ofs1]\n"
        "11:\n"
        BRT_COUNT_INT1
        "ds_read_b32 v100, v111\n"                              // this lane's plane (a clamp lane: any plane, times 0)
        "ds_read_b64 v[108:109], v113 offset:96\n"              // children L, R
        "ds_read_i16 v110, %[rspa]\n"                           // the would-be pop
        "s_waitcnt lgkmcnt(2)\n"
        "v_sub_f32_e32 v101, v100, %[oj]\n"                     // (b - o) * (1 / d), raytrace.wgsl:388-390; a plane lane adds -0 (exact), a clamp
        "v_fma_f32 v101, v101, %[mj1], %[cj1]\n"                // lane multiplies a finite number by 0 and adds its clamp
        "s_nop 1\n"
        "v_max_f32_dpp v102, v101, v101 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_max_f32_dpp v103, v102, v102 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_add_f32_dpp v104, v103, v103 row_shr:4 row_mask:0xf bank_mask:0xf\n"   // quads 1, 3: max(t_near, denorm_min) - min(t_far, below)
        "s_nop 1\n"
        "v_mov_b32_dpp v105, v104 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"
        "v_mov_b32_dpp v104, v104 row_newbcast:12 row_mask:0xf bank_mask:0xf\n"
        "v_cmp_ge_f32_e32 vcc, 0, v105\n"                       // p1: t_near <= t_far of child L (x - y <= 0 iff x <= y: no flushing, y finite)
        "v_cmp_ge_f32_e64 %[s_p2], 0, v104\n"                   // p2
        "s_waitcnt lgkmcnt(0)\n"
        "ds_write_b16 %[rspa], v108 offset:128\n"               // child L above the top: dead unless both are pushed
        "v_cndmask_b32_e32 %[rcur], v110, v108, vcc\n"          // neither: pop; only L: L
        "v_cndmask_b32_e64 v105, v112, 64, vcc\n"               // stack top: + 64 + 64 (both), + 64 - 64 (one), - 64 - 64 (neither)
        "v_cndmask_b32_e64 %[rcur], %[rcur], v109, %[s_p2]\n"   // R, if pushed, is popped first
        "v_cndmask_b32_e64 v104, v112, 64, %[s_p2]\n"
        "v_mul_u32_u24_e32 v113, %[rec_bytes], %[rcur]\n"
        "v_add3_u32 %[rspa], %[rspa], v104, v105\n"
        "v_add_u32_e32 v111, v113, %[ofs1]\n"                   // this lane's word of the next record
        "v_cmp_lt_i32_e32 vcc, -1, %[rcur]\n"
        "s_cbranch_vccnz 11b\n"
        "14:\n"                                                  // ---- a leaf, or the end of the walk
        "v_cmp_gt_i32_e32 vcc, -1, %[rcur]\n"
        "s_cbranch_vccz 7f\n"
        "s_mov_b64 %[s_take], exec\n"
        BRT_COUNT_LEAF
        BRT_ROWS_LEAF_STEP
        "v_cndmask_b32_e64 %[cj1], %[cj1], %[cj], %[s_farpad]\n"   // (closest may have changed: the far clamp)
        "v_cmp_lt_i32_e32 vcc, -1, %[rcur]\n"
        "s_cbranch_vccnz 10b\n"
        "s_branch 14b\n"
        // ---- loop (two to four rays) ------------------------------------------------------------------------------------------------------
        "3:\n"
        "v_cmp_lt_i32_e32 vcc, -1, %[rcur]\n"                   // rows at an interior node
        "s_and_b64 %[s_take], vcc, %[s_rows]\n"
        "s_cbranch_scc0 4f\n"
        "s_mov_b64 exec, %[s_take]\n"
        BRT_COUNT_INT
        "ds_read_i16 v110, %[rspa]\n"                           // the would-be pop
        "v_mul_u32_u24_e32 v111, %[rec_bytes], %[rcur]\n"
        "v_add_u32_e32 v111, v111, %[ofs]\n"
        "ds_read_b32 v100, v111\n"                              // this lane's plane (clamp lanes: a child descriptor)
        "s_waitcnt lgkmcnt(0)\n"
        "v_sub_f32_e32 v101, v100, %[oj]\n"                     // (b - o) * (1 / d), raytrace.wgsl:388-390
        "v_mul_f32_e32 v101, v101, %[mj]\n"
        "v_cndmask_b32_e64 v101, v101, %[cj], %[s_pad]\n"
        "v_mov_b32_dpp v108, v100 row_newbcast:7 row_mask:0xf bank_mask:0xf\n"      // child L
        "v_mov_b32_dpp v109, v100 row_newbcast:15 row_mask:0xf bank_mask:0xf\n"     // child R
        "v_max_f32_dpp v102, v101, v101 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
        "ds_write_b16 %[rspa], v108 offset:128\n"               // child L above the top: dead unless both are pushed
        "s_nop 0\n"
        "v_max_f32_dpp v103, v102, v102 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
        "s_nop 1\n"
        "v_mov_b32_dpp v104, v103 row_newbcast:0 row_mask:0xf bank_mask:0xf\n"      // max(t_near L, denorm_min)
        "v_mov_b32_dpp v105, v103 row_newbcast:4 row_mask:0xf bank_mask:0xf\n"      // -min(t_far L, below)
        "v_mov_b32_dpp v106, v103 row_newbcast:8 row_mask:0xf bank_mask:0xf\n"
        "v_mov_b32_dpp v107, v103 row_newbcast:12 row_mask:0xf bank_mask:0xf\n"
        "v_cmp_le_f32_e64 vcc, v104, -v105\n"                   // p1: child L is pushed (raytrace.wgsl:331)
        "v_cmp_le_f32_e64 %[s_p2], v106, -v107\n"               // p2: child R is pushed (raytrace.wgsl:338)
        "s_or_b64 %[s_any], vcc, %[s_p2]\n"
        "s_and_b64 %[s_both], vcc, %[s_p2]\n"
        "s_andn2_b64 exec, %[s_take], %[s_any]\n"               // no child pushed: pop
        "v_mov_b32_e32 %[rcur], v110\n"
        "v_add_u32_e32 %[rspa], 0xffffff80, %[rspa]\n"
        "s_andn2_b64 exec, vcc, %[s_p2]\n"                      // only L
        "v_mov_b32_e32 %[rcur], v108\n"
        "s_mov_b64 exec, %[s_p2]\n"                             // R (pushed last, popped first)
        "v_mov_b32_e32 %[rcur], v109\n"
        "s_mov_b64 exec, %[s_both]\n"                           // both: L stays on the stack
        "v_add_u32_e32 %[rspa], 0x80, %[rspa]\n"
        "s_mov_b64 exec, %[s_rows]\n"
        // ---- leaf step for the rows that wait at a leaf -----------------------------------------------------------------------------
        "4:\n"
        "v_cmp_gt_i32_e32 vcc, -1, %[rcur]\n"                   // leaf descriptors are < -1
        "s_and_b64 %[s_take], vcc, %[s_rows]\n"
        "s_cbranch_scc0 6f\n"
        "s_mov_b64 exec, %[s_take]\n"
        BRT_COUNT_LEAF
        BRT_ROWS_LEAF_STEP
        "6:\n"
        "v_cmp_ne_u32_e32 vcc, -1, %[rcur]\n"
        "s_cbranch_vccnz 3b\n"
        "7:\n"
        // ---- results back to the walking lanes -------------------------------------------------------------------------------------
        "ds_write_b32 %[rowb], %[rclosest] offset:28\n"
        "ds_write_b32 %[rowb], %[rcidx] offset:44\n"
        "s_mov_b64 exec, %[s_walk]\n"
        "ds_read_b32 %[closest], %[wa] offset:28\n"
        "ds_read_b32 %[cidx], %[wa] offset:44\n"
        "v_mov_b32_e32 %[cur], -1\n"
        "s_waitcnt lgkmcnt(0)\n"
        "s_mov_b64 exec, %[s_all]\n"
        : [cur] "+v"(cur), [closest] "+v"(closest), [cidx] "+v"(closest_idx), [wa] "=&v"(wa), [rowb] "=&v"(rowb), [ofs] "=&v"(ofs),
          [sphk] "=&v"(sphk), [rcur] "=&v"(rcur), [rspa] "=&v"(rspa), [rcidx] "=&v"(rcidx), [below] "=&v"(below), [lane] "=&v"(lane),
          [oj] "=&v"(oj), [mj] "=&v"(mj), [dj] "=&v"(dj), [cj] "=&v"(cj), [ra] "=&v"(ra), [rclosest] "=&v"(rclosest), [mj1] "=&v"(mj1), [cj1] "=&v"(cj1), [ofs1] "=&v"(ofs1), [n4] "=&s"(n4),
          [s_all] "=&s"(s_all), [s_walk] "=&s"(s_walk), [s_rows] "=&s"(s_rows), [s_take] "=&s"(s_take), [s_p2] "=&s"(s_p2),
          [s_any] "=&s"(s_any), [s_both] "=&s"(s_both), [s_pad] "=&s"(s_pad), [s_farpad] "=&s"(s_farpad)
#if BRT_ASM_COUNT
          , [c_ie] "=&s"(c_ie), [c_il] "=&s"(c_il), [c_le] "=&s"(c_le), [c_ll] "=&s"(c_ll), [c_tmp] "=&s"(c_tmp)
#endif
        : [spa] "v"(spa), [gofs_x] "v"(gofs_x), [gofs_y] "v"(gofs_y), [gofs_z] "v"(gofs_z), [ox] "v"(o.x), [oy] "v"(o.y), [oz] "v"(o.z),
          [ix] "v"(inv.x), [iy] "v"(inv.y), [iz] "v"(inv.z), [dx] "v"(d.x), [dy] "v"(d.y), [dz] "v"(d.z), [a] "v"(a), [sph] "s"(sph),
          [scratch] "s"(scratch), [rec_bytes] "s"(rec_bytes), [c_tiny] "s"(c_tiny), [c_eps] "s"(c_eps), [c_cls] "s"(c_cls), [c80] "s"(c80),
          [c96] "s"(96u)
        : "vcc", "scc", "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112",
          "v113");
#if BRT_ASM_COUNT
    if (first_active_lane()) { ac->int_exec += c_ie; ac->int_lanes += c_il; ac->leaf_exec += c_le; ac->leaf_lanes += c_ll; ac->rows_int_exec += c_ie; ac->rows_calls += 1u; }
#endif
#undef BRT_COUNT_INT
#undef BRT_COUNT_LEAF
#endif
}


template <bool D16, bool SIMPLE_TREE, typename StackT>
BRT_DEV void walk_loop_wave_lds(const ScenePtrs& sc, f3 o, f3 d, float a, f3 inv, uint32_t ox, uint32_t oy, uint32_t oz,
                                float& closest, uint32_t& closest_idx, uint32_t& cur, StackT*& sp, uint32_t& n,
                                uint32_t exit_at, uint32_t vote, HitCounters& hc) {
    static_assert(D16 && SIMPLE_TREE && sizeof(StackT) == 2, "LDS-resident simple tree: 16-bit descriptors, no overflow rule, no leaf table");
    typedef __attribute__((address_space(3))) StackT lds_stack;
    uint32_t spa = (uint32_t)(uintptr_t)(lds_stack*)sp;
    const uint32_t sph = (uint32_t)__builtin_amdgcn_readfirstlane((int)sc.sph_base);
    // (the leaf step runs when at least `vote` of the lanes that walked at the last leaf step wait at a leaf -- walk_loop_wave
    //  counts the leaf lanes instead; the rules differ only when a lane ends its walk inside a run of interior steps, and only
    //  in when the leaf step runs)
#if BRT_ASM_COUNT
    const unsigned long long t0 = __builtin_readcyclecounter();
#endif
    walk_wave_lds_asm(cur, spa, closest, closest_idx, ox, oy, oz, o, inv, d, a, sph, exit_at, vote, &hc.asm_counts);
#if BRT_ASM_COUNT
    if (first_active_lane()) hc.asm_counts.wide_cycles += (uint32_t)(__builtin_readcyclecounter() - t0);
#endif
    sp = (StackT*)reinterpret_cast<lds_stack*>((uintptr_t)spa);
}
// ... and of a wave that walks at most four rays: every walk to its end, a ray to a row of 16 lanes (walk_rows_asm)
template <bool D16, bool SIMPLE_TREE, typename StackT>
BRT_DEV void walk_rows_lds(const ScenePtrs& sc, f3 o, f3 d, float a, f3 inv, uint32_t ox, uint32_t oy, uint32_t oz,
                           float& closest, uint32_t& closest_idx, uint32_t& cur, StackT* sp, HitCounters& hc) {
    static_assert(D16 && SIMPLE_TREE && sizeof(StackT) == 2, "LDS-resident simple tree: 16-bit descriptors, no overflow rule, no leaf table");
    typedef __attribute__((address_space(3))) StackT lds_stack;
    const uint32_t spa = (uint32_t)(uintptr_t)(lds_stack*)sp;
    const uint32_t sph = (uint32_t)__builtin_amdgcn_readfirstlane((int)sc.sph_base);
    const uint32_t scratch = (uint32_t)__builtin_amdgcn_readfirstlane((int)sc.rows_scratch);
#if BRT_ASM_COUNT
    const unsigned long long t0 = __builtin_readcyclecounter();
#endif
    walk_rows_asm(cur, spa, closest, closest_idx, ox, oy, oz, o, inv, d, a, sph, scratch, &hc.asm_counts);
#if BRT_ASM_COUNT
    if (first_active_lane()) hc.asm_counts.rows_cycles += (uint32_t)(__builtin_readcyclecounter() - t0);
#endif
}

// ... and of a scene walked from the tile / from global memory (walk_wave_top_asm)
template <bool D16, bool SIMPLE_TREE, int MODE, typename StackT>
BRT_DEV void walk_loop_wave_top(const ScenePtrs& sc, f3 o, f3 d, float a, f3 inv, uint32_t ox, uint32_t oy, uint32_t oz,
                                float& closest, uint32_t& closest_idx, uint32_t& cur, StackT*& sp, uint32_t& n,
                                uint32_t exit_at, uint32_t vote, HitCounters& hc) {
    static_assert(D16 && SIMPLE_TREE && sizeof(StackT) == 2 && MODE != SCENE_LDS, "16-bit descriptors, no overflow rule, no leaf table");
    typedef __attribute__((address_space(3))) StackT lds_stack;
    uint32_t spa = (uint32_t)(uintptr_t)(lds_stack*)sp;
    const uint32_t near_bytes = MODE == SCENE_LDS_TOP ? (uint32_t)__builtin_amdgcn_readfirstlane((int)sc.near_bytes) : 0u;
    const uint64_t far = (uint64_t)(uintptr_t)sc.pairs_far, sphg = (uint64_t)(uintptr_t)sc.spheres;
    walk_wave_top_asm(cur, spa, closest, closest_idx, ox, oy, oz, o, inv, d, a, near_bytes, far, sphg, exit_at, vote, &hc.asm_counts);
    sp = (StackT*)reinterpret_cast<lds_stack*>((uintptr_t)spa);
}

// HITS: the instantiation can count interior visits per record (sc.hits; the pre-pass of a SCENE_LDS_TOP scene): such a launch walks
// in the compiler's loop, the repairing form (a superset of the plain one: the min / max it adds change nothing for safe rays).
// POLICY: the instantiation can walk under the compare-select reading of min / max (sc.minmax_select): also in the compiler's loop.
// ROWS: a wave that walks at most four rays takes the row-mode loop (walk_rows_asm).  The kernel asks for it in the instantiations
// that can hold a CRITICAL pixel chain (LEAN 0 / 1: config 3 ends in one 11 064-ray pixel); the steady-state instantiation of views
// without such chains (LEAN 2: the headline frame) leaves it out -- there the thin phase of a wave is too short to pay for the second
// loop's registers and dispatch (same box, headline frame: 8.885 ms without, 9.026 ms with; profiles/r06/thin_wave_rows.txt).
template <int STRIDE, bool COUNTERS, bool D16, bool SIMPLE_TREE, int MODE, typename StackT, bool HITS = false, bool POLICY = false, bool ROWS = false>
BRT_DEV void walk_run(const ScenePtrs& sc, WalkState<StackT>& w, StackT* stk, f3 o, f3 d,
                      uint32_t exit_lanes, uint32_t leaf_vote, HitCounters& hc) {
    using DS = Desc<D16>;
    const float a = w.a;
    const f3 inv = w.inv;
    float closest = w.closest;
    uint32_t closest_idx = w.closest_idx;
    uint32_t cur = w.cur;
    StackT* sp = w.sp;
    uint32_t n = w.n;
    const uint32_t ox = w.ox, oy = w.oy, oz = w.oz;
    const bool pending = cur != DS::DONE && (SIMPLE_TREE || n < 31u);
    const bool unsafe = !sc.boxes_ordered || (pending && !ray_is_safe(o, inv));
    if (STRIDE == 64) {
        // Wave-level loop.  The kernel is bound by instruction issue, the two bodies cost the wave ~60-70
        // instructions each however few lanes take part, and a ray needs ~6 interior steps per leaf step:
        //   inner loop   interior steps only, until `leaf_vote` lanes wait at a leaf or no lane has an
        //                interior node left (a waiting lane sits those iterations out);
        //   then         ONE leaf step for all waiting lanes -- a lane that pops an interior node there
        //                continues in the next inner iteration;
        //   leave        when no more than exit_at lanes still walk.  exit_at < the number that entered,
        //                so every call makes progress; exit_lanes == 0 runs all walks to the end.
        // Only the interleaving of lanes changes, never a lane's own sequence of steps.  With the signed
        // descriptor form (brt_layout.h) a finished lane fails both body tests by itself.
        // The loop exists twice: without and with the min/max repair of the near/far reads
        // (walk_interior_step); the wave takes the second one only when one of its rays needs it.
        const uint32_t n_walking = wave_count(pending);
        // (readfirstlane: both thresholds are wave-uniform by construction, but the compiler cannot see that through the
        //  drain logic they come from, and a loop exit on a "divergent" value is compiled with exec-mask bookkeeping)
        uint32_t exit_at = n_walking >> 1;
        exit_at = exit_at < exit_lanes ? exit_at : exit_lanes;
        exit_at = (uint32_t)__builtin_amdgcn_readfirstlane((int)exit_at);
        const uint32_t vote = (uint32_t)__builtin_amdgcn_readfirstlane((int)(leaf_vote < 1u ? 1u : leaf_vote));
        if (n_walking > exit_at) {
            // by hand where the pair records start at LDS address 0: the kernel has no static LDS, which launch_persistent_t
            // (brt_trace.h) checks on the host -- a run-time test here would give the loop's results two homes and a copy each.
            // For the same reason the (rare) wave with an unsafe ray does not take the repairing loop INSTEAD of the hand-written one
            // but BEFORE it: that loop leaves with at most exit_at lanes walking, and the hand-written one then returns at its first test.
            constexpr bool kByHand = BRT_WALK_FAST && MODE == SCENE_LDS && !COUNTERS && SIMPLE_TREE && D16;
            const bool counting = HITS && D16 && sc.hits != nullptr;           // (wave-uniform: a kernel argument)
            const bool select = POLICY && sc.minmax_select;                    // (likewise)
            const bool any_unsafe = __ballot(unsafe) != 0ull || counting || select;
            if (select)
                walk_loop_wave<COUNTERS, D16, SIMPLE_TREE, true, MODE, StackT, false, POLICY>(sc, o, d, a, inv, ox, oy, oz, closest, closest_idx, cur, sp,
                                                                                             n, exit_at, vote, hc);
            else if (counting)
                walk_loop_wave<COUNTERS, D16, SIMPLE_TREE, true, MODE, StackT, HITS && D16>(sc, o, d, a, inv, ox, oy, oz, closest, closest_idx, cur, sp,
                                                                                            n, exit_at, vote, hc);
            else if (any_unsafe)
                walk_loop_wave<COUNTERS, D16, SIMPLE_TREE, true, MODE>(sc, o, d, a, inv, ox, oy, oz, closest, closest_idx, cur, sp,
                                                                       n, exit_at, vote, hc);
            constexpr bool kByHandTop = BRT_WALK_FAST && BRT_WALK_FAST_TOP && MODE != SCENE_LDS && !COUNTERS && SIMPLE_TREE && D16;
            // a thin wave (the tail of a pixel chain): a ray to a row of 16 lanes, every walk to its end (walk_rows_asm)
            constexpr bool kRows = kByHand && BRT_WALK_ROWS && ROWS;
            bool rows = false;
            if constexpr (kRows) rows = n_walking <= 4u && !any_unsafe && sc.rows_scratch != 0u;       // (wave-uniform)
            if constexpr (kByHand) {
                if (rows) walk_rows_lds<D16, SIMPLE_TREE>(sc, o, d, a, inv, ox, oy, oz, closest, closest_idx, cur, sp, hc);
                else walk_loop_wave_lds<D16, SIMPLE_TREE>(sc, o, d, a, inv, ox, oy, oz, closest, closest_idx, cur, sp, n, exit_at, vote, hc);
            }
            else if constexpr (kByHandTop)
                walk_loop_wave_top<D16, SIMPLE_TREE, MODE>(sc, o, d, a, inv, ox, oy, oz, closest, closest_idx, cur, sp, n, exit_at, vote, hc);
            else if (!any_unsafe)
                walk_loop_wave<COUNTERS, D16, SIMPLE_TREE, false, MODE>(sc, o, d, a, inv, ox, oy, oz, closest, closest_idx, cur, sp,
                                                                        n, exit_at, vote, hc);
        }
    } else {
        while (cur != DS::DONE && (SIMPLE_TREE || n < 31u)) {
            if (DS::is_leaf(cur))
                walk_leaf_step<STRIDE, COUNTERS, D16, SIMPLE_TREE>(sc, o, d, a, closest, closest_idx, cur, sp, n, hc);
            else if (unsafe)
                walk_interior_step<STRIDE, COUNTERS, true, D16, MODE>(sc, o, inv, ox, oy, oz, float_below(closest), cur, sp, n, hc);
            else
                walk_interior_step<STRIDE, COUNTERS, false, D16, MODE>(sc, o, inv, ox, oy, oz, float_below(closest), cur, sp, n, hc);
        }
    }
    w.closest = closest;
    w.closest_idx = closest_idx;
    w.cur = cur;
    w.sp = sp;
    w.n = n;
}

// whole walk in one call (bring-up kernel, probes)
template <int STRIDE, bool COUNTERS, bool D16, bool SIMPLE_TREE, typename StackT>
BRT_DEV void raycast(const ScenePtrs& sc, uint32_t root_desc, StackT* stk, f3 o, f3 d,
                     float& t_out, uint32_t& idx_out, HitCounters& hc) {
    WalkState<StackT> w;
    walk_begin<D16>(w, sc, root_desc, stk, d);
    walk_run<STRIDE, COUNTERS, D16, SIMPLE_TREE, SCENE_GLOBAL>(sc, w, stk, o, d, 0u, 0u, hc);
    t_out = w.closest;
    idx_out = w.closest_idx;
}

// raytrace.wgsl:400-402
BRT_DEV f3 reflect3(f3 v, f3 n) { return v - (2.0f * dot3(v, n)) * n; }
// raytrace.wgsl:404-409
BRT_DEV f3 refract3(f3 v, f3 n, float eta, bool select = false) {
    const float cos_theta = select ? min_sel(dot3(neg3(v), n), 1.0f) : min_f(dot3(neg3(v), n), 1.0f);
    const f3 perp = eta * (v + cos_theta * n);
    const float par = -__builtin_sqrtf(__builtin_fabsf(1.0f - dot3(perp, perp)));
    return perp + par * n;
}
// raytrace.wgsl:411-416
// pow5: pow(x, 5.0) as exp2(5 * log2(x)) in f64, rounded to f32 once (BRT_POLICY_POW_EXP2_LOG2: what WGSL defines pow as; the
// oracle's pow_exp2_log2 policy) instead of the multiplies
BRT_DEV float schlick(float cosine, float ri, bool pow5 = false) {
    float r0 = (1.0f - ri) / (1.0f + ri);
    r0 = r0 * r0;
    const float x = 1.0f - cosine;
    float p5;
    if (pow5) {
        if (x != x || x < 0.0f) p5 = __builtin_nanf("");
        else if (x == 0.0f) p5 = 0.0f;
        else if (__builtin_isinf(x)) p5 = __builtin_inff();
        else p5 = (float)exp2(5.0 * log2((double)x));
    } else {
        const float x2 = x * x;
        p5 = (x2 * x2) * x;
    }
    return r0 + (1.0f - r0) * p5;
}

// raytrace.wgsl:231-299 applied to the hit (t, idx) of ray (o,d); the HitInfo fields are
// rebuilt here from (t, idx) exactly as raytrace.wgsl:355-358 builds them.
// Returns absorbed; writes the scattered ray and the attenuation.
//
// Written for a divergent wave: the three material branches of the shader would each carry
// their own copy of the rejection sampler and of normalize().  Here the lottery first fixes
// the kind, then ONE rejection loop serves every lane that still needs a ball (metal 1,
// diffuse 2, glass 0) and ONE normalize serves metal (reflected direction) and glass (incoming
// direction).  Per lane the RNG draws and the arithmetic are the shader's, in its order.
// or_short_circuit: the alternative reading of raytrace.wgsl:269 (no RNG draw when cannot_refract); only the
// TUNABLE kernel instantiation can switch it on (BRT_POLICY_OR_SHORT_CIRCUIT=1), for the alternative-policy
// fixtures -- the product's policy is "always draw" (DESIGN.md section 2).
template <bool COUNTERS>
BRT_DEV bool scatter(const ScenePtrs& sc, f3& o, f3& d, float t, uint32_t idx, uint32_t& rng, f3& attenuation, HitCounters& hc,
                     bool or_short_circuit = false) {
    const float4 s = sc.spheres[idx];
    const f3 pos = mk3(o.x + t * d.x, o.y + t * d.y, o.z + t * d.z);          // ray_at, :130-132
    const f3 nrm = normalize3(mk3(pos.x - s.x, pos.y - s.y, pos.z - s.z));    // :356
    const float4 m0 = sc.sphere_mats[2 * idx];      // base_color.rgb, metallic
    const float4 m1 = sc.sphere_mats[2 * idx + 1];  // roughness, reflectance, ior, specular_transmission
    const bool metal = rng_float(rng) < m0.w;                                  // :234
    const bool glass = !metal && (rng_float(rng) < m1.w);                      // :249 (drawn only when not metal)
    const bool diffuse = !metal && !glass;

    // balls: metal fuzz = roughness * ball (:238); diffuse normal + ball + roughness * ball (:285)
    uint32_t need = metal ? 1u : (diffuse ? 2u : 0u);
    // Branch-free accept with ONE running sum: candidate = acc + scale * p.  Diffuse starts from
    // (acc, scale) = (normal, 1) -- 1 * p == p -- and continues with (normal + ball, roughness); metal
    // starts from (-0, roughness): adding to -0 returns the other operand bit for bit, signed zeros
    // included.  Each accepted candidate is the shader's own expression.
    f3 acc = diffuse ? nrm : mk3(-0.0f, -0.0f, -0.0f);
    float scale = diffuse ? 1.0f : m1.x;
    unsigned long long t_ball = 0;
    if (COUNTERS) t_ball = wall_clock64();
    while (need != 0u) {                                                      // random.wgsl:19-24
        if (COUNTERS) prof_section<COUNTERS>(hc, SEC_BALL, true);
        const float px = rng_ball_coord(rng);
        const float py = rng_ball_coord(rng);
        const float pz = rng_ball_coord(rng);
        const f3 p = mk3(px, py, pz);
        const bool ok = dot3(p, p) <= 1.0f;
        const f3 cand = acc + scale * p;
#if BRT_EXEC_MOVES & 2
        if (ok) {
            acc = cand;
            scale = m1.x;
            need -= 1u;
            asm volatile("" : "+v"(acc.x), "+v"(acc.y), "+v"(acc.z), "+v"(scale), "+v"(need));
        }
#else
        acc = mk3(ok ? cand.x : acc.x, ok ? cand.y : acc.y, ok ? cand.z : acc.z);
        scale = ok ? m1.x : scale;
        need -= ok ? 1u : 0u;
#endif
    }
    if (COUNTERS) {   // booked by the first lane of the section, like prof_section
        const uint64_t m = __ballot(true);
        if (__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u)) == 0u)
            hc.ticks_ball += wall_clock64() - t_ball;
    }

    bool absorbed = false;
    attenuation = mk3(m0.x, m0.y, m0.z);
    if (diffuse) {                                                            // :281-297
        const float eps = 1e-8f;
        if (__builtin_fabsf(acc.x) < eps && __builtin_fabsf(acc.y) < eps && __builtin_fabsf(acc.z) < eps) acc = nrm;
        absorbed = dot3(acc, nrm) < 0.0f;
        d = acc;
    } else {
        const bool front_face = dot3(d, nrm) < 0.0f;                          // :358 (incoming direction)
        const f3 u = normalize3(metal ? reflect3(d, nrm) : d);                // :238 / :261
        if (metal) {                                                          // :234-245
            d = u + acc;
            absorbed = dot3(d, nrm) < 0.0f;
        } else {                                                              // glass, :249-280
            const float ri = front_face ? (1.0f / m1.z) : m1.z;
            const float cos_theta = min_f(dot3(neg3(u), nrm), 1.0f);
            const float sin_theta = __builtin_sqrtf(1.0f - cos_theta * cos_theta);
            const bool cannot_refract = ri * sin_theta > 1.0f;
            const float refl = schlick(cos_theta, ri);
            bool reflects = cannot_refract;
            if (!(or_short_circuit && cannot_refract)) {
                const float draw = rng_float(rng);                            // default policy: always drawn
                reflects = reflects || (refl > draw);
            }
            d = reflects ? reflect3(u, nrm) : refract3(u, nrm, ri);
            attenuation = mk3(1.0f, 1.0f, 1.0f);
        }
    }
    o = pos;
    return absorbed;
}

// raytrace.wgsl:364-369
BRT_DEV f3 background_gradient(f3 d) {
    const f3 u = normalize3(d);
    const float a = 0.5f * (u.y + 1.0f);
    const float b = 1.0f - a;
    return mk3(b * 1.0f + a * 0.5f, b * 1.0f + a * 0.7f, b * 1.0f + a * 1.0f);
}

// raytrace.wgsl:95 (seed) -- uv at the pixel centre
BRT_DEV uint32_t pixel_seed(const FrameParams& fp, float uvx, float uvy) {
    return f32_to_u32_sat((fp.seed_scaled * (uvx * 402.0f)) * (uvy * 31.5f));
}

// raytrace.wgsl:139-156 with frame-uniform terms hoisted into FrameParams.
// ndc0x = uv.x*2-1, ndc0y = 1-uv.y*2.
// camera_dir_raw: the direction before normalize() (the persistent kernel normalises it together with the
// reflected / refracted directions of the same round: shade_landed, brt_trace.h)
BRT_DEV f3 camera_dir_raw(const FrameParams& fp, float ndc0x, float ndc0y, uint32_t& rng) {
    const float rx = rng_float(rng) - 0.5f;
    const float ry = rng_float(rng) - 0.5f;
    const float ndc_x = ndc0x + fp.inv_width * rx;
    const float ndc_y = ndc0y + fp.inv_height * ry;
    const float sx = (ndc_x * fp.aspect) * fp.tan_half_fov;
    const float sy = ndc_y * fp.tan_half_fov;
    const f3 cd = mk3(fp.cam_dir[0], fp.cam_dir[1], fp.cam_dir[2]);
    const f3 cr = mk3(fp.cam_right[0], fp.cam_right[1], fp.cam_right[2]);
    const f3 cu = mk3(fp.cam_up[0], fp.cam_up[1], fp.cam_up[2]);
    return (cd + sx * cr) + sy * cu;
}
BRT_DEV f3 camera_ray_dir(const FrameParams& fp, float ndc0x, float ndc0y, uint32_t& rng) {
    return normalize3(camera_dir_raw(fp, ndc0x, ndc0y, rng));
}

// raytrace.wgsl:104-122: final colour of a pixel from the averaged sample colour/depth.
BRT_DEV float4 resolve_pixel(const FrameParams& fp, f3 avg, float avg_depth, const float* raster_rgba,
                             const float* raster_depth, size_t frame_pix) {
    if (fp.level == 1u || fp.level == 2u) {
        const float depth = raster_depth ? raster_depth[frame_pix] : 0.0f;
        float rd = avg_depth;
        if (rd > fp.far_) rd = -1.0f;
        else rd = fp.near_ / rd;
        if (depth > rd) {
            if (raster_rgba) return reinterpret_cast<const float4*>(raster_rgba)[frame_pix];
            return make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
    }
    return make_float4(avg.x, avg.y, avg.z, 1.0f);
}

}  // namespace brt
