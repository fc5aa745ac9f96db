// pair_step_bench.hip -- what would a LANE-PAIR interior step buy a thin wave?  (development aid, not part of the product)
//
//   hipcc --offload-arch=gfx950 -O3 -o pair_step_bench tests/tools/pair_step_bench.hip && ./pair_step_bench
//
// VERDICT r3 task 3: for waves with <= 32 live paths let lane l take child L and lane l + 32 child R of the same ray -- 12 sub /
// mul, 4 min / max and one compare per lane instead of 24 / 8 / 2.  Before rebuilding the walk around it, the two forms of the
// interior step are timed here in isolation: the production step of walk_wave_lds_asm (brt_device.h) verbatim, and the pair
// step, each walking a chain of pair records in LDS (child L always pushed, child R never: one `v_mov cur, descL` per step,
// the stack pointer stands still), as ONE ray in one wave on an otherwise idle CU (the critical chain of config 3, the last
// waves of a rank's share at 8 GPUs) and with 16 such waves on the CU (4 per SIMD).  Time per step from HIP events.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int NREC = 256;          // 256 x 112 B = 28 KB at LDS address 0
constexpr unsigned REC = 112;

__device__ __forceinline__ void fill_records(float* lds) {
    for (int r = threadIdx.x; r < NREC; r += blockDim.x) {
        float* rec = lds + r * 28;
        for (int k = 0; k < 3; k++) {
            float* g = rec + 8 * k;           // G0 = { min L, min R, max L, max R }, G1 = { max L, max R, min L, min R }
            g[0] = -1e3f; g[1] = 1e6f; g[2] = 1e3f; g[3] = 2e6f;
            g[4] = 1e3f; g[5] = 2e6f; g[6] = -1e3f; g[7] = 1e6f;
        }
        reinterpret_cast<unsigned*>(rec)[24] = (unsigned)((r * 37 + 11) % NREC);   // desc L: the next record of the chain
        reinterpret_cast<unsigned*>(rec)[25] = (unsigned)((r * 13 + 5) % NREC);    // desc R (never taken)
    }
}

// the production interior step (walk_wave_lds_asm, brt_device.h), `steps` times
__global__ void __launch_bounds__(1024) k_full(unsigned steps, unsigned* out, unsigned long long lanes) {
    extern __shared__ float lds[];
    fill_records(lds);
    __syncthreads();
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned cur = lane % NREC, spa = NREC * REC + wave * 4096u + lane * 2u + 256u, t0, tx, ty, tz, pop, cnt, n = steps;
    float closest = 1e5f, below;
    const float ox = 0.0f, oy = 0.0f, oz = 0.0f, ix = 1.0f, iy = 1.0f, iz = 1.0f;
    const unsigned gx = 0u, gy = 32u, gz = 64u, rec_bytes = REC, thr = 0u;
    unsigned long long s_all, s_take, s_p2, s_any, s_both;
    asm volatile(
        "s_mov_b64 exec, %[lanes]\n"
        "s_mov_b64 %[s_all], exec\n"
        "v_add_u32_e32 %[below], -1, %[closest]\n"
        "v_cmp_lt_i32_e32 vcc, -1, %[cur]\n"
        "1:\n"
        "s_mov_b64 %[s_take], vcc\n"
        "s_mov_b64 exec, vcc\n"
        "ds_read_i16 %[pop], %[spa]\n"
        "v_mul_lo_u32 %[t0], %[cur], %[rec_bytes]\n"
        "v_add_u32_e32 %[tx], %[t0], %[gx]\n"
        "ds_read_b128 v[100:103], %[tx]\n"
        "v_add_u32_e32 %[ty], %[t0], %[gy]\n"
        "ds_read_b128 v[104:107], %[ty]\n"
        "v_add_u32_e32 %[tz], %[t0], %[gz]\n"
        "ds_read_b128 v[108:111], %[tz]\n"
        "ds_read_b64 v[112:113], %[t0] offset:96\n"
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
        "v_max3_f32 v100, v100, v108, 1\n"
        "v_max3_f32 v101, v101, v109, 1\n"
        "v_min3_f32 v102, v102, v110, %[below]\n"
        "v_min3_f32 v103, v103, v111, %[below]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "ds_write_b16 %[spa], v112 offset:128\n"
        "v_cmp_le_f32_e32 vcc, v100, v102\n"
        "v_cmp_le_f32_e64 %[s_p2], v101, v103\n"
        "s_or_b64 %[s_any], vcc, %[s_p2]\n"
        "s_and_b64 %[s_both], vcc, %[s_p2]\n"
        "s_andn2_b64 exec, %[s_take], %[s_any]\n"
        "v_mov_b32_e32 %[cur], %[pop]\n"
        "v_add_u32_e32 %[spa], 0xffffff80, %[spa]\n"
        "s_andn2_b64 exec, vcc, %[s_p2]\n"
        "v_mov_b32_e32 %[cur], v112\n"
        "s_mov_b64 exec, %[s_p2]\n"
        "v_mov_b32_e32 %[cur], v113\n"
        "s_mov_b64 exec, %[s_both]\n"
        "v_add_u32_e32 %[spa], 0x80, %[spa]\n"
        "s_mov_b64 exec, %[s_all]\n"
        "v_cmp_lt_i32_e32 vcc, -1, %[cur]\n"
        "s_bcnt1_i32_b64 %[cnt], vcc\n"
        "s_cmp_gt_u32 %[cnt], %[thr]\n"
        "s_sub_u32 %[n], %[n], 1\n"
        "s_cmp_lg_u32 %[n], 0\n"
        "s_cbranch_scc1 1b\n"
        "s_waitcnt lgkmcnt(0)\n"
        "s_mov_b64 exec, -1\n"
        : [cur] "+v"(cur), [spa] "+v"(spa), [below] "=&v"(below), [t0] "=&v"(t0), [tx] "=&v"(tx), [ty] "=&v"(ty), [tz] "=&v"(tz), [pop] "=&v"(pop),
          [cnt] "=&s"(cnt), [n] "+s"(n), [s_all] "=&s"(s_all), [s_take] "=&s"(s_take), [s_p2] "=&s"(s_p2), [s_any] "=&s"(s_any), [s_both] "=&s"(s_both)
        : [closest] "v"(closest), [gx] "v"(gx), [gy] "v"(gy), [gz] "v"(gz), [ox] "v"(ox), [oy] "v"(oy), [oz] "v"(oz), [ix] "v"(ix), [iy] "v"(iy),
          [iz] "v"(iz), [rec_bytes] "s"(rec_bytes), [thr] "s"(thr), [lanes] "s"(lanes)
        : "vcc", "scc", "memory", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113");
    if (out) out[blockIdx.x * blockDim.x + threadIdx.x] = cur + spa;
}

// the pair step: lane l tests child L, lane l + 32 child R of the same ray; both keep the ray's walk state
__global__ void __launch_bounds__(1024) k_pair(unsigned steps, unsigned* out, unsigned long long lanes) {
    extern __shared__ float lds[];
    fill_records(lds);
    __syncthreads();
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, ray = lane & 31u, side = lane >> 5;
    unsigned cur = ray % NREC, spa = NREC * REC + wave * 4096u + ray * 2u + 256u, t0, tx, ty, tz, pop, cnt, n = steps;
    float closest = 1e5f, below;
    const float ox = 0.0f, oy = 0.0f, oz = 0.0f, ix = 1.0f, iy = 1.0f, iz = 1.0f;
    const unsigned gx = 0u + 4u * side, gy = 32u + 4u * side, gz = 64u + 4u * side, rec_bytes = REC, thr = 0u;   // {near, far} of MY child: dwords side, side + 2
    unsigned long long s_all;
    asm volatile(
        "s_mov_b64 exec, %[lanes]\n"
        "s_mov_b64 %[s_all], exec\n"
        "v_add_u32_e32 %[below], -1, %[closest]\n"
        "v_cmp_lt_i32_e32 vcc, -1, %[cur]\n"
        "1:\n"
        "s_mov_b64 s[40:41], vcc\n"                               // the lanes that take the step (both of a pair)
        "s_mov_b64 exec, vcc\n"
        "ds_read_i16 %[pop], %[spa]\n"
        "v_mul_lo_u32 %[t0], %[cur], %[rec_bytes]\n"
        "v_add_u32_e32 %[tx], %[t0], %[gx]\n"
        "ds_read2_b32 v[100:101], %[tx] offset1:2\n"              // { near, far } of my child, x
        "v_add_u32_e32 %[ty], %[t0], %[gy]\n"
        "ds_read2_b32 v[102:103], %[ty] offset1:2\n"
        "v_add_u32_e32 %[tz], %[t0], %[gz]\n"
        "ds_read2_b32 v[104:105], %[tz] offset1:2\n"
        "ds_read_b64 v[112:113], %[t0] offset:96\n"
        "s_waitcnt lgkmcnt(3)\n"
        "v_sub_f32_e32 v100, v100, %[ox]\n v_sub_f32_e32 v101, v101, %[ox]\n v_mul_f32_e32 v100, v100, %[ix]\n v_mul_f32_e32 v101, v101, %[ix]\n"
        "s_waitcnt lgkmcnt(2)\n"
        "v_sub_f32_e32 v102, v102, %[oy]\n v_sub_f32_e32 v103, v103, %[oy]\n v_mul_f32_e32 v102, v102, %[iy]\n v_mul_f32_e32 v103, v103, %[iy]\n"
        "s_waitcnt lgkmcnt(1)\n"
        "v_sub_f32_e32 v104, v104, %[oz]\n v_sub_f32_e32 v105, v105, %[oz]\n v_mul_f32_e32 v104, v104, %[iz]\n v_mul_f32_e32 v105, v105, %[iz]\n"
        "v_max_f32_e32 v100, v100, v102\n"
        "v_min_f32_e32 v101, v101, v103\n"
        "v_max3_f32 v100, v100, v104, 1\n"
        "v_min3_f32 v101, v101, v105, %[below]\n"
        "s_waitcnt lgkmcnt(0)\n"
        "ds_write_b16 %[spa], v112 offset:128\n"
        "v_cmp_le_f32_e32 vcc, v100, v101\n"                      // vcc_lo: child L pushed, vcc_hi: child R pushed, bit = ray
        "s_or_b32 s42, vcc_lo, vcc_hi\n"                          // any
        "s_and_b32 s43, vcc_lo, vcc_hi\n"                         // both
        "s_andn2_b32 s44, vcc_lo, vcc_hi\n"                       // only L
        "s_andn2_b32 exec_lo, s40, s42\n s_andn2_b32 exec_hi, s41, s42\n"     // no child pushed: pop
        "v_mov_b32_e32 %[cur], %[pop]\n"
        "v_add_u32_e32 %[spa], 0xffffff80, %[spa]\n"
        "s_mov_b32 exec_lo, s44\n s_mov_b32 exec_hi, s44\n"
        "v_mov_b32_e32 %[cur], v112\n"
        "s_mov_b32 exec_lo, vcc_hi\n s_mov_b32 exec_hi, vcc_hi\n"
        "v_mov_b32_e32 %[cur], v113\n"
        "s_mov_b32 exec_lo, s43\n s_mov_b32 exec_hi, s43\n"
        "v_add_u32_e32 %[spa], 0x80, %[spa]\n"
        "s_mov_b64 exec, %[s_all]\n"
        "v_cmp_lt_i32_e32 vcc, -1, %[cur]\n"
        "s_bcnt1_i32_b64 %[cnt], vcc\n"
        "s_cmp_gt_u32 %[cnt], %[thr]\n"
        "s_sub_u32 %[n], %[n], 1\n"
        "s_cmp_lg_u32 %[n], 0\n"
        "s_cbranch_scc1 1b\n"
        "s_waitcnt lgkmcnt(0)\n"
        "s_mov_b64 exec, -1\n"
        : [cur] "+v"(cur), [spa] "+v"(spa), [below] "=&v"(below), [t0] "=&v"(t0), [tx] "=&v"(tx), [ty] "=&v"(ty), [tz] "=&v"(tz), [pop] "=&v"(pop),
          [cnt] "=&s"(cnt), [n] "+s"(n), [s_all] "=&s"(s_all)
        : [closest] "v"(closest), [gx] "v"(gx), [gy] "v"(gy), [gz] "v"(gz), [ox] "v"(ox), [oy] "v"(oy), [oz] "v"(oz), [ix] "v"(ix), [iy] "v"(iy),
          [iz] "v"(iz), [rec_bytes] "s"(rec_bytes), [thr] "s"(thr), [lanes] "s"(lanes)
        : "vcc", "scc", "memory", "s40", "s41", "s42", "s43", "s44", "v100", "v101", "v102", "v103", "v104", "v105", "v112", "v113");
    if (out) out[blockIdx.x * blockDim.x + threadIdx.x] = cur + spa;
}

template <typename K>
double time_ns_per_step(K kern, int threads, unsigned long long lanes, unsigned steps, unsigned* d_out) {
    const size_t lds = NREC * REC + 16 * 4096 + 1024;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    double best = 1e30;
    for (int rep = 0; rep < 5; rep++) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, dim3(1), dim3(threads), lds, 0, steps, d_out, lanes);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms * 1e6 / steps < best) best = ms * 1e6 / steps;
    }
    return best;
}

int main() {
    unsigned* d_out = nullptr;
    CHECK(hipMalloc(&d_out, 1024 * 4));
    const unsigned steps = 2000000;
    std::printf("one interior step of the walk, ns per step (HIP events over %u steps; 2.4 GHz: 1 ns = 2.4 cycles)\n", steps);
    struct Case { const char* name; int threads; unsigned long long full_lanes, pair_lanes; };
    const Case cases[] = {
        {"ONE ray, one wave alone on the CU", 64, 1ull, 1ull | (1ull << 32)},
        {"8 rays, one wave alone on the CU", 64, 0xffull, 0xffull | (0xffull << 32)},
        {"32 rays, one wave alone on the CU", 64, 0xffffffffull, ~0ull},
        {"ONE ray per wave, 16 waves on the CU (4 per SIMD)", 1024, 1ull, 1ull | (1ull << 32)},
        {"32 rays per wave, 16 waves on the CU (4 per SIMD)", 1024, 0xffffffffull, ~0ull},
    };
    for (const Case& c : cases) {
        const double a = time_ns_per_step(k_full, c.threads, c.full_lanes, steps, d_out);
        const double b = time_ns_per_step(k_pair, c.threads, c.pair_lanes, steps, d_out);
        std::printf("%-52s production step %7.1f ns   pair step %7.1f ns   %+5.1f %%\n", c.name, a, b, (b / a - 1.0) * 100.0);
    }
    return 0;
}
