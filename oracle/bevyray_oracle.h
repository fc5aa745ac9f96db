/*
 * bevyray_oracle.h -- entry points of the CPU ORACLE (test infrastructure).
 * See bevyray_oracle.c for what it restates and who may call it.
 */
#ifndef BEVYRAY_ORACLE_H
#define BEVYRAY_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Render rows [row_begin,row_end) of a width x height frame into out_rgba (full-frame
 * indexing, RGBA f32, top row first).  Byte layouts as in extract.rs (32/32/48/80/16 B).
 * counters5 (optional) = {rays, node_pops, interior_visits, sphere_tests, hits}.
 * n_threads worker threads take rows from a shared counter.  Returns 0 on success. */
int oracle_render(const void* models, uint32_t n_models, const void* materials, uint32_t n_materials,
                  const void* bvh_nodes, uint32_t n_nodes, const void* camera80, const void* window16,
                  uint32_t level, uint32_t width, uint32_t height, uint32_t row_begin, uint32_t row_end,
                  const float* raster_rgba, const float* raster_depth, float* out_rgba,
                  uint64_t* counters5, int n_threads);

/* Same, but only rows row_begin, row_begin+row_step, ... < row_end (an evenly spread sample). */
int oracle_render_strided(const void* models, uint32_t n_models, const void* materials, uint32_t n_materials,
                          const void* bvh_nodes, uint32_t n_nodes, const void* camera80, const void* window16,
                          uint32_t level, uint32_t width, uint32_t height, uint32_t row_begin, uint32_t row_end,
                          uint32_t row_step, const float* raster_rgba, const float* raster_depth, float* out_rgba,
                          uint64_t* counters5, int n_threads);

/* Alternative readings of three implementation-defined points of the shader (bevyray_oracle.c, "alternative
 * policies"); all 0 = the default policy, which is what the product implements.  Process-wide. */
void oracle_set_policy(int or_short_circuit, int minmax_select, int pow_exp2_log2);

float oracle_tan_half_fov(float fov);
uint32_t oracle_rng_next(uint32_t state);
float oracle_rng_float(uint32_t* state);
uint32_t oracle_seed(float random_seed, uint32_t px, uint32_t py, uint32_t W, uint32_t H);
void oracle_unit_ball(uint32_t* state, float* out3);
float oracle_min(float a, float b);
float oracle_max(float a, float b);
float oracle_ray_bounding_dst(const float* o3, const float* d3, const float* bmin3, const float* bmax3);
float oracle_hit_sphere(const float* o3, const float* d3, const float* center3, float radius);
int oracle_raycast(const void* models, uint32_t n_models, const void* bvh_nodes, uint32_t n_nodes,
                   const float* o3, const float* d3, float* out7, uint32_t* out_material, int* out_front);

/* Diagnostic: interior visits per BVH node of the renders that follow go into per_node[n_nodes] (NULL: off). */
void oracle_set_visit_counts(uint64_t* per_node);
void oracle_set_sphere_counts(uint64_t* per_model);      /* sphere tests per model */

/* The colour target's store conversion of an RGBA f32 frame (checker side; bevyray_oracle.c): format 1 = RGBA8 sRGB
 * (4 bytes per pixel), 2 = RGBA16F (8 bytes), 3 = RGBA8 unorm (4 bytes).  Returns 0 on success. */
int oracle_encode_frame(const float* rgba, uint64_t n_pixels, int format, void* out);

#ifdef __cplusplus
}
#endif
#endif
