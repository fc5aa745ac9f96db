/*
 * exp_traversal.c -- EXPERIMENT (test infrastructure, not product): wave-level cost of BVH traversal variants.
 *
 * Simulates what one 64-lane wave of k_trace_persistent does on an 8x8 tile -- a lane per pixel, one ray segment
 * per round, the nested walk loop with its leaf vote and early exit -- on the CPU, with the ORACLE's own arithmetic
 * for rays, hits and scattering (this file includes oracle/bevyray_oracle.c), and counts how often the wave
 * executes the interior body and the leaf body, for:
 *   width 2 / 4 / 8 nodes (the caller's binary tree collapsed, or a binned-SAH tree built here),
 *   reference order vs near-first order, with or without culling at pop time,
 *   leaves visited as a separate step, or their spheres tested inside the parent's step.
 * All variants return the same hit (closest t); only the work differs.  Used to decide which kernel to build.
 */
#include "../../oracle/bevyray_oracle.c"

#include <stdio.h>

#define MAXW 8
typedef struct {
    int n;                       /* children in use */
    float bmin[MAXW][3], bmax[MAXW][3];
    int child[MAXW];             /* >= 0: wide node id; < 0: leaf, sphere id = -child-1 */
} WNode;

typedef struct { WNode* nodes; int n_nodes; int root_is_leaf; int root_leaf; } WTree;

static float box_area(const float* lo, const float* hi) {
    float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
    return 2.0f * (dx * dy + dy * dz + dz * dx);
}

/* collapse the binary tree (BVHNode array, leaves of 1 sphere) rooted at `bn` into a wide node */
static int collapse(const BVHNode* b, uint32_t bn, int width, WTree* t) {
    int id = t->n_nodes++;
    uint32_t kids[MAXW]; int nk = 2;
    kids[0] = b[bn].index; kids[1] = b[bn].index + 1;
    for (;;) {
        if (nk >= width) break;
        int best = -1; float best_a = -1.0f;
        for (int i = 0; i < nk; i++) {
            if (b[kids[i]].model_count > 0) continue;
            float a = box_area(&b[kids[i]].minx, &b[kids[i]].maxx);
            if (a > best_a) { best_a = a; best = i; }
        }
        if (best < 0) break;
        uint32_t k = kids[best];
        /* keep the reference's child order: the pair replaces its parent in place */
        for (int i = nk; i > best + 1; i--) kids[i] = kids[i - 1];
        kids[best] = b[k].index; kids[best + 1] = b[k].index + 1;
        nk++;
    }
    WNode* w = &t->nodes[id];
    w->n = nk;
    for (int i = 0; i < nk; i++) {
        const BVHNode* c = &b[kids[i]];
        w->bmin[i][0] = c->minx; w->bmin[i][1] = c->miny; w->bmin[i][2] = c->minz;
        w->bmax[i][0] = c->maxx; w->bmax[i][1] = c->maxy; w->bmax[i][2] = c->maxz;
    }
    for (int i = 0; i < nk; i++) {
        const BVHNode* c = &b[kids[i]];
        int ch = c->model_count > 0 ? -(int)c->index - 1 : collapse(b, kids[i], width, t);
        t->nodes[id].child[i] = ch;
    }
    return id;
}

/* ---- multi-sphere leaves (raytrace.wgsl:348-362 loops `model_count` spheres of a leaf) ----------------------
 * g_leaf_max > 1: the SAH builder below may end a range of at most g_leaf_max spheres as ONE leaf -- always (g_leaf_rule 0), or when
 * the surface area heuristic says so (g_leaf_rule 1: n * C_sphere * area <= best split cost * C_sphere + C_step * area, the wave costs
 * of a sphere test and of an interior step).  A leaf is then a code >= LEAF_MULTI into the table {first, count} over g_leaf_ids.
 * g_leaf_units counts, per leaf-step execution of a wave, the LARGEST count among the lanes at a leaf: the step loops that often. */
#define LEAF_MULTI (1 << 24)
static int g_leaf_max = 1, g_leaf_rule = 0, g_pad_mode = 0;
static float g_c_sphere = 61.0f, g_c_step = 43.0f;
static int *g_leaf_first = NULL, *g_leaf_cnt = NULL, *g_leaf_ids = NULL, g_n_leaves = 0, g_n_leaf_ids = 0;
static uint64_t g_leaf_units = 0, g_leaf_hist[9];
void exp_set_leaf(int leaf_max, int rule, int pad_mode, float c_sphere, float c_step) {
    g_leaf_max = leaf_max; g_leaf_rule = rule; g_pad_mode = pad_mode; g_c_sphere = c_sphere; g_c_step = c_step;
}
void exp_get_leaf(uint64_t* out12) { out12[0] = g_leaf_units; out12[1] = (uint64_t)g_n_leaves; for (int i = 0; i < 9; i++) out12[2 + i] = g_leaf_hist[i]; }

/* ---- binned SAH BVH2 over the padded sphere boxes (Model::aabb, extract.rs:220-227) -> BVHNode array ---- */
typedef struct { float lo[3], hi[3]; uint32_t id; } Prim;
static void grow(float* lo, float* hi, const float* plo, const float* phi) {
    for (int k = 0; k < 3; k++) { if (plo[k] < lo[k]) lo[k] = plo[k]; if (phi[k] > hi[k]) hi[k] = phi[k]; }
}
static void sah_build(Prim* p, int n, BVHNode* out, uint32_t slot, uint32_t* n_out) {
    float lo[3] = {INF, INF, INF}, hi[3] = {-INF, -INF, -INF};
    for (int i = 0; i < n; i++) grow(lo, hi, p[i].lo, p[i].hi);
    BVHNode* nd = &out[slot];
    memset(nd, 0, sizeof *nd);
    nd->minx = lo[0]; nd->miny = lo[1]; nd->minz = lo[2]; nd->maxx = hi[0]; nd->maxy = hi[1]; nd->maxz = hi[2];
    if (n == 1) { nd->index = p[0].id; nd->model_count = 1; g_leaf_hist[1]++; return; }
    if (n <= g_leaf_max && g_leaf_rule == 0) {
multi_leaf:
        nd->index = (uint32_t)(LEAF_MULTI + g_n_leaves); nd->model_count = (uint32_t)n;
        g_leaf_first[g_n_leaves] = g_n_leaf_ids; g_leaf_cnt[g_n_leaves] = n; g_n_leaves++;
        for (int i = 0; i < n; i++) g_leaf_ids[g_n_leaf_ids++] = (int)p[i].id;
        g_leaf_hist[n < 8 ? n : 8]++;
        return;
    }
    /* full sweep SAH on each axis (n is small) */
    int best_axis = 0, best_split = n / 2; float best_cost = INF;
    float* right_area = (float*)malloc(sizeof(float) * (size_t)n);
    for (int axis = 0; axis < 3; axis++) {
        for (int i = 1; i < n; i++) {           /* insertion sort by centroid */
            Prim key = p[i]; float kc = key.lo[axis] + key.hi[axis]; int j = i - 1;
            while (j >= 0 && p[j].lo[axis] + p[j].hi[axis] > kc) { p[j + 1] = p[j]; j--; }
            p[j + 1] = key;
        }
        float rlo[3] = {INF, INF, INF}, rhi[3] = {-INF, -INF, -INF};
        for (int i = n - 1; i > 0; i--) { grow(rlo, rhi, p[i].lo, p[i].hi); right_area[i] = box_area(rlo, rhi); }
        float llo[3] = {INF, INF, INF}, lhi[3] = {-INF, -INF, -INF};
        for (int i = 1; i < n; i++) {
            grow(llo, lhi, p[i - 1].lo, p[i - 1].hi);
            float cost = box_area(llo, lhi) * (float)i + right_area[i] * (float)(n - i);
            if (cost < best_cost) { best_cost = cost; best_axis = axis; best_split = i; }
        }
    }
    free(right_area);
    if (n <= g_leaf_max && g_leaf_rule == 1) {
        const float area = box_area(lo, hi);
        if ((float)n * g_c_sphere * area <= best_cost * g_c_sphere + g_c_step * area) goto multi_leaf;
    }
    for (int i = 1; i < n; i++) {
        Prim key = p[i]; float kc = key.lo[best_axis] + key.hi[best_axis]; int j = i - 1;
        while (j >= 0 && p[j].lo[best_axis] + p[j].hi[best_axis] > kc) { p[j + 1] = p[j]; j--; }
        p[j + 1] = key;
    }
    uint32_t first = *n_out; *n_out += 2;
    nd->index = first; nd->model_count = 0;
    sah_build(p, best_split, out, first, n_out);
    sah_build(p + best_split, n - best_split, out, first + 1, n_out);
}

/* ---- per-lane walk over a wide tree ---- */
typedef struct { int id; float tnear; } SEnt;
typedef struct {
    int cur;            /* >= 0 wide node, <= -2: leaf (sphere -cur-2), -1: done */
    float cur_tnear;
    SEnt stack[256]; int sp;
    float closest; int closest_idx;
    Ray ray; vec3 inv;
} Walk;
enum { DONE = -1 };
static inline int is_int(int c) { return c >= 0; }
static inline int is_leaf(int c) { return c <= -2; }

typedef struct {
    int width, near_first, pop_cull, leaf_in_parent;
    int vote, exit_lanes;
} Variant;

typedef struct {
    uint64_t rounds, round_lanes, int_exec, int_lanes, leaf_exec, leaf_lanes, rays, pop_skips, sphere_tests, box_tests;
} SimCnt;

static float slab(const Walk* w, const float* lo, const float* hi) {
    return ray_bounding_dst(w->ray, V(lo[0], lo[1], lo[2]), V(hi[0], hi[1], hi[2]));
}
static void sphere(const Scene* s, Walk* w, int idx, SimCnt* c) {
    c->sphere_tests++;
    float t = hit_sphere(&s->models[idx], w->ray);
    if (t != -1.0f && t > 0.001f && t < w->closest) { w->closest = t; w->closest_idx = idx; }
}
static void pop(Walk* w, const Variant* v, SimCnt* c) {
    for (;;) {
        if (w->sp == 0) { w->cur = DONE; return; }
        SEnt e = w->stack[--w->sp];
        if (v->pop_cull && !(e.tnear < w->closest)) { c->pop_skips++; continue; }
        w->cur = e.id; w->cur_tnear = e.tnear; return;
    }
}
static void step_interior(const Scene* s, const WTree* t, Walk* w, const Variant* v, SimCnt* c) {
    const WNode* n = &t->nodes[w->cur];
    SEnt hit[MAXW]; int nh = 0;
    for (int i = 0; i < n->n; i++) {
        if (v->leaf_in_parent && n->child[i] < 0) { sphere(s, w, -n->child[i] - 1, c); continue; }
        c->box_tests++;
        float d = slab(w, n->bmin[i], n->bmax[i]);
        if (d != INF && d < w->closest) { hit[nh].id = n->child[i] < 0 ? n->child[i] - 1 : n->child[i]; hit[nh].tnear = d; nh++; }
    }
    if (v->near_first) {   /* farthest pushed first */
        for (int i = 1; i < nh; i++) { SEnt k = hit[i]; int j = i - 1; while (j >= 0 && hit[j].tnear < k.tnear) { hit[j + 1] = hit[j]; j--; } hit[j + 1] = k; }
    }
    /* leaf_in_parent may have shrunk closest after some boxes were accepted: cull again when pushing */
    for (int i = 0; i < nh; i++) if (!v->pop_cull || hit[i].tnear < w->closest) w->stack[w->sp++] = hit[i];
    pop(w, v, c);
}
static int leaf_count(int cur) { const int code = -cur - 2; return code >= LEAF_MULTI ? g_leaf_cnt[code - LEAF_MULTI] : 1; }
static void step_leaf(const Scene* s, Walk* w, const Variant* v, SimCnt* c) {
    const int code = -w->cur - 2;
    if (code >= LEAF_MULTI) {
        const int lf = code - LEAF_MULTI;
        for (int i = 0; i < g_leaf_cnt[lf]; i++) sphere(s, w, g_leaf_ids[g_leaf_first[lf] + i], c);     /* raytrace.wgsl:348-362 */
    } else sphere(s, w, code, c);
    pop(w, v, c);
}

/* ---- study of the top-of-tree LDS tile (SCENE_LDS_TOP): nodes whose breadth-first rank is < g_tile_k are "near"
 * (LDS), the others "far" (L2).  g_split = 0: the kernel's loop (one interior step for near and far lanes alike; a
 * step that contains a far lane waits for L2).  g_split = 1: near lanes and far lanes step separately; far lanes
 * wait (their loads in flight) until no near lane is left or g_far_vote of them have gathered. ---- */
static int g_tile_k = 0, g_split = 0, g_far_vote = 64;
static int* g_rank = NULL;
static uint64_t g_far[8];   /* 0 mixed steps containing a far lane, 1 far lanes in them, 2 near steps, 3 near lanes, 4 far steps, 5 far lanes */
void exp_set_far(int tile_k, int split, int far_vote) { g_tile_k = tile_k; g_split = split; g_far_vote = far_vote; }
void exp_get_far(uint64_t* out8) { memcpy(out8, g_far, sizeof g_far); }
static inline int is_far(int c) { return g_tile_k > 0 && c >= 0 && g_rank[c] >= g_tile_k; }

typedef struct {
    int active, in_flight, pend;
    uint32_t rng, sample, bounce;
    float uvx, uvy, first_depth;
    vec3 tput;
    Walk w;
} Lane;

/* ---- study of a leaf step in two parts.  Part A (cheap: the discriminant) for the lanes at a leaf: a lane whose ray misses the
 * sphere (discriminant < 0 or NaN) or has it behind its origin (h <= 0: then h - sqrt(..) <= 0 whatever the root, and t > 0.001
 * fails) pops and walks on; the others WAIT (pending) for part B (sqrt, divide, accept), which runs when g_vote_b of them have
 * gathered, when nobody else can move, or before the wave leaves the loop.  A pending lane does not walk (its `closest` is not
 * known yet: walking on with the old one would visit boxes the reference culls).  g_ab = 0: the kernel's single leaf step. ---- */
static int g_ab = 0, g_vote_a = 12, g_vote_b = 24;
static uint64_t g_abc[4];   /* A executions, A lanes, B executions, B lanes */
void exp_set_ab(int on, int vote_a, int vote_b) { g_ab = on; g_vote_a = vote_a; g_vote_b = vote_b; }
void exp_get_ab(uint64_t* out4) { memcpy(out4, g_abc, sizeof g_abc); }
static int part_a_passes(const Scene* s, const Walk* w, int idx) {
    const Model* m = &s->models[idx];
    vec3 oc = vsub(V(m->px, m->py, m->pz), w->ray.origin);
    float a = dot(w->ray.direction, w->ray.direction);
    float h = dot(w->ray.direction, oc);
    float c = dot(oc, oc) - m->radius * m->radius;
    float disc = h * h - a * c;
    return disc >= 0.0f && h > 0.0f;
}

/* one tile = one wave; runs until the wave has thinned to `stop_live` live lanes (the real kernel hands the rest over) */
static void sim_tile(const Scene* s, const WTree* t, const Variant* v, uint32_t tx, uint32_t ty, uint32_t W, uint32_t H,
                     int stop_live, SimCnt* c) {
    static __thread Lane L[64];
    int live = 0;
    for (int l = 0; l < 64; l++) {
        uint32_t px = tx * 8 + (l & 7), py = ty * 8 + (l >> 3);
        Lane* a = &L[l];
        memset(a, 0, sizeof *a);
        if (px >= W || py >= H) continue;
        a->uvx = ((float)px + 0.5f) / (float)W; a->uvy = ((float)py + 0.5f) / (float)H;
        a->rng = f32_to_u32((s->window.random_seed * 10000.0f) * (a->uvx * 402.0f) * (a->uvy * 31.5f));
        a->active = 1; live++;
    }
    while (live > stop_live) {
        c->rounds++; c->round_lanes += (uint64_t)live;
        int n_walking = 0;
        for (int l = 0; l < 64; l++) {
            Lane* a = &L[l];
            if (!a->active) continue;
            if (!a->in_flight) {
                if (a->bounce == 0) {
                    a->w.ray = random_ray_from_uv(s, a->uvx, a->uvy, &a->rng);
                    a->tput = V(1, 1, 1); a->first_depth = INF;
                }
                Walk* w = &a->w;
                w->inv = V(1.0f / w->ray.direction.x, 1.0f / w->ray.direction.y, 1.0f / w->ray.direction.z);
                w->closest = INF; w->closest_idx = -1; w->sp = 0;
                w->cur = t->root_is_leaf ? -t->root_leaf - 2 : 0; w->cur_tnear = 0.0f;
            }
            n_walking++;
        }
        int exit_at = n_walking >> 1; if (exit_at > v->exit_lanes) exit_at = v->exit_lanes;
        if (g_ab) {
            for (;;) {
                int ni = 0, nl = 0, np = 0;
                for (;;) {
                    ni = 0;
                    for (int l = 0; l < 64; l++) if (L[l].active && !L[l].pend && is_int(L[l].w.cur)) ni++;
                    if (!ni) break;
                    c->int_exec++; c->int_lanes += (uint64_t)ni;
                    for (int l = 0; l < 64; l++) if (L[l].active && !L[l].pend && is_int(L[l].w.cur)) step_interior(s, t, &L[l].w, v, c);
                    nl = 0; np = 0;
                    for (int l = 0; l < 64; l++) if (L[l].active) { if (L[l].pend) np++; else if (is_leaf(L[l].w.cur)) nl++; }
                    if (nl >= g_vote_a || np >= g_vote_b) break;
                }
                ni = nl = np = 0;
                for (int l = 0; l < 64; l++) if (L[l].active) { if (L[l].pend) np++; else if (is_leaf(L[l].w.cur)) nl++; else if (is_int(L[l].w.cur)) ni++; }
                if (nl && (nl >= g_vote_a || !ni)) {
                    g_abc[0]++; g_abc[1] += (uint64_t)nl;
                    for (int l = 0; l < 64; l++) if (L[l].active && !L[l].pend && is_leaf(L[l].w.cur)) {
                        if (part_a_passes(s, &L[l].w, -L[l].w.cur - 2)) { L[l].pend = 1; np++; }
                        else { c->sphere_tests++; pop(&L[l].w, v, c); }
                    }
                }
                int nw = 0, movers = 0;
                for (int l = 0; l < 64; l++) if (L[l].active && L[l].w.cur != DONE) { nw++; if (!L[l].pend) movers++; }
                if (np && (np >= g_vote_b || !movers || nw <= exit_at)) {
                    g_abc[2]++; g_abc[3] += (uint64_t)np;
                    c->leaf_exec++; c->leaf_lanes += (uint64_t)np;
                    for (int l = 0; l < 64; l++) if (L[l].active && L[l].pend) { L[l].pend = 0; step_leaf(s, &L[l].w, v, c); }
                    nw = 0;
                    for (int l = 0; l < 64; l++) if (L[l].active && L[l].w.cur != DONE) nw++;
                }
                if (nw <= exit_at) break;
            }
        } else
        for (;;) {
            for (;;) {
                int ni = 0, nf = 0;
                for (int l = 0; l < 64; l++) if (L[l].active && is_int(L[l].w.cur)) { ni++; nf += is_far(L[l].w.cur); }
                if (!ni) break;
                c->int_exec++;
                if (g_split && g_tile_k > 0) {
                    const int far_step = nf > 0 && (nf == ni || nf >= g_far_vote);
                    const int n = far_step ? nf : ni - nf;
                    c->int_lanes += (uint64_t)n;
                    g_far[far_step ? 4 : 2]++; g_far[far_step ? 5 : 3] += (uint64_t)n;
                    for (int l = 0; l < 64; l++)
                        if (L[l].active && is_int(L[l].w.cur) && is_far(L[l].w.cur) == far_step) step_interior(s, t, &L[l].w, v, c);
                } else {
                    c->int_lanes += (uint64_t)ni;
                    if (nf) { g_far[0]++; g_far[1] += (uint64_t)nf; }
                    for (int l = 0; l < 64; l++) if (L[l].active && is_int(L[l].w.cur)) step_interior(s, t, &L[l].w, v, c);
                }
                int nl = 0;
                for (int l = 0; l < 64; l++) if (L[l].active && is_leaf(L[l].w.cur)) nl++;
                if (nl >= v->vote) break;
            }
            int nl = 0;
            for (int l = 0; l < 64; l++) if (L[l].active && is_leaf(L[l].w.cur)) nl++;
            if (nl) {
                c->leaf_exec++; c->leaf_lanes += (uint64_t)nl;
                int mx = 1;
                for (int l = 0; l < 64; l++) if (L[l].active && is_leaf(L[l].w.cur)) { const int k = leaf_count(L[l].w.cur); if (k > mx) mx = k; }
                g_leaf_units += (uint64_t)mx;
                for (int l = 0; l < 64; l++) if (L[l].active && is_leaf(L[l].w.cur)) step_leaf(s, &L[l].w, v, c);
            }
            int nw = 0;
            for (int l = 0; l < 64; l++) if (L[l].active && L[l].w.cur != DONE) nw++;
            if (nw <= exit_at) break;
        }
        for (int l = 0; l < 64; l++) {
            Lane* a = &L[l];
            if (!a->active) continue;
            a->in_flight = a->w.cur != DONE;
            if (a->in_flight) continue;
            c->rays++;
            Walk* w = &a->w;
            if (a->bounce == 0) a->first_depth = w->closest;
            int ended = 0;
            if (w->closest == INF) ended = 1;
            else {
                HitInfo h;
                const Model* m = &s->models[w->closest_idx];
                h.distance = w->closest;
                h.position = ray_at(w->ray, w->closest);
                h.normal = normalize(vsub(h.position, V(m->px, m->py, m->pz)));
                h.material = m->material_id;
                h.front_face = dot(w->ray.direction, h.normal) < 0.0f;
                vec3 att;
                int absorbed = scatter(s, &w->ray, &att, &h, &a->rng);
                if (absorbed) ended = 1;
                else { a->bounce++; if (a->bounce > s->camera.bounce_count) ended = 1; }
            }
            if (ended) {
                a->bounce = 0; a->sample++;
                if (a->sample == s->camera.sample_count) { a->active = 0; live--; }
            }
        }
    }
}


/* ---- two pixels per lane: a lane walks the ray of its slot 0, then (in the same loop) the ray of slot 1 ---- */
typedef struct {
    int active, in_flight, has_ray;
    uint32_t rng, sample, bounce;
    float uvx, uvy, first_depth;
    vec3 tput;
    Walk w;
} Slot;

static void slot_shade(const Scene* s, Slot* a, SimCnt* c, int* live) {
    Walk* w = &a->w;
    c->rays++;
    if (a->bounce == 0) a->first_depth = w->closest;
    int ended = 0;
    if (w->closest == INF) ended = 1;
    else {
        HitInfo h;
        const Model* m = &s->models[w->closest_idx];
        h.distance = w->closest;
        h.position = ray_at(w->ray, w->closest);
        h.normal = normalize(vsub(h.position, V(m->px, m->py, m->pz)));
        h.material = m->material_id;
        h.front_face = dot(w->ray.direction, h.normal) < 0.0f;
        vec3 att;
        int absorbed = scatter(s, &w->ray, &att, &h, &a->rng);
        if (absorbed) ended = 1;
        else { a->bounce++; if (a->bounce > s->camera.bounce_count) ended = 1; }
    }
    if (ended) {
        a->bounce = 0; a->sample++;
        if (a->sample == s->camera.sample_count) { a->active = 0; (*live)--; }
    }
}

static void sim_tile_p2(const Scene* s, const WTree* t, const Variant* v, uint32_t tx, uint32_t ty, uint32_t W, uint32_t H,
                        int stop_live, SimCnt* c) {
    static __thread Slot S[64][2];
    int live = 0;
    for (int l = 0; l < 64; l++) for (int k = 0; k < 2; k++) {
        uint32_t px = (tx + (uint32_t)k) * 8 + (l & 7), py = ty * 8 + (l >> 3);
        Slot* a = &S[l][k];
        memset(a, 0, sizeof *a);
        if (px >= W || py >= H) continue;
        a->uvx = ((float)px + 0.5f) / (float)W; a->uvy = ((float)py + 0.5f) / (float)H;
        a->rng = f32_to_u32((s->window.random_seed * 10000.0f) * (a->uvx * 402.0f) * (a->uvy * 31.5f));
        a->active = 1; live++;
    }
    while (live > 2 * stop_live) {
        c->rounds++; c->round_lanes += (uint64_t)live;
        int cur_slot[64];
        for (int l = 0; l < 64; l++) {
            for (int k = 0; k < 2; k++) {
                Slot* a = &S[l][k];
                if (!a->active) { a->has_ray = 0; continue; }
                if (!a->in_flight) {
                    if (a->bounce == 0) { a->w.ray = random_ray_from_uv(s, a->uvx, a->uvy, &a->rng); a->tput = V(1, 1, 1); a->first_depth = INF; }
                    Walk* w = &a->w;
                    w->closest = INF; w->closest_idx = -1; w->sp = 0;
                    w->cur = t->root_is_leaf ? -t->root_leaf - 2 : 0;
                }
                a->has_ray = 1;
            }
            /* continue the suspended one first */
            cur_slot[l] = (S[l][1].has_ray && S[l][1].in_flight && !(S[l][0].has_ray && S[l][0].in_flight)) ? 1 : 0;
            if (!S[l][cur_slot[l]].has_ray) cur_slot[l] ^= 1;
        }
        #define CUR(l) (S[l][cur_slot[l]])
        #define WALKING(l) (CUR(l).has_ray && CUR(l).w.cur != DONE)
        int n_walking = 0;
        for (int l = 0; l < 64; l++) if (S[l][0].has_ray || S[l][1].has_ray) n_walking++;
        int exit_at = n_walking >> 1; if (exit_at > v->exit_lanes) exit_at = v->exit_lanes;
        for (;;) {
            for (;;) {
                /* a lane whose current walk has ended moves on to its other ray (ideal: no cost) */
                for (int l = 0; l < 64; l++) if (CUR(l).has_ray && CUR(l).w.cur == DONE) {
                    Slot* o = &S[l][cur_slot[l] ^ 1];
                    if (o->has_ray && o->w.cur != DONE) cur_slot[l] ^= 1;
                }
                int ni = 0;
                for (int l = 0; l < 64; l++) if (WALKING(l) && is_int(CUR(l).w.cur)) ni++;
                if (!ni) break;
                c->int_exec++; c->int_lanes += (uint64_t)ni;
                for (int l = 0; l < 64; l++) if (WALKING(l) && is_int(CUR(l).w.cur)) step_interior(s, t, &CUR(l).w, v, c);
                int nl = 0;
                for (int l = 0; l < 64; l++) if (WALKING(l) && is_leaf(CUR(l).w.cur)) nl++;
                if (nl >= v->vote) break;
            }
            int nl = 0;
            for (int l = 0; l < 64; l++) if (WALKING(l) && is_leaf(CUR(l).w.cur)) nl++;
            if (nl) {
                c->leaf_exec++; c->leaf_lanes += (uint64_t)nl;
                for (int l = 0; l < 64; l++) if (WALKING(l) && is_leaf(CUR(l).w.cur)) step_leaf(s, &CUR(l).w, v, c);
            }
            int nw = 0;
            for (int l = 0; l < 64; l++) {
                int busy = 0;
                for (int k = 0; k < 2; k++) if (S[l][k].has_ray && S[l][k].w.cur != DONE) busy = 1;
                nw += busy;
            }
            if (nw <= exit_at) break;
        }
        for (int l = 0; l < 64; l++) for (int k = 0; k < 2; k++) {
            Slot* a = &S[l][k];
            if (!a->has_ray) continue;
            a->in_flight = a->w.cur != DONE;
            if (!a->in_flight) slot_shade(s, a, c, &live);
        }
    }
}

/* Python entry: runs `n_tiles` tiles (tile coordinates in tiles_xy) through one variant; out10 = SimCnt */
int exp_run(const void* models, uint32_t n_models, const void* materials, uint32_t n_materials, const void* bvh_nodes,
            uint32_t n_nodes, const void* camera80, const void* window16, uint32_t W, uint32_t H, const uint32_t* tiles_xy,
            uint32_t n_tiles, int width, int near_first, int pop_cull, int leaf_in_parent, int vote, int exit_lanes,
            int use_sah, int stop_live, uint64_t* out10) {
    Scene s;
    s.models = (const Model*)models; s.n_models = n_models;
    s.materials = (const Material*)materials; s.n_materials = n_materials;
    s.bvh = (const BVHNode*)bvh_nodes; s.n_nodes = n_nodes;
    memcpy(&s.camera, camera80, 80); memcpy(&s.window, window16, 16);
    s.level = 3; s.tan_half_fov = oracle_tan_half_fov(s.camera.fov);
    BVHNode* own = NULL;
    const BVHNode* b = s.bvh;
    if (use_sah & 1) {
        Prim* p = (Prim*)malloc(sizeof(Prim) * n_models);
        for (uint32_t i = 0; i < n_models; i++) {
            const Model* m = &s.models[i];
            /* g_pad_mode 1: the callee's leaf pads (brt_sah.h sah_model_pad at the scenes' scale: the 0.01 floor) instead of the reference's 0.1 */
            float r = m->radius + ((g_pad_mode == 1 && m->radius <= 100.0f) ? 0.01f : 0.1f);
            p[i].lo[0] = m->px - r; p[i].lo[1] = m->py - r; p[i].lo[2] = m->pz - r;
            p[i].hi[0] = m->px + r; p[i].hi[1] = m->py + r; p[i].hi[2] = m->pz + r; p[i].id = i;
        }
        own = (BVHNode*)calloc(2 * (size_t)n_models, sizeof(BVHNode));
        free(g_leaf_first); free(g_leaf_cnt); free(g_leaf_ids);
        g_leaf_first = (int*)calloc(n_models + 1, sizeof(int)); g_leaf_cnt = (int*)calloc(n_models + 1, sizeof(int)); g_leaf_ids = (int*)calloc(n_models + 1, sizeof(int));
        g_n_leaves = 0; g_n_leaf_ids = 0; memset(g_leaf_hist, 0, sizeof g_leaf_hist);
        uint32_t n_out = 1;
        sah_build(p, (int)n_models, own, 0, &n_out);
        free(p);
        b = own;
    }
    WTree t; t.nodes = (WNode*)calloc(n_models + 1, sizeof(WNode)); t.n_nodes = 0;
    t.root_is_leaf = b[0].model_count > 0; t.root_leaf = (int)b[0].index;
    if (!t.root_is_leaf) collapse(b, 0, width, &t);
    /* breadth-first rank of every wide node (the device layout puts the top levels first) */
    free(g_rank); g_rank = (int*)calloc((size_t)t.n_nodes + 1, sizeof(int));
    if (!t.root_is_leaf) {
        int* q = (int*)malloc(sizeof(int) * (size_t)t.n_nodes); int qh = 0, qt = 0; q[qt++] = 0;
        while (qh < qt) { int id = q[qh]; g_rank[id] = qh++; for (int i = 0; i < t.nodes[id].n; i++) if (t.nodes[id].child[i] >= 0) q[qt++] = t.nodes[id].child[i]; }
        free(q);
    }
    memset(g_far, 0, sizeof g_far);
    memset(g_abc, 0, sizeof g_abc);
    g_leaf_units = 0;
    Variant v = {width, near_first, pop_cull, leaf_in_parent, vote, exit_lanes};
    SimCnt c; memset(&c, 0, sizeof c);
    for (uint32_t i = 0; i < n_tiles; i++) {
        if (use_sah & 2) sim_tile_p2(&s, &t, &v, tiles_xy[2 * i] & ~1u, tiles_xy[2 * i + 1], W, H, stop_live, &c);
        else sim_tile(&s, &t, &v, tiles_xy[2 * i], tiles_xy[2 * i + 1], W, H, stop_live, &c);
    }
    memcpy(out10, &c, sizeof c);
    out10[10] = (uint64_t)t.n_nodes;
    free(t.nodes); free(own);
    return 0;
}
