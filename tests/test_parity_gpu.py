"""Parity of the HIP path (through the C ABI) against the CPU oracle -- needs an MI355X.

Bar: BIT-EXACT frames (compared as u32) and exactly equal ray counters.  The arithmetic is
f32 with one numeric policy on both sides (no FMA contraction, correctly rounded sqrt/divide,
minNum/maxNum), so nothing is left to a tolerance; north_star's "stated per-channel float
tolerance" is therefore 0 here (TOL below), and any mismatch is a bug.
"""
import json
import os

import numpy as np
import pytest

import bevyray_amd as brt
from helpers import (GOLDEN, chain_bvh, fixture_buffers, make_buffers, median_split_bvh, single_leaf_bvh, sky_color,
                     tiny_frame_cases, uniforms)

pytestmark = pytest.mark.gpu

TOL = 0.0   # per-channel absolute tolerance on the RGBA f32 output
COUNTER_KEYS = ("rays", "node_pops", "interior_visits", "sphere_tests", "hits")
DBG_MINMAX, DBG_SQRT_DIV, DBG_RNG, DBG_SLAB, DBG_SPHERE, DBG_SEED = range(6)


def assert_frames_equal(got, want):
    """Bit-exact (NaN == NaN regardless of payload); TOL documents the bar."""
    got, want = np.asarray(got, np.float32), np.asarray(want, np.float32)
    assert got.shape == want.shape
    same = got.view(np.uint32) == want.view(np.uint32)
    both_nan = np.isnan(got) & np.isnan(want)
    bad = ~(same | both_nan)
    if bad.any():
        with np.errstate(all="ignore"):
            diff = np.nanmax(np.abs(got[bad].astype(np.float64) - want[bad].astype(np.float64)))
        if not (diff <= TOL):
            idx = np.argwhere(bad)[:5]
            raise AssertionError(f"{bad.sum()} of {bad.size} values differ (max abs {diff:.3g}); first at {idx.tolist()}")


def render_both(plugin, oracle, b, lvl, cam, win, w, h, flags=brt.FLAG_COUNTERS, raster=None, depth=None):
    got = plugin.node.run(lvl, cam, win, w, h, buffers=b, raster_rgba=raster, raster_depth=depth, flags=flags)
    want, cnt = oracle.render(b, lvl, cam, win, w, h, raster_rgba=raster, raster_depth=depth)
    assert_frames_equal(got, want)
    stats = plugin.node.last_stats
    assert stats["rays"] == cnt["rays"]
    if flags & brt.FLAG_COUNTERS:
        assert {k: stats[k] for k in COUNTER_KEYS} == cnt
    return got, stats


# ---- single functions -------------------------------------------

def _inputs(cols):
    a = np.zeros((len(cols[0]), 16), np.float32)
    for i, c in enumerate(cols):
        a[:, i] = c
    return a


def test_min_max_sqrt_divide_bitwise(plugin, oracle):
    rng = np.random.default_rng(1)
    n = 200000
    a = rng.standard_normal(n).astype(np.float32) * np.float32(10.0) ** rng.integers(-20, 20, n).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32) * np.float32(10.0) ** rng.integers(-20, 20, n).astype(np.float32)
    special = np.array([0.0, -0.0, np.nan, np.inf, -np.inf, 1.0, -1.0, 3.4028235e38, 1e-45, -1e-45, 1.1754944e-38], np.float32)
    sa, sb = np.meshgrid(special, special)
    a = np.concatenate([a, sa.ravel()]); b = np.concatenate([b, sb.ravel()])
    out = plugin.debug_eval(DBG_MINMAX, _inputs([a, b]))
    want_min = np.array([oracle.lib.oracle_min(float(x), float(y)) for x, y in zip(a[-121:], b[-121:])], np.float32)
    want_max = np.array([oracle.lib.oracle_max(float(x), float(y)) for x, y in zip(a[-121:], b[-121:])], np.float32)
    assert_frames_equal(out[-121:, 0], want_min)
    assert_frames_equal(out[-121:, 1], want_max)
    assert_frames_equal(out[:n, 0], np.minimum(a[:n], b[:n]))
    assert_frames_equal(out[:n, 1], np.maximum(a[:n], b[:n]))
    # sqrt / divide: correctly rounded, denormals kept (numpy on x86 SSE is IEEE-exact)
    out = plugin.debug_eval(DBG_SQRT_DIV, _inputs([np.abs(a), b]))
    with np.errstate(all="ignore"):
        assert_frames_equal(out[:, 0], np.sqrt(np.abs(a)))
        assert_frames_equal(out[:, 1], np.abs(a) / b)


def test_rng_seed_and_unit_ball_bitwise(plugin, oracle):
    kat = json.load(open(os.path.join(GOLDEN, "rng_kat.json")))
    starts = np.array([c["start"] for c in kat["chains"]], np.uint32)
    out = plugin.debug_eval(DBG_RNG, _inputs([starts.view(np.float32)]))
    for i, c in enumerate(kat["chains"]):
        assert out[i, 0] == np.float32(c["floats"][0]) and out[i, 1].view(np.uint32) == c["states"][0]
        ball, end = oracle.unit_ball(c["states"][0])
        assert np.array_equal(out[i, 2:5], ball) and out[i, 5].view(np.uint32) == end
    s = kat["seeds"]
    inp = _inputs([np.array([x[k] for x in s], np.float32) for k in ("random_seed", "px", "py", "w", "h")])
    out = plugin.debug_eval(DBG_SEED, inp)
    assert [int(x) for x in out[:, 0].view(np.uint32)] == [x["seed"] for x in s]
    # the rejection sampler over many states
    rng = np.random.default_rng(3)
    states = rng.integers(0, 2**32, 5000, dtype=np.uint64).astype(np.uint32)
    out = plugin.debug_eval(DBG_RNG, _inputs([states.view(np.float32)]))
    for i in range(0, 5000, 37):
        f, s1 = oracle.rng_floats(int(states[i]), 1)
        ball, s2 = oracle.unit_ball(s1)
        assert out[i, 0] == f[0] and np.array_equal(out[i, 2:5], ball) and out[i, 5].view(np.uint32) == s2


def test_slab_and_sphere_tests_bitwise(plugin, oracle):
    import ctypes as C
    rng = np.random.default_rng(4)
    n = 4000
    o = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    d[::7, 0] = 0.0; d[::11, 1] = -0.0; d[::13] *= np.float32(1e-3)
    lo = rng.uniform(-4, 2, (n, 3)).astype(np.float32)
    hi = lo + rng.uniform(0, 3, (n, 3)).astype(np.float32)
    o[::17, 0] = lo[::17, 0]; o[::19, 1] = hi[::19, 1]          # origin on a slab plane, incl. with d == 0
    closest = np.where(rng.random(n) < 0.3, np.float32(3.40282347e+38), rng.uniform(0, 5, n)).astype(np.float32)
    inp = np.zeros((n, 16), np.float32)
    inp[:, 0:3], inp[:, 3:6], inp[:, 6:9], inp[:, 9:12], inp[:, 12] = o, d, lo, hi, closest
    out = plugin.debug_eval(DBG_SLAB, inp)
    f3 = lambda v: (C.c_float * 3)(*[float(x) for x in v])
    INF = np.float32(3.40282347e+38)
    for i in range(n):
        dst = np.float32(oracle.lib.oracle_ray_bounding_dst(f3(o[i]), f3(d[i]), f3(lo[i]), f3(hi[i])))
        assert out[i, 0] == (1.0 if (dst != INF and dst < closest[i]) else 0.0), i
    c = rng.uniform(-3, 3, (n, 3)).astype(np.float32)
    r = rng.uniform(0.05, 2.0, n).astype(np.float32)
    inp = np.zeros((n, 16), np.float32)
    inp[:, 0:3], inp[:, 3:6], inp[:, 6:9], inp[:, 9] = o, d, c, r
    out = plugin.debug_eval(DBG_SPHERE, inp)
    for i in range(n):
        t = np.float32(oracle.lib.oracle_hit_sphere(f3(o[i]), f3(d[i]), f3(c[i]), float(r[i])))
        want = t if (t != np.float32(-1.0) and t > np.float32(0.001)) else INF
        assert out[i, 0].view(np.uint32) == np.float32(want).view(np.uint32), i


# ---- frames --------------------------------------------------------------------------------------------

def test_golden_cover_fixture(plugin, oracle):
    b, lvl, cam, win, frame, counters = fixture_buffers()
    got = plugin.node.run(lvl, cam, win, 64, 36, buffers=b, flags=brt.FLAG_COUNTERS)
    assert_frames_equal(got, frame)
    assert [plugin.node.last_stats[k] for k in COUNTER_KEYS] == counters


def test_numpy_restatement_frames(plugin):
    # frames rendered by the second, independent restatement of the shader (numpy f32 scalars)
    for name, b, lvl, cam, win, w, h, raster, depth, frame, rays in tiny_frame_cases():
        got = plugin.node.run(lvl, cam, win, w, h, buffers=b, raster_rgba=raster, raster_depth=depth)
        assert_frames_equal(got, frame)
        assert plugin.node.last_stats["rays"] == rays, name


@pytest.mark.parametrize("seed", [0.0, 0.25, 0.5, 0.999])
def test_config1_cover_400x225_1spp_4bounces(plugin, oracle, seed):
    # BASELINE.json configs[0]: the reference's own CPU-runnable case
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    lvl, cam, win = brt.cover_camera(400, 225, 1, 4, brt.Raytracing.Pure, seed)
    _, stats = render_both(plugin, oracle, b, lvl, cam, win, 400, 225)
    assert stats["scene_in_lds"] == 1 and stats["paths"] == 400 * 225


@pytest.mark.parametrize("kind,w,h,spp,bounces", [
    (brt.SCENE_COVER, 192, 108, 8, 8),          # configs[1] shape, reduced size
    (brt.SCENE_RTIOW_FINAL, 160, 90, 4, 50),    # configs[2] shape: 50 bounces
    (brt.SCENE_STRESS_GRID, 160, 90, 4, 8),     # configs[4]: 10k spheres, scene NOT LDS resident
    (brt.SCENE_COVER, 61, 35, 3, 2),            # ragged: neither dimension a multiple of 8
    (brt.SCENE_COVER, 1, 1, 5, 3),
])
def test_scenes_match_oracle(plugin, oracle, kind, w, h, spp, bounces):
    b = brt.generate_scene(kind, 1)
    lvl, cam, win = brt.cover_camera(w, h, spp, bounces)
    _, stats = render_both(plugin, oracle, b, lvl, cam, win, w, h)
    assert stats["scene_in_lds"] == (2 if kind == brt.SCENE_STRESS_GRID else 1)   # 2: top of the tree in LDS, the rest from L2
    # timing build (no counters) gives the same pixels and ray count
    render_both(plugin, oracle, b, lvl, cam, win, w, h, flags=0)


def test_bringup_kernel_matches_oracle(plugin, oracle):
    b = brt.generate_scene(brt.SCENE_COVER, 3)
    lvl, cam, win = brt.cover_camera(96, 54, 4, 6, brt.Raytracing.Pure, 0.25)
    render_both(plugin, oracle, b, lvl, cam, win, 96, 54, flags=brt.FLAG_COUNTERS | brt.FLAG_KERNEL_SIMPLE)


def test_hand_made_bvh_topologies(plugin, oracle):
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    lvl, cam, win = brt.cover_camera(64, 36, 2, 4)
    ref, _ = render_both(plugin, oracle, b, lvl, cam, win, 64, 36)
    for bvh in (single_leaf_bvh(b.models), median_split_bvh(b.models, 3), median_split_bvh(b.models, 1)):
        bb = brt.Buffers(b.models, b.materials, bvh)
        got, _ = render_both(plugin, oracle, bb, lvl, cam, win, 64, 36)
        assert_frames_equal(got, ref)          # pixels do not depend on the topology
    # callee-built BVH (bvh = None): binned SAH inside brt_upload_scene, or PLOC on the GPU with the quality knob off --
    # same pixels, and exactly the oracle's counters on the tree the library says it builds
    for quality, tree in ((1, brt.build_bvh_sah(b.models)), (0, brt.build_bvh(b.models))):
        with plugin.tuning(BRT_BVH_QUALITY=quality):
            got = plugin.node.run(lvl, cam, win, 64, 36, buffers=brt.Buffers(b.models, b.materials, None), flags=brt.FLAG_COUNTERS)
            stats = dict(plugin.node.last_stats)
        assert_frames_equal(got, ref)
        _, cnt = oracle.render(brt.Buffers(b.models, b.materials, tree), lvl, cam, win, 64, 36)
        assert {k: stats[k] for k in COUNTER_KEYS} == cnt, quality


def test_stack_overflow_rule(plugin, oracle):
    # raytrace.wgsl:320: a 40-deep caterpillar overflows the 32-entry stack and drops subtrees
    data = [((0.0, 0.0, -5.0 - i), 0.5, brt.StandardMaterial(base_color=(0.8, 0.3, 0.3))) for i in range(40)]
    for n in (40, 33, 32, 31, 30, 28):
        deep = make_buffers(data[:n], chain_bvh)
        lvl, cam, win = uniforms(32, 32, spp=2, bounces=3, pos=(0, 0, 0), target=(0, 0, -1), fov=0.3, seed=0.5)
        got, _ = render_both(plugin, oracle, deep, lvl, cam, win, 32, 32)
        flat = make_buffers(data[:n], single_leaf_bvh)
        want_flat, _ = oracle.render(flat, lvl, cam, win, 32, 32)
        if n <= 31:
            assert_frames_equal(got, want_flat)
        else:
            assert not np.array_equal(got, want_flat)   # the overflow is visible


def test_near_far_read_offsets_and_their_repair(plugin, oracle):
    """The slab test reads {near, far} planes by the sign of the ray direction (brt_layout.h); rays with
    a zero direction component (1/d infinite, 0 * inf = NaN on a slab plane) and boxes that are not
    finite or not ordered must take the min/max repair loop and still match raytrace.wgsl:387-398."""
    spheres = [((0.0, 0.0, -5.0), 0.5, brt.StandardMaterial(base_color=(0.8, 0.3, 0.3), perceptual_roughness=0.0)),
               ((0.25, 0.25, -8.0), 1.0, brt.StandardMaterial(metallic=1.0, perceptual_roughness=0.3)),
               ((-0.5, 0.1, -3.0), 0.25, brt.StandardMaterial(specular_transmission=1.0, ior=1.5)),
               ((0.0, -100.5, -5.0), 100.0, brt.StandardMaterial(base_color=(0.5, 0.5, 0.5)))]
    b = make_buffers(spheres, lambda m: median_split_bvh(m, 1))
    # every primary ray parallel to -Z: `up` parallel to the view direction makes `right` the zero vector and
    # the image-plane offsets multiples of (0, 0, -1); the camera x sits exactly on a padded slab plane
    lvl, cam, win = uniforms(24, 24, spp=3, bounces=4, pos=(0.0, 0.0, 0.0), target=(0.0, 0.0, -1.0), fov=0.6, seed=0.5)
    cam = cam.copy()
    cam["up"] = (0.0, 0.0, -1.0)
    plane = np.float32(0.0) - (np.float32(0.5) + np.float32(0.1))      # Model::aabb min.x of the first sphere
    for x in (0.0, float(plane), -0.5):
        cam["position"] = (x, 0.0, 0.0)
        render_both(plugin, oracle, b, lvl, cam, win, 24, 24)
        # without counters the wave with an unsafe ray runs the repairing loop and THEN the hand-written one (walk_run): also at
        # the extremes of the two thresholds both loops take
        render_both(plugin, oracle, b, lvl, cam, win, 24, 24, flags=0)
        for vote, exit_at in ((0, 0), (64, 63), (63, 1), (1, 62)):
            with plugin.tuning(BRT_LEAF_VOTE=vote, BRT_WALK_EXIT=exit_at):
                render_both(plugin, oracle, b, lvl, cam, win, 24, 24, flags=0)
    # boxes the reference never validates: swapped bounds, infinite bounds, a NaN bound
    lvl, cam, win = uniforms(40, 24, spp=3, bounces=5, pos=(0.3, 0.4, 1.0), target=(0.0, 0.0, -5.0), fov=0.7, seed=0.25)
    for victim, edit in ((1, "swap"), (2, "inf"), (3, "nan"), (4, "swap")):
        bvh = median_split_bvh(b.models, 1).copy()
        victim = min(victim, len(bvh) - 1)
        if edit == "swap":
            lo, hi = bvh[victim]["bounds_min"].copy(), bvh[victim]["bounds_max"].copy()
            bvh[victim]["bounds_min"][0], bvh[victim]["bounds_max"][0] = hi[0], lo[0]
        elif edit == "inf":
            bvh[victim]["bounds_min"] = (-np.inf, -np.inf, -np.inf)
            bvh[victim]["bounds_max"] = (np.inf, np.inf, np.inf)
        else:
            bvh[victim]["bounds_min"][1] = np.nan
        render_both(plugin, oracle, brt.Buffers(b.models, b.materials, bvh), lvl, cam, win, 40, 24)


def test_analytic_cases(plugin, oracle):
    b = make_buffers([((0, 0, 50), 0.5, brt.StandardMaterial())], single_leaf_bvh)
    lvl, cam, win = uniforms(16, 9, spp=2, bounces=3, pos=(0, 0, 0), target=(0, 0, -1), fov=0.8, seed=0.0)
    got = plugin.node.run(lvl, cam, win, 16, 9, buffers=b)
    assert np.array_equal(got[..., :3], sky_color(oracle, cam, win, 16, 9, 2)) and np.all(got[..., 3] == 1.0)
    # zero samples -> NaN colour, alpha 1 (raytrace.wgsl:169)
    lvl, cam, win = uniforms(8, 8, spp=0, bounces=2, pos=(0, 0, 0), target=(0, 0, -1), fov=0.5, seed=0.5)
    got, stats = render_both(plugin, oracle, b, lvl, cam, win, 8, 8)
    assert np.all(np.isnan(got[..., :3])) and stats["rays"] == 0
    # zero bounces on a diffuse hit -> black
    wall = make_buffers([((0, 0, -3), 100.0, brt.StandardMaterial(perceptual_roughness=0.0))], single_leaf_bvh)
    lvl, cam, win = uniforms(8, 8, spp=4, bounces=0, pos=(0, 0, 200), target=(0, 0, -3), fov=0.2, seed=0.5)
    got, _ = render_both(plugin, oracle, wall, lvl, cam, win, 8, 8)
    assert np.all(got[..., :3] == 0.0)
    # materials: all-metal, all-glass (ior below and above 1), rough metal
    for mat in (brt.StandardMaterial(metallic=1.0, perceptual_roughness=0.0), brt.StandardMaterial(metallic=1.0, perceptual_roughness=0.9),
                brt.StandardMaterial(specular_transmission=1.0, ior=1.5), brt.StandardMaterial(specular_transmission=1.0, ior=0.6),
                brt.StandardMaterial(metallic=0.5, specular_transmission=0.5, base_color=(0.9, 0.5, 0.2))):
        data = [((0, -100.5, -1), 100.0, brt.StandardMaterial(base_color=(0.5, 0.5, 0.5))), ((0, 0, -1), 0.5, mat),
                ((-1.0, 0, -1), 0.5, mat), ((1.0, 0, -1.2), 0.5, brt.StandardMaterial(base_color=(0.2, 0.3, 0.8)))]
        bb = make_buffers(data)
        lvl, cam, win = uniforms(48, 27, spp=6, bounces=10, pos=(0, 0.3, 1.5), target=(0, 0, -1), fov=0.9, seed=0.37)
        render_both(plugin, oracle, bb, lvl, cam, win, 48, 27)


@pytest.mark.parametrize("bounces", [0, 1, 3])
def test_late_sample_ends_and_paths_taken_over(oracle, bounces):
    """shade_landed (brt_trace.h) makes the next sample's camera ray where a sample ends -- except for samples that end LATE
    (a metal hit absorbed or at the bounce limit), which get it at the top of the next round -- and the round loop leaves for
    the management code only when a wave has paths to hand over or to take over (the `need_cam` flag travels in the pool's
    records).  A scene of rough metal (absorbs often), glass and diffuse spheres at bounce limits 0 / 1 / 3, with the drain pool
    forced on (BRT_POOL_FORCE: a frame of this size runs without it by default) and off, three frames per context so that the general, the measuring and the LEAN instantiations all
    run: pixels and ray counts equal the oracle's; the COUNTERS instantiation (compiler-built rejection sampler and walk
    loop) agrees on all five counters."""
    rng = np.random.default_rng(77 + bounces)
    data = [((0, -1000.0, 0), 1000.0, brt.StandardMaterial(base_color=(0.5, 0.5, 0.5)))]
    for k in range(90):
        x, z = rng.uniform(-6, 6, 2)
        kind = k % 3
        mat = (brt.StandardMaterial(metallic=1.0, perceptual_roughness=float(rng.uniform(0.6, 1.0)), base_color=tuple(rng.uniform(0.3, 1.0, 3))),
               brt.StandardMaterial(specular_transmission=1.0, ior=float(rng.uniform(1.1, 1.8))),
               brt.StandardMaterial(base_color=tuple(rng.uniform(0.1, 0.9, 3)), perceptual_roughness=float(rng.uniform(0.0, 1.0))))[kind]
        data.append(((float(x), 0.3, float(z)), 0.3, mat))
    b = make_buffers(data)
    w, h = 320, 200
    lvl, cam, win = uniforms(w, h, spp=12, bounces=bounces, pos=(9.0, 2.2, 4.0), target=(0, 0.2, 0), fov=0.6, seed=0.41)
    want, cnt = oracle.render(b, lvl, cam, win, w, h)
    for force in (1, 0):      # a frame of this size runs WITHOUT the pool by default (plan_launch): BRT_POOL_FORCE=1 is the case under test
        with brt.RaytracePlugin([0]) as p:
            p.set_tuning("BRT_POOL_FORCE", force)
            for frame in range(3):
                got = p.node.run(lvl, cam, win, w, h, buffers=b)
                assert_frames_equal(got, want)
                assert p.node.last_stats["rays"] == cnt["rays"], frame
            lds_plain = p.node.last_stats["lds_bytes"]
            got = p.node.run(lvl, cam, win, w, h, buffers=b, flags=brt.FLAG_COUNTERS)
            assert_frames_equal(got, want)
            assert {k: p.node.last_stats[k] for k in COUNTER_KEYS} == cnt
            if force:
                with p.tuning(BRT_POOL_FORCE=0):
                    p.node.run(lvl, cam, win, w, h)
                    assert p.node.last_stats["lds_bytes"] < lds_plain      # the pool really was part of the forced launches


def test_levels_with_raster_inputs(plugin, oracle):
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    w, h = 40, 24
    rng = np.random.default_rng(5)
    raster = rng.random((h, w, 4), dtype=np.float32)
    depth = rng.random((h, w), dtype=np.float32) * np.float32(0.02)
    depth[0, :] = 0.0
    for level in brt.Raytracing:
        lvl, cam, win = brt.cover_camera(w, h, 2, 3, level, 0.5)
        got = plugin.node.run(lvl, cam, win, w, h, buffers=b, raster_rgba=raster, raster_depth=depth)
        want, _ = oracle.render(b, lvl, cam, win, w, h, raster_rgba=raster, raster_depth=depth)
        assert_frames_equal(got, want)
        got = plugin.node.run(lvl, cam, win, w, h)     # no raster inputs: cleared to 0
        want, _ = oracle.render(b, lvl, cam, win, w, h)
        assert_frames_equal(got, want)


def test_window_height_differs_from_target_height(plugin, oracle):
    # raytrace.wgsl:141-142: the jitter uses window.height, not the render target's size
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    lvl, cam, win = uniforms(64, 40, spp=3, bounces=3, pos=(13, 2, 3), target=(0, 0, 0), fov=0.4, seed=0.5, window_height=720)
    render_both(plugin, oracle, b, lvl, cam, win, 64, 40)


# ---- errors: the node skips the pass (pipeline.rs:82-151) ---------------------------------------------

def test_error_behaviour(plugin):
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    lvl, cam, win = brt.cover_camera(32, 18, 1, 1)
    assert plugin.node.run(None, None, None, 32, 18) is None
    assert plugin.node.run(lvl, cam, win, 32, 18, buffers=brt.Buffers(b.models[:0], b.materials, b.bvh)) is None   # empty -> skip
    bad = b.bvh.copy(); bad[0]["index"] = len(bad)
    with pytest.raises(brt.BrtError) as e:
        plugin.node.run(lvl, cam, win, 32, 18, buffers=brt.Buffers(b.models, b.materials, bad))
    assert e.value.code == -4
    with pytest.raises(brt.BrtError) as e:      # failed upload leaves no scene behind
        plugin.node.run(lvl, cam, win, 32, 18)
    assert e.value.code == -7
    ortho = cam.copy(); ortho["projection"] = 1
    plugin.node.write_buffers(b)
    with pytest.raises(brt.BrtError) as e:
        plugin.node.run(lvl, ortho, win, 32, 18)
    assert e.value.code == -8
    with pytest.raises(brt.BrtError):
        plugin.node.run(lvl, cam, win, 0, 18)


def test_sampler_stage_of_the_workgroup_renders_the_same_pixels(oracle):
    """The rejection sampler as a STAGE of the workgroup (knob BRT_BALL_SERVERS; brt_trace.h SRV, brt_device.h ball_server_asm): fourteen
    waves trace, two serve the other waves' samplers through mailboxes in LDS.  A lane's draws are the same hash chain whichever wave
    runs them: whole frames and ray counts equal the oracle's, in the steady-state instantiation (kernel_variant 2 + 32) the stage exists
    for; no pick-up runs into its bound."""
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    bb = brt.Buffers(b.models, b.materials, None)
    for (w, h, spp, bounces) in ((1920, 1080, 4, 2), (1920, 1080, 8, 8)):
        lvl, cam, win = brt.cover_camera(w, h, spp, bounces)
        want, cnt = oracle.render(b, lvl, cam, win, w, h)
        with brt.RaytracePlugin([0]) as p:
            p.set_tuning("BRT_BALL_SERVERS", 1)
            seen = set()
            for i in range(6):
                f = p.node.run(lvl, cam, win, w, h, buffers=bb if i == 0 else None)
                st = dict(p.node.last_stats)
                seen.add(st["kernel_variant"])
                assert st["rays"] == cnt["rays"], (i, st["kernel_variant"])
                assert_frames_equal(f, want)
            assert st["kernel_variant"] == 2 + 32, seen                    # the last frames ran in the instantiation with the stage
            p.debug_profile()
            stage = p.last_sampler_stage
            assert stage["gave_up"] == 0 and stage["iterations"] > 0 and stage["lanes"] > stage["iterations"], stage


def test_exception_barrier_on_exports_that_own_a_context(plugin, oracle):
    """include/bevyray_amd.h: no export throws across the boundary.  The knob BRT_TEST_THROW makes the next brt_upload_scene / brt_render*
    throw std::bad_alloc (1), std::logic_error (2) or a non-standard type (3) from inside the call: an error code and a text come back,
    the context stays usable and the next frame is the oracle's."""
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    w, h = 64, 40
    lvl, cam, win = brt.cover_camera(w, h, 2, 3)
    plugin.node.write_buffers(b)
    for kind, code, text in ((1, -11, "out of memory"), (2, -12, "logic_error"), (3, -12, "unknown error")):
        plugin.set_tuning("BRT_TEST_THROW", kind)
        with pytest.raises(brt.BrtError) as e:
            plugin.node.run(lvl, cam, win, w, h)
        assert e.value.code == code and text in str(e.value), (kind, e.value.code, str(e.value))
        assert plugin.get_tuning("BRT_TEST_THROW")[0] == 0
    plugin.set_tuning("BRT_TEST_THROW", 1)
    changed = b.models.copy(); changed["radius"][5] *= 1.5
    with pytest.raises(brt.BrtError) as e:                      # through brt_upload_scene (the bytes differ: no dirty-tracking shortcut)
        plugin.node.write_buffers(brt.Buffers(changed, b.materials, None))
    assert e.value.code == -11 and "out of memory" in str(e.value)
    render_both(plugin, oracle, b, lvl, cam, win, w, h)


# ---- strips / devices ---------------------------------------------------------------------------------------

@pytest.mark.parametrize("n_parts", [2, 3, 8])
def test_parts_assemble_to_the_full_frame(plugin, oracle, n_parts):
    import torch
    from bevyray_amd.parallel import frame_rows_of_part
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    w, h = 72, 45
    lvl, cam, win = brt.cover_camera(w, h, 2, 4)
    full = plugin.node.run(lvl, cam, win, w, h, buffers=b)
    rows = brt.tile_rows(h, n_parts)
    tiles = torch.zeros((n_parts, rows, w, 4), dtype=torch.float32, device="cuda")
    total_rays = 0
    for p in range(n_parts):
        st = plugin.node.render_part_device(lvl, cam, win, w, h, p, n_parts, tiles[p].data_ptr())
        total_rays += st["rays"]
        t = tiles[p].cpu().numpy()
        fr = frame_rows_of_part(h, p, n_parts)
        assert_frames_equal(t[fr >= 0], full[fr[fr >= 0]])
    assert total_rays == plugin.node.last_stats["rays"]
    frame = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    plugin.node.deinterleave_device(tiles.data_ptr(), n_parts, w, h, frame.data_ptr())
    torch.cuda.synchronize()
    assert_frames_equal(frame.cpu().numpy(), full)


@pytest.mark.parametrize("n_parts", [2, 3, 8])
def test_parts_by_a_strip_table_assemble_to_the_full_frame(oracle, n_parts):
    """Strips dealt out by a table instead of s % n_parts (brt_set_strip_table; brt_plan_strips makes one from measured costs): every
    part's tile holds the rows the table says, the tiles assemble to the oracle's frame through brt_deinterleave_device and through
    the RCCL gather's one-rank form, the rays add up, a table with a part twice in a group is refused, NULL brings s % n_parts back."""
    import torch
    from bevyray_amd.parallel import check_strip_table, frame_rows_of_part
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    w, h = 200, 149                                  # 19 strips: the last group of every split is partial
    lvl, cam, win = brt.cover_camera(w, h, 2, 4)
    want, cnt = oracle.render(b, lvl, cam, win, w, h)
    rows = brt.tile_rows(h, n_parts)
    strips = (h + 7) // 8
    rng = np.random.default_rng(40 + n_parts)
    shuffled = np.zeros(strips, np.uint32)
    for g in range(0, strips, n_parts):
        n = min(n_parts, strips - g)
        shuffled[g:g + n] = rng.permutation(n_parts)[:n]
    with brt.RaytracePlugin([0]) as p:
        p.node.write_buffers(b)

        def render_parts(table):
            tiles = torch.zeros((n_parts, rows, w, 4), dtype=torch.float32, device="cuda")
            total = 0
            for part in range(n_parts):
                st = p.node.render_part_device(lvl, cam, win, w, h, part, n_parts, tiles[part].data_ptr())
                total += st["rays"]
                fr = frame_rows_of_part(h, part, n_parts, table)
                t = tiles[part].cpu().numpy()
                assert_frames_equal(t[fr >= 0], want[fr[fr >= 0]])
                assert np.all(t[fr < 0] == 0.0)                                   # padding rows are never written
                assert st["paths"] == int((fr >= 0).sum()) * w * 2
            assert total == cnt["rays"]
            frame = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
            p.node.deinterleave_device(tiles.data_ptr(), n_parts, w, h, frame.data_ptr())
            torch.cuda.synchronize()
            assert_frames_equal(frame.cpu().numpy(), want)

        render_parts(None)
        p.set_strip_table(n_parts, shuffled)
        render_parts(shuffled)
        planned = p.plan_strips(lvl, cam, win, w, h, n_parts, probe_spp=2)
        check_strip_table(planned, h, n_parts)
        assert np.array_equal(planned, p.plan_strips(lvl, cam, win, w, h, n_parts, probe_spp=2))      # deterministic
        render_parts(planned)
        bad = shuffled.copy(); bad[1] = bad[0]
        with pytest.raises(brt.BrtError) as e:
            p.set_strip_table(n_parts, bad)
        assert e.value.code == -1
        p.set_strip_table(n_parts, None)
        render_parts(None)


def test_multi_device_context_on_one_gpu(oracle):
    # brt_create([0, 0, 0]): three sub-contexts on the same GPU exercise the strip split + copy-out
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    lvl, cam, win = brt.cover_camera(80, 50, 2, 4)
    with brt.RaytracePlugin([0, 0, 0]) as p3:
        got = p3.node.run(lvl, cam, win, 80, 50, buffers=b, flags=brt.FLAG_COUNTERS)
        want, cnt = oracle.render(b, lvl, cam, win, 80, 50)
        assert_frames_equal(got, want)
        assert {k: p3.node.last_stats[k] for k in COUNTER_KEYS} == cnt


def _render_device_frames(device_ids, oracle):
    """brt_render_device on a context over `device_ids`: frames with a new seed each (a stale tile or a gather buffer
    reused too early would show), the synchronous form and the caller-stream form, level 3 and level 2 with device
    raster inputs -- all against the oracle, bit for bit, counters included."""
    import torch
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    w, h = 200, 117                                       # 15 strips, the last one 5 rows: ragged shares
    rng = np.random.default_rng(3)
    raster = rng.random((h, w, 4), dtype=np.float32)
    depth = (rng.random((h, w), dtype=np.float32) * np.float32(0.05)).astype(np.float32)
    with brt.RaytracePlugin(device_ids) as p:
        p.node.write_buffers(b)
        torch.cuda.set_device(device_ids[0])
        frame = torch.full((h, w, 4), -1.0, dtype=torch.float32, device="cuda")
        d_raster, d_depth = torch.from_numpy(raster).cuda(), torch.from_numpy(depth).cuda()
        torch.cuda.synchronize()
        for i in range(5):
            lvl, cam, win = brt.cover_camera(w, h, 3, 5, seed=0.07 + 0.17 * i)
            st = p.node.render_device(lvl, cam, win, w, h, frame.data_ptr(), flags=brt.FLAG_COUNTERS)
            want, cnt = oracle.render(b, lvl, cam, win, w, h)
            assert_frames_equal(frame.cpu().numpy(), want)
            assert {k: st[k] for k in COUNTER_KEYS} == cnt and st["paths"] == w * h * 3
        # asynchronous on torch's current stream (handle 0 = the default stream), a new seed per frame, no sync in between
        stream = torch.cuda.current_stream().cuda_stream
        wants = []
        frames = [torch.empty_like(frame) for _ in range(4)]
        for i, f in enumerate(frames):
            lvl, cam, win = brt.cover_camera(w, h, 2, 4, seed=0.21 + 0.19 * i)
            p.node.render_device(lvl, cam, win, w, h, f.data_ptr(), stream=stream)
            wants.append(oracle.render(b, lvl, cam, win, w, h)[0])
        torch.cuda.synchronize()
        for f, want in zip(frames, wants):
            assert_frames_equal(f.cpu().numpy(), want)
        # depth blend with raster inputs on the first device (forwarded to the others), and the passthrough level
        n = len(device_ids)
        for level in (brt.Raytracing.FallbackRaytraced, brt.Raytracing.FallbackRaster, brt.Raytracing.Skip):
            lvl, cam, win = brt.cover_camera(w, h, 2, 4, level)
            st = p.node.render_device(lvl, cam, win, w, h, frame.data_ptr(), d_raster.data_ptr(), d_depth.data_ptr())
            want, _ = oracle.render(b, lvl, cam, win, w, h, raster_rgba=raster, raster_depth=depth)
            assert_frames_equal(frame.cpu().numpy(), want)
            # only each device's own strips of the raster inputs travel (the whole frame went to every device): the
            # tiles of devices 1 .. n-1, colour (+ depth unless the level is the passthrough, which reads no depth)
            strips = brt.tile_rows(h, n) * w * (16 + (4 if level != brt.Raytracing.Skip else 0)) * (n - 1)
            assert st["forwarded_bytes"] == strips and strips <= 1.1 * (n - 1) / n * (h + 8 * n) * w * 20
        # ... and the host-pointer form (brt_render): every device is sent its strips by one strided copy
        for level in (brt.Raytracing.FallbackRaytraced, brt.Raytracing.FallbackRaster, brt.Raytracing.Skip):
            lvl, cam, win = brt.cover_camera(w, h, 2, 4, level, seed=0.61)
            got = p.node.run(lvl, cam, win, w, h, raster_rgba=raster, raster_depth=depth)
            want, _ = oracle.render(b, lvl, cam, win, w, h, raster_rgba=raster, raster_depth=depth)
            assert_frames_equal(got, want)


@pytest.mark.parametrize("ids", [[0], [0, 0], [0, 0, 0, 0, 0]])
def test_render_device_assembles_the_frame_on_the_first_device(oracle, ids):
    """The library's own N-device path with the gather ON the device (include/bevyray_amd.h brt_render_device): on one
    GPU with a repeated ordinal the peer copies degenerate to device copies; shares, gather buffer and de-interleave are
    the N-GPU ones."""
    _render_device_frames(ids, oracle)


def test_render_device_over_two_gpus(oracle):
    """The same over two real devices (tiles cross xGMI by hipMemcpyPeerAsync); skipped on the one-GPU boxes."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    _render_device_frames([0, 1], oracle)


# ---- BASELINE.json full size: size-independent properties + sampled rows --------------------------------------

def _timed_combination(plugin, b, lvl, cam, win, w, h, want, rays):
    """The exact combination bench.py times, against the oracle frame the caller already has: the tree the CALLEE builds (bvh = None),
    no flags, and the steady state of the view -- after the frames that measure (first frame of a view, a few frames after an upload)
    the dispatch order is the learned one, the tail runs as half-sample jobs and the production LEAN instantiation (hand-written walk
    loops and sampler) renders.  EVERY one of the six frames: whole frame + ray count."""
    bb = brt.Buffers(b.models, b.materials, None)
    st = None
    for i in range(6):
        f = plugin.node.run(lvl, cam, win, w, h, buffers=bb if i == 0 else None)
        st = dict(plugin.node.last_stats)
        assert st["rays"] == rays, (i, st["rays"], rays)
        assert_frames_equal(f, want)
    assert st["kernel_variant"] in (1, 2) and st["measured_tile_costs"] == 0, st       # LEAN instantiation, nothing measured: the timed state
    return st


def test_config2_full_size_properties(plugin, oracle):
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    w, h, spp, bounces = 1920, 1080, 64, 8
    lvl, cam, win = brt.cover_camera(w, h, spp, bounces)
    f1 = plugin.node.run(lvl, cam, win, w, h, buffers=b)
    s1 = dict(plugin.node.last_stats)
    f2 = plugin.node.run(lvl, cam, win, w, h)
    assert np.array_equal(f1.view(np.uint32), f2.view(np.uint32)) and plugin.node.last_stats["rays"] == s1["rays"]   # idempotent
    assert s1["paths"] == w * h * spp and w * h * spp <= s1["rays"] <= w * h * spp * (bounces + 1)
    assert np.all(f1[..., 3] == 1.0) and np.all(np.isfinite(f1)) and f1[..., :3].min() >= 0.0 and f1[..., :3].max() <= 1.0
    # the WHOLE frame and its ray count against the oracle, bit for bit (the headline config: ~5 s of oracle on the box's 16 CPUs)
    want, cnt = oracle.render(b, lvl, cam, win, w, h)
    assert s1["rays"] == cnt["rays"]
    assert_frames_equal(f1, want)
    # ... and through the tree the callee builds on the GPU (what bench.py times): same pixels, the oracle's counters on that tree
    f3 = plugin.node.run(lvl, cam, win, w, h, buffers=brt.Buffers(b.models, b.materials, None), flags=brt.FLAG_COUNTERS)
    s3 = dict(plugin.node.last_stats)
    assert_frames_equal(f3, want)
    _, cnt3 = oracle.render(brt.Buffers(b.models, b.materials, brt.build_bvh_sah(b.models)), lvl, cam, win, w, h)
    assert {k: s3[k] for k in COUNTER_KEYS} == cnt3
    st = _timed_combination(plugin, b, lvl, cam, win, w, h, want, cnt["rays"])
    assert st["kernel_variant"] == 2 and st["scene_in_lds"] == 1


def _frame_properties(f, stats, w, h, spp, bounces, n_px=None):
    n_px = w * h if n_px is None else n_px
    assert stats["paths"] == n_px * spp and n_px * spp <= stats["rays"] <= n_px * spp * (bounces + 1)
    assert np.all(f[..., 3] == 1.0) and np.all(np.isfinite(f)) and f[..., :3].min() >= 0.0 and f[..., :3].max() <= 1.0


def test_config3_rtiow_full_size(plugin, oracle):
    """BASELINE.json configs[2]: RTIOW final scene 1920x1080, 256 spp, 50 bounces -- the WHOLE frame and its ray
    count against the oracle (pixel chains of > 10 000 sequential rays, critical-pixel waves)."""
    b = brt.generate_scene(brt.SCENE_RTIOW_FINAL, 1)
    w, h, spp, bounces = 1920, 1080, 256, 50
    lvl, cam, win = brt.rtiow_camera(w, h, spp, bounces)
    f1 = plugin.node.run(lvl, cam, win, w, h, buffers=b)          # first frame of the view: raster order
    s1 = dict(plugin.node.last_stats)
    f2 = plugin.node.run(lvl, cam, win, w, h)                     # second: order learned from the first (critical tiles first)
    assert np.array_equal(f1.view(np.uint32), f2.view(np.uint32)) and plugin.node.last_stats["rays"] == s1["rays"]
    _frame_properties(f1, s1, w, h, spp, bounces)
    # the oracle over EVERY row (about 1.3 G rays: ~20-30 s on the box's 16 CPUs): whole frame + ray count
    want, cnt = oracle.render(b, lvl, cam, win, w, h)
    assert s1["rays"] == cnt["rays"]
    assert_frames_equal(f1, want)
    # what bench.py times -- the callee-built tree, no flags, steady state -- against the oracle ON THAT TREE (a second pass of the
    # oracle).  The two trees' frames differ in ONE pixel here, (704, 396): its 205th ray meets spheres 166 and 482 at the same f32
    # distance 16.89793, an exact tie that the strict `<` of raytrace.wgsl:353 gives to whichever sphere the walk reaches first --
    # topology dependent in the reference too (brute force in model order sides with the caller's tree).
    want_sah, cnt_sah = oracle.render(brt.Buffers(b.models, b.materials, brt.build_bvh_sah(b.models)), lvl, cam, win, w, h)
    assert int((want_sah.view(np.uint32) != want.view(np.uint32)).any(axis=2).sum()) <= 1
    _timed_combination(plugin, b, lvl, cam, win, w, h, want_sah, cnt_sah["rays"])


@pytest.mark.parametrize("part", [0, 7])
def test_config4_4k_1024spp_one_rank_share(plugin, oracle, part):
    """BASELINE.json configs[3]: 3840x2160, 1024 spp, 8 bounces, row-tiled over 8 ranks -- rank `part`'s share
    (every 8th strip of 8 rows) rendered into a device tile as bench.py does it, 33+ of its rows against the oracle."""
    import torch
    from bevyray_amd.parallel import frame_rows_of_part
    b = brt.generate_scene(brt.SCENE_RTIOW_FINAL, 1)
    w, h, spp, bounces, n_parts = 3840, 2160, 1024, 8, 8
    lvl, cam, win = brt.rtiow_camera(w, h, spp, bounces)
    plugin.node.write_buffers(b)
    fr = frame_rows_of_part(h, part, n_parts)
    tile = torch.zeros((len(fr), w, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    st = plugin.node.render_part_device(lvl, cam, win, w, h, part, n_parts, tile.data_ptr())
    t = tile.cpu().numpy()
    _frame_properties(t[fr >= 0], st, w, h, spp, bounces, n_px=int((fr >= 0).sum()) * w)
    assert np.all(t[fr < 0] == 0.0)                               # padding rows of the last strip are never written
    # rows part*8 + 3 + 64 j all lie in strips of this part (strip s -> part s % 8): one oracle call, a thread per row
    r0 = part * 8 + 3
    want, _ = oracle.render(b, lvl, cam, win, w, h, rows=(r0, h), row_step=64)
    ys = np.arange(r0, h, 64)
    assert len(ys) >= 12
    tile_row_of = {int(y): k for k, y in enumerate(fr) if y >= 0}
    assert_frames_equal(t[[tile_row_of[int(y)] for y in ys]], want[ys])


def test_config5_10k_spheres_full_size(plugin, oracle):
    """BASELINE.json configs[4]: 10 004 spheres, 1920x1080, 64 spp, 8 bounces -- the scene does not fit LDS
    (pair records from L2); the WHOLE frame and its ray count against the oracle."""
    b = brt.generate_scene(brt.SCENE_STRESS_GRID, 1)
    assert len(b.models) == 10004
    w, h, spp, bounces = 1920, 1080, 64, 8
    lvl, cam, win = brt.cover_camera(w, h, spp, bounces)
    f1 = plugin.node.run(lvl, cam, win, w, h, buffers=b)
    s1 = dict(plugin.node.last_stats)
    assert s1["scene_in_lds"] in (0, 2)   # not LDS resident: all from L2, or the top of the tree in an LDS tile
    f2 = plugin.node.run(lvl, cam, win, w, h)
    assert np.array_equal(f1.view(np.uint32), f2.view(np.uint32)) and plugin.node.last_stats["rays"] == s1["rays"]
    _frame_properties(f1, s1, w, h, spp, bounces)
    want, cnt = oracle.render(b, lvl, cam, win, w, h)
    assert s1["rays"] == cnt["rays"]
    assert_frames_equal(f1, want)
    # the same frame through the tree the CALLEE builds when the caller passes none (binned SAH, what INTEGRATION.md
    # recommends): same pixels -- the topology only matters through ties and overflow -- and exactly the oracle's
    # counters on that tree, with fewer node visits than in the caller's PLOC tree
    f3 = plugin.node.run(lvl, cam, win, w, h, buffers=brt.Buffers(b.models, b.materials, None), flags=brt.FLAG_COUNTERS)
    s3 = dict(plugin.node.last_stats)
    assert_frames_equal(f3, want)
    want3, cnt3 = oracle.render(brt.Buffers(b.models, b.materials, brt.build_bvh_sah(b.models)), lvl, cam, win, w, h)
    assert_frames_equal(want3, want)
    assert {k: s3[k] for k in COUNTER_KEYS} == cnt3
    assert cnt3["interior_visits"] < 0.85 * cnt["interior_visits"]
    st = _timed_combination(plugin, b, lvl, cam, win, w, h, want, cnt["rays"])
    assert st["scene_in_lds"] == 2 and st["hot_records"] != 0      # the top of the tree in the LDS tile, records numbered by use


# ---- GPU BVH build --------------------------------------------------------------------

def test_gpu_ploc_grid_build_of_a_large_scene(plugin):
    """200 000 spheres (above the one-workgroup limit: radix sort + grid-wide PLOC rounds): byte-identical to the
    CPU builder, and a valid tree for brt_upload_scene."""
    rng = np.random.default_rng(5)
    n = 200_000
    m = np.zeros(n, brt.MODEL_DTYPE)
    m["position"] = rng.uniform(-300, 300, (n, 3)).astype(np.float32)
    m["position"][:, 1] = np.abs(m["position"][:, 1]) * 0.02
    m["radius"] = rng.uniform(0.05, 0.6, n).astype(np.float32)
    m["position"][::1000] = m["position"][1::1000]            # duplicates: equal Morton codes and areas
    m["radius"][::1000] = m["radius"][1::1000]
    gpu, ms = plugin.build_bvh(m)
    cpu = brt.build_bvh(m)
    assert len(gpu) == 2 * n - 1 and np.array_equal(cpu.view(np.uint8), gpu.view(np.uint8))
    mats = np.zeros(1, brt.MATERIAL_DTYPE)
    mats["base_color"] = 0.5
    assert brt.validate_scene(m, mats, gpu) < 64



@pytest.mark.parametrize("grid", [False, True])
def test_gpu_ploc_build_is_byte_identical_to_cpu_build(oracle, grid):
    # grid: the multi-kernel version of the builder (used above 6 000 spheres) forced on every size
    with brt.RaytracePlugin([0]) as plugin:
        if grid:
            plugin.set_tuning("BRT_PLOC_ONE_BLOCK_MAX", 0)
        _ploc_build_cases(plugin, oracle)


def _ploc_build_cases(plugin, oracle):
    rng = np.random.default_rng(11)
    scenes = [brt.generate_scene(brt.SCENE_COVER, s).models for s in (1, 2, 3)]
    scenes += [brt.generate_scene(brt.SCENE_RTIOW_FINAL, 1).models, brt.generate_scene(brt.SCENE_STRESS_GRID, 1).models]
    for n in (1, 2, 3, 5, 24, 25, 26, 49, 50, 51, 1023, 1024, 1025):
        m = np.zeros(n, brt.MODEL_DTYPE)
        m["position"] = rng.uniform(-20, 20, (n, 3)).astype(np.float32)
        m["radius"] = rng.uniform(0.05, 2.0, n).astype(np.float32)
        scenes.append(m)
    dup = np.zeros(300, brt.MODEL_DTYPE); dup["position"] = (1.0, 2.0, 3.0); dup["radius"] = 0.5    # all Morton codes and areas tie
    line = np.zeros(200, brt.MODEL_DTYPE); line["position"][:, 0] = np.arange(200, dtype=np.float32); line["radius"] = 0.25
    scenes += [dup, line]
    for bad in (np.nan, np.inf, -np.inf, 3e38):       # non-finite spheres: same tree, same (canonical) NaN bits
        m = np.zeros(150, brt.MODEL_DTYPE)
        m["position"] = rng.uniform(-5, 5, (150, 3)).astype(np.float32)
        m["radius"] = rng.uniform(0.1, 1.0, 150).astype(np.float32)
        m["position"][::7, 2] = bad
        m["radius"][3::11] = bad
        scenes.append(m)
    for models in scenes:
        cpu = brt.build_bvh(models)
        gpu, ms = plugin.build_bvh(models)
        assert len(cpu) == len(gpu) == 2 * len(models) - 1
        assert np.array_equal(cpu.view(np.uint8), gpu.view(np.uint8)), f"n={len(models)}"
        assert ms >= 0.0
    # and the callee-built path renders the same pixels as the caller-built one
    b = brt.generate_scene(brt.SCENE_STRESS_GRID, 1)
    lvl, cam, win = brt.cover_camera(96, 54, 2, 4)
    f1 = plugin.node.run(lvl, cam, win, 96, 54, buffers=b, flags=brt.FLAG_COUNTERS)
    s1 = dict(plugin.node.last_stats)
    f2 = plugin.node.run(lvl, cam, win, 96, 54, buffers=brt.Buffers(b.models, b.materials, None), flags=brt.FLAG_COUNTERS)
    assert_frames_equal(f1, f2) and all(plugin.node.last_stats[k] == s1[k] for k in COUNTER_KEYS)


def _sah_build_scenes():
    rng = np.random.default_rng(23)
    scenes = [brt.generate_scene(brt.SCENE_COVER, s).models for s in (1, 2)]
    scenes += [brt.generate_scene(brt.SCENE_RTIOW_FINAL, 1).models, brt.generate_scene(brt.SCENE_STRESS_GRID, 1).models]
    # sizes around every hand-over of the builder: 1 (a leaf root), wave split (<= 192), block split, the 1024-sphere subtree
    # limit (k_sah_top above it), several subtrees, the 65 536-sphere limit of the SAH path
    for n in (1, 2, 3, 4, 5, 63, 64, 65, 191, 192, 193, 500, 1023, 1024, 1025, 2049, 4100, 20000, 65536):
        m = np.zeros(n, brt.MODEL_DTYPE)
        m["position"] = rng.uniform(-20, 20, (n, 3)).astype(np.float32)
        m["radius"] = rng.uniform(0.05, 2.0, n).astype(np.float32)
        scenes.append(m)
    dup = np.zeros(300, brt.MODEL_DTYPE); dup["position"] = (1.0, 2.0, 3.0); dup["radius"] = 0.5     # no axis has an extent: halves
    line = np.zeros(1500, brt.MODEL_DTYPE); line["position"][:, 0] = np.arange(1500, dtype=np.float32); line["radius"] = 0.25   # one usable axis
    plane = np.zeros(3000, brt.MODEL_DTYPE); plane["position"][:, [0, 2]] = rng.uniform(-9, 9, (3000, 2)).astype(np.float32); plane["radius"] = 0.1
    row = np.zeros(2200, brt.MODEL_DTYPE)      # geometric row: lopsided SAH splits run into the depth budget (28 levels)
    row["position"][:, 0] = (1.5 ** (np.arange(2200) % 180)).astype(np.float32); row["position"][:, 1] = np.arange(2200) // 180; row["radius"] = 0.01
    clump = np.zeros(5000, brt.MODEL_DTYPE)    # many equal centroids among distinct ones: bins of very different fill
    clump["position"] = np.round(rng.normal(0, 3, (5000, 3))).astype(np.float32); clump["radius"] = 0.3
    zeros = np.zeros(700, brt.MODEL_DTYPE)     # box bounds of both zero signs: -0.0 from 0.25 - (0.15 + 0.1) style sums
    zeros["position"] = rng.choice(np.array([-0.35, 0.35, 0.0, -0.0], np.float32), (700, 3)); zeros["radius"] = 0.25
    scenes += [dup, line, plane, row, clump, zeros]
    for bad in (np.nan, np.inf, -np.inf, 3e38, -1.0):      # non-finite spheres, negative radii: same tree, same (canonical) NaN bits
        for n in (150, 2500):
            m = np.zeros(n, brt.MODEL_DTYPE)
            m["position"] = rng.uniform(-5, 5, (n, 3)).astype(np.float32)
            m["radius"] = rng.uniform(0.1, 1.0, n).astype(np.float32)
            m["position"][::7, 2] = bad
            m["radius"][3::11] = bad
            scenes.append(m)
    return scenes


def test_gpu_sah_build_is_byte_identical_to_cpu_build(plugin, oracle):
    """SURVEY 8(f1) for the tree the product recommends: brt_sah.hip (what brt_upload_scene runs when the caller passes no BVH)
    writes the bytes of the CPU statement of the rule (brt_build_bvh_sah; extract.rs:315-332 is what both replace)."""
    for models in _sah_build_scenes():
        cpu = brt.build_bvh_sah(models)
        gpu, ms = plugin.build_bvh_sah(models)
        assert len(cpu) == len(gpu) == 2 * len(models) - 1
        assert np.array_equal(cpu.view(np.uint8), gpu.view(np.uint8)), f"n={len(models)}"
        assert brt.validate_scene(models, np.zeros(max(1, int(models["material_id"].max()) + 1), brt.MATERIAL_DTYPE), gpu) <= 28
        # and again: the build is deterministic (LDS atomics on integer keys, nothing depends on arrival order)
        again, _ = plugin.build_bvh_sah(models)
        assert np.array_equal(gpu.view(np.uint8), again.view(np.uint8))
        # ... and for a camera further out than the scene's own extent (larger leaf pads: `reach`, brt_sah.h)
        for reach in (150.0, 1000.0, 3.0e38):
            assert np.array_equal(brt.build_bvh_sah(models, reach).view(np.uint8), plugin.build_bvh_sah(models, reach)[0].view(np.uint8)), (len(models), reach)
    # the upload path: pixels and all five counters through the GPU-built tree equal the oracle's on the CPU twin's tree
    b = brt.generate_scene(brt.SCENE_STRESS_GRID, 1)
    lvl, cam, win = brt.cover_camera(96, 54, 2, 4)
    got = plugin.node.run(lvl, cam, win, 96, 54, buffers=brt.Buffers(b.models, b.materials, None), flags=brt.FLAG_COUNTERS)
    stats = dict(plugin.node.last_stats)
    want, cnt = oracle.render(brt.Buffers(b.models, b.materials, brt.build_bvh_sah(b.models)), lvl, cam, win, 96, 54)
    assert_frames_equal(got, want)
    assert {k: stats[k] for k in COUNTER_KEYS} == cnt
    with plugin.tuning(BRT_CPU_BVH=1):          # the CPU twin inside brt_upload_scene: the same
        got2 = plugin.node.run(lvl, cam, win, 96, 54, buffers=brt.Buffers(b.models, b.materials, None), flags=brt.FLAG_COUNTERS)
        assert {k: plugin.node.last_stats[k] for k in COUNTER_KEYS} == cnt
    assert_frames_equal(got2, want)


def test_bringup_kernel_after_frames_with_half_sample_jobs(plugin, oracle):
    """a dispatch order with half-sample jobs has n_tiles + n_split entries laid out [non-sky | second halves | sky], which
    only the persistent kernel understands; the bring-up kernel (BRT_FLAG_KERNEL_SIMPLE) indexed it as a plain order -- split tiles
    twice, the last sky tiles never, stale pixels of the previous camera there.  It now runs in raster order whatever the history."""
    b = fixture_buffers()[0]
    w, h = 192, 104
    with plugin.tuning(BRT_SPLIT_FORCE=40):
        lvl, cam, win = uniforms(w, h, 16, 4, (13.0, 2.0, 3.0), (0.0, 0.0, 0.0), 0.4, 0.5)
        f1 = plugin.node.run(lvl, cam, win, w, h, buffers=b)
        f2 = plugin.node.run(lvl, cam, win, w, h)                       # in the order the first frame measured: half-sample jobs
        plugin.debug_profile()
        assert plugin.last_order_meta["split_tiles"] > 0
        assert_frames_equal(f1, f2)
        lvl, cam, win = uniforms(w, h, 16, 4, (11.0, 3.5, -4.0), (0.5, 0.0, 0.0), 0.5, 0.25)      # another camera, another seed
        got = plugin.node.run(lvl, cam, win, w, h, flags=brt.FLAG_KERNEL_SIMPLE | brt.FLAG_COUNTERS)
        st = dict(plugin.node.last_stats)
        want, cnt = oracle.render(b, lvl, cam, win, w, h)
        assert_frames_equal(got, want)
        assert {q: st[q] for q in COUNTER_KEYS} == cnt
        assert_frames_equal(plugin.node.run(lvl, cam, win, w, h), want)      # and the persistent kernel again, same view


def test_records_numbered_by_visits_same_pixels_and_counters(oracle):
    """a scene whose tree does not fit the LDS is walked from a tile of the first K pair records in LDS and the rest
    from L2.  The pre-pass of a first frame counts the interior visits per record and the records are re-numbered by them (most
    visited first), so that the tile holds what THIS view walks instead of the breadth-first top of the tree
    (brt_api.cpp apply_hot_order).  Only the numbering changes: pixels, ray count and all five counters equal the oracle's, with
    the order on and off, after a camera jump (counted again, permutations compose), with a tile of a handful of records, over three
    sub-contexts, and in the counting and the production instantiations."""
    b = brt.generate_scene(brt.SCENE_STRESS_GRID, 1)
    nb = brt.Buffers(b.models, b.materials, None)
    w, h, spp, bounces = 640, 360, 64, 4
    views = [brt.cover_camera(w, h, spp, bounces), uniforms(w, h, spp, bounces, (-30.0, 6.0, 22.0), (10.0, 0.0, -5.0), 0.7, 0.31)]
    # (the second camera stands further out than the scene's own extent covers: its frames run in the tree rebuilt for its reach)
    wants = [oracle.render(brt.Buffers(b.models, b.materials, brt.build_bvh_sah(b.models, brt.tree_reach(b.models, v[1])[2])), *v, w, h) for v in views]
    assert brt.tree_reach(b.models, views[0][1])[1] == 0 and brt.tree_reach(b.models, views[1][1])[1] > 0
    for ids, knobs in (([0], {}), ([0], {"BRT_HOT_RECORDS": 0}), ([0], {"BRT_FORCE_LDS_TOP": 70}), ([0, 0, 0], {})):
        with brt.RaytracePlugin(ids) as p:
            for k, v in knobs.items():
                p.set_tuning(k, v)
            for (lvl, cam, win), (want, cnt) in zip(views, wants):         # the second view is a camera jump: pre-pass, counted again
                for frame in range(3):
                    got = p.node.run(lvl, cam, win, w, h, buffers=nb if frame == 0 else None)
                    st = dict(p.node.last_stats)
                    assert_frames_equal(got, want)
                    assert st["rays"] == cnt["rays"] and st["scene_in_lds"] == 2
                    assert (st["hot_records"] > 1000) == (knobs.get("BRT_HOT_RECORDS", 1) == 1), st
                got = p.node.run(lvl, cam, win, w, h, flags=brt.FLAG_COUNTERS)
                assert_frames_equal(got, want)
                assert {q: p.node.last_stats[q] for q in COUNTER_KEYS} == cnt
    # an animated scene (the reference re-uploads every frame, extract.rs:299-336): a tree of the same shape goes up in the numbering
    # the device last counted -- no pre-pass per upload, the frames are the oracle's on each new scene
    with brt.RaytracePlugin([0]) as p:
        lvl, cam, win = views[0]
        for frame in range(3):
            p.node.run(lvl, cam, win, w, h, buffers=nb if frame == 0 else None)
        assert p.node.last_stats["hot_records"] > 1000
        moving = b.models.copy()
        kept = 0
        for step in range(4):
            moving["position"][7 + step, 0] += np.float32(0.002)
            got = p.node.run(lvl, cam, win, w, h, buffers=brt.Buffers(moving, b.materials, None))
            st = dict(p.node.last_stats)
            want, cnt = oracle.render(brt.Buffers(moving, b.materials, brt.build_bvh_sah(moving)), lvl, cam, win, w, h)
            assert_frames_equal(got, want)
            assert st["rays"] == cnt["rays"] and st["prepass_ms"] == 0.0
            kept += 1 if st["hot_records"] > 1000 else 0
        assert kept >= 2, kept            # (a sphere that crosses a bin border of the SAH build changes the tree's shape: breadth first then)
        got = p.node.run(lvl, cam, win, w, h, flags=brt.FLAG_COUNTERS)
        assert {q: p.node.last_stats[q] for q in COUNTER_KEYS} == cnt
    # a camera that keeps moving: the numbering is counted again once the picture has moved a quarter of the frame's height since
    # (a pre-pass; frames stay the oracle's)
    with brt.RaytracePlugin([0]) as p:
        tree0 = brt.Buffers(b.models, b.materials, brt.build_bvh_sah(b.models))
        prepasses = 0
        for i in range(7):
            a = np.deg2rad(3.0 * i)
            pos = (13.0 * np.cos(a) - 3.0 * np.sin(a), 2.0, 13.0 * np.sin(a) + 3.0 * np.cos(a))
            lvl, cam, win = uniforms(w, h, spp, bounces, tuple(float(x) for x in pos), (0.0, 0.0, 0.0), 0.4, 0.5)
            assert brt.tree_reach(b.models, cam)[1] == 0
            got = p.node.run(lvl, cam, win, w, h, buffers=nb if i == 0 else None)
            st = dict(p.node.last_stats)
            want, cnt = oracle.render(tree0, lvl, cam, win, w, h)
            assert_frames_equal(got, want)
            assert st["rays"] == cnt["rays"] and st["hot_records"] > 1000
            prepasses += 1 if (i > 0 and st["prepass_ms"] > 0.0) else 0
        assert 1 <= prepasses <= 4, prepasses
    # a scene that fits the LDS is never re-numbered (nothing to gain), whatever the pre-pass does
    c = brt.generate_scene(brt.SCENE_COVER, 1)
    with brt.RaytracePlugin([0]) as p:
        lvl, cam, win = brt.cover_camera(320, 180, 64, 4)
        p.node.run(lvl, cam, win, 320, 180, buffers=brt.Buffers(c.models, c.materials, None))
        assert p.node.last_stats["hot_records"] == 0 and p.node.last_stats["scene_in_lds"] == 1


def _far_camera(k, w, h, spp, bounces):
    """the cover view from k times the distance, field of view narrowed by k: the same picture, rays k times as long"""
    return uniforms(w, h, spp, bounces, (13.0 * k, 2.0 * k, 3.0 * k), (0.0, 0.0, 0.0), 0.4 / k, 0.5, far=1.0e5)


@pytest.mark.parametrize("kind", [brt.SCENE_COVER, brt.SCENE_STRESS_GRID])
def test_far_camera_callee_tree_is_rebuilt_for_the_camera(plugin, oracle, kind):
    """VERDICT r4 #1 (row H2, Model::aabb, extract.rs:220-227): the tree the callee builds pads leaf boxes by what the f32 tests need
    for the distances rays travel, and those depend on the camera.  Round 4's tree ignored it: 7 / 53 / 377 of 14 400 pixels wrong at
    x 20 / x 25 / x 30 the cover distance, where the reference's 0.1-padded tree is still exact.  Now every render call checks its
    camera and rebuilds the tree on the GPU when it needs larger pads (brt_stats::tree_rebuilt, tree_reach).  With bvh = None:
      * the frame and all five counters equal the oracle's in the CPU twin of the tree the context says it built;
      * the frame equals the oracle's frame in the reference's 0.1-padded PLOC tree at every distance where THAT tree equals the
        brute-force loop (one leaf of all spheres) -- x 1, x 20, x 30; at x 60 the reference's own tree disagrees with brute force in
        ~8 % of the pixels (f32 cannot resolve r = 0.2 at 800 units: the culling of ANY padded tree is marginal there, and which rays
        it loses depends on the visiting order), so there the callee's tree -- the same 0.1-padded leaf boxes in another topology --
        must agree with the reference's tree on all but a handful of the pixels the reference's tree gets right;
      * coming back to x 1 rebuilds the tight tree; small moves do not rebuild (steps of 2^(1/4) in reach, two steps of hysteresis)."""
    b = brt.generate_scene(kind, 1)
    nb = brt.Buffers(b.models, b.materials, None)
    brute = single_leaf_bvh(b.models)
    w, h, spp, bounces = 160, 90, 4, 4
    plugin.node.write_buffers(brt.generate_scene(brt.SCENE_RTIOW_FINAL, 1))     # (another scene first: the upload below is a real one)
    rebuilt = {}
    for k in (1, 20, 30, 60, 57, 1):
        lvl, cam, win = _far_camera(k, w, h, spp, bounces)
        got = plugin.node.run(lvl, cam, win, w, h, buffers=nb, flags=brt.FLAG_COUNTERS)
        st = dict(plugin.node.last_stats)
        rebuilt.setdefault(k, []).append(st["tree_rebuilt"])
        level, reach = brt.tree_reach(b.models, cam)[1:]
        twin = brt.build_bvh_sah(b.models, st["tree_reach"])
        # never a tree whose pads are below what this camera needs: built for at least its reach -- or, where every pad has already
        # reached the reference's 0.1 (the grid's pads do at x 30), the very same bytes
        assert st["tree_reach"] >= reach or np.array_equal(twin.view(np.uint8), brt.build_bvh_sah(b.models, reach).view(np.uint8))
        assert st["tree_reach"] == reach or k in (57, 60)
        want, cnt = oracle.render(brt.Buffers(b.models, b.materials, twin), lvl, cam, win, w, h)
        assert_frames_equal(got, want)
        assert {q: st[q] for q in COUNTER_KEYS} == cnt
        # the production instantiation too (hand-written walk loops): frame + ray count
        got2 = plugin.node.run(lvl, cam, win, w, h)
        assert_frames_equal(got2, want)
        assert plugin.node.last_stats["rays"] == cnt["rays"] and plugin.node.last_stats["tree_rebuilt"] == 0
        ref = oracle.render(b, lvl, cam, win, w, h)[0]
        truth = oracle.render(brt.Buffers(b.models, b.materials, brute), lvl, cam, win, w, h)[0]
        ref_ok = ~(ref.view(np.uint32) != truth.view(np.uint32)).any(axis=2)
        differs = (got.view(np.uint32) != ref.view(np.uint32)).any(axis=2)
        if k <= 30:
            assert ref_ok.all(), (k, int((~ref_ok).sum()))
            assert not differs.any(), (k, int(differs.sum()))
        else:
            assert (~ref_ok).sum() > 100                                     # the reference's own tree is past its limit here
            assert (differs & ref_ok).sum() <= 32, (k, int((differs & ref_ok).sum()))
            assert (got.view(np.uint32) != truth.view(np.uint32)).any(axis=2).sum() <= 1.05 * (~ref_ok).sum() + 16
    assert rebuilt[20] == [1] and rebuilt[30] == [1] and rebuilt[60] == [1 if kind == brt.SCENE_COVER else 0] and rebuilt[57] == [0]
    assert rebuilt[1] == [0, 1]                  # the upload's tree serves the cover camera; coming back from x 57 rebuilds the tight one
    # the tree of the scene's own extent does lose pixels out there (what round 4 shipped): the case is real
    lvl, cam, win = _far_camera(30, w, h, spp, bounces)
    blind = oracle.render(brt.Buffers(b.models, b.materials, brt.build_bvh_sah(b.models)), lvl, cam, win, w, h)[0]
    truth = oracle.render(brt.Buffers(b.models, b.materials, brute), lvl, cam, win, w, h)[0]
    if kind == brt.SCENE_COVER:
        assert (blind.view(np.uint32) != truth.view(np.uint32)).any(axis=2).sum() > 100


def test_far_camera_rebuild_on_the_device_entry_points(plugin, oracle):
    """The same check in front of brt_render_part_device / brt_render_device (every launch path looks at its camera), and a caller's
    tree is never touched."""
    import torch
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    w, h = 96, 56
    lvl, cam, win = _far_camera(25, w, h, 2, 4)
    plugin.node.write_buffers(brt.Buffers(b.models, b.materials, None))
    tile = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    st = plugin.node.render_part_device(lvl, cam, win, w, h, 0, 1, tile.data_ptr())
    assert st["tree_rebuilt"] == 1 and st["tree_reach"] == brt.tree_reach(b.models, cam)[2] > 0
    want, cnt = oracle.render(brt.Buffers(b.models, b.materials, brt.build_bvh_sah(b.models, st["tree_reach"])), lvl, cam, win, w, h)
    assert_frames_equal(tile.cpu().numpy(), want)
    assert st["rays"] == cnt["rays"]
    assert_frames_equal(want, oracle.render(b, lvl, cam, win, w, h)[0])        # = the frame in the reference's tree
    plugin.node.write_buffers(brt.Buffers(b.models, b.materials, None))         # same bytes: dirty tracking keeps the rebuilt tree
    frame = torch.zeros((h, w, 4), dtype=torch.float32, device="cuda")
    st = plugin.node.render_device(lvl, cam, win, w, h, frame.data_ptr())
    assert st["tree_rebuilt"] == 0 and st["tree_reach"] > 0
    assert_frames_equal(frame.cpu().numpy(), want)
    # a caller's tree is honoured as it comes
    plugin.node.write_buffers(b)
    st = plugin.node.render_device(lvl, cam, win, w, h, frame.data_ptr())
    assert st["tree_rebuilt"] == 0 and st["tree_reach"] == 0.0
    assert_frames_equal(frame.cpu().numpy(), want)


def test_gpu_sah_build_time(plugin):
    """VERDICT r3 task 1: <= 1.5 ms at 10 004 spheres, <= 0.3 ms at 506 (kernel time, HIP events; best of 5 after a warm-up).
    The bounds asserted here carry a 2x margin for a busy box; scripts/sah_time.py prints the figures (profiles/r04/)."""
    for kind, bound in ((brt.SCENE_COVER, 0.6), (brt.SCENE_STRESS_GRID, 3.0)):
        m = brt.generate_scene(kind, 1).models
        plugin.build_bvh_sah(m)
        best = min(plugin.build_bvh_sah(m)[1] for _ in range(5))
        print(f"GPU SAH build, {len(m)} spheres: {best:.3f} ms")
        assert best <= bound, (len(m), best)


@pytest.mark.parametrize("level", [0, 1, 2, 3])
def test_sample_count_zero_is_zero_over_zero(plugin, oracle, level):
    """raytrace.wgsl:161-170 with sample_count 0: every channel is 0/0 = NaN (and the depth test of
    levels 1/2 sees NaN, so the raster never wins).  Regression: a compiler fold once dropped two channels."""
    b = fixture_buffers()[0]
    w, h = 23, 9
    lvl, cam, win = uniforms(w, h, spp=0, bounces=4, pos=(13.0, 2.0, 3.0), target=(0.0, 0.0, 0.0), fov=0.4, seed=0.5,
                             level=brt.Raytracing(level))
    rng = np.random.default_rng(5)
    for raster, depth in ((None, None), (rng.random((h, w, 4), dtype=np.float32), rng.random((h, w), dtype=np.float32))):
        got, stats = render_both(plugin, oracle, b, lvl, cam, win, w, h, raster=raster, depth=depth)
        assert stats["rays"] == 0
        if level != 0:
            assert np.isnan(got[..., :3]).all() and (got[..., 3] == 1.0).all()


# ---- launch-shape and scheduling knobs must never change a pixel -------------------------------------------------

@pytest.mark.parametrize("env", [
    {"BRT_BLOCK_THREADS": "512", "BRT_WG_PER_CU": "2"}, {"BRT_BLOCK_THREADS": "256", "BRT_WG_PER_CU": "1"},
    {"BRT_FORCE_GLOBAL_SCENE": "1"}, {"BRT_REFILL_MIN": "64"}, {"BRT_REFILL_MIN": "17", "BRT_BOTTOM_UP": "1"},
    {"BRT_CPU_BVH": "1"}, {"BRT_WALK_EXIT": "0"}, {"BRT_WALK_EXIT": "1"}, {"BRT_WALK_EXIT": "23"},
    {"BRT_WALK_EXIT": "63", "BRT_REFILL_MIN": "9"}, {"BRT_WALK_EXIT": "40", "BRT_FORCE_GLOBAL_SCENE": "1"},
    {"BRT_LEAF_VOTE": "0"}, {"BRT_LEAF_VOTE": "64"}, {"BRT_LEAF_VOTE": "20", "BRT_WALK_EXIT": "0"},
    {"BRT_LEAF_VOTE": "3", "BRT_WALK_EXIT": "30", "BRT_BLOCK_THREADS": "256"},
    # drain pool: off, eager, tiny pool (donations that do not fit), other workgroup shapes.  BRT_POOL_FORCE=1: plan_launch
    # switches the pool off for launches of at most ~2.5 tiles per wave slot (every frame of this size), so the pool cases force it on
    {"BRT_DRAIN_DONATE": "0"}, {"BRT_DRAIN_DONATE": "56", "BRT_POOL_FORCE": "1"}, {"BRT_DRAIN_DONATE": "8", "BRT_POOL_CAP": "16", "BRT_POOL_FORCE": "1"},
    {"BRT_DRAIN_DONATE": "40", "BRT_BLOCK_THREADS": "256", "BRT_POOL_FORCE": "1"}, {"BRT_DRAIN_DONATE": "33", "BRT_FORCE_GLOBAL_SCENE": "1", "BRT_POOL_FORCE": "1"},
    # (two 512-thread workgroups per CU hold the cover scene twice since round 5 -- no room for a pool beside it: one per CU here)
    {"BRT_DRAIN_DONATE": "48", "BRT_WALK_EXIT": "0", "BRT_BLOCK_THREADS": "512", "BRT_WG_PER_CU": "1", "BRT_POOL_FORCE": "1"},
    {"BRT_POOL_FORCE": "1"}, {"BRT_POOL_FORCE": "1", "BRT_TUNABLE": "1"}, {"BRT_POOL_FORCE": "1", "BRT_POOL_CAP": "40"},
    {"BRT_POOL_FORCE": "1", "BRT_LEAF_VOTE": "64", "BRT_WALK_EXIT": "63"},
    # workgroup share of the pixel queue: one tile at a time, the maximum, with other workgroup shapes
    {"BRT_WGQ_BATCH": "64"}, {"BRT_WGQ_BATCH": "512"}, {"BRT_WGQ_BATCH": "192", "BRT_BLOCK_THREADS": "256"},
    {"BRT_WGQ_BATCH": "512", "BRT_FORCE_GLOBAL_SCENE": "1", "BRT_REFILL_MIN": "5"},
    # who takes pooled paths over: nearly full waves too, only thin ones (thinner than the donors: paths wait in the pool)
    {"BRT_POOL_ADOPT": "62", "BRT_POOL_FORCE": "1"}, {"BRT_POOL_ADOPT": "20", "BRT_DRAIN_DONATE": "40", "BRT_POOL_FORCE": "1"},
    {"BRT_POOL_ADOPT": "0", "BRT_POOL_FORCE": "1"},
    # the knobs-live instantiation with every knob at its default; top-of-tree LDS tile of 1 / 37 / 300 / all records
    {"BRT_TUNABLE": "1"}, {"BRT_FORCE_LDS_TOP": "1"}, {"BRT_FORCE_LDS_TOP": "37"},
    {"BRT_FORCE_LDS_TOP": "300", "BRT_TUNABLE": "1"}, {"BRT_FORCE_LDS_TOP": "100000"},
    {"BRT_FORCE_LDS_TOP": "64", "BRT_BLOCK_THREADS": "512", "BRT_WALK_EXIT": "5"},
])
def test_tuning_knobs_do_not_change_results(plugin, oracle, env):
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    lvl, cam, win = brt.cover_camera(120, 68, 3, 6, brt.Raytracing.Pure, 0.25)
    knobs = {k: int(v) for k, v in env.items()}
    if "BRT_CPU_BVH" in env:
        knobs["BRT_BVH_QUALITY"] = 0          # the PLOC builders (the callee's default tree is binned SAH)
    with plugin.tuning(**knobs):
        bb = brt.Buffers(b.models, b.materials, None) if "BRT_CPU_BVH" in env else b
        got = plugin.node.run(lvl, cam, win, 120, 68, buffers=bb, flags=brt.FLAG_COUNTERS)
        stats = dict(plugin.node.last_stats)
        # ... and without the counters: the COUNTERS instantiation runs the compiler-built walk loop and rejection sampler, the
        # shipped kernel the hand-written ones (walk_wave_lds_asm, ball_loop_asm: brt_device.h), whose leaf-vote rule differs
        got_plain = plugin.node.run(lvl, cam, win, 120, 68, flags=0)
        rays_plain = plugin.node.last_stats["rays"]
        if "BRT_POOL_FORCE" in env:        # the forced launches really carried a pool
            plugin.set_tuning("BRT_POOL_FORCE", 0)
            plugin.node.run(lvl, cam, win, 120, 68)
            assert plugin.node.last_stats["lds_bytes"] < stats["lds_bytes"]
    want, cnt = oracle.render(b, lvl, cam, win, 120, 68)
    assert_frames_equal(got, want)
    assert_frames_equal(got_plain, want)
    assert rays_plain == cnt["rays"]
    assert {k: stats[k] for k in COUNTER_KEYS} == cnt
    if "BRT_FORCE_GLOBAL_SCENE" in env:
        assert stats["scene_in_lds"] == 0
    if "BRT_FORCE_LDS_TOP" in env:
        assert stats["scene_in_lds"] == 2
    if "BRT_BLOCK_THREADS" in env:
        assert stats["threads_per_workgroup"] == int(env["BRT_BLOCK_THREADS"])


# ---- randomized scenes: materials, sizes, cameras, topologies -------------------------------------------------------

def _random_case(rng):
    n = int(rng.integers(1, 60))
    data = []
    for _ in range(n):
        kind = rng.random()
        mat = brt.StandardMaterial(base_color=tuple(float(x) for x in rng.random(3)),
                                   metallic=float(rng.choice([0.0, 1.0, rng.random()])),
                                   perceptual_roughness=float(rng.choice([0.0, 0.5, rng.random()])),
                                   ior=float(rng.uniform(0.5, 2.5)),
                                   specular_transmission=float(rng.choice([0.0, 1.0, rng.random()])))
        r = float(rng.uniform(0.05, 1.5)) if kind < 0.9 else float(rng.uniform(20, 200))
        pos = tuple(float(x) for x in rng.uniform(-4, 4, 3)) if kind < 0.9 else (float(rng.uniform(-3, 3)), -r - 1.0, float(rng.uniform(-3, 3)))
        data.append((pos, r, mat))
    topo = rng.integers(0, 4)
    bvh_fn = [None, single_leaf_bvh, lambda m: median_split_bvh(m, int(rng.integers(1, 5))), chain_bvh][topo]
    if bvh_fn is chain_bvh and n < 2:
        bvh_fn = single_leaf_bvh
    b = make_buffers(data, bvh_fn)
    w, h = int(rng.integers(1, 70)), int(rng.integers(1, 50))
    pos = tuple(float(x) for x in rng.uniform(-8, 8, 3))
    lvl, cam, win = uniforms(w, h, spp=int(rng.integers(1, 6)), bounces=int(rng.integers(0, 12)), pos=pos,
                             target=tuple(float(x) for x in rng.uniform(-1, 1, 3)), fov=float(rng.uniform(0.2, 1.5)),
                             seed=float(np.float32(rng.random())), level=brt.Raytracing(int(rng.integers(1, 4))),
                             window_height=int(rng.integers(1, 1200)))
    return b, lvl, cam, win, w, h


@pytest.mark.parametrize("lds_top", [None, "5"])
def test_randomized_scenes_bit_exact(oracle, lds_top):
    # lds_top: the same scenes through the SCENE_LDS_TOP kernel with a 5-record tile (most records then come from L2)
    with brt.RaytracePlugin([0]) as plugin:
        if lds_top:
            plugin.set_tuning("BRT_FORCE_LDS_TOP", int(lds_top))
        _randomized_scenes(plugin, oracle)


def _randomized_scenes(plugin, oracle):
    rng = np.random.default_rng(2024)
    for case in range(40):
        b, lvl, cam, win, w, h = _random_case(rng)
        raster = rng.random((h, w, 4), dtype=np.float32) if case % 3 == 0 else None
        depth = (rng.random((h, w), dtype=np.float32) * np.float32(0.05)) if case % 3 == 0 else None
        try:
            render_both(plugin, oracle, b, lvl, cam, win, w, h, raster=raster, depth=depth)
        except AssertionError as e:
            raise AssertionError(f"case {case}: {len(b.models)} spheres, {len(b.bvh)} nodes, {w}x{h}: {e}")


# ---- large frames and the asynchronous device entry point ------------------------------------------------------------

def test_4k_frame_rows_match_oracle(plugin, oracle):
    b = brt.generate_scene(brt.SCENE_RTIOW_FINAL, 1)
    w, h = 3840, 2160
    lvl, cam, win = brt.cover_camera(w, h, 1, 3)
    got = plugin.node.run(lvl, cam, win, w, h, buffers=b)
    assert plugin.node.last_stats["paths"] == w * h
    for r0 in (0, 1081, 2157):
        want, _ = oracle.render(b, lvl, cam, win, w, h, rows=(r0, r0 + 3))
        assert_frames_equal(got[r0:r0 + 3], want[r0:r0 + 3])


def test_async_render_on_a_caller_stream(plugin, oracle):
    import torch
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    w, h = 128, 72
    lvl, cam, win = brt.cover_camera(w, h, 2, 4)
    plugin.node.write_buffers(b)
    want, _ = oracle.render(b, lvl, cam, win, w, h)
    stream = torch.cuda.Stream()
    raster = torch.rand((h, w, 4), dtype=torch.float32, device="cuda")
    depth = torch.rand((h, w), dtype=torch.float32, device="cuda") * 0.02
    with torch.cuda.stream(stream):
        tile = torch.zeros((brt.tile_rows(h, 1), w, 4), dtype=torch.float32, device="cuda")
        st = plugin.node.render_part_device(lvl, cam, win, w, h, 0, 1, tile.data_ptr(), stream=stream.cuda_stream)
        assert st["kernel_ms"] == 0.0          # asynchronous: no timing, no counters yet
        frame = tile.clone()                   # ordered behind the kernel on the same stream
    stream.synchronize()
    assert_frames_equal(frame.cpu().numpy()[:h], want)
    # device-resident raster inputs (levels 1/2) through the same entry point
    lvl2, cam2, win2 = brt.cover_camera(w, h, 2, 4, brt.Raytracing.FallbackRaytraced)
    tile2 = torch.zeros_like(tile)
    plugin.node.render_part_device(lvl2, cam2, win2, w, h, 0, 1, tile2.data_ptr(), d_raster_rgba=raster.data_ptr(),
                                   d_raster_depth=depth.data_ptr())
    want2, _ = oracle.render(b, lvl2, cam2, win2, w, h, raster_rgba=raster.cpu().numpy(), raster_depth=depth.cpu().numpy())
    assert_frames_equal(tile2.cpu().numpy()[:h], want2)


def test_expensive_first_dispatch_never_changes_pixels(plugin, oracle):
    # the tile order of a frame comes from the previous frame's per-tile ray counts (brt_api.cpp
    # attach_tile_order): first frame raster order, later frames expensive tiles first
    b = brt.generate_scene(brt.SCENE_COVER, 4)
    w, h = 200, 120
    lvl, cam, win = brt.cover_camera(w, h, 3, 6, brt.Raytracing.Pure, 0.61)
    want, cnt = oracle.render(b, lvl, cam, win, w, h)
    plugin.node.write_buffers(b)                       # new scene epoch: no history
    for frame_no in range(4):
        got = plugin.node.run(lvl, cam, win, w, h, flags=brt.FLAG_COUNTERS)
        assert_frames_equal(got, want)
        assert {k: plugin.node.last_stats[k] for k in COUNTER_KEYS} == cnt, frame_no
    with plugin.tuning(BRT_LPT=0):
        assert_frames_equal(plugin.node.run(lvl, cam, win, w, h), want)
    # the ingredients of the order one by one: ranking key, sky tiles first, critical pixels (waves at raised
    # priority that stop taking pixels -- on a frame this small nearly every ranked tile is critical)
    for env in ({"BRT_LPT_SORT": "0"}, {"BRT_LPT_LANE_PERMILLE": "100"}, {"BRT_LPT_LANE_PERMILLE": "1000"},
                {"BRT_LPT_LANE_PERMILLE": "500", "BRT_DRAIN_DONATE": "0"}, {"BRT_CRIT": "0"}, {"BRT_DRAIN_DONATE": "0"},
                {"BRT_LPT_LANE_PERMILLE": "1000", "BRT_DRAIN_DONATE": "56"}):
        with plugin.tuning(**{k: int(v) for k, v in env.items()}):                # (setting a knob forgets the history ...)
            for frame_no in range(3):                                              # ... measure, then use the order
                got = plugin.node.run(lvl, cam, win, w, h, flags=brt.FLAG_COUNTERS)
                assert_frames_equal(got, want)
                assert {k: plugin.node.last_stats[k] for k in COUNTER_KEYS} == cnt, (env, frame_no)
    # another view of the same scene reuses nothing wrongly (different size -> history key mismatch)
    lvl2, cam2, win2 = brt.cover_camera(96, 54, 2, 4)
    want2, _ = oracle.render(b, lvl2, cam2, win2, 96, 54)
    assert_frames_equal(plugin.node.run(lvl2, cam2, win2, 96, 54), want2)
    assert_frames_equal(plugin.node.run(lvl, cam, win, w, h), want)


def test_pinned_frame_fast_path(plugin, oracle):
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    w, h = 136, 77
    lvl, cam, win = brt.cover_camera(w, h, 2, 4)
    want, _ = oracle.render(b, lvl, cam, win, w, h)
    frame = plugin.alloc_frame(w, h)
    frame[:] = -1.0
    got = plugin.node.run(lvl, cam, win, w, h, buffers=b, out=frame)
    assert got is frame
    assert_frames_equal(frame, want)
    with brt.RaytracePlugin([0, 0]) as p2:          # strips of two sub-contexts DMA'd into one pinned frame
        f2 = p2.alloc_frame(w, h)
        p2.node.run(lvl, cam, win, w, h, buffers=b, out=f2)
        assert_frames_equal(f2, want)


def test_unchanged_scene_is_not_reuploaded_and_changes_are(plugin, oracle):
    import time
    b = brt.generate_scene(brt.SCENE_STRESS_GRID, 1)
    nb = brt.Buffers(b.models, b.materials, None)             # callee builds the BVH: the expensive upload
    plugin.node.write_buffers(brt.generate_scene(brt.SCENE_COVER, 9))
    t0 = time.perf_counter(); plugin.node.write_buffers(nb); first = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(10):
        plugin.node.write_buffers(nb)
    again = (time.perf_counter() - t0) / 10
    assert again < first / 3, (first, again)                  # byte compare only
    lvl, cam, win = brt.cover_camera(64, 36, 2, 4)
    f1 = plugin.node.run(lvl, cam, win, 64, 36)
    moved = b.models.copy(); moved[-1]["position"] = (4.0, 1.5, 0.0)   # one sphere moves: must be noticed
    f2 = plugin.node.run(lvl, cam, win, 64, 36, buffers=brt.Buffers(moved, b.materials, None))
    want2, _ = oracle.render(brt.Buffers(moved, b.materials, brt.build_bvh(moved)), lvl, cam, win, 64, 36)
    assert_frames_equal(f2, want2)
    assert not np.array_equal(f1, f2)
    # a failed upload invalidates the scene even if the next upload repeats earlier bytes
    bad = brt.build_bvh(moved); bad[0]["index"] = len(bad)
    with pytest.raises(brt.BrtError):
        plugin.node.write_buffers(brt.Buffers(moved, b.materials, bad))
    f3 = plugin.node.run(lvl, cam, win, 64, 36, buffers=brt.Buffers(moved, b.materials, None))
    assert_frames_equal(f3, want2)


# ---- the frame loop of bench.py: render on the context's stream, gather + de-interleave on torch's ------------------

def test_frame_loop_with_a_new_seed_every_frame(plugin, oracle):
    """world = 1 leg of bevyray_amd.parallel.gather_frame: the de-interleave kernel is enqueued on torch's current
    stream (the default stream, handle 0, through BRT_FLAG_CALLER_STREAM) behind whatever produced the tiles; the
    seed changes every frame so that a stale tile or frame would show."""
    import torch
    from bevyray_amd.parallel import end_of_frame, gather_frame
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    w, h = 200, 120
    plugin.node.write_buffers(b)
    tile = torch.zeros((brt.tile_rows(h, 1), w, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for i in range(6):
        lvl, cam, win = brt.cover_camera(w, h, 3, 5, seed=0.1 + 0.13 * i)
        plugin.node.render_part_device(lvl, cam, win, w, h, 0, 1, tile.data_ptr())
        frame = gather_frame(tile, h, 0, 1, node=plugin.node)
        end_of_frame(tile)
        want, _ = oracle.render(b, lvl, cam, win, w, h)
        assert_frames_equal(frame.cpu().numpy(), want)


def _nccl_rank(rank, world, port, w, h, seeds, out_path):
    import torch
    import torch.distributed as dist
    from bevyray_amd.parallel import RcclGather, end_of_frame, gather_frame
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        b = brt.generate_scene(brt.SCENE_COVER, 1)
        frames = []
        with brt.RaytracePlugin([rank]) as p:
            p.node.write_buffers(b)
            rccl = RcclGather.create(p, rank, world)       # the collective behind the C ABI (brt_gather_rccl)
            assert rccl is not None
            tile = torch.zeros((brt.tile_rows(h, world), w, 4), dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            for k, seed in enumerate(seeds):
                lvl, cam, win = brt.cover_camera(w, h, 3, 5, seed=seed)
                p.node.render_part_device(lvl, cam, win, w, h, rank, world, tile.data_ptr())
                # odd frames through torch.distributed's gather: both legs must assemble the same frame
                frame = gather_frame(tile, h, rank, world, node=p.node, rccl=rccl if k % 2 == 0 else None)
                end_of_frame(tile)
                if rank == 0:
                    frames.append(frame.cpu().numpy())
            rccl.close()
            if rank == 0:
                np.save(out_path, np.stack(frames))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_ranks_over_rccl_match_one_gpu(plugin, oracle, tmp_path):
    """RCCL leg (needs two GPUs; skipped on the one-GPU boxes): two ranks, a new seed every frame, rank 0's
    gathered frame against the oracle and against the one-GPU render."""
    import socket
    import torch
    import torch.multiprocessing as mp
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    w, h, seeds = 200, 120, [0.1, 0.3, 0.55, 0.8]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "frames.npy")
    mp.spawn(_nccl_rank, args=(2, port, w, h, seeds, out), nprocs=2, join=True)
    frames = np.load(out)
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    for seed, f in zip(seeds, frames):
        lvl, cam, win = brt.cover_camera(w, h, 3, 5, seed=seed)
        want, _ = oracle.render(b, lvl, cam, win, w, h)
        assert_frames_equal(f, want)
        assert_frames_equal(plugin.node.run(lvl, cam, win, w, h, buffers=b), want)


def test_rccl_gather_behind_the_c_abi_on_one_rank(plugin, oracle):
    """SURVEY 8(e) through the product boundary: brt_rccl_unique_id / brt_rccl_comm_create / brt_gather_rccl (librccl resolved
    by dlopen inside libbevyray_amd.so) with a ONE-rank communicator on the one GPU of the box -- the same calls every rank
    makes on an N-GPU node: the tile goes through ncclGather (rccl.h:745) into the receive buffer and through the
    de-interleave kernel into the frame, on the context's own stream and on torch's current stream, a new seed every frame."""
    import torch
    from bevyray_amd.parallel import RcclGather, end_of_frame, gather_frame
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    w, h = 200, 120
    plugin.node.write_buffers(b)
    rccl = RcclGather.create(plugin, 0, 1)
    assert rccl is not None, "librccl did not resolve inside the library"
    tile = torch.zeros((brt.tile_rows(h, 1), w, 4), dtype=torch.float32, device="cuda")
    tiles = torch.full((1,) + tuple(tile.shape), -1.0, dtype=torch.float32, device="cuda")
    frame = torch.full((h, w, 4), -1.0, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for k, seed in enumerate((0.1, 0.35, 0.6, 0.85)):
        lvl, cam, win = brt.cover_camera(w, h, 3, 5, seed=seed)
        want, _ = oracle.render(b, lvl, cam, win, w, h)
        plugin.node.render_part_device(lvl, cam, win, w, h, 0, 1, tile.data_ptr())
        if k % 2 == 0:      # the context's own stream: synchronous
            plugin.node.gather_rccl(rccl.comm, 0, 1, tile.data_ptr(), tiles.data_ptr(), w, h, frame.data_ptr())
            got = frame.cpu().numpy()
            assert_frames_equal(tiles[0, :h].cpu().numpy(), want)        # what ncclGather delivered
        else:               # torch's current stream, as bench.py does
            got = gather_frame(tile, h, 0, 1, node=plugin.node, rccl=rccl)
            end_of_frame(tile)
            got = got.cpu().numpy()
        assert_frames_equal(got, want)
    # argument errors come back as codes, not crashes
    with pytest.raises(brt.BrtError):
        plugin.node.gather_rccl(0, 0, 1, tile.data_ptr(), tiles.data_ptr(), w, h, frame.data_ptr())
    with pytest.raises(brt.BrtError):
        plugin.node.gather_rccl(rccl.comm, 1, 1, tile.data_ptr(), tiles.data_ptr(), w, h, frame.data_ptr())
    rccl.close()


def test_frame_target_in_imported_external_memory(plugin, oracle):
    """SURVEY 8(f3), pipeline.rs:191-203 (the pass writes straight into post_process.destination): device memory allocated and
    exported as a file descriptor OUTSIDE the render path (hipMemCreate + hipMemExportToShareableHandle, the stand-in for the
    host's Vulkan allocation), imported through brt_import_frame_fd, rendered into with brt_render_device, read back through
    the EXPORTER's own mapping of the same memory: the frame is the oracle's, i.e. the kernel wrote where the other API reads."""
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    w, h = 160, 90
    nbytes = w * h * 16
    lvl, cam, win = brt.cover_camera(w, h, 3, 5, seed=0.3)
    want, _ = oracle.render(b, lvl, cam, win, w, h)
    plugin.node.write_buffers(b)
    fd, d_exported = plugin.debug_export_frame_fd(nbytes)
    try:
        done = []
        for handle_type in (brt.EXTMEM_DMABUF_FD, brt.EXTMEM_OPAQUE_FD):
            try:
                d_frame = plugin.import_frame_fd(os.dup(fd) if handle_type == brt.EXTMEM_OPAQUE_FD else fd, nbytes, handle_type)
            except brt.BrtError as e:
                if handle_type == brt.EXTMEM_OPAQUE_FD:     # a dma-buf of a HIP allocation is not what every driver accepts as "opaque fd"
                    print(f"opaque-fd import of a HIP dma-buf refused by this runtime: {e.text}")
                    continue
                raise
            assert d_frame != d_exported                    # a second mapping of the same memory
            for seed in (0.3, 0.7):
                lvl, cam, win = brt.cover_camera(w, h, 3, 5, seed=seed)
                want, _ = oracle.render(b, lvl, cam, win, w, h)
                plugin.node.render_device(lvl, cam, win, w, h, d_frame)
                assert_frames_equal(plugin.debug_copy_to_host(d_exported, (h, w, 4)), want)
            plugin.release_frame(d_frame)
            done.append(handle_type)
        assert brt.EXTMEM_DMABUF_FD in done
        with pytest.raises(brt.BrtError):
            plugin.release_frame(12345)
    finally:
        os.close(fd)
        plugin.release_frame(d_exported)


def test_moving_camera_measures_every_frame_in_the_lean_kernel(oracle):
    """The reference's demo flies the camera (src/main.rs:40) and re-extracts it every frame (extract.rs:118-157).  A view whose
    camera moves measures its tile costs EVERY frame, in the LEAN instantiation, and runs each frame in the order measured
    on the frame before; a view that stands still stops measuring.  Every frame of a 14-step orbit bit-exact against the
    oracle; which instantiation ran and whether it measured comes back in the stats."""
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    w, h, spp, bounces = 256, 144, 32, 4
    cams = []
    for k in range(14):
        a = np.deg2rad(0.5 * k)
        pos = (13.0 * np.cos(a) - 3.0 * np.sin(a), 2.0, 13.0 * np.sin(a) + 3.0 * np.cos(a))
        cams.append(uniforms(w, h, spp=spp, bounces=bounces, pos=tuple(float(x) for x in pos), target=(0, 0, 0), fov=0.4, seed=0.1 + 0.05 * k))
    with brt.RaytracePlugin([0]) as p:
        p.node.write_buffers(b)
        seen = []
        for k, (lvl, cam, win) in enumerate(cams):
            got = p.node.run(lvl, cam, win, w, h)
            st = dict(p.node.last_stats)
            want, cnt = oracle.render(b, lvl, cam, win, w, h)
            assert_frames_equal(got, want)
            assert st["rays"] == cnt["rays"]
            seen.append((st["kernel_variant"], st["measured_tile_costs"], st["prepass_ms"] > 0))
        assert seen[0][2] and not any(s[2] for s in seen[1:])            # one pre-pass, for the first frame of the view
        assert all(s[1] == 1 for s in seen)                                # the camera moves: every frame measures ...
        assert all(s[0] in (1, 2) for s in seen[2:])                       # ... in a LEAN instantiation
        lvl, cam, win = cams[-1]
        still = []
        for k in range(4):                                                 # the view stands still: nothing new to measure
            p.node.run(lvl, cam, win, w, h)
            still.append(p.node.last_stats["measured_tile_costs"])
        assert still == [0, 0, 0, 0]
        # a scene upload (one sphere moves) asks for a measurement within four frames, once
        moved = b.models.copy(); moved[5]["position"][0] += np.float32(0.01)
        p.node.write_buffers(brt.Buffers(moved, b.materials, b.bvh))
        after = []
        for k in range(8):
            p.node.run(lvl, cam, win, w, h)
            after.append(p.node.last_stats["measured_tile_costs"])
        assert sum(after) == 1 and after.index(1) < 4
        with p.tuning(BRT_LEAN_MEASURE=0):                                 # the round-3 behaviour: measuring frames in the general instantiation
            lvl, cam, win = cams[3]
            got = p.node.run(lvl, cam, win, w, h)
            assert p.node.last_stats["measured_tile_costs"] == 1 and p.node.last_stats["kernel_variant"] == 0
            assert_frames_equal(got, oracle.render(brt.Buffers(moved, b.materials, b.bvh), lvl, cam, win, w, h)[0])


def test_short_circuit_policy_switch_matches_the_oracle_under_that_policy(oracle, monkeypatch):
    """The alternative reading of `||` (raytrace.wgsl:269) exists in the kernel too, chosen through the API
    (brt_set_policy(BRT_POLICY_OR_SHORT_CIRCUIT)): on the ior < 1 fixture it must give the oracle's frame under that
    policy, which differs from the default one.  The environment cannot switch it: BRT_POLICY_OR_SHORT_CIRCUIT set
    before brt_create (with or without BRT_ENABLE_TUNING) or after it does not change a pixel."""
    monkeypatch.setenv("BRT_POLICY_OR_SHORT_CIRCUIT", "1")
    monkeypatch.setenv("BRT_ENABLE_TUNING", "1")
    with brt.RaytracePlugin([0]) as plugin:
        _short_circuit_policy_case(plugin, oracle, monkeypatch)


def _short_circuit_policy_case(plugin, oracle, monkeypatch):
    z = np.load(os.path.join(GOLDEN, "policy_frames.npz"))
    g = lambda k: z[f"glass_tir.{k}"]
    b = brt.Buffers(g("models").view(brt.MODEL_DTYPE), g("materials").view(brt.MATERIAL_DTYPE), g("bvh").view(brt.BVH_NODE_DTYPE))
    lvl, cam, win = g("level").view(brt.LEVEL_DTYPE), g("camera").view(brt.CAMERA_DTYPE), g("window").view(brt.WINDOW_DTYPE)
    w, h = (int(x) for x in g("size"))
    got = plugin.node.run(lvl, cam, win, w, h, buffers=b)
    assert_frames_equal(got, g("frame.default"))          # the environment variable (set before brt_create) is not the switch
    monkeypatch.setenv("BRT_POLICY_OR_SHORT_CIRCUIT", "0")
    plugin.set_policy(brt.POLICY_OR_SHORT_CIRCUIT)
    got = plugin.node.run(lvl, cam, win, w, h, buffers=b, flags=brt.FLAG_COUNTERS)
    assert_frames_equal(got, g("frame.or_short_circuit"))
    assert plugin.node.last_stats["rays"] == int(g("rays.or_short_circuit")[0])
    with oracle.policy(or_short_circuit=True):
        want, cnt = oracle.render(b, lvl, cam, win, w, h)
    assert_frames_equal(got, want)
    assert {k: plugin.node.last_stats[k] for k in COUNTER_KEYS} == cnt
    assert not np.array_equal(g("frame.default").view(np.uint32), g("frame.or_short_circuit").view(np.uint32))
    plugin.set_policy(0)
    assert_frames_equal(plugin.node.run(lvl, cam, win, w, h, buffers=b), g("frame.default"))
    with pytest.raises(brt.BrtError):
        plugin.set_policy(8)


def test_every_policy_switch_matches_the_fixtures_and_the_oracle_under_that_policy(oracle):
    """VERDICT r4, What's missing #4: the oracle has three switches for the readings WGSL leaves open (`||`, min / max with a NaN, pow);
    the product had one.  brt_set_policy now takes all three (and their combinations): on every policy fixture
    (tests/golden/policy_frames.npz: frames of the independent numpy restatement under each policy -- glass with ior < 1, NaN bounds,
    grazing glass) the GPU frame equals the committed frame and the oracle's frame + counters under the same policy, through the
    LDS-resident walk, the tile + global walk and the all-global walk.  The default stays what the parity suite is stated on."""
    z = np.load(os.path.join(GOLDEN, "policy_frames.npz"))
    names = sorted({k.split(".")[0] for k in z.files})
    pols = {"default": (0, {}), "or_short_circuit": (brt.POLICY_OR_SHORT_CIRCUIT, dict(or_short_circuit=True)),
            "minmax_select": (brt.POLICY_MINMAX_SELECT, dict(minmax="select")), "pow_exp2log2": (brt.POLICY_POW_EXP2_LOG2, dict(pow="exp2log2")),
            "all_three": (7, dict(or_short_circuit=True, minmax="select", pow="exp2log2"))}
    moved = {k: 0 for k in pols}
    with brt.RaytracePlugin([0]) as p:
        for knobs in ({}, {"BRT_FORCE_LDS_TOP": 3}, {"BRT_FORCE_GLOBAL_SCENE": 1}):
            for k in ("BRT_FORCE_LDS_TOP", "BRT_FORCE_GLOBAL_SCENE"):
                p.set_tuning(k, knobs.get(k, 0))
            for name in names:
                g = lambda k: z[f"{name}.{k}"]
                b = brt.Buffers(g("models").view(brt.MODEL_DTYPE), g("materials").view(brt.MATERIAL_DTYPE), g("bvh").view(brt.BVH_NODE_DTYPE))
                lvl, cam, win = g("level").view(brt.LEVEL_DTYPE), g("camera").view(brt.CAMERA_DTYPE), g("window").view(brt.WINDOW_DTYPE)
                w, h = (int(x) for x in g("size"))
                for pname, (flags, okw) in pols.items():
                    p.set_policy(flags)
                    for fl in (brt.FLAG_COUNTERS, 0):
                        got = p.node.run(lvl, cam, win, w, h, buffers=b, flags=fl)
                        st = dict(p.node.last_stats)
                        assert_frames_equal(got, g(f"frame.{pname}"))
                        assert st["rays"] == int(g(f"rays.{pname}")[0]), (name, pname)
                        with oracle.policy(**okw):
                            want, cnt = oracle.render(b, lvl, cam, win, w, h)
                        assert_frames_equal(got, want)
                        if fl:
                            assert {q: st[q] for q in COUNTER_KEYS} == cnt, (name, pname, knobs)
                    same = (g(f"frame.{pname}").view(np.uint32) == g("frame.default").view(np.uint32)) | (np.isnan(g(f"frame.{pname}")) & np.isnan(g("frame.default")))
                    moved[pname] += int((~same).sum())
        p.set_policy(0)
        # the bring-up kernel implements the default reading only and says so
        p.set_policy(brt.POLICY_MINMAX_SELECT)
        with pytest.raises(brt.BrtError):
            p.node.run(lvl, cam, win, w, h, flags=brt.FLAG_KERNEL_SIMPLE)
        p.set_policy(0)
    # the `||` and the min / max switches do change pixels of their fixtures (pow only moves the last bits of a reflectance, which
    # rarely flips a reflect / refract decision at this size)
    assert moved["default"] == 0 and moved["or_short_circuit"] > 0 and moved["minmax_select"] > 0 and moved["all_three"] > 0, moved


def test_first_frame_prepass_orders_tiles_without_changing_pixels(oracle, monkeypatch):
    """The first frame of a view runs a dispatch-order pre-pass of min(4, samples / 16) samples per pixel into the frame's own tile buffer and then
    the frame itself: same pixels and counters as the oracle, prepass_ms reported; BRT_PREPASS_SPP=0 turns it
    off; the second frame of the view needs none."""
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    w, h = 320, 180
    lvl, cam, win = brt.cover_camera(w, h, 32, 8)
    want, cnt = oracle.render(b, lvl, cam, win, w, h)
    with brt.RaytracePlugin([0]) as p:
        f1 = p.node.run(lvl, cam, win, w, h, buffers=b, flags=brt.FLAG_COUNTERS)
        s1 = dict(p.node.last_stats)
        f2 = p.node.run(lvl, cam, win, w, h, flags=brt.FLAG_COUNTERS)
        s2 = dict(p.node.last_stats)
    assert_frames_equal(f1, want)
    assert_frames_equal(f2, want)
    assert {k: s1[k] for k in COUNTER_KEYS} == cnt and {k: s2[k] for k in COUNTER_KEYS} == cnt
    assert s1["prepass_ms"] > 0.0 and s2["prepass_ms"] == 0.0
    with brt.RaytracePlugin([0]) as p:
        p.set_tuning("BRT_PREPASS_SPP", 0)
        f3 = p.node.run(lvl, cam, win, w, h, buffers=b)
        assert p.node.last_stats["prepass_ms"] == 0.0
    assert_frames_equal(f3, want)
    # the environment is read once, at brt_create, and only under BRT_ENABLE_TUNING=1
    monkeypatch.setenv("BRT_PREPASS_SPP", "0")
    with brt.RaytracePlugin([0]) as p:
        p.node.run(lvl, cam, win, w, h, buffers=b)
        assert p.node.last_stats["prepass_ms"] > 0.0
    monkeypatch.setenv("BRT_ENABLE_TUNING", "1")
    with brt.RaytracePlugin([0]) as p:
        assert p.get_tuning("BRT_PREPASS_SPP") == (0, 4)
        monkeypatch.setenv("BRT_PREPASS_SPP", "2")          # after brt_create: ignored
        p.node.run(lvl, cam, win, w, h, buffers=b)
        assert p.node.last_stats["prepass_ms"] == 0.0
    monkeypatch.delenv("BRT_ENABLE_TUNING")
    monkeypatch.delenv("BRT_PREPASS_SPP")
    # frames of fewer than 32 samples (the rule would leave a pre-pass of 0 or 1) run without one
    for spp in (8, 24):
        lvl, cam, win = brt.cover_camera(w, h, spp, 8)
        with brt.RaytracePlugin([0]) as p:
            p.node.run(lvl, cam, win, w, h, buffers=b)
            assert p.node.last_stats["prepass_ms"] == 0.0


# ---- the shared-reciprocal division (brt_device.h) against the compiler's correctly rounded `/` ---------------------

DBG_DIV, DBG_DIV_SWEEP, DBG_SQRT_SWEEP = 6, 7, 8


def test_short_sqrt_is_the_compilers_sqrt_on_every_float_of_its_range(plugin):
    """sqrt_plain (hipcc's correctly rounded sqrt without its tiny-argument scaling and class test) against
    __builtin_sqrtf on EVERY float in [2^-80, 2^80]: 1.34e9 values, bit for bit -- and on +0 and -0, which the sample colour's
    sqrt3 also sends through it."""
    lo = int(np.float32(2.0 ** -80).view(np.uint32))
    hi = int(np.float32(2.0 ** 80).view(np.uint32))
    per = 65536
    n = (hi - lo) // per + 2
    inp = np.zeros((n, 16), np.float32)
    inp[:, 0] = np.full(n, lo - 3, np.uint32).view(np.float32)     # element i covers [lo - 3 + i * per, +per)
    inp[:, 1] = float(per)
    out = plugin.debug_eval(DBG_SQRT_SWEEP, inp)
    assert out[:, 0].sum() == 0, f"first mismatching argument bits {out[out[:, 0] > 0][:1, 1].view(np.uint32)}"


def test_shared_reciprocal_division_is_the_compilers_division_in_the_plain_range(plugin):
    """div_plain / recip_plain are hipcc's own expansion of `/` without the operand scaling and the special-case
    fixup, valid for operands in [2^-40, 2^40]: bit-identical there -- 2.1e9 pseudo-random pairs over every
    exponent combination on the device, and every boundary value from the host -- and the callers' range guards
    keep everything else (zeros, denormals, huge, NaN) on `/`."""
    seeds = np.arange(1, 32769, dtype=np.uint32)
    inp = np.zeros((len(seeds), 16), np.float32)
    inp[:, 0] = seeds.view(np.float32)
    inp[:, 1] = 65536.0
    out = plugin.debug_eval(DBG_DIV_SWEEP, inp)
    bad = out[:, 0].sum()
    first = out[out[:, 0] > 0][:1]
    assert bad == 0, f"{bad} mismatches, first n/d bits {first[:, 1:3].view(np.uint32)}"
    # boundaries of the plain range and a dense mantissa sweep at fixed exponents, checked on the host
    rng = np.random.default_rng(3)
    edge = np.array([2.0 ** -40, np.nextafter(np.float32(2.0 ** -40), np.float32(1)), 1.0, np.nextafter(np.float32(1), np.float32(2)),
                     np.nextafter(np.float32(1), np.float32(0)), 3.0, 2.0 ** 40, np.nextafter(np.float32(2.0 ** 40), np.float32(0)),
                     1.9999999, 0.33333334, 7.0, 1e-9, 1e9], np.float32)
    edge = np.concatenate([edge, -edge])
    n, d = np.meshgrid(edge, edge)
    n = np.concatenate([n.ravel(), (rng.random(400000, dtype=np.float32) + 1) * np.float32(2.0) ** rng.integers(-40, 40, 400000).astype(np.float32)])
    d = np.concatenate([d.ravel(), (rng.random(400000, dtype=np.float32) + 1) * np.float32(2.0) ** rng.integers(-40, 40, 400000).astype(np.float32)])
    out = plugin.debug_eval(DBG_DIV, _inputs([n, d]))
    assert_frames_equal(out[:, 0], n / d)                 # the compiler's `/` is IEEE division (numpy on x86 is too)
    assert_frames_equal(out[:, 1], n / d)
    assert_frames_equal(out[:, 2], np.float32(1.0) / d)
    assert_frames_equal(out[:, 3], np.float32(1.0) / d)


# ---- the dispatch order built on the GPU (brt_order.hip) against the host statement of the rule (brt_host.cpp) ----------

def test_gpu_tile_order_equals_host_tile_order(plugin):
    import ctypes as C
    from bevyray_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(17)
    for n, spp, lanes in [(1, 64, 262144), (200, 64, 262144), (32400, 64, 262144), (129600, 1024, 262144), (5000, 8, 1000),
                          (777, 64, 40), (4096, 256, 262144)]:
        for flavour in range(3):
            longest = rng.integers(spp, spp * 9 + 1, n).astype(np.uint32)
            ray_sum = (64 * spp + 200 + longest.astype(np.uint64) * 30).astype(np.uint32)
            sky = rng.random(n) < (0.0, 0.3, 1.0)[flavour]
            longest[sky], ray_sum[sky] = spp, 64 * spp
            if flavour == 1 and n > 10:
                k = len(longest[1::5])
                longest[0:5 * k:5] = longest[1::5]           # ties: raster order among equals
            # ranked by itself, and by the neighbourhood of radius 2 / 5 in a tile grid of some width that divides n (brt_render's default / a moved camera)
            widths = [t for t in (240, 120, 37, 16, 8, 5, 1) if n % t == 0]
            # ... and with the last `split` non-sky tiles handed out as two half-sample jobs (none, fewer than there are, more than there are)
            for tiles_x, dilate, split in [(0, 0, 0), (0, 0, max(1, n // 7)), (0, 0, n + 3)] + [(widths[0], d, s) for d in (2, 5) for s in (0, n // 3)]:
                want = np.zeros(n + split, np.uint32); info5 = np.zeros(5, np.uint32)
                _lib.check(lib.brt_host_tile_order(ray_sum.ctypes.data, longest.ctypes.data, n, spp, lanes, 1, 0, tiles_x, dilate, split,
                                                   want.ctypes.data, info5.ctypes.data))
                got = np.zeros(n + split, np.uint32); info4 = np.zeros(4, np.uint32)
                _lib.check(lib.brt_debug_tile_order(plugin._ctx, ray_sum.ctypes.data, longest.ctypes.data, n, spp, lanes, tiles_x, dilate, split,
                                                    got.ctypes.data, info4.ctypes.data), plugin._ctx)
                assert info4.tolist() == info5[1:].tolist(), (n, spp, flavour, tiles_x, dilate, split, info4, info5)
                n_split = int(info5[4])
                assert n_split == min(split, int(info5[3]) - int(info5[1]))      # (non-sky tiles that are not critical)
                assert np.array_equal(got[:n + n_split], want[:n + n_split]), (n, spp, flavour, tiles_x, dilate, split)


def test_half_sample_jobs_hand_the_pixel_state_over_without_changing_pixels_or_ray_counts(oracle):
    """The last tiles of the dispatch order are handed out as two half-sample jobs (brt_host.cpp build_tile_order): the first leaves
    {rng, sums, rays} per pixel in HBM, the second picks them up -- or, when the record is not there yet, leaves the pixel to the
    first-half lane, which then renders the second half too (one atomic exchange per side decides; nobody waits, nothing is rendered
    twice).  Forced on small frames (BRT_SPLIT_FORCE tiles; by default only frames of >= 6 tiles per wave slot split): few tiles -> the
    second job runs beside the first and leaves, many -> it finds the records.
    Pixels and the frame's ray count equal the oracle's either way, with the order built on the GPU and on the host."""
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    seen = {"taken": 0, "left": 0}
    for (w, h, spp, bounces) in ((320, 180, 32, 8), (96, 40, 64, 4), (640, 360, 16, 8), (200, 120, 33, 3)):
        lvl, cam, win = brt.cover_camera(w, h, spp, bounces)
        want, cnt = oracle.render(b, lvl, cam, win, w, h)
        n_tiles = ((w + 7) // 8) * ((h + 7) // 8)
        for force, host in ((n_tiles, 0), (3, 0), (max(1, n_tiles // 2), 1), (n_tiles, 1)):
            with brt.RaytracePlugin([0]) as p:
                p.set_tuning("BRT_SPLIT_FORCE", force)
                p.set_tuning("BRT_ORDER_ON_HOST", host)
                for frame in range(3):          # pre-pass + frame in its order, then frames in the measured order
                    got = p.node.run(lvl, cam, win, w, h, buffers=b if frame == 0 else None)
                    assert_frames_equal(got, want)
                    assert p.node.last_stats["rays"] == cnt["rays"], (w, h, spp, force, host, frame)
                p.debug_profile()
                meta = p.last_order_meta
                assert 0 < meta["split_tiles"] <= force, (meta, force)
                assert meta["second_halves_taken"] + meta["second_halves_left"] > 0, meta
                seen["taken"] += meta["second_halves_taken"]
                seen["left"] += meta["second_halves_left"]
                # the counting instantiation takes a first half as the whole tile and skips the second: same frame, all five counters
                got = p.node.run(lvl, cam, win, w, h, flags=brt.FLAG_COUNTERS)
                assert_frames_equal(got, want)
                assert {k: p.node.last_stats[k] for k in COUNTER_KEYS} == cnt
    assert seen["left"] > 0, seen
    # the depth blend levels (the general instantiation: the depth sum is part of the state that changes hands), with raster inputs
    w, h, spp = 200, 120, 32
    rng = np.random.default_rng(11)
    raster = rng.random((h, w, 4), dtype=np.float32)
    depth = rng.random((h, w), dtype=np.float32) * np.float32(0.02)
    for level in (brt.Raytracing.FallbackRaster, brt.Raytracing.FallbackRaytraced):
        lvl, cam, win = brt.cover_camera(w, h, spp, 4, level, 0.5)
        want, cnt = oracle.render(b, lvl, cam, win, w, h, raster_rgba=raster, raster_depth=depth)
        with brt.RaytracePlugin([0]) as p:
            p.set_tuning("BRT_SPLIT_FORCE", 150)
            for frame in range(3):
                got = p.node.run(lvl, cam, win, w, h, buffers=b if frame == 0 else None, raster_rgba=raster, raster_depth=depth)
                assert_frames_equal(got, want)
                assert p.node.last_stats["rays"] == cnt["rays"]
            p.debug_profile()
            assert p.last_order_meta["split_tiles"] > 0
    # a frame of many tiles per wave slot, default settings: the second halves come up long after the first ones and take the states over
    w, h, spp, bounces = 1920, 1080, 16, 2
    lvl, cam, win = brt.cover_camera(w, h, spp, bounces)
    want, cnt = oracle.render(b, lvl, cam, win, w, h)
    with brt.RaytracePlugin([0]) as p:
        for frame in range(3):
            got = p.node.run(lvl, cam, win, w, h, buffers=b if frame == 0 else None)
            assert_frames_equal(got, want)
            assert p.node.last_stats["rays"] == cnt["rays"]
        p.debug_profile()
        meta = p.last_order_meta
        assert meta["split_tiles"] > 0 and meta["second_halves_taken"] > 50 * meta["second_halves_left"], meta


def test_store_conversions_on_single_values(plugin, oracle):
    """BRT_DBG_ENCODE: the three store conversions of BRT_FLAG_OUT_* on the device, value by value, against the oracle's restatement:
    every sRGB threshold and its predecessor, the unorm ties (k + 0.5) / 255 neighbourhoods, f16 ties / overflow / denormals, specials."""
    t = brt.srgb_thresholds()
    rng = np.random.default_rng(4)
    ties = ((np.arange(0, 255) + 0.5) / 255.0).astype(np.float32)
    x = np.concatenate([t, np.nextafter(t, np.float32(-1)), ties, np.nextafter(ties, np.float32(2)), np.nextafter(ties, np.float32(-1)),
                        rng.random(200000).astype(np.float32), (np.float32(10.0) ** rng.uniform(-12, 6, 100000)).astype(np.float32) * rng.choice([-1, 1], 100000).astype(np.float32),
                        np.array([0.0, -0.0, 1.0, np.inf, -np.inf, np.nan, 65504.0, 65519.9, 65520.0, 2.9802322e-8, 2.9802326e-8, 5.96e-8, 6.1e-5, 6.0975e-5], np.float32)])
    out = plugin.debug_eval(9, _inputs([x]))
    f = np.stack([x, x, x, x], axis=-1).reshape(-1, 1, 4)          # (the value in every channel; channel 0 is a colour channel)
    assert np.array_equal(out[:, 0].astype(np.int64), oracle.encode_frame(f, "srgb8")[:, 0, 0].astype(np.int64))
    assert np.array_equal(out[:, 1].astype(np.int64), oracle.encode_frame(f, "unorm8")[:, 0, 0].astype(np.int64))
    assert np.array_equal(oracle.encode_frame(f, "srgb8")[:, 0, 3], oracle.encode_frame(f, "unorm8")[:, 0, 3])      # alpha is linear in both
    got16, want16 = out[:, 2].astype(np.int64), oracle.encode_frame(f, "f16")[:, 0, 0].astype(np.int64)
    nan = np.isnan(x)
    assert np.array_equal(got16[~nan], want16[~nan]) and np.all((got16[nan] & 0x7c00) == 0x7c00) and np.all((got16[nan] & 0x3ff) != 0)


@pytest.mark.parametrize("ids", [[0], [0, 0, 0]])
def test_frame_stored_in_the_colour_targets_own_format(oracle, ids):
    """SURVEY 8(f3), VERDICT r4 #5: the reference's pass writes into post_process.destination, whose format is
    TextureFormat::bevy_default() (pipeline.rs:311-315) -- 8-bit sRGB, or Rgba16Float under HDR.  brt_render_device stores the
    assembled frame in that format (BRT_FLAG_OUT_*), here into an IMPORTED buffer (a dma-buf, as a Vulkan export would hand over),
    read back through the exporter's mapping: bit for bit the oracle's f32 frame put through the oracle's store conversion.  The
    f32 frame stays the parity contract; the conversion is exact (brt_srgb_table.h)."""
    import os as _os
    import torch
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    w, h = 200, 117
    rng = np.random.default_rng(5)
    raster = (rng.random((h, w, 4), dtype=np.float32) * np.float32(1.5) - np.float32(0.2)).astype(np.float32)      # some values outside [0, 1]
    raster[::7, ::5, 0] = np.nan
    depth = (rng.random((h, w), dtype=np.float32) * np.float32(0.05)).astype(np.float32)
    with brt.RaytracePlugin(ids) as p:
        p.node.write_buffers(b)
        torch.cuda.set_device(ids[0])
        d_raster, d_depth = torch.from_numpy(raster).cuda(), torch.from_numpy(depth).cuda()
        torch.cuda.synchronize()
        formats = ((brt.FLAG_OUT_RGBA8_UNORM_SRGB, "srgb8", np.uint8), (brt.FLAG_OUT_RGBA16F, "f16", np.uint16),
                   (brt.FLAG_OUT_RGBA8_UNORM, "unorm8", np.uint8), (brt.FLAG_OUT_RGBA32F, None, np.float32))
        # (all four targets exist side by side and are released at the end: an exporter's mapping that lands on virtual addresses
        #  another mapping has just left was seen to read stale pages through hipMemcpy on one box -- HIP's VMM, not this library)
        targets = []
        for fmt, _, _ in formats:
            nbytes = w * h * brt.OUT_PIXEL_BYTES[fmt]
            fd, d_exported = p.debug_export_frame_fd(nbytes)
            targets.append((fd, d_exported, p.import_frame_fd(fd, nbytes, brt.EXTMEM_DMABUF_FD)))
        for (fmt, name, dt), (fd, d_exported, d_imported) in zip(formats, targets):
            for level in (brt.Raytracing.Pure, brt.Raytracing.FallbackRaster, brt.Raytracing.Skip):
                lvl, cam, win = brt.cover_camera(w, h, 3, 4, level, seed=0.31)
                p.node.render_device(lvl, cam, win, w, h, d_imported, d_raster.data_ptr(), d_depth.data_ptr(), flags=fmt)
                want, _ = oracle.render(b, lvl, cam, win, w, h, raster_rgba=raster, raster_depth=depth)
                got = p.debug_copy_to_host(d_exported, (h, w, 4), dt)
                if name is None:
                    assert_frames_equal(got, want)
                elif name == "f16":
                    w16 = oracle.encode_frame(want, name)
                    nan = np.isnan(want)
                    assert np.array_equal(got[~nan], w16[~nan]) and np.all((got[nan] & 0x7c00) == 0x7c00)
                else:
                    assert np.array_equal(got, oracle.encode_frame(want, name)), (name, level)
        for fd, d_exported, d_imported in targets:
            p.release_frame(d_imported)
            p.release_frame(d_exported)
            _os.close(fd)
        # the format flags belong to the assembled device frame: the host frame and a rank's tile stay f32
        lvl, cam, win = brt.cover_camera(w, h, 1, 1)
        with pytest.raises(brt.BrtError):
            p.node.run(lvl, cam, win, w, h, flags=brt.FLAG_OUT_RGBA16F)
        # ... and the root side of the one-process-per-GPU form applies it too (brt_deinterleave_device)
        tiles = torch.zeros((2, brt.tile_rows(h, 2), w, 4), dtype=torch.float32, device="cuda")
        for part in range(2):
            p.node.render_part_device(lvl, cam, win, w, h, part, 2, tiles[part].data_ptr())
        frame8 = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        p.node.deinterleave_device(tiles.data_ptr(), 2, w, h, frame8.data_ptr(), out_format=brt.FLAG_OUT_RGBA8_UNORM_SRGB)
        want, _ = oracle.render(b, lvl, cam, win, w, h)
        assert np.array_equal(frame8.cpu().numpy(), oracle.encode_frame(want, "srgb8"))


def test_half_sample_jobs_stress_every_tile_split_at_1080p():
    """The hand-over's ordering (brt_trace.h BRT_SLICE_SYNC: wide agent-scope stores, their acknowledgements, then the flag; the
    taker's exchange, then wide agent-scope loads) under load: EVERY tile of a 1920x1080 frame handed out as two half-sample jobs,
    a new seed per frame, each frame against the same frame rendered without a single split on a second context -- bytes and ray
    count.  (scripts/split_stress.py runs 200 frames of it: profiles/r05/split_stress.txt.)"""
    w, h, spp = 1920, 1080, 16
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    nb = brt.Buffers(b.models, b.materials, None)
    n_tiles = (w // 8) * (h // 8)
    with brt.RaytracePlugin([0]) as split, brt.RaytracePlugin([0]) as plain:
        split.set_tuning("BRT_SPLIT_FORCE", n_tiles)
        plain.set_tuning("BRT_SPLIT_TAIL", 0)
        taken = 0
        for i in range(14):
            seed = 0.5 if i < 2 else float(np.float32(0.01 + 0.07 * i))      # (the first two frames of the view build its order)
            lvl, cam, win = brt.cover_camera(w, h, spp, 8, seed=seed)
            fa = split.node.run(lvl, cam, win, w, h, buffers=nb)
            ra = split.node.last_stats["rays"]
            fb = plain.node.run(lvl, cam, win, w, h, buffers=nb)
            assert np.array_equal(fa.view(np.uint32), fb.view(np.uint32)) and ra == plain.node.last_stats["rays"], i
            split.debug_profile()
            plain.debug_profile()
            assert plain.last_order_meta["split_tiles"] == 0
            if i >= 2:
                assert split.last_order_meta["split_tiles"] > 0.3 * n_tiles, split.last_order_meta
                taken += split.last_order_meta["second_halves_taken"]
        assert taken > 12 * 0.2 * w * h


def test_order_built_on_host_and_on_gpu_render_the_same_frames(oracle, monkeypatch):
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    w, h = 320, 180
    lvl, cam, win = brt.cover_camera(w, h, 32, 8)
    want, cnt = oracle.render(b, lvl, cam, win, w, h)
    for host in (0, 1):
        with brt.RaytracePlugin([0]) as p:
            p.set_tuning("BRT_ORDER_ON_HOST", host)
            for _ in range(3):      # pre-pass + frame, then frames in the order measured by the one before
                got = p.node.run(lvl, cam, win, w, h, buffers=b, flags=brt.FLAG_COUNTERS)
                assert_frames_equal(got, want)
                assert {k: p.node.last_stats[k] for k in COUNTER_KEYS} == cnt


def test_lean_steady_state_kernel_renders_the_same_pixels(oracle, monkeypatch):
    """From the second frame of a Pure-level view on, frames that do not measure tile costs run the LEAN instantiation
    (no raster inputs, no depth average, no critical-tile logic) when the host can rule critical tiles out: the cover
    frame at 64 spp x 9 segments qualifies at this size too.  Same pixels and ray count as the oracle, frame after
    frame, with and without it (BRT_NO_LEAN=1), and at levels 1 / 2, which never use it."""
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    w, h = 480, 270
    lvl, cam, win = brt.cover_camera(w, h, 16, 2)        # 16 spp x 3 segments = 48 < half a lane's share (~69)
    want, cnt = oracle.render(b, lvl, cam, win, w, h)
    for no_lean in (0, 1):
        with brt.RaytracePlugin([0]) as p:
            p.set_tuning("BRT_NO_LEAN", no_lean)
            for frame in range(5):
                got = p.node.run(lvl, cam, win, w, h, buffers=b)
                assert_frames_equal(got, want)
                assert p.node.last_stats["rays"] == cnt["rays"], (no_lean, frame)
    rng = np.random.default_rng(1)
    raster = rng.random((h, w, 4), dtype=np.float32)
    depth = rng.random((h, w), dtype=np.float32) * np.float32(0.05)
    for level in (brt.Raytracing.FallbackRaster, brt.Raytracing.FallbackRaytraced):
        lvl, cam, win = brt.cover_camera(w, h, 16, 2, level)
        want, cnt = oracle.render(b, lvl, cam, win, w, h, raster_rgba=raster, raster_depth=depth)
        with brt.RaytracePlugin([0]) as p:
            for frame in range(3):
                got = p.node.run(lvl, cam, win, w, h, buffers=b, raster_rgba=raster, raster_depth=depth)
                assert_frames_equal(got, want)


def test_two_contexts_driven_from_two_host_threads(oracle):
    """The library keeps its state in the context (brt_ctx.h) -- nothing process-wide but the RCCL loader (std::call_once) and a
    thread-local error string -- so two host threads, each with a context of its own on the same GPU, may upload and render at the
    same time (ctypes releases the GIL for the duration of a call).  Each thread: its own scene, frames at two sizes, a far camera
    (tree rebuilt), an animated upload -- every frame the oracle's."""
    import threading
    jobs = []
    for kind, cam_fn, (w, h) in ((brt.SCENE_COVER, brt.cover_camera, (160, 90)), (brt.SCENE_RTIOW_FINAL, brt.rtiow_camera, (128, 72))):
        b = brt.generate_scene(kind, 1)
        steps = []
        for spp, scale in ((8, 1.0), (33, 1.0), (8, 40.0)):
            lvl, cam, win = cam_fn(w, h, spp, 6)
            if scale != 1.0:
                cam = cam.copy()
                cam["position"] *= np.float32(scale)
                cam["fov"] /= np.float32(scale)
            tree = brt.build_bvh_sah(b.models, brt.tree_reach(b.models, cam)[2])
            steps.append((lvl, cam, win, oracle.render(brt.Buffers(b.models, b.materials, tree), lvl, cam, win, w, h)))
        jobs.append((b, w, h, steps))
    errors = []

    def worker(b, w, h, steps):
        try:
            with brt.RaytracePlugin([0]) as p:
                for rep in range(6):
                    for i, (lvl, cam, win, (want, cnt)) in enumerate(steps):
                        got = p.node.run(lvl, cam, win, w, h, buffers=brt.Buffers(b.models, b.materials, None) if (rep + i) % 3 == 0 else None)
                        st = p.node.last_stats
                        if not np.array_equal(got.view(np.uint32), want.view(np.uint32)) or st["rays"] != cnt["rays"]:
                            errors.append(f"{w}x{h} rep {rep} step {i}: frame or ray count differs")
        except Exception as e:      # noqa: BLE001 -- reported by the asserting thread
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=j) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in threads), "a render thread did not finish"
    assert not errors, errors
