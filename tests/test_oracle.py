"""Pins the CPU oracle (oracle/bevyray_oracle.c) -- runs without a GPU.

The reference has no tests, golden vectors or fixtures and cannot be
executed here, so the oracle is pinned by (1) integer known-answer vectors from an independent
numpy restatement (tests/golden/make_golden.py) and the values hand-evaluated in SURVEY.md
8(a); (2) analytic single-ray cases derived from the WGSL; (3) brute-force == BVH on the
oracle itself; (4) a committed regression fixture of its own output.
"""
import ctypes as C
import json
import os

import numpy as np
import pytest

import bevyray_amd as brt
from helpers import (GOLDEN, chain_bvh, fixture_buffers, make_buffers, median_split_bvh, single_leaf_bvh, sky_color,
                     tiny_frame_cases, uniforms)

INF = np.float32(3.40282347e+38)


# ---- (1) integer known-answer vectors -------------------------------------------------------------

SURVEY_KAT = {  # start state -> states after 1..4 steps, floats
    0: ([0xA8BEEA3C, 0x0A2A1484, 0x1E93BE90, 0x75134D09], [0.6591631, 0.03970459, 0.1194419, 0.45732576]),
    1: ([0xB94DD992, 0x7D3246CC, 0xCB994A9C, 0x4DD1F399], [0.7238442, 0.48904842, 0.7953078, 0.30398485]),
    12345: ([0x21EBFEE8, 0x06C77023, 0x3D3393C9, 0xE142C31A], [0.13250726, 0.026480682, 0.23906825, 0.87992495]),
    0xFFFFFFFF: ([0x982FF7A5, 0xBEFDFCF3, 0x37723005, 0xD7A537C4], [0.59448195, 0.74606305, 0.21658611, 0.8423648]),
}
SURVEY_SEEDS = [  # (random_seed, px, py, W, H) -> seed
    ((0.5, 0, 0, 400, 225), 175), ((0.5, 199, 112, 400, 225), 15789178), ((0.5, 399, 224, 400, 225), 63095332),
    ((0.5, 960, 540, 1920, 1080), 15851657), ((0.999, 1919, 1079, 1920, 1080), 126411880),
    ((0.25, 100, 50, 1920, 1080), 77483),
]


def test_rng_matches_survey_table(oracle):
    for start, (states, floats) in SURVEY_KAT.items():
        s = start
        for want_s, want_f in zip(states, floats):
            s = oracle.lib.oracle_rng_next(s)
            assert s == want_s
        got, _ = oracle.rng_floats(start, 4)
        assert np.array_equal(got, np.array(floats, np.float32))


def test_rng_and_seed_match_numpy_fixture(oracle):
    kat = json.load(open(os.path.join(GOLDEN, "rng_kat.json")))
    for chain in kat["chains"]:
        got, end = oracle.rng_floats(chain["start"], len(chain["states"]))
        assert end == chain["states"][-1]
        assert np.array_equal(got, np.array(chain["floats"], np.float32))
    for s in kat["seeds"]:
        assert oracle.lib.oracle_seed(s["random_seed"], s["px"], s["py"], s["w"], s["h"]) == s["seed"]


def test_seed_matches_survey_table(oracle):
    for args, want in SURVEY_SEEDS:
        assert oracle.lib.oracle_seed(*args) == want


def test_rng_float_range_is_closed(oracle):
    # random.wgsl:5: f32(state)/f32(0xffffffff) reaches exactly 1.0 for states >= 0xFFFFFF80
    assert np.float32(0xFFFFFF80) / np.float32(0xFFFFFFFF) == np.float32(1.0)
    assert np.float32(0xFFFFFF7F) / np.float32(0xFFFFFFFF) < np.float32(1.0)


def test_unit_ball_is_inside_and_not_normalised(oracle):
    s = 7
    for _ in range(200):
        p, s = oracle.unit_ball(s)
        assert float(np.dot(p.astype(np.float64), p.astype(np.float64))) <= 1.0 + 1e-6
    # random.wgsl:28-30 returns the ball sample itself: lengths are spread, not all 1
    lens = []
    s = 99
    for _ in range(100):
        p, s = oracle.unit_ball(s)
        lens.append(np.linalg.norm(p))
    assert min(lens) < 0.8 and max(lens) <= 1.0 + 1e-6


def test_min_max_are_minnum_maxnum(oracle):
    nan = float("nan")
    assert oracle.lib.oracle_min(nan, 2.0) == 2.0 and oracle.lib.oracle_min(2.0, nan) == 2.0
    assert oracle.lib.oracle_max(nan, -3.0) == -3.0 and oracle.lib.oracle_max(-3.0, nan) == -3.0
    assert np.signbit(np.float32(oracle.lib.oracle_min(0.0, -0.0))) and np.signbit(np.float32(oracle.lib.oracle_min(-0.0, 0.0)))
    assert not np.signbit(np.float32(oracle.lib.oracle_max(0.0, -0.0))) and not np.signbit(np.float32(oracle.lib.oracle_max(-0.0, 0.0)))
    assert oracle.lib.oracle_min(1.0, 2.0) == 1.0 and oracle.lib.oracle_max(1.0, 2.0) == 2.0


# ---- (2) analytic cases derived from the WGSL -------------------------------------------------------

def _f3(v):
    return (C.c_float * 3)(*v)


def test_slab_test_cases(oracle):
    dst = oracle.lib.oracle_ray_bounding_dst
    # origin outside, box ahead: t_near (raytrace.wgsl:392-396)
    assert dst(_f3([0, 0, 0]), _f3([0, 0, -1]), _f3([-1, -1, -5]), _f3([1, 1, -3])) == 3.0
    # origin inside: 0.0 (select(0.0, t_near, t_near > 0.0))
    assert dst(_f3([0, 0, -4]), _f3([0, 0, -1]), _f3([-1, -1, -5]), _f3([1, 1, -3])) == 0.0
    # box behind: INF
    assert np.float32(dst(_f3([0, 0, 0]), _f3([0, 0, 1]), _f3([-1, -1, -5]), _f3([1, 1, -3]))) == INF
    # parameter is in units of |d| (direction not normalised)
    assert dst(_f3([0, 0, 0]), _f3([0, 0, -2]), _f3([-1, -1, -5]), _f3([1, 1, -3])) == 1.5
    # zero direction component, origin strictly inside that slab: +-inf sorts out, still a hit
    assert dst(_f3([0, 0, 0]), _f3([0, 0, -1]), _f3([-1, -1, -5]), _f3([1, 1, -3])) == 3.0
    # zero component and origin ON the slab's max plane: t_min.x = -inf, t_max.x = 0*inf = NaN;
    # minNum/maxNum drop the NaN, so t1.x = t2.x = -inf -> t_far = -inf -> miss
    assert np.float32(dst(_f3([1, 0, 0]), _f3([0, 0, -1]), _f3([-1, -1, -5]), _f3([1, 1, -3]))) == INF
    # ... ON the min plane: t_min.x = NaN, t_max.x = +inf -> t1.x = t2.x = +inf -> t_near = +inf -> miss
    assert np.float32(dst(_f3([-1, 0, 0]), _f3([0, 0, -1]), _f3([-1, -1, -5]), _f3([1, 1, -3]))) == INF


def test_hit_sphere_near_root_only(oracle):
    hs = oracle.lib.oracle_hit_sphere
    assert hs(_f3([0, 0, 0]), _f3([0, 0, -1]), _f3([0, 0, -5]), 1.0) == 4.0       # (h - sqrt(disc)) / a
    assert hs(_f3([0, 0, 0]), _f3([0, 0, -2]), _f3([0, 0, -5]), 1.0) == 2.0       # a = dot(d,d) = 4
    assert hs(_f3([0, 0, 0]), _f3([0, 1, 0]), _f3([0, 0, -5]), 1.0) == -1.0       # disc < 0
    assert hs(_f3([0, 0, -5]), _f3([0, 0, -1]), _f3([0, 0, -5]), 1.0) == -1.0 + 0  # inside: near root is negative (-1)


def _raycast(oracle, b, o, d):
    out7 = (C.c_float * 7)()
    mat, front = C.c_uint32(0), C.c_int(0)
    oracle.lib.oracle_raycast(b.models.ctypes.data, len(b.models), b.bvh.ctypes.data, len(b.bvh), _f3(o), _f3(d), out7,
                              C.byref(mat), C.byref(front))
    return np.array(list(out7), np.float32), mat.value, front.value


def test_ray_from_inside_a_sphere_does_not_hit_it(oracle):
    # raytrace.wgsl:353,382: only the near root, rejected by t > 0.001
    b = make_buffers([((0, 0, -5), 1.0, brt.StandardMaterial())], single_leaf_bvh)
    hit, _, _ = _raycast(oracle, b, (0, 0, -5), (0, 0, -1))
    assert hit[0] == INF
    hit, mat, front = _raycast(oracle, b, (0, 0, 0), (0, 0, -1))
    assert hit[0] == 4.0 and mat == 0 and front == 1
    assert np.allclose(hit[4:7], [0, 0, 1])


def test_first_encountered_sphere_wins_exact_ties(oracle):
    # raytrace.wgsl:354: strict `<`
    data = [((0, 0, -5), 1.0, brt.StandardMaterial()), ((0, 0, -5), 1.0, brt.StandardMaterial())]
    b = make_buffers(data, single_leaf_bvh)
    _, mat, _ = _raycast(oracle, b, (0, 0, 0), (0, 0, -1))
    assert mat == 0


def test_miss_returns_gamma_sky_and_fallback_depth(oracle):
    # raytrace.wgsl:198-201, 219-223, 364-369: every pixel misses -> sqrt(sky(dir)), alpha 1
    b = make_buffers([((0, 0, 50), 0.5, brt.StandardMaterial())], single_leaf_bvh)   # behind the camera
    w, h = 16, 9
    lvl, cam, win = uniforms(w, h, spp=2, bounces=3, pos=(0, 0, 0), target=(0, 0, -1), fov=0.8, seed=0.0)
    frame, cnt = oracle.render(b, lvl, cam, win, w, h, threads=1)
    assert cnt["rays"] == w * h * 2 and cnt["hits"] == 0
    # seed 0.0 -> every pixel's RNG state is 0 -> same jitter sequence; recompute the expected colour
    want = sky_color(oracle, cam, win, w, h, spp=2)
    assert np.array_equal(frame[..., :3], want)
    assert np.all(frame[..., 3] == 1.0)


def test_zero_bounces_on_a_diffuse_hit_is_black(oracle):
    # raytrace.wgsl:189,215-217: bounce_count 0 -> one segment; a hit that scatters exhausts the loop
    mat = brt.StandardMaterial(base_color=(0.8, 0.8, 0.8), perceptual_roughness=0.0)
    b = make_buffers([((0, 0, -3), 100.0, mat)], single_leaf_bvh)   # fills the whole view
    lvl, cam, win = uniforms(8, 8, spp=4, bounces=0, pos=(0, 0, 200), target=(0, 0, -3), fov=0.2, seed=0.5)
    frame, cnt = oracle.render(b, lvl, cam, win, 8, 8, threads=1)
    assert cnt["rays"] == 8 * 8 * 4
    assert np.all(frame[..., :3] == 0.0) and np.all(frame[..., 3] == 1.0)


def test_zero_samples_is_nan(oracle):
    # raytrace.wgsl:169: 0/0
    b = make_buffers([((0, 0, -3), 1.0, brt.StandardMaterial())], single_leaf_bvh)
    lvl, cam, win = uniforms(4, 4, spp=0, bounces=2, pos=(0, 0, 0), target=(0, 0, -1), fov=0.5, seed=0.5)
    frame, cnt = oracle.render(b, lvl, cam, win, 4, 4, threads=1)
    assert cnt["rays"] == 0 and np.all(np.isnan(frame[..., :3])) and np.all(frame[..., 3] == 1.0)


def test_stack_overflow_drops_pending_subtrees(oracle):
    # raytrace.wgsl:320: the loop exits when stack_index reaches 32.  A 40-deep "caterpillar"
    # whose every box is hit grows the stack by one per level.
    n = 40
    data = [((0.0, 0.0, -5.0 - i), 0.5, brt.StandardMaterial()) for i in range(n)]
    deep = make_buffers(data, lambda m: chain_bvh(m, far_first=False))
    flat = make_buffers(data, single_leaf_bvh)
    o, d = (0, 0, 0), (0, 0, -1)
    hit_flat, mat_flat, _ = _raycast(oracle, flat, o, d)
    assert hit_flat[0] == np.float32(4.5) and mat_flat == 0
    hit_deep, _, _ = _raycast(oracle, deep, o, d)
    # nearest sphere sits at the bottom of the chain: the walk overflows before reaching it
    assert hit_deep[0] == INF
    # same chain but shallow enough: identical to brute force
    data30 = data[:28]
    deep30 = make_buffers(data30, lambda m: chain_bvh(m, far_first=False))
    flat30 = make_buffers(data30, single_leaf_bvh)
    assert np.array_equal(_raycast(oracle, deep30, o, d)[0], _raycast(oracle, flat30, o, d)[0])


# ---- (3) brute force == BVH ---------------------------------------------------------------------------

@pytest.mark.parametrize("seed", [0.25, 0.999])
def test_brute_force_equals_bvh_on_cover_scene(oracle, seed):
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    flat = brt.Buffers(b.models, b.materials, single_leaf_bvh(b.models))
    med = brt.Buffers(b.models, b.materials, median_split_bvh(b.models, leaf_size=3))
    w, h = 48, 27
    lvl, cam, win = brt.cover_camera(w, h, 2, 5, brt.Raytracing.Pure, seed)
    f_bvh, c_bvh = oracle.render(b, lvl, cam, win, w, h)
    f_flat, c_flat = oracle.render(flat, lvl, cam, win, w, h)
    f_med, c_med = oracle.render(med, lvl, cam, win, w, h)
    assert np.array_equal(f_bvh, f_flat) and np.array_equal(f_bvh, f_med)
    assert c_bvh["rays"] == c_flat["rays"] == c_med["rays"] and c_bvh["hits"] == c_flat["hits"]
    assert c_flat["sphere_tests"] == c_flat["rays"] * len(b.models)
    assert c_bvh["sphere_tests"] < c_flat["sphere_tests"] / 10


def test_threads_and_row_ranges_do_not_change_pixels(oracle):
    b = brt.generate_scene(brt.SCENE_COVER, 2)
    w, h = 40, 24
    lvl, cam, win = brt.cover_camera(w, h, 2, 3)
    full, cnt = oracle.render(b, lvl, cam, win, w, h, threads=1)
    par, cnt2 = oracle.render(b, lvl, cam, win, w, h, threads=4)
    assert np.array_equal(full, par) and cnt == cnt2
    part, _ = oracle.render(b, lvl, cam, win, w, h, rows=(5, 9), threads=2)
    assert np.array_equal(part[5:9], full[5:9]) and np.all(part[:5] == 0) and np.all(part[9:] == 0)


# ---- (3b) whole frames from the second, independent restatement (numpy f32) -----------------------------

def test_oracle_matches_numpy_restatement_frames(oracle):
    n = 0
    for name, b, lvl, cam, win, w, h, raster, depth, frame, rays in tiny_frame_cases():
        got, cnt = oracle.render(b, lvl, cam, win, w, h, raster_rgba=raster, raster_depth=depth, threads=2)
        assert np.array_equal(got.view(np.uint32), frame.view(np.uint32)), name
        assert cnt["rays"] == rays, name
        n += 1
    assert n == 6


# ---- (4) regression fixture ----------------------------------------------------------------------------

def test_golden_cover_fixture(oracle):
    b, lvl, cam, win, frame, counters = fixture_buffers()
    got, cnt = oracle.render(b, lvl, cam, win, 64, 36, threads=2)
    assert np.array_equal(got.view(np.uint32), frame.view(np.uint32))
    assert [cnt[k] for k in ("rays", "node_pops", "interior_visits", "sphere_tests", "hits")] == list(counters)


# ---- levels (raytrace.wgsl:97-122) ----------------------------------------------------------------------

def test_levels_blend_against_raster(oracle):
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    w, h = 32, 18
    rng = np.random.default_rng(5)
    raster = rng.random((h, w, 4), dtype=np.float32)
    depth = rng.random((h, w), dtype=np.float32) * np.float32(0.02)   # reverse-Z: near/dist, small = far
    depth[0, :] = 0.0                                                  # nothing rasterised over the top (sky) row
    frames = {}
    for level in brt.Raytracing:
        lvl, cam, win = brt.cover_camera(w, h, 2, 3, level, 0.5)
        frames[level], _ = oracle.render(b, lvl, cam, win, w, h, raster_rgba=raster, raster_depth=depth)
    assert np.array_equal(frames[brt.Raytracing.Skip], raster)
    pure = frames[brt.Raytracing.Pure]
    for level in (brt.Raytracing.FallbackRaster, brt.Raytracing.FallbackRaytraced):
        f = frames[level]
        is_raster = np.all(f == raster, axis=-1)
        is_traced = np.all(f == pure, axis=-1)
        assert np.all(is_raster | is_traced) and is_raster.any() and is_traced.any()
    # misses (raytrace.wgsl:177-182,108-113): level 1 -> far+10 > far -> -1, the raster wins even where
    # its depth is the cleared 0; level 2 -> near/(far-1) > 0, the traced sky wins over cleared depth
    sky = np.all(frames[brt.Raytracing.FallbackRaytraced][0] == pure[0], axis=-1)
    assert sky.all()
    assert np.all(frames[brt.Raytracing.FallbackRaster][0] == raster[0])
    # without raster inputs: depth 0 -> level 2 == level 3
    lvl, cam, win = brt.cover_camera(w, h, 2, 3, brt.Raytracing.FallbackRaytraced, 0.5)
    f2, _ = oracle.render(b, lvl, cam, win, w, h)
    assert np.array_equal(f2, pure)


# ---- alternative readings of the three implementation-defined points (tests/golden/policy_frames.npz) ----------------

POLICIES = {
    "default": {},
    "or_short_circuit": dict(or_short_circuit=True),
    "minmax_select": dict(minmax="select"),
    "pow_exp2log2": dict(pow="exp2log2"),
    "all_three": dict(or_short_circuit=True, minmax="select", pow="exp2log2"),
}


def _policy_cases():
    z = np.load(os.path.join(GOLDEN, "policy_frames.npz"))
    for name in sorted({k.split(".")[0] for k in z.files}):
        g = lambda k: z[f"{name}.{k}"]
        b = brt.Buffers(g("models").view(brt.MODEL_DTYPE), g("materials").view(brt.MATERIAL_DTYPE), g("bvh").view(brt.BVH_NODE_DTYPE))
        w, h = (int(x) for x in g("size"))
        yield name, b, g("level").view(brt.LEVEL_DTYPE), g("camera").view(brt.CAMERA_DTYPE), g("window").view(brt.WINDOW_DTYPE), w, h, z


def _moved(a, b):
    same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
    return int((~same).any(axis=2).sum())


def test_oracle_reproduces_every_policy_fixture(oracle):
    """The frames were rendered by the independent numpy restatement under each policy; the C oracle under the
    same policy must give the same bits (and the same ray count: the `||` policy changes the RNG stream)."""
    for name, b, lvl, cam, win, w, h, z in _policy_cases():
        for pname, pol in POLICIES.items():
            with oracle.policy(**pol):
                got, cnt = oracle.render(b, lvl, cam, win, w, h)
            want = z[f"{name}.frame.{pname}"]
            assert _moved(got, want) == 0, (name, pname)
            assert cnt["rays"] == int(z[f"{name}.rays.{pname}"][0]), (name, pname)
    # the policy switch is process-wide: it must be back at the default
    name, b, lvl, cam, win, w, h, z = next(_policy_cases())
    got, _ = oracle.render(b, lvl, cam, win, w, h)
    assert _moved(got, z[f"{name}.frame.default"]) == 0


def test_which_pixels_each_policy_moves():
    """What DESIGN.md section 2 states about the three unpinned choices, as data:
      * `||` short circuit: moves pixels wherever glass with ior < 1 is hit (ri = 1/ior > 1, total internal
        reflection on ENTRY); with ior >= 1 `cannot_refract` needs a hit from inside, which hit_sphere's
        near-root-only rule (raytrace.wgsl:382) never produces;
      * compare-select min/max: moves pixels only when a bound (or the ray) is NaN;
      * pow as exp2(5 log2 x): differs from the multiplies in the last bits of the reflectance, which only matters
        when `reflectance > rng` flips -- no pixel of these frames."""
    z = np.load(os.path.join(GOLDEN, "policy_frames.npz"))
    moved = {(s, p): _moved(z[f"{s}.frame.{p}"], z[f"{s}.frame.default"])
             for s in ("glass_tir", "axis_parallel", "nan_box") for p in POLICIES}
    assert moved[("glass_tir", "or_short_circuit")] > 0 and moved[("glass_tir", "all_three")] > 0
    assert moved[("glass_tir", "minmax_select")] == 0 and moved[("glass_tir", "pow_exp2log2")] == 0
    assert all(moved[("axis_parallel", p)] == 0 for p in POLICIES)      # 0 * inf = NaN in the slab test: same boxes entered
    assert moved[("nan_box", "minmax_select")] > 0 and moved[("nan_box", "pow_exp2log2")] == 0


def test_short_circuit_policy_cannot_be_seen_on_the_benchmark_scenes(oracle):
    """Every glass sphere of the cover and RTIOW scenes has ior 1.5: `cannot_refract` is never true there (no hits
    from inside a sphere, raytrace.wgsl:382), so the `||` policy does not change a single bit of those frames."""
    for kind in (brt.SCENE_COVER, brt.SCENE_RTIOW_FINAL):
        b = brt.generate_scene(kind, 1)
        lvl, cam, win = brt.cover_camera(240, 135, 6, 8)
        base, c0 = oracle.render(b, lvl, cam, win, 240, 135)
        with oracle.policy(or_short_circuit=True):
            alt, c1 = oracle.render(b, lvl, cam, win, 240, 135)
        assert _moved(base, alt) == 0 and c0 == c1


def test_compare_wgpu_frame_script_names_the_policy_of_a_dump(tmp_path):
    """scripts/compare_wgpu_frame.py (the one-command wgpu comparison for a machine with a Rust toolchain, SURVEY 8(f)
    rank 4) on dumps made from the ior < 1 policy fixture: a frame rendered under the short-circuit reading of `||` is
    recognised as exactly that policy (f32 dump, bit for bit), a default-policy frame quantised to sRGB8 as the default."""
    import subprocess
    import sys
    z = np.load(os.path.join(GOLDEN, "policy_frames.npz"))
    g = lambda k: z[f"glass_tir.{k}"]
    w, h = (int(x) for x in g("size"))
    d = tmp_path / "dump"
    d.mkdir()
    for name, key in (("models.bin", "models"), ("materials.bin", "materials"), ("bvh.bin", "bvh"), ("camera.bin", "camera"),
                      ("window.bin", "window"), ("level.bin", "level")):
        g(key).tofile(str(d / name))
    script = os.path.join(os.path.dirname(GOLDEN), "..", "scripts", "compare_wgpu_frame.py")
    g("frame.or_short_circuit").astype(np.float32).tofile(str(d / "frame.bin"))
    proc = subprocess.run([sys.executable, script, "--dir", str(d), "--width", str(w), "--height", str(h)], capture_output=True, text=True)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("MATCH")]
    assert lines and all("short-circuits" in ln for ln in lines) and any("min/max = minnum, pow = mul" in ln for ln in lines)
    assert any(ln.startswith("differs") and "default policy" in ln for ln in proc.stdout.splitlines())
    # an 8-bit sRGB dump of the default frame
    f = np.clip(np.nan_to_num(g("frame.default").astype(np.float64)), 0.0, 1.0)
    enc = np.where(f <= 0.0031308, 12.92 * f, 1.055 * np.power(f, 1.0 / 2.4) - 0.055)
    q = np.floor(enc * 255.0 + 0.5).astype(np.uint8)
    q[..., 3] = 255
    q.tofile(str(d / "frame8.bin"))
    proc = subprocess.run([sys.executable, script, "--dir", str(d), "--width", str(w), "--height", str(h), "--frame", "frame8.bin",
                           "--srgb8", "--tolerance", "0"], capture_output=True, text=True)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    assert any(ln.startswith("MATCH") and "default policy" in ln for ln in proc.stdout.splitlines())
