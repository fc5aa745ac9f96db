#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/.

1. rng_kat.json -- known-answer vectors for the integer RNG (reference
   assets/shaders/random.wgsl:3-15) and the per-pixel seed formula (reference
   assets/shaders/raytrace.wgsl:95), computed HERE with numpy uint32/float32 arithmetic,
   independently of the C oracle and of the HIP kernels.  The reference has no tests or golden
   vectors of its own, and it cannot be executed in this environment, so
   these are derived from reading the WGSL.
2. tiny_frames.npz -- whole tiny frames rendered by a SECOND independent restatement of the
   shader (numpy_restatement.py: numpy f32 scalars, written from the WGSL, not from the C
   oracle): hits, metal / glass / diffuse scatter, multi-sphere leaves, depth blend.  Pins the
   C oracle (and through it the HIP kernels) on complete paths.
3. cover_64x36.npz -- a small regression fixture: scene bytes + uniforms (inputs) and the
   frame + counters the C oracle produced for them (expected outputs).  It pins the oracle and
   the kernels against silent drift; it is NOT reference output ("parity unpinned").

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)

U32 = np.uint32
F32 = np.float32


def rng_next_int(state: int) -> int:
    """random.wgsl:8-15, wrapping u32"""
    with np.errstate(over="ignore"):
        old = U32(state) + U32(747796405) + U32(2891336453)
        word = ((old >> ((old >> U32(28)) + U32(4))) ^ old) * U32(277803737)
        return int((word >> U32(22)) ^ word)


def rng_next_float(state: int):
    """random.wgsl:3-6 -> (float32 value, new state)"""
    s = rng_next_int(state)
    return F32(s) / F32(0xFFFFFFFF), s


def seed(random_seed: float, px: int, py: int, w: int, h: int) -> int:
    """raytrace.wgsl:95 with uv = ((px+0.5)/W, (py+0.5)/H), all f32, left-assoc, truncating"""
    uvx = (F32(px) + F32(0.5)) / F32(w)
    uvy = (F32(py) + F32(0.5)) / F32(h)
    v = (F32(random_seed) * F32(10000.0)) * (uvx * F32(402.0)) * (uvy * F32(31.5))
    if not (v > 0):
        return 0
    if v >= F32(4294967296.0):
        return 0xFFFFFFFF
    return int(np.floor(v))


def main():
    rng = np.random.default_rng(20241022)
    starts = [0, 1, 12345, 0xFFFFFFFF, 0x80000000, 0xFFFFFF80, 0xFFFFFF7F] + [int(x) for x in rng.integers(0, 2**32, 24)]
    chains = []
    for s0 in starts:
        states, floats = [], []
        s = s0
        for _ in range(8):
            f, s = rng_next_float(s)
            states.append(s)
            floats.append(float(f))
        chains.append({"start": s0, "states": states, "floats": floats})
    seeds = []
    cases = [(0.5, 0, 0, 400, 225), (0.5, 199, 112, 400, 225), (0.5, 399, 224, 400, 225), (0.5, 960, 540, 1920, 1080),
             (0.999, 1919, 1079, 1920, 1080), (0.25, 100, 50, 1920, 1080), (0.0, 7, 9, 64, 36), (1.0, 3839, 2159, 3840, 2160)]
    for _ in range(24):
        w, h = int(rng.integers(1, 4000)), int(rng.integers(1, 2200))
        cases.append((float(F32(rng.random())), int(rng.integers(0, w)), int(rng.integers(0, h)), w, h))
    for c in cases:
        seeds.append({"random_seed": c[0], "px": c[1], "py": c[2], "w": c[3], "h": c[4], "seed": seed(*c)})
    with open(os.path.join(HERE, "rng_kat.json"), "w") as f:
        json.dump({"chains": chains, "seeds": seeds}, f, indent=1)
    print("wrote rng_kat.json")

    # tiny frames from the independent numpy restatement
    import bevyray_amd as brt
    from helpers import make_buffers, median_split_bvh, uniforms
    import numpy_restatement as npr
    M = brt.StandardMaterial
    data = [((0.0, -100.5, -1.0), 100.0, M(base_color=(0.5, 0.5, 0.5))),
            ((0.0, 0.0, -1.2), 0.5, M(base_color=(0.7, 0.3, 0.3), perceptual_roughness=0.0)),
            ((-1.05, 0.0, -1.0), 0.5, M(specular_transmission=1.0, ior=1.5)),
            ((1.05, 0.0, -1.0), 0.5, M(base_color=(0.8, 0.6, 0.2), metallic=1.0, perceptual_roughness=0.3)),
            ((0.3, -0.3, -0.4), 0.2, M(base_color=(0.2, 0.4, 0.9), metallic=0.5, specular_transmission=0.5, ior=0.8)),
            ((-0.4, 0.6, -1.6), 0.35, M(base_color=(0.9, 0.9, 0.9), metallic=1.0, perceptual_roughness=0.0))]
    cases = {}
    rng = np.random.default_rng(7)
    from helpers import chain_bvh

    def add_case(name, b, lvl, cam, win, w, h, level):
        raster = rng.random((h, w, 4), dtype=np.float32) if level != brt.Raytracing.Pure else None
        depth = (rng.random((h, w), dtype=np.float32) * np.float32(0.2)) if level != brt.Raytracing.Pure else None
        frame, rays = npr.render(b.models, b.materials, b.bvh, cam[0], win[0], int(level), w, h, raster, depth)
        for k, v in dict(models=b.models.view(np.uint8), materials=b.materials.view(np.uint8), bvh=b.bvh.view(np.uint8),
                         level=lvl.view(np.uint8), camera=cam.view(np.uint8), window=win.view(np.uint8), frame=frame,
                         rays=np.array([rays], np.uint64), size=np.array([w, h], np.uint32)).items():
            cases[f"{name}.{k}"] = v
        if raster is not None:
            cases[f"{name}.raster"] = raster
            cases[f"{name}.depth"] = depth
        with np.errstate(all="ignore"):
            print("numpy restatement:", name, frame.shape, rays, "rays, mean", np.nanmean(frame[..., :3]))

    for name, bvh_fn, w, h, spp, bounces, level, rseed in [
            ("ploc_pure", None, 16, 12, 3, 6, brt.Raytracing.Pure, 0.37),
            ("leaf2_blend", lambda m: median_split_bvh(m, 2), 12, 9, 2, 4, brt.Raytracing.FallbackRaytraced, 0.81),
            ("leaf3_raster", lambda m: median_split_bvh(m, 3), 10, 8, 2, 3, brt.Raytracing.FallbackRaster, 0.12)]:
        b = make_buffers(data, bvh_fn)
        lvl, cam, win = uniforms(w, h, spp=spp, bounces=bounces, pos=(0.2, 0.4, 1.6), target=(0.0, 0.0, -1.0), fov=0.9, seed=rseed,
                                 level=level, window_height=h * 3)
        add_case(name, b, lvl, cam, win, w, h, level)

    # (added later; the three cases above keep their bytes)
    # a piece of the 506-sphere cover scene through its PLOC tree, cover camera
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    lvl, cam, win = brt.cover_camera(12, 7, 2, 5, brt.Raytracing.Pure, 0.5)
    add_case("cover_ploc", b, lvl, cam, win, 12, 7, brt.Raytracing.Pure)
    # a 36-deep caterpillar tree: the 32-entry stack overflows and drops subtrees (raytrace.wgsl:320)
    deep = make_buffers([((0.0, 0.0, -5.0 - i), 0.5, M(base_color=(0.8, 0.3, 0.3))) for i in range(36)], chain_bvh)
    lvl, cam, win = uniforms(8, 8, spp=2, bounces=3, pos=(0, 0, 0), target=(0, 0, -1), fov=0.3, seed=0.5)
    add_case("deep_chain", deep, lvl, cam, win, 8, 8, brt.Raytracing.Pure)
    # every primary ray parallel to -Z (up parallel to the view direction: right = 0) from an origin ON a padded
    # slab plane: 1/d = inf, 0 * inf = NaN in the slab test (raytrace.wgsl:387-398 with minNum/maxNum)
    b = make_buffers(data[:4], lambda m: median_split_bvh(m, 1))
    lvl, cam, win = uniforms(6, 6, spp=2, bounces=3, pos=(0.0, 0.0, 0.5), target=(0.0, 0.0, -1.0), fov=0.6, seed=0.25)
    cam = cam.copy()
    cam["up"] = (0.0, 0.0, -1.0)
    cam["position"] = (float(np.float32(0.0) - (np.float32(0.5) + np.float32(0.1))), 0.0, 0.5)
    add_case("axis_parallel", b, lvl, cam, win, 6, 6, brt.Raytracing.Pure)
    np.savez_compressed(os.path.join(HERE, "tiny_frames.npz"), **cases)
    print("wrote tiny_frames.npz")

    # regression fixture from the C oracle
    import oracle_loader
    o = oracle_loader.load()
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    lvl, cam, win = brt.cover_camera(64, 36, 3, 4, brt.Raytracing.Pure, 0.5)
    frame, cnt = o.render(b, lvl, cam, win, 64, 36, threads=1)
    np.savez_compressed(os.path.join(HERE, "cover_64x36.npz"), models=b.models.view(np.uint8), materials=b.materials.view(np.uint8),
                        bvh=b.bvh.view(np.uint8), level=lvl.view(np.uint8), camera=cam.view(np.uint8),
                        window=win.view(np.uint8), frame=frame,
                        counters=np.array([cnt[k] for k in ("rays", "node_pops", "interior_visits", "sphere_tests", "hits")], np.uint64))
    print("wrote cover_64x36.npz", cnt)


if __name__ == "__main__":
    main()
