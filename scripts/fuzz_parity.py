#!/usr/bin/env python3
"""Soak test: adversarial random scenes, HIP path (through the C ABI) vs the CPU oracle, bit for bit.

    python scripts/fuzz_parity.py --cases 600 --seed 7 [--log gpurun_out/fuzz.txt]

Beyond tests/test_parity_gpu.py::test_randomized_scenes_bit_exact this injects the values the
reference never guards against (raytrace.wgsl has no validation): zero / negative / huge / NaN /
infinite radii and positions, coincident spheres (exact ties, raytrace.wgsl:353 strict `<`),
materials outside [0,1], ior 0 and inf, cameras inside spheres, degenerate up vectors, window
height 0 (infinite jitter), NaN / huge / negative random_seed, sample_count 0, bounce counts past the
stack size, hand-made trees deeper than the 32-entry stack, multi-sphere leaves, and callee-built
trees (GPU SAH and GPU PLOC, each checked byte for byte against its CPU builder).  A failing case is dumped to
gpurun_out/fuzz_fail_<n>.npz.  Needs an MI355X; the oracle is the checker (test infrastructure).
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import bevyray_amd as brt                      # noqa: E402
import oracle_loader                           # noqa: E402
from helpers import chain_bvh, make_buffers, median_split_bvh, single_leaf_bvh, uniforms   # noqa: E402

COUNTER_KEYS = ("rays", "node_pops", "interior_visits", "sphere_tests", "hits")
SPECIAL = [0.0, -0.0, np.nan, np.inf, -np.inf, 3.0e38, -3.0e38, 1e-38, 1e-45, 1.0, -1.0, 1e6, 1e-6]


def special(rng):
    return float(SPECIAL[int(rng.integers(0, len(SPECIAL)))])


def random_material(rng, wild):
    def unit():
        if wild and rng.random() < 0.15:
            return special(rng) if rng.random() < 0.5 else float(rng.uniform(-2, 3))
        return float(rng.choice([0.0, 1.0, rng.random()]))
    ior = float(rng.uniform(0.3, 3.0))
    if wild and rng.random() < 0.2:
        ior = special(rng)
    return brt.StandardMaterial(base_color=tuple(float(x) for x in rng.random(3)), metallic=unit(),
                                perceptual_roughness=unit(), ior=ior, specular_transmission=unit())


def random_case(rng):
    wild = rng.random() < 0.5            # half of the cases stay "sane" but structurally varied
    n = int(rng.choice([1, 2, 3, int(rng.integers(4, 40)), int(rng.integers(40, 400)), int(rng.integers(400, 3000)),
                        int(rng.integers(8200, 12000)),       # does not fit LDS: top-of-tree tile, 16-bit descriptors
                        int(rng.integers(16400, 18000))],     # > 16382 spheres: 32-bit descriptors, all from L2
                       p=[0.05, 0.05, 0.05, 0.45, 0.285, 0.1, 0.01, 0.005]))
    spread = float(rng.choice([0.5, 4.0, 30.0]))
    data = []
    for i in range(n):
        r = float(rng.uniform(0.05, 1.5)) if rng.random() < 0.9 else float(rng.uniform(20, 2000))
        pos = [float(x) for x in rng.uniform(-spread, spread, 3)]
        if r > 10:
            pos[1] = -r - 1.0
        data.append((tuple(pos), r, random_material(rng, wild)))
    if n >= 2 and rng.random() < 0.3:    # coincident spheres: exact ties in t
        for _ in range(int(rng.integers(1, 4))):
            a, b = rng.integers(0, n, 2)
            data[int(b)] = (data[int(a)][0], data[int(a)][1], data[int(b)][2])

    topo = int(rng.integers(0, 5))
    if n > 400 and topo in (1, 3):
        topo = 0                          # single leaf / chain of 1000s of spheres: too slow for the oracle
    bvh_fn = [None, single_leaf_bvh, lambda m: median_split_bvh(m, int(rng.integers(1, 9))), chain_bvh,
              lambda m: chain_bvh(m, far_first=True)][topo]
    if topo in (3, 4) and n < 2:
        bvh_fn = single_leaf_bvh
    b = brt.prepare_buffers([(p, brt.RaytracedSphere(r), m) for p, r, m in data])
    models = b.models.copy()
    if wild:
        for _ in range(int(rng.integers(0, 4))):
            i = int(rng.integers(0, n))
            if rng.random() < 0.5:
                models["radius"][i] = special(rng)
            else:
                models["position"][i, int(rng.integers(0, 3))] = special(rng)
    bvh = None if bvh_fn is None else bvh_fn(models)
    if bvh is not None and wild and rng.random() < 0.2:      # a poisoned box
        k = int(rng.integers(0, len(bvh)))
        bvh["bounds_min" if rng.random() < 0.5 else "bounds_max"][k, int(rng.integers(0, 3))] = special(rng)
    b = brt.Buffers(models, b.materials, bvh)

    big = n > 400
    w, h = (int(rng.integers(1, 40)), int(rng.integers(1, 30))) if big else (int(rng.integers(1, 90)), int(rng.integers(1, 60)))
    if rng.random() < 0.1:
        h = int(rng.integers(60, 200))    # enough 8-row strips for every part of an 8-way split
    # (32+ samples: the first frame of a view runs a pre-pass, which for a scene walked from an LDS tile + L2 also re-numbers the tree's
    #  records by their use -- brt_api.cpp apply_hot_order)
    spp = int(rng.choice([0, 1, 2, 3, 5, 9, 33, 64], p=[0.03, 0.29, 0.24, 0.19, 0.11, 0.06, 0.05, 0.03]))
    bounces = int(rng.choice([0, 1, 3, 8, 20, 70], p=[0.1, 0.15, 0.3, 0.3, 0.1, 0.05]))
    per_ray = n if topo in (1, 3, 4) else 40          # sphere/box tests per ray, roughly
    while w * h * max(spp, 1) * (bounces + 1) * per_ray > 2e8 and w * h > 1:   # keep the oracle under about a second
        w, h = max(1, w // 2), max(1, (h * 2) // 3)
    pos = tuple(float(x) for x in rng.uniform(-8, 8, 3))
    far_cam = rng.random() < 0.35
    if far_cam:                           # a camera far outside the scene: distance log-uniform in [1, 2000] (the callee's tree must follow it:
        dist = float(np.exp(rng.uniform(0.0, np.log(2000.0))))        # brt_sah.h "leaf boxes")
        v = rng.standard_normal(3)
        pos = tuple(float(x) for x in dist * v / np.linalg.norm(v))
    if rng.random() < 0.15:               # camera inside (or on) a sphere
        i = int(rng.integers(0, n))
        pos = tuple(float(x) for x in np.nan_to_num(models["position"][i], nan=0.0, posinf=9.0, neginf=-9.0))
    target = tuple(float(x) for x in rng.uniform(-1, 1, 3))
    up = (0.0, 1.0, 0.0)
    if wild and rng.random() < 0.1:
        up = tuple(float(x) for x in rng.uniform(-1, 1, 3))
    fov = float(rng.uniform(0.2, 1.5))
    if far_cam:                           # ... looking at the scene through a field of view as narrow as 1e-3
        fov = float(np.exp(rng.uniform(np.log(1e-3), np.log(1.5))))
    if wild and rng.random() < 0.15:
        fov = float(rng.choice([1e-4, 3.1, 3.14159274, 0.0, 6.0]))
    seed = float(np.float32(rng.random()))
    if wild and rng.random() < 0.2:
        seed = float(rng.choice([0.0, 1.0, -0.5, 1e9, np.nan, np.inf, 1e-30]))
    window_height = int(rng.integers(1, 2400))
    if wild and rng.random() < 0.05:
        window_height = 0
    near, far = 0.1, 1000.0
    if wild and rng.random() < 0.1:
        near, far = float(rng.choice([0.0, 5.0, 1e4])), float(rng.choice([0.0, 1.0, 1e30]))
    lvl, cam, win = uniforms(w, h, spp=spp, bounces=bounces, pos=pos, target=target, fov=fov, seed=seed,
                             level=brt.Raytracing(int(rng.integers(0, 4))), near=near, far=far, up=up,
                             window_height=window_height)
    raster = depth = None
    if rng.random() < 0.4:
        raster = rng.random((h, w, 4), dtype=np.float32)
        depth = rng.random((h, w), dtype=np.float32) * np.float32(rng.choice([0.05, 1.0]))
        if wild and rng.random() < 0.3:
            depth[rng.random((h, w)) < 0.1] = np.float32(special(rng))
    mode = str(rng.choice(["run", "multi", "parts", "simple"], p=[0.6, 0.15, 0.15, 0.1]))
    return dict(buffers=b, level=lvl, camera=cam, window=win, w=w, h=h, raster=raster, depth=depth, wild=wild, topo=topo,
                mode=mode, n_parts=int(rng.choice([2, 3, 5, 8])), far_cam=far_cam)


def frames_differ(got, want):
    got, want = np.asarray(got, np.float32), np.asarray(want, np.float32)
    same = got.view(np.uint32) == want.view(np.uint32)
    both_nan = np.isnan(got) & np.isnan(want)
    return int((~(same | both_nan)).sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=300)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--log", default=None)
    ap.add_argument("--seconds", type=float, default=1e9, help="stop after this much wall time")
    args = ap.parse_args()

    import torch
    from bevyray_amd.parallel import frame_rows_of_part
    oracle = oracle_loader.load()
    plugin = brt.RaytracePlugin([0])
    multi = brt.RaytracePlugin([0, 0, 0])      # three sub-contexts on one GPU: the in-process strip split of brt_render
    node = plugin.node

    def render(c, b, counters=True):
        """-> (frame, stats or None): the four ways a frame can be produced through the C ABI"""
        args = (c["level"], c["camera"], c["window"], c["w"], c["h"])
        if not counters:        # the production instantiation (the only one that runs half-sample jobs): frame + ray count
            got = node.run(*args, buffers=b, raster_rgba=c["raster"], raster_depth=c["depth"])
            return got, node.last_stats
        if c["mode"] == "multi":
            got = multi.node.run(*args, buffers=b, raster_rgba=c["raster"], raster_depth=c["depth"], flags=brt.FLAG_COUNTERS)
            return got, multi.node.last_stats
        if c["mode"] == "simple":
            got = node.run(*args, buffers=b, raster_rgba=c["raster"], raster_depth=c["depth"],
                           flags=brt.FLAG_COUNTERS | brt.FLAG_KERNEL_SIMPLE)
            return got, node.last_stats
        if c["mode"] == "parts":
            w, h, n_parts = c["w"], c["h"], c["n_parts"]
            node.write_buffers(b)
            rows = brt.tile_rows(h, n_parts)
            tiles = torch.zeros((n_parts, rows, w, 4), dtype=torch.float32, device="cuda")
            d_r = None if c["raster"] is None else torch.from_numpy(c["raster"]).cuda()
            d_d = None if c["depth"] is None else torch.from_numpy(c["depth"]).cuda()
            rays, reach = 0, 0.0
            for p in range(n_parts):
                st = node.render_part_device(*args, p, n_parts, tiles[p].data_ptr(),
                                             d_raster_rgba=0 if d_r is None else d_r.data_ptr(),
                                             d_raster_depth=0 if d_d is None else d_d.data_ptr())
                rays += st["rays"]
                reach = max(reach, st["tree_reach"])
            frame = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
            node.deinterleave_device(tiles.data_ptr(), n_parts, w, h, frame.data_ptr())
            torch.cuda.synchronize()
            return frame.cpu().numpy(), {"rays": rays, "tree_reach": reach}
        got = node.run(*args, buffers=b, raster_rgba=c["raster"], raster_depth=c["depth"], flags=brt.FLAG_COUNTERS)
        return got, node.last_stats
    rng = np.random.default_rng(args.seed)
    t_start = time.time()
    fails, done, pixels, rays, ploc_checked, rejected, tight_checked, split_checked, prod_checked = 0, 0, 0, 0, 0, 0, 0, 0, 0
    reach_checked, ref_marginal, hot_checked = 0, 0, 0
    lines = []
    t_progress = time.time()
    for case in range(args.cases):
        if time.time() - t_start > args.seconds:
            break
        if time.time() - t_progress > 60:      # a line a minute: the GPU boxes take a silent command for a hung one
            t_progress = time.time()
            print(f"... {done} cases so far, {fails} failed, {time.time() - t_start:.0f} s", flush=True)
        c = random_case(rng)
        b = c["buffers"]
        # kernel variant of this case: default plan, top-of-tree tile of a few records, everything from L2, knobs live
        # ... the last tiles of the order as two half-sample jobs (production instantiations only: "parts" cases and the extra render below)
        variant = [{}, {}, {"BRT_FORCE_LDS_TOP": str(int(rng.integers(1, 200)))}, {"BRT_FORCE_GLOBAL_SCENE": "1"},
                   {"BRT_TUNABLE": "1"}, {"BRT_FORCE_LDS_TOP": "100000", "BRT_TUNABLE": "1"},
                   {"BRT_SPLIT_FORCE": str(int(rng.integers(1, 400)))},
                   {"BRT_SPLIT_FORCE": str(int(rng.integers(1, 400))), "BRT_FORCE_GLOBAL_SCENE": "1"}][int(rng.integers(0, 8))]
        for pl in (plugin, multi):          # knobs live in the context (brt_set_tuning), not in the environment
            for k in ("BRT_FORCE_LDS_TOP", "BRT_FORCE_GLOBAL_SCENE", "BRT_TUNABLE", "BRT_SPLIT_FORCE"):
                pl.set_tuning(k, int(variant.get(k, 0)))
        try:
            if b.bvh is None:             # callee-built: binned SAH (default) or PLOC on the GPU, which must equal the CPU builder byte for byte
                quality = int(rng.integers(0, 2))
                for pl in (plugin, multi):
                    pl.set_tuning("BRT_BVH_QUALITY", quality)
                if quality:     # the tree brt_upload_scene builds on the GPU (brt_sah.hip) must be the CPU statement of the rule, byte for byte
                    cpu_nodes = brt.build_bvh_sah(b.models)
                    gpu_nodes, _ = plugin.build_bvh_sah(b.models)
                    ploc_checked += 1
                    ob = brt.Buffers(b.models, b.materials, cpu_nodes)
                    if cpu_nodes.tobytes() != gpu_nodes.tobytes():
                        raise AssertionError("GPU SAH tree differs from the CPU builder's")
                else:
                    cpu_nodes = brt.build_bvh(b.models)
                    gpu_nodes, _ = plugin.build_bvh(b.models)
                    ploc_checked += 1
                    ob = brt.Buffers(b.models, b.materials, cpu_nodes)
                    if cpu_nodes.tobytes() != gpu_nodes.tobytes():
                        raise AssertionError("GPU PLOC tree differs from the CPU builder's")
            else:
                ob = b
            try:
                got, stats = render(c, b)
            except brt.BrtError as e:   # a poisoned scene may be refused: then the oracle's validator must agree
                rejected += 1
                lines.append(f"case {case}: refused by the library ({e})")
                continue
            if b.bvh is None and quality and int(c["level"][0]["level"]) != 0:
                # the callee's SAH tree follows the camera (larger leaf pads for a camera further out): the oracle walks the CPU twin of
                # the tree the context says it built, and that tree must cover what this camera needs
                need = brt.tree_reach(b.models, c["camera"])[2]
                if stats["tree_reach"] != 0.0:
                    reach_checked += 1
                    ob = brt.Buffers(b.models, b.materials, brt.build_bvh_sah(b.models, stats["tree_reach"]))
                # (built for at least this camera's reach -- or the very same bytes, where every pad is at its clamp either way)
                if stats["tree_reach"] < need and ob.bvh.tobytes() != brt.build_bvh_sah(b.models, need).tobytes():
                    raise AssertionError(f"resident tree built for reach {stats['tree_reach']}, this camera needs {need}")
            want, cnt = oracle.render(ob, c["level"], c["camera"], c["window"], c["w"], c["h"], raster_rgba=c["raster"],
                                      raster_depth=c["depth"])
            bad = frames_differ(got, want)
            if bad:
                raise AssertionError(f"{bad} of {got.size} frame values differ")
            if c["mode"] not in ("multi", "simple", "parts"):
                # the same frame from the production instantiation (the hand-written walk loops and the half-sample jobs only run there):
                # frame + ray count; with forced half-sample jobs twice (the second frame runs in the order the first one measured)
                for rep in range(2 if "BRT_SPLIT_FORCE" in variant else 1):
                    got2, st2 = render(c, b, counters=False)
                    bad = frames_differ(got2, want)
                    if bad or st2["rays"] != cnt["rays"]:
                        raise AssertionError(f"production instantiation: {bad} of {got2.size} frame values differ, rays {st2['rays']} vs {cnt['rays']}")
                prod_checked += 1
                split_checked += 1 if "BRT_SPLIT_FORCE" in variant else 0
                hot_checked += 1 if st2.get("hot_records", 0) else 0
            # the callee's SAH tree pads its leaf boxes by less than the reference's 0.1 (brt_sah.h sah_model_pad): on well-conditioned scenes
            # without coincident spheres (exact ties are decided by the visiting order) the frame must also be the one of the caller's
            # 0.1-padded PLOC tree -- i.e. the tighter boxes culled nothing a ray is accepted by.  Well-conditioned = no sphere of radius
            # > 10: this generator's giant spheres (radius 20-2000, hundreds of them overlapping at coordinates in the thousands) are
            # beyond what the f32 sphere test resolves -- there the reference's OWN 0.1-padded tree disagrees with its brute-force loop
            # (profiles/r04/fuzz_soak.txt: PLOC tree vs one leaf of all spheres differ in up to half the pixels), so no two trees agree.
            if b.bvh is None and quality and not c["wild"] and len(b.models) <= 3000 and float(b.models["radius"].max()) <= 10.0 and \
                    len(np.unique(b.models.view(np.uint8).reshape(len(b.models), -1)[:, :16], axis=0)) == len(b.models):
                want_ref, _ = oracle.render(brt.Buffers(b.models, b.materials, brt.build_bvh(b.models)), c["level"], c["camera"], c["window"],
                                            c["w"], c["h"], raster_rgba=c["raster"], raster_depth=c["depth"])
                # (a far camera can be past what the reference's own 0.1-padded tree resolves -- cover scene: from ~600 units on it
                #  disagrees with the brute-force loop; the cross-tree check then has no reference to hold the callee's tree to)
                marginal = False
                if c["far_cam"] and len(b.models) <= 400:
                    truth, _ = oracle.render(brt.Buffers(b.models, b.materials, single_leaf_bvh(b.models)), c["level"], c["camera"], c["window"],
                                             c["w"], c["h"], raster_rgba=c["raster"], raster_depth=c["depth"])
                    marginal = frames_differ(want_ref, truth) != 0
                elif c["far_cam"]:
                    marginal = True
                if marginal:
                    ref_marginal += 1
                else:
                    tight_checked += 1
                    if frames_differ(got, want_ref):
                        raise AssertionError("frame in the callee's tight-box SAH tree differs from the frame in the 0.1-padded PLOC tree")
            keys = COUNTER_KEYS if c["mode"] != "parts" else ("rays",)
            if {k: stats[k] for k in keys} != {k: cnt[k] for k in keys}:
                raise AssertionError(f"counters differ: gpu { {k: stats[k] for k in keys} } oracle {cnt}")
            done += 1
            pixels += c["w"] * c["h"]
            rays += cnt["rays"]
        except AssertionError as e:
            fails += 1
            msg = (f"case {case} FAILED: {e} | variant {variant} | {len(b.models)} spheres, topo {c['topo']}, wild {c['wild']}, mode {c['mode']}/{c['n_parts']}, {c['w']}x{c['h']}, "
                   f"level {int(c['level'][0]['level']) if hasattr(c['level'], 'dtype') else c['level']}")
            print(msg, flush=True)
            lines.append(msg)
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            np.savez(os.path.join(ROOT, "gpurun_out", f"fuzz_fail_{case}.npz"), models=b.models.view(np.uint8),
                     materials=b.materials.view(np.uint8), bvh=(ob.bvh if b.bvh is None else b.bvh).view(np.uint8),
                     level=np.asarray(c["level"]).view(np.uint8), camera=np.asarray(c["camera"]).view(np.uint8),
                     window=np.asarray(c["window"]).view(np.uint8), size=np.array([c["w"], c["h"]]),
                     raster=np.zeros(0) if c["raster"] is None else c["raster"], depth=np.zeros(0) if c["depth"] is None else c["depth"])
    summary = (f"fuzz_parity seed {args.seed}: {done} cases bit-exact (frames + 5 counters), {fails} failed, {rejected} refused; "
               f"{pixels} pixels, {rays} rays; {ploc_checked} callee-built trees byte-identical CPU vs GPU, {tight_checked} tight-box SAH frames equal to the PLOC-tree frame ({ref_marginal} skipped: the reference's own tree differs from brute force there), {reach_checked} frames in a tree rebuilt for a far camera, {hot_checked} with the tree's records numbered by use, {prod_checked} cases also in the production instantiation (frame + rays), {split_checked} of them through half-sample jobs; "
               f"{time.time() - t_start:.0f} s")
    print(summary, flush=True)
    if args.log:
        os.makedirs(os.path.dirname(os.path.abspath(args.log)), exist_ok=True)
        with open(args.log, "a") as f:
            f.write("\n".join(lines + [summary]) + "\n")
    multi.close()
    plugin.close()
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
