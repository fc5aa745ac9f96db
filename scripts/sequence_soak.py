#!/usr/bin/env python3
"""Soak test of what a context REMEMBERS between frames: random scenes walked from an LDS tile + L2, rendered as SEQUENCES of frames
through one context, every frame against the CPU oracle, bit for bit.

    python scripts/sequence_soak.py --sequences 200 --seed 3 [--log gpurun_out/sequence_soak.txt]

scripts/fuzz_parity.py draws independent frames; the state this round added lives ACROSS frames (brt_api.cpp):
  * the pair records and spheres re-numbered by a view's visit counts (apply_hot_order), counted again after a camera jump -- the
    permutations compose --, kept over the upload of a tree of the same shape (an animated scene), dropped when the shape changes;
  * the callee's tree rebuilt when the camera leaves its reach and again when it comes back (ensure_tree_reach);
  * dispatch order, half-sample jobs and tile costs that follow the view (prepass_order, update_tile_order).
A step of a sequence is one of: the same view again, a small camera move, a camera jump, a far camera (x 5 .. x 300 the scene's
extent, field of view narrowed to match), a few spheres moved and the scene uploaded again, a sphere added or removed, a frame in
the counting instantiation (all five counters), a frame at another size or sample count, a frame in another store format.
The oracle walks the CPU twin of the tree the context says it holds (`tree_reach` of brt_stats).  Needs an MI355X; the oracle is the
checker (test infrastructure).
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
from helpers import uniforms                   # noqa: E402

COUNTER_KEYS = ("rays", "node_pops", "interior_visits", "sphere_tests", "hits")


def frames_differ(got, want):
    got, want = np.asarray(got, np.float32), np.asarray(want, np.float32)
    same = got.view(np.uint32) == want.view(np.uint32)
    return int((~(same | (np.isnan(got) & np.isnan(want)))).sum())


def random_scene(rng):
    n = int(np.exp(rng.uniform(np.log(70), np.log(6000))))
    spread = float(rng.choice([3.0, 12.0, 40.0]))
    data = []
    ground = rng.random() < 0.5
    for i in range(n):
        if i == 0 and ground:
            data.append(((0.0, -1000.0, 0.0), 1000.0, brt.StandardMaterial(base_color=(0.5, 0.5, 0.5))))
            continue
        r = float(rng.uniform(0.05, 0.6)) * (spread / 12.0) ** 0.5
        pos = rng.uniform(-spread, spread, 3)
        if ground:
            pos[1] = r if rng.random() < 0.7 else float(rng.uniform(r, spread / 3))
        u = rng.random()
        mat = brt.StandardMaterial(base_color=tuple(float(x) for x in rng.random(3)), metallic=1.0 if u < 0.2 else 0.0,
                                   perceptual_roughness=float(rng.random()), ior=float(rng.uniform(1.1, 2.4)),
                                   specular_transmission=1.0 if 0.2 <= u < 0.35 else 0.0)
        data.append((tuple(float(x) for x in pos), r, mat))
    b = brt.prepare_buffers([(p, brt.RaytracedSphere(r), m) for p, r, m in data])
    return b, spread


def random_view(rng, spread, w, h, spp, bounces, far=False, near_to=None):
    if near_to is not None:                # a small move: a few percent of the distance
        pos, target, fov, seed = near_to
        d = float(np.linalg.norm(np.subtract(pos, target)))
        pos = tuple(float(x) for x in np.add(pos, rng.normal(0.0, 0.01 * d, 3)))
    else:
        dist = spread * (float(np.exp(rng.uniform(np.log(5.0), np.log(300.0)))) if far else float(rng.uniform(0.3, 2.5)))
        v = rng.standard_normal(3)
        v[1] = abs(v[1]) * 0.5 + 0.05
        pos = tuple(float(x) for x in dist * v / np.linalg.norm(v))
        target = tuple(float(x) for x in rng.uniform(-0.3, 0.3, 3) * spread)
        fov = float(rng.uniform(0.3, 1.2)) if not far else float(min(1.2, rng.uniform(1.0, 3.0) * spread / dist))
        seed = float(np.float32(rng.random()))
    lvl, cam, win = uniforms(w, h, spp=spp, bounces=bounces, pos=pos, target=target, fov=fov, seed=seed,
                             level=brt.Raytracing(int(rng.choice([1, 2, 3], p=[0.15, 0.15, 0.7]))))
    return (lvl, cam, win), (pos, target, fov, seed)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sequences", type=int, default=100)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--log", default=None)
    ap.add_argument("--seconds", type=float, default=1e9, help="stop after this much wall time")
    args = ap.parse_args()
    import torch
    oracle = oracle_loader.load()
    rng = np.random.default_rng(args.seed)
    t0 = t_progress = time.time()
    seqs = frames = fails = hot_frames = recounts = rebuilt = kept_numbering = dropped_numbering = counted = encoded = 0
    kinds = {}
    lines = []
    for seq in range(args.sequences):
        if time.time() - t0 > args.seconds:
            break
        b, spread = random_scene(rng)
        models, materials = b.models.copy(), b.materials
        n_pairs = len(models) - 1
        # the scene is walked from a tile of `tile` records + L2 (a scene of this size would otherwise sit in LDS whole)
        tile = int(rng.choice([0, int(rng.integers(1, 64)), int(rng.integers(64, max(65, n_pairs)))], p=[0.1, 0.3, 0.6]))
        ids = [0] if rng.random() < 0.8 else [0, 0]
        split = int(rng.integers(1, 300)) if rng.random() < 0.3 else 0
        w, h = int(rng.integers(48, 200)), int(rng.integers(32, 120))
        spp, bounces = int(rng.choice([32, 40, 64])), int(rng.integers(1, 7))
        view, vdesc = random_view(rng, spread, w, h, spp, bounces)
        raster = depth = None
        try:
            with brt.RaytracePlugin(ids) as p:
                p.set_tuning("BRT_FORCE_LDS_TOP", tile if tile else 100000)
                p.set_tuning("BRT_SPLIT_FORCE", split)
                upload = True
                for step in range(int(rng.integers(4, 10))):
                    kind = "first" if step == 0 else str(rng.choice(
                        ["same", "move", "jump", "far", "animate", "resize", "add", "counters", "encode"],
                        p=[0.2, 0.15, 0.15, 0.1, 0.15, 0.05, 0.05, 0.1, 0.05]))
                    flags = 0
                    if os.environ.get("SOAK_VERBOSE"):
                        print(f"seq {seq} step {step} {kind}: {len(models)} spheres tile {tile} devices {ids} split {split} {w}x{h} {spp} spp", flush=True)
                    if kind == "move":
                        view, vdesc = random_view(rng, spread, w, h, spp, bounces, near_to=vdesc)
                    elif kind in ("jump", "far"):
                        view, vdesc = random_view(rng, spread, w, h, spp, bounces, far=kind == "far")
                    elif kind == "animate":
                        for i in rng.integers(0, len(models), int(rng.integers(1, 6))):
                            if abs(float(models["radius"][i])) < 100.0:
                                models["position"][i] += rng.normal(0.0, float(rng.choice([1e-3, 0.05])), 3).astype(np.float32)
                        upload = True
                    elif kind == "add":
                        if rng.random() < 0.5 and len(models) > 80:
                            models = np.delete(models, int(rng.integers(1, len(models))))
                        else:
                            models = np.ascontiguousarray(np.concatenate([models, models[-1:]]), brt.MODEL_DTYPE)   # (concatenate drops the padding)
                            models["position"][-1] += np.float32(0.37)
                        upload = True
                    elif kind == "resize":
                        w, h = int(rng.integers(48, 200)), int(rng.integers(32, 120))
                        spp = int(rng.choice([8, 32, 64]))
                        view, vdesc = random_view(rng, spread, w, h, spp, bounces)
                    elif kind == "counters":
                        flags = brt.FLAG_COUNTERS
                    lvl, cam, win = view
                    if int(lvl[0]["level"]) in (1, 2):
                        if raster is None or raster.shape[:2] != (h, w):
                            raster = rng.random((h, w, 4), dtype=np.float32)
                            depth = rng.random((h, w), dtype=np.float32) * np.float32(0.05)
                        r_in, d_in = raster, depth
                    else:
                        r_in = d_in = None
                    nb = brt.Buffers(models, materials, None) if upload else None
                    had_hot = p.node.last_stats.get("hot_records", 0) if step else 0
                    if kind == "encode":               # the device entry point, frame stored in the colour target's own format
                        fmt = str(rng.choice(["srgb8", "unorm8", "f16"]))
                        flag = {"srgb8": brt.FLAG_OUT_RGBA8_UNORM_SRGB, "unorm8": brt.FLAG_OUT_RGBA8_UNORM, "f16": brt.FLAG_OUT_RGBA16F}[fmt]
                        if nb is not None:
                            p.node.write_buffers(nb)
                        d_frame = torch.zeros(h * w * brt.OUT_PIXEL_BYTES[flag], dtype=torch.uint8, device="cuda")
                        d_r = None if r_in is None else torch.from_numpy(r_in).cuda()
                        d_d = None if d_in is None else torch.from_numpy(d_in).cuda()
                        p.node.render_device(lvl, cam, win, w, h, d_frame.data_ptr(), d_raster_rgba=0 if d_r is None else d_r.data_ptr(),
                                             d_raster_depth=0 if d_d is None else d_d.data_ptr(), flags=flag)
                        got = d_frame.cpu().numpy()
                    else:
                        got = p.node.run(lvl, cam, win, w, h, buffers=nb, raster_rgba=r_in, raster_depth=d_in, flags=flags)
                    st = dict(p.node.last_stats)
                    upload = False
                    tree = brt.build_bvh_sah(models, st["tree_reach"])
                    need = brt.tree_reach(models, cam)[2]
                    if st["tree_reach"] < need and tree.tobytes() != brt.build_bvh_sah(models, need).tobytes():
                        raise AssertionError(f"step {step} ({kind}): resident tree built for reach {st['tree_reach']}, this camera needs {need}")
                    want, cnt = oracle.render(brt.Buffers(models, materials, tree), lvl, cam, win, w, h, raster_rgba=r_in, raster_depth=d_in)
                    if kind == "encode":
                        want_px = oracle.encode_frame(want, fmt)
                        bad = int((got != want_px.view(np.uint8).reshape(-1)).sum())
                        encoded += 1
                    else:
                        bad = frames_differ(got, want)
                    if bad or st["rays"] != cnt["rays"]:
                        raise AssertionError(f"step {step} ({kind}): {bad} frame values differ, rays {st['rays']} vs {cnt['rays']}, stats {st}")
                    if flags:
                        if {k: st[k] for k in COUNTER_KEYS} != cnt:
                            raise AssertionError(f"step {step} ({kind}): counters { {k: st[k] for k in COUNTER_KEYS} } vs {cnt}")
                        counted += 1
                    frames += 1
                    kinds[kind] = kinds.get(kind, 0) + 1
                    hot_frames += 1 if st["hot_records"] else 0
                    rebuilt += 1 if st["tree_rebuilt"] else 0
                    recounts += 1 if (step and st["prepass_ms"] > 0.0 and st["hot_records"]) else 0
                    if kind == "animate" and had_hot:
                        kept_numbering += 1 if st["hot_records"] else 0
                        dropped_numbering += 0 if st["hot_records"] else 1
            seqs += 1
        except AssertionError as e:
            fails += 1
            msg = (f"sequence {seq} FAILED: {e} | {len(models)} spheres, spread {spread}, tile {tile}, devices {ids}, split {split}, "
                   f"{w}x{h} {spp} spp {bounces} bounces")
            print(msg, flush=True)
            lines.append(msg)
        if time.time() - t_progress > 60:
            t_progress = time.time()
            print(f"... {seqs} sequences, {frames} frames, {fails} failed, {time.time() - t0:.0f} s", flush=True)
    summary = (f"sequence_soak seed {args.seed}: {seqs} sequences / {frames} frames bit-exact (frame + ray count), {fails} failed; "
               f"{hot_frames} frames with the records numbered by use, {recounts} of them counted again (camera moved on), {rebuilt} tree rebuilds "
               f"for a camera's reach, animated uploads: numbering kept {kept_numbering} / dropped {dropped_numbering} (tree changed shape), "
               f"{counted} frames in the counting instantiation (5 counters), {encoded} in an 8/16-bit store format; steps {kinds}; {time.time() - t0:.0f} s")
    print(summary, flush=True)
    if args.log:
        os.makedirs(os.path.dirname(os.path.abspath(args.log)), exist_ok=True)
        with open(args.log, "a") as f:
            f.write("\n".join(lines + [summary]) + "\n")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
