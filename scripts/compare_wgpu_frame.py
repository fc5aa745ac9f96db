#!/usr/bin/env python3
"""One-command comparison of a frame rendered by the REFERENCE (bevyray's wgpu fragment pass) with this repo's reading
of the shader -- SURVEY.md 8(f) rank 4, for a machine that has a Rust toolchain and a Vulkan device (this build
environment has neither, so nothing here pins parity today; this makes the day it becomes possible a one-command job).

Checker tooling, not product: it renders the CPU oracle (oracle/, through tests/oracle_loader.py) under every
combination of the three readings that WGSL leaves to the implementation (the same switches as
tests/golden/policy_frames.npz) and says which one, if any, reproduces the dump.

Inputs: the byte buffers a ~20-line Bevy system dumps from `prepare_buffers` and the node:

    models.bin      n * 32 B          extract.rs:213-218        camera.bin   80 B   extract.rs:83-97
    materials.bin   n * 32 B          extract.rs:181-189        window.bin   16 B   extract.rs:56-61
    bvh.bin         (2n-1) * 48 B     extract.rs:229-237        level.bin    32 B   extract.rs:100-104 (or --level N)
    frame.bin       W*H*16 B RGBA32F (a Rgba32Float copy of post_process.destination), or W*H*4 B with --srgb8
                    (the default Rgba8UnormSrgb target: the shader's value, sRGB-encoded and quantised by the ROP)

    python scripts/compare_wgpu_frame.py --dir dump/ --width 400 --height 225 [--srgb8] [--raster-rgba r.bin --raster-depth d.bin]

Exit code 0 when one policy matches (bit for bit for f32 dumps; within --tolerance codes for sRGB8), 1 otherwise; prints,
per policy, the number of differing pixels and the first differing pixel with both values.  `--gpu` also renders the same
bytes through libbevyray_amd.so under the default policy and compares that frame too (needs an MI355X).
"""
import argparse
import itertools
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def srgb_encode(x):
    """Linear -> sRGB OETF as a Rgba8UnormSrgb target applies it on store (IEC 61966-2-1), then 8-bit quantisation."""
    x = np.clip(np.nan_to_num(x.astype(np.float64), nan=0.0), 0.0, 1.0)
    y = np.where(x <= 0.0031308, 12.92 * x, 1.055 * np.power(x, 1.0 / 2.4) - 0.055)
    return np.floor(y * 255.0 + 0.5).astype(np.int32)


def load_dump(args):
    import bevyray_amd as brt
    d = args.dir
    rd = lambda name, dt: np.fromfile(os.path.join(d, name), dtype=dt)
    models, materials, bvh = rd("models.bin", brt.MODEL_DTYPE), rd("materials.bin", brt.MATERIAL_DTYPE), rd("bvh.bin", brt.BVH_NODE_DTYPE)
    camera, window = rd("camera.bin", brt.CAMERA_DTYPE), rd("window.bin", brt.WINDOW_DTYPE)
    if args.level is not None:
        level = np.zeros(1, brt.LEVEL_DTYPE)
        level["level"] = args.level
    else:
        level = rd("level.bin", brt.LEVEL_DTYPE)
    assert len(camera) == 1 and len(window) == 1 and len(level) == 1, "camera.bin / window.bin / level.bin must hold one struct each"
    w, h = args.width, args.height
    if args.srgb8:
        frame = np.fromfile(os.path.join(d, args.frame), np.uint8).reshape(h, w, 4).astype(np.int32)
    else:
        frame = np.fromfile(os.path.join(d, args.frame), np.float32).reshape(h, w, 4)
    raster = np.fromfile(args.raster_rgba, np.float32).reshape(h, w, 4) if args.raster_rgba else None
    depth = np.fromfile(args.raster_depth, np.float32).reshape(h, w) if args.raster_depth else None
    return brt.Buffers(models, materials, bvh), level, camera, window, frame, raster, depth


def differing(got, want, srgb8, tol):
    """(number of differing pixels, first differing (x, y) or None)"""
    if srgb8:
        bad = np.any(np.abs(srgb_encode(got[..., :3]) - want[..., :3]) > tol, axis=-1)
    else:
        bad = np.any(got.view(np.uint32) != want.view(np.uint32), axis=-1)
    idx = np.argwhere(bad)
    return int(bad.sum()), (None if len(idx) == 0 else (int(idx[0][1]), int(idx[0][0])))


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--dir", required=True)
    ap.add_argument("--width", type=int, required=True)
    ap.add_argument("--height", type=int, required=True)
    ap.add_argument("--frame", default="frame.bin")
    ap.add_argument("--level", type=int, default=None)
    ap.add_argument("--srgb8", action="store_true")
    ap.add_argument("--tolerance", type=int, default=1, help="sRGB8 dumps: allowed difference in codes (the ROP's rounding is not specified)")
    ap.add_argument("--raster-rgba")
    ap.add_argument("--raster-depth")
    ap.add_argument("--gpu", action="store_true")
    args = ap.parse_args()
    import oracle_loader
    oracle = oracle_loader.load()
    b, level, camera, window, frame, raster, depth = load_dump(args)
    w, h = args.width, args.height
    print(f"{len(b.models)} spheres, {len(b.bvh)} BVH nodes, {w}x{h}, level {int(level['level'][0])}, "
          f"{int(camera['sample_count'][0])} spp, {int(camera['bounce_count'][0])} bounces, random_seed {float(window['random_seed'][0])!r}")
    matches = []
    for osc, mm, pw in itertools.product((False, True), ("minnum", "select"), ("mul", "exp2log2")):
        with oracle.policy(or_short_circuit=osc, minmax=mm, pow=pw):
            got, cnt = oracle.render(b, level, camera, window, w, h, raster_rgba=raster, raster_depth=depth)
        n, first = differing(got, frame, args.srgb8, args.tolerance)
        name = f"`||` {'short-circuits' if osc else 'evaluates both sides'}, min/max = {mm}, pow = {pw}"
        default = " [this repo's default policy]" if (not osc and mm == "minnum" and pw == "mul") else ""
        if n == 0:
            matches.append(name)
            print(f"MATCH     {name}{default}: every pixel equal ({cnt['rays']} rays)")
        else:
            x, y = first
            theirs = frame[y, x]
            ours = srgb_encode(got[y, x, :3]) if args.srgb8 else got[y, x]
            print(f"differs   {name}{default}: {n} of {w * h} pixels; first at (x {x}, y {y}): dump {theirs.tolist()} vs oracle {ours.tolist()}")
    if args.gpu:
        import bevyray_amd as brt
        with brt.RaytracePlugin([0]) as p:
            g = p.node.run(level, camera, window, w, h, buffers=b, raster_rgba=raster, raster_depth=depth)
        n, first = differing(g, frame, args.srgb8, args.tolerance)
        print(f"libbevyray_amd.so (default policy): {'every pixel equal' if n == 0 else f'{n} pixels differ, first at {first}'}")
    if not matches:
        print("no policy reproduces the dump: check the uv convention (pixel centres, raytrace.wgsl:20 / SURVEY 8(a) R2), the target format "
              "and DESIGN.md section 3 (division / sqrt / normalize lowering of the wgpu backend that rendered it)")
    return 0 if matches else 1


if __name__ == "__main__":
    sys.exit(main())
