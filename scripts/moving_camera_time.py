#!/usr/bin/env python3
"""Kernel time per frame of the headline workload while the camera moves (the reference's demo is a fly-camera app, src/main.rs:40):
orbits of several speeds and a dolly, with and without the neighbourhood re-ranking of the one-frame-old tile costs
(BRT_LPT_DILATE), against the static view.  usage: moving_camera_time.py [scene] [frames]   (needs an MI355X)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bevyray_amd as brt

scene = int(sys.argv[1]) if len(sys.argv) > 1 else 0
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 24
W, H, spp, bounces = 1920, 1080, 64, 8
b = brt.generate_scene(scene, 1)


def cam_at(pos, target=(0.0, 0.0, 0.0)):
    return brt.CameraExtract.extract_component(brt.RaytracedCamera(level=brt.Raytracing.Pure, sample_count=spp, bounces=bounces),
                                               brt.Transform(tuple(float(x) for x in pos), target, (0.0, 1.0, 0.0)),
                                               brt.PerspectiveProjection(fov=0.4, aspect_ratio=W / H, near=0.1, far=1000.0))


def orbit(deg):
    for i in range(1, frames + 1):
        a = np.deg2rad(deg * i)
        yield (13.0 * np.cos(a) - 3.0 * np.sin(a), 2.0, 13.0 * np.sin(a) + 3.0 * np.cos(a)), (0.0, 0.0, 0.0)


def dolly(step):
    for i in range(1, frames + 1):
        k = 1.0 - step * i
        yield (13.0 * k, 2.0, 3.0 * k), (0.0, 0.0, 0.0)


def pan(deg):            # the camera stands, the target moves sideways
    for i in range(1, frames + 1):
        yield (13.0, 2.0, 3.0), (0.0, 0.0, 0.3 * deg * i)


with brt.RaytracePlugin([0]) as p:
    p.node.write_buffers(brt.Buffers(b.models, b.materials, None))
    out = p.alloc_frame(W, H)
    lvl, cam = cam_at((13.0, 2.0, 3.0))
    ks = []
    for i in range(10):
        p.node.run(lvl, cam, brt.WindowExtract.extract_component(H, 0.1 + 0.05 * i), W, H, out=out)
        ks.append(p.node.last_stats["kernel_ms"])
    print(f"static view                        median {np.median(ks[3:]):7.3f} ms", flush=True)
    for name, path in (("orbit 0.1 deg/frame", lambda: orbit(0.1)), ("orbit 0.5 deg/frame", lambda: orbit(0.5)), ("orbit 2 deg/frame", lambda: orbit(2.0)),
                       ("dolly 0.5 %/frame", lambda: dolly(0.005)), ("pan 0.3 units x 0.5/frame", lambda: pan(0.5))):
        for dilate in (1, 0):
            p.set_tuning("BRT_LPT_DILATE", dilate)
            p.node.run(lvl, cam, brt.WindowExtract.extract_component(H, 0.5), W, H, out=out)      # (a knob forgets the history: first frame + measure)
            p.node.run(lvl, cam, brt.WindowExtract.extract_component(H, 0.5), W, H, out=out)
            ks, pre, var = [], 0, set()
            for i, (pos, tgt) in enumerate(path()):
                l, c = cam_at(pos, tgt)
                p.node.run(l, c, brt.WindowExtract.extract_component(H, 0.03 + 0.04 * i), W, H, out=out)
                s = p.node.last_stats
                ks.append(s["kernel_ms"] + s["prepass_ms"])
                pre += int(s["prepass_ms"] > 0)
                var.add(s["kernel_variant"])
            print(f"{name:28s} dilate {dilate}  median {np.median(ks):7.3f} ms  max {max(ks):7.3f}  pre-passes {pre}  variants {sorted(var)}", flush=True)
