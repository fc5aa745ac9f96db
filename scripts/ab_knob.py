#!/usr/bin/env python3
"""A/B of one tuning knob over the BASELINE configs in ONE process: kernel ms (best of N) per config and value.
usage: ab_knob.py KNOB v0 v1 ... [--configs 2 3 5] [--reps 4]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bevyray_amd as brt

CFG = {"1": (brt.SCENE_COVER, 400, 225, 1, 4, "cover"), "2": (brt.SCENE_COVER, 1920, 1080, 64, 8, "cover"),
       "3": (brt.SCENE_RTIOW_FINAL, 1920, 1080, 256, 50, "rtiow"), "5": (brt.SCENE_STRESS_GRID, 1920, 1080, 64, 8, "cover"),
       "t": (brt.SCENE_COVER, 64, 36, 64, 8, "cover"), "p1": (brt.SCENE_COVER, 1, 1, 2048, 8, "cover"), "p2": (brt.SCENE_COVER, 2, 1, 2048, 8, "cover"),
       "p8": (brt.SCENE_COVER, 8, 1, 1024, 8, "cover")}

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("knob"); ap.add_argument("values", type=int, nargs="+")
    ap.add_argument("--configs", nargs="*", default=["2", "3"]); ap.add_argument("--reps", type=int, default=4)
    ap.add_argument("--callee-tree", action="store_true", help="upload without a BVH (the callee's SAH tree) instead of the scene's PLOC tree")
    ap.add_argument("--tunable", action="store_true", help="every value in the knobs-live instantiation (else the default value runs the folded one)")
    a = ap.parse_args()
    with brt.RaytracePlugin([0]) as p:
        for c in a.configs:
            kind, w, h, spp, bounces, cam = CFG[c]
            b = brt.generate_scene(kind, 1)
            lvl, camx, win = (brt.rtiow_camera if cam == "rtiow" else brt.cover_camera)(w, h, spp, bounces)
            p.node.write_buffers(brt.Buffers(b.models, b.materials, None) if a.callee_tree else b)
            out = p.alloc_frame(w, h)
            ref = None
            for rnd in range(2):
                for v in a.values:
                    with p.tuning(**({a.knob: v, "BRT_TUNABLE": 1} if a.tunable else {a.knob: v})):
                        ks = []
                        for _ in range(a.reps + 2):
                            p.node.run(lvl, camx, win, w, h, out=out)
                            ks.append(p.node.last_stats["kernel_ms"])
                    if ref is None: ref = out.copy()
                    same = bool(np.array_equal(ref.view(np.uint32), out.view(np.uint32)))
                    if rnd == 1:
                        print(f"config {c}  {a.knob}={v:<4d} best {min(ks[2:]):8.3f} ms  median {np.median(ks[2:]):8.3f} ms  same_pixels {same}", flush=True)

if __name__ == "__main__":
    main()
