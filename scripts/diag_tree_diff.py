#!/usr/bin/env python3
"""Which pixels differ between the caller's PLOC tree and the tree the callee builds (bvh = None)?  GPU part: both frames of a config,
the differing pixels to gpurun_out/tree_diff_<scene>.json.  usage: diag_tree_diff.py scene w h spp bounces"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bevyray_amd as brt

def main():
    scene, w, h, spp, bounces = (int(x) for x in sys.argv[1:6])
    b = brt.generate_scene(scene, 1)
    lvl, cam, win = (brt.rtiow_camera if scene == 1 else brt.cover_camera)(w, h, spp, bounces)
    with brt.RaytracePlugin([0]) as p:
        f1 = p.node.run(lvl, cam, win, w, h, buffers=b).copy()
        r1 = p.node.last_stats["rays"]
        f2 = p.node.run(lvl, cam, win, w, h, buffers=brt.Buffers(b.models, b.materials, None)).copy()
        r2 = p.node.last_stats["rays"]
    bad = np.argwhere((f1.view(np.uint32) != f2.view(np.uint32)).any(axis=2))
    out = {"scene": scene, "w": w, "h": h, "spp": spp, "bounces": bounces, "rays_caller_tree": int(r1), "rays_callee_tree": int(r2),
           "differing_pixels_yx": bad.tolist()}
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open(f"gpurun_out/tree_diff_{scene}.json", "w"))
    print(out)

if __name__ == "__main__":
    main()
