#!/usr/bin/env python3
"""The last pass of the callee's SAH rule ("giant leaves first", brt_sah.h) on scenes it was NOT shaped on: the
cover scene with its ground of radius 1000, with a ground of radius 50, and with no ground at all; seeds 1-3; 1920x1080, 64 spp,
8 bounces; every frame is checked against the frame in the caller's PLOC tree (same pixels).  A/B of library builds:
    python scripts/exp_tree_rules.py ab/libgiant0.so bevyray_amd/libbevyray_amd.so      (giant0: -DBRT_GIANT_RULE=0, round 5's "radius > 100")"""
import os, subprocess, sys
CHILD = r'''
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import bevyray_amd as brt
w, h, spp, bn = 1920, 1080, 64, 8
lvl, cam, win = brt.cover_camera(w, h, spp, bn)
out = []
for ground in ("r1000", "r50", "none"):
    ms, same = [], True
    for seed in (1, 2, 3):
        b = brt.generate_scene(brt.SCENE_COVER, seed)
        models = b.models.copy()
        mats = b.materials
        if ground == "r50":
            models[0]["position"] = (0.0, -50.0, 0.0); models[0]["radius"] = 50.0
        elif ground == "none":
            models = models[1:].copy()
        with brt.RaytracePlugin([0]) as p:
            o = p.alloc_frame(w, h)
            p.node.write_buffers(brt.Buffers(models, mats, brt.build_bvh(models)))
            p.node.run(lvl, cam, win, w, h, out=o)
            ref = o.copy()
            p.node.write_buffers(brt.Buffers(models, mats, None))
            ks = []
            for i in range(6):
                p.node.run(lvl, cam, win, w, h, out=o)
                ks.append(p.node.last_stats["kernel_ms"])
            same = same and bool(np.array_equal(o.view(np.uint32), ref.view(np.uint32)))
            ms.append(min(ks[2:]))
    out.append(f"{ground}: {np.mean(ms):6.3f} (" + " ".join(f"{x:.2f}" for x in ms) + f") same pixels as the caller's tree: {same}")
print("   ".join(out), flush=True)
'''
for lib in sys.argv[1:]:
    env = dict(os.environ, BRT_LIB_PATH=os.path.abspath(lib))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print(f"{lib:32s} {r.stdout.strip() or r.stderr.strip()[-400:]}", flush=True)
