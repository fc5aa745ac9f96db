#!/usr/bin/env python3
"""EXPERIMENT: like exp_tree_cost.py, but over several seeds of each scene (the greedy top-down build is chaotic: one scene's gain
is mostly luck): cover and RTIOW scenes seeds 1-6, the grid seeds 1-2, all 1920x1080, 64 spp, 8 bounces; mean kernel ms per variant."""
import os, subprocess, sys
CHILD = r'''
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import bevyray_amd as brt
out = []
for scene, seeds, camf in ((brt.SCENE_COVER, range(1, 7), brt.cover_camera), (brt.SCENE_RTIOW_FINAL, range(1, 7), brt.rtiow_camera), (brt.SCENE_STRESS_GRID, range(1, 3), brt.cover_camera)):
    w, h, spp, b = 1920, 1080, 64, 8
    lvl, cam, win = camf(w, h, spp, b)
    ms = []
    for seed in seeds:
        bb = brt.generate_scene(scene, seed)
        with brt.RaytracePlugin([0]) as p:
            p.set_tuning("BRT_CPU_BVH", 1)
            p.node.write_buffers(brt.Buffers(bb.models, bb.materials, None))
            o = p.alloc_frame(w, h)
            ks = []
            for i in range(6):
                p.node.run(lvl, cam, win, w, h, out=o)
                ks.append(p.node.last_stats["kernel_ms"])
            ms.append(min(ks[2:]))
    out.append(f"{np.mean(ms):7.3f} (" + " ".join(f"{x:.2f}" for x in ms) + ")")
print("  ".join(out), flush=True)
'''
lib = sys.argv[1]
for spec in sys.argv[2:]:          # "NAME=value NAME=value" (environment of the CPU-twin variant), or the older "kind C"
    if "=" in spec:
        extra = dict(kv.split("=", 1) for kv in spec.split())
    else:
        k, c = spec.split()
        extra = {"BRT_EXP_KIND": k, "BRT_EXP_C": c}
    env = dict(os.environ, BRT_LIB_PATH=os.path.abspath(lib), **extra)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print(f"{spec:36s}: cover / rtiow / grid mean ms  {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
