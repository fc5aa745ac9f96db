#!/usr/bin/env python3
"""EXPERIMENT (kept for the record, docs/experiments.md): kernel time of configs 2, 3, 5 in trees built by a CPU-twin variant whose
split-cost weight is chosen by BRT_EXP_KIND / BRT_EXP_C (a build of brt_host.cpp with the experiment's exp_f; BRT_CPU_BVH=1 makes the
upload use the CPU builder).  usage: exp_tree_cost.py lib.so "kind C" ..."""
import os, subprocess, sys
CHILD = r'''
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, zlib
import bevyray_amd as brt
out = []
for scene, (w, h, spp, b), camf in ((brt.SCENE_COVER, (1920, 1080, 64, 8), brt.cover_camera), (brt.SCENE_RTIOW_FINAL, (1920, 1080, 256, 50), brt.rtiow_camera), (brt.SCENE_STRESS_GRID, (1920, 1080, 64, 8), brt.cover_camera)):
    bb = brt.generate_scene(scene, 1)
    lvl, cam, win = camf(w, h, spp, b)
    with brt.RaytracePlugin([0]) as p:
        p.set_tuning("BRT_CPU_BVH", 1)
        p.node.write_buffers(brt.Buffers(bb.models, bb.materials, None))
        o = p.alloc_frame(w, h)
        ks = []
        for i in range(5 if scene == brt.SCENE_RTIOW_FINAL else 8):
            p.node.run(lvl, cam, win, w, h, out=o)
            ks.append(p.node.last_stats["kernel_ms"])
        out.append(f"{min(ks[2:]):7.3f}")
print(" ".join(out), flush=True)
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
    print(f"{spec:36s}: config 2 / 3 / 5 kernel ms  {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
