#!/usr/bin/env python3
"""Kernel time of one workload in the caller's PLOC tree and in the callee-built SAH tree, for A/B runs of two library builds on one box
(BRT_LIB_PATH selects the build).  usage: ab_trees.py <scene> <w> <h> <spp> <bounces> ["K=V ..."]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bevyray_amd as brt

scene, w, h, spp, bounces = (int(x) for x in sys.argv[1:6])
knobs = dict(kv.split("=", 1) for kv in sys.argv[6].split()) if len(sys.argv) > 6 and sys.argv[6] else {}
reps = int(os.environ.get("SWEEP_REPS", "8"))
b = brt.generate_scene(scene, 1)
lvl, cam, win = (brt.rtiow_camera if scene == 1 else brt.cover_camera)(w, h, spp, bounces)
with brt.RaytracePlugin([0]) as p:
    for k, v in knobs.items():
        p.set_tuning(k, int(v))
    out = p.alloc_frame(w, h)
    for name, bufs in (("caller PLOC tree", b), ("callee SAH tree", brt.Buffers(b.models, b.materials, None))):
        p.node.write_buffers(bufs)
        ks = []
        for i in range(reps + 2):
            p.node.run(lvl, cam, win, w, h, out=out)
            ks.append(p.node.last_stats["kernel_ms"])
        s = p.node.last_stats
        print(f"{os.path.basename(os.environ.get('BRT_LIB_PATH', 'libbevyray_amd.so')):24s} {name:18s} best {min(ks[2:]):7.3f} median {np.median(ks[2:]):7.3f} ms  "
              f"mode {s['scene_in_lds']} lds {s['lds_bytes']} rays {s['rays']}", flush=True)
