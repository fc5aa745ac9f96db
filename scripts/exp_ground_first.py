#!/usr/bin/env python3
"""EXPERIMENT: the ground sphere's leaf is a child of the root in the callee's tree (slot 1: pushed first, popped LAST, so every lane
tests it at the end of its own walk, at a time of its own).  Swapped into slot 2 it is popped FIRST: all lanes of a wave test it together
in one full-width leaf step right after the root.  The twin of the callee's tree is uploaded as a caller's BVH, as built and with the
root's two children swapped; kernel ms of configs 2, 3, 5 (and that the frame is the same)."""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bevyray_amd as brt

for scene, (w, h, spp, b), camf, name in ((brt.SCENE_COVER, (1920, 1080, 64, 8), brt.cover_camera, "config 2"),
                                          (brt.SCENE_RTIOW_FINAL, (1920, 1080, 256, 50), brt.rtiow_camera, "config 3"),
                                          (brt.SCENE_STRESS_GRID, (1920, 1080, 64, 8), brt.cover_camera, "config 5")):
    bb = brt.generate_scene(scene, 1)
    tree = brt.build_bvh_sah(bb.models)
    a = int(tree[0]["index"])
    swapped = tree.view(np.uint8).reshape(len(tree), -1).copy()
    swapped[[a, a + 1]] = swapped[[a + 1, a]]
    swapped = swapped.reshape(-1).view(brt.BVH_NODE_DTYPE)
    lvl, cam, win = camf(w, h, spp, b)
    leaf_first = tree.view(np.uint8).reshape(len(tree), -1).copy()          # every node: a leaf child beside an interior one goes first
    for i in range(len(tree)):
        if tree[i]["model_count"] == 0:
            c = int(tree[i]["index"])
            if tree[c]["model_count"] > 0 and tree[c + 1]["model_count"] == 0:
                leaf_first[[c, c + 1]] = leaf_first[[c + 1, c]]
    leaf_first = leaf_first.reshape(-1).view(brt.BVH_NODE_DTYPE)
    leaf_last = swapped.view(np.uint8).reshape(len(tree), -1).copy()        # ... goes last (but the ground first)
    for i in range(1, len(tree)):
        if tree[i]["model_count"] == 0:
            c = int(tree[i]["index"])
            if tree[c]["model_count"] == 0 and tree[c + 1]["model_count"] > 0:
                leaf_last[[c, c + 1]] = leaf_last[[c + 1, c]]
    leaf_last = leaf_last.reshape(-1).view(brt.BVH_NODE_DTYPE)
    for rnd in range(2):
        for tag, t in (("callee's tree as built", tree), ("root's children swapped", swapped), ("leaf child first, all nodes", leaf_first), ("ground first, leaves last", leaf_last)):
            with brt.RaytracePlugin([0]) as p:
                p.node.write_buffers(brt.Buffers(bb.models, bb.materials, t))
                o = p.alloc_frame(w, h)
                ks = []
                for i in range(5 if scene == brt.SCENE_RTIOW_FINAL else 8):
                    p.node.run(lvl, cam, win, w, h, out=o)
                    ks.append(p.node.last_stats["kernel_ms"])
                if rnd == 1:
                    print(f"{name} {tag:26s}: best {min(ks[2:]):7.3f} ms  median {float(np.median(ks[2:])):7.3f}  rays {p.node.last_stats['rays']}  crc {zlib.crc32(o.tobytes()):08x}", flush=True)
