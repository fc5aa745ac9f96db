#!/usr/bin/env python3
"""Which records of the tree does a view walk?  Interior visits per node (the oracle's diagnostic counter) of a sampled config-5 frame,
and what share of them a tile of K records in LDS would serve when the records are ordered (a) breadth first, as now, (b) by how
often this view visits them.  usage: visit_hist.py [scene] [w] [h] [spp]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bevyray_amd as brt
import oracle_loader

scene = int(sys.argv[1]) if len(sys.argv) > 1 else 2
w, h, spp = (int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (480, 270, 16)
o = oracle_loader.load()
b = brt.generate_scene(scene, 1)
tree = brt.build_bvh_sah(b.models)
lvl, cam, win = brt.cover_camera(w, h, spp, 8)
counts = np.zeros(len(tree), np.uint64)
o.lib.oracle_set_visit_counts.argtypes = [C.c_void_p]
o.lib.oracle_set_visit_counts(counts.ctypes.data)
sph = np.zeros(len(b.models), np.uint64)
o.lib.oracle_set_sphere_counts.argtypes = [C.c_void_p]
o.lib.oracle_set_sphere_counts(sph.ctypes.data)
_, cnt = o.render(brt.Buffers(b.models, b.materials, tree), lvl, cam, win, w, h)
o.lib.oracle_set_visit_counts(None)
o.lib.oracle_set_sphere_counts(None)
interior = np.flatnonzero(tree["model_count"] == 0)
# breadth-first rank of the interior nodes (what the encoder numbers pair records by)
order, q = [], [0]
while q:
    nxt = []
    for n in q:
        if tree[n]["model_count"] == 0:
            order.append(n)
            nxt += [int(tree[n]["index"]), int(tree[n]["index"]) + 1]
    q = nxt
order = np.array(order)
c_bfs = counts[order].astype(np.float64)
total = c_bfs.sum()
assert total == cnt["interior_visits"]
c_hot = np.sort(c_bfs)[::-1]
print(f"scene {scene}: {len(order)} pair records, {cnt['rays']} rays, {total / cnt['rays']:.2f} interior visits per ray; records never visited: {(c_bfs == 0).sum()}")
print(f"{'records in LDS':>15s} {'breadth-first':>14s} {'by visits':>10s}   (share of the interior visits served from LDS; the rest per ray)")
for k in (400, 879, 1240, 1550, 1774, 2400, 3100, 4096, len(order)):
    a, bb = c_bfs[:k].sum() / total, c_hot[:k].sum() / total
    print(f"{k:15d} {a:14.3f} {bb:10.3f}   global steps per ray: {(1 - a) * total / cnt['rays']:.2f} -> {(1 - bb) * total / cnt['rays']:.2f}")

s_hot = np.sort(sph.astype(np.float64))[::-1]
print(f"sphere tests per ray {s_hot.sum() / cnt['rays']:.2f}; spheres never tested: {(s_hot == 0).sum()} of {len(s_hot)}")
print(f"{'spheres in LDS':>15s} {'by tests':>10s}   (share of the sphere tests served from LDS)")
for k in (256, 512, 1024, 1536, 2048, 3072, 4096, 6144):
    print(f"{k:15d} {s_hot[:k].sum() / s_hot.sum():10.3f}")
