#!/usr/bin/env python3
"""Wall time of brt_upload_scene for a scene that changes every frame (the reference rebuilds and re-uploads every
frame, extract.rs:280-337 / pipeline.rs:136-138): with the caller's BVH, and with the callee building it (GPU PLOC)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bevyray_amd as brt

with brt.RaytracePlugin([0]) as p:
    for kind, name in ((brt.SCENE_COVER, "cover 506"), (brt.SCENE_STRESS_GRID, "grid 10004")):
        b = brt.generate_scene(kind, 1)
        for with_bvh in (True, False):
            ts = []
            for i in range(12):
                m = b.models.copy()
                m["position"][:, 1] += np.float32(1e-3 * (i + 1))        # every sphere moves: nothing is reusable
                bb = brt.Buffers(m, b.materials, brt.build_bvh(m) if with_bvh else None)
                t0 = time.perf_counter()
                p.node.write_buffers(bb)
                ts.append((time.perf_counter() - t0) * 1e3)
            print(f"{name:11s} {'caller BVH ' if with_bvh else 'callee BVH '} upload {np.median(ts[2:]):7.3f} ms (min {min(ts[2:]):.3f})", flush=True)
