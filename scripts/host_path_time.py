#!/usr/bin/env python3
"""Wall time of the whole brt_render call (host buffers in and out, PCIe included) on the headline
frame: pageable numpy output vs a page-locked frame from brt_host_alloc (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bevyray_amd as brt
W, H, spp, bounces = 1920, 1080, 64, 8
b = brt.generate_scene(brt.SCENE_COVER, 1)
lvl, cam, win = brt.cover_camera(W, H, spp, bounces)
with brt.RaytracePlugin([0]) as p:
    p.node.write_buffers(b)
    pinned = p.alloc_frame(W, H)
    pageable = np.zeros((H, W, 4), np.float32)
    for name, out in (("pageable", pageable), ("page-locked", pinned), ("pageable", pageable), ("page-locked", pinned)):
        ts, ks = [], []
        for i in range(6):
            t0 = time.perf_counter()
            p.node.run(lvl, cam, win, W, H, out=out)
            ts.append((time.perf_counter() - t0) * 1e3)
            ks.append(p.node.last_stats["kernel_ms"])
        print(f"{name:12s} call wall {np.mean(ts[1:]):6.2f} ms (min {min(ts[1:]):6.2f})  kernel {np.mean(ks[1:]):6.2f} ms  library total_ms {p.node.last_stats['total_ms']:.2f}")
