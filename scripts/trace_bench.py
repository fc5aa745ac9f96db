#!/usr/bin/env python3
"""Trace-only microbenchmark (diagnostic): records real rays of the headline frame and times
walking them (a) 64 per wave without refill, (b) with in-loop lane refill at several thresholds."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevyray_amd as brt
from bevyray_amd import _lib
W, H, spp, bounces = 1920, 1080, 8, 8
b = brt.generate_scene(brt.SCENE_COVER, 1)
lvl, cam, win = brt.cover_camera(W, H, spp, bounces)
lib = _lib.load()
with brt.RaytracePlugin([0]) as p:
    p.node.write_buffers(b)
    for refill in (64, 32, 16, 8, 4):
        out = (C.c_double * 6)()
        _lib.check(lib.brt_debug_trace_bench(p._ctx, cam.ctypes.data, win.ctypes.data, W, H, 40_000_000, refill, out), p._ctx)
        n, ms0, ms1, it, ln, eq = list(out)
        print(f"refill_min {refill:2d}: {int(n)} rays | no refill {ms0:7.3f} ms ({n/ms0/1e3:8.1f} Mrays/s) | in-loop refill {ms1:7.3f} ms "
              f"({n/ms1/1e3:8.1f} Mrays/s) x{ms0/ms1:4.2f} | walking lanes per iteration {ln/max(1,it):5.1f} | same results {bool(eq)}", flush=True)
