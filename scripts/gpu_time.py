#!/usr/bin/env python3
"""Quick timing of the trace kernel on one GPU (development aid; bench.py is the contract)."""
import argparse
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bevyray_amd as brt

ap = argparse.ArgumentParser()
ap.add_argument("--scene", type=int, default=0)
ap.add_argument("--w", type=int, default=1920)
ap.add_argument("--h", type=int, default=1080)
ap.add_argument("--spp", type=int, default=64)
ap.add_argument("--bounces", type=int, default=8)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--flags", type=int, default=0)
ap.add_argument("--camera", default="cover")
a = ap.parse_args()
b = brt.generate_scene(a.scene, 1)
lvl, cam, win = (brt.rtiow_camera if a.camera == "rtiow" else brt.cover_camera)(a.w, a.h, a.spp, a.bounces)
with brt.RaytracePlugin([0]) as p:
    # the tree the callee builds (what bench.py times); BRT_CALLER_TREE=1: the caller's PLOC tree
    p.node.write_buffers(b if os.environ.get("BRT_CALLER_TREE") == "1" else brt.Buffers(b.models, b.materials, None))
    for i in range(a.reps):
        f = p.node.run(lvl, cam, win, a.w, a.h, flags=a.flags)
        s = p.node.last_stats
        print(f"rep {i}: kernel {s['kernel_ms']:.2f} ms  rays {s['rays']}  {s['rays']/s['kernel_ms']/1e3:.1f} Mrays/s  "
              f"lds {s['lds_bytes']} in_lds {s['scene_in_lds']} grid {s['n_workgroups']}x{s['threads_per_workgroup']} total {s['total_ms']:.1f} ms", flush=True)
    p.node.run(lvl, cam, win, a.w, a.h, flags=1)
    print({k: v for k, v in p.node.last_stats.items()})
    prof = p.debug_profile()
    print("   timeline", getattr(p, "last_timeline", None))
    print("   order", getattr(p, "last_order_meta", None))
    for k, (ex, ln) in prof.items():
        if ex:
            print(f"   section {k:9s} executions {ex:12d}  lanes {ln:14d}  avg lanes/exec {ln/ex:6.2f}")
