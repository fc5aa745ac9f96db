#!/usr/bin/env python3
"""Renders a few frames of the headline workload (cover scene 1920x1080, 64 spp, 8 bounces) through the C ABI
without importing torch: the program bench.py, scripts/pmc_sets.py and scripts/collect_profiles.sh put behind `rocprofv3 --pmc ... --`.
The first frame runs in raster order and measures the tile costs, the later ones use the order built from
them (brt_api.cpp); counter readers take the LAST dispatch of k_trace_persistent."""
import os
import sys

os.environ.setdefault("BRT_NO_TORCH", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevyray_amd as brt  # noqa: E402


def main():
    scene = int(os.environ.get("BRT_PMC_SCENE", str(brt.SCENE_COVER)))
    w, h, spp, bounces = 1920, 1080, 64, 8
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    b = brt.generate_scene(scene, 1)
    lvl, cam, win = brt.cover_camera(w, h, spp, bounces)
    with brt.RaytracePlugin([0]) as p:
        p.node.write_buffers(b)
        out = p.alloc_frame(w, h)
        for i in range(frames):
            p.node.run(lvl, cam, win, w, h, out=out)
            s = p.node.last_stats
            print(f"frame {i}: kernel {s['kernel_ms']:.3f} ms, {s['rays']} rays", flush=True)


if __name__ == "__main__":
    main()
