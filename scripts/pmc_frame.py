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
    workload = os.environ.get("BRT_PMC_WORKLOAD", "cover_1920x1080_64spp_8b")
    if workload == "rtiow_3840x2160_1024spp_8b":      # BASELINE.json configs[3], the whole frame on one GPU
        scene, (w, h, spp, bounces), cam_fn, dflt_frames = brt.SCENE_RTIOW_FINAL, (3840, 2160, 1024, 8), brt.rtiow_camera, 2
    elif workload == "rtiow_1920x1080_256spp_50b":    # BASELINE.json configs[2]
        scene, (w, h, spp, bounces), cam_fn, dflt_frames = brt.SCENE_RTIOW_FINAL, (1920, 1080, 256, 50), brt.rtiow_camera, 3
    elif workload == "grid10k_1920x1080_64spp_8b":    # BASELINE.json configs[4]
        scene, (w, h, spp, bounces), cam_fn, dflt_frames = brt.SCENE_STRESS_GRID, (1920, 1080, 64, 8), brt.cover_camera, 3
    else:
        scene = int(os.environ.get("BRT_PMC_SCENE", str(brt.SCENE_COVER)))
        (w, h, spp, bounces), cam_fn, dflt_frames = (1920, 1080, 64, 8), brt.cover_camera, 3
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else dflt_frames
    b = brt.generate_scene(scene, 1)
    lvl, cam, win = cam_fn(w, h, spp, bounces)
    with brt.RaytracePlugin([0]) as p:
        p.node.write_buffers(brt.Buffers(b.models, b.materials, None))   # as bench.py: the callee builds the tree
        out = p.alloc_frame(w, h)
        for i in range(frames):
            p.node.run(lvl, cam, win, w, h, out=out)
            s = p.node.last_stats
            print(f"frame {i}: kernel {s['kernel_ms']:.3f} ms, {s['rays']} rays", flush=True)


if __name__ == "__main__":
    main()
