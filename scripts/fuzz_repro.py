#!/usr/bin/env python3
"""Re-runs one dumped fuzz case (gpurun_out/fuzz_fail_<n>.npz from scripts/fuzz_parity.py) and prints the differences."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bevyray_amd as brt
import oracle_loader

def main():
    oracle = oracle_loader.load()
    plugin = brt.RaytracePlugin([0])
    for path in sys.argv[1:]:
        z = np.load(path)
        b = brt.Buffers(z["models"].view(brt.MODEL_DTYPE), z["materials"].view(brt.MATERIAL_DTYPE), z["bvh"].view(brt.BVH_NODE_DTYPE))
        lvl, cam, win = z["level"].view(brt.LEVEL_DTYPE), z["camera"].view(brt.CAMERA_DTYPE), z["window"].view(brt.WINDOW_DTYPE)
        w, h = (int(x) for x in z["size"])
        raster = z["raster"] if z["raster"].size else None
        depth = z["depth"] if z["depth"].size else None
        print(path, len(b.models), "spheres", w, h, cam, win, lvl)
        cpu_nodes = brt.build_bvh(b.models)
        gpu_nodes, _ = plugin.build_bvh(b.models)
        if cpu_nodes.tobytes() != gpu_nodes.tobytes():
            d = np.flatnonzero((cpu_nodes.view(np.uint32).reshape(len(cpu_nodes), -1) != gpu_nodes.view(np.uint32).reshape(len(gpu_nodes), -1)).any(1))
            print(" PLOC differs at nodes", d[:10], "of", len(cpu_nodes))
            for k in d[:3]:
                print("  cpu", cpu_nodes[k]); print("  gpu", gpu_nodes[k])
            for m in b.models:
                if not np.isfinite(m["position"]).all() or not np.isfinite(m["radius"]):
                    print("  non-finite model", m["position"], m["radius"])
        got = plugin.node.run(lvl, cam, win, w, h, buffers=b, raster_rgba=raster, raster_depth=depth, flags=brt.FLAG_COUNTERS)
        want, cnt = oracle.render(b, lvl, cam, win, w, h, raster_rgba=raster, raster_depth=depth)
        bad = (got.view(np.uint32) != want.view(np.uint32)) & ~(np.isnan(got) & np.isnan(want))
        print(" frame values differing:", int(bad.sum()), "of", bad.size)
        for y, x, c in np.argwhere(bad)[:4]:
            print("  ", y, x, "gpu", got[y, x], "oracle", want[y, x])
    plugin.close()

if __name__ == "__main__":
    main()
