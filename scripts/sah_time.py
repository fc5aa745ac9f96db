"""Kernel time of the GPU binned-SAH build (brt_sah.hip) against the CPU twin and the GPU PLOC build, on the benchmark
scenes and on random scenes of growing size.  Usage: python scripts/sah_time.py  (needs an MI355X)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bevyray_amd as brt

rng = np.random.default_rng(1)
cases = [("cover", brt.generate_scene(brt.SCENE_COVER, 1).models), ("rtiow", brt.generate_scene(brt.SCENE_RTIOW_FINAL, 1).models),
         ("grid 10k", brt.generate_scene(brt.SCENE_STRESS_GRID, 1).models)]
for n in (2000, 4096, 16384, 65536):
    m = np.zeros(n, brt.MODEL_DTYPE)
    m["position"] = rng.uniform(-50, 50, (n, 3)).astype(np.float32)
    m["radius"] = rng.uniform(0.05, 0.6, n).astype(np.float32)
    cases.append((f"random {n}", m))
with brt.RaytracePlugin([0]) as p:
    for name, m in cases:
        p.build_bvh_sah(m)
        gpu = sorted(p.build_bvh_sah(m)[1] for _ in range(9))
        t0 = time.perf_counter(); brt.build_bvh_sah(m); cpu = (time.perf_counter() - t0) * 1e3
        ploc = min(p.build_bvh(m)[1] for _ in range(3))
        # whole upload (build + read back + validate + re-encode + copies), every sphere moved so that dirty tracking does not skip it
        b = brt.Buffers(m.copy(), np.zeros(len(m), brt.MATERIAL_DTYPE), None)
        b.models["material_id"] = np.arange(len(m))
        ups = []
        for i in range(6):
            b.models["position"][:, 1] += np.float32(1e-3)
            t0 = time.perf_counter(); p.node.write_buffers(b); ups.append((time.perf_counter() - t0) * 1e3)
        print(f"{name:>14} ({len(m):6d} spheres): GPU SAH best {gpu[0]:.3f} median {gpu[len(gpu)//2]:.3f} ms | CPU twin {cpu:.2f} ms | "
              f"GPU PLOC {ploc:.3f} ms | brt_upload_scene wall (median of 5) {sorted(ups[1:])[2]:.3f} ms", flush=True)
