#!/usr/bin/env python3
"""A/B of tuning knobs in ONE process (brt_set_tuning; names are those of the BRT_* variables): for every setting, a few frames of one
workload, the best and the median kernel time.  usage: sweep_env.py <scene> <w> <h> <spp> <bounces> "K=V K2=V2" ..."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bevyray_amd as brt


def main():
    scene, w, h, spp, bounces = (int(x) for x in sys.argv[1:6])
    settings = sys.argv[6:] or [""]
    reps = int(os.environ.get("SWEEP_REPS", "6"))
    b = brt.generate_scene(scene, 1)
    lvl, cam, win = brt.cover_camera(w, h, spp, bounces)
    with brt.RaytracePlugin([0]) as p:
        # SWEEP_CALLEE_TREE=1: upload without a BVH (the callee builds its SAH tree on the GPU: what bench.py times)
        p.node.write_buffers(brt.Buffers(b.models, b.materials, None) if os.environ.get("SWEEP_CALLEE_TREE") == "1" else b)
        out = p.alloc_frame(w, h)
        ref = None
        for rnd in range(2):          # two passes over the settings: the second is the one to read (clocks warm)
            for st in settings:
                env = dict(kv.split("=", 1) for kv in st.split()) if st else {}
                with p.tuning(**{k: int(v) for k, v in env.items()}):
                    ks = []
                    for i in range(reps):
                        p.node.run(lvl, cam, win, w, h, out=out)
                        ks.append(p.node.last_stats["kernel_ms"])
                    s = p.node.last_stats
                if ref is None:
                    ref = out.copy()
                same = bool(np.array_equal(ref.view(np.uint32), out.view(np.uint32)))
                if rnd == 1:
                    print(f"{st or '(default)':60s} best {min(ks[1:]):7.3f} ms  median {np.median(ks[1:]):7.3f} ms  mode {s['scene_in_lds']} "
                          f"grid {s['n_workgroups']}x{s['threads_per_workgroup']} lds {s['lds_bytes']} same_pixels {same}", flush=True)


if __name__ == "__main__":
    main()
