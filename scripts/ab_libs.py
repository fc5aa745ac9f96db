#!/usr/bin/env python3
"""Same-box A/B of library BUILDS: for every .so given, a child process (BRT_LIB_PATH) renders one workload a few times; two passes over
the list, the second is the one to read.  usage: ab_libs.py <scene> <w> <h> <spp> <bounces> lib1.so lib2.so ..."""
import os, subprocess, sys
CHILD = r'''
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(sys.argv[0]))) if False else os.getcwd())
import numpy as np, zlib
import bevyray_amd as brt
scene, w, h, spp, bounces, reps = (int(x) for x in sys.argv[1:7])
b = brt.generate_scene(scene, 1)
lvl, cam, win = (brt.rtiow_camera if scene == 1 else brt.cover_camera)(w, h, spp, bounces)
with brt.RaytracePlugin([0]) as p:
    p.node.write_buffers(brt.Buffers(b.models, b.materials, None))
    out = p.alloc_frame(w, h)
    ks = []
    for i in range(reps):
        p.node.run(lvl, cam, win, w, h, out=out)
        ks.append(p.node.last_stats["kernel_ms"])
    ks = ks[2:]
    print(f"best {min(ks):8.3f} ms  median {float(np.median(ks)):8.3f} ms  crc {zlib.crc32(out.tobytes()):08x}", flush=True)
'''
def main():
    args, libs = sys.argv[1:6], sys.argv[6:]
    reps = os.environ.get("AB_REPS", "10")
    for rnd in range(2):
        for lib in libs:
            env = dict(os.environ, BRT_LIB_PATH=os.path.abspath(lib))
            r = subprocess.run([sys.executable, "-c", CHILD, *args, reps], env=env, capture_output=True, text=True)
            if rnd == 1:
                print(f"{lib:28s} {r.stdout.strip() or r.stderr.strip()[-300:]}", flush=True)
if __name__ == "__main__":
    main()
