#!/usr/bin/env python3
"""Sampler stage (knob BRT_BALL_SERVERS; brt_trace.h SRV) against the one-path-per-lane kernel, same process, same box:
    python scripts/srv_ab.py [scene w h spp bounces]"""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bevyray_amd as brt
a = [int(x) for x in sys.argv[1:6]] if len(sys.argv) >= 6 else [0, 1920, 1080, 64, 8]
scene, w, h, spp, bounces = a
b = brt.generate_scene(scene, 1)
lvl, cam, win = (brt.rtiow_camera if scene == 1 else brt.cover_camera)(w, h, spp, bounces)
with brt.RaytracePlugin([0]) as p:
    p.node.write_buffers(brt.Buffers(b.models, b.materials, None))
    out = p.alloc_frame(w, h)
    for rep in range(2):
        for srv in (0, 1):
            p.set_tuning("BRT_BALL_SERVERS", srv)
            ks, var = [], None
            for i in range(10):
                p.node.run(lvl, cam, win, w, h, out=out)
                st = p.node.last_stats
                ks.append(st["kernel_ms"]); var = st["kernel_variant"]
            p.debug_profile()
            print(f"servers {srv}: best {min(ks[3:]):7.3f} median {float(np.median(ks[3:])):7.3f} ms  variant {var} lds {st['lds_bytes']} rays {st['rays']} "
                  f"crc {zlib.crc32(out.tobytes()):08x}  stage {p.last_sampler_stage}", flush=True)
