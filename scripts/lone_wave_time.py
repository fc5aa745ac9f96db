#!/usr/bin/env python3
"""Latency of ONE wave alone on the GPU: frames of 1x1 ... 8x8 pixels (one tile = one wave; the other CUs idle), many
samples per pixel, so that the kernel time is a chain of sequential rounds.  Prints us per round and the wave's time
by phase (COUNTERS build: 100 MHz ticks) -- what bounds the critical pixels of config 3 and a rank's share at 8 GPUs.

    python scripts/lone_wave_time.py [scene 0|1|2] [spp] [bounces]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevyray_amd as brt

def main():
    scene = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    bounces = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    b = brt.generate_scene(scene, 1)
    with brt.RaytracePlugin([0]) as p:
        # the tree the callee builds (what bench.py times); BRT_CALLER_TREE=1: the caller's PLOC tree
        p.node.write_buffers(b if os.environ.get("BRT_CALLER_TREE") == "1" else brt.Buffers(b.models, b.materials, None))
        for (w, h) in ((1, 1), (2, 1), (4, 1), (8, 1), (8, 2), (8, 4), (8, 8)):
            # a tiny frame of the same camera sees the scene centre: spheres and ground, paths of several bounces
            lvl, cam, win = (brt.rtiow_camera if scene == 1 else brt.cover_camera)(w, h, spp, bounces)
            best = None
            for _ in range(3):
                p.node.run(lvl, cam, win, w, h)
                s = p.node.last_stats
                best = s if best is None or s["kernel_ms"] < best["kernel_ms"] else best
            p.node.run(lvl, cam, win, w, h, flags=brt.FLAG_COUNTERS)
            prof = p.debug_profile()
            rounds = prof["round"][0]
            ph = (getattr(p, "last_timeline", None) or {}).get("wave_ms_refill_walk_shade_ball_pre")
            print(f"{w}x{h}: kernel {best['kernel_ms']:8.3f} ms  rays {best['rays']:8d}  rounds {rounds:7d}  "
                  f"{best['kernel_ms'] * 1e3 / max(1, rounds):6.2f} us/round  "
                  f"interior execs/round {prof['interior'][0] / max(1, rounds):5.1f} leaf {prof['leaf'][0] / max(1, rounds):4.1f} "
                  f"ball {prof['ball'][0] / max(1, rounds):4.1f}  phase ms (COUNTERS build) {ph}", flush=True)

if __name__ == "__main__":
    main()
