#!/usr/bin/env python3
"""Strong-scaling emulation on ONE GPU: renders EVERY rank's share (part p of N, interleaved 8-row strips) of
BASELINE.json configs 2 and 4 one after the other and reports, per N, the slowest share (= the frame time an
N-GPU node would see, gather aside), the sum over the shares (against the 1-GPU frame: what the split itself
costs) and the expected speed-up.  With no multi-GPU box available this is the hardware evidence for
SURVEY 8(e); bench.py --gpus N is the real thing.

    python scripts/gpu_parts.py [--configs 2 4] [--parts 1 2 4 8 16] [--reps 3]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bevyray_amd as brt

CONFIGS = {
    2: ("config 2: cover 1920x1080, 64 spp, 8 bounces", brt.SCENE_COVER, 1920, 1080, 64, 8, "cover"),
    4: ("config 4: RTIOW 3840x2160, 1024 spp, 8 bounces", brt.SCENE_RTIOW_FINAL, 3840, 2160, 1024, 8, "rtiow"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", type=int, nargs="*", default=[2, 4])
    ap.add_argument("--parts", type=int, nargs="*", default=[1, 2, 4, 8, 16])
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--knob", nargs="*", default=[], help="tuning knobs NAME=VALUE")
    ap.add_argument("--plan", type=int, default=0, help="strips by measured cost: brt_plan_strips with this many probe samples per pixel (0: strip s -> part s %% n)")
    a = ap.parse_args()
    with brt.RaytracePlugin([0]) as p:
        for kv in a.knob:
            k, v = kv.split("=", 1)
            p.set_tuning(k, int(v))
        for c in a.configs:
            name, kind, W, H, spp, bounces, camera = CONFIGS[c]
            b = brt.generate_scene(kind, 1)
            cam_fn = brt.rtiow_camera if camera == "rtiow" and hasattr(brt, "rtiow_camera") else brt.cover_camera
            lvl, cam, win = cam_fn(W, H, spp, bounces)
            # the tree the callee builds (what bench.py times); BRT_CALLER_TREE=1: the caller's PLOC tree
            p.node.write_buffers(b if os.environ.get("BRT_CALLER_TREE") == "1" else brt.Buffers(b.models, b.materials, None))
            print(f"== {name}", flush=True)
            one = None
            for n in a.parts:
                rows = brt.tile_rows(H, n)
                if a.plan and n > 1:
                    t = p.plan_strips(lvl, cam, win, W, H, n, probe_spp=a.plan)          # strips by measured cost (installs the table)
                    moved = int(sum(1 for s_, q in enumerate(t) if q != s_ % n))
                    print(f"   (strip table from a {a.plan}-spp probe: {moved} of {len(t)} strips leave their s % n part)", flush=True)
                else:
                    p.set_strip_table(n, None)
                tile = torch.zeros((rows, W, 4), dtype=torch.float32, device="cuda")
                torch.cuda.synchronize()
                reps = a.reps if c == 2 else max(2, a.reps - 1)
                per_part, rays = [], 0
                for part in range(n):
                    best = None
                    for _ in range(reps):     # first call of a (part, n) view = pre-pass + measuring frame; the rest are steady state
                        st = p.node.render_part_device(lvl, cam, win, W, H, part, n, tile.data_ptr())
                        best = st if best is None or st["kernel_ms"] < best["kernel_ms"] else best
                    per_part.append(best["kernel_ms"])
                    rays += best["rays"]
                slow, total = max(per_part), sum(per_part)
                if n == 1:
                    one = slow
                print(f"n_parts {n:2d}: slowest share {slow:8.3f} ms  fastest {min(per_part):8.3f}  sum over shares {total:8.3f} ms "
                      f"({total / one:5.2f} x the 1-GPU frame)  ideal {one / n:7.3f} ms  speed-up {one / slow:5.2f} of {n}  "
                      f"rays {rays}  {rays / slow / 1e3:9.1f} Mrays/s", flush=True)


if __name__ == "__main__":
    main()
