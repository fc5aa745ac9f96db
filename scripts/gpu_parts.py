#!/usr/bin/env python3
"""Emulates the N-GPU strong-scaling split on ONE GPU: times part 0 of N of the headline frame
(development aid for the tail behaviour when a GPU only gets 1/N of the pixels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bevyray_amd as brt
W, H, spp, bounces = 1920, 1080, 64, 8
b = brt.generate_scene(brt.SCENE_COVER, 1)
lvl, cam, win = brt.cover_camera(W, H, spp, bounces)
with brt.RaytracePlugin([0]) as p:
    p.node.write_buffers(b)
    for n in (1, 2, 4, 8, 16):
        rows = brt.tile_rows(H, n)
        tile = torch.zeros((rows, W, 4), dtype=torch.float32, device="cuda")
        best = None
        for part in (0, n - 1):
            for _ in range(3):
                st = p.node.render_part_device(lvl, cam, win, W, H, part, n, tile.data_ptr())
                best = st if best is None or st["kernel_ms"] < best["kernel_ms"] else best
        cs = p.node.render_part_device(lvl, cam, win, W, H, 0, n, tile.data_ptr(), flags=brt.FLAG_COUNTERS)
        prof = p.debug_profile()
        rl = prof["round"]; it = prof["interior"]
        print(f"      rounds {rl[0]:9d} avg live lanes {rl[1]/max(1,rl[0]):5.1f}; interior execs {it[0]:10d} avg lanes {it[1]/max(1,it[0]):5.1f}")
        full = 27.5
        print(f"n_parts {n:2d}: kernel {best['kernel_ms']:7.3f} ms  rays {best['rays']:10d}  {best['rays']/best['kernel_ms']/1e3:8.1f} Mrays/s  "
              f"grid {best['n_workgroups']}x{best['threads_per_workgroup']}  ideal {full/n:6.3f} ms", flush=True)
