#!/usr/bin/env python3
"""Config 5 (10 004-sphere grid, 1920x1080, 64 spp, 8 bounces) with LDS tiles of different sizes, records numbered by use: what a
step that has to leave the tile costs.  With tests/tools/visit_hist.py (share of the interior visits served from LDS for each tile
size) the slope says what is left to gain by serving EVERY step from LDS.  usage: tile_slope.py [tile sizes ...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bevyray_amd as brt


def main():
    tiles = [int(x) for x in sys.argv[1:]] or [200, 400, 600, 879]
    b = brt.generate_scene(brt.SCENE_STRESS_GRID, 1)
    nb = brt.Buffers(b.models, b.materials, None)
    w, h = 1920, 1080
    lvl, cam, win = brt.cover_camera(w, h, 64, 8)
    for rnd in range(2):
        for tile in tiles:
            with brt.RaytracePlugin([0]) as p:
                p.set_tuning("BRT_FORCE_LDS_TOP", tile)
                out = p.alloc_frame(w, h)
                ks = []
                for i in range(8):
                    p.node.run(lvl, cam, win, w, h, buffers=nb if i == 0 else None, out=out)
                    ks.append(p.node.last_stats["kernel_ms"])
                st = p.node.last_stats
                if rnd == 1:
                    print(f"tile {tile:5d} records: best {min(ks[3:]):7.3f} ms  median {float(np.median(ks[3:])):7.3f} ms  "
                          f"hot_records {st['hot_records']}  scene_in_lds {st['scene_in_lds']}", flush=True)


if __name__ == "__main__":
    main()
