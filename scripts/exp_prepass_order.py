#!/usr/bin/env python3
"""Experiment: how good is a dispatch order measured on a 1-spp frame for the 64-spp frame?
(the order key does not include the sample count, so the 64-spp frames below reuse it)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bevyray_amd as brt


def main():
    W, H, spp, bounces = 1920, 1080, 64, 8
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    with brt.RaytracePlugin([0]) as p:
        for pre in (1, 2, 4, 64):
            p.node.write_buffers(brt.generate_scene(brt.SCENE_COVER, 2))   # forget the history
            p.node.write_buffers(b)
            lvl, cam, win = brt.cover_camera(W, H, pre, bounces)
            p.node.run(lvl, cam, win, W, H)
            pre_ms = p.node.last_stats["kernel_ms"]
            lvl, cam, win = brt.cover_camera(W, H, spp, bounces)
            ks = []
            for i in range(10):
                p.node.run(lvl, cam, win, W, H)
                ks.append(p.node.last_stats["kernel_ms"])
            print(f"order from a {pre:2d}-spp frame ({pre_ms:5.2f} ms): 64-spp frames {np.mean(ks[3:]):6.2f} ms (min {min(ks):6.2f})", flush=True)


if __name__ == "__main__":
    main()
