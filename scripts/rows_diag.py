#!/usr/bin/env python3
"""-DBRT_ASM_COUNT build: how much of a thin frame's walk runs in the row-mode loop (walk_rows_asm)?
    BRT_LIB_PATH=ab/libasmcount.so python scripts/rows_diag.py [scene]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevyray_amd as brt
scene = int(sys.argv[1]) if len(sys.argv) > 1 else 1
b = brt.generate_scene(scene, 1)
with brt.RaytracePlugin([0]) as p:
    p.node.write_buffers(brt.Buffers(b.models, b.materials, None))
    for (w, h) in ((1, 1), (2, 1), (4, 1), (8, 1)):
        lvl, cam, win = (brt.rtiow_camera if scene == 1 else brt.cover_camera)(w, h, 512, 8)
        for _ in range(3):
            p.node.run(lvl, cam, win, w, h)
        st = dict(p.node.last_stats)
        p.debug_profile()
        print(f"{w}x{h}: kernel {st['kernel_ms']:.3f} ms variant {st['kernel_variant']} rays {st['rays']}", p.last_asm_counts, flush=True)
