#!/usr/bin/env python3
"""What the TIMED kernel executes against what the counting kernel counts (What's weak #8).

The production instantiations walk in hand-written loops (walk_wave_lds_asm / walk_wave_top_asm, ball_loop_asm) and run half-sample
jobs; the COUNTERS instantiation -- where roofline.work's interior visits, sphere tests and sampler iterations come from -- runs the
compiler's loops and no half-sample jobs.  A -DBRT_ASM_COUNT build (scripts/build_variant.sh asmcount "-DBRT_ASM_COUNT=1") lets the
hand-written loops count their own executions and active lanes in scalar registers; this script renders BASELINE.json's configs 2 and
5 in that build, production frame and counting frame, and prints both sets side by side.  The LANE counts must be equal (the same
per-lane steps in the same order: they are the oracle's counters); the EXECUTION counts may differ (who shares a wave differs:
half-sample jobs, the leaf vote counted on walkers instead of leaf lanes) and the difference is printed.
    BRT_LIB_PATH=ab/libasmcount.so python scripts/asm_count.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevyray_amd as brt  # noqa: E402


def main():
    out = {}
    for name, scene in (("config2_cover", brt.SCENE_COVER), ("config5_grid10k", brt.SCENE_STRESS_GRID)):
        w, h, spp, bounces = 1920, 1080, 64, 8
        b = brt.generate_scene(scene, 1)
        lvl, cam, win = brt.cover_camera(w, h, spp, bounces)
        with brt.RaytracePlugin([0]) as p:
            p.node.write_buffers(brt.Buffers(b.models, b.materials, None))
            frame = p.alloc_frame(w, h)
            for _ in range(4):                                  # pre-pass + measuring frame, then steady-state frames
                p.node.run(lvl, cam, win, w, h, out=frame)
            st = dict(p.node.last_stats)
            p.debug_profile()
            asm = dict(p.last_asm_counts)
            meta = dict(p.last_order_meta)
            p.node.run(lvl, cam, win, w, h, out=frame, flags=brt.FLAG_COUNTERS)
            cs = dict(p.node.last_stats)
            prof = p.debug_profile()
        rec = {"production_frame": {"kernel_variant": st["kernel_variant"], "rays": st["rays"], "split_tiles": meta["split_tiles"],
                                    "interior_exec_lanes": asm["interior"], "leaf_exec_lanes": asm["leaf"], "ball_exec_lanes": asm["ball"]},
               "counting_frame": {"rays": cs["rays"], "interior_visits": cs["interior_visits"], "sphere_tests": cs["sphere_tests"],
                                  "interior_exec_lanes": prof["interior"], "leaf_exec_lanes": prof["leaf"], "ball_exec_lanes": prof["ball"]}}
        # (a wave that carries an unsafe ray -- an axis-parallel direction, say -- first walks in the compiler's repairing loop:
        #  those few steps are counted there)
        fix_i, fix_l = asm["repairing_loop_lanes"]
        rec["production_frame"]["of_which_in_the_repairing_loop"] = {"interior_lanes": fix_i, "leaf_lanes": fix_l}
        lanes_equal = (asm["interior"][1] + fix_i == cs["interior_visits"] == prof["interior"][1] and
                       asm["leaf"][1] + fix_l == cs["sphere_tests"] == prof["leaf"][1] and
                       asm["ball"][1] == prof["ball"][1] and st["rays"] == cs["rays"])
        rec["lane_counts_equal"] = bool(lanes_equal)
        rec["executions_production_over_counting"] = {k: round(asm[k][0] / max(1, prof[k][0]), 4) for k in ("interior", "leaf", "ball")}
        out[name] = rec
        print(name, json.dumps(rec), flush=True)
    ok = all(r["lane_counts_equal"] for r in out.values())
    print("lane counts of the hand-written loops == the counting kernel's == the oracle's counters:", ok)
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
