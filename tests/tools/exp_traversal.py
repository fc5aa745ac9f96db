#!/usr/bin/env python3
"""Driver of exp_traversal.c: wave-level execution counts of traversal variants on a tile sample of the headline frame."""
import ctypes as C
import os
import subprocess
import sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
os.environ.setdefault("BRT_NO_TORCH", "1")
import bevyray_amd as brt  # noqa: E402

so = os.path.join(HERE, "_exp_traversal.so")
src = os.path.join(HERE, "exp_traversal.c")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
    subprocess.check_call(["gcc", "-O2", "-std=c11", "-ffp-contract=off", "-fPIC", "-shared", "-pthread", "-o", so, src, "-lm"])
lib = C.CDLL(so)
VP, U32, I = C.c_void_p, C.c_uint32, C.c_int
lib.exp_run.argtypes = [VP, U32, VP, U32, VP, U32, VP, VP, U32, U32, VP, U32, I, I, I, I, I, I, I, I, VP]


def run(b, cam, win, W, H, tiles, width=2, near=0, cull=0, lip=0, vote=12, exit_lanes=12, sah=0, stop_live=40):
    out = np.zeros(16, np.uint64)
    t = np.ascontiguousarray(tiles, np.uint32)
    lib.exp_run(b.models.ctypes.data, len(b.models), b.materials.ctypes.data, len(b.materials), b.bvh.ctypes.data, len(b.bvh),
                cam.ctypes.data, win.ctypes.data, W, H, t.ctypes.data, len(t), width, near, cull, lip, vote, exit_lanes, sah,
                stop_live, out.ctypes.data)
    k = ["rounds", "round_lanes", "int_exec", "int_lanes", "leaf_exec", "leaf_lanes", "rays", "pop_skips", "sphere_tests", "box_tests", "nodes"]
    return dict(zip(k, [int(x) for x in out[:11]]))


COST_INT = {2: 64, 4: 135, 8: 260}     # wave instructions per interior-body execution (estimates; width 2 measured)
COST_LEAF = 75


def far_study(n=40):
    """Config 5 (10 004 spheres, top 879 records in LDS): would stepping near (LDS) and far (L2) lanes separately pay?"""
    W, H, spp, bounces = 1920, 1080, 64, 8
    b = brt.generate_scene(brt.SCENE_STRESS_GRID, 1)
    lvl, cam, win = brt.cover_camera(W, H, spp, bounces)
    rng = np.random.default_rng(5)
    tiles = np.stack([rng.integers(0, W // 8, n), rng.integers(0, H // 8, n)], 1)
    lib.exp_set_far.argtypes = [I, I, I]
    lib.exp_get_far.argtypes = [VP]
    far = np.zeros(8, np.uint64)
    print(f"{'policy':40s} {'int exec/rnd':>12s} {'lanes':>6s} {'far-containing':>14s} {'near steps':>10s} {'lanes':>6s} {'far steps':>9s} {'lanes':>6s} {'leaf/rnd':>8s}")
    for name, k, split, fv in [("kernel (one step for all), tile 879", 879, 0, 64), ("split, far when no near lane", 879, 1, 64),
                               ("split, far vote 32", 879, 1, 32), ("split, far vote 16", 879, 1, 16), ("split, far vote 8", 879, 1, 8),
                               ("kernel, tile 300", 300, 0, 64), ("split vote 16, tile 300", 300, 1, 16)]:
        lib.exp_set_far(k, split, fv)
        r = run(b, cam, win, W, H, tiles)
        lib.exp_get_far(far.ctypes.data)
        f = [int(x) for x in far]
        R = r["rounds"]
        print(f"{name:40s} {r['int_exec']/R:12.2f} {r['int_lanes']/max(1,r['int_exec']):6.1f} {f[0]/R:14.2f} {f[2]/R:10.2f} {f[3]/max(1,f[2]):6.1f} "
              f"{f[4]/R:9.2f} {f[5]/max(1,f[4]):6.1f} {r['leaf_exec']/R:8.2f}", flush=True)
    lib.exp_set_far(0, 0, 64)


def leaf_split_study(n=60, scene=None):
    """Would a leaf step in two parts pay?  Part A (the discriminant, ~55 cycles of a SIMD) for the lanes at a leaf; the lanes whose
    ray can still hit (discriminant >= 0, sphere ahead: 48 % of the tests on the cover frame) wait for part B (sqrt, divide, accept:
    ~150 cycles) until `vote_b` of them have gathered.  Costs: interior step 134 cycles, the kernel's single leaf step 200."""
    W, H, spp, bounces = 1920, 1080, 64, 8
    b = brt.generate_scene(brt.SCENE_COVER if scene is None else scene, 1)
    b = brt.Buffers(b.models, b.materials, brt.build_bvh_sah(b.models))      # the callee's tree, as the product walks it
    lvl, cam, win = brt.cover_camera(W, H, spp, bounces)
    rng = np.random.default_rng(5)
    tiles = np.stack([rng.integers(0, W // 8, n), rng.integers(0, H // 8, n)], 1)
    lib.exp_set_ab.argtypes = [I, I, I]
    lib.exp_get_ab.argtypes = [VP]
    ab = np.zeros(4, np.uint64)
    C_INT, C_LEAF, C_A, C_B = 134, 200, 55, 150
    print(f"{'policy':34s} {'int/rnd':>8s} {'lanes':>6s} {'leaf|A/rnd':>10s} {'lanes':>6s} {'B/rnd':>7s} {'lanes':>6s} {'rounds':>8s} {'walk cycles/ray':>15s}")
    base = None
    for name, on, va, vb in [("kernel: one leaf step, vote 12", 0, 12, 0), ("A vote 12, B vote 16", 1, 12, 16), ("A vote 12, B vote 24", 1, 12, 24),
                             ("A vote 12, B vote 32", 1, 12, 32), ("A vote 8, B vote 24", 1, 8, 24), ("A vote 4, B vote 24", 1, 4, 24),
                             ("A vote 4, B vote 32", 1, 4, 32), ("A vote 1, B vote 24", 1, 1, 24), ("A vote 8, B vote 40", 1, 8, 40),
                             ("A vote 12, B vote 1 (A then B at once)", 1, 12, 1)]:
        lib.exp_set_ab(on, va, vb)
        r = run(b, cam, win, W, H, tiles)
        lib.exp_get_ab(ab.ctypes.data)
        a = [int(x) for x in ab]
        R = r["rounds"]
        if on:
            cyc = r["int_exec"] * C_INT + a[0] * C_A + a[2] * C_B
            le, ll = a[0], a[1]
        else:
            cyc = r["int_exec"] * C_INT + r["leaf_exec"] * C_LEAF
            le, ll = r["leaf_exec"], r["leaf_lanes"]
        base = base or cyc / r["rays"]
        print(f"{name:34s} {r['int_exec']/R:8.2f} {r['int_lanes']/max(1,r['int_exec']):6.1f} {le/R:10.2f} {ll/max(1,le):6.1f} {a[2]/R:7.2f} {a[3]/max(1,a[2]):6.1f} "
              f"{R:8d} {cyc/r['rays']:15.1f}  ({100.0 * (cyc / r['rays'] / base - 1.0):+.1f} %)", flush=True)
    lib.exp_set_ab(0, 12, 24)


def leaf_size_study(n=60):
    """multi-sphere leaves in the callee's tree (raytrace.wgsl:348-362 loops `model_count` spheres): SAH trees whose
    leaves hold up to 1 / 2 / 4 spheres, forced or decided by the surface area heuristic with the wave costs of the two steps, on the
    views of configs 2, 3 (at 64 spp) and 5; callee's leaf pads.  Wave-level cost of a round's walk in SIMD cycles: interior step 112,
    leaf step 20 + 142 per sphere of the LARGEST leaf among the lanes of the step (measured issue costs)."""
    W, H = 1920, 1080
    lib.exp_set_leaf.argtypes = [I, I, I, C.c_float, C.c_float]
    lib.exp_get_leaf.argtypes = [VP]
    C_INT, C_LEAF0, C_SPH = 112.0, 20.0, 142.0
    out = np.zeros(12, np.uint64)
    for name, scene, camf, spp, bounces in (("config 2 (cover)", brt.SCENE_COVER, brt.cover_camera, 64, 8),
                                            ("config 3 view (RTIOW, 64 spp, 50 bounces)", brt.SCENE_RTIOW_FINAL, brt.rtiow_camera, 64, 50),
                                            ("config 5 (10 004-sphere grid)", brt.SCENE_STRESS_GRID, brt.cover_camera, 64, 8)):
        b = brt.generate_scene(scene, 1)
        lvl, cam, win = camf(W, H, spp, bounces)
        rng = np.random.default_rng(5)
        tiles = np.stack([rng.integers(0, W // 8, n), rng.integers(0, H // 8, n)], 1)
        print(f"== {name}: {len(b.models)} spheres, {n} random tiles")
        print(f"{'leaves':28s} {'leaves by size 1/2/3/4':>24s} {'int/ray':>8s} {'sph/ray':>8s} {'intX/rnd':>8s} {'lanes':>6s} {'leafX/rnd':>9s} {'lanes':>6s} {'units/rnd':>9s} {'walk cycles/round':>17s}")
        base = None
        for label, mx, rule in (("1 sphere (the product)", 1, 0), ("<= 2, forced", 2, 0), ("<= 2, by SAH", 2, 1), ("<= 4, forced", 4, 0), ("<= 4, by SAH", 4, 1)):
            lib.exp_set_leaf(mx, rule, 1, C_SPH, C_INT)
            r = run(b, cam, win, W, H, tiles, sah=1)
            lib.exp_get_leaf(out.ctypes.data)
            units = int(out[0]); hist = [int(x) for x in out[3:7]]
            R = r["rounds"]
            cyc = (r["int_exec"] * C_INT + r["leaf_exec"] * C_LEAF0 + units * C_SPH) / R
            base = base or cyc
            print(f"{label:28s} {'/'.join(map(str, hist)):>24s} {r['int_lanes']/r['rays']:8.2f} {r['sphere_tests']/r['rays']:8.2f} {r['int_exec']/R:8.2f} "
                  f"{r['int_lanes']/max(1,r['int_exec']):6.1f} {r['leaf_exec']/R:9.2f} {r['leaf_lanes']/max(1,r['leaf_exec']):6.1f} {units/R:9.2f} "
                  f"{cyc:12.0f} ({100.0 * (cyc / base - 1.0):+.1f} %)", flush=True)
    lib.exp_set_leaf(1, 0, 0, C_SPH, C_INT)


def policy_grid(n=60):
    """Leaf-vote x walk-exit thresholds of the width-2 walk: total wave instructions per ray (1060 non-walk per round)."""
    W, H, spp, bounces = 1920, 1080, 64, 8
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    lvl, cam, win = brt.cover_camera(W, H, spp, bounces)
    rng = np.random.default_rng(5)
    tiles = np.stack([rng.integers(0, W // 8, n), rng.integers(0, H // 8, n)], 1)
    exits = (4, 8, 12, 16, 24)
    print("rows: leaf vote; columns: exit lanes " + ", ".join(map(str, exits)) + "; cell: walk instr per round / instr per ray")
    for vote in (6, 8, 12, 16, 24, 32, 64):
        row = []
        for ex in exits:
            r = run(b, cam, win, W, H, tiles, vote=vote, exit_lanes=ex)
            walk = r["int_exec"] * COST_INT[2] + r["leaf_exec"] * COST_LEAF
            row.append(f"{walk / r['rounds']:6.0f}/{(walk + r['rounds'] * 1060) / r['rays']:5.1f}")
        print(f"vote {vote:2d}: " + "  ".join(row), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--far-study":
        return far_study(int(sys.argv[2]) if len(sys.argv) > 2 else 40)
    if len(sys.argv) > 1 and sys.argv[1] == "--leaf-split":
        return leaf_split_study(int(sys.argv[2]) if len(sys.argv) > 2 else 60, int(sys.argv[3]) if len(sys.argv) > 3 else None)
    if len(sys.argv) > 1 and sys.argv[1] == "--leaf-size":
        return leaf_size_study(int(sys.argv[2]) if len(sys.argv) > 2 else 60)
    if len(sys.argv) > 1 and sys.argv[1] == "--policy-grid":
        return policy_grid(int(sys.argv[2]) if len(sys.argv) > 2 else 60)
    scene = int(sys.argv[1]) if len(sys.argv) > 1 else brt.SCENE_COVER
    W, H, spp, bounces = 1920, 1080, 64, 8
    b = brt.generate_scene(scene, 1)
    lvl, cam, win = brt.cover_camera(W, H, spp, bounces)
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    rng = np.random.default_rng(5)
    tiles = np.stack([rng.integers(0, W // 8, n), rng.integers(0, H // 8, n)], 1)
    print(f"{n} tiles, scene {scene}, {len(b.models)} spheres")
    hdr = f"{'variant':44s} {'int/ray':>8s} {'leaf/ray':>8s} {'box/ray':>8s} {'sph/ray':>8s} {'intX/rnd':>8s} {'lanes':>6s} {'leafX/rnd':>9s} {'lanes':>6s} {'walk instr/round':>16s}"
    print(hdr)
    for name, kw in [
        ("w2 reference order (baseline)", dict()),
        ("w2 reference order, TWO pixels per lane", dict(sah=2)),
        ("w2 two pixels per lane, exit 6", dict(sah=2, exit_lanes=6)),
        ("w2 two pixels per lane, vote 16 exit 8", dict(sah=2, vote=16, exit_lanes=8)),
        ("w2 near-first + pop cull", dict(near=1, cull=1)),
        ("w2 near-first + cull + leaf-in-parent", dict(near=1, cull=1, lip=1)),
        ("w2 SAH tree, reference order", dict(sah=1)),
        ("w2 SAH near-first + cull", dict(sah=1, near=1, cull=1)),
        ("w4 reference order", dict(width=4)),
        ("w4 near-first + cull", dict(width=4, near=1, cull=1)),
        ("w4 unordered + cull", dict(width=4, near=0, cull=1)),
        ("w4 near-first + cull + leaf-in-parent", dict(width=4, near=1, cull=1, lip=1)),
        ("w4 unordered + cull + leaf-in-parent", dict(width=4, near=0, cull=1, lip=1)),
        ("w4 SAH near-first + cull", dict(width=4, near=1, cull=1, sah=1)),
        ("w4 SAH near-first + cull + lip", dict(width=4, near=1, cull=1, sah=1, lip=1)),
        ("w8 reference order", dict(width=8)),
        ("w8 near-first + cull", dict(width=8, near=1, cull=1)),
        ("w8 unordered + cull", dict(width=8, near=0, cull=1)),
        ("w8 near-first + cull + leaf-in-parent", dict(width=8, near=1, cull=1, lip=1)),
        ("w8 SAH near-first + cull + lip", dict(width=8, near=1, cull=1, sah=1, lip=1)),
    ]:
        r = run(b, cam, win, W, H, tiles, **kw)
        w = kw.get("width", 2)
        rounds = r["rounds"]
        cost = (r["int_exec"] * COST_INT[w] + r["leaf_exec"] * COST_LEAF) / rounds
        print(f"{name:44s} {r['int_lanes']/r['rays']:8.2f} {r['leaf_lanes']/r['rays']:8.2f} {r['box_tests']/r['rays']:8.1f} {r['sphere_tests']/r['rays']:8.2f} "
              f"{r['int_exec']/rounds:8.2f} {r['int_lanes']/max(1,r['int_exec']):6.1f} {r['leaf_exec']/rounds:9.2f} {r['leaf_lanes']/max(1,r['leaf_exec']):6.1f} {cost:16.0f}", flush=True)


if __name__ == "__main__":
    main()
