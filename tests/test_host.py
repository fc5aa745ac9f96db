"""Host-side logic and the C-ABI surface -- runs without a GPU (no compute calls)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import bevyray_amd as brt
from bevyray_amd import _lib
from helpers import chain_bvh, make_buffers, median_split_bvh, single_leaf_bvh, uniforms

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "bevyray_amd.h")).read()
    declared = set(re.findall(r"\b(brt_[a-z0-9_]+)\s*\(", header))
    declared -= {"brt_ctx", "brt_stats"}
    assert len(declared) >= 16
    lib = C.CDLL(_lib.build())
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/bevyray_amd.h but not exported"
    assert declared == set(_lib.EXPORTS), "ctypes prototypes out of sync with the header"
    assert _lib.load().brt_abi_version() == 6
    import ctypes
    assert ctypes.sizeof(_lib.BrtStats) == 128         # 9 x 8 + 4 x 4 + prepass_ms + kernel_variant, measured_tile_costs + tree_rebuilt, tree_reach, forwarded_bytes (include/bevyray_amd.h, since ABI 5)


def test_wire_layouts_match_the_wgsl_structs():
    # raytrace.wgsl:30-87 / extract.rs:56-237
    assert brt.MODEL_DTYPE.itemsize == 32 and brt.MODEL_DTYPE.fields["radius"][1] == 12 and brt.MODEL_DTYPE.fields["material_id"][1] == 16
    assert brt.MATERIAL_DTYPE.itemsize == 32 and brt.MATERIAL_DTYPE.fields["specular_transmission"][1] == 28
    assert brt.BVH_NODE_DTYPE.itemsize == 48 and brt.BVH_NODE_DTYPE.fields["bounds_max"][1] == 16
    assert brt.BVH_NODE_DTYPE.fields["index"][1] == 28 and brt.BVH_NODE_DTYPE.fields["model_count"][1] == 32
    assert brt.CAMERA_DTYPE.itemsize == 80 and brt.CAMERA_DTYPE.fields["position"][1] == 32
    assert brt.CAMERA_DTYPE.fields["direction"][1] == 48 and brt.CAMERA_DTYPE.fields["up"][1] == 64
    assert brt.WINDOW_DTYPE.itemsize == 16 and brt.LEVEL_DTYPE.itemsize == 32
    assert [int(x) for x in brt.Raytracing] == [0, 1, 2, 3]   # mod.rs:94-101


def test_material_prepare_asset_decodes_srgb():
    # extract.rs:201 base_color.to_linear(); values from SURVEY.md H4
    want = {0.5: 0.21404114, 0.4: 0.13286832, 0.2: 0.033104766, 0.1: 0.010022826, 0.7: 0.44798842, 0.6: 0.31854677}
    for srgb, lin in want.items():
        m = brt.RaytraceMaterial.prepare_asset(brt.StandardMaterial(base_color=(srgb, srgb, srgb)))
        assert np.allclose(m["base_color"][0], lin, rtol=2e-7, atol=0)
    m = brt.RaytraceMaterial.prepare_asset(brt.StandardMaterial())[0]
    assert tuple(m["base_color"]) == (1.0, 1.0, 1.0) and m["metallic"] == 0.0 and m["roughness"] == 0.5
    assert m["reflectance"] == 0.5 and m["ior"] == 1.5 and m["specular_transmission"] == 0.0
    m = brt.RaytraceMaterial.prepare_asset(brt.StandardMaterial(metallic=1.0, perceptual_roughness=0.25, ior=1.33, specular_transmission=1.0))[0]
    assert m["metallic"] == 1.0 and m["roughness"] == 0.25 and m["ior"] == np.float32(1.33) and m["specular_transmission"] == 1.0


def test_camera_extract_is_forward_and_up_of_looking_at():
    cam = brt.RaytracedCamera(brt.Raytracing.FallbackRaytraced, 4, 4)   # main.rs:66-70
    proj = brt.PerspectiveProjection(fov=0.4, aspect_ratio=16 / 9, near=0.1, far=1000.0)
    lvl, c = brt.CameraExtract.extract_component(cam, brt.Transform((0, 0, 5), (0, 0, 0), (0, 1, 0)), proj)  # main.rs:57-58
    c = c[0]
    assert lvl["level"][0] == 2 and c["sample_count"] == 4 and c["bounce_count"] == 4 and c["projection"] == 0
    assert np.allclose(c["direction"], [0, 0, -1]) and np.allclose(c["up"], [0, 1, 0]) and np.allclose(c["position"], [0, 0, 5])
    assert c["fov"] == np.float32(0.4) and c["near"] == np.float32(0.1) and c["far"] == 1000.0
    _, c2 = brt.CameraExtract.extract_component(cam, brt.Transform((13, 2, 3), (0, 0, 0), (0, 1, 0)), proj)
    d, u = c2[0]["direction"].astype(np.float64), c2[0]["up"].astype(np.float64)
    assert abs(np.linalg.norm(d) - 1) < 1e-6 and abs(np.linalg.norm(u) - 1) < 1e-6 and abs(d @ u) < 1e-6
    assert np.allclose(d, -np.array([13, 2, 3]) / np.linalg.norm([13, 2, 3]), atol=1e-6)
    # orthographic -> None (extract.rs:148); the node then skips the pass
    assert brt.CameraExtract.extract_component(cam, brt.Transform(), brt.OrthographicProjection()) is None


def test_window_extract():
    w = brt.WindowExtract.extract_component(1080, 0.5)[0]
    assert w["height"] == 1080 and w["random_seed"] == 0.5


def _check_bvh_contract(models, bvh):
    n = len(models)
    assert len(bvh) == 2 * n - 1
    seen_models = np.zeros(n, int)
    seen_nodes = np.zeros(len(bvh), int)
    stack = [0]
    pos, rad = models["position"], models["radius"]
    while stack:
        i = stack.pop()
        seen_nodes[i] += 1
        nd = bvh[i]
        if nd["model_count"] > 0:
            assert nd["model_count"] == 1
            m = int(nd["index"])
            seen_models[m] += 1
            # Model::aabb pads by 0.1 (extract.rs:220-227)
            pad = rad[m] + np.float32(0.1)
            assert np.array_equal(nd["bounds_min"], pos[m] - pad) and np.array_equal(nd["bounds_max"], pos[m] + pad)
        else:
            a, b = int(nd["index"]), int(nd["index"]) + 1
            assert b < len(bvh)
            for c in (a, b):
                assert np.all(bvh[c]["bounds_min"] >= nd["bounds_min"]) and np.all(bvh[c]["bounds_max"] <= nd["bounds_max"])
            stack += [a, b]
    assert np.all(seen_models == 1) and np.all(seen_nodes == 1)


@pytest.mark.parametrize("kind,count", [(brt.SCENE_COVER, None), (brt.SCENE_RTIOW_FINAL, None), (brt.SCENE_STRESS_GRID, 10004)])
def test_scene_generators_and_ploc_contract(kind, count):
    b = brt.generate_scene(kind, 1)
    n = len(b.models)
    if count is not None:
        assert n == count
    else:
        assert 400 < n <= (23 * 22 + 4 if kind == brt.SCENE_COVER else 22 * 22 + 4)
    assert np.array_equal(b.models["material_id"], np.arange(n))            # extract.rs:301-310
    assert tuple(b.models[0]["position"]) == (0.0, -1000.0, 0.0) and b.models[0]["radius"] == 1000.0   # main.rs:96-100
    assert np.all(b.models[1:-3]["radius"] == np.float32(0.2)) and np.all(b.models[-3:]["radius"] == 1.0)
    assert [tuple(p) for p in b.models[-3:]["position"]] == [(0, 1, 0), (-4, 1, 0), (4, 1, 0)]           # main.rs:184-239
    _check_bvh_contract(b.models, b.bvh)
    depth = brt.validate_scene(b.models, b.materials, b.bvh)
    assert depth < 31, "PLOC tree should stay well below the shader's 32-entry stack"
    # deterministic in the seed, different across seeds
    b2 = brt.generate_scene(kind, 1)
    assert np.array_equal(b.models.view(np.uint8), b2.models.view(np.uint8)) and np.array_equal(b.bvh.view(np.uint8), b2.bvh.view(np.uint8))
    b3 = brt.generate_scene(kind, 2)
    assert not np.array_equal(b.models["position"][1:50], b3.models["position"][1:50])


def test_ploc_builder_terminates_on_non_finite_spheres():
    """The reference never validates spheres, so NaN / inf positions and radii reach the builder.
    Merge costs must stay symmetric and totally ordered (brt_ploc.h) or a round merges nothing and
    the build spins -- found by scripts/fuzz_parity.py with one NaN coordinate among 12 spheres."""
    rng = np.random.default_rng(58)
    for trial in range(200):
        n = int(rng.integers(2, 80))
        m = np.zeros(n, brt.MODEL_DTYPE)
        m["position"] = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
        m["radius"] = rng.uniform(0.05, 1.5, n).astype(np.float32)
        for _ in range(int(rng.integers(1, 5))):
            bad = rng.choice([np.nan, np.inf, -np.inf, 3e38, -3e38, 0.0])
            if rng.random() < 0.5:
                m["position"][int(rng.integers(0, n)), int(rng.integers(0, 3))] = bad
            else:
                m["radius"][int(rng.integers(0, n))] = bad
        if rng.random() < 0.3:
            m[int(rng.integers(0, n))] = m[int(rng.integers(0, n))]
        nodes = brt.build_bvh(m)
        assert len(nodes) == 2 * n - 1
        assert brt.validate_scene(m, np.zeros(1, brt.MATERIAL_DTYPE), nodes) >= 0
        leaves = nodes[nodes["model_count"] > 0]
        assert sorted(leaves["index"].tolist()) == list(range(n))


def test_cover_scene_material_lottery():
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    small = b.materials[1:-3]
    metal = small["metallic"] == 1.0
    glass = small["specular_transmission"] == 1.0
    diffuse = ~metal & ~glass
    n = len(small)
    assert 0.7 < diffuse.sum() / n < 0.9 and 0.08 < metal.sum() / n < 0.22 and 0.01 < glass.sum() / n < 0.1   # main.rs:116,137,159
    assert np.all(small["roughness"][diffuse] == 0.5) and np.all(small["roughness"][glass] == 0.5)        # defaults kept
    assert np.all(small["ior"] == 1.5) and np.all(small["base_color"][glass] == 1.0)
    # big metal: srgb (0.7,0.6,0.5), roughness 0 (main.rs:222-227)
    assert np.allclose(b.materials[-1]["base_color"], [0.44798842, 0.31854677, 0.21404114], rtol=1e-6) and b.materials[-1]["roughness"] == 0.0
    # skip rule main.rs:115
    d = np.linalg.norm(b.models[1:-3]["position"] - np.array([4, 0.2, 0], np.float32), axis=1)
    assert np.all(d > 0.9)


def test_prepare_buffers_mirrors_extract_stage():
    data = [((0.0, -1000.0, 0.0), brt.RaytracedSphere(1000.0), brt.StandardMaterial(base_color=(0.5, 0.5, 0.5))),
            ((1.0, 0.2, 0.0), brt.RaytracedSphere(0.2), brt.StandardMaterial(metallic=1.0, perceptual_roughness=0.1)),
            ((-1.0, 0.2, 0.5), brt.RaytracedSphere(0.2), brt.StandardMaterial(specular_transmission=1.0))]
    b = brt.prepare_buffers(data)
    assert len(b.models) == 3 and len(b.materials) == 3 and len(b.bvh) == 5
    assert list(b.models["material_id"]) == [0, 1, 2] and b.bvh[0]["model_count"] == 0
    _check_bvh_contract(b.models, b.bvh)
    one = brt.prepare_buffers(data[:1])
    assert len(one.bvh) == 1 and one.bvh[0]["model_count"] == 1 and one.bvh[0]["index"] == 0
    assert len(brt.prepare_buffers([]).bvh) == 0


def test_validate_rejects_malformed_input():
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    assert brt.validate_scene(b.models, b.materials, single_leaf_bvh(b.models)) == 0
    assert brt.validate_scene(b.models, b.materials, median_split_bvh(b.models, 3)) >= 7
    assert brt.validate_scene(b.models[:40], b.materials, chain_bvh(b.models[:40])) == 39

    def rejected(models, materials, bvh, code):
        with pytest.raises(brt.BrtError) as e:
            brt.validate_scene(models, materials, bvh)
        assert e.value.code == code, e.value

    bad = b.bvh.copy(); bad[0]["index"] = len(bad) - 1          # child index+1 out of range
    rejected(b.models, b.materials, bad, -4)
    bad = b.bvh.copy(); bad[int(b.bvh[0]["index"])]["index"] = 0; bad[int(b.bvh[0]["index"])]["model_count"] = 0   # cycle to the root
    rejected(b.models, b.materials, bad, -4)
    leaf = int(np.flatnonzero(b.bvh["model_count"] > 0)[0])
    bad = b.bvh.copy(); bad[leaf]["index"] = len(b.models)       # leaf range out of range
    rejected(b.models, b.materials, bad, -4)
    bad = b.bvh.copy(); bad[leaf]["model_count"] = 0xFFFFFFFF    # u32 overflow of index + count
    rejected(b.models, b.materials, bad, -4)
    badm = b.models.copy(); badm[7]["material_id"] = len(b.materials)
    rejected(badm, b.materials, b.bvh, -5)
    rejected(b.models[:0], b.materials, b.bvh, -6)               # empty scene: the reference skips the pass
    rejected(b.models, b.materials, b.bvh[:0], -4)


def test_tile_rows_and_strip_mapping():
    from bevyray_amd.parallel import frame_rows_of_part
    assert brt.tile_rows(1080, 1) == 1080 and brt.tile_rows(1080, 8) == 136 and brt.tile_rows(225, 2) == 120
    assert brt.tile_rows(2160, 8) == 272 and brt.tile_rows(7, 3) == 8
    for h, n in [(1080, 8), (225, 2), (36, 3), (7, 3), (64, 5)]:
        seen = np.zeros(h, int)
        for p in range(n):
            rows = frame_rows_of_part(h, p, n)
            assert len(rows) == brt.tile_rows(h, n)
            valid = rows[rows >= 0]
            seen[valid] += 1
            assert np.all((valid // 8) % n == p)
        assert np.all(seen == 1)


def test_create_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(brt.BrtError) as e:
        brt.RaytracePlugin([0])
    assert e.value.code == -2 and "no CPU path" in e.value.text


def test_new_exports_fail_with_codes_not_crashes_without_a_context():
    """ABI 4 entry points called the wrong way (no GPU needed): null contexts and pointers come back as BRT_ERR_INVALID_ARGUMENT,
    never as a crash; the RCCL id either works (librccl resolves by dlopen: rank 0 of any host may call it without a context) or
    reports BRT_ERR_RCCL with a text."""
    import ctypes as C
    lib = _lib.load()
    ptr, fd = C.c_void_p(), C.c_int32(-1)
    assert lib.brt_gather_rccl(None, None, 0, 1, None, None, 8, 8, None, None, 0) == -1
    assert lib.brt_rccl_comm_create(None, None, 0, 1, C.byref(ptr)) == -1
    assert lib.brt_rccl_comm_destroy(None, None) == 0                      # nothing to destroy
    assert lib.brt_import_frame_fd(None, 3, 64, 2, C.byref(ptr)) == -1
    assert lib.brt_release_frame(None, None) == -1
    assert lib.brt_debug_export_frame_fd(None, 64, C.byref(fd), C.byref(ptr)) == -1
    assert lib.brt_debug_copy_to_host(None, None, None, 0) == -1
    assert lib.brt_build_bvh_sah_device(None, None, 0, 0.0, None, 0, None, None) == -1
    assert lib.brt_host_tree_reach(None, 3, None, None, None, None) == -1
    assert lib.brt_rccl_unique_id(None) == -1
    buf = (C.c_char * 128)()
    rc = lib.brt_rccl_unique_id(buf)
    assert rc in (0, -10)
    if rc == -10:
        assert lib.brt_last_error(None)


def _tile_order(ray_sum, longest, spp, grid_lanes, sorted_=1, lane_permille=0, tiles_x=0, dilate=0, split_tail=0):
    ray_sum = np.ascontiguousarray(ray_sum, np.uint32)
    longest = np.ascontiguousarray(longest, np.uint32)
    order = np.zeros(len(ray_sum) + split_tail, np.uint32)
    info = np.zeros(5, np.uint32)
    _lib.check(_lib.load().brt_host_tile_order(ray_sum.ctypes.data, longest.ctypes.data, len(ray_sum), spp, grid_lanes, sorted_,
                                               lane_permille, tiles_x, dilate, split_tail, order.ctypes.data, info.ctypes.data))
    info = dict(zip(("n_lane", "n_critical", "longest_pixel", "n_nonsky", "n_split"), (int(x) for x in info)))
    return order[:len(ray_sum) + info["n_split"]], info


def test_dispatch_order_from_tile_costs():
    """brt_host.cpp build_tile_order: non-sky tiles by their longest pixel (longest first), sky tiles last in raster
    order; the front of the order can go to the lane queue; critical tiles are a prefix."""
    spp, n = 64, 200
    rng = np.random.default_rng(3)
    longest = rng.integers(64, 577, n).astype(np.uint32)           # rays of a tile's longest pixel (<= spp * 9)
    ray_sum = (64 * spp + 200 + longest.astype(np.uint64) * 30).astype(np.uint32)   # 64 pixels, most shorter than the longest
    sky = rng.random(n) < 0.25
    longest[sky], ray_sum[sky] = spp, 64 * spp                     # one ray per sample in every pixel
    lanes = 256 * 1024
    order, info = _tile_order(ray_sum, longest, spp, lanes)
    assert sorted(order.tolist()) == list(range(n))                # a permutation
    n_sky = int(sky.sum())
    front, back = order[:n - n_sky], order[n - n_sky:]
    assert set(back.tolist()) == set(np.flatnonzero(sky).tolist()) and list(back) == sorted(back)   # sky last, raster order
    assert list(longest[front]) == sorted(longest[front], reverse=True)             # longest pixel first
    assert info["n_lane"] == 0 and info["longest_pixel"] == int(longest.max())
    # ties keep raster order
    same = np.flatnonzero(longest[front][1:] == longest[front][:-1])
    assert all(front[i] < front[i + 1] for i in same)
    # this frame is far longer than any pixel (sum / lanes >> longest): nothing is critical ...
    big = np.full(n, 4_000_000, np.uint32)
    assert _tile_order(big, longest, spp, 1000)[1]["n_critical"] == 0
    # ... a frame whose lanes work through ~200 rays each while pixels need up to 576 has critical tiles: the front
    # of the order down to half the longest pixel
    order2, info2 = _tile_order(ray_sum, longest, spp, int(ray_sum.sum()) // 200)
    assert info2["n_critical"] > 0 and np.array_equal(order2, order)
    assert longest[order2[:info2["n_critical"]]].min() >= info2["longest_pixel"] // 2
    assert info2["n_critical"] == n - n_sky or longest[order2[info2["n_critical"]]] < info2["longest_pixel"] // 2
    # a tenth of the non-sky tiles to the lane queue; raster order instead of the ranking
    assert _tile_order(ray_sum, longest, spp, lanes, lane_permille=100)[1]["n_lane"] == (n - n_sky) // 10
    order3, info3 = _tile_order(ray_sum, longest, spp, lanes, sorted_=0)
    assert list(order3[:n - n_sky]) == sorted(order3[:n - n_sky]) and info3["n_critical"] == 0
    # half-sample jobs: the last k non-sky tiles appear twice -- [other non-sky | first halves | second halves | sky] -- k capped by the
    # non-sky tiles that are not critical; the order without the repeats is the plain order
    for k in (0, 1, 37, n - n_sky, n):
        order4, info4 = _tile_order(ray_sum, longest, spp, lanes, split_tail=k)
        ks = min(k, n - n_sky - info4["n_critical"])     # (never a critical tile)
        assert info4["n_nonsky"] == n - n_sky and info4["n_split"] == ks and len(order4) == n + ks
        assert np.array_equal(order4[:n - n_sky], order[:n - n_sky])
        assert np.array_equal(order4[n - n_sky:n - n_sky + ks], order[n - n_sky - ks:n - n_sky])
        assert np.array_equal(order4[n - n_sky + ks:], order[n - n_sky:])
    # a critical tile is never split
    order6, info6 = _tile_order(ray_sum, longest, spp, int(ray_sum.sum()) // 200, split_tail=n)
    assert info6["n_critical"] > 0 and info6["n_split"] == n - n_sky - info6["n_critical"]
    assert np.array_equal(order6[n - n_sky:n - n_sky + info6["n_split"]], order[info6["n_critical"]:n - n_sky])


def test_sah_builder_contract_depth_cap_and_determinism():
    """brt_build_bvh_sah: the reference's node contract (extract.rs:323-332: root 0, children adjacent, single-sphere
    leaves that address the model buffer), every sphere exactly once, deterministic, depth below the shader's 32-entry
    stack even where SAH alone would build a caterpillar, and it terminates on NaN / inf spheres."""
    rng = np.random.default_rng(5)
    cases = [brt.generate_scene(k, 1).models for k in (brt.SCENE_COVER, brt.SCENE_RTIOW_FINAL, brt.SCENE_STRESS_GRID)]
    for n in (1, 2, 3, 7, 64, 1000):
        m = np.zeros(n, brt.MODEL_DTYPE)
        m["position"] = rng.uniform(-30, 30, (n, 3)).astype(np.float32)
        m["radius"] = rng.uniform(0.05, 2.0, n).astype(np.float32)
        cases.append(m)
    geo = np.zeros(200, brt.MODEL_DTYPE)                       # centres at 1.3^k: every SAH split peels one sphere off
    geo["position"][:, 0] = (1.3 ** np.arange(200)).astype(np.float32)
    geo["radius"] = 0.1
    dup = np.zeros(300, brt.MODEL_DTYPE); dup["position"] = (1.0, 2.0, 3.0); dup["radius"] = 0.5
    bad = np.zeros(64, brt.MODEL_DTYPE)
    bad["position"] = rng.uniform(-5, 5, (64, 3)).astype(np.float32); bad["radius"] = 0.3
    bad["position"][::5, 1] = np.nan; bad["position"][3::7, 0] = np.inf; bad["radius"][1::9] = np.nan
    cases += [geo, dup, bad]
    for m in cases:
        n = len(m)
        nodes = brt.build_bvh_sah(m)
        assert len(nodes) == 2 * n - 1
        assert np.array_equal(nodes.view(np.uint8), brt.build_bvh_sah(m).view(np.uint8))      # deterministic
        mats = np.zeros(max(1, int(m["material_id"].max()) + 1), brt.MATERIAL_DTYPE)
        depth = brt.validate_scene(m, mats, nodes)
        assert depth <= 28, (n, depth)
        leaves = nodes[nodes["model_count"] > 0]
        assert np.all(leaves["model_count"] == 1) and sorted(leaves["index"].tolist()) == list(range(n))
        inner = nodes[nodes["model_count"] == 0]
        assert len(inner) == n - 1 and np.all(inner["index"] % 2 == 1)                         # children at odd / even slot pairs
    # finite scenes: every leaf box is the sphere's bounds padded by the rule of brt_sah.h -- clamp(2^-24 (2 S)^2 / r, 0.01, 0.1), S the
    # largest |c|_1 + r over the spheres of radius <= 100; larger spheres keep Model::aabb's 0.1 (extract.rs:220-227) -- and every
    # parent holds its children
    m = cases[0]
    nodes = brt.build_bvh_sah(m)
    f32 = np.float32
    ordinary = m["radius"] <= 100
    ap = np.abs(m["position"][ordinary]).astype(f32)
    scale = (((ap[:, 0] + ap[:, 1]) + ap[:, 2]) + m["radius"][ordinary]).max()
    pads = set()
    for nd in nodes:
        if nd["model_count"]:
            c, r = m[nd["index"]]["position"], f32(m[nd["index"]]["radius"])
            if r <= 100:
                d = f32(2.0) * f32(scale)
                p = (f32(5.9604645e-8) * (d * d)) / r
                p = f32(min(max(p, f32(0.01)), f32(0.1)))
            else:
                p = f32(0.1)
            pads.add(float(p))
            pad = r + p
            assert np.array_equal(nd["bounds_min"], c - pad) and np.array_equal(nd["bounds_max"], c + pad)
        else:
            for ch in (nodes[nd["index"]], nodes[nd["index"] + 1]):
                assert np.all(ch["bounds_min"] >= nd["bounds_min"]) and np.all(ch["bounds_max"] <= nd["bounds_max"])
    assert min(pads) == float(f32(0.01)) and max(pads) == float(f32(0.1))      # the small spheres and the ground sphere of the cover scene


def _leaf_pads(models, nodes):
    """leaf box half-width minus radius, per sphere (x axis of bounds_max: exact in f32 only up to the rounding of c + (r + pad))"""
    pads = np.zeros(len(models), np.float64)
    for nd in nodes[nodes["model_count"] > 0]:
        m = models[nd["index"]]
        pads[nd["index"]] = (float(nd["bounds_max"][1]) - float(nd["bounds_min"][1])) / 2 - float(m["radius"])
    return pads


def test_tree_reach_rule_and_pads_follow_the_camera():
    """The callee-built SAH tree pads its leaf boxes for the distances rays travel -- camera included (brt_sah.h):
    brt_host_tree_reach is the rule brt_render* applies, brt_build_bvh_sah(reach) the tree it then builds.  Replaces what
    Model::aabb's flat 0.1 (extract.rs:220-227) does for the reference whatever the camera."""
    f32 = np.float32
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    m = b.models
    ordinary = m["radius"] <= 100
    ap = np.abs(m["position"][ordinary]).astype(f32)
    S_want = float((((ap[:, 0] + ap[:, 1]) + ap[:, 2]) + m["radius"][ordinary]).max())
    _, cam, _ = brt.cover_camera(160, 90, 1, 1)
    S, level, reach = brt.tree_reach(m, cam)
    assert S == S_want
    # the cover camera: |(13, 2, 3)|_1 + S + the tangent to the ground sphere ~ 105 > 2 S, yet every pad of that reach is still the
    # 0.01 floor -> the tree of the scene's own extent already is that tree: level 0
    assert (level, reach) == (0, 0.0)
    base = brt.build_bvh_sah(m)
    assert np.array_equal(base.view(np.uint8), brt.build_bvh_sah(m, 105.0).view(np.uint8))
    # moving out along the view axis: the level never falls, the reach covers |cam|_1 + S + tangent, the pads grow up to the reference's 0.1
    last_level, last_pads = 0, _leaf_pads(m, base)
    for k in (2, 5, 10, 20, 30, 60, 1000, 1e6):
        _, cam, _ = uniforms(160, 90, 1, 1, (13.0 * k, 2.0 * k, 3.0 * k), (0, 0, 0), 0.4 / k, 0.5)
        S2, level, reach = brt.tree_reach(m, cam)
        assert S2 == S and level >= last_level and level <= 80
        if level:
            c = np.array([13.0 * k, 2.0 * k, 3.0 * k])
            hgt = np.linalg.norm(c - np.array([0.0, -1000.0, 0.0])) - 1000.0
            need = np.abs(c).sum() + S + np.sqrt(hgt * (2000.0 + hgt))
            assert reach >= need * (1 - 1e-6) and reach <= need * 2 ** 0.25 * (1 + 1e-6)
            assert abs(reach - 2 * S * 2 ** (level / 4)) <= 1e-5 * reach
        nodes = brt.build_bvh_sah(m, reach)
        assert brt.validate_scene(m, b.materials, nodes) <= 28
        pads = _leaf_pads(m, nodes)
        assert np.all(pads >= last_pads - 1e-4) and pads.max() <= 0.1 + 1e-4 and pads.min() >= 0.01 - 1e-4
        # the rule of brt_sah.h at this reach, in f32, for the r = 0.2 spheres
        small = np.isclose(m["radius"], 0.2)
        d = f32(2.0) * max(f32(S), f32(0.5) * f32(reach))
        want = min(max((f32(5.9604645e-8) * (d * d)) / f32(0.2), f32(0.01)), f32(0.1))
        assert np.allclose(pads[small], float(want), atol=2e-6)
        last_level, last_pads = level, pads
    assert last_level >= 70 and np.allclose(last_pads, 0.1, atol=1e-4)           # far enough out: the reference's own boxes
    # a camera that is not a number needs the most: level 80, not "no floor"
    _, cam, _ = brt.cover_camera(160, 90, 1, 1)
    cam = cam.copy(); cam["position"][0, 1] = np.nan
    assert brt.tree_reach(m, cam)[1] == 80
    # a scene without ordinary spheres has nothing the camera could change
    g = m[m["radius"] > 100]
    assert brt.tree_reach(g, cam)[1] == 0


def test_far_camera_frames_in_the_callee_tree_equal_the_reference_tree(oracle):
    """CPU statement of the -m gpu far-camera test: the oracle's frame in the callee's tree built for the camera's reach equals its
    frame in the reference's 0.1-padded PLOC tree (and in one leaf of all spheres) at x 1 ... x 30 the cover distance -- where the
    tree of the scene's own extent (round 4's only tree) loses 7 / 53 / 377 pixels from x 20 on."""
    b = brt.generate_scene(brt.SCENE_COVER, 1)
    w, h = 160, 90
    brute = single_leaf_bvh(b.models)
    lost = 0
    for k in (1, 20, 25, 30):
        lvl, cam, win = uniforms(w, h, 4, 4, (13.0 * k, 2.0 * k, 3.0 * k), (0, 0, 0), 0.4 / k, 0.5, far=1e5)
        _, level, reach = brt.tree_reach(b.models, cam)
        assert (level > 0) == (k > 1)
        frames = [oracle.render(brt.Buffers(b.models, b.materials, t), lvl, cam, win, w, h)[0]
                  for t in (brt.build_bvh_sah(b.models, reach), b.bvh, brute, brt.build_bvh_sah(b.models))]
        assert np.array_equal(frames[0].view(np.uint32), frames[1].view(np.uint32))
        assert np.array_equal(frames[0].view(np.uint32), frames[2].view(np.uint32))
        lost += int((frames[3].view(np.uint32) != frames[2].view(np.uint32)).any(axis=2).sum())
    assert lost > 100            # the camera-blind tree does lose pixels out there: the case is real


def test_rccl_not_found_is_an_error_code_not_a_crash():
    """a host without librccl (the single-GPU case the header says needs none) must get BRT_ERR_RCCL from every
    brt_rccl_* / brt_gather_rccl call -- the message built from ONE dlerror() call (a second one returns NULL).  Own process: the
    library resolves librccl once.  BRT_RCCL_LIB names the only library tried."""
    import subprocess
    import sys
    code = (
        "import ctypes as C, sys\n"
        "from bevyray_amd import _lib\n"
        "lib = _lib.load()\n"
        "buf = (C.c_char * 128)()\n"
        "rc = lib.brt_rccl_unique_id(buf)\n"
        "msg = lib.brt_last_error(None).decode()\n"
        "rc2 = lib.brt_rccl_unique_id(buf)\n"
        "print(rc, rc2, msg)\n")
    env = dict(os.environ, BRT_RCCL_LIB="/nonexistent/librccl_not_here.so", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    proc = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    rc, rc2, msg = proc.stdout.strip().split(" ", 2)
    assert (rc, rc2) == ("-10", "-10")
    assert "librccl not found" in msg and "librccl_not_here" in msg


def test_srgb_store_thresholds_and_the_oracle_formula_agree_at_every_decision_point(oracle):
    """BRT_FLAG_OUT_RGBA8_UNORM_SRGB (the store into TextureFormat::bevy_default(), pipeline.rs:311-315): the product counts 255
    precomputed f32 thresholds (exact: scripts/gen_srgb_table.py evaluates them in 60-digit arithmetic), the oracle evaluates
    round(255 * OETF(c)) in double.  They must agree on every f32 -- it suffices to look where the code changes: at each threshold the
    oracle says k, one float below it k - 1; plus a dense sample, the segment joint and the specials."""
    t = brt.srgb_thresholds()
    assert t.shape == (255,) and np.all(np.diff(t) > 0) and 0 < t[0] and t[-1] < 1
    below = np.nextafter(t, np.float32(-1.0), dtype=np.float32)
    enc = lambda x: oracle.encode_frame(np.stack([x, x, x, x], axis=-1).reshape(-1, 1, 4), "srgb8")[:, 0, 0].astype(np.int64)
    assert np.array_equal(enc(t), np.arange(1, 256)) and np.array_equal(enc(below), np.arange(0, 255))
    rng = np.random.default_rng(9)
    x = np.concatenate([rng.random(300000).astype(np.float32), (rng.random(100000) * 0.01).astype(np.float32),
                        np.array([0.0, -0.0, 1.0, 2.0, -1.0, np.inf, -np.inf, np.nan, 0.0031308, 0.00313080495, 0.04045, 1e-30], np.float32)])
    with np.errstate(invalid="ignore"):
        want = np.searchsorted(t, x, side="right")       # thresholds <= x  (NaN sorts last: fixed below)
    want[np.isnan(x)] = 0
    assert np.array_equal(enc(x), want)
    # the regenerated table is the committed one
    import subprocess, sys
    header = os.path.join(ROOT, "bevyray_amd", "csrc", "brt_srgb_table.h")
    before = open(header).read()
    subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gen_srgb_table.py")], check=True, capture_output=True)
    assert open(header).read() == before


_OOM_CHILD = r'''
import ctypes as C, resource, sys
import numpy as np
from bevyray_amd import _lib
lib = _lib.load()
n = int(sys.argv[1])
rng = np.random.default_rng(7)
models = np.zeros((n, 8), np.float32)
models[:, 0:3] = rng.uniform(-100, 100, (n, 3)).astype(np.float32)
models[:, 3] = 0.2
vm = 0
for ln in open("/proc/self/status"):
    if ln.startswith("VmSize:"):
        vm = int(ln.split()[1]) * 1024
# the builders need (2 n - 1) x 48 bytes for the nodes alone: leave them 32 MB of address space
resource.setrlimit(resource.RLIMIT_AS, (vm + (32 << 20), resource.RLIM_INFINITY))
n_nodes = C.c_uint32(123)
for name, args in (("brt_build_bvh_sah", (C.c_float(0.0),)), ("brt_build_bvh", ())):
    rc = getattr(lib, name)(models.ctypes.data, n, *args, None, 0, C.byref(n_nodes))
    print(name, rc, lib.brt_last_error(None).decode(), n_nodes.value, flush=True)
# the library is still usable afterwards
resource.setrlimit(resource.RLIMIT_AS, (resource.RLIM_INFINITY, resource.RLIM_INFINITY))
out = np.zeros((19, 12), np.uint32)
rc = lib.brt_build_bvh_sah(models.ctypes.data, 10, C.c_float(0.0), out.ctypes.data, 19, C.byref(n_nodes))
print("after", rc, n_nodes.value, flush=True)
'''


def test_exception_barrier_turns_a_failed_host_allocation_into_an_error_code():
    """include/bevyray_amd.h: no export throws across the boundary (pipeline.rs:82-85: the node skips the pass).  A child process whose
    address space is capped lets the two host builders run out of memory on 3 M spheres: BRT_ERR_OUT_OF_MEMORY + text, no abort."""
    import subprocess
    import sys
    if "asan" in os.environ.get("LD_PRELOAD", ""):
        pytest.skip("an address-space cap and AddressSanitizer's shadow memory do not go together (scripts/asan_host.sh)")
    r = subprocess.run([sys.executable, "-c", _OOM_CHILD, "3000000"], capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = r.stdout.strip().splitlines()
    assert lines[0].split()[:2] == ["brt_build_bvh_sah", "-11"] and "out of memory" in lines[0], r.stdout
    assert lines[1].split()[:2] == ["brt_build_bvh", "-11"] and "out of memory" in lines[1], r.stdout
    assert lines[0].split()[-1] == "0" and lines[1].split()[-1] == "0"        # out_n_nodes was reset before the failure
    assert lines[2] == "after 0 19", r.stdout
