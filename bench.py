#!/usr/bin/env python3
"""Headline benchmark: Mrays/s on the README cover scene at 1920x1080, 64 spp, 8 bounces
(BASELINE.json metric, configs[1]).

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts its own N ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = one full frame: every rank traces its interleaved row strips with the persistent HIP
kernel (scene already resident in HBM), the tiles meet on rank 0 in ONE gather over RCCL, and
rank 0 de-interleaves.  The frame is fixed while N grows, so scaling is "strong".
Rank 0 prints one JSON line.  `value` = rays of all ranks / max-over-ranks wall time of the K
timed steps.

`roofline` names the resource that binds the trace kernel: the f32 vector pipes (VALU
lane-cycles).  achieved = VALU instructions x average active lanes / kernel time, peak = 256 CUs
x 4 SIMDs x 32 lanes x 2.4 GHz; the instruction and lane counts come from a rocprofv3 --pmc
pass over the same workload made by THIS run after the timed region (fallback: the committed
profiles/pmc_summary.json, only if it was taken on the same device code), the kernel time from
HIP events on the kernel's own stream inside the timed steps.  The SURVEY.md 8(d) algorithmic
bytes and the measured HBM bytes are carried as secondary keys: the scene lives in LDS and ray
state in registers, so HBM is idle and is not the bound.
For N > 1 the same block is filled from the counters of the WHOLE frame on one GPU (its useful lane-operations do not
depend on how the rows are dealt out): achieved = those lane-ops / the slowest rank's kernel time, peak = N x one GPU's.
`config4` (BASELINE.json's own multi-GPU config, N > 1 only) carries its own block from a config-4 counter pass.
`cpu_baseline` = the C oracle (a port, not the reference: the reference cannot be built here)
on rank 0's host cores over a bounded, evenly spread row sample of the same frame, for every N.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
N_CUS, SIMDS_PER_CU, SIMD_LANES, CLOCK_HZ = 256, 4, 32, 2.4e9
VALU_PEAK_TLANEOPS = N_CUS * SIMDS_PER_CU * SIMD_LANES * CLOCK_HZ / 1e12   # 78.6: one f32 lane-op per lane per cycle

# Lane-operations one unit of work costs when every lane of a wave does useful work: the per-lane VALU instruction counts of the
# kernel's bodies as they stood when this table was fixed (round 4; counted in the ISA).  FIXED from here on:
# `roofline.frac_work` = sum(units x cost) / kernel time / peak prices the WORK a frame contains (counted exactly by the
# COUNTERS instantiation, equal to the oracle's counters), so a faster kernel at equal work raises it -- unlike `frac`, which
# prices the lane-operations the kernel happened to EXECUTE and falls when an optimisation removes instructions.
WORK_COST = {"rays": 150.0,             # per raycast call: walk begin (1 / d, granule offsets), hit normal or sky normalize, path bookkeeping
             "interior_visits": 54.0,   # one interior step: 24 sub / mul, 8 min / max, 2 compares, address, push / pop moves
             "sphere_tests": 63.0,      # one leaf step: discriminant, correctly rounded sqrt and divide, accept
             "hits": 180.0,             # material lottery, second normalize, reflect / refract / Schlick, throughput
             "paths": 100.0,            # per sample: camera ray (two draws, normalize), gamma sqrt x 3, accumulation
             "ball_iterations": 43.0}   # one iteration of the rejection sampler: three hash draws, |p|^2, accept

WORKLOAD = dict(width=1920, height=1080, spp=64, bounces=8, scene_seed=1, random_seed=0.5)
WORKLOAD3 = dict(width=1920, height=1080, spp=256, bounces=50, scene_seed=1, random_seed=0.5)   # BASELINE.json configs[2]
WORKLOAD5 = dict(width=1920, height=1080, spp=64, bounces=8, scene_seed=1, random_seed=0.5)     # BASELINE.json configs[4]
PMC_WORKLOAD3_TAG = "rtiow_1920x1080_256spp_50b"
PMC_WORKLOAD5_TAG = "grid10k_1920x1080_64spp_8b"
WORKLOAD4 = dict(width=3840, height=2160, spp=1024, bounces=8, scene_seed=1, random_seed=0.5)   # BASELINE.json configs[3]
PMC_WORKLOAD_TAG = "cover_1920x1080_64spp_8b"
PMC_WORKLOAD4_TAG = "rtiow_3840x2160_1024spp_8b"


def bytes_alg(stats, width, rows):
    """what an uncached machine would move for this launch."""
    return (stats["rays"] * 96 + stats["node_pops"] * 48 + stats["interior_visits"] * 96 +
            stats["sphere_tests"] * 32 + stats["hits"] * 32 + width * rows * 16)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)   # the first ~7 frames after idle run 4 % slower (clock ramp)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="wall-time target of the CPU baseline sample")
    ap.add_argument("--no-pmc", action="store_true", help="skip the live rocprofv3 --pmc pass (N = 1 only)")
    ap.add_argument("--no-extras", action="store_true", help="skip first-frame / reseeded / moving-camera / config-4 measurements")
    ap.add_argument("--no-configs", action="store_true", help="skip the config 3 / config 5 blocks (N = 1)")
    ap.add_argument("--plan-strips", type=int, default=0, help="N > 1: deal the strips out to the ranks by measured cost (brt_plan_strips with this many "
                    "probe samples per pixel, outside the timed region; 0: strip s -> rank s %% N).  Config 4 in 8 shares 91.9 -> 89.1 ms, the headline config unchanged")
    ap.add_argument("--sync-steps", action="store_true", help="timed steps through the synchronous form of brt_render_part_device (a host round "
                                                              "trip per frame) instead of enqueueing them on the stream")
    return ap.parse_args(argv)


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start N ranks as CHILD processes (this parent has not
    imported torch or touched HIP, and never execs), relay rank 0's JSON line, exit with their code."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC for RCCL (see the environment notes)
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env)
    line = None
    for ln in proc.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    if proc.returncode == 0 and line is None:
        print("bench.py: the ranks ended without a JSON line", file=sys.stderr)
        return 1
    return proc.returncode


def load_tracer():
    """None, or the factory of a stand-in tracer.  The product tracer needs an MI355X; BRT_BENCH_TRACER=module:factory
    swaps in a stand-in (tests/stub_tracer.py: CPU tensors over gloo) so that the launch / gather / reporting
    logic of this file can be tested without a GPU.  A stub run says so in its JSON line (`data`)."""
    hook = os.environ.get("BRT_BENCH_TRACER")
    if hook:
        mod, fn = hook.split(":")
        return getattr(__import__(mod), fn)


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))

    import numpy as np
    import torch
    import torch.distributed as dist
    import bevyray_amd as brt
    from bevyray_amd.parallel import RcclGather, end_of_frame, frame_rows_of_part, gather_frame

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    stub = load_tracer()
    if stub is None:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: no HIP device (there is no CPU path)")
        if local_rank >= torch.cuda.device_count():
            raise SystemExit(f"bench.py --gpus {args.gpus}: rank {rank} has no GPU (this node shows {torch.cuda.device_count()})")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        backend = "nccl"
    else:
        dev = torch.device("cpu")
        backend = "gloo"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    plugin = brt.RaytracePlugin([local_rank]) if stub is None else stub(rank, world)
    node = plugin.node
    # the one collective of the path through the product boundary (brt_gather_rccl: ncclGather + de-interleave inside the library);
    # BRT_BENCH_GATHER=torch keeps torch.distributed's gather (which is also the fallback when the library's RCCL leg is not usable)
    rccl = None
    if stub is None and world > 1 and os.environ.get("BRT_BENCH_GATHER", "abi") != "torch":
        rccl = RcclGather.create(plugin, rank, world)

    def sync():
        if dev.type == "cuda":
            torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        sync()

    def all_max(x):
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def all_sum(x):
        t = torch.tensor([float(x)], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    def all_list(x):
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        if world == 1:
            return [float(x)]
        out = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        return [float(o.item()) for o in out]

    def run_workload(wl, scene_kind, steps, warmup, counters=True):
        """Times `steps` frames of one workload; returns a dict (meaningful on every rank after the reductions)."""
        W, H, spp, bounces = wl["width"], wl["height"], wl["spp"], wl["bounces"]
        scene = brt.generate_scene(scene_kind, wl["scene_seed"])
        # The scene goes up WITHOUT a BVH: the callee builds the tree (binned SAH), the integration INTEGRATION.md recommends --
        # the reference's own obvhs PLOC build (extract.rs:315-332) is then not needed at all.  `buffers` carries that same tree
        # (brt_build_bvh_sah is the builder brt_upload_scene uses) for the CPU oracle; the caller's-PLOC-tree variant is timed
        # below as an extra.
        buffers = brt.Buffers(scene.models, scene.materials, brt.build_bvh_sah(scene.models))
        upload = brt.Buffers(scene.models, scene.materials, None)
        cam_fn = brt.rtiow_camera if scene_kind == brt.SCENE_RTIOW_FINAL else brt.cover_camera   # configs 3 / 4: the book's view
        lvl, cam, win = cam_fn(W, H, spp, bounces, brt.Raytracing.Pure, wl["random_seed"])
        node.write_buffers(upload)              # scene resident in HBM before anything is timed
        if stub is None and world > 1:
            # every rank computes the same table from the same probe frame (deterministic): no second collective
            if args.plan_strips > 0:
                plugin.plan_strips(lvl, cam, win, W, H, world, args.plan_strips)
            else:
                plugin.set_strip_table(world, None)
        rows = brt.tile_rows(H, world)
        tile = torch.zeros((rows, W, 4), dtype=torch.float32, device=dev)
        sync()                                  # the zero fill ran on torch's stream, the trace kernel has its own
        gather_events = []

        def step(flags=0, timed=False, window=win):
            # the trace kernel runs on the context's own stream and the call returns when it is done
            st = node.render_part_device(lvl, cam, window, W, H, rank, world, tile.data_ptr(), flags=flags)
            if timed and dev.type == "cuda":   # gather + de-interleave run on torch's current stream: time them there
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            frame = gather_frame(tile, H, rank, world, node=node, rccl=rccl)
            if timed and dev.type == "cuda":
                e1.record()
                gather_events.append((e0, e1))
            end_of_frame(tile)                  # the RCCL send / the copy kernel are done before `tile` is reused
            return st, frame

        counted, frame = step(brt.FLAG_COUNTERS) if counters else (None, None)
        if counted is not None and stub is None:
            # lane-level iterations of the rejection sampler = active lanes summed over the executions of its section
            counted["ball_iterations"] = plugin.debug_profile()["ball"][1]
        for _ in range(warmup):
            st_w = step()[0]
        async_steps = dev.type == "cuda" and stub is None and not args.sync_steps and warmup >= 2
        barrier()
        t0 = time.perf_counter()
        kernel_ms, rays, variants, measuring = [], 0, [], 0
        if async_steps:
            # The K timed frames are ENQUEUED on torch's current stream (the caller-stream form of brt_render_part_device: no host round
            # trip per frame; the dispatch order and the kernel instantiation are the steady state the synchronous warm-up frames
            # left), each followed by its gather / de-interleave on the same stream; ONE synchronisation, at the barrier behind them.
            # Kernel time: HIP events on that stream around each launch.  Every frame has the same inputs, hence the same ray count
            # as the last warm-up frame, which the library counted.
            cur = torch.cuda.current_stream().cuda_stream
            evs = []
            for _ in range(steps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                st = node.render_part_device(lvl, cam, win, W, H, rank, world, tile.data_ptr(), stream=cur)
                e1.record()
                g0, g1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                g0.record()
                frame = gather_frame(tile, H, rank, world, node=node, rccl=rccl)
                g1.record()
                gather_events.append((g0, g1))
                evs.append((e0, e1))
                variants.append(st.get("kernel_variant", 0))
                measuring += st.get("measured_tile_costs", 0)
            barrier()
            elapsed = all_max(time.perf_counter() - t0)
            kernel_ms = [a.elapsed_time(b) for a, b in evs]
            rays = st_w["rays"] * steps
            st = dict(st_w, **{k: st[k] for k in ("kernel_variant", "measured_tile_costs") if k in st})
        else:
            for _ in range(steps):
                st, frame = step(timed=True)
                kernel_ms.append(st["kernel_ms"])
                rays += st["rays"]
                variants.append(st.get("kernel_variant", 0))
                measuring += st.get("measured_tile_costs", 0)
            barrier()
            elapsed = all_max(time.perf_counter() - t0)
        total_rays = all_sum(rays)
        res = {"W": W, "H": H, "spp": spp, "bounces": bounces, "buffers": buffers, "scene": scene, "upload": upload, "lvl": lvl, "cam": cam, "win": win,
               "frame": frame, "counted": counted, "elapsed": elapsed, "total_rays": total_rays, "steps": steps,
               "kernel_ms": float(np.mean(kernel_ms)), "kernel_ms_per_rank": all_list(float(np.mean(kernel_ms))),
               "rays_per_rank": all_list(rays / steps),
               "gather_ms": float(np.mean([a.elapsed_time(b) for a, b in gather_events])) if gather_events else 0.0,
               "step": step, "tile": tile, "last_stats": st, "variants": sorted(set(variants)), "measuring_frames": measuring,
               "async_steps": async_steps}
        return res

    # every rank is there and rank r renders on device ordinal r (one process per GPU: a launcher that put two ranks on one card, or
    # fewer ranks than --gpus, must not produce a line that looks like an N-GPU figure)
    ranks_seen = dist.get_world_size() if world > 1 else 1
    ordinals = [int(x) for x in all_list(float(torch.cuda.current_device() if stub is None else rank))]
    if ranks_seen != args.gpus or ordinals != list(range(args.gpus)):
        raise SystemExit(f"bench.py --gpus {args.gpus}: {ranks_seen} ranks on device ordinals {ordinals}")

    head = run_workload(WORKLOAD, brt.SCENE_COVER, args.steps, args.warmup)
    W, H, spp = head["W"], head["H"], head["spp"]

    # ---- outside the timed region: what a frame costs when the view or the seed is new ----------------
    extras = {}
    if not args.no_extras and stub is None:
        step = head["step"]
        # (a) a new seed every frame, as the reference draws one (extract.rs:72-73); the dispatch order of the
        #     view is reused (it does not depend on the seed)
        ks = []
        for i in range(8):
            w2 = brt.WindowExtract.extract_component(H, 0.05 + 0.11 * i)
            ks.append(step(window=w2)[0]["kernel_ms"])
        extras["reseeded_ms"] = [round(all_max(k), 3) for k in ks]
        # (b) the first frame of a view: no history (setting a knob -- to its own value -- forgets it)
        plugin.set_tuning("BRT_LPT", plugin.get_tuning("BRT_LPT")[0])
        st1 = step()[0]
        second = step()[0]["kernel_ms"]
        # kernel time of the first frame of a view = its dispatch-order pre-pass (2 spp) + the frame in that order
        extras["first_frame_ms"] = round(all_max(st1["kernel_ms"] + st1.get("prepass_ms", 0.0)), 3)
        extras["first_frame_prepass_ms"] = round(all_max(st1.get("prepass_ms", 0.0)), 3)
        extras["first_frame_call_wall_ms"] = round(all_max(st1["total_ms"]), 3)   # host wall time of that call (incl. building the order)
        extras["second_frame_ms"] = round(all_max(second), 3)
        # (c) the caller's own tree: the PLOC BVH that the reference's extract stage would hand over (extract.rs:315-332)
        node.write_buffers(head["scene"])
        ks = [step()[0]["kernel_ms"] for _ in range(8)]
        extras["caller_ploc_tree_ms"] = round(all_max(float(np.median(ks[3:]))), 3)
        node.write_buffers(head["upload"])
        for _ in range(3):
            step()
        # (d) an animated scene: the reference re-extracts and re-uploads every frame (extract.rs:299-336, README.md:17);
        #     one sphere moves a little per frame, the caller's BVH is rebuilt by the callee (NULL BVH upload)
        import copy
        moving = copy.deepcopy(head["scene"])
        ks, ups = [], []
        for i in range(12):
            moving.models["position"][7, 0] += np.float32(0.002)
            t0 = time.perf_counter()
            node.write_buffers(brt.Buffers(moving.models, moving.materials, None))
            ups.append((time.perf_counter() - t0) * 1e3)
            ks.append(step()[0]["kernel_ms"])
        extras["animated_scene_ms"] = [round(all_max(k), 3) for k in ks[4:]]
        extras["animated_scene_upload_wall_ms"] = round(float(np.median(ups)), 3)
        node.write_buffers(head["upload"])
        for _ in range(3):
            step()
        # (e) a moving camera: the reference's demo is a fly-camera app (src/main.rs:40) and re-extracts the camera every frame
        #     (extract.rs:118-157).  The cover camera orbits the origin 0.5 degrees per frame for 24 frames, a new seed each; every
        #     such frame measures its tile costs (in the LEAN instantiation) and runs in the order measured one frame earlier
        lvl0 = head["lvl"]
        ks, kinds = [], {"prepass": 0, "general": 0, "lean": 0, "measuring": 0}
        for i in range(1, 25):
            a = np.deg2rad(0.5 * i)
            pos = (13.0 * np.cos(a) - 3.0 * np.sin(a), 2.0, 13.0 * np.sin(a) + 3.0 * np.cos(a))
            cam_i = brt.CameraExtract.extract_component(brt.RaytracedCamera(level=brt.Raytracing.Pure, sample_count=spp, bounces=WORKLOAD["bounces"]),
                                                        brt.Transform(tuple(float(x) for x in pos), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0)),
                                                        brt.PerspectiveProjection(fov=0.4, aspect_ratio=W / H, near=0.1, far=1000.0))[1]
            st = node.render_part_device(lvl0, cam_i, brt.WindowExtract.extract_component(H, 0.03 + 0.04 * i), W, H, rank, world, head["tile"].data_ptr())
            ks.append(st["kernel_ms"] + st.get("prepass_ms", 0.0))
            kinds["prepass"] += int(st.get("prepass_ms", 0.0) > 0)
            kinds["lean" if st.get("kernel_variant", 0) in (1, 2) else "general"] += 1
            kinds["measuring"] += st.get("measured_tile_costs", 0)
        extras["moving_camera_ms"] = [round(all_max(k), 3) for k in ks]
        extras["moving_camera_median_ms"] = round(float(np.median([all_max(k) for k in ks])), 3)
        extras["moving_camera_frames"] = kinds
        for _ in range(3):
            step()
        # (f) the same re-upload-every-frame loop on the 10 004-sphere scene (config 5's): the callee builds the SAH tree ON THE GPU
        #     (brt_sah.hip) inside brt_upload_scene; round 3 built it on one host core (8-9 ms per upload)
        if world == 1:
            g = brt.generate_scene(brt.SCENE_STRESS_GRID, 1)
            gm = g.models.copy()
            node.write_buffers(brt.Buffers(gm, g.materials, None))
            lvl5, cam5, win5 = brt.cover_camera(W, H, spp, WORKLOAD["bounces"], brt.Raytracing.Pure, WORKLOAD["random_seed"])
            ks, ups = [], []
            for i in range(10):
                gm["position"][7, 0] += np.float32(0.002)
                t0 = time.perf_counter()
                node.write_buffers(brt.Buffers(gm, g.materials, None))
                ups.append((time.perf_counter() - t0) * 1e3)
                ks.append(node.render_part_device(lvl5, cam5, win5, W, H, rank, world, head["tile"].data_ptr())["kernel_ms"])
            extras["animated_scene_10k_ms"] = [round(k, 3) for k in ks[4:]]
            extras["animated_scene_10k_upload_wall_ms"] = round(float(np.median(ups[2:])), 3)
            extras["animated_scene_10k_gpu_bvh_build_ms"] = round(min(plugin.build_bvh_sah(gm)[1] for _ in range(3)), 3)
            node.write_buffers(head["upload"])
            for _ in range(3):
                step()
        # (g) what a following frame started before this one has ended would buy (lever (b)) -- measured beside
        #     `value`, whose timed frames run one after the other on one stream.  K frames asynchronously on ONE context and stream (no
        #     host round trip between frames: what the timed region does), then alternating between TWO contexts on two streams (the second frame's workgroups start
        #     on the CUs the first one's tail frees; each context has its own control block, order and pixel-state buffers)
        if world == 1:
            try:
                K = 24
                p2 = brt.RaytracePlugin([local_rank])
                p2.node.write_buffers(head["upload"])
                tile2 = torch.zeros_like(head["tile"])
                sync()
                lvl0, cam0, win0 = head["lvl"], head["cam"], head["win"]
                for _ in range(4):      # the second context learns its dispatch order (pre-pass, measuring frame) synchronously
                    p2.node.render_part_device(lvl0, cam0, win0, W, H, 0, 1, tile2.data_ptr())
                streams = [torch.cuda.Stream(), torch.cuda.Stream()]
                def run_async(ctxs):
                    sync()
                    t0 = time.perf_counter()
                    for i in range(K):
                        nd, tl, st = ctxs[i % len(ctxs)]
                        nd.render_part_device(lvl0, cam0, win0, W, H, 0, 1, tl.data_ptr(), stream=st.cuda_stream)
                    sync()
                    return (time.perf_counter() - t0) / K * 1e3
                one = [(node, head["tile"], streams[0])]
                two = [(node, head["tile"], streams[0]), (p2.node, tile2, streams[1])]
                run_async(one); run_async(two)
                extras["frames_in_flight"] = {"frames": K, "one_context_async_ms_per_frame": round(min(run_async(one) for _ in range(3)), 3),
                                              "two_contexts_alternating_ms_per_frame": round(min(run_async(two) for _ in range(3)), 3),
                                              "note": "beside `value` (frames enqueued on one stream, one after the other): what a following frame of a SECOND context, started before this one's tail has drained, would add -- nothing"}
                p2.close()
                for _ in range(3):
                    step()
            except Exception as e:   # noqa: BLE001  (a measurement beside the contract line must never take it down)
                extras["frames_in_flight"] = {"error": str(e)[:200]}
    cfg4 = None
    if not args.no_extras and world > 1:
        # config 2's longest pixel chains take ~5 ms whatever N is; BASELINE.json's own
        # multi-GPU config is 4K x 1024 spp, reported here beside the headline
        r4 = run_workload(WORKLOAD4, brt.SCENE_RTIOW_FINAL if stub is None else brt.SCENE_COVER, 2, 1, counters=False)
        cfg4 = {"workload": "RTIOW random spheres 3840x2160, 1024 spp, 8 bounces (BASELINE.json configs[3])",
                "value": r4["total_rays"] / r4["elapsed"] / 1e6, "unit": "Mrays/s", "ms_per_step": r4["elapsed"] / r4["steps"] * 1e3,
                "steps": r4["steps"], "warmup": 1, "kernel_ms_per_rank": r4["kernel_ms_per_rank"], "gather_ms": r4["gather_ms"]}

    cfg_blocks = {}
    if world == 1 and stub is None and not args.no_configs:
        # BASELINE.json's other one-GPU configs under the driver's own run: a few frames each, kernel time from HIP events, own
        # counter pass for the roofline, sampled rows against the oracle (the -m gpu suite checks the WHOLE frames)
        for key, wl, kind, tag, name in (("config3", WORKLOAD3, brt.SCENE_RTIOW_FINAL, PMC_WORKLOAD3_TAG,
                                          "RTIOW final scene 1920x1080, 256 spp, 50 bounces (BASELINE.json configs[2])"),
                                         ("config5", WORKLOAD5, brt.SCENE_STRESS_GRID, PMC_WORKLOAD5_TAG,
                                          "10 004-sphere grid 1920x1080, 64 spp, 8 bounces (BASELINE.json configs[4])")):
            r = run_workload(wl, kind, 3, 2)
            cfg_blocks[key] = (name, tag, r)
        node.write_buffers(head["upload"])

    if rank == 0:
        counted = head["counted"]
        my_rows = int((frame_rows_of_part(H, 0, world) >= 0).sum())
        alg = bytes_alg(counted, W, my_rows)
        kernel_ms = max(head["kernel_ms_per_rank"])          # the slowest rank's kernel bounds the frame
        roof = roofline_block(args, world, stub is not None, kernel_ms, alg, PMC_WORKLOAD_TAG, counted=counted, paths=W * H * spp)
        out = {
            "metric": "Mrays/s at 1920x1080, 64 spp, 8 bounces", "value": head["total_rays"] / head["elapsed"] / 1e6,
            "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["elapsed"] / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic" if stub is None else "synthetic (STUB tracer: no rays traced)",
            "config": {"workload": "README cover scene 1920x1080, 64 spp, 8 bounces (BASELINE.json configs[1])",
                       "spheres": int(len(head["buffers"].models)), "bvh_nodes": int(len(head["buffers"].bvh)),
                       "bvh": "built by the callee (brt_upload_scene without a BVH: binned SAH, INTEGRATION.md section 3); "
                              "`caller_ploc_tree_ms` = the same frame in the caller's PLOC tree",
                       "scene_seed": WORKLOAD["scene_seed"], "random_seed": WORKLOAD["random_seed"], "level": "Pure",
                       "parallelism": (f"8-row strips over {world} GPU(s), " + ("dealt out by measured cost (brt_plan_strips)" if args.plan_strips > 0 and world > 1 else "strip s -> rank s % N") +
                                       ", one RCCL gather per frame")},
            "n_ranks_seen": ranks_seen, "device_ordinals": ordinals,
            "rays_per_frame": head["total_rays"] / args.steps, "paths_per_frame": W * H * spp,
            "mpaths_per_s": W * H * spp * args.steps / head["elapsed"] / 1e6,
            "kernel_ms_per_rank": head["kernel_ms_per_rank"], "rays_per_frame_per_rank": head["rays_per_rank"],
            "gather_ms": head["gather_ms"],   # rank 0: RCCL gather (N > 1) + de-interleave copy kernel, per frame
            "gather_via": ("brt_gather_rccl (ncclGather + de-interleave inside libbevyray_amd.so)" if rccl is not None else
                           ("torch.distributed.gather + brt_deinterleave_device" if world > 1 else "brt_deinterleave_device (one rank)")),
            # the timed frames: which instantiation(s) of the trace kernel ran and how many of them measured tile costs.  A view whose
            # camera and scene stand still never measures again (round 4), so the timed window IS the amortised steady state; a moving
            # camera measures every frame: `moving_camera_ms`
            "steady_state": {"kernel_variants_timed": head["variants"], "measuring_frames_timed": head["measuring_frames"],
                             "amortised_ms_per_frame": head["elapsed"] / args.steps * 1e3,
                             "submission": ("the K timed frames are enqueued on one stream (caller-stream form of brt_render_part_device, each followed by its "
                                            "gather / de-interleave), one synchronisation at the closing barrier; frames run one after the other, none overlaps "
                                            "(`frames_in_flight`)") if head["async_steps"] else "one synchronous call per frame (a host round trip each)"},
            "roofline": roof,
            "kernel": {"lds_bytes": counted["lds_bytes"], "scene_in_lds": counted["scene_in_lds"],
                       "workgroups": counted["n_workgroups"], "threads": counted["threads_per_workgroup"]},
        }
        out.update(extras)
        # both trees side by side in the contract's config block: `value` is measured in the tree the callee
        # builds; the same frame in the tree the reference's extract stage would hand over (PLOC, 0.1 pads):
        if "caller_ploc_tree_ms" in extras and extras["caller_ploc_tree_ms"] > 0:
            out["config"]["caller_tree_ms"] = extras["caller_ploc_tree_ms"]
            out["config"]["caller_tree_value"] = round(head["total_rays"] / args.steps / (extras["caller_ploc_tree_ms"] * 1e-3) / 1e6, 1)
            out["config"]["caller_tree_note"] = ("Mrays/s (kernel time, this rank's share) of the same frame in the caller's PLOC tree, 0.1 pads "
                                                 "(extract.rs:220-227, 315-332); `value` is in the callee-built tree")
        if cfg4 is not None:
            cfg4["roofline"] = roofline_block(args, world, stub is not None, max(cfg4["kernel_ms_per_rank"]), None, PMC_WORKLOAD4_TAG)
            out["config4"] = cfg4
        for key, (name, tag, r) in cfg_blocks.items():
            c = r["counted"]
            alg_c = bytes_alg(c, r["W"], r["H"])
            blk = {"workload": name, "spheres": int(len(r["buffers"].models)), "steps": r["steps"], "warmup": 2,
                   "value": r["total_rays"] / r["elapsed"] / 1e6, "unit": "Mrays/s", "ms_per_step": r["elapsed"] / r["steps"] * 1e3,
                   "kernel_ms": r["kernel_ms"], "rays_per_frame": r["total_rays"] / r["steps"], "scene_in_lds": c["scene_in_lds"],
                   "roofline": roofline_block(args, 1, False, r["kernel_ms"], alg_c, tag, counted=c, paths=r["W"] * r["H"] * r["spp"],
                                              memory_side=(key == "config5")),
                   "sampled_rows_bit_exact": sampled_rows_exact(r)}
            out[key] = blk
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(head["buffers"], head["lvl"], head["cam"], head["win"], W, H, head["frame"],
                                               args.cpu_seconds, is_stub=stub is not None)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
    if rccl is not None:
        rccl.close()
    if world > 1:
        dist.destroy_process_group()
    plugin.close()


# ---- roofline ---------------------------------------------------------------------------------------------

PMC_PASSES = [
    "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES",
    "FETCH_SIZE",
    "WRITE_SIZE",
]
# the memory side of a launch (config 5: BASELINE.json's "BVH-bandwidth bound, rocprof HBM GB/s focus"): where the wave
# cycles go (issuing / issue-stalled / parked in s_waitcnt), the vector-memory path (L1 accesses, L1 -> L2 requests, texture
# addresser busy and stalled by the cache) and the L2's hit rate -- one rocprofv3 pass each (8 SQ slots, 4 TCC slots per pass)
PMC_PASSES_MEMORY = [
    "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE",
    "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_BUSY_avr",
    "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum",
]


def under_a_profiler():
    """True when this process already runs under rocprofv3 / rocprof (its tool library is preloaded and its ROCP* /
    ROCPROF* variables are set): a nested counter pass would inherit them -- conflicting tool libraries at worst, a
    confusing note at best -- so the live pass is skipped and the committed, hash-checked summary is used."""
    if any(k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) for k in os.environ):
        return True
    return any(t in os.environ.get(v, "") for v in ("LD_PRELOAD", "HSA_TOOLS_LIB") for t in ("rocprofiler", "roctracer", "rocprof"))


def live_pmc(timeout_s=90.0, workload=PMC_WORKLOAD_TAG, passes=None):
    """rocprofv3 --pmc over scripts/pmc_frame.py (the workload named by `workload`, torch-free), one child process per
    counter set (SQ: 8 slots; FETCH_SIZE and WRITE_SIZE do not fit one TCC pass).  Returns {counter: value of
    the LAST dispatch of the production kernel} or None.  The program comes directly after `--` (no shell, no
    env wrapper: the profiler has initialised the GPU by then)."""
    import csv
    import glob
    import shutil
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, "rocprofv3 not found"
    if under_a_profiler():
        return None, "this run is itself under a profiler: no nested counter pass"
    tmp = tempfile.mkdtemp(prefix="brt_pmc_", dir="/tmp")
    env = {k: v for k, v in os.environ.items()
           if not k.startswith(("ROCPROF", "ROCP_", "ROCTRACER")) and k not in ("LD_PRELOAD", "HSA_TOOLS_LIB", "BRT_BENCH_TRACER")}
    env.update(TMPDIR="/tmp", BRT_NO_TORCH="1", BRT_PMC_WORKLOAD=workload)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):     # the child is a one-GPU program, whatever launched this rank
        env.pop(k, None)
    got = {}
    t_end = time.time() + timeout_s
    try:
        for i, counters in enumerate(passes or PMC_PASSES):
            left = t_end - time.time()
            if left < 10:
                return (got or None), "time budget of the PMC passes used up"
            out_dir = os.path.join(tmp, f"pass{i}")
            cmd = [rocprof, "--pmc"] + counters.split() + ["--output-format", "csv", "-d", out_dir, "--",
                                                           sys.executable, os.path.join(ROOT, "scripts", "pmc_frame.py")]
            try:
                proc = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                                      timeout=left)
            except subprocess.TimeoutExpired:
                return (got or None), f"pass {i} timed out"
            if proc.returncode != 0:
                return (got or None), f"pass {i}: rocprofv3 exit {proc.returncode}: {proc.stdout[-300:]}"
            per = {}
            for f in glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if "k_trace_persistent" not in r["Kernel_Name"]:
                        continue
                    d = per.setdefault(int(r["Dispatch_Id"]), {})
                    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            if not per:
                return (got or None), f"pass {i}: no k_trace_persistent dispatch in the counter CSV"
            got.update(per[max(per)])
        return got, None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def sampled_rows_exact(r, n_rows=4):
    """A few rows of the last timed frame of a config block against the oracle, bit for bit (the oracle as checker)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_loader
    oracle = oracle_loader.load()
    H = r["H"]
    g = r["frame"].cpu().numpy()
    ok = True
    for y in [int(x) for x in np.linspace(H // 3, H - 5, n_rows)]:
        want, _ = oracle.render(r["buffers"], r["lvl"], r["cam"], r["win"], r["W"], H, rows=(y, y + 1), threads=oracle_loader.usable_cores())
        ok = ok and bool(np.array_equal(g[y].view(np.uint32), want[y].view(np.uint32)))
    return ok


def roofline_block(args, world, is_stub, kernel_ms, alg, workload, counted=None, paths=None, memory_side=False):
    """The roofline object of one workload.  kernel_ms: the slowest rank's mean kernel time.  Counters: of the WHOLE frame
    on ONE GPU (a live rocprofv3 --pmc pass made now by rank 0 in a one-GPU child process, or the committed summary taken on
    the same device code); for N > 1 the frame's lane-operations are set against N GPUs' peak."""
    from bevyray_amd import _lib
    code_hash = None if is_stub else _lib.kernel_code_hash()
    pmc, source, why = None, None, None
    head = workload == PMC_WORKLOAD_TAG
    if not is_stub and not args.no_pmc and workload != PMC_WORKLOAD4_TAG:
        # (config 4 renders for seconds per frame and minutes per counter pass: its block takes the committed, hash-checked summary
        #  directly instead of a live pass that would hit its time budget; config 5 also takes the memory-side sets)
        pmc, why = live_pmc(workload=workload, passes=PMC_PASSES + (PMC_PASSES_MEMORY if memory_side else []),
                            timeout_s=90.0 if head else (150.0 if memory_side else 75.0))
        if pmc and "SQ_INSTS_VALU" in pmc:
            source = "live: rocprofv3 --pmc passes made by this run after the timed region (scripts/pmc_frame.py, one GPU, last dispatch)"
        else:
            pmc = None
    if pmc is None and not is_stub:
        path = os.path.join(ROOT, "profiles", {PMC_WORKLOAD_TAG: "pmc_summary.json", PMC_WORKLOAD4_TAG: "pmc_summary_config4.json",
                                               PMC_WORKLOAD3_TAG: "pmc_summary_config3.json",
                                               PMC_WORKLOAD5_TAG: "pmc_summary_config5.json"}[workload])
        try:
            rec = json.load(open(path))
            if rec.get("workload") == workload and rec.get("kernel_code_hash") == code_hash:
                pmc = rec["counters"]
                source = f"{os.path.relpath(path, ROOT)} (same device code {code_hash})"
            else:
                why = (why + "; " if why else "") + f"{os.path.relpath(path, ROOT)} was taken on device code {rec.get('kernel_code_hash')}, this is {code_hash}"
        except Exception as e:   # noqa: BLE001
            why = (why + "; " if why else "") + f"{os.path.relpath(path, ROOT)}: {e}"
    secs = kernel_ms * 1e-3
    peak = VALU_PEAK_TLANEOPS * world
    roof = {"bound": "valu", "achieved": None, "peak": peak, "unit": "Tlane-op/s", "frac": None, "frac_work": None, "traffic": None,
            "kernel": "k_trace_persistent", "kernel_ms": kernel_ms, "kernel_code_hash": code_hash, "n_gpus": world,
            "algorithmic_bytes_per_launch": alg, "algorithmic_GBs": alg / secs / 1e9 if (alg and secs > 0) else None,
            "hbm_peak_GBs": HBM_PEAK_GBS * world, "hbm_frac_measured": None,
            "note": "bound = f32 vector pipes: achieved = VALU instructions x mean active lanes of the whole frame (counted on one GPU) / "
                    "the slowest rank's kernel time; peak = n_gpus x 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz (the guide's 2-cycle wave64 issue; "
                    "tests/tools/issue_bench.hip measures 2.2-2.4 cycles for the f32 add/mul/fma group in a pure stream and 4 for min/max, "
                    "compares, v_cndmask, conversions and integer multiplies, 8 for v_rcp/v_sqrt -- DESIGN.md section 5). The scene is LDS "
                    "resident and ray state lives in registers, so HBM only sees the scene load per workgroup, one 16-B store per pixel and the 32-B pixel states that change hands between the two half-sample jobs of a tile (two wide agent-scope stores / loads per pixel, DESIGN.md section 6): "
                    "`traffic` (measured HBM bytes of the frame) / kernel time is `hbm_frac_measured` of the HBM peak; the SURVEY 8(d) "
                    "algorithmic bytes are informational (they are served from LDS/registers)",
            "counter_source": source, "counter_note": why}
    if world > 1:
        roof["estimate"] = ("N > 1: `achieved` is an ESTIMATE -- the counters are those of the WHOLE frame on one GPU (its useful lane-operations do "
                            "not depend on how the rows are dealt out) set against the slowest rank's kernel time and N GPUs' peak; per-rank "
                            "counters are not collected")
    if counted is not None and secs > 0 and "ball_iterations" in counted:
        units = {k: float(counted[k]) for k in ("rays", "interior_visits", "sphere_tests", "hits", "ball_iterations")}
        units["paths"] = float(paths)
        work = sum(units[k] * WORK_COST[k] for k in WORK_COST)
        roof["work"] = {"units_per_launch": units, "lane_ops_per_unit": WORK_COST, "lane_ops_per_launch": work,
                        "note": "frac_work = (exactly counted work x a FIXED lane-op cost per unit) / kernel time / peak: rises when the kernel "
                                "gets faster at equal work; `frac` prices the lane-operations actually executed (counters) and falls when "
                                "an optimisation removes instructions"}
        roof["achieved_work"] = work / secs / 1e12
        roof["frac_work"] = roof["achieved_work"] / peak
    if pmc:
        valu, act, thr = pmc.get("SQ_INSTS_VALU"), pmc.get("SQ_ACTIVE_INST_VALU"), pmc.get("SQ_THREAD_CYCLES_VALU")
        lanes = thr / act if act else None
        if valu and lanes and secs > 0:
            roof["achieved"] = valu * lanes / secs / 1e12
            roof["frac"] = roof["achieved"] / peak
            simd_cycles = secs * CLOCK_HZ * N_CUS * SIMDS_PER_CU * world
            n_inst = sum(pmc.get(k, 0.0) for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH"))
            roof["issue"] = {"valu_instructions_per_launch": valu, "instructions_per_launch": n_inst,
                             "active_lanes_per_valu": lanes, "valu_per_simd_cycle": valu / simd_cycles,
                             "valu_issue_frac": valu / simd_cycles / 0.5, "instructions_per_simd_cycle": n_inst / simd_cycles}
        if "FETCH_SIZE" in pmc and "WRITE_SIZE" in pmc:
            # rocprofv3 reports KB; on gfx950 FETCH_SIZE counts 64 B per 128-B request: doubled (MI355X_MICROARCH.md, HBM)
            roof["traffic"] = (2.0 * pmc["FETCH_SIZE"] + pmc["WRITE_SIZE"]) * 1024.0
            roof["hbm_frac_measured"] = roof["traffic"] / secs / 1e9 / (HBM_PEAK_GBS * world) if secs > 0 else None
        if memory_side and "SQ_WAIT_ANY" in pmc:
            wc = pmc.get("SQ_WAVE_CYCLES") or 0.0
            mem = {"wave_cycles_frac": {"issuing": pmc.get("SQ_ACTIVE_INST_ANY", 0.0) / wc if wc else None,
                                        "issue_stalled": pmc.get("SQ_WAIT_INST_ANY", 0.0) / wc if wc else None,
                                        "parked_in_waitcnt": pmc["SQ_WAIT_ANY"] / wc if wc else None},
                   "vmem_read_instructions": pmc.get("SQ_INSTS_VMEM_RD"), "lds_bank_conflict_frac": (pmc.get("SQ_LDS_BANK_CONFLICT", 0.0) / pmc["SQ_LDS_IDX_ACTIVE"]) if pmc.get("SQ_LDS_IDX_ACTIVE") else None}
            if "TCP_TOTAL_CACHE_ACCESSES_sum" in pmc:
                mem.update({"l1_accesses": pmc["TCP_TOTAL_CACHE_ACCESSES_sum"], "l1_to_l2_read_requests": pmc.get("TCP_TCC_READ_REQ_sum"),
                            "l1_hit_frac": 1.0 - pmc.get("TCP_TCC_READ_REQ_sum", 0.0) / pmc["TCP_TOTAL_CACHE_ACCESSES_sum"] if pmc["TCP_TOTAL_CACHE_ACCESSES_sum"] else None,
                            "l1_pending_stall_cycles": pmc.get("TCP_PENDING_STALL_CYCLES_sum"),
                            "ta_busy_avr_cycles": pmc.get("TA_BUSY_avr"), "ta_addr_stalled_by_tc_cycles": pmc.get("TA_ADDR_STALLED_BY_TC_CYCLES_sum"),
                            "l2_read_GBs_at_64B_per_request": pmc.get("TCP_TCC_READ_REQ_sum", 0.0) * 64.0 / secs / 1e9 if secs > 0 else None})
            if "TCC_HIT_sum" in pmc and (pmc["TCC_HIT_sum"] + pmc.get("TCC_MISS_sum", 0.0)) > 0:
                mem["l2_hit_frac"] = pmc["TCC_HIT_sum"] / (pmc["TCC_HIT_sum"] + pmc.get("TCC_MISS_sum", 0.0))
                mem["l2_requests"] = pmc.get("TCC_REQ_sum")
            roof["memory_side"] = mem
    return roof


def cpu_baseline(buffers, lvl, cam, win, W, H, gpu_frame, target_seconds, is_stub=False):
    """The C oracle on this host's cores over every `row_step`-th row of the same frame; also
    checks those rows of the GPU frame bit for bit (the oracle as checker, never as product)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_loader
    oracle = oracle_loader.load()
    cores = oracle_loader.usable_cores()
    probe_rows = list(range(3, H, max(1, H // 8)))[:8]
    t0 = time.perf_counter()
    _, cnt = oracle.render(buffers, lvl, cam, win, W, H, rows=(3, H), row_step=max(1, H // 8), threads=cores)
    probe = max(time.perf_counter() - t0, 1e-3)
    per_row = probe / max(1, len(probe_rows))
    n_rows = int(max(8, min(H, target_seconds / per_row)))
    row_step = max(1, H // n_rows)
    t0 = time.perf_counter()
    frame, cnt = oracle.render(buffers, lvl, cam, win, W, H, rows=(0, H), row_step=row_step, threads=cores)
    dt = time.perf_counter() - t0
    sampled = np.arange(0, H, row_step)
    g = gpu_frame.cpu().numpy()
    exact = None if is_stub else bool(np.array_equal(g[sampled].view(np.uint32), frame[sampled].view(np.uint32)))
    return {"value": cnt["rays"] / dt / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"every {row_step}th row of the same 1920x1080x64spp frame ({len(sampled)} rows, {cnt['rays']} rays, "
                      f"{dt:.1f} s wall on {cores} threads = the CPUs granted to this process; {os.cpu_count()} logical CPUs on the host); "
                      f"scalar C restatement of the WGSL loop (no Rust toolchain here)",
            "gpu_rows_bit_exact": exact}


if __name__ == "__main__":
    main()
