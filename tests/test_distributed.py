"""The N>1 path on CPU: world_size-2 (and 3) gloo runs of the row tiling + the single gather.

No GPU here, so the ranks cannot trace; each rank fills its tile with a pure function of the
FRAME coordinates (what a correct render_part_device would put there: row k*8+r of part p is
frame row (k*n+p)*8+r).  The test then checks that gather_frame() reassembles exactly the
full-frame pattern on rank 0 -- the same code path bench.py uses with RCCL, with gloo instead.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import bevyray_amd as brt
from bevyray_amd.parallel import frame_rows_of_part, gather_frame


def _pattern(rows, width):
    """value at (frame row y, column x, channel c) = y * 10000 + x * 4 + c; padding rows = -1"""
    y = torch.from_numpy(rows.astype(np.float32))[:, None, None]
    x = torch.arange(width, dtype=torch.float32)[None, :, None]
    c = torch.arange(4, dtype=torch.float32)[None, None, :]
    t = y * 10000.0 + x * 4.0 + c
    t[torch.from_numpy(rows < 0)] = -1.0
    return t.contiguous()


def _table(height, world, seed):
    """a strip table (brt_set_strip_table): the parts shuffled inside every group of `world` strips; seed None: no table"""
    if seed is None:
        return None
    rng = np.random.default_rng(seed)
    strips = (height + 7) // 8
    t = np.zeros(strips, np.uint32)
    for g in range(0, strips, world):
        n = min(world, strips - g)
        t[g:g + n] = rng.permutation(world)[:n]
    return t


def _worker(rank, world, port, width, height, out_path, table_seed=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        table = _table(height, world, table_seed)          # (every rank makes the same table, as brt_plan_strips does)
        rows = frame_rows_of_part(height, rank, world, table)
        assert len(rows) == brt.tile_rows(height, world)
        tile = _pattern(rows, width)
        frame = gather_frame(tile, height, rank, world, table=table)
        if rank == 0:
            want = _pattern(np.arange(height), width)
            ok = frame is not None and frame.shape == (height, width, 4) and torch.equal(frame, want)
            with open(out_path, "w") as f:
                f.write("ok" if ok else "mismatch")
        else:
            assert frame is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("world,width,height", [(2, 40, 36), (2, 17, 45), (3, 24, 100)])
def test_gather_frame_gloo(tmp_path, world, width, height):
    out = tmp_path / "result.txt"
    mp.spawn(_worker, args=(world, _free_port(), width, height, str(out)), nprocs=world, join=True)
    assert out.read_text() == "ok"


@pytest.mark.parametrize("world,width,height,seed", [(2, 24, 45, 1), (3, 16, 100, 2)])
def test_gather_frame_gloo_with_a_strip_table(tmp_path, world, width, height, seed):
    """The strips dealt out by a table (brt_set_strip_table / brt_plan_strips: a permutation of the parts inside every group of `world`
    strips) instead of s % world: same tiles, same single gather, the frame assembled through the table."""
    out = tmp_path / "result.txt"
    mp.spawn(_worker, args=(world, _free_port(), width, height, str(out), seed), nprocs=world, join=True)
    assert out.read_text() == "ok"


def test_strip_table_rows_partition_the_frame_and_bad_tables_are_refused():
    from bevyray_amd.parallel import check_strip_table
    for height, world, seed in ((45, 2, 3), (100, 3, 4), (1080, 8, 5), (2160, 8, 6), (7, 4, 7)):
        t = _table(height, world, seed)
        seen = np.zeros(height, np.int64)
        for p in range(world):
            rows = frame_rows_of_part(height, p, world, t)
            assert len(rows) == brt.tile_rows(height, world)
            v = rows[rows >= 0]
            seen[v] += 1
            # the k-th local strip of a part lies in group k (the tile layout of s % world)
            k = np.flatnonzero(rows >= 0) // 8
            assert np.array_equal(v // 8 // world, k)
        assert np.all(seen == 1)
        assert np.array_equal(np.concatenate([frame_rows_of_part(height, p, world, None) for p in range(world)]) >= 0,
                              np.concatenate([frame_rows_of_part(height, p, world, t) for p in range(world)]) >= 0) or height % (8 * world) != 0
    bad = _table(100, 3, 1).copy(); bad[1] = bad[0]
    with pytest.raises(ValueError):
        check_strip_table(bad, 100, 3)
    with pytest.raises(ValueError):
        check_strip_table(np.zeros(5, np.uint32), 100, 3)


def test_planned_strip_tables_are_valid_and_balance_the_parts():
    """The rule of brt_plan_strips on given costs (brt_host_plan_strips, host arithmetic): a valid table -- a permutation of the parts inside
    every group --, deterministic, and on costs that fall off from the bottom of the frame to the top (ground below, sky above, as the
    benchmark views) the dearest part is no dearer than under s % n_parts and within one strip of the mean."""
    from bevyray_amd import _lib
    from bevyray_amd.parallel import check_strip_table
    lib = _lib.load()
    rng = np.random.default_rng(11)
    for height, world in ((1080, 8), (2160, 8), (1080, 3), (45, 2), (149, 4)):
        strips = (height + 7) // 8
        cost = (np.linspace(1.0, 6.0, strips) ** 2 * 1e6 * rng.uniform(0.8, 1.25, strips)).astype(np.uint64)
        t = np.zeros(strips, np.uint32)
        assert lib.brt_host_plan_strips(cost.ctypes.data, strips, world, t.ctypes.data) == 0
        check_strip_table(t, height, world)
        t2 = np.zeros(strips, np.uint32)
        assert lib.brt_host_plan_strips(cost.ctypes.data, strips, world, t2.ctypes.data) == 0 and np.array_equal(t, t2)
        planned = np.array([cost[t == p].sum() for p in range(world)], np.float64)
        inter = np.array([cost[np.arange(strips) % world == p].sum() for p in range(world)], np.float64)
        assert planned.sum() == inter.sum()
        assert planned.max() <= inter.max()
        assert planned.max() - planned.mean() <= float(cost.max())
    assert lib.brt_host_plan_strips(None, 10, 2, t.ctypes.data) == -1


def test_single_rank_is_identity():
    rows = frame_rows_of_part(24, 0, 1)
    tile = _pattern(rows, 8)
    frame = gather_frame(tile, 24, 0, 1)
    assert torch.equal(frame, _pattern(np.arange(24), 8))


class _StubRcclPlugin:
    """what RcclGather.create needs of a RaytracePlugin, with a rank that cannot resolve librccl / cannot create its communicator"""

    def __init__(self, rank, fail_probe_on, fail_create_on, log):
        self.rank, self.fail_probe_on, self.fail_create_on, self.log = rank, fail_probe_on, fail_create_on, log

    def rccl_unique_id(self):
        if self.rank == self.fail_probe_on:
            raise RuntimeError("librccl not found")
        return bytes([self.rank]) * 128

    def rccl_comm_create(self, uid, rank, world):
        self.log.append(("create", uid[0]))
        if self.rank == self.fail_create_on:
            raise RuntimeError("device error")
        return 1000 + rank

    def rccl_comm_destroy(self, comm):
        self.log.append(("destroy", comm))


def _create_worker(rank, world, port, fail_probe_on, fail_create_on, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bevyray_amd.parallel import RcclGather
        log = []
        g = RcclGather.create(_StubRcclPlugin(rank, fail_probe_on, fail_create_on, log), rank, world)
        with open(os.path.join(out_dir, f"r{rank}.txt"), "w") as f:
            f.write(repr((None if g is None else g.comm, log)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("fail_probe_on,fail_create_on", [(-1, -1), (1, -1), (0, -1), (-1, 1)])
def test_rccl_gather_create_every_rank_takes_the_same_path(tmp_path, fail_probe_on, fail_create_on):
    """RcclGather.create probed librccl on rank 0 only -- a rank != 0 without it fell back alone and left the others in
    ncclCommInitRank.  Now every rank probes, the flags are min-reduced before and after the communicator is created, and all
    ranks return a gather object or all return None (a communicator a rank did get is given back)."""
    import ast
    mp.spawn(_create_worker, args=(2, _free_port(), fail_probe_on, fail_create_on, str(tmp_path)), nprocs=2, join=True)
    res = [ast.literal_eval((tmp_path / f"r{r}.txt").read_text()) for r in range(2)]
    if fail_probe_on < 0 and fail_create_on < 0:
        assert [r[0] for r in res] == [1000, 1001]
        assert res[0][1] == [("create", 0)] and res[1][1] == [("create", 0)]          # rank 0's id is the communicator's
    else:
        assert [r[0] for r in res] == [None, None]
        if fail_probe_on >= 0:
            assert res[0][1] == [] and res[1][1] == []                                # nobody entered ncclCommInitRank
        else:
            assert res[0][1] == [("create", 0), ("destroy", 1000)] and res[1][1] == [("create", 0)]


def _run_bench(extra_env, *argv):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BRT_BENCH_TRACER="stub_tracer:make",
               PYTHONPATH=os.path.join(root, "tests") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env)
    proc = subprocess.run([sys.executable, os.path.join(root, "bench.py"), *argv], capture_output=True, text=True, env=env,
                          timeout=600)
    assert proc.returncode == 0, proc.stdout + proc.stderr
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, proc.stdout
    return json.loads(lines[0])


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the driver's command shape): the parent starts the
    two ranks itself (gloo + a stub tracer here), relays ONE JSON line and the line describes both ranks."""
    out = _run_bench({}, "--gpus", "2", "--steps", "2", "--warmup", "1", "--cpu-seconds", "1")
    assert out["n_gpus"] == 2 and out["n_ranks_seen"] == 2 and out["steps"] == 2
    assert out["scaling"] == "strong" and out["unit"] == "Mrays/s"
    assert len(out["kernel_ms_per_rank"]) == 2 and out["kernel_ms_per_rank"] == [1.0, 2.0]
    # 1080 rows = 135 strips: rank 0 gets 68 of them, rank 1 67; one "ray" per pixel sample in the stub
    assert out["rays_per_frame_per_rank"] == [68 * 8 * 1920 * 64, 67 * 8 * 1920 * 64]
    assert out["rays_per_frame"] == 1920 * 1080 * 64
    assert "STUB" in out["data"] and out["roofline"]["frac"] is None
    assert out["config4"]["steps"] == 2 and len(out["config4"]["kernel_ms_per_rank"]) == 2
    # an N > 1 line is gradeable: the roofline names its bound, unit and the N-GPU peak (the stub has no counters, so
    # `achieved` stays null), config 4 carries its own block, and rank 0 reports the CPU baseline
    for roof in (out["roofline"], out["config4"]["roofline"]):
        assert roof["bound"] == "valu" and roof["unit"] == "Tlane-op/s" and roof["n_gpus"] == 2
        assert abs(roof["peak"] - 2 * 256 * 4 * 32 * 2.4e9 / 1e12) < 1e-6
        assert set(("achieved", "frac", "traffic", "kernel_ms")) <= set(roof)
    assert out["roofline"]["kernel_ms"] == 2.0          # the slowest rank
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "Mrays/s" and cb["value"] > 0 and cb["cores"] >= 1 and "sample" in cb


def test_bench_single_rank_line_has_the_contract_keys():
    out = _run_bench({}, "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["n_ranks_seen"] == 1 and out["vs_baseline"] is None
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(out["roofline"])
    assert out["roofline"]["bound"] == "valu" and "workload" in out["config"] and "model" not in out["config"]
