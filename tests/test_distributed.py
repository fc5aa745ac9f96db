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


def _worker(rank, world, port, width, height, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rows = frame_rows_of_part(height, rank, world)
        assert len(rows) == brt.tile_rows(height, world)
        tile = _pattern(rows, width)
        frame = gather_frame(tile, height, rank, world)
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


def test_single_rank_is_identity():
    rows = frame_rows_of_part(24, 0, 1)
    tile = _pattern(rows, 8)
    frame = gather_frame(tile, 24, 0, 1)
    assert torch.equal(frame, _pattern(np.arange(24), 8))
