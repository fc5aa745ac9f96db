"""Row tiling of one frame over ranks and the single collective of the path.

The frame is cut into strips of STRIP_ROWS (= 8) rows; strip s belongs to part s % n_parts
(interleaved, because contiguous bands are badly imbalanced: sky rows end after one ray).
Every rank traces its strips into a dense tile (bevyray_amd.h: brt_render_part_device) and the
tiles meet on rank 0 in ONE gather over RCCL/xGMI (torch.distributed backend "nccl"); rank 0
de-interleaves with a copy kernel.  Pixels are independent (seeds depend on absolute pixel
coordinates only, reference raytrace.wgsl:95), so there is no other exchange.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from .raytracing import STRIP_ROWS, tile_rows


def check_strip_table(table, height: int, n_parts: int) -> np.ndarray:
    """A strip table (include/bevyray_amd.h brt_set_strip_table): table[s] = the part of frame strip s, a permutation of the parts inside
    every group of n_parts consecutive strips.  Returns it as an array, raises ValueError if it is not one."""
    t = np.asarray(table, np.int64)
    strips = (height + STRIP_ROWS - 1) // STRIP_ROWS
    if t.shape != (strips,) or t.min(initial=0) < 0 or t.max(initial=0) >= n_parts:
        raise ValueError("strip table: one part < n_parts per strip of the frame")
    for g in range(0, strips, n_parts):
        grp = t[g:g + n_parts]
        if len(set(grp.tolist())) != len(grp):
            raise ValueError(f"strip table: group {g // n_parts} holds a part twice")
    return t


def frame_rows_of_part(height: int, part: int, n_parts: int, table=None) -> np.ndarray:
    """Frame row of every tile row of `part` (-1 for padding rows past the frame).  table: a strip table (brt_set_strip_table /
    brt_plan_strips), or None: strip s belongs to part s % n_parts.  Either way the part's k-th local strip lies in group k."""
    rows = np.full(tile_rows(height, n_parts), -1, np.int64)
    strips = (height + STRIP_ROWS - 1) // STRIP_ROWS
    t = None if table is None else check_strip_table(table, height, n_parts)
    for s in range(strips):
        if (s % n_parts if t is None else int(t[s])) != part:
            continue
        k = s // n_parts
        for r in range(STRIP_ROWS):
            y = s * STRIP_ROWS + r
            if y < height:
                rows[k * STRIP_ROWS + r] = y
    return rows


class RcclGather:
    """The gather of the path through the C ABI (brt_gather_rccl: ncclGather + de-interleave inside libbevyray_amd.so) -- what a
    host without an RCCL binding of its own calls; here the communicator's unique id travels over the
    torch.distributed group that is up anyway.  `RcclGather.create` returns None when the library's RCCL leg is not usable
    (every rank decides the same way, so that nobody waits in ncclCommInitRank alone); gather_frame then falls back to
    torch.distributed.gather."""

    def __init__(self, plugin, comm: int, rank: int, world: int):
        self.plugin, self.comm, self.rank, self.world = plugin, comm, rank, world

    @staticmethod
    def create(plugin, rank: int, world: int, group=None):
        import torch
        import torch.distributed as dist

        def all_ok(ok: bool) -> bool:
            """every rank takes the same path: min over the ranks' flags (a CPU tensor under gloo, a device tensor under nccl)"""
            if world == 1:
                return ok
            dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
            t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
            return bool(int(t.item()))

        # EVERY rank proves that librccl resolves (dlopen) in its process -- a rank without it would otherwise fall back alone and
        # leave the others waiting in ncclCommInitRank
        uid, ok = b"", True
        try:
            uid = plugin.rccl_unique_id()
        except Exception:                              # noqa: BLE001
            ok = False
        if not all_ok(ok):
            return None
        if world > 1:
            box = [uid]
            dist.broadcast_object_list(box, src=0, group=group)      # rank 0's id is the communicator's
            uid = box[0]
        comm = 0
        try:
            comm = plugin.rccl_comm_create(uid, rank, world)
        except Exception:                              # noqa: BLE001
            ok = False
        if not all_ok(ok):                             # (a device error on one rank: the ranks that did get a communicator give it back)
            if comm:
                try:
                    plugin.rccl_comm_destroy(comm)
                except Exception:                      # noqa: BLE001
                    pass
            return None
        return RcclGather(plugin, comm, rank, world)

    def close(self):
        if self.comm:
            self.plugin.rccl_comm_destroy(self.comm)
            self.comm = 0


def gather_frame(tile, height: int, rank: int, world: int, node=None, group=None, rccl: Optional["RcclGather"] = None, table=None):
    """Gathers the per-rank tiles (torch tensors [tile_rows, W, 4] f32, same shape on every
    rank) on rank 0 and returns the de-interleaved frame [height, W, 4] there (None elsewhere).

    CUDA tensors: one RCCL gather, then the de-interleave HIP kernel of `node`
    (brt_deinterleave_device) -- with `rccl` (an RcclGather) both inside the library, one call (brt_gather_rccl); without it
    the gather is torch.distributed's.  CPU tensors (gloo, used by the world_size-2 tests): one gloo
    gather, then an index copy -- the row mapping under test is the same.

    Stream ordering on the GPU path (everything is enqueued on torch's CURRENT stream):
      * `tile` must be complete, or be produced by work already enqueued on the current stream
        (render_part_device(stream=None) is synchronous; with stream=<current stream> it is ordered);
      * dist.gather makes the current stream wait for RCCL's own stream, and the de-interleave kernel is
        launched on that same current stream (FLAG_CALLER_STREAM: also when it is the default stream, handle
        0), so it runs behind the gather;
      * the host is NOT blocked: before `tile` is overwritten (next frame) or the returned frame is read
        from another stream, synchronise the current stream -- `end_of_frame()` does that."""
    import torch
    import torch.distributed as dist

    width = tile.shape[1]
    if rccl is not None and tile.is_cuda:
        # ONE call: ncclGather of the tiles to rank 0 + the de-interleave kernel, both on torch's current stream
        stream = int(torch.cuda.current_stream().cuda_stream)
        tiles = torch.empty((world,) + tuple(tile.shape), dtype=tile.dtype, device=tile.device) if rank == 0 else None
        frame = torch.empty((height, width, 4), dtype=torch.float32, device=tile.device) if rank == 0 else None
        node.gather_rccl(rccl.comm, rank, world, tile.data_ptr(), tiles.data_ptr() if rank == 0 else 0, width, height,
                         frame.data_ptr() if rank == 0 else 0, stream=stream)
        if tiles is not None:
            tiles.record_stream(torch.cuda.current_stream())
        return frame
    if world == 1:
        tiles = tile.unsqueeze(0)
    else:
        # receive straight into one [world, rows, W, 4] buffer (the layout the de-interleave kernel reads)
        tiles = torch.empty((world,) + tuple(tile.shape), dtype=tile.dtype, device=tile.device) if rank == 0 else None
        dist.gather(tile, list(tiles.unbind(0)) if rank == 0 else None, dst=0, group=group)
        if rank != 0:
            return None
    if tiles.is_cuda:
        if node is None:
            raise RuntimeError("gather_frame on CUDA tensors needs the RayTracingNode (de-interleave kernel)")
        frame = torch.empty((height, width, 4), dtype=torch.float32, device=tiles.device)
        node.deinterleave_device(tiles.data_ptr(), world, width, height, frame.data_ptr(),
                                 stream=int(torch.cuda.current_stream().cuda_stream))
        # `tiles` is freed by Python when this function returns; the caching allocator only hands the block to
        # later work on the SAME stream, which is ordered behind the kernel that reads it
        return frame
    frame = torch.empty((height, width, 4), dtype=torch.float32)
    for p in range(world):
        rows = frame_rows_of_part(height, p, world, table)      # (table: the strip table every rank rendered with; the GPU paths take it from the context)
        valid = rows >= 0
        frame[torch.from_numpy(rows[valid])] = tiles[p][torch.from_numpy(np.flatnonzero(valid))]
    return frame


def end_of_frame(tile) -> None:
    """Blocks the host until everything enqueued on torch's current stream has finished: rank 0's gather
    + de-interleave, and on the other ranks the RCCL send that still reads `tile`.  Call it before the
    next frame is rendered into the same `tile` (the trace kernel runs on the context's own stream, which
    is not ordered with torch's)."""
    if tile.is_cuda:
        import torch
        torch.cuda.current_stream().synchronize()
