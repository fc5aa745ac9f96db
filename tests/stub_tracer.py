"""Stand-in for the GPU tracer in bench.py's CPU test (BRT_BENCH_TRACER=stub_tracer:make): the same
node interface over CPU tensors.  It traces nothing: a rank's tile row is filled with a pure function
of its FRAME row, so that the gathered frame can be checked; ray counts are made up (one per pixel
sample).  Test infrastructure only."""
import ctypes as C

import numpy as np

import bevyray_amd as brt
from bevyray_amd.parallel import frame_rows_of_part


class _Node:
    def __init__(self, rank, world):
        self.rank, self.world = rank, world

    def write_buffers(self, buffers):
        self.n_models = len(buffers.models)

    def render_part_device(self, level, camera, window, width, height, part, n_parts, d_out_tile, flags=0, **_):
        rows = frame_rows_of_part(height, part, n_parts)
        tile = np.ctypeslib.as_array((C.c_float * (len(rows) * width * 4)).from_address(d_out_tile)).reshape(len(rows), width, 4)
        tile[:] = np.where(rows >= 0, rows, -1).astype(np.float32)[:, None, None]
        px = int((rows >= 0).sum()) * width
        spp = int(camera["sample_count"][0])
        return {"rays": px * spp, "node_pops": 0, "interior_visits": 0, "sphere_tests": 0, "hits": 0, "paths": px * spp,
                "kernel_ms": 1.0 + part, "gather_ms": 0.0, "total_ms": 1.0, "lds_bytes": 0, "scene_in_lds": 0,
                "n_workgroups": 0, "threads_per_workgroup": 0}


class _Plugin:
    def __init__(self, rank, world):
        self.node = _Node(rank, world)

    def close(self):
        pass


def make(rank, world):
    return _Plugin(rank, world)
